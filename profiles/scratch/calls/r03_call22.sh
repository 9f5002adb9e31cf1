#!/bin/bash
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_c22
mkdir -p $OUT
cd $ROOT
for E in 4096 1024 64; do timeout -k 10 120 python3 profiles/scratch/step_k1.py $E >> $OUT/step_k1.txt 2>&1; done
grep -v amdgpu.ids $OUT/step_k1.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/prof -o k1 -- python3 $ROOT/profiles/scratch/step_k1.py 4096 > $OUT/prof.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/prof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:90], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
