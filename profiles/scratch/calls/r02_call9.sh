#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export CCX_PACE_MEMORY=0
python3 profiles/scratch/tstamps.py 2>&1 | grep -v amdgpu.ids
python3 profiles/scratch/stepwise.py 2>&1 | grep -v amdgpu.ids
mkdir -p gpurun_out/r02_call9 && cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02_call9/kt -o kt -- python3 $GRAFT_REPO_ROOT/profiles/scratch/stepwise.py > /dev/null 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$GRAFT_REPO_ROOT/gpurun_out/r02_call9/kt/**/kt_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r["Name"][:90], r["Calls"], r["AverageNs"], r["MinNs"])
PY
