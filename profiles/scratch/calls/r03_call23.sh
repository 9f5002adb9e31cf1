#!/bin/bash
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_c23
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o k1 -- python3 $ROOT/profiles/scratch/step_k1.py 4096 > $OUT/prof.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/prof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:100], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
for f in glob.glob("$OUT/prof/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "rollout_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rows, rows[1:])]
    durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
    import statistics as st
    print("rollout kernels", len(rows), "median dur", st.median(durs), "median gap", st.median(gaps))
    # the graph part: last 2000 launches
    print("last 2000: median dur", st.median(durs[-2000:]), "median gap", st.median(gaps[-2000:]))
PY
