#!/bin/bash
# Extra PMC evidence for the rollout kernel (one derived metric per rocprofv3 pass; --pmc only, no traces):
#   bash profiles/collect_extra.sh r01      (run through gpurun from the repo root)
set -e
R=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_extra
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in LdsBankConflict LdsUtil VALUBusy SALUBusy MemUnitStalled MeanOccupancyPerCU; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/$C -o p -- python3 $ROOT/bench.py --no-cpu-baseline --steps 4 --warmup 80 > $OUT/$C.json 2> $OUT/$C.err || { tail -3 $OUT/$C.err; exit 1; }
done
python3 - <<PY
import csv, glob, os
out = open("$OUT/${R}_pmc_extra.csv", "w")
out.write("counter,dispatches,mean_over_the_last_4_rollout_dispatches\n")
for f in sorted(glob.glob("$OUT/*/p_counter_collection.csv")):
    rows = [r for r in csv.DictReader(open(f)) if "rollout_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    vals = [float(r["Counter_Value"]) for r in rows]
    name = rows[0]["Counter_Name"] if rows else os.path.basename(os.path.dirname(f))
    out.write(f"{name},{len(vals)},{sum(vals[-4:]) / max(1, len(vals[-4:])):.4f}\n")
out.close()
print(open("$OUT/${R}_pmc_extra.csv").read())
PY
