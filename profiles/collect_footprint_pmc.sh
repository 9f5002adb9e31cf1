#!/bin/bash
# why do rollouts that cycle through > 3 GB of output memory sustain 3-4 % less?  Hardware counters of
# the C2 rollout kernel writing ONE 2.65 GB trajectory buffer vs FOUR used round-robin (bench.py --buffers), one rocprofv3
# --pmc pass per counter group (the program itself follows `--`), plus the un-profiled rates of both in the same call.
#   bash profiles/collect_footprint_pmc.sh        ->  gpurun_out/footprint/r03_footprint_pmc.json
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/footprint
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export CCX_BENCH_NO_SETTLE=1
ARGS="--no-cpu-baseline --no-secondary"
for B in 1 4; do
  python3 $ROOT/bench.py $ARGS --buffers $B --steps 40 --warmup 80 > $OUT/rate_b$B.json 2>> $OUT/err.txt
done
G1="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum"
G2="TCC_TAG_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum"
G3="TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum"
G4="TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum"
G5="TCC_WRITE_REQ_LATENCY_sum TCC_WRITE_REQ_sum GRBM_GUI_ACTIVE"
n=0
for G in "$G1" "$G2" "$G3" "$G4" "$G5"; do
  n=$((n+1))
  for B in 1 4; do
    rocprofv3 --pmc $G --output-format csv -d $OUT/g${n}_b$B -o pmc -- python3 $ROOT/bench.py $ARGS --buffers $B --steps 8 --warmup 60 \
      > $OUT/bench_g${n}_b$B.json 2> $OUT/g${n}_b$B.err || { echo "group $n buffers $B failed:"; tail -3 $OUT/g${n}_b$B.err; }
  done
done
python3 - <<PY
import csv, glob, json
out = "$OUT"
res = {"what": "C2 rollout kernel (4096 envs x 8 agents, 500 steps per launch, 2.654 GB of outputs per launch), mean per launch over the LAST 8 "
               "dispatches of each rocprofv3 --pmc pass; buffers = trajectory buffers the launches cycle through (bench.py --buffers)",
       "unprofiled": {}, "counters": {}}
for b in (1, 4):
    d = json.loads(open(f"{out}/rate_b{b}.json").read().strip().splitlines()[-1])
    res["unprofiled"][f"buffers_{b}"] = {"frac": d["roofline"]["frac"], "kernel_ms_per_launch": d["roofline"]["kernel_ms_per_launch"],
                                         "step_pace_ns": d["config"]["step_pace_ns"], "GB_cycled": 2.654 * b}
for path in sorted(glob.glob(f"{out}/g*_b*/**/pmc_counter_collection.csv", recursive=True)):
    b = int(path.split("_b")[1].split("/")[0])
    rows = [r for r in csv.DictReader(open(path)) if "rollout_kernel" in r["Kernel_Name"]]
    by = {}
    for r in rows:
        by.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for name, vals in by.items():
        res["counters"].setdefault(name, {})[f"buffers_{b}"] = sum(vals[-8:]) / len(vals[-8:])
for name, v in res["counters"].items():
    if "buffers_1" in v and "buffers_4" in v and v["buffers_1"]:
        v["ratio_4_over_1"] = v["buffers_4"] / v["buffers_1"]
json.dump(res, open(f"{out}/r03_footprint_pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
