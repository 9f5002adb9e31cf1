#!/bin/bash
# round-2 baseline of the round-1 kernel: all workloads + the latency-bound regimes, one gpurun call
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_base
mkdir -p $OUT
cd $ROOT
python3 bench.py > $OUT/c2.json 2> $OUT/c2.err
python3 bench.py --no-cpu-baseline --no-obs > $OUT/c2_noobs.json 2>> $OUT/c2.err
python3 bench.py --no-cpu-baseline --envs-per-gpu 2048 > $OUT/c2_2048.json 2>> $OUT/c2.err
python3 bench.py --no-cpu-baseline --warmup 5 --steps 20 > $OUT/c2_w5.json 2>> $OUT/c2.err
CCX_BENCH_NO_SETTLE=1 python3 bench.py --no-cpu-baseline --warmup 5 --steps 20 > $OUT/c2_w5_cold.json 2>> $OUT/c2.err
python3 profiles/scratch/stepwise.py > $OUT/stepwise.txt 2>&1
for W in "c3 random" "c5_50 greedy" "c5_64 greedy"; do
  set -- $W
  bash profiles/collect_workload.sh r02base $1 $2 > $OUT/collect_$1.txt 2>&1 || { tail -5 $OUT/collect_$1.txt; exit 1; }
done
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], f"{d['value']:.4g}", f"frac {d['roofline']['frac']:.3f}", f"ms/launch {d['roofline']['kernel_ms_per_launch']:.4f}")
    except Exception as e:
        print(f, "ERR", e)
PY
cat $OUT/stepwise.txt $OUT/collect_*.txt
