#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_c29
mkdir -p $OUT
cd $ROOT
B="timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 30"
for E in 1280 1536 1792; do
  for LW in "64 3" "64 4" "32 3" "32 4"; do
    set -- $LW
    $B --envs-per-gpu $E --lanes $1 --writers $2 > $OUT/e${E}_l$1_w$2.json 2>> $OUT/err.txt || echo "fail $E $LW"
  done
done
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], f"{d['value']:.4g}", f"frac {d['roofline']['frac']:.3f}", "us/step %.4f" % (d['roofline']['kernel_ms_per_launch']*1e3/d['config']['steps_per_launch']))
PY
