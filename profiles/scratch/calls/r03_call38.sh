#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_c38
mkdir -p $OUT
cd $ROOT
B="timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-secondary --pool 512"
for C in 100 37 18; do
  $B --workload c3 --chunk $C --steps 40 --warmup 80 > $OUT/c3_k$C.json 2>> $OUT/err.txt
  $B --workload c5_64 --policy greedy --chunk $C --steps 40 --warmup 80 > $OUT/c5_64_k$C.json 2>> $OUT/err.txt
done
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], f"{d['value']:.4g}", f"frac {d['roofline']['frac']:.3f}", "GB per launch %.2f" % (d['roofline']['bytes_per_launch']/1e9), d['config'].get('step_pace_ns'))
PY
