#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_c39
mkdir -p $OUT
cd $ROOT
B="timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-secondary --pool 512"
for P in 9300 9600 9900 10200; do
  $B --workload c3 --chunk 37 --steps 40 --warmup 80 --pace $P > $OUT/c3_k37_p$P.json 2>> $OUT/err.txt
  $B --workload c3 --chunk 100 --steps 40 --warmup 40 --pace $P > $OUT/c3_k100_p$P.json 2>> $OUT/err.txt
done
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], f"{d['value']:.4g}", f"frac {d['roofline']['frac']:.3f}", "max/median %.3f" % d['roofline']['kernel_ms_max_over_median'])
PY
