#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_c33
mkdir -p $OUT
cd $ROOT
B="timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 30"
for E in 1024 1536 2048 2560 3072; do
  for W in 3 4 5 6; do
    $B --envs-per-gpu $E --writers $W > $OUT/e${E}_w${W}.json 2>> $OUT/err.txt || echo "fail $E $W"
  done
done
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    s = d['config']['launch_shape']
    print(f.split("/")[-1], f"{d['value']:.4g}", f"frac {d['roofline']['frac']:.3f}", "us/step %.4f" % (d['roofline']['kernel_ms_per_launch']*1e3/d['config']['steps_per_launch']), s['lanes_per_wave'], s['writers_per_tile'], s['num_blocks'], d['config'].get('step_pace_ns'))
PY
