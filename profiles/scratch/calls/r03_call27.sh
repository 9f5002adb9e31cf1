#!/bin/bash
# small batches: tile size (lanes carrying agents per wave) x writers per tile, C2 geometry
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_c27
mkdir -p $OUT
cd $ROOT
B="timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 30"
for E in 1024 2048 512; do
  for L in 64 32 16; do
    for W in 1 2 3 4; do
      $B --envs-per-gpu $E --lanes $L --writers $W > $OUT/e${E}_l${L}_w${W}.json 2>> $OUT/err.txt || echo "fail $E $L $W"
    done
  done
done
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], f"{d['value']:.4g}", f"frac {d['roofline']['frac']:.3f}", "us/step %.4f" % (d['roofline']['kernel_ms_per_launch']*1e3/d['config']['steps_per_launch']), d['config']['launch_shape']['num_blocks'], d['config']['launch_shape']['waves_per_block'], d['config'].get('step_pace_ns'))
    except Exception as e:
        print(f, "ERR", e)
PY
