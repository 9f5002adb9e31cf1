#!/bin/bash
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_c6
mkdir -p $OUT
cd $ROOT
export CCX_PACE_CACHE=$OUT/pace_cache.json
for W in 2 3 4 5; do timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-secondary --envs-per-gpu 2048 --writers $W > $OUT/c2_2048_w$W.json 2>> $OUT/err.txt; done
for W in 3 4; do timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-secondary --envs-per-gpu 1024 --writers $W > $OUT/c2_1024_w$W.json 2>> $OUT/err.txt; done
for W in 1 2; do timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-secondary --no-obs --writers $W > $OUT/c2_noobs_w$W.json 2>> $OUT/err.txt; done
timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-secondary --no-obs --only-obs > $OUT/c2_nothing.json 2>> $OUT/err.txt || true
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], f"{d['value']:.4g}", f"frac {d['roofline']['frac']:.3f}", f"ms/launch {d['roofline']['kernel_ms_per_launch']:.4f}", "us/step %.4f" % (d['config']['ms_per_env_step']*1e3), d['config']['launch_shape'])
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 $OUT/err.txt
