#!/bin/bash
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_c28
mkdir -p $OUT
cd $ROOT
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1 || { tail -60 $OUT/pytest.txt; exit 1; }
tail -2 $OUT/pytest.txt
B="timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-secondary"
for E in 256 512 1024 1536 2048; do $B --envs-per-gpu $E > $OUT/c2_$E.json 2>> $OUT/err.txt; done
$B --envs-per-gpu 1024 --compact-obs > $OUT/c2_1024_compact.json 2>> $OUT/err.txt
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        s = d['config']['launch_shape']
        print(f.split("/")[-1], f"{d['value']:.4g}", f"frac {d['roofline']['frac']:.3f}", "us/step %.4f" % (d['roofline']['kernel_ms_per_launch']*1e3/d['config']['steps_per_launch']), s['lanes_per_wave'], s['writers_per_tile'], s['num_blocks'])
    except Exception as e:
        print(f, "ERR", e)
PY
