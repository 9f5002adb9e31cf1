#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_c35
mkdir -p $OUT
cd $ROOT
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1 || { tail -40 $OUT/pytest.txt; exit 1; }
tail -2 $OUT/pytest.txt
B="timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 30"
for rep in 1 2; do
for E in 512 1024 2048 3072; do
  for P in 0 1; do
    $B --envs-per-gpu $E --tunable pair_rows=$P > $OUT/e${E}_p${P}_$rep.json 2>> $OUT/err.txt || echo "fail $E $P"
  done
done
done
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], f"{d['value']:.4g}", f"frac {d['roofline']['frac']:.3f}", "us/step %.4f" % (d['roofline']['kernel_ms_per_launch']*1e3/d['config']['steps_per_launch']), d['config']['launch_shape']['writers_per_tile'])
PY
