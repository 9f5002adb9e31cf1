#!/bin/bash
# new sim loop: microbench extras, parity, sim-bound timings
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_c3
mkdir -p $OUT
cd $ROOT
export CCX_PACE_CACHE=$OUT/pace_cache.json
true

timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1 || { tail -40 $OUT/pytest.txt; exit 1; }
tail -3 $OUT/pytest.txt
for E in 4096 2048 1024; do timeout -k 10 120 python3 profiles/scratch/sim_only.py $E >> $OUT/sim_only.txt 2>&1; done
for E in 4096 2048; do timeout -k 10 120 python3 profiles/scratch/sim_only.py $E hand2=0 >> $OUT/sim_only.txt 2>&1; done
cat $OUT/sim_only.txt | grep -v amdgpu.ids
timeout -k 10 200 python3 bench.py --no-cpu-baseline > $OUT/c2.json 2> $OUT/c2.err
timeout -k 10 120 python3 bench.py --no-cpu-baseline --compact-obs > $OUT/c2_compact.json 2>> $OUT/c2.err
timeout -k 10 120 python3 bench.py --no-cpu-baseline --envs-per-gpu 2048 > $OUT/c2_2048.json 2>> $OUT/c2.err
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], f"{d['value']:.4g}", f"frac {d['roofline']['frac']:.3f}", f"ms/launch {d['roofline']['kernel_ms_per_launch']:.4f}", d.get("secondary"))
    except Exception as e:
        print(f, "ERR", e)
PY
