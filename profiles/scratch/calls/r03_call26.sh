#!/bin/bash
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_c26
mkdir -p $OUT
cd $ROOT
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1 || { tail -60 $OUT/pytest.txt; exit 1; }
tail -2 $OUT/pytest.txt
timeout -k 10 120 python3 profiles/scratch/tstamps.py 2>&1 | grep -v amdgpu.ids > $OUT/tstamps.txt; cat $OUT/tstamps.txt
for E in 4096 64; do timeout -k 10 120 python3 profiles/scratch/step_k1.py $E >> $OUT/step_k1.txt 2>&1; done; grep -v amdgpu.ids $OUT/step_k1.txt
for E in 4096 2048 1024; do timeout -k 10 120 python3 profiles/scratch/sim_only.py $E >> $OUT/sim_only.txt 2>&1; done; grep -v amdgpu.ids $OUT/sim_only.txt
B="timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-secondary"
timeout -k 10 200 python3 bench.py --no-cpu-baseline > $OUT/c2.json 2>> $OUT/err.txt
$B --compact-obs > $OUT/c2_compact.json 2>> $OUT/err.txt
$B --envs-per-gpu 2048 > $OUT/c2_2048.json 2>> $OUT/err.txt
$B --envs-per-gpu 1024 > $OUT/c2_1024.json 2>> $OUT/err.txt
$B --workload c5_64 --policy greedy --chunk 100 --steps 24 --warmup 24 --pool 512 > $OUT/c5_64.json 2>> $OUT/err.txt
$B --workload c5_50 --policy greedy --chunk 100 --steps 24 --warmup 24 --pool 512 > $OUT/c5_50.json 2>> $OUT/err.txt
$B --workload c3 --chunk 100 --steps 24 --warmup 24 --pool 512 > $OUT/c3.json 2>> $OUT/err.txt
$B --workload c3 --compact-obs --chunk 100 --steps 24 --warmup 24 --pool 512 > $OUT/c3_compact.json 2>> $OUT/err.txt
$B --workload c5_64 --policy greedy --compact-obs --chunk 100 --steps 24 --warmup 24 --pool 512 > $OUT/c5_64_compact.json 2>> $OUT/err.txt
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], f"{d['value']:.4g}", f"frac {d['roofline']['frac']:.3f}", "us/step %.4f" % (d['roofline']['kernel_ms_per_launch']*1e3/d['config']['steps_per_launch']), d.get('secondary'))
    except Exception as e:
        print(f, "ERR", e)
PY
