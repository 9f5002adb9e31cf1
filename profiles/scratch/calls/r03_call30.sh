#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_c30
mkdir -p $OUT
cd $ROOT
B="timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-secondary"
$B --envs-per-gpu 8192 --chunk 250 > $OUT/e8192.json 2>> $OUT/err.txt
$B --envs-per-gpu 16384 --chunk 125 > $OUT/e16384.json 2>> $OUT/err.txt
$B --envs-per-gpu 32768 --chunk 64 > $OUT/e32768.json 2>> $OUT/err.txt
$B --envs-per-gpu 65536 --chunk 64 > $OUT/e65536.json 2>> $OUT/err.txt
$B --envs-per-gpu 32768 --chunk 128 > $OUT/e32768_k128.json 2>> $OUT/err.txt
timeout -k 10 100 python3 profiles/scratch/dict_env_latency.py > $OUT/dict.txt 2>&1
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    s = d['config']['launch_shape']
    print(f.split("/")[-1], f"{d['value']:.4g}", f"frac {d['roofline']['frac']:.3f}", "us/step %.4f" % (d['roofline']['kernel_ms_per_launch']*1e3/d['config']['steps_per_launch']), s['num_blocks'], s['resident_blocks'], s['waves_per_block'], s['writers_per_tile'])
PY
grep -v amdgpu $OUT/dict.txt | tail -6
