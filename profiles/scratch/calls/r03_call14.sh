#!/bin/bash
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_c14
mkdir -p $OUT
cd $ROOT
B="timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-secondary --envs-per-gpu 2048"
$B > $OUT/a_default.json 2>> $OUT/err.txt
$B --only-obs > $OUT/b_only_obs.json 2>> $OUT/err.txt
$B --no-obs > $OUT/c_no_obs.json 2>> $OUT/err.txt
$B --only-obs --writers 2 > $OUT/d_only_obs_w2.json 2>> $OUT/err.txt
$B --only-obs --writers 3 > $OUT/e_only_obs_w3.json 2>> $OUT/err.txt
$B --tunable writer_roles=0 > $OUT/f_roles0.json 2>> $OUT/err.txt
$B --lanes 32 > $OUT/g_lanes32.json 2>> $OUT/err.txt
$B --tunable hand2=0 > $OUT/h_barrier.json 2>> $OUT/err.txt
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], f"{d['value']:.4g}", "us/step %.4f" % (d['roofline']['kernel_ms_per_launch']*1e3/d['config']['steps_per_launch']), "GB/s %.0f" % d['roofline']['achieved'], d['config']['launch_shape']['writers_per_tile'], d['config']['launch_shape']['lanes_per_wave'])
    except Exception as e:
        print(f, "ERR", e)
PY
