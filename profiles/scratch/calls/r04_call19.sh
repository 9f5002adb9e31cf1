#!/bin/bash
# round 4, call 19: do sim waves share their SIMD with writer waves?  rollouts without rows at 4096 / 1024 envs over (writers, tiles per workgroup)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c19
mkdir -p $OUT
cd $ROOT
B="timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 20"
for rep in 1 2; do
 for E in 4096 1024; do
  for wt in "0 0" "2 1" "2 2" "3 1" "3 2" "1 1" "1 2" "1 4" "4 1"; do
    set -- $wt
    $B --no-obs --envs-per-gpu $E --writers $1 --wpb $2 > $OUT/noobs_E${E}_w$1_t$2_$rep.json 2>> $OUT/err.txt || echo "fail $E $wt"
  done
 done
done
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable"); continue
    ls = d["config"]["launch_shape"]
    print(f.split("/")[-1], "us/env-step %.4f" % (d["roofline"]["kernel_ms_per_launch"] * 1e3 / d["config"]["steps_per_launch"]), (ls["lanes_per_wave"], ls["writers_per_tile"], ls["waves_per_block"], ls["num_blocks"]))
PY
