#!/bin/bash
# round 4, call 3: the team-structured short-launch kernel -- parity, stamps, launch-shape A/B; rllib adapter tests + host cost
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c03
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_step_kernel.py tests/test_gpu_rllib.py -x -q > $OUT/pytest_step.txt 2>&1 || { tail -60 $OUT/pytest_step.txt; exit 1; }
tail -2 $OUT/pytest_step.txt
timeout -k 10 120 python3 profiles/scratch/step_tstamps.py 4096 > $OUT/tstamps.txt 2>&1 || { tail -5 $OUT/tstamps.txt; exit 1; }
cat $OUT/tstamps.txt
timeout -k 10 300 python3 profiles/scratch/step_ab.py 4096 64 > $OUT/step_ab.txt 2>&1 || { tail -5 $OUT/step_ab.txt; exit 1; }
cat $OUT/step_ab.txt
timeout -k 10 200 python3 profiles/scratch/step_ab.py c3 4096 > $OUT/step_ab_c3.txt 2>&1 || { tail -5 $OUT/step_ab_c3.txt; exit 1; }
cat $OUT/step_ab_c3.txt
timeout -k 10 200 python3 profiles/scratch/rllib_host_cost.py > $OUT/rllib_host_cost.txt 2>&1 || { tail -5 $OUT/rllib_host_cost.txt; exit 1; }
cat $OUT/rllib_host_cost.txt
