#!/bin/bash
# round 4, call 2: the short-launch kernel -- parity, then where its time goes
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c02
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_step_kernel.py -x -q > $OUT/pytest_step.txt 2>&1 || { tail -60 $OUT/pytest_step.txt; exit 1; }
tail -2 $OUT/pytest_step.txt
timeout -k 10 120 python3 profiles/scratch/step_k1.py 4096 > $OUT/step_k1.txt 2>&1 || { tail -5 $OUT/step_k1.txt; exit 1; }
cat $OUT/step_k1.txt
timeout -k 10 120 python3 profiles/scratch/step_k1.py 64 > $OUT/step_k1_64.txt 2>&1 || { tail -5 $OUT/step_k1_64.txt; exit 1; }
cat $OUT/step_k1_64.txt
timeout -k 10 120 python3 profiles/scratch/step_tstamps.py 4096 > $OUT/tstamps.txt 2>&1 || { tail -5 $OUT/tstamps.txt; exit 1; }
cat $OUT/tstamps.txt
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1 || { tail -40 $OUT/pytest.txt; exit 1; }
tail -2 $OUT/pytest.txt
