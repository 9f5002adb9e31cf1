#!/bin/bash
# round 4, call 51: step tiles halved only below CUs / 2 tiles -- step tests, the scan again
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c51
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_step_kernel.py tests/test_gpu_env_api.py tests/test_gpu_vector.py tests/test_gpu_rllib.py -m gpu -q -x > $OUT/pytest.txt 2>&1 || { tail -30 $OUT/pytest.txt; exit 1; }
tail -2 $OUT/pytest.txt
timeout -k 10 400 python3 profiles/scratch/step_scan.py 3 2>&1 | grep -v amdgpu | tee $OUT/step_scan_n3.txt | cut -c1-60
timeout -k 10 400 python3 profiles/scratch/step_scan.py 8 2>&1 | grep -v amdgpu | tee $OUT/step_scan_n8.txt | cut -c1-60
timeout -k 10 400 python3 profiles/scratch/step_scan.py 32 2>&1 | grep -v amdgpu | tee $OUT/step_scan_n32.txt
