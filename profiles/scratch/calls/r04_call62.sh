#!/bin/bash
# round 4, call 63: the same with the 100 x 100 case fixed (tables that do not fit are not counted)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c63
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py tests/test_gpu_shape_guard.py tests/test_gpu_large_grid_policy.py tests/test_gpu_policy_stream.py tests/test_gpu_step_kernel.py -m gpu -q > $OUT/pytest.txt 2>&1; tail -4 $OUT/pytest.txt; grep -n "^FAILED\|^ERROR\|AssertionError" $OUT/pytest.txt | cut -c1-600 | head
timeout -k 10 600 python3 profiles/scratch/big_grid_tpb.py 2>&1 | grep -v amdgpu | tee $OUT/big_grid_tpb.txt | cut -c1-420
