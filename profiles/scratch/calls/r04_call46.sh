#!/bin/bash
# round 4, call 46: single-agent envs without tables when the tables force rounds; hypothesis draws of occ_tables; scan again
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c46
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py tests/test_gpu_shape_guard.py -m gpu -q > $OUT/pytest.txt 2>&1; tail -4 $OUT/pytest.txt; grep -n "^FAILED\|^ERROR\|AssertionError: (" $OUT/pytest.txt | cut -c1-600 | head
CCX_HYP_EXAMPLES=3000 timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k arbitrary_valid_configs -p no:cacheprovider 2>&1 | tail -1
timeout -k 10 600 python3 profiles/scratch/cliff_scan.py $OUT/cliff_scan.json 2>&1 | grep -v amdgpu | tee $OUT/cliff_scan.txt | cut -c1-600
