#!/bin/bash
# round 4, call 79: singles where pairs cost a round -- N = 4 / 3 / 8 checks, guards, round-2 tests
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c79
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_round2.py tests/test_gpu_shape_guard.py -m gpu -q > $OUT/pytest.txt 2>&1; tail -2 $OUT/pytest.txt; grep -n "^FAILED\|^ERROR\|AssertionError" $OUT/pytest.txt | cut -c1-400 | head
for n in 4 3 8; do
  timeout -k 10 300 python3 profiles/scratch/noobs_scan.py 16384,17776,19000,20480,24576 $n rows 2>&1 | grep -v "amdgpu\|arn" | tee -a $OUT/rows_438.txt
done
timeout -k 10 300 python3 profiles/scratch/noobs_scan.py 16384,17776,19000,20480 4 noobs 2>&1 | grep -v "amdgpu\|arn" | tee -a $OUT/rows_438.txt
