#!/bin/bash
# round 4, call 48: balanced rounds only for two-round grids with a thin second round -- tests, table, full suite later
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c48
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_round2.py -m gpu -q > $OUT/pytest.txt 2>&1; tail -4 $OUT/pytest.txt; grep -n "^FAILED\|^ERROR\|AssertionError: (" $OUT/pytest.txt | cut -c1-600 | head
timeout -k 10 400 python3 profiles/scratch/balanced.py c2 2>&1 | grep -v amdgpu | tee $OUT/balanced_c2.txt
timeout -k 10 300 python3 profiles/scratch/balanced.py c3 2>&1 | grep -v amdgpu | tee $OUT/balanced_c3.txt
