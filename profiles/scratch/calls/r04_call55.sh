#!/bin/bash
# round 4, call 55: the same after the no-rows rules (two-writer tiles never in pairs, one writer from 1281 tiles, pairs of one-writer tiles from 1500)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c55
mkdir -p $OUT
cd $ROOT
for n in 8 3 12 32; do
  timeout -k 10 300 python3 profiles/scratch/noobs_scan.py 512,1024,2048,3000,6000,8192,12288,16384,24576,32768,65536 $n noobs 2>&1 | grep -v amdgpu | tee -a $OUT/noobs_scan.txt
done
timeout -k 10 300 python3 profiles/scratch/noobs_scan.py 2500,3072,3840,4096,4328,6160 8 compact 2>&1 | grep -v amdgpu | tee -a $OUT/noobs_scan.txt
timeout -k 10 600 python3 -m pytest tests/test_gpu_shape_guard.py tests/test_gpu_round2.py -m gpu -q > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt; grep -n "^FAILED\|^ERROR\|AssertionError: (" $OUT/pytest.txt | cut -c1-600 | head
