#!/bin/bash
# round 4, call 88: the GPU suite twice more on a fresh box (flaky timing tests?)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c88
mkdir -p $OUT
cd $ROOT
for i in 1 2; do
  timeout -k 10 900 python3 -m pytest tests -m gpu -q -x > $OUT/pytest_$i.txt 2>&1; tail -2 $OUT/pytest_$i.txt; grep -n "^FAILED\|^ERROR" $OUT/pytest_$i.txt | head -5
done
