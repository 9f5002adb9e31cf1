#!/bin/bash
# round 4, call 81: the new shape-parity tests
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c81
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_round4_shapes.py -m gpu -q > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt; grep -n "^FAILED\|^ERROR\|Error\|assert " $OUT/pytest.txt | cut -c1-300 | head -20
