#!/bin/bash
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_call2
mkdir -p $OUT
cd $ROOT
export CCX_PACE_MEMORY=0
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
tail -3 $OUT/pytest.log
timeout -k 10 600 python3 profiles/scratch/sweep_knobs.py c3,c5_64,c2 > $OUT/sweep.txt 2>&1 || { tail -20 $OUT/sweep.txt; exit 1; }
cat $OUT/sweep.txt
