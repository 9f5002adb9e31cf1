#!/bin/bash
# round 4, call 21: the sweep once more with 3 / 4 writers among the candidates of the no-rows modes
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c21
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python3 profiles/scratch/shape_sweep.py $OUT/shape_sweep.json > $OUT/shape_sweep.txt 2>&1 || { tail -20 $OUT/shape_sweep.txt; exit 1; }
tail -3 $OUT/shape_sweep.txt
