#!/bin/bash
# round 4, call 1: the launch floor microbenchmarks + the full GPU suite on the re-split build
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c01
mkdir -p $OUT
cd $ROOT
timeout -k 10 120 profiles/scratch/launch_floor > $OUT/floor.txt 2>&1 || { tail -5 $OUT/floor.txt; exit 1; }
timeout -k 10 120 profiles/scratch/launch_floor_preload > $OUT/floor_preload.txt 2>&1 || { tail -5 $OUT/floor_preload.txt; exit 1; }
cat $OUT/floor.txt
echo ---- preload; cat $OUT/floor_preload.txt
timeout -k 10 120 python3 profiles/scratch/step_k1.py 4096 > $OUT/step_k1.txt 2>&1 || { tail -5 $OUT/step_k1.txt; exit 1; }
cat $OUT/step_k1.txt
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1 || { tail -40 $OUT/pytest.txt; exit 1; }
tail -2 $OUT/pytest.txt
