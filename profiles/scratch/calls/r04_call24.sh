#!/bin/bash
# round 4, call 24: why do batches of several rounds lose with MORE steps per launch? (steps per launch x pacing x shape)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c24
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python3 profiles/scratch/multi_round.py $OUT/multi_round.json > $OUT/multi_round.txt 2>&1 || { tail -20 $OUT/multi_round.txt; exit 1; }
cat $OUT/multi_round.txt
