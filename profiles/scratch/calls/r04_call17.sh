#!/bin/bash
# round 4, call 17: rocprofv3 evidence for the other BASELINE workloads, then the hypothesis soak with the round's new draws
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c17
mkdir -p $OUT
cd $ROOT
for wp in "c3 random" "c5_50 greedy" "c5_64 greedy"; do
  set -- $wp
  timeout -k 10 300 bash profiles/collect_workload.sh r04 $1 $2 > $OUT/collect_$1.txt 2>&1; tail -3 $OUT/collect_$1.txt
done
timeout -k 10 1100 bash profiles/scratch/r04_soak.sh 8 10000
