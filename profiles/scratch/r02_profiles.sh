#!/bin/bash
# round-2 rocprofv3 evidence for all four bench workloads, one gpurun call
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
export CCX_PACE_MEMORY=0
for W in "c2 random" "c3 random" "c5_50 greedy" "c5_64 greedy"; do
  set -- $W
  bash profiles/collect_workload.sh r02 $1 $2 2>&1 | grep -v amdgpu.ids
done
python3 bench.py > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err
tail -c 600 gpurun_out/r02_bench_default.json
