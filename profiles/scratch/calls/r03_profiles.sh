#!/bin/bash
# round-3 rocprofv3 evidence for all four bench workloads + the counters rocprofv3 offers for the footprint question
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out/r03_prof
for W in "c2 random" "c3 random" "c5_50 greedy" "c5_64 greedy"; do
  set -- $W
  bash profiles/collect_workload.sh r03 $1 $2 2>&1 | grep -v amdgpu.ids
done
python3 bench.py > gpurun_out/r03_prof/r03_bench_default.json 2> gpurun_out/r03_prof/bench_default.err
python3 bench.py --no-cpu-baseline --warmup 5 --steps 20 > gpurun_out/r03_prof/r03_bench_driver_flags.json 2>> gpurun_out/r03_prof/bench_default.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $ROOT/gpurun_out/r03_prof/list_avail.txt 2>&1 || true
grep -i -o "TCC_EA[A-Z0-9_]*\|TCC_TAG[A-Z0-9_]*\|UTCL2[A-Z0-9_]*\|TLB[A-Z0-9_]*\|TCP_UTCL1[A-Z0-9_]*\|TCC_[A-Z0-9_]*WR[A-Z0-9_]*" $ROOT/gpurun_out/r03_prof/list_avail.txt | sort -u | tr '\n' ' '
