#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export CCX_PACE_MEMORY=0
for W in c3 c5_64; do
timeout -k 10 200 python3 profiles/scratch/pace_trace.py $W 160 2>&1 | grep -v amdgpu.ids > gpurun_out/pace_trace_$W.txt
tail -1 gpurun_out/pace_trace_$W.txt
done
