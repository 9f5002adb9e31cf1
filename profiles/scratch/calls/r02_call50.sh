#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export CCX_PACE_MEMORY=0
for E in 16384 32768 65536; do
CCX_SWEEP_E=$E CCX_TRACE_K=125 timeout -k 10 200 python3 profiles/scratch/pace_trace.py c2 120 2>&1 | grep -v amdgpu.ids > gpurun_out/pace_trace_E$E.txt
tail -1 gpurun_out/pace_trace_E$E.txt
done
