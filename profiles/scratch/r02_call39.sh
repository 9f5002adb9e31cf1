#!/bin/bash
# where does a collapse start (lag of 16 tiles behind their schedule, step by step), and does a LOCAL slip by whole
# step periods (phase among the tiles kept) turn it into a blip?
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
D=collectivecrossing_amd/csrc/_diag
for L in lag slip3 slip8; do
  for P in 670 690; do
    echo "===== $L pace $P"
    CCX_DIAG_LIB=$D/libccx_$L.so timeout -k 10 120 python3 profiles/scratch/lag_trace.py $P 40 2>&1 | grep -v amdgpu.ids > gpurun_out/lag_${L}_$P.txt
    head -2 gpurun_out/lag_${L}_$P.txt
  done
done
