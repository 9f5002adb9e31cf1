#!/bin/bash
# chip-wide pauses of the write stream inside long launches (the controller's plan includes them this time)
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export CCX_PACE_MEMORY=0
D=collectivecrossing_amd/csrc/_diag
for K in 500 1000 2000; do
  for L in "" $D/libccx_gap500.so $D/libccx_gap250.so $D/libccx_gap500b.so; do
    echo "E 4096 K $K lib ${L:-shipped}: $(CCX_DIAG_LIB=$L timeout -k 10 300 python3 profiles/scratch/sweep_knobs.py c2 $K 40 16 '[{"split":0}]' 2>&1 | grep -v amdgpu.ids | grep rep1 | sed 's/  */ /g' | cut -c24-120)"
  done
done
