#!/bin/bash
# is it the PAUSE at a kernel boundary that lets short launches run faster?  schedule gaps inside long launches
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export CCX_PACE_MEMORY=0
D=collectivecrossing_amd/csrc/_diag
for EK in "16384 250" "32768 125" "32768 250" "65536 125"; do
  set -- $EK
  for L in "" $D/libccx_gap125.so $D/libccx_gap60.so; do
    echo "E $1 K $2 lib ${L:-shipped}: $(CCX_DIAG_LIB=$L CCX_SWEEP_E=$1 timeout -k 10 300 python3 profiles/scratch/sweep_knobs.py c2 $2 60 20 '[{}]' 2>&1 | grep -v amdgpu.ids | grep rep1 | sed 's/  */ /g' | cut -c12-110)"
  done
done
