#!/bin/bash
# soak with the launch-shape draws + s_setprio(3) on the sim wave vs the shipped build (A/B inside one call)
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
bash profiles/scratch/r02_soak.sh 5 8000
export CCX_PACE_MEMORY=0
for rep in 1 2; do
for L in "" collectivecrossing_amd/csrc/_diag/libccx_prio.so; do
  echo "== lib ${L:-shipped}"
  CCX_DIAG_LIB=$L timeout -k 10 300 python3 profiles/scratch/sweep_knobs.py c2,c3,c5_64 250 60 30 '[{}]' 2>&1 | grep -v amdgpu.ids | grep rep1 | cut -c1-200
  CCX_DIAG_LIB=$L timeout -k 10 100 python3 profiles/scratch/sim_only.py 2>&1 | grep -v amdgpu.ids | tail -6
done
done
