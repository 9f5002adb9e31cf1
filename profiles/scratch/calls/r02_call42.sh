#!/bin/bash
# s_setprio(3) on the sim wave, A/B in one call: sim chain alone at 4096 / 2048 / 1024 envs, the bench workloads
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export CCX_PACE_MEMORY=0
for rep in 1 2 3; do
for L in "" collectivecrossing_amd/csrc/_diag/libccx_prio.so; do
  for E in 4096 2048 1024; do
    CCX_DIAG_LIB=$L timeout -k 10 100 python3 profiles/scratch/sim_only.py $E 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
done
for rep in 1 2; do
for L in "" collectivecrossing_amd/csrc/_diag/libccx_prio.so; do
  echo "== lib ${L:-shipped}"
  CCX_DIAG_LIB=$L timeout -k 10 300 python3 profiles/scratch/sweep_knobs.py c2,c3,c5_50,c5_64 250 60 30 '[{}]' 2>&1 | grep -v amdgpu.ids | grep rep1 | cut -c1-150
done
done
