#!/bin/bash
# round 4, call 45: small lane groups -- LDS occupancy tables vs all-pairs compares (tunable occ_tables)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c45
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python3 profiles/scratch/occ_off.py 2>&1 | grep -v amdgpu | tee $OUT/occ_off.txt
