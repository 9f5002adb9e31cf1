#!/bin/bash
# round 4, call 87: 2 agents at 36 032 .. 64 928 envs: shapes and candidates
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c87
mkdir -p $OUT
cd $ROOT
timeout -k 10 300 python3 - 2>&1 <<'PY' | grep -v "amdgpu\|arn\|alloc_rollout" | tee $OUT/n2.txt
import sys
sys.path.insert(0, "profiles/scratch")
import big_grid_scan as b, cliff_scan, cliff_scan2
cfg = cliff_scan2.config_for(2)
for E in (36032, 40528, 45600, 51296, 57712, 64928):
    res = {}
    for occ in (-1, 1, 0):
        b._occ[0] = occ
        r = cliff_scan.measure(cfg, E, 2, "rows")
        res[occ] = (round(r["frac"], 3), tuple(r["shape"]))
    print(f"N=2 rows E={E}: default {res[-1]}  tables {res[1]}  all-pairs {res[0]}", flush=True)
PY
timeout -k 10 300 python3 profiles/scratch/noobs_scan.py 45600,51296,64928 2 rows 2>&1 | grep -v "amdgpu\|arn" | tee -a $OUT/n2.txt
