#!/bin/bash
# round 4, call 86: 2 / 5 agents over large batches once more (after the tables-vs-rounds rule)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c86
mkdir -p $OUT
cd $ROOT
timeout -k 10 400 python3 - 2>&1 <<'PY' | grep -v "amdgpu\|arn\|alloc_rollout" | tee $OUT/rows_2_5.txt
import sys
sys.path.insert(0, "profiles/scratch")
import cliff_scan, cliff_scan2
Es = [e for e in sorted({int(round(256 * 1.125 ** k / 16) * 16) for k in range(34, 48)}) if e <= 70000]
for N in (2, 5, 3):
    cfg = cliff_scan2.config_for(N) if N != 3 else __import__("shape_sweep").config_for(3)
    for mode in ("rows", "noobs"):
        rows = [cliff_scan.measure(cfg, E, N, mode) for E in Es]
        print(f"N={N} {mode}: " + " ".join(f"{r['E']}:{(r['frac'] if r['frac'] else r['us_per_env_step']):.3f}" for r in rows), flush=True)
        print("   shapes: " + " ".join(f"{r['E']}:{tuple(r['shape'][1:])}" for r in rows[::3]), flush=True)
PY
