#!/bin/bash
# round 4, call 70: C5 class (50 / 64 agents): dense default scan + candidates
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c70
mkdir -p $OUT
cd $ROOT
timeout -k 10 500 python3 - 2>&1 <<'PY' | grep -v amdgpu | tee $OUT/c5_scan.txt
import sys
sys.path.insert(0, "profiles/scratch")
import cliff_scan, cliff_scan2
Es = [e for e in sorted({int(round(128 * 1.125 ** k / 4) * 4) for k in range(0, 40)}) if e <= 6000]
for N in (64, 50):
    cfg = cliff_scan2.config_for(N)
    for mode in ("rows", "noobs"):
        rows = []
        for E in Es:
            if mode == "rows" and E * N * (6 + 4 * N) * 4 * 24 > 5.5e9:
                continue
            rows.append(cliff_scan.measure(cfg, E, N, mode))
        print(f"N={N} {mode}: " + " ".join(f"{r['E']}:{(r['frac'] if r['frac'] else r['us_per_env_step']):.3f}" for r in rows), flush=True)
        print("   shapes: " + " ".join(f"{r['E']}:{tuple(r['shape'][1:])}" for r in rows[::3]), flush=True)
    rows = [cliff_scan2.other(cfg, E, N, "greedy") for E in Es if E * N * (6 + 4 * N) * 4 * 24 <= 5.5e9]
    print(f"N={N} greedy (us per env-step): " + " ".join(f"{r['E']}:{r['us_per_env_step']:.3f}" for r in rows), flush=True)
PY
for n in 64 50; do
  timeout -k 10 300 python3 profiles/scratch/noobs_scan.py 512,1024,1536,2048,3072 $n rows 2>&1 | grep -v amdgpu | tee -a $OUT/c5_cands.txt
  timeout -k 10 300 python3 profiles/scratch/noobs_scan.py 512,1024,2048,4096,8192 $n noobs 2>&1 | grep -v amdgpu | tee -a $OUT/c5_cands.txt
done
