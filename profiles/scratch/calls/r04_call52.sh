#!/bin/bash
# round 4, call 52: balanced rounds only where the scaled schedule is feasible; no two-writer pairs beyond 8192 tiles -- scan 2 again (rows of 2 / 5 / 16 agents), round-2 tests
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c52
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_round2.py tests/test_gpu_shape_guard.py -m gpu -q > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt; grep -n "^FAILED\|^ERROR\|AssertionError: (" $OUT/pytest.txt | cut -c1-600 | head
timeout -k 10 600 python3 - > $OUT/rows_2_5_16.txt 2>&1 <<'PY'
import sys
sys.path.insert(0, "profiles/scratch")
import cliff_scan, cliff_scan2
Es = [e for e in sorted({int(round(256 * 1.125 ** k / 16) * 16) for k in range(30, 48)}) if e <= 70000]
for N in (2, 5, 16, 12):
    cfg = cliff_scan2.config_for(N) if N != 12 else __import__("shape_sweep").config_for(12)
    rows = []
    for E in Es:
        if E * N * (6 + 4 * N) * 4 * 24 > 5.5e9:
            continue
        rows.append(cliff_scan.measure(cfg, E, N, "rows"))
    print(f"N={N} rows: " + " ".join(f"{r['E']}:{r['frac']:.3f}" for r in rows), flush=True)
    print("   shapes: " + " ".join(f"{r['E']}:{tuple(r['shape'][1:])}" for r in rows[::3]), flush=True)
PY
grep -v amdgpu $OUT/rows_2_5_16.txt
