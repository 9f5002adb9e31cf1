#!/bin/bash
# round 4, call 69: the two-writer class beyond 8192 tiles: one writer IN PAIRS -- the 20-agent tables again, 16 agents on 12 x 8, round-2 / guard tests
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c69
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_round2.py tests/test_gpu_shape_guard.py -m gpu -q > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt; grep -n "^FAILED\|^ERROR\|AssertionError" $OUT/pytest.txt | cut -c1-600 | head
timeout -k 10 800 python3 profiles/scratch/big_grid_tpb.py 40x30x20 24x16x20 2>&1 | grep -v amdgpu | tee $OUT/mid_grid_tpb.txt | cut -c1-200
timeout -k 10 300 python3 - 2>&1 <<'PY' | grep -v amdgpu | tee $OUT/n16.txt
import sys
sys.path.insert(0, "profiles/scratch")
import cliff_scan, cliff_scan2
cfg = cliff_scan2.config_for(16)
print("N=16 12x8 rows: " + " ".join(f"{E}:{cliff_scan.measure(cfg, E, 16, 'rows')['frac']:.3f}" for E in (16384, 28464, 32032, 36032, 45600, 65536)))
PY
