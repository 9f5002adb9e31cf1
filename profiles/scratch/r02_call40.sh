#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export CCX_PACE_MEMORY=0
timeout -k 10 200 python3 profiles/scratch/pace_trace.py c2 260 2>&1 | grep -v amdgpu.ids > gpurun_out/pace_trace_c2_260.txt
python3 - <<'PY'
import re, numpy as np
us = np.array([float(re.search(r":\s+([0-9.]+) us", l).group(1)) for l in open("gpurun_out/pace_trace_c2_260.txt") if l.startswith("launch")])
for a in range(0, 260, 52):
    w = us[a:a + 52]; med = np.median(us[100:])
    print(f"launches {a:3d}-{a+51:3d}: mean {w.mean():.4f} med {np.median(w):.4f} max {w.max():.4f} over1.05 {(w > 1.05 * med).sum()}  frac(mean) {1296 * 4096 / (w.mean() * 1e-6) / 8e12:.3f}")
PY
timeout -k 10 300 python3 -m pytest tests/test_gpu_round2.py -m gpu -q -k "outliers" -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -3
