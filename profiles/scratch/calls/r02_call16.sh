#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export CCX_PACE_MEMORY=0
python3 profiles/scratch/pace_trace.py c2 260 2>&1 | grep -v amdgpu.ids > gpurun_out/r02_pace_trace_c2.txt
python3 - <<PY
import re
rows = [l for l in open("gpurun_out/r02_pace_trace_c2.txt") if l.startswith("launch")]
us = [float(l.split()[2]) for l in rows]
import statistics as st
for a, b in ((0, 40), (40, 100), (100, 180), (180, 260)):
    seg = us[a:b]; med = st.median(seg)
    print(f"launches {a}-{b}: median {med:.4f} mean {st.mean(seg):.4f} max {max(seg):.4f} ({max(seg)/med:.3f}x) outliers>1.05x: {sum(1 for v in seg if v > 1.05*med)}")
print(rows[39].strip()); print(rows[99].strip()); print(rows[179].strip()); print(rows[-1].strip())
PY
