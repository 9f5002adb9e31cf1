#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export CCX_PACE_MEMORY=0
python3 profiles/scratch/pace_trace.py c2 300 2>&1 | grep -v amdgpu.ids > gpurun_out/r02_pace_trace_c2.txt
python3 - <<PY
rows = [l for l in open("gpurun_out/r02_pace_trace_c2.txt") if l.startswith("launch")]
us = [float(l.split()[2]) for l in rows]
import statistics as st
for a, b in ((0, 40), (40, 100), (100, 200), (200, 300)):
    seg = us[a:b]; med = st.median(seg)
    print(f"launches {a}-{b}: median {med:.4f} mean {st.mean(seg):.4f} max {max(seg):.4f} ({max(seg)/med:.3f}x) outliers>1.05x: {sum(1 for v in seg if v > 1.05*med)}")
for i in (39, 99, 199, 299): print(rows[i].strip())
PY
for rep in 1 2 3; do python3 bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench frac %.3f pace %.1f max/med %.3f' % (d['roofline']['frac'], d['config']['step_pace_ns'], d['roofline']['kernel_ms_max_over_median']))"; done
