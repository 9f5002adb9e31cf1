#!/bin/bash
# round 4, call 28: translation prefetch of the observation rows (tunable row_prefetch) on outputs beyond the ~4 GB reach
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c28
mkdir -p $OUT
cd $ROOT
timeout -k 10 300 python3 -m pytest tests/test_gpu_round2.py -m gpu -q -x > $OUT/pytest.txt 2>&1 || { tail -30 $OUT/pytest.txt; exit 1; }
tail -2 $OUT/pytest.txt
timeout -k 10 600 python3 - > $OUT/prefetch.txt 2>&1 <<'PY' || { tail -20 $OUT/prefetch.txt; exit 1; }
import sys
sys.path.insert(0, "profiles/scratch")
import multi_round as m
for E, K in ((4096, 500), (4096, 1000), (4096, 2000), (16384, 256), (65536, 64), (65536, 128)):
    for name, prep in (("off", None), ("prefetch", lambda e: e.set_tunable("row_prefetch", 1)),
                       ("prefetch_pace_off", lambda e: (e.set_tunable("row_prefetch", 1), e.set_step_pace(-1)))):
        r = m.measure(E, K, prep, settle=24, timed=8)
        gb = r["ms"] * 1e-3 * r["frac"] * 8e12 / 1e9
        print(f"{E:6d} x {K:4d} {gb:5.1f} GB {name:18s} frac {r['frac']:.3f} (best {r['frac_best']:.3f}) {r['ms']:.3f} ms pace {r['pace_ns']:.0f} shape {r['shape']}", flush=True)
PY
cat $OUT/prefetch.txt
