#!/bin/bash
# round 4, call 26: is the loss of long multi-round rollouts a matter of the OUTPUT SIZE alone? one-round batches with 2.7 ... 11 GB of rows
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c26
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python3 - > $OUT/size.txt 2>&1 <<'PY' || { tail -20 $OUT/size.txt; exit 1; }
import sys
sys.path.insert(0, "profiles/scratch")
import multi_round as m
for E, K in ((4096, 500), (4096, 1000), (4096, 2000), (8192, 250), (8192, 500), (8192, 1000), (16384, 128), (16384, 256), (16384, 512), (2048, 2000), (2048, 4000)):
    for name, prep in (("default", None), ("pace_off", lambda e: e.set_step_pace(-1))):
        r = m.measure(E, K, prep, settle=20, timed=6)
        gb = r["ms"] * 1e-3 * r["frac"] * 8e12 / 1e9
        print(f"{E:6d} x {K:4d} {gb:5.1f} GB {name:9s} frac {r['frac']:.3f} (best {r['frac_best']:.3f}) {r['ms']:.3f} ms pace {r['pace_ns']:.0f} shape {r['shape']}", flush=True)
PY
cat $OUT/size.txt
