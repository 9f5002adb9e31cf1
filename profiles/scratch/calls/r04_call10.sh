#!/bin/bash
# round 4, call 10: where does the 12 % of the no-obs rollout go?  head / cur / cur with the rows shape / cur without the reward-table read
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c10
mkdir -p $OUT
cd $ROOT
D=collectivecrossing_amd/csrc/_diag
B="timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-secondary --no-obs --steps 20 --warmup 20"
for rep in 1 2; do
  CCX_DIAG_LIB=$D/libccx_head.so $B > $OUT/head_$rep.json 2>> $OUT/err.txt || echo fail
  $B > $OUT/cur_$rep.json 2>> $OUT/err.txt || echo fail
  $B --tunable small_shape=0 > $OUT/cur_rowsshape_$rep.json 2>> $OUT/err.txt || echo fail
  CCX_DIAG_LIB=$D/libccx_nortab.so $B > $OUT/nortab_$rep.json 2>> $OUT/err.txt || echo fail
  CCX_DIAG_LIB=$D/libccx_nortab.so $B --compact-obs > $OUT/nortab_compact_$rep.json 2>> $OUT/err.txt || echo fail
  $B --compact-obs > $OUT/cur_compact_$rep.json 2>> $OUT/err.txt || echo fail
  CCX_DIAG_LIB=$D/libccx_head.so $B --compact-obs > $OUT/head_compact_$rep.json 2>> $OUT/err.txt || echo fail
done
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f.split("/")[-1], f"{d['value']:.4g}", "us/env-step %.4f" % (d["roofline"]["kernel_ms_per_launch"] * 1e3 / d["config"]["steps_per_launch"]), d["config"]["launch_shape"])
PY
tail -3 $OUT/err.txt
