#!/bin/bash
# round 4, call 16: small-output writer loops specialised by output set (no-rows rollouts): A/B vs HEAD's library, then parity
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c16
mkdir -p $OUT
cd $ROOT
D=collectivecrossing_amd/csrc/_diag
B="timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 20"
for rep in 1 2; do
  CCX_DIAG_LIB=$D/libccx_head.so $B --no-obs > $OUT/head_noobs_$rep.json 2>> $OUT/err.txt || echo fail
  $B --no-obs > $OUT/cur_noobs_$rep.json 2>> $OUT/err.txt || echo fail
  CCX_DIAG_LIB=$D/libccx_head.so $B --compact-obs > $OUT/head_compact_$rep.json 2>> $OUT/err.txt || echo fail
  $B --compact-obs > $OUT/cur_compact_$rep.json 2>> $OUT/err.txt || echo fail
  CCX_DIAG_LIB=$D/libccx_head.so timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-secondary > $OUT/head_rows_$rep.json 2>> $OUT/err.txt || echo fail
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-secondary > $OUT/cur_rows_$rep.json 2>> $OUT/err.txt || echo fail
  CCX_DIAG_LIB=$D/libccx_head.so $B --no-obs --envs-per-gpu 1024 > $OUT/head_noobs1024_$rep.json 2>> $OUT/err.txt || echo fail
  $B --no-obs --envs-per-gpu 1024 > $OUT/cur_noobs1024_$rep.json 2>> $OUT/err.txt || echo fail
done
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f.split("/")[-1], f"{d['value']:.4g}", "frac %.4f" % d["roofline"]["frac"], "us/env-step %.4f" % (d["roofline"]["kernel_ms_per_launch"] * 1e3 / d["config"]["steps_per_launch"]))
PY
timeout -k 10 1000 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -q -x > $OUT/pytest.txt 2>&1; tail -4 $OUT/pytest.txt
