#!/bin/bash
# round 4, call 11: conditional reward-table read: no-obs A/B vs HEAD, then the full GPU suite and the sweep with the v2 writer rules
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c11
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
timeout -k 10 1000 python3 -m pytest tests -m gpu -q > $OUT/pytest.txt 2>&1; tail -15 $OUT/pytest.txt; grep -n "AssertionError: (" $OUT/pytest.txt | cut -c1-500
timeout -k 10 600 python3 profiles/scratch/shape_sweep.py $OUT/shape_sweep.json > $OUT/shape_sweep.txt 2>&1 || { tail -20 $OUT/shape_sweep.txt; exit 1; }
python3 - <<PY
import json
R = json.load(open("$OUT/shape_sweep.json"))
bad = [r for r in R if r["default_over_best"] and r["default_over_best"] >= 1.06]
print(len(bad), "of", len(R), "points with default >= 1.06 x best")
for r in bad:
    d = r["default"]["shape"]; b = r["best"]
    print(r["N"], r["E"], r["mode"], (d["lanes_per_wave"], d["writers_per_tile"], d["waves_per_block"]), r["default"]["us_per_env_step"], "x", r["default_over_best"], "best", (b["lanes"], b["writers"]), b["us_per_env_step"])
PY
