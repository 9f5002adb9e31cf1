#!/bin/bash
# round 4, call 6: two launch shapes per handle (rows / no rows) with the sweep-derived writer rules: full GPU suite, sweep again
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c06
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1 || { tail -40 $OUT/pytest.txt; exit 1; }
tail -3 $OUT/pytest.txt
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
