#!/bin/bash
# round 4, call 8: the rest of the GPU suite (from test_gpu_round2 on), headline check, sweep with the new rules
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c08
mkdir -p $OUT
cd $ROOT
timeout -k 10 1000 python3 -m pytest tests/test_gpu_round2.py tests/test_gpu_shape_guard.py tests/test_gpu_step_kernel.py tests/test_gpu_vector.py tests/test_gpu_large_grid_policy.py -m gpu -q > $OUT/pytest.txt 2>&1; tail -30 $OUT/pytest.txt
for rep in 1 2; do
timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-secondary > $OUT/bench_$rep.json 2>> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
done
python3 - <<PY
import json
for rep in (1, 2):
    d = json.loads(open("$OUT/bench_%d.json" % rep).read().strip().splitlines()[-1])
    print("headline", rep, d["value"], "frac", d["roofline"]["frac"], "frac_wall", d["roofline"]["frac_wall"])
PY
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
