#!/bin/bash
# round 4, call 77: final binary (after the big-grid, misaligned-slab and step-kernel work) -- full GPU suite, smoke, sweep, bench, soak
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c77
mkdir -p $OUT
cd $ROOT
timeout -k 10 1000 python3 -m pytest tests -m gpu -q > $OUT/pytest.txt 2>&1; tail -4 $OUT/pytest.txt; grep -n "^FAILED\|^ERROR" $OUT/pytest.txt | head
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -k 10 900 python3 profiles/scratch/shape_sweep.py $OUT/shape_sweep.json > $OUT/shape_sweep.txt 2>&1 || { tail -20 $OUT/shape_sweep.txt; exit 1; }
python3 - <<PY
import json
R = json.load(open("$OUT/shape_sweep.json"))
bad = [r for r in R if r["default_over_best"] and r["default_over_best"] >= 1.06]
print(len(bad), "of", len(R), "points with default >= 1.06 x best; worst", max(r["default_over_best"] for r in R))
for r in bad:
    d = r["default"]["shape"]; b = r["best"]
    print(r["N"], r["E"], r["mode"], (d["lanes_per_wave"], d["writers_per_tile"], d["waves_per_block"]), r["default"]["us_per_env_step"], "x", r["default_over_best"], "best", (b["lanes"], b["writers"]), b["us_per_env_step"])
PY
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
python3 - <<PY
import json
d = json.loads(open("$OUT/bench_driver_flags.json").read().strip().splitlines()[-1])
print("value", d["value"], "frac", d["roofline"]["frac"], "cold", d["cold"]["value"])
s = d["secondary"]
for k in ("no_obs", "compact_obs", "step_k1", "step_k1_graph"):
    print(k, {kk: vv for kk, vv in s.get(k).items() if kk != "what"})
print("short", {k: (round(v["us_per_launch"], 2), round(v["frac"], 3)) for k, v in s["short_launches"].items() if k.startswith("k")})
su = s["sustained"]; print("sustained", {k: v for k, v in su.items() if k not in ("buckets", "what")})
for w in s.get("workloads", []):
    print(w.get("workload"), w.get("envs"), w.get("error") or (round(w["frac"],3), round(w["frac_wall"],3), round(w["kernel_ms_per_launch"],4), w["launch_shape"]["lanes_per_wave"], w["launch_shape"]["writers_per_tile"], w["launch_shape"]["waves_per_block"]))
PY
CCX_HYP_EXAMPLES=12000 timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k arbitrary_valid_configs -p no:cacheprovider 2>&1 | tail -1
CCX_HYP_ENVS=257,600,1025,2048,3000 CCX_HYP_EXAMPLES=2500 timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k arbitrary_valid_configs -p no:cacheprovider 2>&1 | tail -1
