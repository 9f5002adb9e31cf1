#!/bin/bash
# round 4, call 33: step shape (two tiles per CU), 65 536 envs x 32 steps in the bench -- full GPU suite, stamps, the driver's bench command
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c33
mkdir -p $OUT
cd $ROOT
timeout -k 10 1000 python3 -m pytest tests -m gpu -q > $OUT/pytest.txt 2>&1; tail -5 $OUT/pytest.txt; grep -n "^FAILED\|^ERROR" $OUT/pytest.txt | head
timeout -k 10 120 python3 profiles/scratch/step_tstamps.py 4096 > $OUT/tstamps.txt 2>&1; grep -v amdgpu $OUT/tstamps.txt
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
