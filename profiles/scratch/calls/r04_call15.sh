#!/bin/bash
# round 4, call 15: move order in the short-launch kernel: full GPU suite; the driver's bench command; rocprofv3 evidence for C2 and the step kernel
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c15
mkdir -p $OUT
cd $ROOT
timeout -k 10 1000 python3 -m pytest tests -m gpu -q > $OUT/pytest.txt 2>&1; tail -6 $OUT/pytest.txt; grep -n "AssertionError: (" $OUT/pytest.txt | cut -c1-400
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
timeout -k 10 300 python3 bench.py > $OUT/bench_default.json 2>> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
python3 - <<PY
import json
for name in ("bench_driver_flags", "bench_default"):
    d = json.loads(open("$OUT/%s.json" % name).read().strip().splitlines()[-1])
    print(name, "value", d["value"], "frac", d["roofline"]["frac"], "frac_wall", d["roofline"]["frac_wall"], "cold", d["cold"]["value"])
    s = d["secondary"]
    print(" step_k1", round(s["step_k1"]["us_per_step"], 3), "graph", round(s["step_k1_graph"]["us_per_step"], 3), "no_obs", round(s["no_obs"]["us_per_env_step"], 4), "compact", s["compact_obs"]["env_steps_per_sec"])
    print(" sustained", {k: v for k, v in s["sustained"].items() if k not in ("buckets", "what")})
    for w in s.get("workloads", []):
        print(" ", w.get("workload"), w.get("envs"), w.get("error") or (round(w["frac"],3), round(w["frac_wall"],3), w["launch_shape"]["lanes_per_wave"], w["launch_shape"]["writers_per_tile"], w["launch_shape"]["waves_per_block"]))
PY
timeout -k 10 400 bash profiles/collect_workload.sh r04 c2 random > $OUT/collect_c2.txt 2>&1; tail -4 $OUT/collect_c2.txt
timeout -k 10 300 bash profiles/collect_step.sh r04 > $OUT/collect_step.txt 2>&1; tail -25 $OUT/collect_step.txt
