#!/bin/bash
# round 4, call 5: the full bench line (new secondary blocks) exactly as the driver runs it + the default invocation
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c05
mkdir -p $OUT
cd $ROOT
( time timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2> $OUT/bench_driver_flags.err ) 2> $OUT/time_driver.txt || { tail -20 $OUT/bench_driver_flags.err; exit 1; }
cat $OUT/time_driver.txt
python3 - <<PY
import json
d = json.loads(open("$OUT/bench_driver_flags.json").read().strip().splitlines()[-1])
print("value", d["value"], "frac", d["roofline"]["frac"], "cold", d["cold"]["value"])
s = d["secondary"]
for k in ("no_obs", "compact_obs", "step_k1", "step_k1_graph"):
    print(k, s.get(k))
print("short", json.dumps(s.get("short_launches")))
su = s.get("sustained"); print("sustained", {k: v for k, v in su.items() if k != "buckets"}); print([ (b["t_s"], round(b["ms_median"],4), round(b["ms_max"],4)) for b in su["buckets"]])
for w in s.get("workloads", []):
    print(w.get("workload"), w.get("envs"), w.get("error") or (round(w["frac"],3), round(w["frac_wall"],3), round(w["kernel_ms_per_launch"],4), w["launch_shape"]["lanes_per_wave"], w["launch_shape"]["writers_per_tile"]))
print("cpu", {k: v for k, v in d["cpu_baseline"].items() if k != "reference_python"})
PY
