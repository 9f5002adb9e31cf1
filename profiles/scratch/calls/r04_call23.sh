#!/bin/bash
# round 4, call 23: step kernel with the row waves' table words really loaded before the barrier (the compiler had sunk them below it
# in the K = 1 instantiations) and the reward-table presence in a preloaded bit; C2 at 65 536 envs with 64 steps per launch
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c23
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_step_kernel.py tests/test_gpu_position_only.py -m gpu -q -x > $OUT/pytest.txt 2>&1 || { tail -30 $OUT/pytest.txt; exit 1; }
tail -3 $OUT/pytest.txt
timeout -k 10 300 python3 - > $OUT/exp.txt 2>&1 <<'PY' || { tail -20 $OUT/exp.txt; exit 1; }
import json, torch, bench
dev = torch.device("cuda:0")
for envs, chunk in ((65536, 24), (65536, 64), (65536, 128), (32768, 64), (32768, 128)):
    r = bench.measure_workload(torch, dev, "c2", envs, chunk, "random")
    print(envs, chunk, round(r["frac"], 4), round(r["frac_wall"], 4), round(r["kernel_ms_per_launch"], 4), r["launch_shape"]["writers_per_tile"], r["launch_shape"]["waves_per_block"], flush=True)
PY
cat $OUT/exp.txt
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
python3 - <<PY
import json
d = json.loads(open("$OUT/bench_driver_flags.json").read().strip().splitlines()[-1])
print("value", d["value"], "frac", d["roofline"]["frac"])
s = d["secondary"]
for k in ("no_obs", "compact_obs", "step_k1", "step_k1_graph"):
    print(k, {kk: vv for kk, vv in s.get(k).items() if kk != "what"})
print("short", {k: (round(v["us_per_launch"], 2), round(v["frac"], 3)) for k, v in s["short_launches"].items() if k.startswith("k")})
PY
