#!/bin/bash
# round 4, call 31: step kernel with a small-output wave -- parity, stamps, short launches
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c31
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_step_kernel.py tests/test_gpu_position_only.py tests/test_gpu_env_api.py tests/test_gpu_rllib.py tests/test_gpu_vector.py -m gpu -q -x > $OUT/pytest.txt 2>&1 || { tail -40 $OUT/pytest.txt; exit 1; }
tail -3 $OUT/pytest.txt
timeout -k 10 120 python3 profiles/scratch/step_tstamps.py 4096 > $OUT/tstamps.txt 2>&1; grep -v amdgpu $OUT/tstamps.txt
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
python3 - <<PY
import json
d = json.loads(open("$OUT/bench_driver_flags.json").read().strip().splitlines()[-1])
print("value", d["value"], "frac", d["roofline"]["frac"])
s = d["secondary"]
for k in ("step_k1", "step_k1_graph"):
    print(k, {kk: vv for kk, vv in s.get(k).items() if kk != "what"})
print("short", {k: (round(v["us_per_launch"], 2), round(v["frac"], 3)) for k, v in s["short_launches"].items() if k.startswith("k")})
PY
