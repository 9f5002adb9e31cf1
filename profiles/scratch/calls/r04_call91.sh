#!/bin/bash
# round 4, call 91: bench contract tests + the driver's bench command after the last bench.py change
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c91
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_bench_contract.py -m gpu -q 2>&1 | tail -2
( time timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2> $OUT/bench.err ) 2>&1 | grep real
python3 - <<PY
import json
d = json.loads(open("$OUT/bench_driver_flags.json").read().strip().splitlines()[-1])
print("value", d["value"], "frac", d["roofline"]["frac"], "cold", d["cold"]["value"])
for w in d["secondary"].get("workloads", []):
    print(w.get("workload"), w.get("envs"), w.get("error") or (round(w["frac"],3), round(w["frac_wall"],3)))
PY
