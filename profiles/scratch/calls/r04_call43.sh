#!/bin/bash
# round 4, call 43: pacing from 0.40 us per step on, the ring in paced launches of near-chain-bound batches: tests, the table again, bench
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c43
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_shape_guard.py tests/test_gpu_round2.py tests/test_gpu_parity.py -m gpu -q > $OUT/pytest.txt 2>&1; tail -4 $OUT/pytest.txt; grep -n "^FAILED\|^ERROR\|AssertionError: (" $OUT/pytest.txt | cut -c1-600 | head
timeout -k 10 600 python3 profiles/scratch/small_batch2.py 2>&1 | grep -v amdgpu | tee $OUT/small_batch2.txt
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
python3 - <<PY
import json
d = json.loads(open("$OUT/bench_driver_flags.json").read().strip().splitlines()[-1])
print("value", d["value"], "frac", d["roofline"]["frac"], "cold", d["cold"]["value"])
s = d["secondary"]
for w in s.get("workloads", []):
    print(w.get("workload"), w.get("envs"), w.get("error") or (round(w["frac"],3), round(w["frac_wall"],3), round(w["kernel_ms_per_launch"],4), w["launch_shape"]["lanes_per_wave"], w["launch_shape"]["writers_per_tile"], w["launch_shape"]["waves_per_block"]))
PY
