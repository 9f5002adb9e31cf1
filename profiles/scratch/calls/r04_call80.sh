#!/bin/bash
# round 4, call 80: the last binary -- full GPU suite, smoke, the driver's bench command, 6000 randomised examples
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c80
mkdir -p $OUT
cd $ROOT
timeout -k 10 1000 python3 -m pytest tests -m gpu -q > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt; grep -n "^FAILED\|^ERROR" $OUT/pytest.txt | head
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
python3 - <<PY
import json
d = json.loads(open("$OUT/bench_driver_flags.json").read().strip().splitlines()[-1])
print("value", d["value"], "frac", d["roofline"]["frac"], "cold", d["cold"]["value"])
s = d["secondary"]
for k in ("no_obs", "compact_obs", "step_k1", "step_k1_graph"):
    print(k, {kk: vv for kk, vv in s.get(k).items() if kk != "what"})
for w in s.get("workloads", []):
    print(w.get("workload"), w.get("envs"), w.get("error") or (round(w["frac"],3), round(w["frac_wall"],3)))
PY
CCX_HYP_EXAMPLES=6000 timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k arbitrary_valid_configs -p no:cacheprovider 2>&1 | tail -1
