#!/bin/bash
# round 4, call 84: the default invocation of bench.py on the last binary
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c84
mkdir -p $OUT
cd $ROOT
( time timeout -k 10 500 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench.err ) 2>&1 | tail -3
python3 - <<PY
import json
d = json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "scaling", "vs_baseline")})
print("roofline", d["roofline"]); print("cpu", {k: v for k, v in d["cpu_baseline"].items() if k != "sample"})
PY
