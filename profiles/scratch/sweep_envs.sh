#!/bin/bash
# throughput of the default launch shape over batch sizes (C2 geometry), one call
mkdir -p gpurun_out/sweep
for E in 64 256 1024 2048 4096 8192 16384 65536; do
  C=500; [ $E -ge 16384 ] && C=100; [ $E -ge 65536 ] && C=40
  timeout -k 10 120 python bench.py --no-cpu-baseline --envs-per-gpu $E --chunk $C --steps 16 --warmup 12 > gpurun_out/sweep/e$E.json 2> gpurun_out/sweep/e$E.err || { tail -3 gpurun_out/sweep/e$E.err; exit 1; }
done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/sweep/e*.json"), key=lambda f:int(f.split("/e")[-1][:-5])):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
    print(d["config"]["envs_per_gpu"], "%.3e"%d["value"], "frac %.3f"%r["frac"], "ach %.3f"%r["frac_of_achievable"], d["config"]["launch_shape"])
PY
