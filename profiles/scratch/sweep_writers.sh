#!/bin/bash
# writers x throttle sweep for one workload, one call: sweep_writers.sh c5_50 "2 3 4 5 7" "-1 8 16"
W=$1; mkdir -p gpurun_out/sw
for w in $2; do for t in $3; do
  timeout -k 10 120 python bench.py --no-cpu-baseline --workload $W --chunk 100 --steps 24 --warmup 30 --pool 512 --writers $w --throttle $t > gpurun_out/sw/${W}_w${w}_t${t}.json 2> gpurun_out/sw/err.log || { tail -3 gpurun_out/sw/err.log; exit 1; }
done; done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/sw/${W}_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]; print(f.split("/")[-1][:-5], "us/step %.2f"%(r["kernel_ms_per_launch"]*1000/50), "frac %.3f"%r["frac"])
PY
