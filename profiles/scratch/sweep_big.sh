#!/bin/bash
mkdir -p gpurun_out/big
run() { # name E chunk extra-env args...
  n=$1; E=$2; C=$3; shift 3
  timeout -k 10 120 python bench.py --no-cpu-baseline --envs-per-gpu $E --chunk $C --steps 8 --warmup 1 "$@" > gpurun_out/big/$n.json 2> gpurun_out/big/$n.err || { tail -3 gpurun_out/big/$n.err; exit 1; }
}
for rep in 1 2; do
for E in 8192 16384; do
  C=250; [ $E -ge 16384 ] && C=125
  for T in 16 8 4 2; do run e${E}_t${T}_$rep $E $C --throttle $T; done
  CCX_LDS_PAD=82000 run e${E}_pad82_$rep $E $C
  CCX_LDS_PAD=54000 run e${E}_pad54_$rep $E $C
  CCX_LDS_PAD=54000 run e${E}_pad54t8_$rep $E $C --throttle 8
done
done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/big/e*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
    print(f.split("/")[-1][:-5], "%.3e"%d["value"], "frac %.3f"%r["frac"], "ach %.3f"%r["frac_of_achievable"])
PY
