#!/bin/bash
# in-call sweep of writer waves per tile
mkdir -p gpurun_out/abw
for rep in 1 2; do for W in $1; do for WR in $2; do
  X="--chunk 250 --steps 1000 --warmup 250"; [ $W != c2 ] && X="--chunk 50 --steps 300 --warmup 50 --pool 512"
  timeout -k 10 200 python bench.py --no-cpu-baseline --workload $W $X --writers $WR > gpurun_out/abw/${W}_w${WR}_$rep.json 2> gpurun_out/abw/e.err || { tail -3 gpurun_out/abw/e.err; exit 1; }
done; done; done
python - <<PY
import json,glob,collections
r=collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/abw/*.json")):
    k=f.split("/")[-1][:-7]
    d=json.loads(open(f).read().strip().splitlines()[-1]); r[k].append(d["roofline"]["frac"])
for k in sorted(r): print(k, ["%.4f"%v for v in r[k]])
PY
