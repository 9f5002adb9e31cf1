#!/bin/bash
# A/B harness: run the bench for several diagnostic builds of libccx INSIDE ONE gpurun call
# (boxes differ by up to 20 % from call to call, so only in-call comparisons are meaningful).
# usage: ab.sh "<lib names>" "<workloads>"   e.g. ab.sh "base contig late" "c2 c3"
mkdir -p gpurun_out/ab
for rep in 1 2; do
  for L in $1; do
    for W in $2; do
      X="--steps 40 --warmup 80"; [ $W != c2 ] && X="--chunk 100 --steps 24 --warmup 24 --pool 512"
      CCX_DIAG_LIB=collectivecrossing_amd/csrc/_diag/libccx_$L.so timeout -k 10 90 python bench.py --no-cpu-baseline --workload $W $X > gpurun_out/ab/${L}_${W}_$rep.json 2> gpurun_out/ab/${L}_${W}_$rep.err || { tail -3 gpurun_out/ab/${L}_${W}_$rep.err; exit 1; }
    done
  done
done
python - <<PY
import json,glob,collections
r=collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/ab/*.json")):
    L,W,_=f.split("/")[-1][:-5].rsplit("_",2)
    d=json.loads(open(f).read().strip().splitlines()[-1]); r[(W,L)].append(d["roofline"]["frac"])
for k in sorted(r): print(k, ["%.4f"%v for v in r[k]])
PY
python - <<PY
import json,glob,collections
c=collections.defaultdict(set)
for f in sorted(glob.glob("gpurun_out/ab/*.json")):
    L,W,_=f.split("/")[-1][:-5].rsplit("_",2)
    d=json.loads(open(f).read().strip().splitlines()[-1]); c[W].add(json.dumps(d["counters"],sort_keys=True))
for W in c: print("counters identical across variants for", W, ":", len(c[W])==1)
PY
