#!/bin/bash
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_c10
mkdir -p $OUT
cd $ROOT
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1 || { tail -60 $OUT/pytest.txt; exit 1; }
tail -2 $OUT/pytest.txt
timeout -k 10 200 python3 bench.py > $OUT/c2.json 2>> $OUT/err.txt
timeout -k 10 200 python3 bench.py --no-cpu-baseline --warmup 5 --steps 20 > $OUT/c2_w5.json 2>> $OUT/err.txt
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], f"{d['value']:.4g}", "frac", round(d['roofline']['frac'],3), "frac_wall", round(d['roofline']['frac_wall'],3), "cold", d['cold']['ratio_to_value'], d['config']['pace_start_source'], d['config']['pace_start_ns'], d['config']['pace_probe_GBs'], d['config']['step_pace_ns'], d.get('secondary'))
PY
