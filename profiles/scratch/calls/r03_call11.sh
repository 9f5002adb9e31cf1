#!/bin/bash
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_c11
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_round2.py tests/test_gpu_bench_contract.py -x -q -k "new_shape or calibration or capturing or launch_mode or bench_line or pace_cache" > $OUT/pytest.txt 2>&1 || { tail -60 $OUT/pytest.txt; exit 1; }
tail -2 $OUT/pytest.txt
timeout -k 10 200 python3 bench.py --no-cpu-baseline > $OUT/c2.json 2>> $OUT/err.txt
timeout -k 10 200 python3 bench.py --no-cpu-baseline --warmup 5 --steps 20 > $OUT/c2_w5.json 2>> $OUT/err.txt
for W in "c3 random" "c5_64 greedy" "c5_50 greedy"; do set -- $W; timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-secondary --workload $1 --policy $2 --chunk 100 --steps 24 --warmup 24 --pool 512 > $OUT/$1.json 2>> $OUT/err.txt; done
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], f"{d['value']:.4g}", "frac", round(d['roofline']['frac'],3), "frac_wall", round(d['roofline']['frac_wall'],3), "cold", round(d['cold']['ratio_to_value'],3), d['config']['pace_start_source'], round(d['config']['pace_start_ns'],1), round(d['config']['pace_probe_GBs']), round(d['config']['step_pace_ns'],1), "fill", round(d['roofline']['achievable_write_GBs_this_box']))
PY
