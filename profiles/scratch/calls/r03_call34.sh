#!/bin/bash
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_c34
mkdir -p $OUT
cd $ROOT
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1 || { tail -60 $OUT/pytest.txt; exit 1; }
tail -2 $OUT/pytest.txt
for E in 4096 64; do timeout -k 10 120 python3 profiles/scratch/step_k1.py $E 2>&1 | grep -v amdgpu; done
timeout -k 10 300 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench.err
python3 - <<PY
import json
d = json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1])
print(f"{d['value']:.4g}", d['roofline']['frac'], d['roofline']['frac_wall'], d['cold']['ratio_to_value'], {k: round(v.get('us_per_step', v.get('us_per_env_step', 0)), 3) for k, v in d['secondary'].items()}, d['cpu_baseline']['value'])
PY
