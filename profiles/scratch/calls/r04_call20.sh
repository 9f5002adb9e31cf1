#!/bin/bash
# round 4, call 20: the driver's tiers on the committed tree: pytest -m gpu, smoke(), the bench command
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c20
mkdir -p $OUT
cd $ROOT
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $OUT/pytest.txt 2>&1; tail -5 $OUT/pytest.txt
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3
( time timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | grep real
python3 -c "
import json
d = json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'frac', d['roofline']['frac'], 'traffic_from_profiles', d['roofline']['traffic_from_profiles'])
print('keys', sorted(d['secondary'].keys()))
"
