#!/bin/bash
# a remembered start pace is also the first floor: cold window of the driver's invocation, three fresh processes + a trace
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for i in 1 2 3; do
  timeout -k 10 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/drv_$i.json 2> gpurun_out/drv_$i.err
  python3 - gpurun_out/drv_$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.4e frac %.3f  cold %.4e frac_cold %.3f ratio %.3f pace %.1f" % (d["value"], d["roofline"]["frac"], d["cold"]["value"], d["roofline"]["frac_cold"], d["cold"]["ratio_to_value"], d["config"]["step_pace_ns"]))
PY
done
timeout -k 10 200 python3 profiles/scratch/pace_trace.py c2 50 2>&1 | grep -v amdgpu.ids | awk '/^launch/{print $2,$3,$7,$10}' | paste - - - - -
