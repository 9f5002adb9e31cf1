#!/bin/bash
# the driver's own invocation (5 warm-ups, 20 timed launches) in three fresh processes: cold window vs settled window
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for i in 1 2 3; do
  timeout -k 10 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/drv_$i.json 2> gpurun_out/drv_$i.err
  python3 - gpurun_out/drv_$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.4e frac %.3f  cold %.4e frac_cold %.3f  pace %.1f  max/med %.3f" % (d["value"], d["roofline"]["frac"], d["cold"]["value"], d["roofline"]["frac_cold"], d["config"]["step_pace_ns"], d["roofline"]["kernel_ms_max_over_median"]), d["cold"])
PY
done
