#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export CCX_PACE_CACHE=/tmp/ccx_pace_test.json
for rep in 1 2; do
for M in 0 1; do
    CCX_PACE_MEMORY=$M python3 bench.py --no-cpu-baseline --no-secondary --warmup 5 --steps 20 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pace_memory=$M  value %.4g frac %.3f  cold %.4g frac_cold %.3f ratio %.3f  pace %.1f name %s' % (d['value'], d['roofline']['frac'], d['cold']['value'], d['roofline']['frac_cold'], d['cold']['ratio_to_value'], d['config']['step_pace_ns'], d['device']['name']))"
done
done
cat /tmp/ccx_pace_test.json
