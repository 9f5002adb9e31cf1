#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export CCX_PACE_MEMORY=0
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r02_call18_pytest.log 2>&1 || { tail -30 gpurun_out/r02_call18_pytest.log; exit 1; }
tail -2 gpurun_out/r02_call18_pytest.log
for A in "" "--workload c3" "--workload c5_50 --policy greedy" "--workload c5_64 --policy greedy"; do
  python3 bench.py --no-cpu-baseline --no-secondary $A 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-36s value %.4g frac %.3f cold %.3f max/med %.3f %s' % ('$A', d['value'], d['roofline']['frac'], d['roofline']['frac_cold'], d['roofline']['kernel_ms_max_over_median'], d['config']['launch_shape']))"
done
