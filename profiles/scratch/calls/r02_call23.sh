#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export CCX_PACE_MEMORY=0
for rep in 1 2; do for L in "" collectivecrossing_amd/csrc/_diag/libccx_edge.so; do for W in "c5_50 --policy greedy" "c5_64 --policy greedy"; do
CCX_DIAG_LIB=$L python3 bench.py --no-cpu-baseline --no-secondary --warmup 40 --steps 20 --workload $W 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('lib=%-8s %-24s frac %.3f pace %.0f' % ('$L'[-7:] or 'shipped', '$W', d['roofline']['frac'], d['config']['step_pace_ns']))"
done; done; done
