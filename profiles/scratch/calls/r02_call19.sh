#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export CCX_PACE_MEMORY=0
for rep in 1 2 3; do
for L in "" collectivecrossing_amd/csrc/_diag/libccx_cur.so; do
    CCX_DIAG_LIB=$L python3 bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('lib=%-10s value %.4g frac %.3f  pace %.1f  fill %.0f' % ('$L'[-6:] or 'new', d['value'], d['roofline']['frac'], d['config']['step_pace_ns'], d['roofline']['achievable_write_GBs_this_box']))"
done
done
