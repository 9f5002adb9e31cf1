#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export CCX_PACE_MEMORY=0
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r02_call20_pytest.log 2>&1 || { tail -30 gpurun_out/r02_call20_pytest.log; exit 1; }
tail -2 gpurun_out/r02_call20_pytest.log
bash profiles/scratch/r02_profiles.sh
