#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export CCX_PACE_MEMORY=0
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_env_api.py -m gpu -x -q > gpurun_out/r02_call10_pytest.log 2>&1 || { tail -30 gpurun_out/r02_call10_pytest.log; exit 1; }
tail -1 gpurun_out/r02_call10_pytest.log
python3 profiles/scratch/tstamps.py 2>&1 | grep -v amdgpu.ids
python3 profiles/scratch/stepwise.py 2>&1 | grep -v amdgpu.ids
python3 profiles/scratch/dict_env_latency.py 2>&1 | grep -v amdgpu.ids | tail -4
