#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export CCX_PACE_MEMORY=0
timeout -k 10 900 python3 -m pytest tests/test_gpu_vector.py tests/test_gpu_env_api.py tests/test_gpu_custom_strategies.py -m gpu -x -q 2>&1 | tail -30
