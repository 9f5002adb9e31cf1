#!/bin/bash
# one-off soak: randomised hypothesis examples of kernel-vs-oracle on the GPU (tests/test_gpu_parity.py), log kept
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
N=${1:-6000}
( echo "CCX_HYP_EXAMPLES=$N python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k arbitrary_valid_configs  (randomised; $(date -u +%Y-%m-%dT%H:%MZ))"; \
  CCX_HYP_EXAMPLES=$N timeout -k 10 1000 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k arbitrary_valid_configs --hypothesis-show-statistics 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r02_hypothesis_soak.txt
tail -25 gpurun_out/r02_hypothesis_soak.txt
