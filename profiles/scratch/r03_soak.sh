#!/bin/bash
# one-off soak: randomised hypothesis examples of kernel-vs-oracle on the GPU (tests/test_gpu_parity.py), in chunks
# (a line of progress per chunk: a silent GPU command is taken to be hung after 7 minutes), log kept
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
CHUNKS=${1:-6}
N=${2:-10000}
LOG=gpurun_out/r03_hypothesis_soak.txt
echo "soak: $CHUNKS x (CCX_HYP_EXAMPLES=$N python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k arbitrary_valid_configs), randomised; $(date -u +%Y-%m-%dT%H:%MZ)" > $LOG
for c in $(seq 1 $CHUNKS); do
  CCX_HYP_EXAMPLES=$N timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k arbitrary_valid_configs -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/chunk $c: /" | tee -a $LOG
done
