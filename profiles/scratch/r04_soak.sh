#!/bin/bash
# one-off soak (round 4): randomised hypothesis examples of kernel-vs-oracle on the GPU (tests/test_gpu_parity.py) with the round's new
# draws -- short-launch kernel on / off with 1-7 row waves and 8-64 lanes, the no-rows launch shape, random user reward / terminated
# tables -- in chunks (a line of progress per chunk), log kept
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
CHUNKS=${1:-6}
N=${2:-6000}
LOG=gpurun_out/r04_hypothesis_soak.txt
echo "soak: $CHUNKS x (CCX_HYP_EXAMPLES=$N python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k arbitrary_valid_configs), randomised; $(date -u +%Y-%m-%dT%H:%MZ)" > $LOG
for c in $(seq 1 $CHUNKS); do
  CCX_HYP_EXAMPLES=$N timeout -k 10 420 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k arbitrary_valid_configs -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/chunk $c: /" | tee -a $LOG
done
# the small-batch / multi-tile shapes: larger batches
for c in $(seq 1 2); do
  CCX_HYP_ENVS=257,600,1025,2048,3000 CCX_HYP_EXAMPLES=$((N / 3)) timeout -k 10 420 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k arbitrary_valid_configs -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/large-batch chunk $c: /" | tee -a $LOG
done
