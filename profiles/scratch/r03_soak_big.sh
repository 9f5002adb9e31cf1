#!/bin/bash
# soak of the small-batch launch shapes: larger batches per example (the oracle on the host takes most of the time)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/r03_soak_big
for c in $(seq 1 ${1:-4}); do
  CCX_HYP_ENVS=257,600,1025,2048,3000 CCX_HYP_EXAMPLES=${2:-1500} timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k arbitrary_valid_configs -p no:cacheprovider > gpurun_out/r03_soak_big/chunk_$c.txt 2>&1
  rc=$?
  tail -1 gpurun_out/r03_soak_big/chunk_$c.txt | sed "s/^/chunk $c: /"
  if [ $rc -ne 0 ]; then grep -n "Falsifying\|AssertionError\|Error\|error" gpurun_out/r03_soak_big/chunk_$c.txt | head -20; exit 0; fi
done
