#!/bin/bash
# hunt for a failing hypothesis example: full pytest output of every chunk kept, stop at the first failure
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/r03_soak_find
for c in $(seq 1 ${1:-8}); do
  CCX_HYP_EXAMPLES=${2:-10000} timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k arbitrary_valid_configs -p no:cacheprovider > gpurun_out/r03_soak_find/chunk_$c.txt 2>&1
  rc=$?
  tail -1 gpurun_out/r03_soak_find/chunk_$c.txt | sed "s/^/chunk $c: /"
  if [ $rc -ne 0 ]; then grep -n "Falsifying\|AssertionError\|Error\|error" gpurun_out/r03_soak_find/chunk_$c.txt | head -20; exit 0; fi
done
