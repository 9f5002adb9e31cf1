#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | grep -v amdgpu.ids > gpurun_out/gpu_suite.txt || { tail -60 gpurun_out/gpu_suite.txt; exit 1; }
tail -3 gpurun_out/gpu_suite.txt
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -2
bash profiles/scratch/r02_profiles.sh
