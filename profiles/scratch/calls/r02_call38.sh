#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -4
bash profiles/scratch/r02_soak.sh 6 10000
