#!/bin/bash
# round 4, call 92: randomised examples with batches of 3001 .. 9001 envs (several rounds on the larger drawn grids: the tables-vs-rounds rule inside the soak)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
CCX_HYP_ENVS=3001,5000,9001 CCX_HYP_EXAMPLES=1200 timeout -k 10 1000 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k arbitrary_valid_configs -p no:cacheprovider 2>&1 | tail -1
