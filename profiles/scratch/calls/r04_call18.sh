#!/bin/bash
# round 4, call 18: the hypothesis soak proper (8 x 10 000 randomised examples of kernel-vs-oracle with the round's new draws;
# r04_soak.sh appends 2 x 3333 examples at batches of 257-3000 envs)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout -k 10 1150 bash profiles/scratch/r04_soak.sh 6 10000
