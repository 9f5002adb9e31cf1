#!/bin/bash
# round 4, call 85: a last long soak on the last binary (occ_tables among the draws)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout -k 10 1150 bash profiles/scratch/r04_soak.sh 6 8000
