#!/bin/bash
# round 4, call 39: hypothesis soak on the final binary
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout -k 10 1150 bash profiles/scratch/r04_soak.sh 6 8000
