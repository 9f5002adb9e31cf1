#!/bin/bash
# round 4, call 38: final binary -- the new round-by-round test, rocprofv3 evidence (C2 rollout kernel, step kernel), full GPU suite
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c38
mkdir -p $OUT
cd $ROOT
timeout -k 10 300 python3 -m pytest tests/test_gpu_round2.py -m gpu -q -x -k "round_by_round or partial_last" > $OUT/pytest_rounds.txt 2>&1 || { tail -30 $OUT/pytest_rounds.txt; exit 1; }
tail -2 $OUT/pytest_rounds.txt
timeout -k 10 400 bash profiles/collect_workload.sh r04 c2 random > $OUT/collect_c2.txt 2>&1; tail -3 $OUT/collect_c2.txt
cd $ROOT
timeout -k 10 400 bash profiles/collect_step.sh r04 > $OUT/collect_step.txt 2>&1; tail -4 $OUT/collect_step.txt
cd $ROOT
timeout -k 10 1000 python3 -m pytest tests -m gpu -q > $OUT/pytest.txt 2>&1; tail -4 $OUT/pytest.txt; grep -n "^FAILED\|^ERROR" $OUT/pytest.txt | head
