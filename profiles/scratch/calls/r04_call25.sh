#!/bin/bash
# round 4, call 25: grids of several rounds launched round by round (block_base / grid_blocks) -- parity, then the same
# steps-per-launch x pacing x shape table as call 24
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c25
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py tests/test_gpu_shape_guard.py -m gpu -q -x > $OUT/pytest.txt 2>&1 || { tail -30 $OUT/pytest.txt; exit 1; }
tail -3 $OUT/pytest.txt
timeout -k 10 900 python3 profiles/scratch/multi_round.py $OUT/multi_round.json > $OUT/multi_round.txt 2>&1 || { tail -20 $OUT/multi_round.txt; exit 1; }
cat $OUT/multi_round.txt
