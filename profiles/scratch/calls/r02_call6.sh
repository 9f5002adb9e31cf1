#!/bin/bash
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_call6
mkdir -p $OUT
cd $ROOT
export CCX_PACE_MEMORY=0
timeout -k 10 600 python3 -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py -m gpu -x -q -k "tunables or pacing or adaptive or full_size or consecutive or autoreset" > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
P='"pace_phase":1'
A='[{}, {'$P',"tile_map":1}, {'$P',"tile_map":2}, {'$P',"tile_map":3}, {'$P',"tile_map":4}, {'$P',"tile_map":5}, {'$P',"tile_map":6}, {'$P',"tile_map":0}]'
for W in c5_50 c5_64 c3; do
timeout -k 10 500 python3 profiles/scratch/sweep_knobs.py $W 250 40 20 "$A" > $OUT/sweep_$W.txt 2>&1 || { tail -20 $OUT/sweep_$W.txt; exit 1; }
cut -c1-172 $OUT/sweep_$W.txt
done
# does the slip rescue paces beyond the cliff?
S='[{"pace":700}, {"pace":700,"pace_slip":0}, {"pace":680}, {"pace":680,"pace_slip":0}, {"pace":660}, {"pace":660,"pace_slip":0}, {}, {"pace_slip":0}]'
timeout -k 10 300 python3 profiles/scratch/sweep_knobs.py c2 250 40 20 "$S" > $OUT/slip_c2.txt 2>&1
cut -c1-172 $OUT/slip_c2.txt
S3='[{'$P',"tile_map":1,"pace":5000}, {'$P',"tile_map":1,"pace":5000,"pace_slip":0}, {'$P',"tile_map":1,"pace":4800}, {'$P',"tile_map":1,"pace":4800,"pace_slip":0}]'
timeout -k 10 300 python3 profiles/scratch/sweep_knobs.py c3 250 40 20 "$S3" > $OUT/slip_c3.txt 2>&1
cut -c1-172 $OUT/slip_c3.txt
