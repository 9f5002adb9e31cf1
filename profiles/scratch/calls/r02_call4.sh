#!/bin/bash
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_call4
mkdir -p $OUT
cd $ROOT
export CCX_PACE_MEMORY=0
python3 profiles/diag_stamps.py > $OUT/stamps.txt 2>&1 || { tail $OUT/stamps.txt; exit 1; }
python3 profiles/scratch/tstamps.py > $OUT/tstamps.txt 2>&1 || { tail $OUT/tstamps.txt; exit 1; }
cat $OUT/stamps.txt $OUT/tstamps.txt
A='[{}, {"pace_phase":1,"tile_map":1}, {"pace_phase":1,"tile_map":1,"writer_split":1}, {"pace_phase":3}, {"pace_phase":3,"writer_split":1}, {"pace_phase":1,"tile_map":1,"writers":7}, {"pace_phase":3,"writers":7}]'
timeout -k 10 500 python3 profiles/scratch/sweep_knobs.py c5_50 250 40 20 "$A" > $OUT/sweep_c5_50.txt 2>&1 || { tail -20 $OUT/sweep_c5_50.txt; exit 1; }
cut -c1-175 $OUT/sweep_c5_50.txt
