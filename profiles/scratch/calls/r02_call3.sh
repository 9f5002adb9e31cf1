#!/bin/bash
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_call3
mkdir -p $OUT
cd $ROOT
export CCX_PACE_MEMORY=0
B='{"pace_phase":1,"tile_map":1}'
C3='[{}, '$B', {"pace_phase":1,"tile_map":1,"writer_split":1}, {"pace_phase":3}, {"pace_phase":3,"writer_split":1}, {"pace_phase":1,"tile_map":1,"writers":2}, {"pace_phase":1,"tile_map":1,"writers":4}, {"pace_phase":1,"tile_map":1,"writers":5}, {"pace_phase":1,"tile_map":1,"pace":5200}, {"pace_phase":1,"tile_map":1,"pace":5100}, {"pace_phase":1,"tile_map":1,"pace":5000}, {"pace_phase":1,"tile_map":1,"pace":4900}, {"pace_phase":1,"tile_map":1,"lanes":32}]'
C5='[{}, '$B', {"pace_phase":1,"tile_map":1,"writer_split":1}, {"pace_phase":3}, {"pace_phase":3,"writer_split":1}, {"pace_phase":1,"tile_map":1,"writers":2}, {"pace_phase":1,"tile_map":1,"writers":4}, {"pace_phase":1,"tile_map":1,"writers":5}, {"pace_phase":1,"tile_map":1,"writers":7}, {"pace_phase":1,"tile_map":1,"pace":10200}, {"pace_phase":1,"tile_map":1,"pace":10000}, {"pace_phase":1,"tile_map":1,"pace":9800}, {"pace_phase":1,"tile_map":1,"pace":9600}]'
timeout -k 10 500 python3 profiles/scratch/sweep_knobs.py c3 250 40 20 "$C3" > $OUT/sweep_c3.txt 2>&1 || { tail -20 $OUT/sweep_c3.txt; exit 1; }
timeout -k 10 500 python3 profiles/scratch/sweep_knobs.py c5_64 250 40 20 "$C5" > $OUT/sweep_c5.txt 2>&1 || { tail -20 $OUT/sweep_c5.txt; exit 1; }
timeout -k 10 300 python3 profiles/scratch/sweep_knobs.py c2 250 40 20 '[{}, {"writer_split":1}]' > $OUT/sweep_c2.txt 2>&1
cut -c1-200 $OUT/sweep_c3.txt $OUT/sweep_c5.txt $OUT/sweep_c2.txt
