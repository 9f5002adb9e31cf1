#!/bin/bash
# round 4, call 32: the step kernel's launch-shape knobs once more, now that the row waves' table words are really preloaded
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c32
mkdir -p $OUT
cd $ROOT
for w in c2 c3 c5_64; do
  E=4096; [ $w = c5_64 ] && E=1024
  timeout -k 10 200 python3 profiles/scratch/step_ab.py $w $E 2>/dev/null > $OUT/shapes_$w.txt || { tail -5 $OUT/shapes_$w.txt; exit 1; }
  echo "== $w"; cat $OUT/shapes_$w.txt
done
