#!/bin/bash
# round 4, call 89: rocprofv3 kernel-trace evidence for two of the late shapes (which instantiation runs, how long)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c89
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in grid64 c5odd; do
  ( cd $ROOT && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$w -o kt -- python3 profiles/scratch/late_shapes_trace.py $w > $OUT/$w.txt 2> $OUT/$w.err ) || { tail -5 $OUT/$w.err; }
  grep -v "amdgpu\|arn" $OUT/$w.txt
  f=$(find $OUT/kt_$w -name "kt_kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/r04_${w}_kernel_stats.csv && head -3 $f | cut -c1-260
done
