#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   bash profiles/collect.sh r01
# kernel-trace/stats pass and the two PMC passes are separate rocprofv3 runs (gpurun refuses mixes).
set -e
R=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py > $OUT/${R}_bench_default.json 2> $OUT/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $ROOT/bench.py --no-cpu-baseline > $OUT/${R}_bench_under_kernel_trace.json 2> $OUT/kt.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- python3 $ROOT/bench.py --no-cpu-baseline --steps 2 > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- python3 $ROOT/bench.py --no-cpu-baseline --steps 2 > $OUT/bench_write.json 2> $OUT/write.err
find $OUT -name "*.csv" | sort
