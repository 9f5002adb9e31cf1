#!/bin/bash
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_c8
mkdir -p $OUT
cd $ROOT
export CCX_PACE_CACHE=$OUT/pace_cache.json
B="timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-secondary"
for P in 380 400 420; do $B --envs-per-gpu 2048 --pace $P --tunable hand2=2 > $OUT/c2_2048_flags_p$P.json 2>> $OUT/err.txt; done
$B --envs-per-gpu 2048 --pace 400 > $OUT/c2_2048_barrier_p400.json 2>> $OUT/err.txt
for rep in 1 2; do
$B > $OUT/c2_default_$rep.json 2>> $OUT/err.txt
$B --tunable hand2=2 > $OUT/c2_flags_$rep.json 2>> $OUT/err.txt
done
$B --workload c5_64 --policy greedy --chunk 100 --steps 24 --warmup 24 --pool 512 --tunable hand2=2 > $OUT/c5_64_flags.json 2>> $OUT/err.txt
$B --workload c5_64 --policy greedy --chunk 100 --steps 24 --warmup 24 --pool 512 > $OUT/c5_64_default.json 2>> $OUT/err.txt
$B --workload c3 --chunk 100 --steps 24 --warmup 24 --pool 512 --tunable hand2=2 > $OUT/c3_flags.json 2>> $OUT/err.txt
$B --workload c3 --chunk 100 --steps 24 --warmup 24 --pool 512 > $OUT/c3_default.json 2>> $OUT/err.txt
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], f"{d['value']:.4g}", f"frac {d['roofline']['frac']:.3f}", f"ms/launch {d['roofline']['kernel_ms_per_launch']:.4f}", "us/step %.4f" % (d['roofline']['kernel_ms_per_launch']*1e3/d['config']['steps_per_launch']), "pace", d['config']['step_pace_ns'])
    except Exception as e:
        print(f, "ERR", e)
PY
