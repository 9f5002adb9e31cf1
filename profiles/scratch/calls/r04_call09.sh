#!/bin/bash
# round 4, call 9: did the reward-table / terminated-bit changes cost the hot kernels anything?  HEAD's library vs the working tree's,
# interleaved, three repeats: headline (rows) and the sim-bound no-obs rollout; then the step kernel's stamps
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c09
mkdir -p $OUT
cd $ROOT
D=collectivecrossing_amd/csrc/_diag
for rep in 1 2 3; do
  for v in head cur; do
    if [ $v = cur ]; then unset CCX_DIAG_LIB; else export CCX_DIAG_LIB=$D/libccx_$v.so; fi
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-secondary > $OUT/rows_${v}_$rep.json 2>> $OUT/err.txt || echo fail
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-secondary --no-obs --steps 20 --warmup 20 > $OUT/noobs_${v}_$rep.json 2>> $OUT/err.txt || echo fail
  done
done
unset CCX_DIAG_LIB
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f.split("/")[-1], f"{d['value']:.4g}", "frac %.4f" % d["roofline"]["frac"], "us/env-step %.4f" % (d["roofline"]["kernel_ms_per_launch"] * 1e3 / d["config"]["steps_per_launch"]))
PY
timeout -k 10 120 python3 profiles/scratch/step_tstamps.py 4096 > $OUT/tstamps.txt 2>&1; grep -v amdgpu $OUT/tstamps.txt
CCX_AB_QUICK=1 timeout -k 10 120 python3 profiles/scratch/step_ab.py 4096 64 2>/dev/null | grep '"step"'
