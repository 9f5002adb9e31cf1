#!/bin/bash
# larger batches of the C2 geometry on one GPU (288 GB: the natural way to use it), final kernel
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export CCX_PACE_MEMORY=0
for EC in "32768 250" "65536 250" "131072 125"; do
  set -- $EC
  timeout -k 10 200 python3 bench.py --envs-per-gpu $1 --chunk $2 --steps 12 --warmup 24 --no-cpu-baseline --no-secondary > gpurun_out/big_$1.json 2> gpurun_out/big_$1.err
  python3 - gpurun_out/big_$1.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c = d["config"]
print("E %6d K %3d: %.3e env-steps/s  frac %.3f  pace %s  shape %s max/med %.3f" % (c["envs_per_gpu"], c["env_steps_per_step"], d["value"], d["roofline"]["frac"], c.get("step_pace_ns"), {k: c["launch_shape"][k] for k in ("waves_per_block", "writers_per_tile", "num_blocks", "resident_blocks")}, d["roofline"]["kernel_ms_max_over_median"]))
PY
done
