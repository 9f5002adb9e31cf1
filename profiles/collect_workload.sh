#!/bin/bash
# rocprofv3 evidence for ONE bench workload on the GPU box (run through gpurun from the repo root):
#   bash profiles/collect_workload.sh r02 c3 random
#   bash profiles/collect_workload.sh r02 c5_50 greedy
# Three separate rocprofv3 runs (gpurun refuses --pmc mixed with traces): kernel-trace + stats, then one
# --pmc pass per counter.  The program itself follows `--` (no env / bash -c hop).  Results land in
# gpurun_out/prof_<workload>/; the summaries to be judged are copied to profiles/ by hand afterwards:
#   <R>_<workload>_kernel_stats.csv   rocprofv3 --stats table of the kernel-trace run
#   <R>_<workload>_kernel_trace.csv   one row per rollout_kernel dispatch (start, end, duration)
#   <R>_<workload>_bench.json         bench.py's own line of the kernel-trace run (HIP events, same process)
#   <R>_<workload>_traffic.json       WRITE_SIZE + 2 x FETCH_SIZE per launch vs the algorithmic bytes
set -e
R=${1:-r02}
W=${2:-c3}
P=${3:-random}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$W
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--workload $W --policy $P --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $ROOT/bench.py $ARGS > $OUT/${R}_${W}_bench.json 2> $OUT/kt.err || { tail -5 $OUT/kt.err; exit 1; }
export CCX_BENCH_NO_SETTLE=1    # the PMC passes time nothing: no settle launches inside the counter window
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- python3 $ROOT/bench.py $ARGS --steps 2 --warmup 6 > $OUT/bench_write.json 2> $OUT/write.err || { tail -5 $OUT/write.err; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- python3 $ROOT/bench.py $ARGS --steps 2 --warmup 6 > $OUT/bench_fetch.json 2> $OUT/fetch.err || { tail -5 $OUT/fetch.err; exit 1; }
python3 - <<PY
import csv, glob, json, subprocess, sys
out = "$OUT"
line = json.loads(open(f"{out}/${R}_${W}_bench.json").read().strip().splitlines()[-1])
cfg = line["config"]
stats = glob.glob(f"{out}/kt/**/kt_kernel_stats.csv", recursive=True)
trace = glob.glob(f"{out}/kt/**/kt_kernel_trace.csv", recursive=True)
if stats:
    open(f"{out}/${R}_${W}_kernel_stats.csv", "w").write(open(stats[0]).read())
if trace:
    rows = [r for r in csv.DictReader(open(trace[0])) if "rollout_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    with open(f"{out}/${R}_${W}_kernel_trace.csv", "w") as f:
        f.write("dispatch,kernel,start_ns,end_ns,duration_us\n")
        for k, r in enumerate(rows):
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            f.write(f'{k},"{r["Kernel_Name"]}",{r["Start_Timestamp"]},{r["End_Timestamp"]},{d:.3f}\n')
    timed = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[-line["steps"]:]]
    print(f"${W}: {len(rows)} dispatches, mean of the last {len(timed)} = {sum(timed) / len(timed):.1f} us; "
          f"bench HIP events {line['roofline']['kernel_ms_per_launch'] * 1e3:.1f} us, frac {line['roofline']['frac']:.3f}")
fetch = glob.glob(f"{out}/fetch/**/fetch_counter_collection.csv", recursive=True)[0]
write = glob.glob(f"{out}/write/**/write_counter_collection.csv", recursive=True)[0]
res = subprocess.run([sys.executable, "$ROOT/profiles/derive_traffic.py", fetch, write, str(cfg["envs_per_gpu"]),
                      str(cfg["agents"]), str(cfg["env_steps_per_step"])], capture_output=True, text=True, check=True)
open(f"{out}/${R}_${W}_traffic.json", "w").write(res.stdout)
t = json.loads(res.stdout)
print(f"${W}: HBM bytes per launch {t['hbm_bytes_per_launch']:.4g} = {t['hbm_bytes_per_launch'] / t['algorithmic_bytes_per_launch']:.4f} x algorithmic")
PY
