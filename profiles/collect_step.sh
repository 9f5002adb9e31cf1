#!/bin/bash
# rocprofv3 evidence for the SHORT-LAUNCH kernel (csrc/ccx_step.hip) on the GPU box (run through gpurun from the repo root):
#   bash profiles/collect_step.sh r04
# kernel-trace + stats of profiles/scratch/step_k1.py (400 eager + 2 x 2000 graph-replayed single steps at 4096 envs), then one
# --pmc pass per counter (gpurun refuses --pmc mixed with traces); the program itself follows `--`.
set -e
R=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_step
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $ROOT/profiles/scratch/step_k1.py 4096 > $OUT/${R}_step_k1_under_trace.txt 2> $OUT/kt.err || { tail -5 $OUT/kt.err; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- python3 $ROOT/profiles/scratch/step_k1.py 4096 > /dev/null 2> $OUT/write.err || { tail -5 $OUT/write.err; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- python3 $ROOT/profiles/scratch/step_k1.py 4096 > /dev/null 2> $OUT/fetch.err || { tail -5 $OUT/fetch.err; exit 1; }
python3 - <<PY
import csv, glob, json
out = "$OUT"
stats = glob.glob(f"{out}/kt/**/kt_kernel_stats.csv", recursive=True)
if stats:
    open(f"{out}/${R}_step_k1_kernel_stats.csv", "w").write(open(stats[0]).read())
trace = glob.glob(f"{out}/kt/**/kt_kernel_trace.csv", recursive=True)
res = {}
if trace:
    rows = [r for r in csv.DictReader(open(trace[0])) if "step_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    gaps = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(rows, rows[1:])]
    with_obs = [d for r, d in zip(rows, dur)]
    res["dispatches"] = len(rows)
    res["kernel_us_mean_all"] = sum(dur) / len(dur)
    srt = sorted(dur)
    res["kernel_us_median"] = srt[len(srt) // 2]
    g = sorted(x for x in gaps if x < 50)
    res["gap_to_next_dispatch_us_median"] = g[len(g) // 2] if g else None
def per_launch(path, counter):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if "step_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)
w = glob.glob(f"{out}/write/**/write_counter_collection.csv", recursive=True)
f = glob.glob(f"{out}/fetch/**/fetch_counter_collection.csv", recursive=True)
if w and f:
    wk, nw = per_launch(w[0], "WRITE_SIZE")
    fk, nf = per_launch(f[0], "FETCH_SIZE")
    E, N = 4096, 8
    L = 6 + 4 * N
    res.update({"envs": E, "agents": N, "WRITE_SIZE_KB_per_launch_mean_over_obs_and_no_obs_launches": wk, "FETCH_SIZE_KB_per_launch": fk,
                "dispatches_write_pass": nw, "dispatches_fetch_pass": nf,
                "algorithmic_bytes_per_step_with_rows": E * N * (4 * L + 1 + 8 + 1 + 22) + 6 * E,
                "algorithmic_bytes_per_step_without_rows": E * N * (1 + 8 + 1 + 22) + 6 * E,
                "note": "step_k1.py launches half of its steps with observation rows and half without; a step also reads and writes "
                        "the state (22 B per agent: the SURVEY 8d figure of 16N + 54 + 4 applies to single-step launches); FETCH_SIZE x 2 "
                        "per the gfx950 correction of MI355X_MICROARCH.md"})
    if wk is not None and fk is not None:
        res["hbm_bytes_per_launch_mean"] = wk * 1024 + 2 * fk * 1024
json.dump(res, open(f"{out}/${R}_step_k1_profile.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
