"""Derive profiles/rNN_traffic.json from the two separate rocprofv3 --pmc passes.

usage: python profiles/derive_traffic.py <fetch_counter.csv> <write_counter.csv> <envs> <agents> <chunk> > profiles/r01_traffic.json

Units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE
count KiB; FETCH_SIZE is doubled (gfx950 reports half of wide coalesced reads -- for our byte loads of
actions that is the conservative side).  Per launch = mean over the rollout_kernel dispatches.
"""
import csv
import json
import sys


def per_launch(path, counter):
    vals = []
    with open(path) as f:
        for row in csv.DictReader(f):
            if "rollout_kernel" in row["Kernel_Name"] and row["Counter_Name"] == counter:
                vals.append(float(row["Counter_Value"]))
    if not vals:
        raise SystemExit(f"no {counter} rows for rollout_kernel in {path}")
    return sum(vals) / len(vals), len(vals)


def main():
    fetch_csv, write_csv, envs, agents, chunk = sys.argv[1], sys.argv[2], *map(int, sys.argv[3:6])
    fk, nf = per_launch(fetch_csv, "FETCH_SIZE")
    wk, nw = per_launch(write_csv, "WRITE_SIZE")
    obs_len = 6 + 4 * agents
    algorithmic = chunk * envs * agents * (4 * obs_len + 8 + 1 + 1)   # obs f32, reward f64, flag, action (bench.py)
    out = {
        "envs": envs, "agents": agents, "chunk": chunk,
        "dispatches_fetch_pass": nf, "dispatches_write_pass": nw,
        "FETCH_SIZE_KB_per_launch": fk, "WRITE_SIZE_KB_per_launch": wk,
        "fetch_bytes_raw": fk * 1024, "fetch_bytes_x2_gfx950_correction": 2 * fk * 1024,
        "write_bytes": wk * 1024,
        "hbm_bytes_per_launch": wk * 1024 + 2 * fk * 1024,
        "algorithmic_bytes_per_launch": algorithmic,
        "note": "separate --pmc passes (FETCH_SIZE, WRITE_SIZE), counter unit KiB; FETCH_SIZE doubled per "
                "MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads; our reads are byte loads "
                "of actions, uncalibrated, so x2 is the conservative side); WRITE_SIZE is exact for "
                "16-B-per-lane streaming stores",
    }
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
