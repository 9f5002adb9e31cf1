"""Dense scan of batch sizes with the DEFAULT launch shape: throughput (envs per us of one env-step) over a geometric grid of E,
per agent count and output mode; prints dips against the neighbours.  usage: cliff_scan.py out.json"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "profiles/scratch")
import shape_sweep  # noqa: E402  (config_for)
from collectivecrossing_amd.batched import BatchedCollectiveCrossing  # noqa: E402


def measure(cfg, E, N, mode, settle=16, timed=8):
    dev = torch.device("cuda:0")
    L = 6 + 4 * N
    step_bytes = E * N * ((4 * L + 10) if mode == "rows" else 10)
    K = int(min(400, max(24, 2.0e9 // step_bytes if mode == "rows" else 300)))
    env = BatchedCollectiveCrossing(cfg, E, device=dev)
    try:
        env.set_tunable("step_kernel", 0)
        env.make_reset_pool(0, 256, on_device=True)
        env.reset_from_pool()
        acts = torch.randint(0, 5, (K, E, N), dtype=torch.uint8, device=dev)
        traj = env.alloc_rollout(K, want_obs=(mode == "rows"))
        for _ in range(settle):
            env.rollout(acts, auto_reset=True, out=traj)
        ev = []
        for _ in range(timed):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); env.rollout(acts, auto_reset=True, out=traj); b.record()
            ev.append((a, b))
        torch.cuda.synchronize()
        us = float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e3 / K
        sh = env.launch_shape()
        return {"E": E, "N": N, "mode": mode, "K": K, "us_per_env_step": us, "envs_per_us": E / us,
                "frac": step_bytes / (us * 1e-6) / 8e12 if mode == "rows" else None, "pace_ns": env.step_pace_ns(),
                "shape": (sh["lanes_per_wave"], sh["writers_per_tile"], sh["waves_per_block"], sh["num_blocks"], sh["resident_blocks"])}
    finally:
        env.close()
        torch.cuda.empty_cache()


if __name__ == "__main__":
    out, t0 = [], time.time()
    Es = sorted({int(round(256 * 1.125 ** k / 8) * 8) for k in range(0, 48)})
    Es = [e for e in Es if e <= 70000]
    for N in (8, 3, 12, 1, 32):
        cfg = shape_sweep.config_for(N)
        for mode in ("rows", "noobs"):
            rows = []
            for E in Es:
                if mode == "rows" and E * N * (6 + 4 * N) * 4 * 24 > 5.5e9:
                    continue
                try:
                    rows.append(measure(cfg, E, N, mode))
                except Exception as exc:
                    print("error", N, mode, E, repr(exc)[:100], flush=True)
            out.extend(rows)
            thr = [r["envs_per_us"] for r in rows]
            for i in range(1, len(rows) - 1):
                ref = min(thr[i - 1], thr[i + 1])
                if thr[i] < 0.88 * ref:
                    r = rows[i]
                    print(f"[{time.time() - t0:4.0f}s] DIP N={N} {mode} E={r['E']}: {thr[i]:.0f} envs/us vs {thr[i - 1]:.0f} at {rows[i - 1]['E']} and {thr[i + 1]:.0f} at {rows[i + 1]['E']}; "
                          f"shape {r['shape']} pace {r['pace_ns']:.0f} frac {r['frac']}", flush=True)
            line = " ".join(f"{r['E']}:{(r['frac'] if r['frac'] else r['us_per_env_step']):.3f}" for r in rows)
            print(f"[{time.time() - t0:4.0f}s] N={N} {mode} ({'frac of peak' if mode == 'rows' else 'us per env-step'}): {line}", flush=True)
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"))
