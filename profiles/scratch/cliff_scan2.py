"""More dense scans with the default shapes: short launches (K = 1, 8) of the step kernel, compact rows, the fused greedy policy,
other agent counts.  usage: cliff_scan2.py out.json"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "profiles/scratch")
import bench  # noqa: E402
import cliff_scan  # noqa: E402
from collectivecrossing_amd import configs as C  # noqa: E402
from collectivecrossing_amd.batched import BatchedCollectiveCrossing  # noqa: E402


def config_for(n):
    if n in (8, 32, 50, 64):
        return bench.c2_config() if n == 8 else bench.workload_config({32: "c3", 50: "c5_50", 64: "c5_64"}[n])[0]
    nb = (n + 1) // 2
    return C.CollectiveCrossingConfig(width=12, height=8, division_y=4, tram_door_left=5, tram_door_right=7, tram_length=9,
                                      num_boarding_agents=nb, num_exiting_agents=n - nb, exiting_destination_area_y=0,
                                      boarding_destination_area_y=8, truncated_config=C.MaxStepsTruncatedConfig(max_steps=100))


def short(cfg, E, N, K, n_launch=200):
    dev = torch.device("cuda:0")
    env = BatchedCollectiveCrossing(cfg, E, device=dev)
    try:
        env.make_reset_pool(0, 256, on_device=True)
        env.reset_from_pool()
        acts = torch.randint(0, 5, (K, E, N), dtype=torch.uint8, device=dev)
        traj = env.alloc_rollout(K)
        for _ in range(20):
            env.rollout(acts, auto_reset=True, out=traj)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n_launch):
            env.rollout(acts, auto_reset=True, out=traj)
        b.record()
        torch.cuda.synchronize()
        return {"E": E, "N": N, "K": K, "us_per_launch": a.elapsed_time(b) * 1e3 / n_launch, "step_shape": env.step_shape()}
    finally:
        env.close()
        torch.cuda.empty_cache()


def other(cfg, E, N, mode, settle=16, timed=8):
    dev = torch.device("cuda:0")
    K = int(min(400, max(24, 2.0e9 // (E * N * (6 + 4 * N) * 4)))) if mode == "greedy" else 300
    env = BatchedCollectiveCrossing(cfg, E, device=dev)
    try:
        env.make_reset_pool(0, 256, on_device=True)
        env.reset_from_pool()
        traj = env.alloc_rollout(K, want_obs=(mode == "greedy"), want_compact=(mode == "compact"))
        acts = torch.randint(0, 5, (K, E, N), dtype=torch.uint8, device=dev) if mode == "compact" else None

        def go():
            if mode == "greedy":
                env.rollout_greedy(K, auto_reset=True, out=traj, want_actions=False)
            else:
                env.rollout(acts, auto_reset=True, out=traj)
        for _ in range(settle):
            go()
        ev = []
        for _ in range(timed):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); go(); b.record()
            ev.append((a, b))
        torch.cuda.synchronize()
        us = float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e3 / K
        return {"E": E, "N": N, "mode": mode, "us_per_env_step": us, "envs_per_us": E / us}
    finally:
        env.close()
        torch.cuda.empty_cache()


if __name__ == "__main__":
    out, t0 = [], time.time()
    Es = sorted({int(round(256 * 1.125 ** k / 16) * 16) for k in range(0, 48)})
    Es = [e for e in Es if e <= 70000]
    for N, K in ((8, 1), (8, 8), (3, 1), (32, 1)):
        cfg = config_for(N)
        rows = []
        for E in Es:
            if E * N * (6 + 4 * N) * 4 * K > 3.0e9:
                continue
            try:
                rows.append(short(cfg, E, N, K))
            except Exception as exc:
                print("error", N, K, E, repr(exc)[:100], flush=True)
        out.extend(rows)
        print(f"[{time.time() - t0:4.0f}s] step kernel N={N} K={K} (us per launch): " + " ".join(f"{r['E']}:{r['us_per_launch']:.2f}" for r in rows), flush=True)
        print("   shapes: " + " ".join(f"{r['E']}:{r['step_shape']['lanes_per_wave']}/{r['step_shape']['row_waves']}" for r in rows[::4]), flush=True)
    for N, mode in ((8, "compact"), (8, "greedy"), (50, "greedy"), (2, "rows"), (5, "rows"), (16, "rows")):
        cfg = config_for(N)
        rows = []
        for E in Es:
            if mode != "compact" and E * N * (6 + 4 * N) * 4 * 24 > 5.5e9:
                continue
            try:
                rows.append(cliff_scan.measure(cfg, E, N, "rows") if mode == "rows" else other(cfg, E, N, mode))
            except Exception as exc:
                print("error", N, mode, E, repr(exc)[:100], flush=True)
        out.extend(rows)
        thr = [r["envs_per_us"] for r in rows]
        for i in range(1, len(rows) - 1):
            if thr[i] < 0.88 * min(thr[i - 1], thr[i + 1]):
                print(f"   DIP N={N} {mode} E={rows[i]['E']}: {thr[i]:.0f} envs/us vs {thr[i - 1]:.0f} / {thr[i + 1]:.0f}", flush=True)
        print(f"[{time.time() - t0:4.0f}s] N={N} {mode}: " + " ".join(f"{r['E']}:{(r.get('frac') or r['us_per_env_step']):.3f}" for r in rows), flush=True)
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"))
