#!/usr/bin/env python3
"""Launch-shape sweep of the rollout kernel: for E x N x output mode, us per env-step with the
library's DEFAULT shape and with a small candidate set (lanes per wave x writer waves per tile); written as JSON.
usage: shape_sweep.py out.json [quick]"""
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import bench  # noqa: E402
from collectivecrossing_amd import configs as C  # noqa: E402
from collectivecrossing_amd.batched import BatchedCollectiveCrossing  # noqa: E402

HBM = 8000.0


def config_for(n):
    if n == 8:
        return bench.c2_config()
    if n in (1, 3, 12):
        nb = {1: 1, 3: 2, 12: 6}[n]
        return C.CollectiveCrossingConfig(width=12, height=8, division_y=4, tram_door_left=5, tram_door_right=7, tram_length=9,
                                          num_boarding_agents=nb, num_exiting_agents=n - nb, exiting_destination_area_y=0,
                                          boarding_destination_area_y=8, truncated_config=C.MaxStepsTruncatedConfig(max_steps=100))
    return bench.workload_config({32: "c3", 50: "c5_50", 64: "c5_64"}[n])[0]


def measure(env, acts, traj, K, warm=4, timed=6):
    for _ in range(warm):
        env.rollout(acts, auto_reset=True, out=traj)
    ev = []
    for _ in range(timed):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        env.rollout(acts, auto_reset=True, out=traj)
        e1.record()
        ev.append((e0, e1))
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e3 / K     # us per env-step


def main():
    out_path = sys.argv[1]
    quick = len(sys.argv) > 2
    dev = torch.device("cuda", 0)
    Ns = [8, 32] if quick else [1, 3, 8, 12, 32, 50, 64]
    Es = [1024, 4096] if quick else [256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536]
    results = []
    t_start = time.time()
    for N in Ns:
        cfg = config_for(N)
        G = 1
        while G < N:
            G *= 2
        L = 6 + 4 * N
        for E in Es:
            step_bytes = {"rows": E * N * (4 * L + 10), "compact": E * N * 26, "noobs": E * N * 10}
            env = BatchedCollectiveCrossing(cfg, E, device=dev)
            env.set_tunable("step_kernel", 0)
            env.make_reset_pool(0, 512, on_device=True)
            env.reset_from_pool()
            for mode in ("rows", "compact", "noobs"):
                # launches of ~1.5 ms at the memory rate / ~0.4 us per step for the sim-bound modes, 24..500 steps, <= 3 GB
                K = int(min(500, max(24, (1.5e-3 * 6.5e12) // step_bytes[mode] if mode == "rows" else 400)))
                if step_bytes[mode] * K > 6.0e9:
                    K = int(6.0e9 // step_bytes[mode])
                if K < 8:
                    continue
                acts = torch.randint(0, 5, (K, E, N), dtype=torch.uint8, device=dev)
                traj = env.alloc_rollout(K, want_obs=(mode == "rows"), want_compact=(mode == "compact"))
                env.set_launch_shape(0, 0)
                env.set_writers(0)
                rec = {"N": N, "E": E, "mode": mode, "K": K, "bytes_per_step": step_bytes[mode]}
                us = measure(env, acts, traj, K)
                rec["default"] = {"shape": env.launch_shape(), "us_per_env_step": round(us, 4),
                                  "frac": round(step_bytes[mode] / (us * 1e-6) / 1e9 / HBM, 4)}
                cands = []
                lanes_set = [l for l in (G, 2 * G, 4 * G, 8 * G, 16 * G, 32 * G, 64 * G) if min(64, max(G, 8)) <= l <= 64]
                for lanes in lanes_set:
                    for writers in (0, 1, 2, 3, 4):
                        try:
                            env.set_launch_shape(lanes, 0)
                            env.set_writers(writers)
                            us_c = measure(env, acts, traj, K, warm=3, timed=4)
                            cands.append({"lanes": lanes, "writers": writers, "shape": env.launch_shape(), "us_per_env_step": round(us_c, 4)})
                        except Exception as exc:
                            cands.append({"lanes": lanes, "writers": writers, "error": repr(exc)[:80]})
                rec["candidates"] = cands
                ok = [c for c in cands if "us_per_env_step" in c]
                best = min(ok, key=lambda c: c["us_per_env_step"]) if ok else None
                rec["best"] = best
                rec["default_over_best"] = round(us / best["us_per_env_step"], 3) if best else None
                results.append(rec)
                print(f"[{time.time() - t_start:6.0f}s] N={N} E={E} {mode} K={K}: default {us:.3f} us ({rec['default']['shape']['lanes_per_wave']} lanes, "
                      f"{rec['default']['shape']['writers_per_tile']} writers, frac {rec['default']['frac']:.3f}); best {best['us_per_env_step']:.3f} "
                      f"({best['lanes']} lanes, {best['writers']} writers) ratio {rec['default_over_best']}", flush=True)
                del traj, acts
                Path(out_path).write_text(json.dumps(results))
            env.close()
            torch.cuda.empty_cache()
    Path(out_path).write_text(json.dumps(results))


if __name__ == "__main__":
    main()
