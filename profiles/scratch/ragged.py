"""Batch sizes between the powers of two the launch-shape sweep knows: fraction of the HBM peak (rows mode) with the default shape
and with (writers, tiles per workgroup) candidates, settled like the bench's workloads.  usage: ragged.py [out.json] [workload]"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from collectivecrossing_amd.batched import BatchedCollectiveCrossing  # noqa: E402


def measure(workload, E, prep, settle=30, timed=10):
    dev = torch.device("cuda:0")
    config, _ = bench.workload_config(workload)
    env = BatchedCollectiveCrossing(config, E, device=dev)
    try:
        if prep:
            prep(env)
        N = env.num_agents
        L = 6 + 4 * N
        K = int(max(16, min(500, 2.5e9 // (E * N * L * 4))))
        env.make_reset_pool(0, 1024, on_device=True)
        env.reset_from_pool()
        gen = torch.Generator(device=dev).manual_seed(4321)
        actions = torch.randint(0, 5, (K, E, N), dtype=torch.uint8, device=dev, generator=gen)
        traj = env.alloc_rollout(K)
        for _ in range(settle):
            env.rollout(actions, auto_reset=True, out=traj)
        torch.cuda.synchronize(dev)
        ev = []
        for _ in range(timed):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); env.rollout(actions, auto_reset=True, out=traj); b.record()
            ev.append((a, b))
        torch.cuda.synchronize(dev)
        ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        nbytes = bench.rollout_bytes_per_agent_step(N) * K * E * N
        sh = env.launch_shape()
        return {"E": E, "K": K, "frac": nbytes / (ms * 1e-3) / 1e9 / bench.HBM_PEAK_GBS, "pace_ns": env.step_pace_ns(),
                "shape": (sh["lanes_per_wave"], sh["writers_per_tile"], sh["waves_per_block"], sh["num_blocks"], sh["resident_blocks"])}
    finally:
        env.close()
        torch.cuda.empty_cache()


def cand(w, t):
    return lambda e: (e.set_writers(w), e.set_launch_shape(0, t))


if __name__ == "__main__":
    workload = sys.argv[2] if len(sys.argv) > 2 else "c2"
    Es = {"c2": (2500, 3000, 4096, 5000, 6000, 7000, 8192, 10000, 12000, 14000, 16384, 20000, 24000, 28000, 40000, 50000),
          "c3": (700, 1000, 1500, 2048, 3000, 4096, 5000, 6001, 8192, 10000)}[workload]
    out, t0 = [], time.time()
    for E in Es:
        row = {}
        for name, prep in (("default", None), ("w1t1", cand(1, 1)), ("w1t2", cand(1, 2)), ("w2t1", cand(2, 1)), ("w2t2", cand(2, 2)),
                           ("w3t1", cand(3, 1)), ("w3t2", cand(3, 2))):
            try:
                r = measure(workload, E, prep)
            except Exception as exc:
                r = {"E": E, "error": repr(exc)[:100]}
            r["variant"] = name
            out.append(r)
            row[name] = r
        d = row["default"]
        best = max((r for r in row.values() if "frac" in r), key=lambda r: r["frac"])
        print(f"[{time.time() - t0:4.0f}s] {workload} E={E:6d} K={d.get('K')}: default {d.get('frac', 0):.3f} {d.get('shape')}  best {best['variant']} {best['frac']:.3f} {best['shape']} | " +
              " ".join(f"{k} {v['frac']:.3f}" if "frac" in v else f"{k} err" for k, v in row.items() if k != "default"), flush=True)
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)
