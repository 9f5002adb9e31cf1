"""Batches of several rounds of workgroups: fraction of the HBM peak by steps per launch, pacing and launch shape.
Usage: python3 profiles/scratch/multi_round.py [out.json]   (GPU; ~1 min)"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from collectivecrossing_amd.batched import BatchedCollectiveCrossing  # noqa: E402


def measure(E, chunk, prep=None, settle=30, timed=10):
    dev = torch.device("cuda:0")
    config, _ = bench.workload_config("c2")
    env = BatchedCollectiveCrossing(config, E, device=dev)
    try:
        if prep:
            prep(env)
        N = env.num_agents
        env.make_reset_pool(0, 1024, on_device=True)
        env.reset_from_pool()
        gen = torch.Generator(device=dev).manual_seed(4321)
        actions = torch.randint(0, 5, (chunk, E, N), dtype=torch.uint8, device=dev, generator=gen)
        traj = env.alloc_rollout(chunk)
        for _ in range(settle):
            env.rollout(actions, auto_reset=True, out=traj)
        torch.cuda.synchronize(dev)
        ev = []
        for _ in range(timed):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); env.rollout(actions, auto_reset=True, out=traj); b.record()
            ev.append((a, b))
        torch.cuda.synchronize(dev)
        ms = [a.elapsed_time(b) for a, b in ev]
        nbytes = bench.rollout_bytes_per_agent_step(N) * chunk * E * N
        sh = env.launch_shape()
        return {"E": E, "K": chunk, "frac": nbytes / (float(np.mean(ms)) * 1e-3) / 1e9 / bench.HBM_PEAK_GBS,
                "frac_best": nbytes / (float(np.min(ms)) * 1e-3) / 1e9 / bench.HBM_PEAK_GBS,
                "ms": float(np.mean(ms)), "pace_ns": env.step_pace_ns(),
                "shape": (sh["lanes_per_wave"], sh["writers_per_tile"], sh["waves_per_block"], sh["num_blocks"], sh["resident_blocks"])}
    finally:
        env.close()
        torch.cuda.empty_cache()


VARIANTS = {
    "default": None,                                                        # (by rounds beyond ~3.5 GB of rows since call 29)
    "one_launch": lambda e: e.set_tunable("round_launches", 0),
    "by_rounds": lambda e: e.set_tunable("round_launches", 2),
    "pace_off": lambda e: e.set_step_pace(-1),
    "w3_tpb1": lambda e: (e.set_writers(3), e.set_launch_shape(0, 1)),
}

if __name__ == "__main__":
    out = []
    t0 = time.time()
    for E in (5000, 20000, 32768, 40000, 65536, 100003):
        for K in (32, 64, 128, 256):
            if E * K > 65536 * 128 or (E not in (32768, 65536) and K != 64):
                continue
            for name, prep in VARIANTS.items():
                try:
                    r = measure(E, K, prep)
                except Exception as exc:
                    r = {"E": E, "K": K, "error": repr(exc)}
                r["variant"] = name
                out.append(r)
                print(f"[{time.time() - t0:5.0f}s] {E:6d} x {K:3d} {name:18s} " +
                      (r.get("error") or f"frac {r['frac']:.3f} (best launch {r['frac_best']:.3f}) {r['ms']:.3f} ms pace {r['pace_ns']:.0f} ns shape {r['shape']}"), flush=True)
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)
