#!/usr/bin/env python3
"""K = 1 stepping through the short-launch kernel: graph-replayed us per step over its launch-shape knobs.
usage: step_ab.py [E ...]"""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import os  # noqa: E402

import bench  # noqa: E402
if os.environ.get("CCX_DIAG_LIB"):   # an experimental build (make variant NAME=...)
    import ctypes
    from collectivecrossing_amd import _abi, _lib
    _lib.LIB_PATH = Path(os.environ["CCX_DIAG_LIB"]).resolve()
    probe = ctypes.CDLL(str(_lib.LIB_PATH))
    _abi.PROTOTYPES = {k: v for k, v in _abi.PROTOTYPES.items() if hasattr(probe, k)}
from collectivecrossing_amd.batched import BatchedCollectiveCrossing  # noqa: E402

cfg, _ = bench.workload_config(sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].isdigit() else "c2")
Es = [int(a) for a in sys.argv[1:] if a.isdigit()] or [4096]
dev = torch.device("cuda", 0)
side = torch.cuda.Stream(device=dev)


def graph_us(env, acts, want_obs, n=100, reps=20):
    with torch.cuda.stream(side):
        env.step(acts[0], want_obs=want_obs)
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            for k in range(n):
                env.step(acts[k % acts.shape[0]], want_obs=want_obs)
        graph.replay()
        side.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side)
            for _ in range(reps):
                graph.replay()
            e1.record(side)
            side.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / (reps * n))
    return best


for E in Es:
    env = BatchedCollectiveCrossing(cfg, E, device=dev)
    env.reset(torch.arange(E, dtype=torch.int64))
    N = env.num_agents
    G = 1
    while G < N:
        G *= 2
    acts = torch.randint(0, 5, (64, E, N), dtype=torch.uint8, device=dev)
    env.use_stream(side)
    rows = []
    env.set_tunable("step_kernel", 0)
    rows.append({"kernel": "rollout", "obs_us": round(graph_us(env, acts, True), 3), "noobs_us": round(graph_us(env, acts, False), 3)})
    env.set_tunable("step_kernel", 1)
    lanes_list = [0] if os.environ.get("CCX_AB_QUICK") else sorted({0} | {l for l in (G, 2 * G, 4 * G, 8 * G) if l <= 64})
    for lanes in lanes_list:
        for rw in ((0,) if os.environ.get("CCX_AB_QUICK") else (0, 1, 2, 3) if N <= 8 else (0, 2, 3, 5, 7)):
            env.set_tunable("step_lanes", lanes)
            env.set_tunable("step_rows", rw)
            sh = env.step_shape()
            rows.append({"kernel": "step", "lanes": lanes, "rows": rw, "shape": [sh["lanes_per_wave"], sh["row_waves"], sh["num_blocks"]],
                         "obs_us": round(graph_us(env, acts, True), 3), "noobs_us": round(graph_us(env, acts, False), 3)})
    for r in rows:
        print(E, json.dumps(r), flush=True)
    env.close()
