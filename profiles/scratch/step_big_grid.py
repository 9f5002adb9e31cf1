"""Graph-replayed single steps on larger grids (does the short-launch kernel still serve them?)."""
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "profiles/scratch")
import big_grid_scan as b  # noqa: E402
import step_scan  # noqa: E402
from collectivecrossing_amd.batched import BatchedCollectiveCrossing  # noqa: E402

for (w, h, n) in ((12, 8, 8), (24, 16, 8), (40, 30, 8), (64, 48, 8), (80, 60, 8), (100, 100, 8), (40, 30, 20), (100, 100, 20)):
    cfg = b.cfg(w, h, n) if (w, h) != (12, 8) else __import__("bench").c2_config()
    for E in (256, 4096):
        env = BatchedCollectiveCrossing(cfg, E, device=step_scan.dev)
        try:
            env.reset(torch.arange(E, dtype=torch.int64))
            env.use_stream(step_scan.side)
            acts = torch.randint(0, 5, (16, E, n), dtype=torch.uint8, device=step_scan.dev)
            sh = env.step_shape()
            us = step_scan.graph_us(env, acts)
            print(f"{w}x{h} N={n} E={E}: {us:.2f} us per step; step kernel ok={sh['ok']} lanes {sh['lanes_per_wave']} rows {sh['row_waves']} blocks {sh['num_blocks']} lds {sh['lds_bytes']}", flush=True)
        finally:
            env.close()
