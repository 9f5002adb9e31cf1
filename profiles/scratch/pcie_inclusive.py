"""PCIe-inclusive rate of the batch API when the caller insists on HOST buffers: per env-step of 4096
envs, actions H2D (32 KB) + ccx_step + all outputs D2H (5.3 MB) through pinned memory."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from bench import c2_config  # noqa: E402
from collectivecrossing_amd.batched import BatchedCollectiveCrossing  # noqa: E402

E, N = 4096, 8
env = BatchedCollectiveCrossing(c2_config(), E)
env.make_reset_pool(0, 1024, on_device=True)
env.reset_from_pool()
h_act = torch.randint(0, 5, (E, N), dtype=torch.uint8).pin_memory()
d_act = torch.empty((E, N), dtype=torch.uint8, device=env.device)
res = env.step(d_act.zero_())
h_out = [torch.empty_like(t, device="cpu").pin_memory() for t in (res.obs, res.reward, res.agent_flags, res.env_flags)]


def one():
    d_act.copy_(h_act, non_blocking=True)
    r = env.step(d_act)
    for h, t in zip(h_out, (r.obs, r.reward, r.agent_flags, r.env_flags)):
        h.copy_(t, non_blocking=True)
    torch.cuda.synchronize()


for _ in range(20):
    one()
t0 = time.perf_counter()
n = 300
for _ in range(n):
    one()
dt = (time.perf_counter() - t0) / n
nbytes = sum(h.numel() * h.element_size() for h in h_out) + h_act.numel()
print(f"host-buffer step of {E} envs: {dt * 1e6:.1f} us ({E / dt:.3e} env-steps/s, {nbytes / dt / 1e9:.1f} GB/s over the host link)")
