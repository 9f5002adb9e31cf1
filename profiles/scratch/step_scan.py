"""Graph-replayed single steps (device time per step) over batch sizes and the step kernel's lanes-per-wave knob."""
import sys
import time

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "profiles/scratch")
import cliff_scan2  # noqa: E402
from collectivecrossing_amd.batched import BatchedCollectiveCrossing  # noqa: E402

dev = torch.device("cuda", 0)
side = torch.cuda.Stream(device=dev)


def graph_us(env, acts, n=100, reps=10):
    with torch.cuda.stream(side):
        env.step(acts[0])
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            for k in range(n):
                env.step(acts[k % acts.shape[0]])
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


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    cfg = cliff_scan2.config_for(N)
    for E in (512, 1024, 1504, 1680, 2128, 2704, 3040, 3424, 4320, 8192, 16384, 32768):
        res = {}
        for lanes in (0, 8, 16, 32, 64):
            env = BatchedCollectiveCrossing(cfg, E, device=dev)
            try:
                env.reset(torch.arange(E, dtype=torch.int64))
                env.use_stream(side)
                if lanes:
                    env.set_tunable("step_lanes", lanes)
                acts = torch.randint(0, 5, (16, E, N), dtype=torch.uint8, device=dev)
                sh = env.step_shape()
                res[lanes] = (round(graph_us(env, acts), 2), sh["lanes_per_wave"], sh["row_waves"], sh["num_blocks"])
            except Exception as exc:
                res[lanes] = repr(exc)[:60]
            finally:
                env.close()
        print(f"N={N} E={E}: " + "  ".join(f"{('auto' if k == 0 else k)}: {v}" for k, v in res.items()), flush=True)
