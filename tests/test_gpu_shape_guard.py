"""Launch-shape guard: `choose_shape` is a set of rules derived from a measured E x N x output-mode sweep
(profiles/r04_shape_sweep.json, profiles/scratch/shape_sweep.py).  On a coarse sub-grid of that sweep the library's DEFAULT
shape must stay within 8 % of the best of a small candidate set (lanes per wave x writer waves per tile; 10 % for batches
of at most 128 full tiles, where the half-tile rule is a measured compromise: 2-4 % better for 1 / 8 agents, 6-8 % worse for
3 / 12) and a batch size must not fall off a cliff between its neighbours: throughput(E) >= 0.88 x min(throughput(E / 2),
throughput(2 E)).

Timing test, so every failing point is measured a second time before it counts (a shared box can lose a few per cent
between two measurements); parity of every shape is the job of test_gpu_parity / test_gpu_round2, not of this file."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ES = [1024, 2048, 4096, 8192, 16384]
NS = [3, 8, 32]


def _config(n):
    import bench
    from collectivecrossing_amd import configs as C
    if n == 8:
        return bench.c2_config()
    if n == 32:
        return bench.workload_config("c3")[0]
    return C.CollectiveCrossingConfig(width=12, height=8, division_y=4, tram_door_left=5, tram_door_right=7, tram_length=9,
                                      num_boarding_agents=2, num_exiting_agents=n - 2, exiting_destination_area_y=0,
                                      boarding_destination_area_y=8, truncated_config=C.MaxStepsTruncatedConfig(max_steps=100))


def _measure(torch, env, acts, traj, warm=5, timed=6):
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
    return float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e3 / acts.shape[0]


@pytest.mark.parametrize("n", NS)
@pytest.mark.parametrize("mode", ["rows", "noobs"])
def test_default_shape_is_near_the_best_candidate_and_has_no_cliffs(n, mode):
    import torch

    from collectivecrossing_amd.batched import BatchedCollectiveCrossing
    cfg = _config(n)
    L = 6 + 4 * n
    us_default = {}
    problems = []
    for E in ES:
        step_bytes = E * n * ((4 * L + 10) if mode == "rows" else 10)
        K = int(min(400, max(24, (1.5e-3 * 6.5e12) // step_bytes if mode == "rows" else 300)))
        if step_bytes * K > 4.0e9:
            K = int(4.0e9 // step_bytes)
        env = BatchedCollectiveCrossing(cfg, E)
        env.set_tunable("step_kernel", 0)
        env.make_reset_pool(0, 256)
        env.reset_from_pool()
        acts = torch.randint(0, 5, (K, E, n), dtype=torch.uint8, device=env.device)
        traj = env.alloc_rollout(K, want_obs=(mode == "rows"))
        us = _measure(torch, env, acts, traj)
        best, best_shape = float("inf"), None
        for lanes in (32, 64):
            for writers in (1, 2, 3, 4):
                env.set_launch_shape(lanes, 0)
                env.set_writers(writers)
                c = _measure(torch, env, acts, traj, warm=3, timed=4)
                if c < best:
                    best, best_shape = c, (lanes, writers)
        g_lanes = 1
        while g_lanes < n:
            g_lanes *= 2
        tol = 1.10 if E * g_lanes <= 128 * 64 else 1.08
        if us > tol * best:                                   # measure both again before it counts
            env.set_launch_shape(0, 0)
            env.set_writers(0)
            us = min(us, _measure(torch, env, acts, traj))
            env.set_launch_shape(best_shape[0], 0)
            env.set_writers(best_shape[1])
            best = max(best, _measure(torch, env, acts, traj))
            if us > tol * best:
                problems.append(f"E={E}: default {us:.3f} us per env-step vs {best:.3f} with (lanes, writers) = {best_shape}")
        us_default[E] = us
        env.close()
        del traj, acts
        torch.cuda.empty_cache()
    thr = {E: E / us_default[E] for E in ES}
    for lo, E, hi in zip(ES, ES[1:], ES[2:]):
        if thr[E] < 0.88 * min(thr[lo], thr[hi]):
            problems.append(f"cliff at E={E}: {thr[E]:.0f} envs/us vs {thr[lo]:.0f} at {lo} and {thr[hi]:.0f} at {hi}")
    assert not problems, (n, mode, problems, us_default)


RAGGED = [("c2", 3000), ("c2", 5000), ("c2", 10000), ("c2", 12000), ("c2", 20000), ("c3", 3000), ("c3", 6001)]


@pytest.mark.parametrize("workload,E", RAGGED)
def test_batch_sizes_between_the_powers_of_two_have_no_cliff(workload, E):
    """The sweep and the test above know powers of two.  Between them a rule's boundary can sit in the wrong place: "two writers
    while three waves per tile fit one round" put 8193 .. 10 920 envs of C2 on six-wave workgroups, three to a CU, at 0.49 of
    the peak (profiles/r04_ragged_c2.txt).  Settled like the bench's workloads, the default shape of a ragged batch stays
    within 12 % of the best of (writers, tiles per workgroup) in {1, 2, 3} x {1, 2} and above 0.70 of the HBM peak."""
    import bench
    import torch

    from collectivecrossing_amd.batched import BatchedCollectiveCrossing
    cfg = bench.workload_config(workload)[0]

    def run(prep, settle=24, timed=8):
        env = BatchedCollectiveCrossing(cfg, E)
        try:
            if prep:
                env.set_writers(prep[0])
                env.set_launch_shape(0, prep[1])
            n = env.num_agents
            K = int(max(16, min(500, 2.0e9 // (E * n * (6 + 4 * n) * 4))))
            env.make_reset_pool(0, 512)
            env.reset_from_pool()
            acts = torch.randint(0, 5, (K, E, n), dtype=torch.uint8, device=env.device)
            traj = env.alloc_rollout(K)
            for _ in range(settle):
                env.rollout(acts, auto_reset=True, out=traj)
            ms = _measure(torch, env, acts, traj, warm=0, timed=timed) * K * 1e-3
            return bench.rollout_bytes_per_agent_step(n) * K * E * n / (ms * 1e-3) / 1e9 / bench.HBM_PEAK_GBS, env.launch_shape()
        finally:
            env.close()
            torch.cuda.empty_cache()

    frac, shape = run(None)
    cands = {(w, t): run((w, t))[0] for w in (1, 2, 3) for t in (1, 2)}
    best = max(cands, key=cands.get)
    if frac < 0.88 * cands[best] or frac < 0.70:                 # measured again before it counts
        frac = max(frac, run(None)[0])
    assert frac >= 0.88 * cands[best] and frac >= 0.70, (workload, E, frac, shape, best, cands)
