#!/usr/bin/env python3
"""Throughput bench of the batched CollectiveCrossing step on MI355X (driver contract).

Workload (BASELINE.json configs[1], SURVEY 8d "C2"): per GPU 4096 envs x 8 agents (5 boarding +
3 exiting) on the 12x8 grid, DefaultReward + DefaultObservation, IndividualAtDestination,
MaxSteps=100, uniform random actions read from a device tensor, auto-reset from a pool of
reference-exact seeded placements.  A bench "step" = ONE pass of the hot path over one batch of
synthetic input: one fused ``ccx_rollout`` launch that advances every env of the batch by `--chunk`
(default 500) env-steps from an action tensor [chunk, E, N] resident in HBM and writes the full
per-step outputs (observations f32 [E,N,L], rewards f64, flag bytes) of every env-step to a trajectory
buffer in HBM.  `--steps K` / `--warmup W` count such launches (K = 40: 20000 env-steps of 4096 envs);
the metric stays env-steps/s = K * chunk * envs / elapsed.  The first ~30 launches of a process run
slower (clock ramp, pace controller): W defaults to 80, and a smaller W is topped up by untimed set-up
launches that the JSON line reports as config.settle_launches.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

Prints ONE JSON line on rank 0.  `value` = env-steps/s over ALL ranks with inputs resident in HBM,
timed between barrier + synchronize pairs, max over ranks.  Weak scaling: every rank owns 4096
envs of a global batch of N*4096; envs are independent, the only collective is the 48-byte
all-reduce of the counters after the timed window.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable


def c2_config(max_steps: int = 100):
    from collectivecrossing_amd.configs import CollectiveCrossingConfig, MaxStepsTruncatedConfig
    return CollectiveCrossingConfig(
        width=12, height=8, division_y=4, tram_door_left=5, tram_door_right=7, tram_length=9,
        num_boarding_agents=5, num_exiting_agents=3, exiting_destination_area_y=0,
        boarding_destination_area_y=8, truncated_config=MaxStepsTruncatedConfig(max_steps=max_steps))


def workload_config(name: str):
    """Named workloads: c2 is THE bench line (BASELINE configs[1]); the others are the remaining
    BASELINE configs, available for diagnostics (`--workload`), never the default."""
    from collectivecrossing_amd import configs as C
    if name == "c2":
        return c2_config(), 4096
    if name == "c3":   # configs[2]: 20x12, 16+16, SimpleDistance (dense collisions)
        return C.CollectiveCrossingConfig(
            width=20, height=12, division_y=6, tram_door_left=6, tram_door_right=10, tram_length=16,
            num_boarding_agents=16, num_exiting_agents=16, exiting_destination_area_y=0,
            boarding_destination_area_y=12,
            reward_config=C.SimpleDistanceRewardConfig(distance_penalty_factor=0.1),
            truncated_config=C.MaxStepsTruncatedConfig(max_steps=100)), 4096
    if name in ("c5_50", "c5_64"):   # configs[4] geometry: 32x16, AllAtDestination, MaxSteps=500
        nb = 25 if name == "c5_50" else 32
        kw = dict(width=32, height=16, division_y=8, tram_door_left=10, tram_door_right=16,
                  tram_length=26, num_boarding_agents=nb, num_exiting_agents=nb,
                  exiting_destination_area_y=0, boarding_destination_area_y=16,
                  terminated_config=C.AllAtDestinationTerminatedConfig(),
                  truncated_config=C.MaxStepsTruncatedConfig(max_steps=500),
                  observation_config=C.DefaultObservationConfig(), reward_config=C.DefaultRewardConfig(),
                  render_mode=None)
        if nb == 32:   # 64 agents exceed the reference's limit of 50 (configs.py:166): skip validation
            return C.CollectiveCrossingConfig.model_construct(**kw), 1024
        return C.CollectiveCrossingConfig(**kw), 1024
    raise SystemExit(f"unknown workload {name}")


def rollout_bytes_per_agent_step(n_agents: int) -> int:
    """Bytes one fused rollout launch MUST move per agent-step (state stays in registers):
    observation row 4*(6+4N) + action 1 + reward f64 8 + agent flag byte 1 (+ 1/N env flag byte,
    ignored).  SURVEY 8d's contract figure 16N+54 additionally counts a state read+write per step
    (22 B) that the fused kernel does not do; both are reported."""
    return 4 * (6 + 4 * n_agents) + 1 + 8 + 1


def write_bandwidth_probe(dev, buf: torch.Tensor) -> float:
    """Achievable pure-WRITE bandwidth of THIS process in GB/s: best of 7 device fills of the very
    buffer the rollout writes its observations to (the sustained write rate varies by 20-30 %
    between boxes and allocations; SURVEY 8d asks for a measured denominator next to the peak)."""
    buf = buf.view(-1)
    nbytes = buf.numel() * buf.element_size()
    best = float("inf")
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        buf.fill_(1.0)
        e1.record()
        torch.cuda.synchronize(dev)
        best = min(best, e0.elapsed_time(e1))
    return nbytes / (best * 1e-3) / 1e9


def cpu_baseline(config, n_agents: int, seconds_target: float = 12.0) -> dict:
    """The CPU oracle (a C port of the reference's sequential algorithm) on the host cores, same
    workload (full trajectory outputs), bounded sample."""
    from collectivecrossing_amd.params import lower_config
    from collectivecrossing_amd.reset import build_reset_pool
    from oracle import oracle as ref  # cpu_baseline leg only

    params = lower_config(config)
    pool = build_reset_pool(config, 0, 128)
    cores = min(os.cpu_count() or 1, 16)
    E_t, K = 64, 500  # per thread: 64 envs x 500 steps, trajectory 39 MB

    def make(seed):
        b = ref.OracleBatch(params, E_t)
        b.set_reset_pool(pool)
        b.reset_from_pool()
        acts = np.random.default_rng(seed).integers(0, 5, size=(K, E_t, n_agents), dtype=np.uint8)
        return b, acts

    # single core
    b, acts = make(0)
    b.rollout(acts[:50], auto_reset=True)  # warm-up
    t0 = time.perf_counter()
    reps1 = 0
    while time.perf_counter() - t0 < seconds_target * 0.4:
        b.rollout(acts, auto_reset=True)
        reps1 += 1
    dt1 = time.perf_counter() - t0
    single = reps1 * E_t * K / dt1
    # all cores: one thread per core, each on its own envs (ctypes releases the GIL)
    workers = [make(100 + i) for i in range(cores)]
    reps = max(1, int(reps1 * 1.2))

    def run(w):
        for _ in range(reps):
            w[0].rollout(w[1], auto_reset=True)

    threads = [threading.Thread(target=run, args=(w,)) for w in workers]
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    dtm = time.perf_counter() - t0
    multi = cores * reps * E_t * K / dtm
    return {"value": multi, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "single_core_value": single,
            "sample": (f"oracle/ccx_oracle.c ccxo_rollout, same config/outputs (full trajectory), "
                       f"{cores} threads x {reps} x ({E_t} envs x {K} steps) in {dtm:.1f}s; "
                       f"1 thread: {reps1} x ({E_t} x {K}) in {dt1:.1f}s"),
            "reference_python": reference_python_speed()}


def reference_python_speed():
    """The pure-Python reference cannot travel to the GPU box; its speed was recorded in the build
    container by tests/golden/ref_speed.py (data file, read here for the record only)."""
    f = ROOT / "profiles" / "r01_reference_python_speed.json"
    try:
        d = json.loads(f.read_text())
        return {"value": d["C1_12x8_5+3"]["env_steps_per_sec_one_core"], "unit": "env-steps/s", "cores": 1,
                "where": "build container (not the GPU box), tests/golden/ref_speed.py, time inside env.step only",
                "config": "BASELINE configs[0] geometry = the per-env workload of this bench"}
    except Exception:
        return None


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40, help="timed rollout launches (bench steps)")
    ap.add_argument("--warmup", type=int, default=80, help="untimed rollout launches (clock ramp + pace controller)")
    ap.add_argument("--envs-per-gpu", type=int, default=0, help="0 = the workload's own size")
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c5_50", "c5_64"])
    ap.add_argument("--chunk", type=int, default=500, help="env-steps fused per rollout launch (= per bench step)")
    ap.add_argument("--lanes", type=int, default=0, help="lanes per wave carrying agents (0=auto)")
    ap.add_argument("--wpb", type=int, default=0, help="env tiles per workgroup (0=auto)")
    ap.add_argument("--writers", type=int, default=0, help="writer waves per env tile (0=auto)")
    ap.add_argument("--pool", type=int, default=4096, help="reset-pool entries (seeds 0..pool-1)")
    ap.add_argument("--throttle", type=int, default=0, help="stores a writer keeps in flight (0=auto, -1=off)")
    ap.add_argument("--pace", type=int, default=0, help="ns per env-step (0=adaptive, -1=off)")
    ap.add_argument("--policy", default="random", choices=["random", "greedy"],
                    help="random = actions from a device tensor (the bench line); greedy = the "
                         "reference's GreedyPolicy(epsilon=0) evaluated inside the rollout kernel (BASELINE configs[4])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-obs", action="store_true", help="diagnostic: skip the observation output")
    ap.add_argument("--only-obs", action="store_true", help="diagnostic: skip reward / flag outputs")
    args = ap.parse_args()

    if os.environ.get("CCX_DIAG_LIB"):   # diagnostics only: an experimental build of libccx
        import ctypes
        from collectivecrossing_amd import _abi, _lib
        _lib.LIB_PATH = Path(os.environ["CCX_DIAG_LIB"]).resolve()
        probe = ctypes.CDLL(str(_lib.LIB_PATH))      # older diagnostic builds lack newer symbols
        _abi.PROTOTYPES = {k: v for k, v in _abi.PROTOTYPES.items() if hasattr(probe, k)}
    from collectivecrossing_amd import sharding
    from collectivecrossing_amd.batched import BatchedCollectiveCrossing
    from collectivecrossing_amd.reset import build_reset_pool

    rank, world, local = sharding.init_from_env()
    if world != args.gpus:
        print(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}: start N>1 with "
              "`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`", file=sys.stderr)
        return 2
    local = local % max(1, torch.cuda.device_count())   # one rank per GPU on a full node
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    config, E_default = workload_config(args.workload)
    E = args.envs_per_gpu or E_default
    total = E * world
    env = BatchedCollectiveCrossing(config, E, device=dev, env_offset=rank * E, total_envs=total)
    N, L = env.num_agents, env.obs_len
    if args.lanes or args.wpb:
        env.set_launch_shape(args.lanes, args.wpb)
    if args.writers:
        env.set_writers(args.writers)
    if args.throttle:
        env.set_store_throttle(args.throttle)
    if args.pace:
        env.set_step_pace(args.pace)
    env.make_reset_pool(0, args.pool, on_device=not os.environ.get("CCX_DIAG_LIB"))  # seeds 0..pool-1
    env.reset_from_pool()

    chunk = max(1, args.chunk)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    # the action stream holds `n_buf` launches worth of steps and wraps around
    n_buf = max(1, min(max(args.steps, args.warmup, 1), 8))
    actions = torch.randint(0, 5, (n_buf * chunk, E, N), dtype=torch.uint8, device=dev, generator=gen)
    traj = env.alloc_rollout(chunk, want_obs=not args.no_obs)
    view = traj if not args.only_obs else type(traj)(traj.obs, None, None, None)
    launched = 0

    def run(nlaunches, events=None):
        nonlocal launched
        for _ in range(nlaunches):
            if events is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            if args.policy == "greedy":
                env.rollout_greedy(chunk, auto_reset=True, out=view, want_actions=False)
            else:
                a0 = (launched % n_buf) * chunk
                env.rollout(actions[a0:a0 + chunk], auto_reset=True, out=view)
            launched += 1
            if events is not None:
                e1.record()
                events.append((e0, e1))

    # The first ~30 launches of a process are slower (clock ramp; the pace controller starts from a
    # conservative value).  The default --warmup covers that; when the caller asks for
    # fewer warm-up steps the difference is run first as untimed set-up and reported as such.
    settle = max(0, 80 - args.warmup) if not os.environ.get("CCX_BENCH_NO_SETTLE") else 0
    run(settle)
    run(args.warmup)
    env.zero_counters()
    torch.cuda.synchronize(dev)
    sharding.barrier()
    torch.cuda.synchronize(dev)
    events: list = []
    t0 = time.perf_counter()
    run(args.steps, events)
    torch.cuda.synchronize(dev)
    sharding.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = sharding.allreduce_max(elapsed)
    counters = sharding.allreduce_counters(env.counters())   # the one RCCL reduction (48 B)

    # kernel time from HIP events recorded on the launch stream, per launch
    full = [a.elapsed_time(b) for a, b in events]
    kern_ms = float(np.mean(full)) if full else float("nan")
    bytes_unit = rollout_bytes_per_agent_step(N) - (4 * L if args.no_obs else 0)
    launch_bytes = bytes_unit * chunk * E * N
    achieved = launch_bytes / (kern_ms * 1e-3) / 1e9 if full else float("nan")
    survey_unit = 16 * N + 54 + 4   # SURVEY 8d contract figure, f64 rewards
    traffic = None
    tf = ROOT / "profiles" / "r01_traffic.json"
    if tf.exists():
        try:
            t = json.loads(tf.read_text())
            if t.get("envs") == E and t.get("chunk") == chunk and t.get("agents") == N:
                traffic = t.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    probe = write_bandwidth_probe(dev, traj.obs if traj.obs is not None else traj.reward) if rank == 0 else None
    if rank == 0:
        assert os.environ.get("CCX_DIAG_LIB") or counters["env_steps"] == args.steps * chunk * total, counters
        props = torch.cuda.get_device_properties(dev)
        env_sps = args.steps * chunk * total / elapsed
        line = {
            "metric": f"env-steps/sec, random-action rollout, {E} envs x {N} agents per GPU",
            "value": env_sps, "unit": "env-steps/s", "agent_steps_per_sec": env_sps * N,
            "live_agent_steps_per_sec": counters["live_agent_steps"] / elapsed,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "collective_backend": (torch.distributed.get_backend() if world > 1 else None),
            "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int32",
            "dtypes": "int32/u8 state and flags, f32 observations (exact integers), f64 rewards (one multiply)",
            "data": "synthetic",
            "config": {"workload": ("C2: 4096 envs x (5 boarding + 3 exiting) per GPU, 12x8 grid, "
                                    "DefaultReward + DefaultObservation, individual_at_destination, "
                                    "max_steps=100, uniform random actions, auto-reset from "
                                    f"{args.pool} reference-exact seeded placements") if args.workload == "c2"
                       else f"{args.workload} (diagnostic, not the bench line)",
                       "policy": args.policy,
                       "envs_per_gpu": E, "global_envs": total, "agents": N, "obs_len": L,
                       "step": "one fused rollout launch over an action batch [env_steps_per_step, envs, agents]",
                       "env_steps_per_step": chunk, "steps_per_launch": chunk,
                       "ms_per_env_step": elapsed * 1e3 / (args.steps * chunk),
                       "settle_launches": settle,
                       "launch_shape": env.launch_shape(), "step_pace_ns": env.step_pace_ns(),
                       "outputs": "full trajectory" + (" (no obs)" if args.no_obs else "")},
            "counters": counters,
            "device": {"name": props.name, "compute_units": props.multi_processor_count,
                       "hbm_GiB": round(props.total_memory / 2**30, 1)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "ccx::rollout_kernel", "kernel_ms_per_launch": kern_ms,
                         "bytes_per_agent_step": bytes_unit, "bytes_per_launch": launch_bytes,
                         "achievable_write_GBs_this_box": probe,
                         "frac_of_achievable": achieved / probe if probe else None,
                         "achieved_survey_8d_GBs": (survey_unit * chunk * E * N / (kern_ms * 1e-3) / 1e9
                                                    if full else None),
                         "survey_8d_bytes_per_agent_step": survey_unit},
        }
        if not args.no_cpu_baseline and world == 1 and args.workload == "c2" and args.policy == "random":
            line["cpu_baseline"] = cpu_baseline(config, N)
        elif world > 1:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    env.close()
    if world > 1:
        torch.distributed.destroy_process_group()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
