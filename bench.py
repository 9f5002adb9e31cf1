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
the metric stays env-steps/s = K * chunk * envs / elapsed.

    python bench.py --gpus N --steps K --warmup W

N > 1 works both ways: started by ``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N``
(WORLD_SIZE is set: this process is one rank) or as the bare command (WORLD_SIZE unset: this process
only LAUNCHES N rank processes -- before anything touches a GPU -- relays rank 0's JSON line and
returns the worst exit code).  One rank per GPU over RCCL; RCCL failing to come up is fatal.

Timing protocol.  Exactly W untimed launches, then K timed launches between barrier + synchronize
pairs, max over ranks: that is the `cold` block of the line.  The first ~30 launches of a process run
slower (clock ramp, pace controller start-up), so when W + K < 80 the difference is
run as further untimed set-up (`config.settle_launches`) and K launches are timed again the same way:
`value` is that steady-state figure (`value_protocol` says so in the line), `cold.value` the one a caller with
exactly W warm-ups sees; with W >= 80 they are the same measurement.  No number depends on state outside the
process: the pace controller's start value is measured in-process (`config.pace_start_source`: "calibration");
a pace cache is used only if the caller names one (CCX_PACE_CACHE -> "user_cache").

Prints ONE JSON line on rank 0.  `value` = env-steps/s over ALL ranks with inputs resident in HBM.
Weak scaling: every rank owns 4096 envs of a global batch of N*4096; envs are independent, the only
collective is the 48-byte all-reduce of the device-resident counters after the timed window (plus the
timing barrier and the max / gather of the per-rank elapsed times).
"""

from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
SETTLE_LAUNCHES = 80   # launches after which a process is in steady state


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
        # 64 agents exceed the reference's limit of 50 (configs.py:166): lift that one cap, keep every other rule
        return C.CollectiveCrossingConfig(**kw, strict_reference_limits=(nb != 32)), 1024
    raise SystemExit(f"unknown workload {name}")


def rollout_bytes_per_agent_step(n_agents: int) -> int:
    """Bytes one fused rollout launch MUST move per agent-step (state stays in registers):
    observation row 4*(6+4N) + action 1 + reward f64 8 + agent flag byte 1 (+ 1/N env flag byte,
    ignored).  SURVEY 8d's contract figure 16N+54 additionally counts a state read+write per step
    (22 B) that the fused kernel does not do; both are reported."""
    return 4 * (6 + 4 * n_agents) + 1 + 8 + 1


def write_bandwidth_probe(torch, dev, buf) -> float:
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


def cpu_baseline(config, n_agents: int, seconds_target: float = 7.0) -> dict:
    """The CPU oracle (a C port of the reference's sequential algorithm) on the host cores, same
    workload (full trajectory outputs), bounded sample: ~2.5 s on one thread, then ~3.5 s on one thread per
    schedulable core (os.cpu_count() workers; both the box's count and the threads used are stated)."""
    import numpy as np

    from collectivecrossing_amd.params import lower_config
    from collectivecrossing_amd.reset import build_reset_pool
    from oracle import oracle as ref  # cpu_baseline leg only

    params = lower_config(config)
    pool = build_reset_pool(config, 0, 128)
    box_cores = os.cpu_count() or 1
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = box_cores
    quota = None                                  # a container's CPU share (cgroup v2 cpu.max / v1 cfs quota), in cores
    try:
        q, per = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        quota = None if q == "max" else float(q) / float(per)
    except Exception:
        try:
            q = float(Path("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read_text())
            quota = q / float(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text()) if q > 0 else None
        except Exception:
            pass
    # one thread per core this process can really run on: os.cpu_count() unless an affinity mask or a cgroup quota says less
    cores = max(1, min(box_cores, affinity, int(quota + 0.999) if quota else box_cores, 128))
    E_t, K = 64, 250  # per thread: 64 envs x 250 steps, trajectory 20 MB

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
    while time.perf_counter() - t0 < seconds_target * 0.35:
        b.rollout(acts, auto_reset=True)
        reps1 += 1
    dt1 = time.perf_counter() - t0
    single = reps1 * E_t * K / dt1
    # all cores: one thread per core, each on its own envs (ctypes releases the GIL)
    workers = [(b, acts)] + [make(100 + i) for i in range(1, cores)]
    reps = max(1, int(reps1 * 1.3))

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
            "cores_of_the_box": box_cores, "cores_in_affinity_mask": affinity, "cgroup_cpu_quota": quota, "threads_used": cores,
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


def traffic_from_profiles(workload: str, E: int, N: int, chunk: int):
    """HBM bytes per launch from the PMC passes of an EARLIER rocprofv3 run of this same command
    (profiles/collect*.sh; counters cannot be collected inside a plain bench run).  Replayed from the
    tracked file, never measured here -- hence not `roofline.traffic`."""
    names = ([f"r04_{workload}_traffic.json", f"r03_{workload}_traffic.json", f"r02_{workload}_traffic.json"] +
             (["r01_traffic.json"] if workload == "c2" else []))
    for name in names:
        f = ROOT / "profiles" / name
        try:
            t = json.loads(f.read_text())
        except Exception:
            continue
        if t.get("envs") == E and t.get("chunk") == chunk and t.get("agents") == N:
            return {"hbm_bytes_per_launch": t.get("hbm_bytes_per_launch"), "file": f"profiles/{name}",
                    "ratio_to_algorithmic": (t["hbm_bytes_per_launch"] / t["algorithmic_bytes_per_launch"]
                                             if t.get("algorithmic_bytes_per_launch") else None)}
    return None


# --------------------------------------------------------------------------------------------------
# N > 1 from the bare command: this process becomes a launcher and never touches a GPU
# --------------------------------------------------------------------------------------------------
def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n: int, argv: list[str]) -> int:
    """Start n fresh rank processes of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their
    environment, like torch.distributed.run does), relay rank 0's stdout, return the worst exit code.
    The launcher has imported neither torch nor libccx: a process that has initialised the GPU must
    never be the one that forks / execs the ranks."""
    port = _free_port()
    base = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                MASTER_PORT=str(port), CCX_BENCH_LAUNCHER="bench.py")
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve()), *argv], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True))
    relay = threading.Thread(target=lambda: [print(ln, end="", flush=True) for ln in procs[0].stdout], daemon=True)
    relay.start()
    # below the harness limits (gpurun 1200 s, pytest 600 s): the launcher's own clean-up must be able to fire
    deadline = time.time() + float(os.environ.get("CCX_BENCH_LAUNCH_TIMEOUT", "1100"))
    worst = 0
    alive = set(range(n))

    def stop_ranks(why: str) -> None:
        """End exactly the processes started here: terminate, then kill what ignores it."""
        live = [r for r in sorted(alive) if procs[r].poll() is None]
        if not live:
            return
        print(f"[bench] {why}: stopping ranks {live}", file=sys.stderr)
        for r in live:
            procs[r].terminate()
        for r in live:
            try:
                procs[r].wait(timeout=20)
            except subprocess.TimeoutExpired:
                procs[r].kill()

    def on_signal(signum, _frame):            # SIGTERM / SIGINT to the launcher must not orphan ranks holding GPUs
        raise KeyboardInterrupt(f"signal {signum}")

    import signal
    old_handlers = {sig: signal.signal(sig, on_signal) for sig in (signal.SIGTERM, signal.SIGINT)}
    try:
        while alive:
            for r in sorted(alive):
                rc = procs[r].poll()
                if rc is not None:
                    alive.discard(r)
                    if rc != 0:
                        worst = worst or rc
            if (worst or time.time() > deadline) and alive:
                # one rank failed (or the job hangs at a rendezvous)
                stop_ranks(f"rank exit code {worst}" if worst else "launcher timeout")
                worst = worst or 124
                break
            time.sleep(0.05)
    except KeyboardInterrupt as stop:
        stop_ranks(f"launcher interrupted ({stop})")
        worst = worst or 130
    finally:
        stop_ranks("launcher exiting")        # (no-op unless something above raised)
        for sig, h in old_handlers.items():
            signal.signal(sig, h)
    relay.join(timeout=5)
    return worst


def rehearse(args) -> int:
    """`--rehearse`: the N>1 plumbing WITHOUT any env stepping (no GPU, no libccx needed): rendezvous,
    barriers, the three collectives of the bench with known per-rank values, one JSON line marked as a
    rehearsal.  This is what the CPU test-suite runs through the launcher with gloo."""
    from collectivecrossing_amd import sharding
    from collectivecrossing_amd._abi import COUNTER_FIELDS

    rank, world, _ = sharding.init_from_env()
    if world != args.gpus:
        print(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}", file=sys.stderr)
        return 2
    sharding.barrier()
    t0 = time.perf_counter()
    sharding.barrier()
    elapsed = sharding.allreduce_max(time.perf_counter() - t0 + rank)     # rank r reports r + epsilon
    counters = sharding.allreduce_counters({k: (rank + 1) * (q + 1) for q, k in enumerate(COUNTER_FIELDS)})
    per_rank = sharding.allgather_float(float(rank))
    if rank == 0:
        print(json.dumps({"metric": "rehearsal of the N>1 plumbing (no env was stepped)", "rehearsal": True,
                          "value": None, "n_gpus": world, "elapsed_max": elapsed, "counters": counters,
                          "per_rank": per_rank, "launcher": os.environ.get("CCX_BENCH_LAUNCHER", "torch.distributed.run"),
                          **sharding.group_info()}), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    return 0


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40, help="timed rollout launches (bench steps)")
    ap.add_argument("--warmup", type=int, default=80, help="untimed rollout launches before the timed ones")
    ap.add_argument("--envs-per-gpu", type=int, default=0, help="0 = the workload's own size")
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c5_50", "c5_64"])
    ap.add_argument("--chunk", type=int, default=500, help="env-steps fused per rollout launch (= per bench step)")
    ap.add_argument("--lanes", type=int, default=0, help="lanes per wave carrying agents (0=auto)")
    ap.add_argument("--wpb", type=int, default=0, help="env tiles per workgroup (0=auto)")
    ap.add_argument("--writers", type=int, default=0, help="writer waves per env tile (0=auto)")
    ap.add_argument("--pool", type=int, default=4096, help="reset-pool entries (seeds 0..pool-1)")
    ap.add_argument("--throttle", type=int, default=0, help="stores a writer keeps in flight (0=auto, -1=off)")
    ap.add_argument("--pace", type=int, default=0, help="ns per env-step (0=adaptive, -1=off)")
    ap.add_argument("--policy", default="random", choices=["random", "greedy", "device-random"],
                    help="random = actions from a device tensor (the bench line); greedy = the "
                         "reference's GreedyPolicy(epsilon=0) evaluated inside the rollout kernel (BASELINE configs[4]); "
                         "device-random = uniform actions drawn inside the kernel (CCX_POLICY_RANDOM), no action tensor")
    ap.add_argument("--epsilon", type=float, default=0.0,
                    help="with --policy greedy: the policy's randomness_factor, drawn on the device "
                         "(ccx_set_policy_epsilon; the reference's create_greedy_policy default is 0.1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the no-obs / K=1 secondary figures")
    ap.add_argument("--no-obs", action="store_true", help="diagnostic: skip the observation output")
    ap.add_argument("--compact-obs", action="store_true",
                    help="diagnostic: CCX_OBS_COMPACT [E][N][4] rows instead of the DefaultObservation rows "
                         "(never the bench line: 16 B instead of 16N+24 B of observation per agent-step)")
    ap.add_argument("--only-obs", action="store_true", help="diagnostic: skip reward / flag outputs")
    ap.add_argument("--direct-rccl", action="store_true",
                    help="reduce the counters with ccx_rccl_allreduce_counters (RCCL through the C-ABI) instead of "
                         "torch.distributed")
    ap.add_argument("--buffers", type=int, default=1,
                    help="diagnostic: cycle the rollouts through this many trajectory buffers (the bench line rewrites ONE; "
                         "DESIGN.md 3.6: launches that cycle through > 3 GB of output memory sustain less)")
    ap.add_argument("--tunable", action="append", default=[], metavar="NAME=VALUE",
                    help="diagnostic: ccx_set_tunable(NAME, VALUE) on the handle (repeatable)")
    ap.add_argument("--rehearse", action="store_true", help="N>1 plumbing only, no env stepping (CPU-runnable)")
    return ap.parse_args(argv)


def secondary_figures(torch, env, actions, chunk: int) -> dict:
    """Two figures next to the headline (rank 0, N=1): the rollout WITHOUT the observation output
    (the sim chain + small outputs: what bounds small batches) and the latency of a single-step
    ccx_step launch (K = 1, full outputs), both from HIP events on the launch stream."""
    dev = env.device
    E, N = env.num_envs, env.num_agents
    out = {}
    small = env.alloc_rollout(chunk, want_obs=False)
    for _ in range(3):
        env.rollout(actions[:chunk], auto_reset=True, out=small)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        env.rollout(actions[:chunk], auto_reset=True, out=small)
    e1.record()
    torch.cuda.synchronize(dev)
    ms = e0.elapsed_time(e1) / reps
    out["no_obs"] = {"env_steps_per_sec": chunk * E / (ms * 1e-3), "us_per_env_step": ms * 1e3 / chunk,
                     "what": f"ccx_rollout, {chunk} steps per launch, rewards + flag bytes only"}
    del small
    # the same rollout with the compact observation output ([E][N][4] rows, include/ccx.h: CCX_OBS_COMPACT) instead of the
    # DefaultObservation rows: what a consumer on the GPU would ask for; bound by the sim wave, not by memory
    comp = env.alloc_rollout(chunk, want_obs=False, want_compact=True)
    for _ in range(3):
        env.rollout(actions[:chunk], auto_reset=True, out=comp)
    e0.record()
    for _ in range(reps):
        env.rollout(actions[:chunk], auto_reset=True, out=comp)
    e1.record()
    torch.cuda.synchronize(dev)
    ms = e0.elapsed_time(e1) / reps
    out["compact_obs"] = {"env_steps_per_sec": chunk * E / (ms * 1e-3), "us_per_env_step": ms * 1e3 / chunk,
                          "what": f"ccx_rollout, {chunk} steps per launch, rewards + flag bytes + compact observation rows"}
    del comp
    for _ in range(20):
        env.step(actions[0])
    reps = 200
    per_step = [actions[k % actions.shape[0]] for k in range(reps)]     # (the views are made outside the timed loop)
    e0.record()
    for a in per_step:
        env.step(a)
    e1.record()
    torch.cuda.synchronize(dev)
    us = e0.elapsed_time(e1) * 1e3 / reps
    out["step_k1"] = {"us_per_step": us, "env_steps_per_sec": E / (us * 1e-6),
                      "what": f"{reps} back-to-back ccx_step launches (K = 1, full outputs), eager"}
    # The eager figure is what the HOST can issue (Python + hipLaunchKernel per call); captured into a HIP graph the
    # same launches run back to back on the device: the kernel's own single-step latency.
    try:
        side = torch.cuda.Stream(device=dev)
        env.use_stream(side)
        with torch.cuda.stream(side):
            env.step(actions[0])
            side.synchronize()
            graph = torch.cuda.CUDAGraph()
            n = min(100, int(actions.shape[0]))
            with torch.cuda.graph(graph, stream=side):
                for k in range(n):
                    env.step(actions[k])
            graph.replay()
            side.synchronize()
            g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            g0.record(side)
            for _ in range(10):
                graph.replay()
            g1.record(side)
            side.synchronize()
        us_g = g0.elapsed_time(g1) * 1e3 / (10 * n)
        out["step_k1_graph"] = {"us_per_step": us_g, "env_steps_per_sec": E / (us_g * 1e-6),
                                "what": f"{n} ccx_step launches (K = 1, full outputs) captured into one HIP graph, replayed"}
        del graph
    except Exception as exc:   # (a secondary figure must never cost the bench line)
        out["step_k1_graph"] = {"error": repr(exc)}
    finally:
        env.use_stream(torch.cuda.current_stream(dev))
    return out


def measure_workload(torch, dev, name: str, envs: int, chunk: int, policy: str, settle: int = 30, timed: int = 10) -> dict:
    """One BASELINE workload / batch size measured like the headline, in this process, on a handle of its own: `settle`
    untimed launches (the first adaptive one calibrates the pace controller in-process), then `timed` launches between
    synchronize pairs with a HIP-event pair around each: frac = algorithmic bytes / mean kernel time / 8 TB/s, frac_wall the
    same bytes over the wall clock of the window.  Full trajectory outputs, auto-reset from 1024 seeded placements."""
    import numpy as np

    from collectivecrossing_amd.batched import BatchedCollectiveCrossing
    config, e_default = workload_config(name)
    E = envs or e_default
    env = BatchedCollectiveCrossing(config, E, device=dev)
    try:
        N = env.num_agents
        env.make_reset_pool(0, 1024, on_device=True)
        env.reset_from_pool()
        actions = None
        if policy == "random":
            gen = torch.Generator(device=dev).manual_seed(4321)
            actions = torch.randint(0, 5, (chunk, E, N), dtype=torch.uint8, device=dev, generator=gen)
        traj = env.alloc_rollout(chunk)

        def launch():
            if policy == "greedy":
                env.rollout_greedy(chunk, auto_reset=True, out=traj, want_actions=False)
            else:
                env.rollout(actions, auto_reset=True, out=traj)

        for _ in range(settle):
            launch()
        torch.cuda.synchronize(dev)
        events = []
        t0 = time.perf_counter()
        for _ in range(timed):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            launch()
            e1.record()
            events.append((e0, e1))
        torch.cuda.synchronize(dev)
        wall = time.perf_counter() - t0
        ms = [a.elapsed_time(b) for a, b in events]
        launch_bytes = rollout_bytes_per_agent_step(N) * chunk * E * N
        kern = float(np.mean(ms))
        return {"workload": name, "envs": E, "agents": N, "policy": policy, "steps_per_launch": chunk,
                "settle_launches": settle, "timed_launches": timed,
                "env_steps_per_sec": timed * chunk * E / wall, "kernel_ms_per_launch": kern,
                "kernel_ms_min_median_max": [float(np.min(ms)), float(np.median(ms)), float(np.max(ms))],
                "bytes_per_launch": launch_bytes,
                "frac": launch_bytes / (kern * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "frac_wall": launch_bytes * timed / wall / 1e9 / HBM_PEAK_GBS,
                "launch_shape": env.launch_shape(), "step_pace_ns": env.step_pace_ns()}
    finally:
        env.close()
        torch.cuda.empty_cache()


# (workload, envs per GPU or 0 = the workload's own, env-steps per launch, policy): BASELINE configs[2] and [4] (both agent
# counts), and the C2 geometry from 1024 to 65 536 envs (small batches, several rounds of workgroups) -- VERDICT r3 item 2.
# The large batches take fewer steps per launch so that their rows stay ~2.7 GB, the headline launch's size: the same launch
# loses 5-8 % once the memory it streams through passes ~4 GB (profiles/r04_output_size.txt)
SECONDARY_WORKLOADS = [("c3", 0, 500, "random"), ("c5_50", 0, 500, "greedy"), ("c5_64", 0, 500, "greedy"),
                       ("c2", 1024, 500, "random"), ("c2", 2048, 500, "random"), ("c2", 16384, 64, "random"),
                       ("c2", 32768, 64, "random"), ("c2", 65536, 40, "random"),
                       # two batch sizes BETWEEN the powers of two every launch-shape rule was tuned on (round 4: 10 000 envs sat on a
                       # rule boundary at 0.49 of the peak, 20 000 envs' partial last round drove the pace controller to 0.67)
                       ("c2", 10000, 200, "random"), ("c2", 20000, 100, "random")]


def secondary_workloads(torch, dev) -> list:
    out = []
    for name, envs, chunk, policy in SECONDARY_WORKLOADS:
        try:
            out.append(measure_workload(torch, dev, name, envs, chunk, policy))
        except Exception as exc:      # (a secondary figure must never cost the bench line)
            out.append({"workload": name, "envs": envs, "error": repr(exc)})
    return out


def sustained_window(torch, env, actions, chunk: int, traj, seconds: float = 2.5, bucket_s: float = 0.25) -> dict:
    """The headline's launches issued continuously for `seconds` (nothing showed the pace controller / the HBM
    write stream holding over more than 0.1 s): HIP-event pair around every launch, the launches grouped into buckets of
    `bucket_s` by their position in the stream; the fraction of the peak by the WALL clock of the whole window."""
    import numpy as np
    dev = env.device
    E, N = env.num_envs, env.num_agents
    n_buf = max(1, actions.shape[0] // chunk)
    launch_bytes = rollout_bytes_per_agent_step(N) * chunk * E * N
    torch.cuda.synchronize(dev)
    events = []
    t0 = time.perf_counter()
    k = 0
    while True:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0 = (k % n_buf) * chunk
        e0.record()
        env.rollout(actions[a0:a0 + chunk], auto_reset=True, out=traj)
        e1.record()
        events.append((e0, e1))
        k += 1
        if k % 64 == 0:
            # (the host must not run unboundedly ahead of the device: wait for the launch issued 64 launches ago)
            events[k - 64][1].synchronize()
            if time.perf_counter() - t0 >= seconds:
                break
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    ms = np.array([a.elapsed_time(b) for a, b in events])
    starts = np.array([events[0][0].elapsed_time(a) for a, _ in events]) * 1e-3     # seconds since the first launch
    buckets = []
    nb = int(np.ceil((starts[-1] + 1e-9) / bucket_s))
    for b in range(nb):
        sel = ms[(starts >= b * bucket_s) & (starts < (b + 1) * bucket_s)]
        if sel.size:
            buckets.append({"t_s": round(b * bucket_s, 3), "launches": int(sel.size), "ms_min": float(sel.min()),
                            "ms_median": float(np.median(sel)), "ms_max": float(sel.max()),
                            "frac_median": launch_bytes / (float(np.median(sel)) * 1e-3) / 1e9 / HBM_PEAK_GBS})
    return {"what": f"{k} back-to-back {chunk}-step launches of the headline workload, issued continuously",
            "seconds": wall, "launches": k, "env_steps_per_sec": k * chunk * E / wall,
            "frac_wall": launch_bytes * k / wall / 1e9 / HBM_PEAK_GBS,
            "frac_kernel_mean": launch_bytes / (float(ms.mean()) * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "kernel_ms_min_median_max": [float(ms.min()), float(np.median(ms)), float(ms.max())],
            "launches_over_1p05_median": int((ms > 1.05 * np.median(ms)).sum()),
            "buckets": buckets}


def short_launches(torch, env, actions) -> dict:
    """Launches of K = 1 ... 64 env-steps (policy-in-the-loop callers: the reference's literal step() is K = 1), full outputs,
    each K captured into a HIP graph of `n` launches and replayed: us per launch on the device and the fraction of the HBM
    peak its algorithmic bytes reach."""
    dev = env.device
    E, N = env.num_envs, env.num_agents
    out = {}
    side = torch.cuda.Stream(device=dev)
    env.use_stream(side)
    try:
        with torch.cuda.stream(side):
            for K in (1, 2, 4, 8, 16, 32, 64):
                n = max(4, min(50, actions.shape[0] // K))
                traj = env.alloc_rollout(K)
                for _ in range(3):
                    env.rollout(actions[:K], auto_reset=True, out=traj)
                side.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side):
                    for k in range(n):
                        env.rollout(actions[k * K:(k + 1) * K], auto_reset=True, out=traj)
                graph.replay()
                side.synchronize()
                g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                g0.record(side)
                for _ in range(10):
                    graph.replay()
                g1.record(side)
                side.synchronize()
                us = g0.elapsed_time(g1) * 1e3 / (10 * n)
                nbytes = rollout_bytes_per_agent_step(N) * K * E * N
                out[f"k{K}"] = {"us_per_launch": us, "us_per_env_step": us / K, "env_steps_per_sec": K * E / (us * 1e-6),
                                "frac": nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS}
                del graph, traj
    except Exception as exc:
        out["error"] = repr(exc)
    finally:
        env.use_stream(torch.cuda.current_stream(dev))
    out["what"] = "ccx_rollout launches of K env-steps with full outputs, captured into HIP graphs and replayed"
    return out


def run_rank(args) -> int:
    import numpy as np
    import torch

    if os.environ.get("CCX_DIAG_LIB"):   # diagnostics only: an experimental build of libccx
        import ctypes
        from collectivecrossing_amd import _abi, _lib
        _lib.LIB_PATH = Path(os.environ["CCX_DIAG_LIB"]).resolve()
        probe = ctypes.CDLL(str(_lib.LIB_PATH))      # older diagnostic builds lack newer symbols
        _abi.PROTOTYPES = {k: v for k, v in _abi.PROTOTYPES.items() if hasattr(probe, k)}
    from collectivecrossing_amd import sharding
    from collectivecrossing_amd.batched import BatchedCollectiveCrossing

    rank, world, local = sharding.init_from_env()
    if world != args.gpus:
        print(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}", file=sys.stderr)
        return 2
    if not torch.cuda.is_available():
        print("[bench] no GPU visible: the hot path has no CPU implementation (use --rehearse for the "
              "N>1 plumbing on CPU)", file=sys.stderr)
        return 3
    local = local % max(1, torch.cuda.device_count())   # several ranks may share a GPU under gloo only
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
    for kv in args.tunable:
        env.set_tunable(kv.split("=")[0], int(kv.split("=")[1]))
    if args.epsilon:
        env.set_rng_seed(1234 + rank)
        env.set_policy_epsilon(args.epsilon)
    env.make_reset_pool(0, args.pool, on_device=not os.environ.get("CCX_DIAG_LIB"))  # seeds 0..pool-1
    env.reset_from_pool()

    chunk = max(1, args.chunk)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    # the action stream holds `n_buf` launches worth of steps and wraps around
    n_buf = max(1, min(max(args.steps, args.warmup, 1), 8))
    actions = torch.randint(0, 5, (n_buf * chunk, E, N), dtype=torch.uint8, device=dev, generator=gen)
    if args.compact_obs:
        args.no_obs = True
    trajs = [env.alloc_rollout(chunk, want_obs=not args.no_obs, want_compact=args.compact_obs) for _ in range(max(1, args.buffers))]
    traj = trajs[0]
    views = [t if not args.only_obs else type(t)(t.obs, None, None, None, t.obs_compact) for t in trajs]
    n_views = len(views)
    launched = 0

    def run(nlaunches, events=None):
        nonlocal launched
        for _ in range(nlaunches):
            if events is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            view = views[launched % len(views)]
            if args.policy == "greedy":
                env.rollout_greedy(chunk, auto_reset=True, out=view, want_actions=False)
            elif args.policy == "device-random":
                env.rollout_policy(chunk, "random", auto_reset=True, out=view, want_actions=False)
            else:
                a0 = (launched % n_buf) * chunk
                env.rollout(actions[a0:a0 + chunk], auto_reset=True, out=view)
            launched += 1
            if events is not None:
                e1.record()
                events.append((e0, e1))

    def timed_window():
        """K launches between barrier + synchronize pairs; (elapsed max over ranks, elapsed of this
        rank, per-launch kernel ms from HIP events on the launch stream, summed counters)."""
        env.zero_counters()
        torch.cuda.synchronize(dev)
        sharding.barrier()
        torch.cuda.synchronize(dev)
        events: list = []
        t0 = time.perf_counter()
        run(args.steps, events)
        torch.cuda.synchronize(dev)
        mine = time.perf_counter() - t0
        sharding.barrier()
        elapsed = sharding.allreduce_max(time.perf_counter() - t0)
        # the one data reduction of the job: 48 bytes, on the device under RCCL
        counters = direct.allreduce(env) if direct else sharding.allreduce_counters(env.counters_tensor())
        return elapsed, mine, [a.elapsed_time(b) for a, b in events], counters

    direct = sharding.RcclCounterReducer(env, rank, world) if args.direct_rccl else None
    bytes_unit = rollout_bytes_per_agent_step(N) - (4 * L if args.no_obs else 0) + (16 if args.compact_obs else 0)
    launch_bytes = bytes_unit * chunk * E * N

    def summarize(elapsed, launch_ms):
        kern_ms = float(np.mean(launch_ms)) if launch_ms else float("nan")
        achieved = launch_bytes / (kern_ms * 1e-3) / 1e9 if launch_ms else float("nan")
        wall = launch_bytes * args.steps / elapsed / 1e9          # the same bytes over the barrier-to-barrier wall clock
        return {"value": args.steps * chunk * total / elapsed, "ms_per_step": elapsed * 1e3 / args.steps,
                "kernel_ms_per_launch": kern_ms, "achieved": achieved, "frac": achieved / HBM_PEAK_GBS,
                "frac_wall": wall / HBM_PEAK_GBS,
                "kernel_ms_max_over_median": (float(np.max(launch_ms) / np.median(launch_ms)) if launch_ms else None)}

    # exactly --warmup untimed launches, then the timed window: what a caller with W warm-ups sees
    run(args.warmup)
    elapsed, mine, launch_ms, counters = timed_window()
    cold = summarize(elapsed, launch_ms)
    pace_start = env.pace_start()     # (after the first long launch: that is where a calibration happens)
    settle = 0
    retimed = None
    if args.warmup + args.steps < SETTLE_LAUNCHES and not os.environ.get("CCX_BENCH_NO_SETTLE"):
        # not in steady state yet: finish the start-up untimed and measure again
        settle = SETTLE_LAUNCHES - args.warmup - args.steps
        run(settle)
        elapsed, mine, launch_ms, counters = timed_window()
        # A host that loses the CPU inside the 8-ms window (shared boxes: seen once in ~20 runs, 0.54 instead of 0.38 ms per
        # launch by the wall clock) leaves the GPU idle between launches: the wall clock of the window then exceeds the
        # summed kernel times by far.  Such a window says nothing about the job: it is timed ONCE more and the line says so
        # (`retimed`).  Slow KERNELS never trigger this.
        busy_ms = float(np.sum(launch_ms)) if launch_ms else 0.0
        if world == 1 and busy_ms > 0 and elapsed * 1e3 > 1.15 * busy_ms:
            retimed = {"first_ms_per_step": elapsed * 1e3 / args.steps, "first_wall_over_kernel_time": elapsed * 1e3 / busy_ms,
                       "why": "wall clock of the window > 1.15 x its summed kernel times: the host stalled, not the GPU"}
            elapsed, mine, launch_ms, counters = timed_window()
    steady = summarize(elapsed, launch_ms)
    per_rank = [args.steps * chunk * E / t for t in sharding.allgather_float(mine)]

    probe = write_bandwidth_probe(torch, dev, traj.obs if traj.obs is not None else traj.reward) if rank == 0 else None
    secondary = None
    cpu_line = None
    if rank == 0 and world == 1 and not args.no_secondary and args.policy == "random" and not args.no_obs:
        secondary = secondary_figures(torch, env, actions, chunk)
        if args.workload == "c2" and not args.envs_per_gpu and not os.environ.get("CCX_DIAG_LIB"):
            secondary["short_launches"] = short_launches(torch, env, actions)
            secondary["sustained"] = sustained_window(torch, env, actions, chunk, traj)
    if rank == 0 and not args.no_cpu_baseline and world == 1 and args.workload == "c2" and args.policy == "random":
        # (between the GPU legs: the driver's activity sampler sees a busy GPU before AND after these ~7 s)
        cpu_line = cpu_baseline(config, N)
    if secondary is not None and args.workload == "c2" and not args.envs_per_gpu and not os.environ.get("CCX_DIAG_LIB"):
        trajs.clear()
        views.clear()
        traj = None
        torch.cuda.empty_cache()
        secondary["workloads"] = secondary_workloads(torch, dev)
    if rank == 0:
        assert os.environ.get("CCX_DIAG_LIB") or counters["env_steps"] == args.steps * chunk * total, counters
        props = torch.cuda.get_device_properties(dev)
        env_sps = steady["value"]
        kern_ms = steady["kernel_ms_per_launch"]
        survey_unit = 16 * N + 54 + 4   # SURVEY 8d contract figure, f64 rewards
        line = {
            "metric": f"env-steps/sec, random-action rollout, {E} envs x {N} agents per GPU",
            "value": env_sps, "unit": "env-steps/s", "agent_steps_per_sec": env_sps * N,
            "live_agent_steps_per_sec": counters["live_agent_steps"] / elapsed,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            **sharding.group_info(),
            "counters_allreduce": (f"ccx_rccl_allreduce_counters ({direct.num_ranks} RCCL rank(s))" if direct
                                   else "torch.distributed" if world > 1 else None),
            "launcher": os.environ.get("CCX_BENCH_LAUNCHER", "torch.distributed.run" if world > 1 else None),
            "per_rank_env_steps_per_sec": per_rank,
            "value_protocol": ("steady state: the requested --steps launches timed again after config.settle_launches "
                               "further untimed launches; the figure for EXACTLY --warmup + --steps is `cold`"
                               if settle else "exactly --warmup untimed launches, then --steps timed ones (= `cold`)"),
            "retimed": retimed,
            "ms_per_step": steady["ms_per_step"], "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int32",
            "dtypes": "int32/u8 state and flags, f32 observations (exact integers), f64 rewards (one multiply)",
            "data": "synthetic",
            "cold": {"what": f"the {args.steps} launches right after exactly {args.warmup} warm-up launches "
                             "(same barrier + synchronize protocol); `value` is the steady state after "
                             "config.settle_launches further untimed launches",
                     "value": cold["value"], "ms_per_step": cold["ms_per_step"],
                     "kernel_ms_per_launch": cold["kernel_ms_per_launch"],
                     "frac": cold["frac"], "frac_wall": cold["frac_wall"],
                     "pace_start_source": pace_start["source"],
                     "ratio_to_value": cold["value"] / env_sps},
            # the same window when nothing outside this process seeded the controller (no pace cache: the default)
            "cold_unseeded": ({"value": cold["value"], "ratio_to_value": cold["value"] / env_sps,
                               "pace_start_source": pace_start["source"]}
                              if pace_start["source"] != "user_cache" else None),
            "config": {"workload": ("C2: 4096 envs x (5 boarding + 3 exiting) per GPU, 12x8 grid, "
                                    "DefaultReward + DefaultObservation, individual_at_destination, "
                                    "max_steps=100, uniform random actions, auto-reset from "
                                    f"{args.pool} reference-exact seeded placements") if args.workload == "c2"
                       else f"{args.workload} (diagnostic, not the bench line)",
                       "policy": args.policy if not args.epsilon else f"{args.policy}, epsilon {args.epsilon} (device draws)",
                       "envs_per_gpu": E, "global_envs": total, "agents": N, "obs_len": L,
                       "step": "one fused rollout launch over an action batch [env_steps_per_step, envs, agents]",
                       "trajectory_buffer": ("every launch rewrites ONE set of [steps, envs, agents] output buffers (a fixed RL rollout "
                                             "buffer, %.2f GB here); launches that cycle through > 3 GB of output memory sustain "
                                             "0.86-0.88 of the peak instead of 0.90 (DESIGN.md 3.6)") % (chunk * E * N * rollout_bytes_per_agent_step(N) / 1e9),
                       "trajectory_buffers": n_views,
                       "env_steps_per_step": chunk, "steps_per_launch": chunk,
                       "ms_per_env_step": elapsed * 1e3 / (args.steps * chunk),
                       "settle_launches": settle,
                       "launch_shape": env.launch_shape(), "step_pace_ns": env.step_pace_ns(),
                       "pace_start_source": pace_start["source"], "pace_start_ns": pace_start["ns"],
                       "pace_probe_GBs": pace_start["probe_GBs"],
                       "outputs": "full trajectory" + (" (compact obs [E][N][4] instead of the rows)" if args.compact_obs
                                                       else " (no obs)" if args.no_obs else "")},
            "counters": counters,
            "device": {"name": props.name, "compute_units": props.multi_processor_count,
                       "hbm_GiB": round(props.total_memory / 2**30, 1)},
            "roofline": {"bound": "hbm", "achieved": steady["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": steady["frac"], "frac_wall": steady["frac_wall"], "frac_cold": cold["frac"],
                         "traffic": None,
                         "traffic_from_profiles": traffic_from_profiles(args.workload, E, N, chunk),
                         "kernel": "ccx::rollout_kernel", "kernel_ms_per_launch": kern_ms,
                         "kernel_ms_max_over_median": steady["kernel_ms_max_over_median"],
                         "bytes_per_agent_step": bytes_unit, "bytes_per_launch": launch_bytes,
                         "achievable_write_GBs_this_box": probe,
                         "frac_of_achievable": steady["achieved"] / probe if probe else None,
                         "achieved_survey_8d_GBs": survey_unit * chunk * E * N / (kern_ms * 1e-3) / 1e9,
                         "survey_8d_bytes_per_agent_step": survey_unit},
        }
        if secondary:
            line["secondary"] = secondary
        if cpu_line is not None:
            line["cpu_baseline"] = cpu_line
        elif world > 1:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if direct:
        direct.close()
    env.close()
    if world > 1:
        torch.distributed.destroy_process_group()
    return 0


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus, argv)     # launcher only: no torch, no GPU in this process
    if args.rehearse:
        return rehearse(args)
    return run_rank(args)


if __name__ == "__main__":
    raise SystemExit(main())
