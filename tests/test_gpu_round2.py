"""GPU tests of the round-2 C-ABI additions and of the code paths the bench numbers come from.

  * full-size C3 (4096 envs x 32 agents: 2048 workgroups in more than one round, the scaled pace of the
    partial last round) against the oracle -- collectivecrossing.py:161-261, observations.py:43-94;
  * tunables never change results; the pace start value; the reset-pool cursor when P divides total_envs;
  * CCX_CHECK_INPUTS = the reference's _check_action_and_agent_validity (collectivecrossing.py:685-711)
    for the array API;
  * ccx_rccl_allreduce_counters in a one-rank RCCL world;
  * inputs produced on another torch stream than the launch stream.
"""

import numpy as np
import pytest
from _fixtures import Golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ccx():
    import torch

    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    from collectivecrossing_amd.batched import BatchedCollectiveCrossing

    return BatchedCollectiveCrossing


def _np(t):
    return None if t is None else t.cpu().numpy()


def _against_oracle(oracle, ccx, g, E, K, seed, *, pool_size=257, setup=None, order=False, total=None, offset=0):
    from collectivecrossing_amd.reset import build_reset_pool

    rng = np.random.default_rng(seed)
    actions = rng.integers(0, 5, size=(K, E, g.N), dtype=np.uint8)
    orders = np.argsort(rng.random((K, E, g.N)), axis=-1).astype(np.uint8) if order else None
    pool = build_reset_pool(g.config, 7000 + seed, pool_size)
    ob = oracle.OracleBatch(g.params, E, env_offset=offset, total_envs=total)
    env = ccx(g.config, E, env_offset=offset, total_envs=total)
    if setup:
        setup(env)
    for b in (ob, env):
        b.set_reset_pool(pool)
        b.reset_from_pool()
    res = env.rollout(actions, orders, auto_reset=True)
    o_obs, o_rew, o_af, o_ef = ob.rollout(actions, orders, auto_reset=True)
    np.testing.assert_array_equal(_np(res.agent_flags), o_af)
    np.testing.assert_array_equal(_np(res.env_flags), o_ef)
    np.testing.assert_array_equal(_np(res.reward).view(np.uint64), o_rew.view(np.uint64))
    got = _np(res.obs).view(np.uint32)
    del res
    np.testing.assert_array_equal(got, o_obs.view(np.uint32))
    del got, o_obs
    st = env.get_state()
    for k in ("x", "y", "active", "terminated", "truncated", "step_count", "episode"):
        np.testing.assert_array_equal(st[k], getattr(ob, k), err_msg=k)
    c = env.counters()
    assert c == ob.counters.as_dict()
    shape = env.launch_shape()
    env.close()
    return c, shape, pool, st


def test_full_size_c3_equals_the_oracle(oracle, ccx):
    """BASELINE configs[2] at its full size: 4096 envs x (16 + 16) agents on the 20x12 grid, SimpleDistance,
    dense collisions -- the exact launch the C3 bench figure comes from (one writer wave per tile and four
    tiles per workgroup so that all 2048 tiles are resident in one round; tiles phased over the step period)."""
    g = Golden("g3_c3_dense_simple_distance")
    assert (g.N, g.config.width, g.config.height) == (32, 20, 12)
    c, shape, _, _ = _against_oracle(oracle, ccx, g, E=4096, K=18, seed=31, order=False)
    assert shape["writers_per_tile"] == 1 and shape["waves_per_block"] == 4
    assert shape["num_blocks"] == shape["resident_blocks"] == 512       # one round
    assert c["env_steps"] == 4096 * 18 and c["moves"] > 0


def test_c3_geometry_in_several_rounds_equals_the_oracle(oracle, ccx):
    """More workgroups than the device holds at once (6001 envs: 3001 tiles x 4 waves, a partial last tile and
    a partial last round whose schedule is scaled) -- the multi-round path of the pace logic."""
    g = Golden("g3_c3_dense_simple_distance")
    c, shape, _, _ = _against_oracle(oracle, ccx, g, E=6001, K=17, seed=33, order=False)
    assert shape["num_blocks"] > shape["resident_blocks"] > 0 and shape["writers_per_tile"] == 2   # (round 4: several rounds anyway -> two writers, one tile per workgroup)
    assert c["env_steps"] == 6001 * 17


def test_full_size_c3_shuffled_orders_equal_the_oracle(oracle, ccx):
    g = Golden("g3_c3_dense_simple_distance")
    _against_oracle(oracle, ccx, g, E=4096, K=6, seed=32, order=True)


@pytest.mark.parametrize("knobs", [{"pace_phase": 0, "tile_map": 0}, {"pace_phase": 1, "tile_map": 1}, {"pace_phase": 2, "tile_map": 4},
                                   {"pace_phase": 3, "tile_map": 0}, {"pace_phase": 1, "tile_map": 6}, {"hand2": 0}, {"hand2": 2}, {"max_launch_steps": 5},
                                   {"pair_rows": 0}, {"pair_rows": 1, "hand2": 2}])
@pytest.mark.parametrize("cfg_name,E,K", [("g1_c1_random", 4096, 70), ("g3_c3_dense_simple_distance", 1500, 20),
                                          ("g1_c1_random", 700, 33)])
def test_tunables_never_change_results(oracle, ccx, cfg_name, E, K, knobs):
    def setup(env):
        for k, v in knobs.items():
            env.set_tunable(k, v)
    _against_oracle(oracle, ccx, Golden(cfg_name), E, K, seed=3, setup=setup)


def test_unknown_tunable_is_rejected(ccx):
    from collectivecrossing_amd._lib import CcxError
    env = ccx(Golden("g7_n3_small").config, 4)
    with pytest.raises(CcxError, match="unknown tunable"):
        env.set_tunable("warp_speed", 1)
    with pytest.raises(CcxError, match="must be -1..3"):
        env.set_tunable("pace_phase", 9)
    env.close()


def test_pace_start_value_is_honoured_and_results_do_not_depend_on_it(oracle, ccx):
    g = Golden("g1_c1_random")
    seen = {}

    def setup(env):
        env.set_step_pace_start(912.0)
        seen["ns"] = env.step_pace_ns()
    _against_oracle(oracle, ccx, g, E=4096, K=70, seed=9, setup=setup)
    assert seen["ns"] == pytest.approx(912.0, rel=0.01)


def test_pool_cursor_walks_the_pool_when_the_pool_size_divides_the_batch(oracle, ccx):
    """total_envs % P == 0 used to pin every env to ONE placement for ever (stride 0); the stride is 1 then."""
    g = Golden("g8_rollout_c1")                      # max_steps = 25: several episodes in 90 steps
    c, _, pool, st = _against_oracle(oracle, ccx, g, E=64, K=90, seed=2, pool_size=32)
    assert c["episodes"] >= 64 * 3 and int(st["episode"].min()) >= 3
    # ... and the same through a 2-way shard view (offset 32 of 64): still the oracle's trajectory
    _against_oracle(oracle, ccx, g, E=32, K=90, seed=2, pool_size=32, total=64, offset=32)


def test_check_inputs_reports_what_the_reference_would_raise_on(ccx):
    from collectivecrossing_amd._lib import CcxInputError
    g = Golden("g1_c1_random")
    E, N, K = 50, g.N, 7
    env = ccx(g.config, E, check_inputs=True)
    env.set_state(**{k: np.repeat(v[:1], E, axis=0) for k, v in g.init_state().items()})
    rng = np.random.default_rng(0)
    good = rng.integers(0, 5, size=(K, E, N), dtype=np.uint8)
    good[rng.random((K, E, N)) < 0.1] = 255                         # absent agents are fine
    orders = np.argsort(rng.random((K, E, N)), axis=-1).astype(np.uint8)
    env.rollout(good, orders)
    env.synchronize()                                               # nothing to report
    env.step(good[0], orders[0])
    env.check_inputs()
    bad = good.copy()
    bad[2, 7, 3], bad[5, 11, 0], bad[5, 11, 1] = 5, 17, 254         # actions.py:8-24 knows 0..4 only
    env.rollout(bad, orders)
    with pytest.raises(CcxInputError, match=r"Invalid action: 3 action byte") as ei:
        env.synchronize()
    assert isinstance(ei.value, ValueError)                          # the reference raises ValueError
    env.synchronize()                                               # reported once
    bad_order = orders.copy()
    bad_order[1, 4] = bad_order[1, 4][0]                            # one slot named N times
    bad_order[3, 9, 2] = N                                          # a slot the env does not have
    env.rollout(good, bad_order)
    with pytest.raises(CcxInputError, match=r"2 move-order row"):
        env.counters()
    env.step(bad[2], None)
    with pytest.raises(CcxInputError, match=r"Invalid action: 1 action byte"):
        env.check_inputs()
    env.close()
    # off by default: the same inputs step silently (documented behaviour of the array API)
    env = ccx(g.config, E)
    env.rollout(bad, orders)
    env.synchronize()
    env.close()


def test_rccl_counter_allreduce_in_a_one_rank_world(ccx):
    """ccx_rccl_allreduce_counters: a real ncclCommInitRank + ncclAllReduce through the C-ABI (one rank is
    all a one-GPU box can host; N ranks differ only in the communicator size)."""
    import torch

    from collectivecrossing_amd import sharding
    g = Golden("g8_rollout_c1")
    env = ccx(g.config, 300)
    env.make_reset_pool(0, 64)
    env.reset_from_pool()
    acts = torch.randint(0, 5, (40, 300, g.N), dtype=torch.uint8, device=env.device)
    env.rollout(acts, auto_reset=True, want_obs=False)
    red = sharding.RcclCounterReducer(env, rank=0, world=1)
    total = red.allreduce(env)
    assert red.num_ranks == 1
    assert total == env.counters() and total["env_steps"] == 300 * 40
    env.rollout(acts, auto_reset=True, want_obs=False)
    assert red.allreduce(env)["env_steps"] == 2 * 300 * 40
    red.close()
    env.close()


def test_inputs_made_on_another_stream_are_ordered_before_the_launch(ccx):
    """The handle launches on the stream captured at construction; a caller inside
    `torch.cuda.stream(other)` hands over tensors whose H2D copies run on `other`."""
    import torch

    g = Golden("g1_c1_random")
    env = ccx(g.config, g.E)
    env.set_state(**g.init_state())
    want = ccx(g.config, g.E)
    want.set_state(**g.init_state())
    ref = want.rollout(g["actions"], g["order"])
    side = torch.cuda.Stream(device=env.device)
    big = torch.empty(64 << 20, dtype=torch.uint8, device=env.device)
    with torch.cuda.stream(side):
        big.fill_(1)                                                  # keeps `side` busy for a while
        res = env.rollout(g["actions"], g["order"])                  # numpy -> device copies on `side`
    env.synchronize()
    side.synchronize()
    np.testing.assert_array_equal(_np(res.obs).view(np.uint32), _np(ref.obs).view(np.uint32))
    np.testing.assert_array_equal(_np(res.agent_flags), _np(ref.agent_flags))
    env.close()
    want.close()


def test_bench_direct_rccl_flag():
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    p = subprocess.run([sys.executable, str(root / "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--no-secondary", "--direct-rccl", "--envs-per-gpu", "1024"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert d["counters_allreduce"].startswith("ccx_rccl_allreduce_counters (1 RCCL rank")
    assert d["counters"]["env_steps"] == 2 * 500 * 1024


# ---- CCX_OBS_COMPACT ------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["g1_c1_random", "g2_c1_shuffled_absent", "g7_n5_odd", "g3_c3_dense_shuffled",
                                  "g4_c5_all_at_dest_greedy_25_25", "g7_n1_boarding_only"])
def test_compact_observations_expand_to_the_reference_rows(ccx, name):
    """obs_compact holds (x, y, type, active) once per agent; ccx_expand_observations must rebuild the
    DefaultObservation rows the reference recorded (observations.py:43-94) bit for bit -- from a rollout
    that wrote ONLY the compact output, and from a step that wrote both."""
    g = Golden(name)
    env = ccx(g.config, g.E)
    env.set_state(**g.init_state())
    res = env.rollout(g["actions"], g["order"], want_obs=False, want_compact=True)
    assert res.obs is None and tuple(res.obs_compact.shape) == (g.K, g.E, g.N, 4)
    comp = _np(res.obs_compact)
    nb = g.config.num_boarding_agents
    np.testing.assert_array_equal(comp[..., 0], g["x"].astype(np.float32))
    np.testing.assert_array_equal(comp[..., 1], g["y"].astype(np.float32))
    np.testing.assert_array_equal(comp[..., 2], np.broadcast_to((np.arange(g.N) >= nb).astype(np.float32), comp[..., 2].shape))
    np.testing.assert_array_equal(comp[..., 3], g["active"].astype(np.float32))
    full = env.expand_observations(res.obs_compact)
    np.testing.assert_array_equal(_np(full).view(np.uint32), g["obs"].view(np.uint32))
    np.testing.assert_array_equal(_np(res.agent_flags), g["agent_flags"])
    # one step with both outputs; expanding a single env's rows works too
    env.set_state(**g.init_state())
    r = env.step(g["actions"][0], g["order"][0], want_obs=True, want_compact=True)
    np.testing.assert_array_equal(_np(env.expand_observations(r.obs_compact)).view(np.uint32), _np(r.obs).view(np.uint32))
    one = env.expand_observations(r.obs_compact[g.E - 1])
    np.testing.assert_array_equal(_np(one).view(np.uint32), g["obs"][0, g.E - 1].view(np.uint32))
    env.close()


def test_compact_rollout_with_autoreset_and_policy_equals_the_full_rollout(ccx):
    g = Golden("g4_small_all_at_dest_greedy")
    a, b = ccx(g.config, 700), ccx(g.config, 700)
    for env in (a, b):
        env.make_reset_pool(5, 97)
        env.reset_from_pool()
    full, acts_a = a.rollout_greedy(90, auto_reset=True)
    out = b.alloc_rollout(90, want_obs=False, want_compact=True)
    comp, acts_b = b.rollout_greedy(90, auto_reset=True, out=out)
    np.testing.assert_array_equal(_np(acts_a), _np(acts_b))
    np.testing.assert_array_equal(_np(b.expand_observations(comp.obs_compact)).view(np.uint32), _np(full.obs).view(np.uint32))
    np.testing.assert_array_equal(_np(comp.reward).view(np.uint64), _np(full.reward).view(np.uint64))
    assert a.counters() == b.counters()
    a.close()
    b.close()


def test_steady_state_launches_have_no_outliers(ccx):
    """The adaptive pace controller in steady state: after the start-up phase, 60 consecutive
    C2 launches (4096 envs x 8 agents, 500 env-steps each, full outputs) stay within a few per cent of their
    median -- a collapse of the drain rate costs 8-13 % of a launch (the chip's write limiter cuts in for ~0.5 ms,
    profiles/r02_lag_trace_*.txt) and must be a rare event, not a rhythm (round 1: one every 15-20 launches).
    Bounds with room for the lease-to-lease spread: at most three launches above 1.05 x the median (the controller's own
    spacing is one per 60-130 launches), none above 1.25 x."""
    import torch

    from bench import c2_config
    E, K = 4096, 500
    env = ccx(c2_config(), E)
    env.set_timing(True)
    env.make_reset_pool(0, 4096)
    env.reset_from_pool()
    acts = torch.randint(0, 5, (K, E, env.num_agents), dtype=torch.uint8, device=env.device)
    traj = env.alloc_rollout(K)
    for _ in range(100):
        env.rollout(acts, auto_reset=True, out=traj)
    ms = []
    for _ in range(60):
        env.rollout(acts, auto_reset=True, out=traj)
        ms.append(env.last_launch_ms())
    ms = np.array(ms)
    med = float(np.median(ms))
    over = int((ms > 1.05 * med).sum())
    st = env.pace_state()
    env.close()
    assert st["paced"] == 1.0 and st["next_pace_ns"] > 0
    assert over <= 3 and ms.max() < 1.25 * med, (over, ms.max() / med, sorted(ms)[-3:], st)


@pytest.mark.parametrize("E", [512, 1024, 2048, 3072])
def test_small_batches_use_full_tiles_with_two_writers_and_equal_the_oracle(oracle, ccx, E):
    """Below the memory-bound regime the default shape is full 64-lane tiles with two to three writer waves each, split
    by role -- half tiles with four writers while full ones would leave half the CUs idle: same results, of
    course."""
    g = Golden("g8_rollout_c1")
    c, shape, _, _ = _against_oracle(oracle, ccx, g, E=E, K=70, seed=17 + E)
    assert (shape["lanes_per_wave"], shape["writers_per_tile"], shape["waves_per_block"]) == (
        (32, 4, 1) if E <= 1024 else (64, 3, 1))          # (round 4: three writers up to one round of full tiles, profiles/r04_shape_sweep.json)
    assert c["episodes"] > 0


def test_a_paced_rollout_captured_into_a_graph_replays_without_touching_the_controller(ccx):
    import torch

    from bench import c2_config
    E, K = 4096, 64
    env = ccx(c2_config(), E)
    env.make_reset_pool(0, 512)
    env.reset_from_pool()
    acts = torch.randint(0, 5, (K, E, env.num_agents), dtype=torch.uint8, device=env.device)
    traj = env.alloc_rollout(K)
    side = torch.cuda.Stream(device=env.device)
    env.use_stream(side)
    with torch.cuda.stream(side):
        for _ in range(3):
            env.rollout(acts, auto_reset=True, out=traj)        # adaptive launches outside the capture
        side.synchronize()
        before = env.pace_state()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            env.rollout(acts, auto_reset=True, out=traj)
        for _ in range(5):
            graph.replay()
        side.synchronize()
    after = env.pace_state()
    assert after["next_pace_ns"] == before["next_pace_ns"] and after["calm_launches"] == before["calm_launches"]
    assert env.counters()["env_steps"] == (3 + 5) * K * E
    env.close()


# ---- CCX_POLICY_RANDOM: uniform actions drawn on the device ------------------------------------------
@pytest.mark.parametrize("cfg_name,E,K", [("g8_rollout_c1", 777, 130), ("g3_c3_dense_simple_distance", 150, 60),
                                          ("g7_n5_odd", 200, 90), ("g4_c5_all_at_dest_greedy_32_32", 40, 70)])
def test_device_random_rollout_equals_the_oracle(oracle, ccx, cfg_name, E, K):
    from collectivecrossing_amd.reset import build_reset_pool
    g = Golden(cfg_name)
    pool = build_reset_pool(g.config, 77, 129)
    seed = 0xDEADBEEF12345678 ^ E
    oracle.OracleBatch.set_rng_seed(seed)
    ob = oracle.OracleBatch(g.params, E, env_offset=5, total_envs=E + 9)
    env = ccx(g.config, E, env_offset=5, total_envs=E + 9)
    env.set_rng_seed(seed)
    for b in (ob, env):
        b.set_reset_pool(pool)
        b.reset_from_pool()
    o_act, o_obs, o_rew, o_af, o_ef = ob.rollout_greedy(K, auto_reset=True, policy="random")
    res, acts = env.rollout_policy(K // 2, "random", auto_reset=True)          # two launches: same stream
    res2, acts2 = env.rollout_policy(K - K // 2, "random", auto_reset=True)
    np.testing.assert_array_equal(np.concatenate([_np(acts), _np(acts2)]), o_act)
    np.testing.assert_array_equal(np.concatenate([_np(res.obs), _np(res2.obs)]).view(np.uint32), o_obs.view(np.uint32))
    np.testing.assert_array_equal(np.concatenate([_np(res.reward), _np(res2.reward)]).view(np.uint64), o_rew.view(np.uint64))
    np.testing.assert_array_equal(np.concatenate([_np(res.agent_flags), _np(res2.agent_flags)]), o_af)
    np.testing.assert_array_equal(np.concatenate([_np(res.env_flags), _np(res2.env_flags)]), o_ef)
    assert env.counters() == ob.counters.as_dict() and ob.counters.moves > 0
    env.close()


@pytest.mark.parametrize("name,eps", [("g4_c5_all_at_dest_greedy_25_25", 0.0), ("g4_c5_all_at_dest_greedy_32_32", 0.0),
                                      ("g4_c5_all_at_dest_greedy_25_25", 0.1)])
def test_full_size_c5_greedy_rollout_equals_the_oracle(oracle, ccx, name, eps):
    """BASELINE configs[4] at its full size -- 1024 envs x 64 (and the reference-legal 50) agents on the 32x16 grid,
    the greedy policy evaluated inside the kernel, AllAtDestination, auto-reset: the exact launch the C5 bench
    figures come from (one env per wave, three writer waves, 1024 workgroups in one round, tiles phased, grouped tile
    map; N = 50: tile regions that begin and end mid-line, i.e. the edge-iteration instantiation), 20 paced steps,
    every output bit-equal to the oracle.  Also with the exploration draws of ccx_set_policy_epsilon."""
    from collectivecrossing_amd.reset import build_reset_pool

    g = Golden(name)
    E, K = 1024, 20
    pool = build_reset_pool(g.config, 500, 300)
    ob, env = oracle.OracleBatch(g.params, E), ccx(g.config, E)
    try:
        for b in (ob, env):
            b.set_reset_pool(pool)
            b.reset_from_pool()
        oracle.OracleBatch.set_rng_seed(9)
        oracle.OracleBatch.set_policy_epsilon(eps)
        env.set_rng_seed(9)
        env.set_policy_epsilon(eps)
        for launch in range(2):           # the second launch starts mid-episode, from the state the first one left
            o_act, o_obs, o_rew, o_af, o_ef = ob.rollout_greedy(K, auto_reset=True)
            res, acts = env.rollout_greedy(K, auto_reset=True)
            np.testing.assert_array_equal(_np(acts), o_act)
            np.testing.assert_array_equal(_np(res.agent_flags), o_af)
            np.testing.assert_array_equal(_np(res.env_flags), o_ef)
            np.testing.assert_array_equal(_np(res.reward).view(np.uint64), o_rew.view(np.uint64))
            got = _np(res.obs).view(np.uint32)
            del res
            np.testing.assert_array_equal(got, o_obs.view(np.uint32))
            del got, o_obs
        shape = env.launch_shape()
        assert (shape["writers_per_tile"], shape["waves_per_block"], shape["num_blocks"]) == (3, 1, 1024)
        assert shape["num_blocks"] == shape["resident_blocks"]
        assert env.step_pace_ns() > 0                           # a paced launch
        assert env.counters() == ob.counters.as_dict()
    finally:
        oracle.OracleBatch.set_policy_epsilon(0.0)
        env.close()


def test_long_rollouts_are_cut_into_launches_without_changing_a_bit(ccx, oracle):
    """The writer waves address rewards / flag bytes / compact rows with 32-bit offsets from the stream's base, so
    `ccx_rollout` cuts a rollout whose small streams would exceed 4 GiB into several launches on the stream
    (ccx_api.hip: run_rollout).  Forced here with the `max_launch_steps` tunable: 45 steps in launches of at most 7,
    tensor actions with a shuffled move order and the in-kernel greedy policy, auto-reset -- every output equals the
    oracle's single 45-step rollout."""
    from collectivecrossing_amd.reset import build_reset_pool
    g = Golden("g1_c1_random")
    E, K = 37, 45
    pool = build_reset_pool(g.config, 7, 64)
    rng = np.random.default_rng(5)
    acts = rng.integers(0, 5, size=(K, E, g.N), dtype=np.uint8)
    order = np.stack([np.stack([rng.permutation(g.N) for _ in range(E)]) for _ in range(K)]).astype(np.uint8)
    ob, env = oracle.OracleBatch(g.params, E), ccx(g.config, E)
    try:
        env.set_tunable("max_launch_steps", 7)
        for b in (ob, env):
            b.set_reset_pool(pool)
            b.reset_from_pool()
        o_obs, o_rew, o_af, o_ef = ob.rollout(acts, order=order, auto_reset=True)
        res = env.rollout(acts, order=order, auto_reset=True, want_compact=True)
        np.testing.assert_array_equal(_np(res.obs).view(np.uint32), o_obs.view(np.uint32))
        np.testing.assert_array_equal(_np(res.reward).view(np.uint64), o_rew.view(np.uint64))
        np.testing.assert_array_equal(_np(res.agent_flags), o_af)
        np.testing.assert_array_equal(_np(res.env_flags), o_ef)
        np.testing.assert_array_equal(_np(env.expand_observations(res.obs_compact)).view(np.uint32), o_obs.view(np.uint32))
        o_act, o_obs, o_rew, o_af, o_ef = ob.rollout_greedy(K, auto_reset=True)
        res, a = env.rollout_greedy(K, auto_reset=True)
        np.testing.assert_array_equal(_np(a), o_act)
        np.testing.assert_array_equal(_np(res.obs).view(np.uint32), o_obs.view(np.uint32))
        np.testing.assert_array_equal(_np(res.reward).view(np.uint64), o_rew.view(np.uint64))
        np.testing.assert_array_equal(_np(res.agent_flags), o_af)
        assert env.counters() == ob.counters.as_dict()
    finally:
        env.close()


# ---- round 3: start-up calibration of the pace controller, capture guard, launch-mode boundaries -------------------
def _shape_config(name):
    from bench import workload_config
    return workload_config(name)[0]


@pytest.mark.parametrize("workload,E,K", [("c2", 6001, 128), ("c3", 3000, 64), ("c5_50", 1000, 64)])
def test_a_new_shape_starts_within_five_percent_of_its_steady_state(ccx, workload, E, K):
    """the first paced launch of a shape nobody has seen measures its start value in-process (a
    ~2.5 ms write probe of the rollout's own observation buffer, ccx_api.hip: calibrate_pace).  Three shapes that never
    had a shipped or cached pace: launches 5-25 of a fresh handle run within 5 % of launches 80-120."""
    import torch
    cfg = _shape_config(workload)

    def fresh_handle():
        env = ccx(cfg, E)
        try:
            env.set_timing(True)
            env.make_reset_pool(0, 512)
            env.reset_from_pool()
            assert env.pace_start()["source"] == "assumed" and env.pace_state()["paced"] == 1.0
            acts = torch.randint(0, 5, (K, E, env.num_agents), dtype=torch.uint8, device=env.device)
            traj = env.alloc_rollout(K)
            ms = []
            for n in range(121):
                env.rollout(acts, auto_reset=True, out=traj)
                ms.append(env.last_launch_ms())
            start = env.pace_start()
            assert start["source"] == "calibration" and 3000 < start["probe_GBs"] < 9000 and start["ns"] > 0
            assert abs(env.step_pace_ns() / start["ns"] - 1.0) < 0.2, (env.step_pace_ns(), start)   # the probe was in the right place
            return float(np.mean(ms[5:26])), float(np.mean(ms[80:121])), start, ms[:30]
        finally:
            env.close()

    early, late, start, head = fresh_handle()
    if early > late * 1.05:                   # a timing test on a shared box: a second fresh handle before it counts
        early, late, start, head = fresh_handle()
    assert early <= late * 1.05, (workload, early, late, start, head)


def test_calibration_can_be_switched_off_and_a_callers_start_value_wins(ccx):
    import torch

    from bench import c2_config
    E, K = 4096, 64
    acts = None
    for mode in ("off", "caller"):
        env = ccx(c2_config(), E)
        try:
            if mode == "off":
                env.set_pace_calibration(False)
            else:
                env.set_step_pace_start(777.0)
            env.make_reset_pool(0, 256)
            env.reset_from_pool()
            acts = torch.randint(0, 5, (K, E, env.num_agents), dtype=torch.uint8, device=env.device)
            env.rollout(acts, auto_reset=True)
            st = env.pace_start()
            assert st["probe_GBs"] == 0.0
            assert st["source"] == ("assumed" if mode == "off" else "caller")
            if mode == "caller":
                assert abs(st["ns"] - 777.0) < 0.1
        finally:
            env.close()


def test_capturing_a_paced_rollout_before_its_controller_started_is_refused(ccx):
    """the (re)start of the pace controller (memsets, calibration) must never become part of a graph -- every
    replay would re-zero the controller.  A paced rollout captured right after a setting changed fails loudly; one eager
    launch later the same capture works and replays leave the controller alone."""
    import torch

    from bench import c2_config
    from collectivecrossing_amd._lib import CcxError
    E, K = 4096, 64
    env = ccx(c2_config(), E)
    env.make_reset_pool(0, 256)
    env.reset_from_pool()
    acts = torch.randint(0, 5, (K, E, env.num_agents), dtype=torch.uint8, device=env.device)
    traj = env.alloc_rollout(K)
    side = torch.cuda.Stream(device=env.device)
    env.use_stream(side)
    with torch.cuda.stream(side):
        env.rollout(acts, auto_reset=True, out=traj)
        env.set_tunable("pace_phase", 1)                      # restarts the controller
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with pytest.raises(CcxError, match="eager rollout"):
            with torch.cuda.graph(graph, stream=side):
                env.rollout(acts, auto_reset=True, out=traj)
        side.synchronize()
        env.rollout(acts, auto_reset=True, out=traj)          # eager: calibrates / restarts
        side.synchronize()
        before = env.pace_state()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            env.rollout(acts, auto_reset=True, out=traj)
        for _ in range(3):
            graph.replay()
        side.synchronize()
    after = env.pace_state()
    assert after["next_pace_ns"] == before["next_pace_ns"] and after["floor_ns"] == before["floor_ns"]
    env.close()


@pytest.mark.parametrize("K", [1, 15, 16, 63, 64, 65])
@pytest.mark.parametrize("pace", [0, -1, 760])
@pytest.mark.parametrize("obs", [True, False])
def test_launch_mode_boundaries_against_the_oracle(oracle, ccx, K, pace, obs):
    """VERDICT r2 item 7.  Whether a launch is paced (pace handle, observations, K >= 16), whether the controller adapts
    (K >= 64) and whether the sim wave hands steps over through sequence words or a barrier are decided by ONE set of
    helpers (ccx_kernels.h: launch_is_paced / launch_is_adaptive / launch_uses_flags) shared by run_rollout and the
    kernel.  Swept here across the boundaries, at a batch size that is paced at all: two launches each (the second one
    starts from the controller state the first one left), every output equal to the oracle."""
    from collectivecrossing_amd.reset import build_reset_pool
    g = Golden("g1_c1_random")
    E = 4096
    pool = build_reset_pool(g.config, 3, 64)
    rng = np.random.default_rng(K * 7 + pace)
    ob, env = oracle.OracleBatch(g.params, E), ccx(g.config, E)
    try:
        env.set_step_pace(pace)
        for b in (ob, env):
            b.set_reset_pool(pool)
            b.reset_from_pool()
        for launch in range(2):
            acts = rng.integers(0, 5, size=(K, E, g.N), dtype=np.uint8)
            o_obs, o_rew, o_af, o_ef = ob.rollout(acts, auto_reset=True)
            res = env.rollout(acts, auto_reset=True, want_obs=obs)
            np.testing.assert_array_equal(_np(res.agent_flags), o_af)
            np.testing.assert_array_equal(_np(res.env_flags), o_ef)
            np.testing.assert_array_equal(_np(res.reward).view(np.uint64), o_rew.view(np.uint64))
            if obs:
                np.testing.assert_array_equal(_np(res.obs).view(np.uint32), o_obs.view(np.uint32))
            del res, o_obs
        assert env.counters() == ob.counters.as_dict()
        paced = env.pace_state()["paced"] == 1.0
        assert paced == (pace != -1)
    finally:
        env.close()


def test_one_step_launches_take_liveness_from_the_state_at_kernel_entry(ccx):
    """Regression (round 3): writer 0 used to read terminated / truncated from the state arrays BEHIND the prologue's
    barrier, while the sim wave of a one-step launch may already be writing the post-step state back -- once the prologue
    got faster, a step that truncates everybody reported random agents as "not live before the step" (no reward /
    truncated entry in the dict API).  Every wave now reads the pair in front of the barrier.  300 single-step launches
    from the last step before truncation, tiny batches (the sim wave finishes first there) and a full one."""
    from collectivecrossing_amd import configs as C
    AF_TRUNC, AF_LIVE = 2, 4
    for E, reps in ((1, 200), (3, 60), (4096, 40)):
        cfg = C.CollectiveCrossingConfig(width=10, height=8, division_y=4, tram_door_left=4, tram_door_right=5, tram_length=8,
                                         num_boarding_agents=1, num_exiting_agents=1, exiting_destination_area_y=1,
                                         boarding_destination_area_y=7,
                                         truncated_config=C.MaxStepsTruncatedConfig(max_steps=65535))
        env = ccx(cfg, E)
        env.reset(np.arange(E, dtype=np.uint64))
        st = env.get_state()
        acts = np.full((E, 2), 4, np.uint8)
        for rep in range(reps):
            env.set_state(x=st["x"], y=st["y"], active=np.ones((E, 2), np.uint8), terminated=np.zeros((E, 2), np.uint8),
                          truncated=np.zeros((E, 2), np.uint8), step_count=np.full(E, 65534, np.int32))
            r = env.step(acts, want_obs=bool(rep & 1))
            af = r.agent_flags.cpu().numpy()
            assert ((af & (AF_TRUNC | AF_LIVE)) == (AF_TRUNC | AF_LIVE)).all(), (E, rep, af[:2])
            assert (r.env_flags.cpu().numpy() & 2).all()
        env.close()


@pytest.mark.parametrize("E", [257, 64, 1000])
def test_small_batches_of_tall_grids_keep_the_occupancy_tables(oracle, ccx, E):
    """Regression (round-3 hypothesis soak): the small-batch shape checked the LDS need of a full 64-lane tile against
    TWO writer slots while the writer rule chose four -- a 6 x 16 grid with one agent per env (64 envs per tile, 88 KB of
    occupancy tables) passed the check, lost the tables and refused every in-kernel policy rollout.  The check now uses
    the writer count that will be chosen; such batches fall back to smaller tiles and keep the tables."""
    from collectivecrossing_amd import configs as C
    from collectivecrossing_amd.params import lower_config
    from collectivecrossing_amd.reset import build_reset_pool

    cfg = C.CollectiveCrossingConfig(width=6, height=16, division_y=2, tram_door_left=0, tram_door_right=1, tram_length=4,
                                     num_boarding_agents=0, num_exiting_agents=1, exiting_destination_area_y=0,
                                     boarding_destination_area_y=10, truncated_config=C.MaxStepsTruncatedConfig(max_steps=30))
    pool = build_reset_pool(cfg, 3, 37)
    ob, env = oracle.OracleBatch(lower_config(cfg), E), ccx(cfg, E)
    try:
        for b in (ob, env):
            b.set_reset_pool(pool)
            b.reset_from_pool()
        for policy in ("greedy", "waiting"):
            o_act, o_obs, o_rew, o_af, o_ef = ob.rollout_greedy(40, auto_reset=True, policy=policy)
            res, acts = env.rollout_greedy(40, auto_reset=True, policy=policy)
            np.testing.assert_array_equal(acts.cpu().numpy(), o_act)
            np.testing.assert_array_equal(res.agent_flags.cpu().numpy(), o_af)
            np.testing.assert_array_equal(res.obs.cpu().numpy().view(np.uint32), o_obs.view(np.uint32))
            np.testing.assert_array_equal(res.reward.cpu().numpy().view(np.uint64), o_rew.view(np.uint64))
        assert env.counters() == ob.counters.as_dict()
    finally:
        env.close()


@pytest.mark.parametrize("cfg_name,E", [("g7_n3_small", 700), ("g7_n5_odd", 1000), ("g8_rollout_c1", 300), ("g8_rollout_c1", 1500)])
@pytest.mark.parametrize("K", [1, 2, 3, 16, 17, 34])
def test_two_step_row_writer_iterations_equal_the_oracle(oracle, ccx, cfg_name, E, K):
    """Small batches: row writers own two staging slots and take two env-steps per iteration whenever the sim wave is
    ahead (tunable pair_rows).  Step counts around the pairing (odd, even, one burst + 1), several agent
    counts (odd row lengths: the edge-iteration instantiation), shuffled move orders, auto-reset -- against the oracle;
    and with pairing switched off the same results."""
    g = Golden(cfg_name)
    for knobs in ({}, {"pair_rows": 0}, {"hand2": 2}):
        def setup(env, knobs=knobs):
            for k, v in knobs.items():
                env.set_tunable(k, v)
        c, shape, _, _ = _against_oracle(oracle, ccx, g, E=E, K=K, seed=5 * K + E, setup=setup, order=bool(K & 1))
        assert shape["writers_per_tile"] >= 2 and c["env_steps"] == E * K


def test_short_rollouts_of_large_tiles_are_adaptive_too(oracle, ccx):
    """The step counts from which a launch is paced / the controller adapts are per launch shape (~12 us / ~50 us of
    planned duration): 16 / 64 steps of the 4096 x 8 shape, 2 / 5 of the 4096 x 32 one.  Rounds 1-2 used 16 / 64 for every
    shape, so 20-step rollouts of C3 never calibrated and stayed at the assumed start pace (0.81 instead of 0.87 of the
    peak).  Here: C3 at 20 steps per launch calibrates at its first launch, its pace moves over the next ones, and the
    trajectory is the oracle's; the C2 shape keeps 16 / 64 (the boundary sweep above)."""
    import torch
    g = Golden("g3_c3_dense_simple_distance")
    E, K = 4096, 20
    env = ccx(g.config, E)
    env.make_reset_pool(0, 512)
    env.reset_from_pool()
    acts = torch.randint(0, 5, (K, E, g.N), dtype=torch.uint8, device=env.device)
    traj = env.alloc_rollout(K)
    assert env.pace_start()["source"] == "assumed"
    env.rollout(acts, auto_reset=True, out=traj)
    env.synchronize()
    ps = env.pace_start()
    assert ps["source"] == "calibration" and 3000 < ps["probe_GBs"] < 9000, ps
    paces = []
    for _ in range(12):
        env.rollout(acts, auto_reset=True, out=traj)
        paces.append(env.pace_state()["next_pace_ns"])
    assert len({round(p, 1) for p in paces}) > 3, (ps, paces)          # the controller moves (down on a healthy box)
    env.close()
    _against_oracle(oracle, ccx, g, E=1500, K=9, seed=77)             # (9 steps: adaptive for this shape) == the oracle


# ---- round 4: ADVICE r3 ---------------------------------------------------------------------------------------------
def test_a_short_paced_rollout_captures_after_one_eager_launch(ccx):
    """on the C2 shape launches of 16..63 steps are paced but not adaptive; with the calibration pending the
    handle stayed `dirty` after every such eager launch and refused the capture the caller had prepared for ("run one eager
    rollout of this shape before capturing").  The pending calibration is tracked apart from the controller's (re)start now:
    one eager K = 32 launch starts the controller, the same launch captures and replays, and the controller's state is
    untouched by the replays; a later eager adaptive launch still calibrates."""
    import torch

    from bench import c2_config
    E, K = 4096, 32
    env = ccx(c2_config(), E)
    env.make_reset_pool(0, 256)
    env.reset_from_pool()
    acts = torch.randint(0, 5, (64, E, env.num_agents), dtype=torch.uint8, device=env.device)
    traj = env.alloc_rollout(K)
    side = torch.cuda.Stream(device=env.device)
    env.use_stream(side)
    with torch.cuda.stream(side):
        env.rollout(acts[:K], auto_reset=True, out=traj)       # eager, paced, not adaptive: starts the controller
        side.synchronize()
        before = env.pace_state()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            env.rollout(acts[:K], auto_reset=True, out=traj)
        for _ in range(3):
            graph.replay()
        side.synchronize()
        after = env.pace_state()
        assert after["next_pace_ns"] == before["next_pace_ns"] and after["floor_ns"] == before["floor_ns"]
        assert env.pace_start()["source"] == "assumed"
        long_traj = env.alloc_rollout(64)
        env.rollout(acts, auto_reset=True, out=long_traj)      # the first adaptive eager launch calibrates and restarts
        side.synchronize()
        assert env.pace_start()["source"] == "calibration"
    env.close()


def test_cut_rollouts_of_an_odd_number_of_agent_slots_stay_aligned(ccx, oracle):
    """L = 6 + 4N is 2 mod 4, so a step's observation slab is a multiple of 16 bytes only if E x N is even; with
    E x N odd a cut after an odd number of steps handed the second sub-launch a misaligned slice ("obs buffer must be
    16-byte aligned") after the first had advanced the state.  Cuts fall on even steps now."""
    g = Golden("g7_n5_odd")
    E, K = 37, 23
    assert (E * g.N) % 2 == 1

    def setup(env):
        env.set_tunable("max_launch_steps", 7)

    _against_oracle(oracle, ccx, g, E, K, seed=11, setup=setup)


# ---- round 4: grids of several rounds -- partial last rounds and round-by-round launches ------------------------------
def test_a_partial_last_round_does_not_drive_the_common_pace_up(ccx):
    """The tiles of a partial last round run on a schedule scaled to their number, which their own step chain may not be
    able to follow (20 000 envs of C2: 226 of 1250 workgroups in the second round).  They voted 'late' all the same and the
    controller raised the COMMON pace until they were on time: 0.67 of the peak for 20 000 envs, 0.27 for 100 003 (pace
    3.5 x its value; profiles/r04_multi_round.txt).  Such tiles no longer vote: the pace of a ragged batch settles where the
    pace of the full batch next to it does."""
    import torch
    cfg = _shape_config("c2")
    paces = {}
    for E in (16384, 20000):
        env = ccx(cfg, E)
        try:
            env.make_reset_pool(0, 512)
            env.reset_from_pool()
            K = 64
            acts = torch.randint(0, 5, (K, E, env.num_agents), dtype=torch.uint8, device=env.device)
            traj = env.alloc_rollout(K)
            for _ in range(60):
                env.rollout(acts, auto_reset=True, out=traj)
            torch.cuda.synchronize()
            shape = env.launch_shape()
            paces[E] = env.step_pace_ns()
            if E == 20000:
                assert shape["num_blocks"] > shape["resident_blocks"] > 0 and shape["num_blocks"] % shape["resident_blocks"] != 0
        finally:
            env.close()
    assert paces[16384] > 0 and paces[20000] < 1.15 * paces[16384], paces


@pytest.mark.parametrize("mode", [0, 2])
def test_round_by_round_launches_equal_the_oracle(oracle, ccx, mode):
    """A grid of more workgroups than the device holds is launched one round per launch when its rows exceed the reach of
    the ~4 GB footprint knee (ccx_api.hip: run_rollout, KParams::block_base / launch_flags; tunable round_launches: 0 = one
    launch, 2 = always by rounds): 6001 envs of the C3 geometry in 3001 tiles -- ragged last tile, partial last round -- are
    bit-equal to the oracle either way, and the counters add up over the rounds."""
    g = Golden("g3_c3_dense_simple_distance")
    c, shape, _, _ = _against_oracle(oracle, ccx, g, E=6001, K=17, seed=35, order=False,
                                     setup=lambda env: env.set_tunable("round_launches", mode))
    assert shape["num_blocks"] > shape["resident_blocks"] > 0


def test_round_by_round_launches_of_small_tiles_equal_the_oracle(oracle, ccx):
    """The same for the bench's geometry (C2: two 64-lane tiles per workgroup, one throttled writer, the pace controller's
    votes collected over the rounds): 40 003 envs -- 2501 workgroups on 1024 slots, a ragged last tile, a partial last round --
    with a shuffled move order (the instantiations that are not PLAIN carry block_base too), by rounds and in one launch."""
    from types import SimpleNamespace

    from collectivecrossing_amd.params import lower_config
    cfg = _shape_config("c2")
    g = SimpleNamespace(config=cfg, params=lower_config(cfg), N=8)
    for mode, order in ((2, False), (2, True), (0, True)):
        c, shape, _, _ = _against_oracle(oracle, ccx, g, E=40003, K=20, seed=36 + mode, order=order,
                                         setup=lambda env: env.set_tunable("round_launches", mode))
        assert shape["num_blocks"] > 2 * shape["resident_blocks"] > 0


def test_large_batches_of_single_agent_envs_run_in_one_round_and_equal_the_oracle(oracle, ccx):
    """One agent per env, 64 envs per wavefront: the LDS occupancy tables (one per env) allowed ONE tile per CU, so more than
    64 x CUs envs ran in rounds -- 17 768 envs at twice the time per env-step of 15 800 (profiles/r04_cliff_scan.txt).  Such a
    batch now takes the all-pairs instantiations (an agent alone in its env collides with nobody): everything is resident, the
    trajectory is the oracle's, and an explicit request keeps either path (tunable occ_tables)."""
    from types import SimpleNamespace

    from collectivecrossing_amd import configs as C
    from collectivecrossing_amd.params import lower_config
    cfg = C.CollectiveCrossingConfig(width=12, height=8, division_y=4, tram_door_left=5, tram_door_right=7, tram_length=9,
                                     num_boarding_agents=1, num_exiting_agents=0, exiting_destination_area_y=0,
                                     boarding_destination_area_y=8, truncated_config=C.MaxStepsTruncatedConfig(max_steps=30))
    g = SimpleNamespace(config=cfg, params=lower_config(cfg), N=1)
    c, shape, _, _ = _against_oracle(oracle, ccx, g, E=20011, K=40, seed=41, order=False)
    assert shape["num_blocks"] == shape["resident_blocks"] > 256
    c1, shape1, _, _ = _against_oracle(oracle, ccx, g, E=20011, K=40, seed=41, order=False,
                                       setup=lambda env: env.set_tunable("occ_tables", 1))
    assert shape1["num_blocks"] > shape1["resident_blocks"] and c1 == c
