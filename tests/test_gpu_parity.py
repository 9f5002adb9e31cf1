"""GPU parity: libccx (HIP, through the C-ABI) vs the reference-recorded goldens and the CPU oracle.

Bit-exact everywhere: positions, flags, observations (u32 bit patterns), rewards (f64 bit
patterns, including the sign of zero).
"""

import numpy as np
import pytest
from _fixtures import ROLLOUT_NPZ, STEP_NPZ, Golden, assert_step_matches

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ccx():
    import torch

    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    from collectivecrossing_amd.batched import BatchedCollectiveCrossing

    return BatchedCollectiveCrossing


def _np(t):
    return None if t is None else t.cpu().numpy()


def _check_rollout_vs_golden(g, res, state, counters=None):
    obs, rew, af, ef = _np(res.obs), _np(res.reward), _np(res.agent_flags), _np(res.env_flags)
    np.testing.assert_array_equal(af, g["agent_flags"], err_msg=f"{g.name} agent_flags")
    np.testing.assert_array_equal(ef, g["env_flags"], err_msg=f"{g.name} env_flags")
    np.testing.assert_array_equal(obs.view(np.uint32), g["obs"].view(np.uint32), err_msg=f"{g.name} obs")
    live = (g["agent_flags"] & 4) != 0
    np.testing.assert_array_equal(np.where(live, rew, 0).view(np.uint64),
                                  np.where(live, g["reward"], 0).view(np.uint64), err_msg=f"{g.name} reward")
    for k in ("x", "y", "active", "terminated", "truncated"):
        np.testing.assert_array_equal(state[k], g[k][-1], err_msg=f"{g.name} final {k}")


@pytest.mark.parametrize("name", STEP_NPZ)
def test_step_by_step_matches_reference_vectors(ccx, name):
    """ccx_step, one launch per step, against every reference-recorded step."""
    g = Golden(name)
    env = ccx(g.config, g.E)
    env.set_state(**g.init_state())
    for s in range(g.K):
        r = env.step(g["actions"][s], g["order"][s])
        assert_step_matches(g, s, _np(r.obs), _np(r.reward), _np(r.agent_flags), _np(r.env_flags),
                            env.get_state())
    env.close()


@pytest.mark.parametrize("name", STEP_NPZ)
def test_fused_rollout_matches_reference_vectors(ccx, name):
    """ccx_rollout (state in registers for K steps) against the same vectors."""
    g = Golden(name)
    env = ccx(g.config, g.E)
    env.set_state(**g.init_state())
    res = env.rollout(g["actions"], g["order"])
    _check_rollout_vs_golden(g, res, env.get_state())
    c = env.counters()
    assert c["env_steps"] == g.K * g.E and c["agent_steps"] == g.K * g.E * g.N
    assert c["live_agent_steps"] == int(((g["agent_flags"] & 4) != 0).sum())
    env.close()


@pytest.mark.parametrize("name", ["g1_c1_random", "g2_c1_shuffled_absent", "g3_c3_dense_shuffled",
                                  "g7_n3_small", "g7_n5_odd", "g7_n12_constant_negative"])
@pytest.mark.parametrize("shape", [(0, 1), (0, 4), (64, 2), (-1, 3)])
def test_launch_shapes_do_not_change_results(ccx, name, shape):
    """Every lanes-per-wave / waves-per-block choice gives identical bits."""
    g = Golden(name)
    env = ccx(g.config, g.E)
    lanes, wpb = shape
    group = env.launch_shape()["group_lanes"]
    if lanes == -1:
        lanes = group            # one env per wavefront
    elif lanes == 64:
        lanes = (64 // group) * group
    env.set_launch_shape(lanes, wpb)
    env.set_state(**g.init_state())
    res = env.rollout(g["actions"], g["order"])
    _check_rollout_vs_golden(g, res, env.get_state())
    env.close()


@pytest.mark.parametrize("name", ["g1_c1_random", "g2_c1_shuffled_absent", "g3_c3_dense_shuffled",
                                  "g4_c5_all_at_dest_greedy_32_32", "g7_n5_odd", "g7_n50_padded_group"])
@pytest.mark.parametrize("writers", [1, 2, 3])
def test_writer_wave_count_does_not_change_results(ccx, name, writers):
    """1, 2 or 3 writer wavefronts per env tile split the observation stores differently."""
    g = Golden(name)
    env = ccx(g.config, g.E)
    env.set_writers(writers)
    env.set_state(**g.init_state())
    res = env.rollout(g["actions"], g["order"])
    _check_rollout_vs_golden(g, res, env.get_state())
    env.close()


@pytest.mark.parametrize("name", ["g1_c1_random", "g3_c3_dense_shuffled", "g7_n50_padded_group"])
@pytest.mark.parametrize("writers,throttle", [(1, 1), (1, 16), (2, 2), (3, 63), (1, -1)])
def test_store_throttle_does_not_change_results(ccx, name, writers, throttle):
    """The store throttle only bounds how many observation stores a writer keeps in flight."""
    g = Golden(name)
    env = ccx(g.config, g.E)
    env.set_writers(writers)
    env.set_store_throttle(throttle)
    shape = env.launch_shape()
    assert shape["writers_per_tile"] == writers
    assert shape["store_throttle"] == (0 if throttle < 0 else throttle)
    env.set_state(**g.init_state())
    res = env.rollout(g["actions"], g["order"])
    _check_rollout_vs_golden(g, res, env.get_state())
    with pytest.raises(Exception):
        env.set_store_throttle(64)
    env.close()


@pytest.mark.parametrize("name", ["g1_c1_random", "g3_c3_dense_shuffled", "g7_n50_padded_group"])
@pytest.mark.parametrize("pace", [0, -1, 40, 3000])
def test_step_pacing_does_not_change_results(ccx, name, pace):
    """Step pacing only delays when a tile starts an env-step (adaptive, off, far too fast, slow)."""
    g = Golden(name)
    env = ccx(g.config, g.E)
    env.set_step_pace(pace)
    if pace > 0:
        assert abs(env.step_pace_ns() - pace) <= 0.05
    if pace == -1:
        assert env.step_pace_ns() == 0.0
    env.set_state(**g.init_state())
    res = env.rollout(g["actions"], g["order"])
    _check_rollout_vs_golden(g, res, env.get_state())
    if pace > 0:
        assert abs(env.step_pace_ns() - pace) <= 0.05      # a fixed pace is never retuned
    with pytest.raises(Exception):
        env.set_step_pace(-2)
    env.close()


def test_adaptive_pace_settles_near_the_drain_rate(ccx):
    """C2-sized rollouts: after a few launches the adaptive pace sits between the controller's bounds
    and within a factor of two of bytes-per-step / 8 TB/s."""
    import torch

    g = Golden("g1_c1_random")
    E, K = 4096, 128
    env = ccx(g.config, E)
    env.make_reset_pool(0, 512, on_device=True)
    env.reset_from_pool()
    acts = torch.randint(0, 5, (K, E, g.N), dtype=torch.uint8, device=env.device)
    traj = env.alloc_rollout(K)
    start = env.step_pace_ns()
    for _ in range(12):
        env.rollout(acts, auto_reset=True, out=traj)
    settled = env.step_pace_ns()
    floor_ns = E * g.N * (4 * (6 + 4 * g.N) + 10) / 8000.0      # bytes per env-step / (8 TB/s in B/ns)
    assert start > 0 and settled != start
    assert floor_ns * 0.9 <= settled <= floor_ns * 2.0, (start, settled, floor_ns)
    env.close()


def test_huge_grid_uses_the_all_pairs_fallback_and_big_lds(oracle, ccx):
    """100x100 grid: the occupancy tables do not fit in LDS (all-pairs path) and the cell table
    alone needs > 64 KiB of dynamic LDS.  Checked against the oracle, with shuffled move order."""
    from collectivecrossing_amd import configs as C
    from collectivecrossing_amd.params import lower_config
    from collectivecrossing_amd.reset import seeded_positions

    cfg = C.CollectiveCrossingConfig(
        width=100, height=100, division_y=50, tram_door_left=40, tram_door_right=60, tram_length=100,
        num_boarding_agents=12, num_exiting_agents=9, exiting_destination_area_y=48,
        boarding_destination_area_y=52, truncated_config=C.MaxStepsTruncatedConfig(max_steps=40))
    E, K, N = 37, 50, 21
    rng = np.random.default_rng(2)
    actions = rng.integers(0, 5, size=(K, E, N), dtype=np.uint8)
    order = np.argsort(rng.random((K, E, N)), axis=-1).astype(np.uint8)
    pos = seeded_positions(cfg, range(E))
    # crowd everybody next to the door so that conflicts actually happen
    pos[:, :, 0] = 45 + (np.arange(N) % 7)[None, :]
    pos[:, :12, 1] = 47 + (np.arange(12) // 7)[None, :]
    pos[:, 12:, 1] = 51 + (np.arange(9) // 7)[None, :]
    for o in (None, order):
        ob = oracle.OracleBatch(lower_config(cfg), E)
        env = ccx(cfg, E)
        ob.set_state(x=pos[..., 0], y=pos[..., 1])
        env.set_state(x=pos[..., 0], y=pos[..., 1])
        o_obs, o_rew, o_af, o_ef = ob.rollout(actions, o)
        res = env.rollout(actions, o)
        np.testing.assert_array_equal(_np(res.agent_flags), o_af)
        np.testing.assert_array_equal(_np(res.env_flags), o_ef)
        np.testing.assert_array_equal(_np(res.obs).view(np.uint32), o_obs.view(np.uint32))
        np.testing.assert_array_equal(_np(res.reward).view(np.uint64), o_rew.view(np.uint64))
        assert env.counters() == ob.counters.as_dict()
        assert env.counters()["moves"] > 0
        env.close()


@pytest.mark.parametrize("name", ROLLOUT_NPZ)
def test_autoreset_rollout_matches_reference(ccx, name):
    g = Golden(name)
    env = ccx(g.config, g.E, env_offset=int(g["env_offset"]), total_envs=int(g["total_envs"]))
    env.set_reset_pool(g["pool_xy"])
    env.reset_from_pool()
    st = env.get_state()
    np.testing.assert_array_equal(st["x"], g["init_x"])
    np.testing.assert_array_equal(st["y"], g["init_y"])
    res = env.rollout(g["actions"], None, auto_reset=True)
    np.testing.assert_array_equal(_np(res.env_flags), g["env_flags"])
    np.testing.assert_array_equal(_np(res.agent_flags), g["agent_flags"])
    np.testing.assert_array_equal(_np(res.obs).view(np.uint32), g["obs"].view(np.uint32))
    live = (g["agent_flags"] & 4) != 0
    np.testing.assert_array_equal(np.where(live, _np(res.reward), 0).view(np.uint64),
                                  np.where(live, g["reward"], 0).view(np.uint64))
    st = env.get_state()
    np.testing.assert_array_equal(st["episode"], g["final_episode"])
    assert env.counters()["episodes"] == int(g["final_episode"].sum())
    # split the same rollout into two launches: state + episode cursor survive the kernel boundary
    env2 = ccx(g.config, g.E, env_offset=int(g["env_offset"]), total_envs=int(g["total_envs"]))
    env2.set_reset_pool(g["pool_xy"])
    env2.reset_from_pool()
    h = g.K // 3
    r1 = env2.rollout(g["actions"][:h], None, auto_reset=True)
    r2 = env2.rollout(g["actions"][h:], None, auto_reset=True)
    np.testing.assert_array_equal(np.concatenate([_np(r1.env_flags), _np(r2.env_flags)]), g["env_flags"])
    np.testing.assert_array_equal(np.concatenate([_np(r1.obs), _np(r2.obs)]).view(np.uint32),
                                  g["obs"].view(np.uint32))
    env.close()
    env2.close()


def test_seeded_reset_and_observe_match_reference(ccx):
    """reset(seed) placement + the observation reset() returns (collectivecrossing.py:153-159)."""
    for name in ("g1_c1_random", "g7_n3_small", "g3_c3_dense_simple_distance"):
        g = Golden(name)
        env = ccx(g.config, g.E)
        obs = _np(env.reset(g["seeds"]))
        st = env.get_state()
        np.testing.assert_array_equal(st["x"], g["init_x"])
        np.testing.assert_array_equal(st["y"], g["init_y"])
        np.testing.assert_array_equal(obs[:, :, 0], g["init_x"].astype(np.float32))
        np.testing.assert_array_equal(obs[:, :, 1], g["init_y"].astype(np.float32))
        env.close()


@pytest.mark.parametrize("name", ["g1_c1_random", "g3_c3_dense_simple_distance", "g4_c5_all_at_dest_greedy_32_32",
                                  "g7_n1_exiting_only", "g7_n5_odd", "g5_sealed_door_greedy", "g7_n50_padded_group"])
def test_device_seeded_placement_equals_numpy_stream(ccx, name):
    """ccx_fill_reset_pool_seeded / ccx_reset_seeded (SeedSequence -> PCG64 -> Lemire on the GPU)
    against numpy itself (reset.py, which is pinned to the reference's placements), 20k seeds
    including seeds >= 2^32."""
    import torch

    from collectivecrossing_amd.reset import build_reset_pool, seeded_positions

    g = Golden(name)
    P = 20000 if g.N <= 8 else 1500
    env = ccx(g.config, 64)
    env.make_reset_pool(1000, P)
    np.testing.assert_array_equal(_np(env.reset_pool()), build_reset_pool(g.config, 1000, P))
    big = [0, 2**32 - 1, 2**32, 2**32 + 1, 2**40 + 12345, 2**63 + 7, 2**64 - 1, 123456789012345678]
    seeds = np.array(big + list(range(77, 77 + 64 - len(big))), dtype=np.uint64)
    obs = _np(env.reset(seeds))
    st = env.get_state()
    pos = seeded_positions(g.config, [int(v) for v in seeds])
    np.testing.assert_array_equal(st["x"], pos[..., 0])
    np.testing.assert_array_equal(st["y"], pos[..., 1])
    np.testing.assert_array_equal(obs[:, :, 0], pos[..., 0].astype(np.float32))
    assert (st["active"] == 1).all() and (st["step_count"] == 0).all()
    # masked reset leaves the other envs alone
    env.step(np.full((64, g.N), 4, np.uint8))
    mask = (np.arange(64) % 3 == 0).astype(np.uint8)
    env.reset(seeds[::-1].copy(), env_mask=mask)
    st2 = env.get_state()
    pos2 = seeded_positions(g.config, [int(v) for v in seeds[::-1]])
    np.testing.assert_array_equal(st2["x"][mask == 1], pos2[mask == 1][..., 0])
    np.testing.assert_array_equal(st2["x"][mask == 0], st["x"][mask == 0])
    assert (st2["step_count"][mask == 1] == 0).all() and (st2["step_count"][mask == 0] == 1).all()
    env.close()


def test_device_placement_reports_impossible_configs(ccx):
    """More agents than free cells: the reference would spin forever; libccx reports it."""
    from collectivecrossing_amd import configs as C
    from collectivecrossing_amd._lib import CcxError

    kw = dict(width=3, height=4, division_y=2, tram_door_left=0, tram_door_right=1, tram_length=2,
              num_boarding_agents=0, num_exiting_agents=3, exiting_destination_area_y=0,
              boarding_destination_area_y=3, observation_config=C.DefaultObservationConfig(),
              reward_config=C.DefaultRewardConfig(), terminated_config=C.IndividualAtDestinationTerminatedConfig(),
              truncated_config=C.MaxStepsTruncatedConfig(max_steps=5), render_mode=None)
    cfg = C.CollectiveCrossingConfig.model_construct(**kw)   # tram interior is x=1 only, rows 2..3 -> 1 legal cell
    env = ccx(cfg, 2)
    with pytest.raises(CcxError, match="no free cell"):
        env.reset(np.array([1, 2], dtype=np.uint64))
    env.close()


@pytest.mark.parametrize("name", [n for n in STEP_NPZ if "greedy" in n])
def test_greedy_policy_kernel_matches_the_reference_policy(ccx, name):
    """ccx_greedy_actions against the actions the reference's GreedyPolicy(epsilon=0) emitted."""
    g = Golden(name)
    env = ccx(g.config, g.E)
    env.set_state(**g.init_state())
    for s in range(g.K):
        np.testing.assert_array_equal(_np(env.greedy_actions()), g["actions"][s], err_msg=f"{name} step {s}")
        env.step(g["actions"][s], g["order"][s], want_obs=False)
    env.close()


@pytest.mark.parametrize("cfg_name,E,K", [("g4_c5_all_at_dest_greedy_32_32", 64, 120), ("g1_c1_random", 512, 150),
                                          ("g3_c3_dense_simple_distance", 128, 130), ("g7_n5_odd", 100, 120)])
def test_greedy_closed_loop_equals_oracle(oracle, ccx, cfg_name, E, K):
    """policy -> step -> policy ... on the GPU vs the oracle, from seeded resets (dense door jams)."""
    from collectivecrossing_amd.reset import seeded_positions

    g = Golden(cfg_name)
    pos = seeded_positions(g.config, range(9000, 9000 + E))
    ob = oracle.OracleBatch(g.params, E)
    env = ccx(g.config, E)
    ob.set_state(x=pos[..., 0], y=pos[..., 1])
    env.set_state(x=pos[..., 0], y=pos[..., 1])
    for s in range(K):
        a_o = ob.greedy_actions()
        a_g = env.greedy_actions()
        np.testing.assert_array_equal(_np(a_g), a_o, err_msg=f"step {s}")
        o = ob.step(a_o, want_obs=False)
        r = env.step(a_g, want_obs=False)
        np.testing.assert_array_equal(_np(r.agent_flags), o[2])
    st = env.get_state()
    np.testing.assert_array_equal(st["x"], ob.x)
    np.testing.assert_array_equal(st["y"], ob.y)
    assert ob.counters.arrivals > 0
    env.close()


@pytest.mark.parametrize("name", [n for n in STEP_NPZ if "greedy" in n])
def test_fused_greedy_rollout_reproduces_reference_episodes(ccx, name):
    """ccx_rollout_policy: the whole policy -> step loop in ONE launch reproduces the episodes the
    reference produced with GreedyPolicy(epsilon=0) + env.step (actions and every output)."""
    g = Golden(name)
    env = ccx(g.config, g.E)
    env.set_state(**g.init_state())
    res, acts = env.rollout_greedy(g.K)
    np.testing.assert_array_equal(_np(acts), g["actions"])
    _check_rollout_vs_golden(g, res, env.get_state())
    env.close()


@pytest.mark.parametrize("name", [n for n in STEP_NPZ if "waiting" in n])
def test_waiting_policy_kernel_and_fused_rollout_match_the_reference(ccx, name):
    """WaitingPolicy(epsilon=0): stand-alone kernel step by step and the fused rollout against the
    episodes the reference produced with its own policy."""
    g = Golden(name)
    env = ccx(g.config, g.E)
    env.set_state(**g.init_state())
    for s in range(g.K):
        np.testing.assert_array_equal(_np(env.policy_actions("waiting")), g["actions"][s], err_msg=f"{name} step {s}")
        env.step(g["actions"][s], g["order"][s], want_obs=False)
    env.set_state(**g.init_state())
    res, acts = env.rollout_greedy(g.K, policy="waiting")
    np.testing.assert_array_equal(_np(acts), g["actions"])
    _check_rollout_vs_golden(g, res, env.get_state())
    env.close()


@pytest.mark.parametrize("policy", ["greedy", "waiting"])
@pytest.mark.parametrize("cfg_name,E,K,writers", [("g4_c5_all_at_dest_greedy_32_32", 96, 140, 0),
                                                  ("g4_c5_all_at_dest_greedy_25_25", 64, 100, 2),
                                                  ("g1_c1_random", 1024, 260, 0), ("g7_n5_odd", 333, 200, 1),
                                                  ("g8_rollout_small_all_at_dest", 500, 150, 0)])
def test_fused_greedy_rollout_with_autoreset_equals_oracle(oracle, ccx, cfg_name, E, K, writers, policy):
    from collectivecrossing_amd.reset import build_reset_pool

    g = Golden(cfg_name)
    pool = build_reset_pool(g.config, 4242, 211)
    ob = oracle.OracleBatch(g.params, E)
    env = ccx(g.config, E)
    if writers:
        env.set_writers(writers)
    for b in (ob, env):
        b.set_reset_pool(pool)
        b.reset_from_pool()
    o_act, o_obs, o_rew, o_af, o_ef = ob.rollout_greedy(K, auto_reset=True, policy=policy)
    res, acts = env.rollout_greedy(K, auto_reset=True, policy=policy)
    np.testing.assert_array_equal(_np(acts), o_act)
    np.testing.assert_array_equal(_np(res.agent_flags), o_af)
    np.testing.assert_array_equal(_np(res.env_flags), o_ef)
    np.testing.assert_array_equal(_np(res.obs).view(np.uint32), o_obs.view(np.uint32))
    np.testing.assert_array_equal(_np(res.reward).view(np.uint64), o_rew.view(np.uint64))
    assert env.counters() == ob.counters.as_dict()
    assert ob.counters.moves > 0
    env.close()


# ---- against the oracle at sizes the goldens do not reach -------------------------------------
def _random_case(oracle, ccx, cfg_name, E, K, seed, shuffle, auto_reset, shape=None, p_absent=0.0,
                 writers=0, throttle=0):
    from collectivecrossing_amd.reset import build_reset_pool, seeded_positions

    g = Golden(cfg_name)
    rng = np.random.default_rng(seed)
    N = g.N
    actions = rng.integers(0, 5, size=(K, E, N), dtype=np.uint8)
    if p_absent:
        actions[rng.random((K, E, N)) < p_absent] = 255
    order = None
    if shuffle:
        order = np.argsort(rng.random((K, E, N)), axis=-1).astype(np.uint8)
    ob = oracle.OracleBatch(g.params, E)
    env = ccx(g.config, E)
    if shape:
        env.set_launch_shape(*shape)
    if writers:
        env.set_writers(writers)
    if throttle:
        env.set_store_throttle(throttle)
    if auto_reset:
        pool = build_reset_pool(g.config, 7000 + seed, 257)
        ob.set_reset_pool(pool)
        env.set_reset_pool(pool)
        ob.reset_from_pool()
        env.reset_from_pool()
    else:
        pos = seeded_positions(g.config, range(seed, seed + E))
        ob.set_state(x=pos[..., 0], y=pos[..., 1])
        env.set_state(x=pos[..., 0], y=pos[..., 1])
    o_obs, o_rew, o_af, o_ef = ob.rollout(actions, order, auto_reset=auto_reset)
    res = env.rollout(actions, order, auto_reset=auto_reset)
    np.testing.assert_array_equal(_np(res.agent_flags), o_af)
    np.testing.assert_array_equal(_np(res.env_flags), o_ef)
    np.testing.assert_array_equal(_np(res.obs).view(np.uint32), o_obs.view(np.uint32))
    np.testing.assert_array_equal(_np(res.reward).view(np.uint64), o_rew.view(np.uint64))
    st = env.get_state()
    for k in ("x", "y", "active", "terminated", "truncated", "step_count", "episode"):
        np.testing.assert_array_equal(st[k], getattr(ob, k), err_msg=k)
    c = env.counters()
    assert c == ob.counters.as_dict()
    env.close()
    return c


@pytest.mark.parametrize("cfg_name,E,K", [
    ("g1_c1_random", 1000, 130), ("g1_c1_random", 4096, 40),
    ("g3_c3_dense_simple_distance", 300, 120), ("g4_c5_all_at_dest_greedy_25_25", 70, 80),
    ("g4_c5_all_at_dest_greedy_32_32", 40, 60), ("g7_n5_odd", 333, 120), ("g7_n1_boarding_only", 200, 60)])
@pytest.mark.parametrize("shuffle", [False, True])
def test_rollout_equals_oracle(oracle, ccx, cfg_name, E, K, shuffle):
    _random_case(oracle, ccx, cfg_name, E, K, seed=11, shuffle=shuffle, auto_reset=False, p_absent=0.05)


@pytest.mark.parametrize("cfg_name,E,K", [("g8_rollout_c1", 777, 160), ("g8_rollout_small_all_at_dest", 501, 90),
                                          ("g7_n5_odd", 129, 260)])
def test_autoreset_rollout_equals_oracle(oracle, ccx, cfg_name, E, K):
    c = _random_case(oracle, ccx, cfg_name, E, K, seed=5, shuffle=False, auto_reset=True)
    assert c["episodes"] > 0


FUZZ = [n for n in STEP_NPZ if n.startswith("g10_fuzz_")]


@pytest.mark.parametrize("name", FUZZ[::2])
def test_fuzz_configs_at_batch_sizes_the_goldens_do_not_reach(oracle, ccx, name):
    """Random reference-valid configs (the g10 family) at a few hundred envs with shuffled move
    order, omitted agents and auto-reset: kernel vs oracle, bit-exact."""
    k = FUZZ.index(name)
    _random_case(oracle, ccx, name, E=130 + 37 * k, K=70, seed=40 + k, shuffle=bool(k & 2), auto_reset=bool(k & 4),
                 p_absent=0.08 if k % 3 == 0 else 0.0)


@pytest.mark.parametrize("seed", range(28))
def test_random_batch_sizes_and_launch_shapes_equal_oracle(oracle, ccx, seed):
    """API fuzz: random env counts (partial last tiles, slab strides that are not line multiples),
    step counts, tile shapes, tiles per workgroup, writer counts and store throttles."""
    rng = np.random.default_rng(1000 + seed)
    name = ["g1_c1_random", "g7_n3_small", "g7_n5_odd", "g7_n12_constant_negative", "g3_c3_dense_simple_distance",
            "g7_n50_padded_group", "g4_c5_all_at_dest_greedy_32_32", "g10_fuzz_05", "g10_fuzz_14",
            "g10_fuzz_17"][seed % 10]
    N = Golden(name).N
    G = 1
    while G < N:
        G *= 2
    lanes = int(rng.choice([0, 0, G, 64, G * max(1, (64 // G) // 2)]))
    E = int(rng.choice([1, 2, 7, 31, 33, 64, 100, 255, 257, 600, int(rng.integers(1, 900))]))
    if N >= 32:
        E = min(E, 150)
    _random_case(oracle, ccx, name, E=E, K=int(rng.integers(1, 70)), seed=seed, shuffle=bool(rng.integers(0, 2)),
                 auto_reset=bool(rng.integers(0, 2)), shape=(lanes, int(rng.integers(0, 5))),
                 p_absent=float(rng.choice([0.0, 0.1])), writers=int(rng.integers(0, 5)),
                 throttle=int(rng.choice([0, -1, 3, 16, 40])))


def test_arbitrary_valid_configs_kernel_equals_oracle(oracle, ccx):
    """Hypothesis over the config space (the generator of tests/test_oracle_properties.py, plus up to
    40 agents): grids, door widths, agent mixes, every strategy -- kernel vs oracle, bit-exact, with
    shuffled move order, omitted agents and auto-reset."""
    from hypothesis import HealthCheck, given, settings
    from hypothesis import strategies as st
    from test_oracle_properties import configs

    from collectivecrossing_amd.params import lower_config
    from collectivecrossing_amd.reset import build_reset_pool

    import os

    soak = int(os.environ.get("CCX_HYP_EXAMPLES", "0"))     # a one-off soak: CCX_HYP_EXAMPLES=1000 (randomised)
    # (a soak of the small-batch shapes -- half tiles, role split, two-step row writers -- draws larger batches:
    #  CCX_HYP_ENVS=257,600,1025,2048,3000)
    e_list = [int(v) for v in os.environ.get("CCX_HYP_ENVS", "1,5,64,130,257").split(",")]

    @settings(max_examples=soak or 50, deadline=None, derandomize=not soak,
              suppress_health_check=[HealthCheck.too_slow, HealthCheck.filter_too_much, HealthCheck.function_scoped_fixture])
    @given(cfg=st.one_of(configs(), configs(max_boarding=25, max_exiting=25)), seed=st.integers(0, 2**20),
           E=st.sampled_from(e_list), K=st.integers(1, 48),
           mode=st.sampled_from(["actions", "actions", "greedy", "waiting", "random"]), compact=st.booleans(),
           writers=st.sampled_from([0, 0, 1, 2, 3, 4]), roles=st.sampled_from([-1, -1, 0, 1]),
           hand2=st.sampled_from([1, 1, 0, 2]), full_tiles=st.booleans(), eps=st.sampled_from([0.0, 0.0, 0.1, 0.5, 1.0]),
           mt=st.sampled_from([False, False, True]), pair_rows=st.sampled_from([-1, -1, 0]),
           step=st.sampled_from([(-1, 0, 0), (-1, 0, 0), (0, 0, 0), (1, 1, 0), (1, 2, 64), (1, 3, 16), (1, 5, 8), (1, 7, 0)]),
           small_shape=st.sampled_from([1, 1, 0]), tables=st.sampled_from([0, 0, 0, 1, 2, 3]), occ=st.sampled_from([-1, -1, -1, 0]))
    def run(cfg, seed, E, K, mode, compact, writers, roles, hand2, full_tiles, eps, mt, pair_rows, step, small_shape, tables, occ):
        p = lower_config(cfg)
        N = p.num_boarding + p.num_exiting
        rng = np.random.default_rng(seed)
        actions = rng.integers(0, 5, size=(K, E, N), dtype=np.uint8)
        actions[rng.random((K, E, N)) < 0.05] = 255
        order = np.argsort(rng.random((K, E, N)), axis=-1).astype(np.uint8) if seed & 1 else None
        pool = build_reset_pool(cfg, seed % 977, 37)
        ob, env = oracle.OracleBatch(p, E), ccx(cfg, E)
        try:
            # launch shape: writer waves per tile, their role split, paired hand-offs, 64-lane tiles
            if writers:
                env.set_writers(writers)
            env.set_tunable("writer_roles", roles)
            env.set_tunable("hand2", hand2)
            env.set_tunable("pair_rows", pair_rows)
            # round 4: the short-launch kernel (on / off, row waves, lanes), the no-rows launch shape, user tables
            env.set_tunable("step_kernel", step[0])
            env.set_tunable("step_rows", step[1])
            env.set_tunable("step_lanes", step[2])
            env.set_tunable("small_shape", small_shape)
            env.set_tunable("occ_tables", occ)            # (0: the all-pairs instantiations of the rollout kernel whatever the grid)
            if tables and (cfg.width + 3) * (cfg.height + 3) <= 1200:
                shape = (cfg.height + 1, cfg.width + 1)
                rew = (rng.normal(size=shape), rng.normal(size=shape)) if tables & 1 else None
                term = ((rng.random(shape) < 0.15), (rng.random(shape) < 0.15)) if (tables & 2 and p.terminated_mode == 0) else None
                if rew is not None:
                    env.set_reward_table(*rew)
                if term is not None:
                    env.set_terminated_table(*term)
                ob.set_user_tables(reward=rew, terminated=term)
            if full_tiles and mode in ("actions", "random"):   # (the scripted policies need the LDS occupancy tables)
                try:
                    env.set_launch_shape(64, 0)
                except Exception:          # a grid whose tables for 64 lanes of envs exceed the LDS: the library's shape
                    env.set_launch_shape(0, 0)
            for b in (ob, env):
                b.set_reset_pool(pool)
                b.reset_from_pool()
            # compact: the rollout writes only the [E][N][4] observation rows; they are expanded afterwards
            out = env.alloc_rollout(K, want_obs=not compact, want_compact=compact)
            if mode == "actions":
                o_obs, o_rew, o_af, o_ef = ob.rollout(actions, order, auto_reset=True)
                res = env.rollout(actions, order, auto_reset=True, out=out)
            else:       # the scripted policies / the device RNG evaluated inside the kernel
                oracle.OracleBatch.set_rng_seed(seed * 2654435761 + 7)
                env.set_rng_seed(seed * 2654435761 + 7)
                oracle.OracleBatch.set_policy_epsilon(eps)      # epsilon-greedy / -waiting (no effect on "random")
                env.set_policy_epsilon(eps)
                if mt and mode != "random":                      # the reference's own stream: a numpy RandomState per env
                    mt_seeds = (np.arange(E, dtype=np.uint32) * np.uint32(40503) + np.uint32(seed)) if seed & 2 else seed
                    ob.set_policy_stream_mt19937(mt_seeds, eps)
                    env.set_policy_stream("mt19937", mt_seeds)
                o_act, o_obs, o_rew, o_af, o_ef = ob.rollout_greedy(K, auto_reset=True, policy=mode)
                res, acts = env.rollout_greedy(K, auto_reset=True, policy=mode, out=out)
                np.testing.assert_array_equal(_np(acts), o_act)
                if mt and mode != "random":
                    np.testing.assert_array_equal(env.policy_stream_state(), ob._mt)
            np.testing.assert_array_equal(_np(res.agent_flags), o_af)
            np.testing.assert_array_equal(_np(res.env_flags), o_ef)
            got_obs = _np(env.expand_observations(res.obs_compact)) if compact else _np(res.obs)
            np.testing.assert_array_equal(got_obs.view(np.uint32), o_obs.view(np.uint32))
            np.testing.assert_array_equal(_np(res.reward).view(np.uint64), o_rew.view(np.uint64))
            assert env.counters() == ob.counters.as_dict()
        finally:
            oracle.OracleBatch.set_policy_epsilon(0.0)
            env.close()

    run()


@pytest.mark.parametrize("name,E,K,policy,eps", [
    ("g4_c5_all_at_dest_greedy_25_25", 96, 130, "greedy", 0.1), ("g4_c5_all_at_dest_greedy_32_32", 64, 100, "greedy", 0.1),
    ("g4_c5_all_at_dest_greedy_32_32", 33, 70, "waiting", 0.5), ("g8_rollout_c1", 700, 150, "greedy", 1.0),
    ("g9_c1_waiting_policy", 300, 120, "waiting", 0.1)])
def test_epsilon_greedy_rollouts_equal_the_oracle(oracle, ccx, name, E, K, policy, eps):
    """ccx_set_policy_epsilon: the reference's create_greedy_policy(epsilon=0.1) / create_waiting_policy shape
    of rollout with the draws on the device (counter-based, include/ccx.h) -- actions, observations, rewards,
    flags and counters bit-equal to the oracle's restatement, across auto-resets and two launches."""
    from collectivecrossing_amd.reset import build_reset_pool

    g = Golden(name)
    pool = build_reset_pool(g.config, 11, 200)
    ob, env = oracle.OracleBatch(g.params, E), ccx(g.config, E)
    try:
        for b in (ob, env):
            b.set_reset_pool(pool)
            b.reset_from_pool()
        oracle.OracleBatch.set_rng_seed(2024)
        oracle.OracleBatch.set_policy_epsilon(eps)
        env.set_rng_seed(2024)
        env.set_policy_epsilon(eps)
        first_acts = None
        for part in (K // 2, K - K // 2):
            o_act, o_obs, o_rew, o_af, o_ef = ob.rollout_greedy(part, auto_reset=True, policy=policy)
            res, acts = env.rollout_greedy(part, auto_reset=True, policy=policy)
            np.testing.assert_array_equal(_np(acts), o_act)
            np.testing.assert_array_equal(_np(res.agent_flags), o_af)
            np.testing.assert_array_equal(_np(res.env_flags), o_ef)
            np.testing.assert_array_equal(_np(res.obs).view(np.uint32), o_obs.view(np.uint32))
            np.testing.assert_array_equal(_np(res.reward).view(np.uint64), o_rew.view(np.uint64))
            first_acts = o_act if first_acts is None else first_acts
            del res
        assert env.counters() == ob.counters.as_dict()
        assert ob.counters.episodes > 0
        # the draws did change the rollout: the epsilon-0 policy takes other actions from the same start
        oracle.OracleBatch.set_policy_epsilon(0.0)
        ob0 = oracle.OracleBatch(g.params, E)
        ob0.set_reset_pool(pool)
        ob0.reset_from_pool()
        assert (ob0.rollout_greedy(K // 2, auto_reset=True, policy=policy)[0] != first_acts).any()
    finally:
        oracle.OracleBatch.set_policy_epsilon(0.0)
        env.close()


def test_policy_epsilon_argument_checks(ccx):
    from collectivecrossing_amd._lib import CcxError

    env = ccx(Golden("g8_rollout_c1").config, 4)
    for bad in (-0.1, 1.5, float("nan")):
        with pytest.raises(CcxError, match="epsilon must be in"):
            env.set_policy_epsilon(bad)
    env.set_policy_epsilon(1.0)
    env.set_policy_epsilon(0.0)
    env.close()


def test_consecutive_long_launches_with_the_adaptive_pace_equal_the_oracle(oracle, ccx):
    """Five launches of 160 steps in a row (the pace controller votes after each and the next launch
    reads the vote), 4096 envs so that the batch is paced at all: every launch bit-exact vs the oracle."""
    from collectivecrossing_amd.reset import build_reset_pool

    g = Golden("g8_rollout_c1")
    E, K = 4096, 160
    rng = np.random.default_rng(5)
    ob = oracle.OracleBatch(g.params, E)
    env = ccx(g.config, E)
    pool = build_reset_pool(g.config, 900, 300)
    for b in (ob, env):
        b.set_reset_pool(pool)
        b.reset_from_pool()
    paces = [env.step_pace_ns()]
    for launch in range(5):
        actions = rng.integers(0, 5, size=(K, E, g.N), dtype=np.uint8)
        o_obs, o_rew, o_af, o_ef = ob.rollout(actions, None, auto_reset=True)
        res = env.rollout(actions, auto_reset=True)
        np.testing.assert_array_equal(_np(res.agent_flags), o_af, err_msg=f"launch {launch}")
        np.testing.assert_array_equal(_np(res.env_flags), o_ef, err_msg=f"launch {launch}")
        np.testing.assert_array_equal(_np(res.obs).view(np.uint32), o_obs.view(np.uint32), err_msg=f"launch {launch}")
        np.testing.assert_array_equal(_np(res.reward).view(np.uint64), o_rew.view(np.uint64), err_msg=f"launch {launch}")
        paces.append(env.step_pace_ns())
        del res
    assert paces[0] > 0 and len(set(paces)) > 1, paces      # the controller did retune
    assert env.counters() == ob.counters.as_dict()
    env.close()


def test_a_hundred_thousand_envs_equal_the_oracle(oracle, ccx):
    """A batch far beyond one round of workgroups (100 003 envs: 12 501 tiles, a partial last tile and a
    partial last round), three auto-reset steps from the end of an episode, shuffled order."""
    _random_case(oracle, ccx, "g8_rollout_c1", E=100003, K=3, seed=77, shuffle=True, auto_reset=True, p_absent=0.05)


def test_full_size_c2_properties(ccx):
    """BASELINE config 2 at full size (4096 x 8): size-independent properties of a long
    auto-reset rollout -- no two active agents ever share a cell, nobody stands in a wall,
    step counters stay below max_steps, the counters add up."""
    import torch

    from collectivecrossing_amd.reset import build_reset_pool

    g = Golden("g1_c1_random")
    E, K, N = 4096, 400, 8
    env = ccx(g.config, E)
    env.set_reset_pool(build_reset_pool(g.config, 0, 1024))
    env.reset_from_pool()
    actions = torch.randint(0, 5, (K, E, N), dtype=torch.uint8, device=env.device,
                            generator=torch.Generator(device=env.device).manual_seed(3))
    res = env.rollout(actions, auto_reset=True)
    obs = res.obs  # [K,E,N,L]
    x, y = obs[..., 0].long(), obs[..., 1].long()
    af = res.agent_flags.long()
    active = (af & 0x40) != 0
    p = g.params
    assert bool(((x >= 0) & (x <= p.width) & (y >= 0) & (y <= p.height)).all())
    on_wall_row = (y == p.division_y) & ~((x > p.door_left) & (x < p.door_right))
    in_tram_rows = (y >= p.division_y) & ~((x > p.tram_left) & (x < p.tram_right))
    assert not bool((on_wall_row | in_tram_rows).any())
    key = torch.where(active, y * 128 + x, -1 - torch.arange(N, device=env.device).expand_as(x))
    srt = key.sort(dim=-1).values
    assert not bool((srt[..., 1:] == srt[..., :-1]).any()), "two active agents share a cell"
    # every observer row agrees with the others about everybody's position
    slots = obs[..., 6:].reshape(K, E, N, N, 4)
    j = torch.arange(N, device=env.device)
    others = slots[..., 0][:, :, (j + 1) % N, j]  # x of agent j as seen by observer j+1
    assert bool((others == obs[..., 0]).all())
    c = env.counters()
    assert c["env_steps"] == K * E and c["agent_steps"] == K * E * N
    assert c["episodes"] == int(((res.env_flags & 4) != 0).sum())
    assert c["live_agent_steps"] == int(((af & 4) != 0).sum())
    st = env.get_state()
    assert (st["step_count"] < p.max_steps).all() and (st["step_count"] >= 0).all()
    env.close()


def test_mid_size_grid_shrinks_tiles_to_keep_the_occupancy_tables(oracle, ccx):
    """40x40 grid, 8 agents: eight envs per wave would need 118 KiB of occupancy tables; the
    library carries fewer envs per wave instead of falling back, so policy rollouts still work."""
    from collectivecrossing_amd import configs as C
    from collectivecrossing_amd.params import lower_config
    from collectivecrossing_amd.reset import build_reset_pool

    cfg = C.CollectiveCrossingConfig(
        width=40, height=40, division_y=20, tram_door_left=12, tram_door_right=18, tram_length=30,
        num_boarding_agents=5, num_exiting_agents=3, exiting_destination_area_y=17,
        boarding_destination_area_y=23, truncated_config=C.MaxStepsTruncatedConfig(max_steps=30),
        terminated_config=C.AllAtDestinationTerminatedConfig())
    E, K = 300, 90
    pool = build_reset_pool(cfg, 5, 97)
    ob = oracle.OracleBatch(lower_config(cfg), E)
    env = ccx(cfg, E)
    assert env.launch_shape()["lanes_per_wave"] < 64
    for b in (ob, env):
        b.set_reset_pool(pool)
        b.reset_from_pool()
    o_act, o_obs, o_rew, o_af, o_ef = ob.rollout_greedy(K, auto_reset=True)
    res, acts = env.rollout_greedy(K, auto_reset=True)
    np.testing.assert_array_equal(_np(acts), o_act)
    np.testing.assert_array_equal(_np(res.agent_flags), o_af)
    np.testing.assert_array_equal(_np(res.env_flags), o_ef)
    np.testing.assert_array_equal(_np(res.obs).view(np.uint32), o_obs.view(np.uint32))
    assert env.counters() == ob.counters.as_dict() and ob.counters.arrivals > 0
    env.close()


def test_c4_sharding_is_bit_invariant_at_full_size(ccx):
    """BASELINE config 4 (32768 envs x 8 agents over 8 GPUs): eight shard handles
    (env_offset = r*4096, total_envs = 32768), run one after the other on this GPU, reproduce the
    single 32768-env batch bit for bit -- what makes results identical at 1/2/4/8 GPUs."""
    import torch

    g = Golden("g1_c1_random")
    total, world, K, N = 32768, 8, 130, 8
    per = total // world
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(11)
    actions = torch.randint(0, 5, (K, total, N), dtype=torch.uint8, device=dev, generator=gen)
    whole = ccx(g.config, total)
    whole.make_reset_pool(77, 5000)
    whole.reset_from_pool()
    ref = whole.rollout(actions, auto_reset=True)
    ref_counters = whole.counters()
    ref_state = whole.get_state()
    pool = whole.reset_pool()
    summed = dict.fromkeys(ref_counters, 0)
    for r in range(world):
        shard = ccx(g.config, per, env_offset=r * per, total_envs=total)
        shard.set_reset_pool(pool)
        shard.reset_from_pool()
        out = shard.rollout(actions[:, r * per:(r + 1) * per].contiguous(), auto_reset=True)
        sl = slice(r * per, (r + 1) * per)
        assert torch.equal(out.obs.view(torch.int32), ref.obs[:, sl].view(torch.int32))
        assert torch.equal(out.reward.view(torch.int64), ref.reward[:, sl].view(torch.int64))
        assert torch.equal(out.agent_flags, ref.agent_flags[:, sl])
        assert torch.equal(out.env_flags, ref.env_flags[:, sl])
        st = shard.get_state()
        for k in ("x", "y", "active", "terminated", "truncated", "step_count", "episode"):
            np.testing.assert_array_equal(st[k], ref_state[k][sl], err_msg=k)
        for k, v in shard.counters().items():
            summed[k] += v
        shard.close()
    assert summed == ref_counters and ref_counters["episodes"] >= total
    whole.close()


def test_c5_full_size_greedy_rollout_properties(ccx):
    """BASELINE config 5 per-GPU share (1024 envs x 64 agents, AllAtDestination, MaxSteps=500,
    greedy actions inside the kernel): 520 steps with auto-reset.  The door jams (the
    reference dead-locks and truncates at 500), so every env truncates exactly at step 500."""
    import torch

    from bench import workload_config

    cfg, E = workload_config("c5_64")
    env = ccx(cfg, E)
    env.make_reset_pool(0, 2048)
    env.reset_from_pool()
    K, N = 520, 64
    res, acts = env.rollout_greedy(K, auto_reset=True, want_obs=False)
    ef = res.env_flags
    af = res.agent_flags.long()
    assert bool(((ef[:499] & 3) == 0).all()), "nobody finishes before max_steps"
    assert bool(((ef[499] & 2) != 0).all()) and bool(((ef[499] & 4) != 0).all()), "all truncate + reset at step 500"
    assert bool(((af[499] & 2) != 0).all())
    assert bool((af[500] & 4).bool().all()), "fresh episode: everybody live again"
    assert bool((acts[:500] != 255).all()) and int(acts.max()) <= 4
    c = env.counters()
    assert c["env_steps"] == K * E and c["episodes"] == E and c["arrivals"] > 0 and c["moves"] > 0
    st = env.get_state()
    assert (st["step_count"] == 20).all() and (st["episode"] == 1).all()
    # no two active agents on one cell at the end
    key = np.where(st["active"] == 1, st["y"] * 128 + st["x"], -1 - np.arange(N)[None, :])
    srt = np.sort(key, axis=1)
    assert not (srt[:, 1:] == srt[:, :-1]).any()
    env.close()


def test_handle_can_be_moved_to_another_stream(ccx):
    import torch

    g = Golden("g7_n3_small")
    env = ccx(g.config, g.E)
    env.set_state(**g.init_state())
    side = torch.cuda.Stream(device=env.device)
    with torch.cuda.stream(side):
        env.use_stream()
        res = env.rollout(g["actions"], g["order"])
    side.synchronize()
    _check_rollout_vs_golden(g, res, env.get_state())
    env.close()


def test_policy_in_the_loop_stepping_captures_into_a_hip_graph(ccx):
    """A step-wise loop (policy kernel -> ``ccx_step``) is launch-bound; both calls only enqueue
    kernels on the handle's stream, so the pair captures into a HIP graph (torch.cuda.graph) that is
    replayed per step.  Result = the fused in-kernel policy rollout, bit for bit."""
    import torch

    g = Golden("g4_c1_individual_greedy")
    K = 40
    ref = ccx(g.config, g.E)
    ref.set_state(**g.init_state())
    want, want_actions = ref.rollout_greedy(K)
    env = ccx(g.config, g.E)
    env.set_state(**g.init_state())
    side = torch.cuda.Stream(device=env.device)
    env.use_stream(side)                       # bind BEFORE the capture starts (set_stream syncs)
    acts = torch.empty((g.E, g.N), dtype=torch.uint8, device=env.device)
    with torch.cuda.stream(side):
        env.greedy_actions(out=acts)           # warm-up outside the capture (allocates the step buffers)
        env.step(acts)
        side.synchronize()
        env.set_state(**g.init_state())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            env.greedy_actions(out=acts)
            out = env.step(acts)
        for s in range(K):
            graph.replay()
            side.synchronize()
            np.testing.assert_array_equal(_np(acts), _np(want_actions[s]), err_msg=f"actions step {s}")
            np.testing.assert_array_equal(_np(out.obs).view(np.uint32), _np(want.obs[s]).view(np.uint32))
            np.testing.assert_array_equal(_np(out.reward).view(np.uint64), _np(want.reward[s]).view(np.uint64))
            np.testing.assert_array_equal(_np(out.agent_flags), _np(want.agent_flags[s]))
            np.testing.assert_array_equal(_np(out.env_flags), _np(want.env_flags[s]))
    st, st_ref = env.get_state(), ref.get_state()
    for k in ("x", "y", "active", "terminated", "truncated", "step_count"):
        np.testing.assert_array_equal(st[k], st_ref[k], err_msg=k)
    env.close()
    ref.close()


def test_errors_are_loud(ccx):
    from collectivecrossing_amd._lib import CcxError

    g = Golden("g7_n3_small")
    env = ccx(g.config, 4)
    with pytest.raises(CcxError, match="outside 0..width"):
        env.set_state(x=np.full((4, 3), 99, np.int32))
    with pytest.raises(CcxError, match="reset pool"):
        env.rollout(np.zeros((2, 4, 3), np.uint8), auto_reset=True)
    with pytest.raises(ValueError):
        env.step(np.zeros((4, 2), np.uint8))
    env.close()


@pytest.mark.parametrize("policy,eps", [("greedy", 0.3), ("waiting", 0.5), ("greedy", 1.0)])
def test_stepwise_policy_loop_takes_the_actions_of_the_fused_rollout(oracle, ccx, policy, eps):
    """ccx_policy_actions honours ccx_set_policy_epsilon with the draws of the fused rollout: K rounds of
    policy_actions + step take exactly the actions (and produce the outputs) of one rollout_policy(K) from the same
    start, and equal the oracle's stand-alone restatement state by state; greedy_actions stays epsilon 0."""
    from collectivecrossing_amd.reset import build_reset_pool

    g = Golden("g4_c5_all_at_dest_greedy_25_25")
    E, K = 40, 45
    pool = build_reset_pool(g.config, 21, 90)
    fused, loop, ob = ccx(g.config, E), ccx(g.config, E), oracle.OracleBatch(g.params, E)
    try:
        for b in (fused, loop, ob):
            b.set_reset_pool(pool)
            b.reset_from_pool()
        oracle.OracleBatch.set_rng_seed(31)
        oracle.OracleBatch.set_policy_epsilon(eps)
        for b in (fused, loop):
            b.set_rng_seed(31)
            b.set_policy_epsilon(eps)
        res, acts = fused.rollout_policy(K, policy, auto_reset=False)
        differs = 0
        for s in range(K):
            a = loop.policy_actions(policy)
            np.testing.assert_array_equal(_np(a), ob.policy_actions(policy, with_epsilon=True), err_msg=f"step {s}")
            np.testing.assert_array_equal(_np(a), _np(acts[s]), err_msg=f"step {s}")
            differs += int((_np(loop.greedy_actions()) != _np(a)).sum()) if policy == "greedy" else 0
            out = loop.step(a)
            ob.step(_np(a))
            np.testing.assert_array_equal(_np(out.obs).view(np.uint32), _np(res.obs[s]).view(np.uint32), err_msg=f"step {s}")
            np.testing.assert_array_equal(_np(out.agent_flags), _np(res.agent_flags[s]))
        assert policy != "greedy" or differs > 0          # the draws did replace greedy choices
        assert loop.counters() == fused.counters() == ob.counters.as_dict()
    finally:
        oracle.OracleBatch.set_policy_epsilon(0.0)
        fused.close()
        loop.close()
