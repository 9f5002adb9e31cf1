"""Size-independent properties of the step, checked on the CPU oracle over random reference-valid
configs (hypothesis): they are the invariants the full-size GPU tests rely on where the oracle is too
slow to follow (tests/test_gpu_parity.py::test_full_size_c2_properties, C4, C5)."""

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from collectivecrossing_amd import configs as C
from collectivecrossing_amd.params import lower_config
from collectivecrossing_amd.reset import seeded_positions


@st.composite
def configs(draw, max_boarding=5, max_exiting=4):
    W, H = draw(st.integers(6, 24)), draw(st.integers(5, 16))
    div = draw(st.integers(2, H - 2))
    Lt = draw(st.integers(4, W))
    dl = draw(st.integers(0, Lt - 2))
    dr = draw(st.integers(dl + 1, Lt))
    # keep the rejection-sampled reset feasible with room to spare (the reference -- and the restated
    # host / device samplers -- would spin forever on a config without free cells)
    tl, tr = W // 2 - Lt // 2, W // 2 + Lt // 2
    cap_b = W * div - (dr - dl + 1)
    cap_e = max(0, tr - tl - 1) * max(0, H - div - 1)
    nb = draw(st.integers(0, max(0, min(max_boarding, cap_b // 3))))
    ne = draw(st.integers(0, max(0, min(max_exiting, cap_e // 3))))
    if nb + ne == 0:
        from hypothesis import reject
        reject()
    reward = draw(st.sampled_from([C.DefaultRewardConfig(), C.SimpleDistanceRewardConfig(distance_penalty_factor=0.3),
                                   C.BinaryRewardConfig(goal_reward=2.0, no_goal_reward=-0.5),
                                   C.ConstantNegativeRewardConfig(step_penalty=-0.25)]))
    term = draw(st.sampled_from([C.IndividualAtDestinationTerminatedConfig(), C.AllAtDestinationTerminatedConfig()]))
    try:
        return C.CollectiveCrossingConfig(
            width=W, height=H, division_y=div, tram_door_left=dl, tram_door_right=dr, tram_length=Lt,
            num_boarding_agents=nb, num_exiting_agents=ne, exiting_destination_area_y=draw(st.integers(0, div - 1)),
            boarding_destination_area_y=draw(st.integers(div, H)), reward_config=reward, terminated_config=term,
            truncated_config=C.MaxStepsTruncatedConfig(max_steps=draw(st.integers(3, 25))))
    except Exception:   # noqa: BLE001 -- our validators mirror the reference's: draw again
        from hypothesis import reject
        reject()


@settings(max_examples=40, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.too_slow, HealthCheck.filter_too_much])
@given(cfg=configs(), seed=st.integers(0, 2**31 - 1))
def test_step_invariants(oracle, cfg, seed):
    p = lower_config(cfg)
    N, E, K = p.num_boarding + p.num_exiting, 6, 30
    rng = np.random.default_rng(seed)
    pos = seeded_positions(cfg, range(seed % 1000, seed % 1000 + E))
    b = oracle.OracleBatch(p, E)
    b.set_state(x=pos[..., 0], y=pos[..., 1])
    was_done = np.zeros((E, N), bool)
    for s in range(K):
        before = np.stack([b.x.copy(), b.y.copy()], -1)
        active_before = b.active.copy().astype(bool)
        actions = rng.integers(0, 5, size=(E, N), dtype=np.uint8)
        obs, rew, af, ef = b.step(actions, None)
        x, y = b.x, b.y
        # positions stay on the grid (bounds are inclusive, collectivecrossing.py:509-534)
        assert (x >= 0).all() and (x <= p.width).all() and (y >= 0).all() and (y <= p.height).all()
        # one cell per step at most, and inactive agents never move
        moved = np.abs(x - before[..., 0]) + np.abs(y - before[..., 1])
        assert (moved <= 1).all() and (moved[~active_before] == 0).all()
        # two ACTIVE agents never share a cell
        for e in range(E):
            cells = [(int(x[e, i]), int(y[e, i])) for i in range(N) if b.active[e, i]]
            assert len(cells) == len(set(cells))
        # flags: done agents are not live, live agents get finite rewards, step counter advances
        live = (af & 4) != 0
        assert not (live & was_done).any() and np.isfinite(rew[live]).all() and (rew[~live] == 0).all()
        assert (b.step_count == s + 1).all()
        # observation rows: own position first, then the constants of the geometry
        assert (obs[..., 0] == x).all() and (obs[..., 1] == y).all()
        assert (obs[..., 3] == p.division_y).all() and (obs[..., 4] == p.door_left).all() and (obs[..., 5] == p.door_right).all()
        # done is sticky; truncation hits every live agent of an env at once, at max_steps
        was_done |= ((af & 1) != 0) & live | ((af & 2) != 0) & live
        assert (((af & 2) != 0) & live == (live & (s + 1 >= p.max_steps))).all()
    assert b.counters.env_steps == E * K


def test_two_identical_runs_are_identical(oracle):
    from _fixtures import Golden

    g = Golden("g1_c1_random")
    outs = []
    for _ in range(2):
        b = oracle.OracleBatch(g.params, g.E)
        b.set_state(**g.init_state())
        outs.append(b.rollout(g["actions"], g["order"]))
    for a, c in zip(*outs):
        np.testing.assert_array_equal(a, c)


def test_device_random_policy_stream_is_uniform_and_independent_of_splits(oracle):
    """CCX_POLICY_RANDOM (include/ccx.h: ccx_set_rng_seed): a counter-based hash of (seed, global env, episode,
    step of the episode, agent).  Restated in the oracle; properties: all five actions ~uniform, a different
    seed gives a different stream, and an env's stream does not depend on how the batch is cut into shards or
    the rollout into launches."""
    from _fixtures import Golden

    from collectivecrossing_amd.reset import build_reset_pool
    g = Golden("g8_rollout_c1")
    pool = build_reset_pool(g.config, 40, 61)

    def run(E, K_parts, off=0, total=24, seed=1234):
        oracle.OracleBatch.set_rng_seed(seed)
        b = oracle.OracleBatch(g.params, E, env_offset=off, total_envs=total)
        b.set_reset_pool(pool)
        b.reset_from_pool()
        outs = [b.rollout_greedy(k, auto_reset=True, policy="random") for k in K_parts]
        return [np.concatenate([o[j] for o in outs], axis=0) for j in range(5)], b

    (acts, obs, rew, af, ef), b = run(24, [120])
    live = acts != 255
    counts = np.bincount(acts[live], minlength=5)
    assert counts.min() > 0.17 * live.sum() and counts.max() < 0.23 * live.sum() and set(np.unique(acts)) <= {0, 1, 2, 3, 4, 255}
    assert b.counters.episodes > 24        # the stream goes on across auto-resets (episode is part of the key)
    (acts2, obs2, *_), _ = run(24, [50, 70])                      # two launches
    np.testing.assert_array_equal(acts, acts2)
    np.testing.assert_array_equal(obs.view(np.uint32), obs2.view(np.uint32))
    (a_lo, o_lo, *_), _ = run(10, [120], off=0)                   # two shards of the same global batch
    (a_hi, o_hi, *_), _ = run(14, [120], off=10)
    np.testing.assert_array_equal(np.concatenate([a_lo, a_hi], axis=1), acts)
    np.testing.assert_array_equal(np.concatenate([o_lo, o_hi], axis=1).view(np.uint32), obs.view(np.uint32))
    (acts3, *_), _ = run(24, [120], seed=1235)
    assert (acts3 != acts).mean() > 0.5


def _mix32(k):
    k &= 0xFFFFFFFF
    k ^= k >> 16
    k = (k * 0x7FEB352D) & 0xFFFFFFFF
    k ^= k >> 15
    k = (k * 0x846CA68B) & 0xFFFFFFFF
    k ^= k >> 16
    return k


def _eps_word(seed, g, j, t, a):
    """include/ccx.h (ccx_set_policy_epsilon): the exploration word of agent a of global env g at step t of episode j."""
    lo, hi = seed & 0xFFFFFFFF, (seed >> 32) ^ 0x5BD1E995
    k = (g * 0x9E3779B1 + j * 0x85EBCA77 + t * 0xC2B2AE3D + a * 0x27D4EB2F + lo) & 0xFFFFFFFF
    return _mix32(_mix32(k) ^ hi)


@pytest.mark.parametrize("policy", ["greedy", "waiting"])
def test_epsilon_greedy_draws_follow_the_documented_formula(oracle, policy):
    """ccx_set_policy_epsilon restated in the oracle: an agent explores iff its counter-based word is below
    epsilon * 2^32 (recomputed here in Python from the header's formula); otherwise it takes the epsilon-0
    policy's action; an exploring action is the k-th valid one, k from the second draw (valid = wait, or a
    target cell on the grid that no other active agent holds -- necessary conditions checked here); the
    stream does not depend on how the rollout is cut into launches or the batch into shards."""
    from _fixtures import Golden

    from collectivecrossing_amd.reset import build_reset_pool
    g = Golden("g4_c5_all_at_dest_greedy_25_25")
    pool = build_reset_pool(g.config, 7, 33)
    seed, eps, E, K, total = (77 << 32) | 12345, 0.3, 6, 60, 9
    thr = int(eps * 2**32)
    DX, DY = [1, 0, -1, 0, 0], [0, 1, 0, -1, 0]

    def fresh(E_, off):
        oracle.OracleBatch.set_rng_seed(seed)
        oracle.OracleBatch.set_policy_epsilon(eps)
        b = oracle.OracleBatch(g.params, E_, env_offset=off, total_envs=total)
        b.set_reset_pool(pool)
        b.reset_from_pool()
        return b

    try:
        b = fresh(E, 0)
        explored = asked = 0
        all_acts = []
        for s in range(K):
            base = b.policy_actions(policy)                      # epsilon = 0 on the same pre-step state
            x, y, active = b.x.copy(), b.y.copy(), b.active.copy()
            t, j = b.step_count.copy(), b.episode.copy()
            acts = b.rollout_greedy(1, auto_reset=True, policy=policy)[0][0]
            all_acts.append(acts)
            for e in range(E):
                for a in range(g.N):
                    if base[e, a] == 255:
                        assert acts[e, a] == 255
                        continue
                    asked += 1
                    u = _eps_word(seed, e, int(j[e]), int(t[e]), a)
                    if u >= thr:
                        assert acts[e, a] == base[e, a], (s, e, a)
                        continue
                    explored += 1
                    act = int(acts[e, a])
                    assert 0 <= act <= 4
                    if act != 4:
                        nx, ny = x[e, a] + DX[act], y[e, a] + DY[act]
                        assert 0 <= nx <= g.config.width and 0 <= ny <= g.config.height   # (:515: inclusive)
                        others = (np.arange(g.N) != a) & (active[e] != 0)
                        assert not np.any(others & (x[e] == nx) & (y[e] == ny))
        assert 0.25 * asked < explored < 0.35 * asked, (explored, asked)
        assert b.counters.episodes > 0
        whole = np.stack(all_acts)
        b2 = fresh(E, 0)                                           # one launch instead of K
        np.testing.assert_array_equal(b2.rollout_greedy(K, auto_reset=True, policy=policy)[0], whole)
        lo, hi = fresh(4, 0), None                                 # two shards of the same global batch
        a_lo = lo.rollout_greedy(K, auto_reset=True, policy=policy)[0]
        hi = fresh(2, 4)
        a_hi = hi.rollout_greedy(K, auto_reset=True, policy=policy)[0]
        np.testing.assert_array_equal(np.concatenate([a_lo, a_hi], axis=1), whole)
        oracle.OracleBatch.set_policy_epsilon(0.0)                 # epsilon 0 = the deterministic policy
        b3 = fresh(E, 0)
        oracle.OracleBatch.set_policy_epsilon(0.0)
        d = b3.rollout_greedy(5, auto_reset=True, policy=policy)[0]
        b4 = fresh(E, 0)
        oracle.OracleBatch.set_policy_epsilon(0.0)
        for s in range(5):
            np.testing.assert_array_equal(b4.policy_actions(policy), d[s])
            b4.rollout_greedy(1, auto_reset=True, policy=policy)
    finally:
        oracle.OracleBatch.set_policy_epsilon(0.0)


def test_epsilon_one_picks_uniformly_among_the_valid_actions(oracle):
    """epsilon = 1: every asked agent explores; a lone agent in the open (all five actions valid) draws each
    action about one time in five."""
    from _fixtures import Golden

    from collectivecrossing_amd.reset import build_reset_pool
    g = Golden("g7_n1_boarding_only")
    pool = build_reset_pool(g.config, 3, 50)
    try:
        oracle.OracleBatch.set_rng_seed(99)
        oracle.OracleBatch.set_policy_epsilon(1.0)
        b = oracle.OracleBatch(g.params, 400)
        b.set_reset_pool(pool)
        b.reset_from_pool()
        acts = b.rollout_greedy(30, auto_reset=True, policy="greedy")[0]
        live = acts != 255
        counts = np.bincount(acts[live], minlength=5)
        assert counts.min() > 0.1 * live.sum(), counts     # border cells have fewer valid actions: loose bounds
        assert counts[4] > 0.19 * live.sum(), counts        # wait is valid everywhere: at least its 1/5 share
    finally:
        oracle.OracleBatch.set_policy_epsilon(0.0)
