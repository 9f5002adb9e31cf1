"""The reference's own epsilon stream on the device (include/ccx.h: ccx_set_policy_stream,
CCX_EPS_STREAM_MT19937).  The reference's policies draw `random_state.random()` / `.choice(valid_actions)` from ONE
`np.random.RandomState(42)` per policy object, agents in env.agents order (greedy_policy.py:31,48-59,
waiting_policy.py:31,48-59, scripts/run_greedy_policy_demo.py:67-109).  With the MT19937 stream the device replays the
epsilon episodes the REFERENCE recorded (g11_epsilon_policy_*) action for action -- stand-alone policy kernel + step, and
the policy rollout -- leaves every env's generator where numpy's is, and equals the oracle's restatement on larger
seeded batches (per-env seeds, auto-reset, several twists of the generator state, 64 agents)."""

import numpy as np
import pytest
from _fixtures import ALL_NPZ, Golden, assert_step_matches

pytestmark = pytest.mark.gpu
EPSILON = [n for n in ALL_NPZ if n.startswith("g11_epsilon_policy_")]


@pytest.fixture(scope="module")
def ccx():
    import torch

    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    from collectivecrossing_amd.batched import BatchedCollectiveCrossing

    return BatchedCollectiveCrossing


def _np(t):
    return None if t is None else t.cpu().numpy()


def _numpy_state(rs):
    _, key, pos = rs.get_state()[:3]
    return np.concatenate([key.astype(np.uint32), np.array([pos], np.uint32)])


@pytest.mark.parametrize("name", EPSILON)
def test_reference_epsilon_episodes_replay_action_for_action(ccx, name):
    g = Golden(name)
    eps = float(g["epsilon"])
    policy = "waiting" if "_w_" in name else "greedy"
    # (a) policy kernel + step, one pair per recorded step
    env = ccx(g.config, g.E)
    env.set_state(**g.init_state())
    env.set_policy_epsilon(eps)
    env.set_policy_stream("mt19937", 42)
    explored = 0
    for s in range(g.K):
        a = env.policy_actions(policy)
        np.testing.assert_array_equal(_np(a), g["actions"][s], err_msg=f"{name} step {s}")
        explored += int((_np(env.greedy_actions()) != _np(a)).sum()) if policy == "greedy" else 1
        r = env.step(a)
        assert_step_matches(g, s, _np(r.obs), _np(r.reward), _np(r.agent_flags), _np(r.env_flags), env.get_state())
    assert explored > 0
    state_a = env.policy_stream_state()
    # (b) the policy rollout (policy and step as separate launches inside one call)
    env.set_state(**g.init_state())
    env.set_policy_stream("mt19937", 42)                      # re-seeds
    res, acts = env.rollout_greedy(g.K, policy=policy)
    np.testing.assert_array_equal(_np(acts), g["actions"])
    np.testing.assert_array_equal(_np(res.agent_flags), g["agent_flags"])
    np.testing.assert_array_equal(_np(res.obs).view(np.uint32), g["obs"].view(np.uint32))
    live = (g["agent_flags"] & 4) != 0
    np.testing.assert_array_equal(np.where(live, _np(res.reward), 0).view(np.uint64),
                                  np.where(live, g["reward"], 0).view(np.uint64))
    np.testing.assert_array_equal(env.policy_stream_state(), state_a)
    env.close()


@pytest.mark.parametrize("name,E,K,policy,eps,seeds", [
    ("g8_rollout_c1", 300, 230, "greedy", 0.3, "per_env"), ("g9_c1_waiting_policy", 130, 150, "waiting", 0.1, 42),
    ("g4_c5_all_at_dest_greedy_32_32", 20, 60, "greedy", 0.25, "per_env"), ("g7_n5_odd", 257, 200, "greedy", 1.0, 7),
    ("g7_n1_boarding_only", 65, 400, "greedy", 0.5, "per_env")])
def test_stream_rollouts_equal_the_oracle(oracle, ccx, name, E, K, policy, eps, seeds):
    """Seeded batches against the oracle's MT19937 restatement (itself pinned against numpy and the g11 episodes):
    actions, every output, counters and the generators' final state (key + position), across auto-resets (the stream
    runs on, as a policy object's does) and two calls."""
    from collectivecrossing_amd.reset import build_reset_pool

    g = Golden(name)
    pool = build_reset_pool(g.config, 5, 150)
    seeds = (np.arange(E, dtype=np.uint32) * np.uint32(2654435761) + np.uint32(17)) if seeds == "per_env" else seeds
    ob, env = oracle.OracleBatch(g.params, E), ccx(g.config, E)
    try:
        for b in (ob, env):
            b.set_reset_pool(pool)
            b.reset_from_pool()
        ob.set_policy_stream_mt19937(seeds, eps)
        env.set_policy_epsilon(eps)
        env.set_policy_stream("mt19937", seeds)
        asked = np.zeros(E, np.int64)
        for part in (K // 3, K - K // 3):
            o_act, o_obs, o_rew, o_af, o_ef = ob.rollout_greedy(part, auto_reset=True, policy=policy)
            res, acts = env.rollout_greedy(part, auto_reset=True, policy=policy)
            np.testing.assert_array_equal(_np(acts), o_act)
            np.testing.assert_array_equal(_np(res.agent_flags), o_af)
            np.testing.assert_array_equal(_np(res.env_flags), o_ef)
            np.testing.assert_array_equal(_np(res.obs).view(np.uint32), o_obs.view(np.uint32))
            np.testing.assert_array_equal(_np(res.reward).view(np.uint64), o_rew.view(np.uint64))
            asked += (o_act != 255).sum(axis=(0, 2))
            del res
        assert env.counters() == ob.counters.as_dict() and ob.counters.episodes > 0
        st = env.policy_stream_state()
        np.testing.assert_array_equal(st, ob._mt)
        assert (st[:, 624] <= 624).all()
        # the generators really twisted several times: more words were drawn than one state holds
        assert asked.max() * 2 > 624
        # a stand-alone policy call continues the same stream
        np.testing.assert_array_equal(_np(env.policy_actions(policy)), ob.policy_actions(policy, with_epsilon=True))
        np.testing.assert_array_equal(env.policy_stream_state(), ob._mt)
    finally:
        ob.set_policy_stream_mt19937(None, 0.0)
        ob._bind_stream()
        env.close()


def test_generator_state_is_numpys(ccx):
    """Seeding = RandomState(seed) (key and pos as get_state() has them); epsilon 1 with every agent alone on an open
    grid: each decision draws one double and one 5-way choice, which numpy replays call for call."""
    from collectivecrossing_amd import configs as C

    cfg = C.CollectiveCrossingConfig(width=12, height=8, division_y=4, tram_door_left=5, tram_door_right=7, tram_length=9,
                                     num_boarding_agents=1, num_exiting_agents=0, exiting_destination_area_y=0,
                                     boarding_destination_area_y=8, truncated_config=C.MaxStepsTruncatedConfig(max_steps=1000))
    E = 5
    seeds = np.array([42, 0, 1, 2**32 - 1, 123456789], np.uint32)
    env = ccx(cfg, E)
    env.set_policy_stream("mt19937", seeds)
    st = env.policy_stream_state()
    for e in range(E):
        np.testing.assert_array_equal(st[e], _numpy_state(np.random.RandomState(int(seeds[e]))))
    env.set_policy_epsilon(1.0)
    rss = [np.random.RandomState(int(s)) for s in seeds]
    # park the agent in the middle of the waiting area: all four moves are valid, the list is [0, 1, 2, 3, 4]
    x = np.full((E, 1), 3, np.int32)
    y = np.full((E, 1), 1, np.int32)
    for k in range(700):                                          # 2100 words per env: three twists
        env.set_state(x=x, y=y, active=np.ones((E, 1), np.uint8), terminated=np.zeros((E, 1), np.uint8),
                      truncated=np.zeros((E, 1), np.uint8), step_count=np.zeros(E, np.int32))
        a = _np(env.policy_actions("greedy"))
        want = []
        for rs in rss:
            assert rs.random() < 1.0
            want.append(rs.choice([0, 1, 2, 3, 4]))
        assert a[:, 0].tolist() == want, k
    st = env.policy_stream_state()
    for e in range(E):
        np.testing.assert_array_equal(st[e], _numpy_state(rss[e]))
    # epsilon 0 draws nothing
    env.set_policy_epsilon(0.0)
    np.testing.assert_array_equal(_np(env.policy_actions("greedy")), _np(env.greedy_actions()))
    np.testing.assert_array_equal(env.policy_stream_state(), st)
    # back to the counter-based draws: the generators are left alone
    env.set_policy_epsilon(0.5)
    env.set_policy_stream("counter")
    env.policy_actions("greedy")
    env.rollout_greedy(3)
    np.testing.assert_array_equal(env.policy_stream_state(), st)
    env.close()


def test_stream_argument_checks(ccx):
    from collectivecrossing_amd._lib import CcxError

    env = ccx(Golden("g8_rollout_c1").config, 4)
    with pytest.raises(ValueError, match="unknown epsilon stream"):
        env.set_policy_stream("pcg64")
    with pytest.raises(CcxError, match="unknown epsilon stream"):
        from collectivecrossing_amd._lib import check
        check(env._lib.ccx_set_policy_stream(env._h, 7, None, 0))
    with pytest.raises(CcxError, match="no MT19937 stream"):
        env.policy_stream_state()
    env.close()


def test_stream_rollout_on_a_grid_without_occupancy_tables(oracle, ccx):
    """A 100 x 100 grid has no LDS occupancy tables; the stepwise loop of the MT19937 stream runs the stand-alone policy
    kernel + ccx_step and works on any shape -- against the oracle, with the all-pairs conflict path under it."""
    from collectivecrossing_amd import configs as C
    from collectivecrossing_amd._lib import CcxError
    from collectivecrossing_amd.params import lower_config
    from collectivecrossing_amd.reset import seeded_positions

    cfg = C.CollectiveCrossingConfig(
        width=100, height=100, division_y=50, tram_door_left=40, tram_door_right=60, tram_length=100,
        num_boarding_agents=12, num_exiting_agents=9, exiting_destination_area_y=48,
        boarding_destination_area_y=52, truncated_config=C.MaxStepsTruncatedConfig(max_steps=40))
    E, K, N = 19, 45, 21
    pos = seeded_positions(cfg, range(E))
    pos[:, :, 0] = 45 + (np.arange(N) % 7)[None, :]
    pos[:, :12, 1] = 47 + (np.arange(12) // 7)[None, :]
    pos[:, 12:, 1] = 51 + (np.arange(9) // 7)[None, :]
    ob, env = oracle.OracleBatch(lower_config(cfg), E), ccx(cfg, E)
    try:
        ob.set_state(x=pos[..., 0], y=pos[..., 1])
        env.set_state(x=pos[..., 0], y=pos[..., 1])
        env.set_policy_epsilon(0.3)
        # (round 4: the fused kernel no longer needs the occupancy tables for its policies -- tests/test_gpu_large_grid_policy.py;
        # this test keeps to the stepwise MT19937 loop on such a grid)
        ob.set_policy_stream_mt19937(42, 0.3)
        env.set_policy_stream("mt19937", 42)
        o_act, o_obs, o_rew, o_af, o_ef = ob.rollout_greedy(K, policy="greedy")
        res, acts = env.rollout_greedy(K, policy="greedy")
        np.testing.assert_array_equal(_np(acts), o_act)
        np.testing.assert_array_equal(_np(res.agent_flags), o_af)
        np.testing.assert_array_equal(_np(res.obs).view(np.uint32), o_obs.view(np.uint32))
        np.testing.assert_array_equal(_np(res.reward).view(np.uint64), o_rew.view(np.uint64))
        np.testing.assert_array_equal(env.policy_stream_state(), ob._mt)
        assert env.counters() == ob.counters.as_dict() and env.counters()["moves"] > 0
    finally:
        ob.set_policy_stream_mt19937(None, 0.0)
        ob._bind_stream()
        env.close()
