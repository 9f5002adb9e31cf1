"""Position-only user strategies on the GPU batch path (ccx_set_reward_table / ccx_set_terminated_table): the g13 episodes
the imported reference recorded with the plugin classes registered replay bit for bit through ccx_step (rollout kernel with a
move order, short-launch kernel without) and through the fused rollout; large batches equal the oracle with the same tables."""

import sys
from pathlib import Path

import numpy as np
import pytest
from _fixtures import PLUGIN_NPZ, Golden, assert_step_matches

sys.path.insert(0, str(Path(__file__).resolve().parent / "golden"))
import custom_strategies as cs  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ccx():
    import torch

    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    from collectivecrossing_amd import strategies
    from collectivecrossing_amd.batched import BatchedCollectiveCrossing
    plugins = cs.make_position_only(strategies.RewardFunction, strategies.TerminatedFunction)
    strategies.REWARD_FUNCTIONS[cs.PO_NAMES["reward"]] = plugins["reward"]
    strategies.TERMINATED_FUNCTIONS[cs.PO_NAMES["terminated"]] = plugins["terminated"]
    yield BatchedCollectiveCrossing
    strategies.REWARD_FUNCTIONS.pop(cs.PO_NAMES["reward"], None)
    strategies.TERMINATED_FUNCTIONS.pop(cs.PO_NAMES["terminated"], None)


def _np(t):
    return None if t is None else t.cpu().numpy()


@pytest.mark.parametrize("name", PLUGIN_NPZ)
@pytest.mark.parametrize("with_order", [True, False])
def test_recorded_plugin_episodes_replay_step_by_step(ccx, name, with_order):
    g = Golden(name)
    identity = bool((g["order"] == np.arange(g.N, dtype=np.uint8)).all())
    if not with_order and not identity:
        pytest.skip("recorded with shuffled dict orders")
    env = ccx(g.config, g.E)                                   # (the tables are installed by the constructor)
    env.set_state(**g.init_state())
    for s in range(g.K):
        r = env.step(g["actions"][s], g["order"][s] if with_order else None)
        assert_step_matches(g, s, _np(r.obs), _np(r.reward), _np(r.agent_flags), _np(r.env_flags), env.get_state())
    env.close()


@pytest.mark.parametrize("name", PLUGIN_NPZ)
def test_recorded_plugin_episodes_replay_through_the_fused_rollout(ccx, name):
    g = Golden(name)
    env = ccx(g.config, g.E)
    env.set_state(**g.init_state())
    res = env.rollout(g["actions"], g["order"])
    af, ef, rew, obs = _np(res.agent_flags), _np(res.env_flags), _np(res.reward), _np(res.obs)
    np.testing.assert_array_equal(af, g["agent_flags"])
    np.testing.assert_array_equal(ef & 3, g["env_flags"] & 3)
    live = (g["agent_flags"] & 4) != 0
    np.testing.assert_array_equal(np.where(live, rew, 0).view(np.uint64), np.where(live, g["reward"], 0).view(np.uint64))
    np.testing.assert_array_equal(obs.view(np.uint32), g["obs"].view(np.uint32))
    env.close()


@pytest.mark.parametrize("name,E,K,want_obs", [("g13_position_only_both", 4096, 40, True), ("g13_position_only_both", 4096, 40, False),
                                               ("g13_position_only_both", 333, 7, True), ("g13_position_only_terminated_c3", 500, 30, True),
                                               ("g13_position_only_reward", 9000, 20, False)])
def test_large_batches_equal_the_oracle_with_the_same_tables(oracle, ccx, name, E, K, want_obs):
    from collectivecrossing_amd.params import position_only_tables
    from collectivecrossing_amd.reset import build_reset_pool
    g = Golden(name)
    rew, term = position_only_tables(g.config)
    rng = np.random.default_rng(E + K)
    actions = rng.integers(0, 5, size=(K, E, g.N), dtype=np.uint8)
    pool = build_reset_pool(g.config, 4242, 97)
    ob = oracle.OracleBatch(g.params, E)
    ob.set_user_tables(reward=rew, terminated=term)
    env = ccx(g.config, E)
    for b in (ob, env):
        b.set_reset_pool(pool)
        b.reset_from_pool()
    for launch in range(2):
        res = env.rollout(actions, auto_reset=True, want_obs=want_obs)
        o_obs, o_rew, o_af, o_ef = ob.rollout(actions, auto_reset=True)
        np.testing.assert_array_equal(_np(res.agent_flags), o_af)
        np.testing.assert_array_equal(_np(res.env_flags), o_ef)
        np.testing.assert_array_equal(_np(res.reward).view(np.uint64), o_rew.view(np.uint64))
        if want_obs:
            np.testing.assert_array_equal(_np(res.obs).view(np.uint32), o_obs.view(np.uint32))
    st = env.get_state()
    for k in ("x", "y", "active", "terminated", "truncated", "step_count", "episode"):
        np.testing.assert_array_equal(st[k], getattr(ob, k), err_msg=k)
    assert env.counters() == ob.counters.as_dict()
    # back to the built-in strategies on the same handle
    env.set_reward_table(None, None)
    env.set_terminated_table(None, None)
    ob.set_user_tables(None, None)
    res = env.rollout(actions[:5], auto_reset=True)
    o_obs, o_rew, o_af, o_ef = ob.rollout(actions[:5], auto_reset=True)
    np.testing.assert_array_equal(_np(res.agent_flags), o_af)
    np.testing.assert_array_equal(_np(res.reward).view(np.uint64), o_rew.view(np.uint64))
    env.close()


def test_tables_that_do_not_fit_the_lds_are_refused(ccx):
    from collectivecrossing_amd import configs as C
    from collectivecrossing_amd._lib import CcxError
    cfg = C.CollectiveCrossingConfig(width=100, height=100, division_y=50, tram_door_left=10, tram_door_right=30, tram_length=60,
                                     num_boarding_agents=3, num_exiting_agents=3, exiting_destination_area_y=0,
                                     boarding_destination_area_y=100)
    env = ccx(cfg, 4)
    t = np.zeros((101, 101))
    with pytest.raises(CcxError, match="does not fit"):
        env.set_reward_table(t, t)
    env.close()
