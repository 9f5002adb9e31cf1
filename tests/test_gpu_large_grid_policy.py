"""Fused policy rollouts on grids whose occupancy tables exceed the LDS.  GreedyPolicy.get_action works
on every legal config (greedy_policy.py:33-60, 238-264; configs.py:39-40 allows 100 x 100); until round 3
ccx_rollout_policy refused such grids (CCX_EINVAL) and only the step-wise loop served them.  The in-kernel policies now take
their `busy` bits from an all-pairs exchange.  Parity: two 100 x 100 episodes of the reference's own GreedyPolicy and one of
its WaitingPolicy (g14_*, recorded from the imported reference) replay ACTION FOR ACTION through the fused rollout; larger
batches with auto-reset and epsilon draws equal the oracle."""

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


@pytest.mark.parametrize("name,policy", [("g14_greedy_100x100", "greedy"), ("g14_waiting_100x100", "waiting")])
def test_reference_policy_episodes_on_the_largest_grid_replay_through_the_fused_rollout(ccx, name, policy):
    g = Golden(name)
    env = ccx(g.config, g.E)
    assert env.step_shape()["ok"] == 0, "100 x 100: the occupancy tables do not fit the LDS, this is the all-pairs path"
    env.set_state(**g.init_state())
    res, acts = env.rollout_greedy(g.K, policy=policy)
    np.testing.assert_array_equal(_np(acts), g["actions"])
    np.testing.assert_array_equal(_np(res.agent_flags), g["agent_flags"])
    np.testing.assert_array_equal(_np(res.obs).view(np.uint32), g["obs"].view(np.uint32))
    live = (g["agent_flags"] & 4) != 0
    np.testing.assert_array_equal(np.where(live, _np(res.reward), 0).view(np.uint64), np.where(live, g["reward"], 0).view(np.uint64))
    st = env.get_state()
    for k in ("x", "y", "active", "terminated", "truncated"):
        np.testing.assert_array_equal(st[k], g[k][-1], err_msg=k)
    env.close()


@pytest.mark.parametrize("policy,eps,E,K", [("greedy", 0.0, 300, 120), ("waiting", 0.0, 64, 150), ("greedy", 0.2, 129, 90)])
def test_large_grid_policy_rollouts_equal_the_oracle(oracle, ccx, policy, eps, E, K):
    from collectivecrossing_amd.reset import build_reset_pool
    g = Golden("g14_greedy_100x100")
    cfg = g.config.model_copy(update={"truncated_config": g.config.truncated_config.model_copy(update={"max_steps": 70})})
    from collectivecrossing_amd.params import lower_config
    pool = build_reset_pool(cfg, 77, 53)
    ob = oracle.OracleBatch(lower_config(cfg), E)
    env = ccx(cfg, E)
    for b in (ob, env):
        b.set_reset_pool(pool)
        b.reset_from_pool()
        if eps:
            b.set_rng_seed(99)
            b.set_policy_epsilon(eps)
    try:
        o_act, o_obs, o_rew, o_af, o_ef = ob.rollout_greedy(K, auto_reset=True, policy=policy)
    finally:
        oracle.OracleBatch.set_policy_epsilon(0.0)             # (process-wide in the oracle)
    res, acts = env.rollout_greedy(K, auto_reset=True, policy=policy)
    np.testing.assert_array_equal(_np(acts), o_act)
    np.testing.assert_array_equal(_np(res.agent_flags), o_af)
    np.testing.assert_array_equal(_np(res.env_flags), o_ef)
    np.testing.assert_array_equal(_np(res.reward).view(np.uint64), o_rew.view(np.uint64))
    np.testing.assert_array_equal(_np(res.obs).view(np.uint32), o_obs.view(np.uint32))
    assert env.counters() == ob.counters.as_dict() and ob.counters.moves > 0
    env.close()
