"""Position-only user strategies: a registered RewardFunction / TerminatedFunction class whose value
depends on the agent's own type and cell only is lowered to a table (params.position_only_tables) that the kernels read like
the built-in strategies (ccx_set_reward_table / ccx_set_terminated_table).  CPU part: the lowering itself, its refusals, and
the ORACLE with those tables against the g13 fixtures -- episodes the imported reference recorded with the very classes
registered (tests/golden/custom_strategies.py: make_position_only; gen_golden.py: run_position_only)."""

import sys
from pathlib import Path

import numpy as np
import pytest
from _fixtures import PLUGIN_NPZ, Golden

sys.path.insert(0, str(Path(__file__).resolve().parent / "golden"))
import custom_strategies as cs  # noqa: E402

from collectivecrossing_amd import strategies  # noqa: E402
from collectivecrossing_amd.params import lower_config, position_only_tables  # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def registered():
    plugins = cs.make_position_only(strategies.RewardFunction, strategies.TerminatedFunction)
    strategies.REWARD_FUNCTIONS[cs.PO_NAMES["reward"]] = plugins["reward"]
    strategies.TERMINATED_FUNCTIONS[cs.PO_NAMES["terminated"]] = plugins["terminated"]
    yield plugins
    strategies.REWARD_FUNCTIONS.pop(cs.PO_NAMES["reward"], None)
    strategies.TERMINATED_FUNCTIONS.pop(cs.PO_NAMES["terminated"], None)


def test_three_fixtures_were_recorded_from_the_reference():
    assert PLUGIN_NPZ == ["g13_position_only_both", "g13_position_only_reward", "g13_position_only_terminated_c3"]


def test_tables_are_the_classes_values_cell_by_cell():
    g = Golden("g13_position_only_both")
    rew, term = position_only_tables(g.config)
    W, H = g.config.width, g.config.height
    assert rew[0].shape == rew[1].shape == (H + 1, W + 1) and rew[0].dtype == np.float64 and term[0].dtype == np.uint8
    p = lower_config(g.config, allow_position_only=True)
    in_tram = lambda x, y: y >= p.division_y and p.tram_left <= x <= p.tram_right   # noqa: E731  (collectivecrossing.py:551-554)
    for y in range(H + 1):
        for x in range(W + 1):
            v = 0.125 * x - 0.3 * y + (2.5 if in_tram(x, y) else 0.0)
            assert rew[0][y, x] == v + (7.0 if y == p.boarding_dest_y else 0.0)
            assert rew[1][y, x] == -(v + (7.0 if y == p.exiting_dest_y else 0.0))
            assert term[0][y, x] == in_tram(x, y) and term[1][y, x] == (y == p.exiting_dest_y)
    only_reward = Golden("g13_position_only_reward")
    r2, t2 = position_only_tables(only_reward.config)
    assert t2 is None and np.array_equal(r2[0], rew[0])
    assert position_only_tables(Golden("g1_c1_random").config) == (None, None)


def test_classes_that_are_not_position_only_keep_raising(registered):
    g = Golden("g13_position_only_both")
    with pytest.raises(ValueError, match="position_only"):
        lower_config(g.config)                                          # the plain lowering (oracle tests, E = 1 env) refuses

    class StepCountReward(strategies.RewardFunction):
        position_only = True                                            # ... but it is not

        def calculate_reward(self, agent_id, env):
            a = env._agents[agent_id]
            return None if (a.terminated or a.truncated) else -0.1 * env._step_count

    class PaysTheDead(strategies.RewardFunction):
        position_only = True

        def calculate_reward(self, agent_id, env):
            return 1.0

    class Undeclared(strategies.RewardFunction):
        def calculate_reward(self, agent_id, env):
            return 0.0

    from collectivecrossing_amd import configs as C
    for name, cls, msg in (("step_count", StepCountReward, "changed with"), ("pays_dead", PaysTheDead, "terminated agent")):
        strategies.REWARD_FUNCTIONS[name] = cls
        try:
            cfg = g.config.model_copy(update={"reward_config": C.CustomRewardConfig(reward_function=name)})
            with pytest.raises(ValueError, match=msg):
                position_only_tables(cfg)
        finally:
            del strategies.REWARD_FUNCTIONS[name]
    strategies.REWARD_FUNCTIONS["undeclared"] = Undeclared
    try:
        cfg = g.config.model_copy(update={"reward_config": C.CustomRewardConfig(reward_function="undeclared")})
        with pytest.raises(ValueError, match="E = 1"):
            lower_config(cfg, allow_position_only=True)
    finally:
        del strategies.REWARD_FUNCTIONS["undeclared"]
    with pytest.raises(ValueError, match="Unknown reward function"):
        lower_config(g.config.model_copy(update={"reward_config": C.CustomRewardConfig(reward_function="nobody_registered_this")}),
                     allow_position_only=True)


@pytest.mark.parametrize("name", PLUGIN_NPZ)
def test_oracle_with_tables_replays_what_the_reference_recorded(oracle, name):
    """The oracle's restatement of the table semantics against episodes the REFERENCE produced with the classes themselves."""
    from _fixtures import assert_step_matches
    g = Golden(name)
    rew, term = position_only_tables(g.config)
    ob = oracle.OracleBatch(lower_config(g.config, allow_position_only=True), g.E)
    ob.set_user_tables(reward=rew, terminated=term)
    ob.set_state(**g.init_state())
    for s in range(g.K):
        obs, reward, af, ef = ob.rollout(g["actions"][s][None], g["order"][s][None])
        st = {k: getattr(ob, k) for k in ("x", "y", "active", "terminated", "truncated", "step_count")}
        assert_step_matches(g, s, obs[0], reward[0], af[0], ef[0], st)
