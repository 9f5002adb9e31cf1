"""The scenarios of the reference's own unit tests (tests/collectivecrossing/envs/*.py), restated
against the GPU-backed drop-in class.  Each test names the reference test(s) it restates; the
assertions are the reference's (key presence, types, constants, error strings), the code is ours."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Env():
    from collectivecrossing_amd import CollectiveCrossingEnv

    return CollectiveCrossingEnv


def _config(**kw):
    """The 10x8 two-agent config of test_rewards.py:17-32 / test_terminateds.py:12-26."""
    from collectivecrossing_amd.configs import CollectiveCrossingConfig, MaxStepsTruncatedConfig

    d = dict(width=10, height=8, division_y=4, tram_door_left=4, tram_door_right=5, tram_length=8,
             num_boarding_agents=1, num_exiting_agents=1, exiting_destination_area_y=1,
             boarding_destination_area_y=6, truncated_config=MaxStepsTruncatedConfig(max_steps=100))
    d.update(kw)
    return CollectiveCrossingConfig(**d)


def _sampled_actions(env, obs):
    """`env.action_spaces[a].sample()` for every active agent in the observation dict."""
    return {a: int(env.action_spaces[a].sample()) for a in obs if env._agents[a].active}


# ---------------------------------------------------------------- test_collective_crossing.py
def test_initialization_and_reset_counts(Env):
    """test_environment_initialization :11-38, test_environment_reset :41-72."""
    cfg = _config(tram_door_left=3, tram_door_right=4, num_boarding_agents=3, num_exiting_agents=2,
                  exiting_destination_area_y=0, boarding_destination_area_y=7, render_mode="human")
    env = Env(config=cfg)
    c = env.config
    assert (c.width, c.height, c.division_y, c.tram_door_left, c.tram_door_right) == (10, 8, 4, 3, 4)
    assert (c.num_boarding_agents, c.num_exiting_agents) == (3, 2)
    assert (c.exiting_destination_area_y, c.boarding_destination_area_y) == (0, 7)
    obs, infos = env.reset(seed=42)
    assert len(obs) == 5 and len(infos) == 5
    assert len([k for k in obs if k.startswith("boarding")]) == 3
    assert len([k for k in obs if k.startswith("exiting")]) == 2
    env.close()


def test_waiting_keeps_everybody_in_place(Env):
    """test_agent_movement :75-113: all agents wait, positions (obs[:2]) unchanged, 4 observations."""
    env = Env(config=_config(tram_door_left=3, tram_door_right=4, num_boarding_agents=3,
                             exiting_destination_area_y=0, boarding_destination_area_y=7))
    obs, _ = env.reset(seed=42)
    before = {a: o[:2].copy() for a, o in obs.items()}
    new_obs, _, _, _, _ = env.step(dict.fromkeys(obs, 4))
    assert len(new_obs) == 4
    for a, o in new_obs.items():
        assert np.array_equal(o[:2], before[a])
    env.close()


def test_agents_on_their_destination_rows_terminate(Env):
    """test_agent_termination :116-152: forced positions via Agent.update_position."""
    env = Env(config=_config(width=8, height=6, division_y=3, tram_door_left=3, tram_door_right=4,
                             exiting_destination_area_y=0, boarding_destination_area_y=5))
    obs, _ = env.reset(seed=42)
    for a in obs:
        env._agents[a].update_position(np.array([4, 5]) if a.startswith("boarding") else np.array([4, 0]))
    _, _, terminated, _, _ = env.step(dict.fromkeys(obs, 4))
    assert all(v for k, v in terminated.items() if k != "__all__") and terminated["__all__"]
    env.close()


def test_action_space_and_observation_function_wiring(Env):
    """test_action_space :207-235 (5 discrete actions, invalid ones raise), test_observation_config
    :238-257, test_default_observation_function :260-277, test_observation_function_integration."""
    from collectivecrossing_amd.configs import DefaultObservationConfig
    from collectivecrossing_amd.observations import DefaultObservationFunction

    env = Env(config=_config(observation_config=DefaultObservationConfig()))
    assert isinstance(env._observation_function, DefaultObservationFunction)
    assert env.config.observation_config.get_observation_function_name() == "default"
    obs, _ = env.reset(seed=42)
    for a in obs:
        assert env.action_spaces[a].n == 5 and env.get_action_space(a).n == 5
        assert all(0 <= env.action_spaces[a].sample() < 5 for _ in range(20))
    for good in range(5):
        env.step({"boarding_0": good})
    with pytest.raises(ValueError):
        env.step({"boarding_0": 5})
    np.testing.assert_array_equal(env._get_agent_observation("boarding_0"), env.step({})[0]["boarding_0"])
    env.close()


def test_observation_layout_and_value_ranges(Env):
    """test_observation_structure :280-352 and test_observation_consistency :355-393."""
    env = Env(config=_config(height=6, division_y=3, tram_door_left=3, tram_door_right=4,
                             num_boarding_agents=2, exiting_destination_area_y=0,
                             boarding_destination_area_y=4))
    obs, _ = env.reset(seed=42)
    assert len(obs) == 3
    for o in obs.values():
        assert isinstance(o, np.ndarray) and o.dtype == np.float32 and o.shape == (2 + 4 + 4 * 3,)
        assert 0 <= o[0] < env.config.width and 0 <= o[1] < env.config.height
        assert o[2] == (env.tram_door_left + env.tram_door_right) // 2 and o[3] == env.config.division_y
        assert o[4] == env.tram_door_left and o[5] == env.tram_door_right
        others = o[6:].reshape(3, 4)
        assert set(others[:, 2].tolist()) <= {-1.0, 0.0, 1.0} and set(others[:, 3].tolist()) <= {-1.0, 0.0, 1.0}
        assert (others == -1.0).all(axis=1).sum() == 1            # exactly one self placeholder
    new_obs, _, _, _, _ = env.step(dict.fromkeys(obs, 4))
    for a in obs:
        assert obs[a].shape == new_obs[a].shape and obs[a].dtype == new_obs[a].dtype
        assert np.array_equal(obs[a][2:6], new_obs[a][2:6])
    env.close()


# ---------------------------------------------------------------------------- test_rewards.py
@pytest.mark.parametrize("make,check", [
    ("default", lambda r: isinstance(r, float)),
    ("simple_distance", lambda r: isinstance(r, float) and r <= 0),
    ("binary", lambda r: r == 0.0),
    ("constant_negative_2", lambda r: r == -2.0),
    ("constant_negative_default", lambda r: r == -1.0),
    ("custom_default", lambda r: isinstance(r, float)),
])
def test_every_reward_function_pays_both_agents(Env, make, check):
    """test_default_reward_function :34, test_simple_distance_reward_function :57,
    test_binary_reward_function :78 (always no_goal_reward), test_constant_negative_reward_function
    :99 / _default :120, test_custom_default_reward_config :181."""
    from collectivecrossing_amd import configs as C

    rc = {"default": C.DefaultRewardConfig(),
          "simple_distance": C.SimpleDistanceRewardConfig(distance_penalty_factor=0.2),
          "binary": C.BinaryRewardConfig(goal_reward=5.0, no_goal_reward=0.0),
          "constant_negative_2": C.ConstantNegativeRewardConfig(step_penalty=-2.0),
          "constant_negative_default": C.ConstantNegativeRewardConfig(),
          "custom_default": C.DefaultRewardConfig(boarding_destination_reward=50.0, tram_door_reward=25.0,
                                                  tram_area_reward=10.0, distance_penalty_factor=0.05)}[make]
    env = Env(config=_config(reward_config=rc))
    obs, _ = env.reset(seed=42)
    _, rewards, _, _, _ = env.step(_sampled_actions(env, obs))
    assert set(rewards) == {"boarding_0", "exiting_0"}
    assert all(check(r) for r in rewards.values()), rewards
    env.close()


def test_constant_negative_reward_is_the_same_every_step(Env):
    """test_constant_negative_reward_consistency :139-160."""
    from collectivecrossing_amd.configs import ConstantNegativeRewardConfig

    env = Env(config=_config(reward_config=ConstantNegativeRewardConfig(step_penalty=-3.0)))
    obs, _ = env.reset(seed=42)
    for _ in range(3):
        obs, rewards, _, _, _ = env.step(_sampled_actions(env, obs))
        assert rewards["boarding_0"] == -3.0 and rewards["exiting_0"] == -3.0
    env.close()


def test_destination_positions(Env):
    """test_get_agent_destination_position :163-178."""
    env = Env(config=_config())
    env.reset(seed=42)
    assert env.get_agent_destination_position("boarding_0") == (None, 6)
    assert env.get_agent_destination_position("exiting_0") == (None, 1)
    env.close()


def _park_on_destination(env, *agent_ids):
    for a in agent_ids:
        env._agents[a].position = np.array([5, env.get_agent_destination_position(a)[1]], dtype=np.int32)
        env._agents[a].deactivate()


def test_no_rewards_after_all_at_destination_termination(Env):
    """test_rewards_not_issued_for_terminated_agents_all_termination :276-327."""
    from collectivecrossing_amd.configs import AllAtDestinationTerminatedConfig, BinaryRewardConfig

    env = Env(config=_config(reward_config=BinaryRewardConfig(goal_reward=10.0, no_goal_reward=-1.0),
                             terminated_config=AllAtDestinationTerminatedConfig()))
    obs, _ = env.reset(seed=42)
    _park_on_destination(env, "boarding_0", "exiting_0")
    obs, rewards, terminateds, _, _ = env.step(_sampled_actions(env, obs))
    assert terminateds["boarding_0"] and terminateds["exiting_0"] and terminateds["__all__"]
    obs, rewards, terminateds, _, _ = env.step(_sampled_actions(env, obs))
    assert rewards == {}
    env.close()


def test_no_rewards_after_truncation(Env):
    """test_rewards_not_issued_for_truncated_agents :330-369 (max_steps = 1)."""
    from collectivecrossing_amd.configs import ConstantNegativeRewardConfig, MaxStepsTruncatedConfig

    env = Env(config=_config(reward_config=ConstantNegativeRewardConfig(step_penalty=-2.5),
                             truncated_config=MaxStepsTruncatedConfig(max_steps=1)))
    obs, _ = env.reset(seed=42)
    obs, rewards, _, truncateds, _ = env.step(_sampled_actions(env, obs))
    assert truncateds["boarding_0"] and truncateds["exiting_0"] and truncateds["__all__"]
    assert rewards == {"boarding_0": -2.5, "exiting_0": -2.5}
    obs, rewards, _, truncateds, _ = env.step(_sampled_actions(env, obs))
    assert rewards == {} and truncateds == {"__all__": False}
    env.close()


@pytest.mark.parametrize("reward", ["default", "simple_distance"])
def test_one_agent_done_the_other_still_paid(Env, reward):
    """test_default_reward_function_respects_termination :372-420, test_mixed_termination_states
    :423-473."""
    from collectivecrossing_amd import configs as C

    rc = C.DefaultRewardConfig() if reward == "default" else C.SimpleDistanceRewardConfig(distance_penalty_factor=0.2)
    env = Env(config=_config(reward_config=rc, terminated_config=C.IndividualAtDestinationTerminatedConfig()))
    obs, _ = env.reset(seed=42)
    _park_on_destination(env, "boarding_0")
    obs, rewards, terminateds, _, _ = env.step(_sampled_actions(env, obs))
    assert terminateds["boarding_0"]
    obs, rewards, terminateds, _, _ = env.step(_sampled_actions(env, obs))
    assert terminateds["boarding_0"] and "boarding_0" not in rewards
    assert not terminateds["exiting_0"] and list(rewards) == ["exiting_0"]
    assert isinstance(rewards["exiting_0"], float)
    if reward == "simple_distance":
        assert rewards["exiting_0"] < 0
    env.close()


# ------------------------------------------------------------------------ test_terminateds.py
@pytest.mark.parametrize("mode", ["all_at_destination", "individual_at_destination", None])
def test_nobody_terminates_on_the_first_random_step(Env, mode):
    """test_all_at_destination_terminated_function :29, test_individual_... :54,
    test_default_terminated_function :82 (default = individual), test_terminated_function_consistency
    :101, test_terminated_function_config_structure :148."""
    from collectivecrossing_amd import configs as C

    kw = {} if mode is None else {"terminated_config": C.get_terminated_config(mode)}
    env = Env(config=_config(**kw))
    assert env.config.terminated_config.get_terminated_function_name() == (mode or "individual_at_destination")
    obs, _ = env.reset(seed=42)
    _, _, terminateds, _, _ = env.step({a: int(env.action_spaces[a].sample()) for a in obs})
    assert not terminateds["boarding_0"] and not terminateds["exiting_0"] and not terminateds["__all__"]
    assert not env._calculate_terminated("boarding_0") and not env._calculate_terminated("exiting_0")
    env.close()


# --------------------------------------------------------------- test_action_agent_validity.py
def test_validity_check_edge_cases(Env):
    """TestActionAgentValidity :33-180: every valid action for every agent type passes; 999, -1,
    1000000 name the action and the agent; "", None, "boarding 0", "invalid_agent" are unknown ids."""
    env = Env(config=_config(num_boarding_agents=2, exiting_destination_area_y=1, boarding_destination_area_y=7))
    env.reset(seed=42)
    for a in env.possible_agents:
        for action in range(5):
            env._check_action_and_agent_validity(a, action)
    for bad in (999, -1, 1000000):
        with pytest.raises(ValueError) as ei:
            env._check_action_and_agent_validity("boarding_0", bad)
        assert "Invalid action" in str(ei.value) and str(bad) in str(ei.value) and "boarding_0" in str(ei.value)
    for who in ("", None, "boarding 0", "invalid_agent"):
        with pytest.raises(ValueError) as ei:
            env._check_action_and_agent_validity(who, 0)
        assert "Unknown agent ID" in str(ei.value) and str(who) in str(ei.value)
    with pytest.raises(ValueError) as ei:      # unknown id is reported before the bad action (:73-85)
        env._check_action_and_agent_validity("invalid_agent", 999)
    assert "Unknown agent ID" in str(ei.value)
    env.close()


# ----------------------------------------------------------------------- test_greedy_policy.py
def test_greedy_policy_is_deterministic_and_makes_progress(Env):
    """test_policy_consistency :87-122 (same state, same action), test_greedy_policy(_simple) :15,
    :125 (20 steps of the demo loop run without error and move agents towards their rows)."""
    from collectivecrossing_amd import baseline_policies as bp
    from collectivecrossing_amd.configs import MaxStepsTruncatedConfig

    env = Env(config=_config(height=6, division_y=3, tram_door_left=3, tram_door_right=4, num_boarding_agents=2,
                             exiting_destination_area_y=0, boarding_destination_area_y=5,
                             truncated_config=MaxStepsTruncatedConfig(max_steps=50)))
    policy = bp.create_greedy_policy(0.0)
    obs, _ = env.reset(seed=42)
    for a in env.agents:
        assert len({policy.get_action(a, obs[a], env) for _ in range(5)}) == 1
    dist0 = {a: abs(env._agents[a].y - env.get_agent_destination_position(a)[1]) for a in env.agents}
    terminateds = dict.fromkeys(obs, False)
    for _ in range(20):
        acts = {a: policy.get_action(a, obs[a], env) for a in env.agents
                if a in obs and not terminateds.get(a, False)}
        assert all(v in range(5) for v in acts.values())
        obs, _, terminateds, truncateds, _ = env.step(acts)
        if terminateds["__all__"] or truncateds["__all__"]:
            break
    dist1 = {a: abs(env._agents[a].y - env.get_agent_destination_position(a)[1]) for a in dist0}
    assert sum(dist1.values()) < sum(dist0.values())
    env.close()
