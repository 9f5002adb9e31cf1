"""The drop-in dict API on the GPU: reference goldens through ``CollectiveCrossingEnv`` and the
behaviours the reference's unit tests pin (tests/collectivecrossing/envs/*.py, cited per test)."""

import json

import numpy as np
import pytest
from _fixtures import GOLDEN, Golden, config_from_dict

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Env():
    from collectivecrossing_amd import CollectiveCrossingEnv

    return CollectiveCrossingEnv


def _cfg(**kw):
    from collectivecrossing_amd.configs import CollectiveCrossingConfig, MaxStepsTruncatedConfig

    d = dict(width=10, height=8, division_y=4, tram_door_left=4, tram_door_right=5, tram_length=8,
             num_boarding_agents=2, num_exiting_agents=1, exiting_destination_area_y=1,
             boarding_destination_area_y=7, truncated_config=MaxStepsTruncatedConfig(max_steps=100))
    d.update(kw)
    return CollectiveCrossingConfig(**d)


@pytest.mark.parametrize("fn", ["golden_basic_trajectory.json", "regression_test.json"])
def test_reference_golden_trajectories_replay_through_the_dict_api(Env, fn):
    """The reference's own VCR replay (test_trajectory_vcr.py:123-197): obs array_equal,
    rewards < 1e-6 (here: exact), terminated/truncated/infos equal -- from reset(seed=42)."""
    d = json.loads((GOLDEN / "reference" / fn).read_text())
    env = Env(config=config_from_dict({k: v for k, v in d["config"].items() if k != "render_mode"}))
    obs, infos = env.reset(seed=42)
    assert {k: v.tolist() for k, v in obs.items()} == d["initial_observations"]
    assert infos == d["initial_infos"]
    for st in d["steps"]:
        obs, rew, term, trunc, infos = env.step(st["active_actions"])
        assert {k: v.tolist() for k, v in obs.items()} == st["next_observations"]
        assert rew == st["next_rewards"]
        assert term == st["next_terminated"] and trunc == st["next_truncated"]
        assert infos == st["next_infos"]
    env.close()


@pytest.mark.parametrize("name", ["g2_c1_shuffled_absent", "g4_small_all_at_dest_greedy", "g5_edges_default"])
def test_dict_api_equals_recorded_reference_steps(Env, name):
    """Full episodes through dicts (incl. shuffled dict order and omitted agents) vs the arrays
    recorded from the reference."""
    from collectivecrossing_amd.params import agent_ids

    g = Golden(name)
    ids = agent_ids(g.config)
    env = Env(config=g.config)
    for e in range(min(g.E, 4)):
        env.reset(seed=0)
        for i, a in enumerate(ids):
            ag = env._agents[a]
            ag.position = np.array([g["init_x"][e, i], g["init_y"][e, i]], dtype=np.int32)
            ag.active, ag.terminated, ag.truncated = (bool(g[k][e, i]) for k in
                                                      ("init_active", "init_terminated", "init_truncated"))
        env._step_count = int(g["init_step_count"][e])
        for s in range(g.K):
            acts = {ids[k]: int(g["actions"][s, e, k]) for k in g["order"][s, e] if g["actions"][s, e, k] != 255}
            obs, rew, term, trunc, infos = env.step(acts)
            af = g["agent_flags"][s, e]
            assert set(obs) == {a for i, a in enumerate(ids) if af[i] & 8} == set(infos)
            assert set(rew) == {a for i, a in enumerate(ids) if af[i] & 4} == set(trunc) - {"__all__"}
            for i, a in enumerate(ids):
                assert term[a] == bool(af[i] & 1)
                if a in rew:
                    assert np.float64(rew[a]).view(np.uint64) == g["reward"][s, e, i].view(np.uint64)
                    assert trunc[a] == bool(af[i] & 2)
                if a in obs:
                    np.testing.assert_array_equal(obs[a], g["obs"][s, e, i])
                assert env._agents[a].x == g["x"][s, e, i] and env._agents[a].y == g["y"][s, e, i]
                assert env._agents[a].active == bool(g["active"][s, e, i])
            assert term["__all__"] == bool(g["env_flags"][s, e] & 1)
            assert trunc["__all__"] == bool(g["env_flags"][s, e] & 2)
            assert env._step_count == g["step_count"][s, e]
    env.close()


def test_construction_spaces_and_reset(Env):
    """test_collective_crossing.py:11-72, :280-352."""
    env = Env(config=_cfg())
    assert env.possible_agents == ["boarding_0", "boarding_1", "exiting_0"]
    assert env.action_space.n == 5 and set(env.action_spaces) == set(env.possible_agents)
    assert env.observation_space.shape == (2 + 4 + 4 * 3,) and env.observation_space.dtype == np.float32
    assert env.get_observation_space("boarding_0") is env.observation_space
    assert (env.tram_left, env.tram_right, env.tram_door_left, env.tram_door_right) == (1, 9, 5, 6)
    obs, infos = env.reset(seed=42)
    assert set(obs) == set(env.possible_agents) == set(env.agents)
    assert infos["exiting_0"] == {"agent_type": "exiting"}
    o = obs["boarding_1"]
    assert o.dtype == np.float32 and o.shape == (18,)
    assert o[2:6].tolist() == [5.0, 4.0, 5.0, 6.0]                 # door centre, division, door l/r
    assert o[10:14].tolist() == [-1.0] * 4                          # own slot
    assert o[6:8].tolist() == obs["boarding_0"][:2].tolist() and o[8:10].tolist() == [0.0, 1.0]
    assert o[14:18].tolist() == obs["exiting_0"][:2].tolist() + [1.0, 1.0]
    obs2, _ = env.reset(seed=42)
    assert all(np.array_equal(obs[a], obs2[a]) for a in obs)       # same seed, same placement
    env.close()


def test_wait_action_keeps_positions_and_observations(Env):
    """test_collective_crossing.py:75-113, :355-393."""
    env = Env(config=_cfg(num_exiting_agents=2))
    obs, _ = env.reset(seed=42)
    new_obs, rew, term, trunc, infos = env.step(dict.fromkeys(obs, 4))
    assert len(new_obs) == 4
    for a in obs:
        assert np.array_equal(new_obs[a], obs[a])
    assert not term["__all__"] and not trunc["__all__"]
    env.close()


def test_forced_positions_terminate(Env):
    """test_collective_crossing.py:116-152."""
    env = Env(config=_cfg(width=8, height=6, division_y=3, tram_door_left=3, tram_door_right=4, tram_length=8,
                          num_boarding_agents=1, num_exiting_agents=1, exiting_destination_area_y=0,
                          boarding_destination_area_y=5, render_mode="human"))
    obs, _ = env.reset(seed=42)
    for a in obs:
        env._agents[a].update_position(np.array([4, 5]) if a.startswith("boarding") else np.array([4, 0]))
    new_obs, rew, term, trunc, infos = env.step(dict.fromkeys(obs, 4))
    assert term["boarding_0"] and term["exiting_0"] and term["__all__"]
    assert infos["boarding_0"]["at_destination"] and not infos["boarding_0"]["active"]
    assert env.agents == []
    env.close()


def test_invalid_actions_and_agents_raise_like_the_reference(Env):
    """test_collective_crossing.py:207-235, test_action_agent_validity.py."""
    env = Env(config=_cfg())
    env.reset(seed=42)
    before = {a: env._agents[a].position.tolist() for a in env.possible_agents}
    with pytest.raises(ValueError, match="Invalid action"):
        env.step({"boarding_0": 0, "boarding_1": 10})
    # whole-dict validation: nothing moved, step counter untouched (documented difference)
    assert before == {a: env._agents[a].position.tolist() for a in env.possible_agents}
    assert env._step_count == 0
    with pytest.raises(ValueError, match="Unknown agent ID"):
        env.step({"invalid_agent": 0})
    env._check_action_and_agent_validity("boarding_0", 0)
    with pytest.raises(ValueError) as ei:
        env._check_action_and_agent_validity("invalid_agent", 999)
    assert "Unknown agent ID" in str(ei.value)
    with pytest.raises(ValueError, match="Unknown agent ID"):
        env._get_agent("nobody")
    env.close()


def test_unknown_strategy_names_raise_at_construction(Env):
    """test_rewards.py:218, test_collective_crossing.py:256, test_terminateds.py:139."""
    from collectivecrossing_amd.configs import (CustomRewardConfig, CustomTerminatedConfig, CustomTruncatedConfig,
                                                ObservationConfig)

    with pytest.raises(ValueError, match="Unknown reward function"):
        Env(config=_cfg(reward_config=CustomRewardConfig(reward_function="invalid")))
    with pytest.raises(ValueError, match="Unknown termination function"):
        Env(config=_cfg(terminated_config=CustomTerminatedConfig(terminated_function="invalid")))
    with pytest.raises(ValueError, match="Unknown truncation function"):
        Env(config=_cfg(truncated_config=CustomTruncatedConfig(truncated_function="invalid")))
    with pytest.raises(ValueError, match="Unknown observation function"):
        Env(config=_cfg(observation_config=ObservationConfig(observation_function="invalid")))


def test_no_rewards_for_terminated_agents(Env):
    """test_rewards.py:222-273: a deactivated agent forced onto its destination terminates on
    the next step, keeps terminateds[id] == True afterwards and stops receiving rewards."""
    from collectivecrossing_amd.configs import SimpleDistanceRewardConfig

    env = Env(config=_cfg(num_boarding_agents=1, reward_config=SimpleDistanceRewardConfig(distance_penalty_factor=0.2)))
    obs, _ = env.reset(seed=42)
    dest = env.get_agent_destination_position("boarding_0")
    assert dest == (None, 7)
    env._agents["boarding_0"].position = np.array([5, dest[1]], dtype=np.int32)
    env._agents["boarding_0"].deactivate()
    with pytest.raises(ValueError, match="already deactivated"):
        env._agents["boarding_0"].deactivate()
    acts = {a: env.action_spaces[a].sample() for a in obs if env._agents[a].active}
    obs, rew, term, trunc, infos = env.step(acts)
    assert term["boarding_0"] and "boarding_0" in rew and "boarding_0" in obs    # final reward + obs
    acts = {a: env.action_spaces[a].sample() for a in obs if env._agents[a].active}
    obs, rew, term, trunc, infos = env.step(acts)
    assert term["boarding_0"] and "boarding_0" not in rew and "boarding_0" not in obs
    assert not term["exiting_0"] and isinstance(rew["exiting_0"], float) and rew["exiting_0"] <= 0
    assert env.agents == ["exiting_0"]
    env.close()


@pytest.mark.parametrize("current_step,max_steps,expected", [
    (5, 10, False), (9, 10, False), (10, 10, True), (15, 10, True), (0, 1, False), (1, 1, True),
    (0, 10, False), (100000, 100000, True)])
def test_truncation_table_via_step_count(Env, current_step, max_steps, expected):
    """test_truncateds.py:14-55 -- both through the strategy object on the host mirror and
    through a real GPU step from step_count - 1."""
    from collectivecrossing_amd.configs import MaxStepsTruncatedConfig

    env = Env(config=_cfg(num_boarding_agents=1, truncated_config=MaxStepsTruncatedConfig(max_steps=max_steps)))
    env.reset(seed=1)
    env._step_count = current_step
    assert env._truncated_function.calculate_truncated("boarding_0", env) is expected
    env._step_count = current_step - 1 if current_step > 0 else 0
    if current_step > 0:
        _, _, _, trunc, _ = env.step({})
        assert trunc["boarding_0"] is expected and trunc["__all__"] is expected
    env.close()


def test_strategy_objects_answer_on_the_mirror_and_on_mocks(Env):
    """test_terminateds.py:29-98 (env._calculate_terminated) and test_rewards.py:476-527 (the
    reference's hand-rolled MockEnv objects work with the strategy classes)."""
    from collectivecrossing_amd.configs import AllAtDestinationTerminatedConfig, BinaryRewardConfig, SimpleDistanceRewardConfig
    from collectivecrossing_amd.rewards import BinaryRewardFunction, SimpleDistanceRewardFunction

    env = Env(config=_cfg(terminated_config=AllAtDestinationTerminatedConfig()))
    obs, _ = env.reset(seed=42)
    _, _, term, _, _ = env.step({a: env.action_spaces[a].sample() for a in obs})
    assert not any(term.values())
    assert not env._calculate_terminated("boarding_0") and not env._calculate_terminated("exiting_0")
    r = env._calculate_reward("boarding_0")
    assert r is not None and r == env._reward_function.calculate_reward("boarding_0", env)
    assert env._is_move_valid("boarding_0", env._get_agent_position("boarding_0"), np.array([-1, 0])) is False
    env.close()

    class MockEnv:
        def __init__(self, **flags):
            self._agents = {"agent_0": type("Agent", (), flags)()}

        def _get_agent_position(self, agent_id):
            return np.array([0, 0])

        def get_agent_destination_position(self, agent_id):
            return np.array([5, 5])

    sd = SimpleDistanceRewardFunction(SimpleDistanceRewardConfig(distance_penalty_factor=0.2))
    assert sd.calculate_reward("agent_0", MockEnv(terminated=True, truncated=False)) is None
    bn = BinaryRewardFunction(BinaryRewardConfig(goal_reward=10.0, no_goal_reward=-1.0))
    assert bn.calculate_reward("agent_0", MockEnv(terminated=False, truncated=True)) is None
    assert bn.calculate_reward("agent_0", MockEnv(terminated=False, truncated=False)) == -1.0


def test_host_reward_strategies_agree_with_the_gpu(Env):
    """The host-side strategy mirror and the kernel give the same f64 on random states."""
    from collectivecrossing_amd.configs import DefaultRewardConfig, SimpleDistanceRewardConfig

    for rc in (DefaultRewardConfig(distance_penalty_factor=0.37), SimpleDistanceRewardConfig(distance_penalty_factor=1.3)):
        env = Env(config=_cfg(num_exiting_agents=2, reward_config=rc))
        obs, _ = env.reset(seed=7)
        rng = np.random.default_rng(0)
        for _ in range(40):
            _, rew, _, _, _ = env.step({a: int(rng.integers(0, 5)) for a in env.agents})
            for a, v in rew.items():
                # the mirror is post-step; done flags were applied after the reward was taken
                ag = env._agents[a]
                t, u = ag.terminated, ag.truncated
                env._mirror.terminated[ag._index] = env._mirror.truncated[ag._index] = 0
                host = env._calculate_reward(a)
                env._mirror.terminated[ag._index], env._mirror.truncated[ag._index] = t, u
                env._mirror.dirty = False
                assert np.float64(host).view(np.uint64) == np.float64(v).view(np.uint64), (a, host, v)
        env.close()


def test_mirror_stays_in_sync_with_the_device_state(Env):
    """step() rebuilds the host mirror from the step outputs (no state read-back): it must equal
    the device state after every step, through arrivals, truncation and forced writes."""
    from collectivecrossing_amd.configs import MaxStepsTruncatedConfig

    env = Env(config=_cfg(num_boarding_agents=3, num_exiting_agents=2, truncated_config=MaxStepsTruncatedConfig(max_steps=25)))
    rng = np.random.default_rng(3)
    for ep in range(3):
        env.reset(seed=ep)
        for t in range(30):
            env.step({a: int(rng.integers(0, 5)) for a in env.possible_agents if rng.random() < 0.9})
            st = env._batch.get_state()
            m = env._mirror
            for k in ("x", "y", "active", "terminated", "truncated"):
                np.testing.assert_array_equal(getattr(m, k), st[k][0], err_msg=f"{k} ep {ep} t {t}")
            assert m.step_count == int(st["step_count"][0])
            if t == 10:
                env._agents["exiting_0"].position = np.array([3, 2])
    env.close()


def test_vector_adapter_views_equal_independent_dict_envs(Env):
    """VectorCollectiveCrossing (one batch, lazy per-env dicts, seeded reset of finished envs on
    the device) against E independent drop-in envs fed the same dicts."""
    from collectivecrossing_amd.configs import MaxStepsTruncatedConfig
    from collectivecrossing_amd.vector import VectorCollectiveCrossing

    cfg = _cfg(num_boarding_agents=3, num_exiting_agents=2, truncated_config=MaxStepsTruncatedConfig(max_steps=12))
    E = 6
    vec = VectorCollectiveCrossing(cfg, E)
    singles = [Env(config=cfg) for _ in range(E)]
    seeds = np.arange(100, 100 + E, dtype=np.uint64)
    obs0 = vec.reset(seeds).cpu().numpy()
    for e, env in enumerate(singles):
        o, _ = env.reset(seed=int(seeds[e]))
        for i, a in enumerate(vec.agent_ids):
            np.testing.assert_array_equal(o[a], obs0[e, i])
    rng = np.random.default_rng(9)
    next_seed = 1000
    for t in range(40):
        dicts = []
        for env in singles:
            ids = list(env.agents)
            rng.shuffle(ids)
            dicts.append({a: int(rng.integers(0, 5)) for a in ids})
        vec.step_dicts(dicts)
        done = vec.done_mask().cpu().numpy()
        for e, env in enumerate(singles):
            o, r, te, tr, inf = env.step(dicts[e])
            vo, vr, vte, vtr, vinf = vec.view(e)
            assert r == vr and te == vte and tr == vtr and inf == vinf
            assert set(o) == set(vo) and all(np.array_equal(o[a], vo[a]) for a in o)
            assert bool(done[e]) == (te["__all__"] or tr["__all__"])
        if done.any():
            new = np.arange(next_seed, next_seed + E, dtype=np.uint64)
            next_seed += E
            vec.reset_done(new)
            for e, env in enumerate(singles):
                if done[e]:
                    env.reset(seed=int(new[e]))
    assert next_seed > 1000
    vec.close()
    for env in singles:
        env.close()


def test_zero_copy_and_staged_step_io_agree(Env, monkeypatch):
    """The default step writes its outputs straight into pinned host memory (ccx_host_device_pointer);
    CCX_ENV_STAGED=1 uses device buffers + two copies.  Same dicts, same lazily synced mirror."""
    rng = np.random.default_rng(3)
    cfg = _cfg()
    runs = []
    for staged in ("0", "1"):
        monkeypatch.setenv("CCX_ENV_STAGED", staged)
        env = Env(config=cfg)
        env.reset(seed=11)
        rng = np.random.default_rng(3)
        trace = []
        for _ in range(60):
            acts = {a: int(rng.integers(0, 5)) for a in env.agents}
            obs, rew, term, trunc, infos = env.step(acts)
            trace.append(({k: v.tolist() for k, v in obs.items()}, rew, term, trunc, infos,
                          {a: (ag.x, ag.y, ag.active, ag.terminated, ag.truncated)
                           for a, ag in env._agents.items()}, list(env.agents)))
        assert env._zero_copy == (staged == "0")
        runs.append(trace)
        env.close()
    assert runs[0] == runs[1]


def test_host_device_pointer_rejects_pageable_memory(Env):
    import ctypes as C

    env = Env(config=_cfg())
    b = env._batch
    pageable = np.zeros(64, np.uint8)
    out = C.c_void_p()
    rc = b._lib.ccx_host_device_pointer(b._h, C.c_void_p(pageable.ctypes.data), C.byref(out))
    assert rc != 0 and b"page-locked" in b._lib.ccx_last_error()
    env.close()


@pytest.mark.parametrize("policy_name", ["greedy", "waiting"])
def test_reference_demo_loop_with_the_host_policy_classes(Env, policy_name):
    """scripts/run_greedy_policy_demo.py:67-109 / run_waiting_policy_demo.py: policy.get_action per
    live agent, env.step, until __all__ -- through the dict API on the GPU, and the same episode
    from the fused on-device policy rollout."""
    from collectivecrossing_amd import baseline_policies as bp
    from collectivecrossing_amd.batched import BatchedCollectiveCrossing
    from collectivecrossing_amd.params import agent_ids

    cfg = _cfg(num_boarding_agents=3, num_exiting_agents=2, tram_door_left=3, tram_door_right=5)
    ids = agent_ids(cfg)
    env = Env(config=cfg)
    obs, _ = env.reset(seed=7)
    start = np.array([env._agents[a].position for a in ids])
    policy = (bp.create_greedy_policy if policy_name == "greedy" else bp.create_waiting_policy)(0.0)
    taken, rewards_seen = [], []
    for _ in range(100):
        acts = {a: policy.get_action(a, obs[a], env) for a in env.agents}
        row = np.full(len(ids), 255, np.uint8)
        for a, v in acts.items():
            row[ids.index(a)] = v
        taken.append(row)
        obs, rew, term, trunc, _ = env.step(acts)
        rewards_seen.append(rew)
        if term["__all__"] or trunc["__all__"]:
            break
    assert term["__all__"] or trunc["__all__"]    # (head-on greedy agents may deadlock at a one-cell door)
    batch = BatchedCollectiveCrossing(cfg, 1)
    batch.set_state(x=start[None, :, 0], y=start[None, :, 1])
    res, fused_actions = batch.rollout_greedy(len(taken), policy=policy_name)
    np.testing.assert_array_equal(fused_actions.cpu().numpy()[:, 0], np.stack(taken))
    last = res.reward.cpu().numpy()[-1, 0]
    for a, r in rewards_seen[-1].items():
        assert r == last[ids.index(a)]
    batch.close()
    env.close()
