"""Host logic of rllib.BatchedMultiAgentEnv without a GPU: the flat action dict -> arrays encoder and the array -> flat
dicts builder against the single-env functions they generalise (env.encode_actions / env.decode_step, which the recorded
reference episodes pin), and the attribute surface RLlib's MultiAgentEnv.__init__ touches."""

import numpy as np
import pytest

from collectivecrossing_amd import configs as C
from collectivecrossing_amd.env import decode_step, encode_actions
from collectivecrossing_amd.rllib import BatchedMultiAgentEnv


def _cfg():
    return C.CollectiveCrossingConfig(width=12, height=8, division_y=4, tram_door_left=5, tram_door_right=7, tram_length=9,
                                      num_boarding_agents=3, num_exiting_agents=2, exiting_destination_area_y=0,
                                      boarding_destination_area_y=8)


def _host(E, **kw):
    env = BatchedMultiAgentEnv(_cfg(), E, _host_only=True, **kw)
    env._started = True
    return env


def test_agents_are_listed_before_the_first_reset_and_assignments_are_harmless():
    """ray's MultiAgentEnv.__init__ does `if not self.agents: self.agents = list(self._agent_ids)`: `agents` must be
    non-empty before reset() and both `agents` / `possible_agents` must tolerate an assignment."""
    env = BatchedMultiAgentEnv(_cfg(), 3, _host_only=True)
    assert env.agents == env.possible_agents and len(env.agents) == 15
    env.agents = ["x"]
    env.possible_agents = ["y"]
    assert env.agents == env.possible_agents == [f"{e}/{a}" for e in range(3)
                                                 for a in ("boarding_0", "boarding_1", "boarding_2", "exiting_0", "exiting_1")]

    class RayLikeBase:                                   # what the real base class does with the attributes
        def __init__(self):
            if not self.agents:
                self.agents = ["inferred"]
            self.possible_agents = list(self.agents)

    RayLikeBase.__init__(env)
    assert len(env.agents) == 15


@pytest.mark.parametrize("seed", range(5))
def test_flat_encoder_equals_the_single_env_encoder(seed):
    rng = np.random.default_rng(seed)
    E = 7
    env = _host(E)
    ids = env._local_ids
    N = len(ids)
    env._finished[2] = True                                       # entries of a finished env are ignored
    per_env = []
    for e in range(E):
        listed = [a for a in ids if rng.random() < 0.8]
        if seed % 2:
            rng.shuffle(listed)
        per_env.append({a: int(rng.integers(0, 5)) for a in listed})
    flat, k = {}, 0                                               # entries of different envs interleaved
    items = [list(d.items()) for d in per_env]
    while any(k < len(v) for v in items):
        for e, v in enumerate(items):
            if k < len(v):
                flat[f"{e}/{v[k][0]}"] = v[k][1]
        k += 1
    a, o = env._encode(flat)
    for e in range(E):
        xa, xo = encode_actions(ids, per_env[e] if e != 2 else {})
        assert np.array_equal(a[e], xa), e
        if o is not None and e != 2:
            # absent agents never move: only the relative order of the listed ones matters
            listed = [i for i in xo.tolist() if xa[i] != 255]
            assert [i for i in o[e].tolist() if xa[i] != 255] == listed, e
            assert sorted(o[e].tolist()) == list(range(N))
    if seed % 2 == 0:
        assert o is None                                          # slot order everywhere: no order tensor


def test_flat_encoder_raises_the_reference_errors():
    env = _host(2)
    with pytest.raises(ValueError, match="Unknown agent ID: 5/boarding_0"):
        env._encode({"0/boarding_0": 1, "5/boarding_0": 1})
    with pytest.raises(ValueError, match="Unknown agent ID"):
        env._encode({"boarding_0": 1})
    with pytest.raises(ValueError, match="Invalid action: 9 for agent 1/exiting_1"):
        env._encode({"0/boarding_0": 1, "1/exiting_1": 9})
    with pytest.raises(ValueError, match="Invalid action"):
        env._encode({"0/boarding_0": "up"})
    a, o = env._encode({})
    assert (a == 255).all() and o is None


@pytest.mark.parametrize("seed", range(4))
def test_flat_dicts_equal_the_single_env_decoder(seed):
    rng = np.random.default_rng(100 + seed)
    E = 6
    env = _host(E)
    ids, N = env._local_ids, len(env._local_ids)
    L = 6 + 4 * N
    af = rng.integers(0, 256, size=(E, N)).astype(np.uint8)
    rew = rng.normal(size=(E, N))
    obs = rng.integers(-1, 12, size=(E, N, L)).astype(np.float32)
    sel = rng.random(E) < 0.7
    types = ["boarding"] * 3 + ["exiting"] * 2
    got = env._dicts(sel, af, rew, obs, env._keys)
    exp = ({}, {}, {}, {}, {})
    for e in np.flatnonzero(sel):
        one = decode_step(ids, obs[e], rew[e], af[e], 0, types)
        one[2].pop("__all__"), one[3].pop("__all__")
        for src, dst in zip(one, exp):
            dst.update({f"{e}/{k}": v for k, v in src.items()})
    assert list(got[0]) == list(exp[0])
    for k in exp[0]:
        assert got[0][k].dtype == np.float32 and np.array_equal(got[0][k], exp[0][k])
    assert got[1] == exp[1] and got[2] == exp[2] and got[3] == exp[3] and got[4] == exp[4]
    a_key = next(iter(got[4]))
    got[4][a_key]["active"] = "mutated"                            # infos are fresh dicts, not shared templates
    assert all(v["active"] != "mutated" for k, v in got[4].items() if k != a_key)
    assert "mutated" not in [t["active"] for t in env._info_tpl.values()]
