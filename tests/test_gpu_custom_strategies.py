"""User-registered strategy classes (no kernel_mode) on the GPU env: the host slow path of SURVEY 8b.

The reference's plugin registries accept any class (rewards.py:186-216, terminateds.py:12-34,86-114,
truncateds.py:12-34,99-128).  tests/golden/g12_custom_strategies.json.gz holds what the imported
reference returned, dict by dict, when the three plugins of tests/golden/custom_strategies.py were
registered with IT (all three, and each one alone next to built-in strategies); here the same classes
are registered with collectivecrossing_amd and the recorded episodes are replayed through the drop-in
env: the GPU does the moves / deactivation, phases collectivecrossing.py:214-259 run on the host mirror."""

import gzip
import json
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = Path(__file__).resolve().parent / "golden"
sys.path.insert(0, str(GOLDEN))


@pytest.fixture(scope="module")
def recorded():
    with gzip.open(GOLDEN / "g12_custom_strategies.json.gz") as z:
        return json.loads(z.read())


@pytest.fixture(scope="module")
def plugins():
    import custom_strategies as cs

    from collectivecrossing_amd import strategies as S
    made = cs.make(S.RewardFunction, S.TerminatedFunction, S.TruncatedFunction)
    S.REWARD_FUNCTIONS[cs.NAMES["reward"]] = made["reward"]
    S.TERMINATED_FUNCTIONS[cs.NAMES["terminated"]] = made["terminated"]
    S.TRUNCATED_FUNCTIONS[cs.NAMES["truncated"]] = made["truncated"]
    yield cs
    for table, key in ((S.REWARD_FUNCTIONS, "reward"), (S.TERMINATED_FUNCTIONS, "terminated"),
                       (S.TRUNCATED_FUNCTIONS, "truncated")):
        table.pop(cs.NAMES[key], None)


@pytest.mark.parametrize("mix", ["all", "reward", "terminated", "truncated"])
def test_recorded_reference_episodes_with_user_strategies_replay_exactly(recorded, plugins, mix):
    from collectivecrossing_amd import CollectiveCrossingEnv
    from collectivecrossing_amd import configs as C
    cs = plugins
    arrivals_paid = 0
    for ep in recorded[mix]:
        env = CollectiveCrossingEnv(config=cs.build_config(C, C, C, C, cs.MIXES[mix]))
        assert env._host_strategies
        obs, _ = env.reset(seed=ep["seed"])
        assert {k: v.tolist() for k, v in obs.items()} == ep["initial"]
        for a, pos in ep["forced"].items():      # the way the reference's tests poke env._agents[...]
            env._agents[a].position = np.array(pos)
        for s, st in enumerate(ep["steps"]):
            o, r, te, tr, inf = env.step(dict(st["actions"]))
            tag = f"{mix} seed {ep['seed']} step {s}"
            assert sorted(o) == sorted(st["observations"]), tag
            for k, v in o.items():
                assert v.dtype == np.float32 and v.tolist() == st["observations"][k], (tag, k)
            assert {k: float(v) for k, v in r.items()} == st["rewards"], tag          # keys AND values
            assert {k: bool(v) for k, v in te.items()} == st["terminateds"], tag
            assert {k: bool(v) for k, v in tr.items()} == st["truncateds"], tag
            assert inf == st["infos"], tag
            assert env.agents == st["agents"] and env._step_count == st["step_count"], tag
            for a, (act, term, trunc) in st["flags"].items():
                ag = env._agents[a]
                assert (ag.active, ag.terminated, ag.truncated) == (act, term, trunc), (tag, a)
            arrivals_paid += sum(1 for v in st["rewards"].values() if v > 50.0)
        env.close()
    if "reward" in cs.MIXES[mix]:
        assert arrivals_paid > 0    # the bonus on the finishing step was really exercised


def test_user_strategy_must_still_be_registered(plugins):
    from collectivecrossing_amd import CollectiveCrossingEnv
    from collectivecrossing_amd import configs as C
    with pytest.raises(ValueError, match="Unknown termination function 'nope'"):
        CollectiveCrossingEnv(config=C.CollectiveCrossingConfig(
            width=12, height=8, division_y=4, tram_door_left=5, tram_door_right=7, tram_length=9,
            num_boarding_agents=1, num_exiting_agents=1, exiting_destination_area_y=0, boarding_destination_area_y=8,
            terminated_config=C.CustomTerminatedConfig(terminated_function="nope")))
