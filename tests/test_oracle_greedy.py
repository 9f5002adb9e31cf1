"""Pins the oracle's restatement of the reference's GreedyPolicy (epsilon = 0) against the actions
the reference's own policy emitted while the g4_/g5_ greedy fixtures were recorded."""

import numpy as np
import pytest
from _fixtures import ALL_NPZ, Golden

GREEDY = [n for n in ALL_NPZ if "greedy" in n]
WAITING = [n for n in ALL_NPZ if "waiting" in n]


def test_there_are_greedy_fixtures():
    assert len(GREEDY) >= 5


@pytest.mark.parametrize("name", GREEDY)
def test_greedy_actions_match_the_reference_policy(oracle, name):
    g = Golden(name)
    b = oracle.OracleBatch(g.params, g.E)
    b.set_state(**g.init_state())
    seen = set()
    for s in range(g.K):
        acts = b.greedy_actions()
        np.testing.assert_array_equal(acts, g["actions"][s], err_msg=f"{name} step {s}")
        seen.update(np.unique(acts).tolist())
        b.step(g["actions"][s], g["order"][s], want_obs=False)
    assert {0, 1, 2, 3, 4} & seen, seen


@pytest.mark.parametrize("name", WAITING)
def test_waiting_policy_actions_match_the_reference_policy(oracle, name):
    g = Golden(name)
    b = oracle.OracleBatch(g.params, g.E)
    b.set_state(**g.init_state())
    waited = 0
    for s in range(g.K):
        acts = b.policy_actions("waiting")
        np.testing.assert_array_equal(acts, g["actions"][s], err_msg=f"{name} step {s}")
        waited += int(((acts == 4) & (b.greedy_actions() != 4)).sum())
        b.step(g["actions"][s], g["order"][s], want_obs=False)
    assert len(WAITING) >= 3
    if name.startswith("g9_"):   # (a fuzz config may have no exiting agents to wait for)
        assert waited > 0, "the fixture must contain steps where waiting != greedy"
