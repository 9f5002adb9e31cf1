"""Pins the oracle's restatement of numpy's legacy RandomState (oracle/ccx_oracle.c: ccxo_mt_*) and of the way the
reference's epsilon policies consume it (greedy_policy.py:31,48-59, waiting_policy.py:31,48-59):

* against numpy itself -- raw MT19937 words, `random()` doubles, `choice(list)` over lists of 1..5 entries (a list of
  one entry draws nothing);
* against the reference: the epsilon episodes recorded from the reference's own GreedyPolicy / WaitingPolicy with
  randomness_factor > 0 and RandomState(42) (g11_epsilon_policy_*, one policy object per env): the oracle's policy rollout
  with the MT19937 stream reproduces every action and every step output."""

import numpy as np
import pytest
from _fixtures import ALL_NPZ, Golden, assert_step_matches

EPSILON = [n for n in ALL_NPZ if n.startswith("g11_epsilon_policy_")]
LISTS = [[4], [1, 4], [0, 2, 4], [0, 1, 3, 4], [0, 1, 2, 3, 4]]


@pytest.mark.parametrize("seed", [42, 0, 1, 2**32 - 1, 20240611])
def test_mt19937_restatement_against_numpy(oracle, seed):
    n, counts = 4000, [1, 2, 3, 4, 5, 5, 3, 1, 2, 4]          # 4000 words = more than six twists of the state
    raw, dbl, picks = oracle.mt_probe(seed, n, counts)
    rs = np.random.RandomState(seed)
    np.testing.assert_array_equal(raw, rs.randint(0, 2**32, size=n, dtype=np.uint32))
    rs = np.random.RandomState(seed)
    np.testing.assert_array_equal(dbl.view(np.uint64), np.array([rs.random() for _ in range(n)]).view(np.uint64))
    rs = np.random.RandomState(seed)
    want = [rs.choice(LISTS[counts[i % len(counts)] - 1]) for i in range(n)]
    got = [LISTS[counts[i % len(counts)] - 1][picks[i]] for i in range(n)]
    assert got == want
    # a one-entry list consumes nothing: the generator is where it was
    a, b = np.random.RandomState(seed), np.random.RandomState(seed)
    a.choice([4])
    assert a.random() == b.random()


def test_there_are_epsilon_fixtures():
    assert len(EPSILON) >= 3


@pytest.mark.parametrize("name", EPSILON)
def test_policy_rollout_with_the_reference_stream_replays_the_recorded_epsilon_episodes(oracle, name):
    g = Golden(name)
    eps = float(g["epsilon"])
    policy = "waiting" if "_w_" in name else "greedy"
    b = oracle.OracleBatch(g.params, g.E)
    b.set_state(**g.init_state())
    b.set_policy_stream_mt19937(42, eps)
    acts, obs, rew, af, ef = b.rollout_greedy(g.K, policy=policy)
    np.testing.assert_array_equal(acts, g["actions"])
    # (the outputs of the LAST step are checked with the final state; the others step by step through a second batch)
    b2 = oracle.OracleBatch(g.params, g.E)
    b2.set_state(**g.init_state())
    b2.set_policy_stream_mt19937(42, eps)
    explored = 0
    for s in range(g.K):
        a1 = b2.policy_actions(policy, with_epsilon=True)              # one policy call = one pass over the stream
        np.testing.assert_array_equal(a1, g["actions"][s], err_msg=f"{name} step {s}")
        explored += int((a1 != b2.policy_actions(policy)).sum())          # (the epsilon-0 call never touches the stream)
        o, r, f, e = b2.step(a1)
        assert_step_matches(g, s, o, r, f, e, dict(x=b2.x, y=b2.y, active=b2.active, terminated=b2.terminated,
                                                   truncated=b2.truncated, step_count=b2.step_count))
        np.testing.assert_array_equal(o.view(np.uint32), obs[s].view(np.uint32))
        np.testing.assert_array_equal(r.view(np.uint64), rew[s].view(np.uint64))
        np.testing.assert_array_equal(f, af[s])
    assert explored > 0, "the fixture must contain exploration moves that differ from the policy's choice"
    # epsilon 0 with the stream attached draws nothing and is the deterministic policy
    b3 = oracle.OracleBatch(g.params, g.E)
    b3.set_state(**g.init_state())
    b3.set_policy_stream_mt19937(42, 0.0)
    np.testing.assert_array_equal(b3.policy_actions(policy, with_epsilon=True), b3.policy_actions(policy))
