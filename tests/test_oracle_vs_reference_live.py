"""LIVE pinning of the oracle where the reference is present (the build container; skipped on the GPU
box): fresh random reference-valid configs are run through the IMPORTED reference and through the
oracle in the same process, step by step -- positions, flags, rewards (bit patterns), observations,
and the scripted policies' actions.  The committed fixtures are a frozen sample of exactly this
comparison; this test redraws it every run with new seeds per round."""

import os
import sys
from pathlib import Path

import numpy as np
import pytest

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE / "golden"))
import gen_golden as G  # noqa: E402

pytestmark = pytest.mark.skipif(not (G.REF / "src" / "collectivecrossing").is_dir(),
                                reason="the reference is only present in the build container")


@pytest.fixture(scope="module")
def reference():
    G.import_reference()
    return G


class _Mem:
    """The arrays of a Recorder with the interface of _fixtures.Golden (no file in between)."""

    def __init__(self, rec, cfg):
        from _fixtures import config_from_dict
        from collectivecrossing_amd.params import lower_config

        self.a, self.name = rec.a, "live"
        self.config = config_from_dict(cfg)
        self.params = lower_config(self.config)
        self.K, self.E, self.N = rec.a["actions"].shape

    def __getitem__(self, k):
        return self.a[k]

    def init_state(self):
        return dict(x=self["init_x"], y=self["init_y"], active=self["init_active"],
                    terminated=self["init_terminated"], truncated=self["init_truncated"],
                    step_count=self["init_step_count"])


# new configs every round of the build, reproducible within one (the seed is printed on failure)
BASE = int(os.environ.get("CCX_LIVE_SEED", "20260000"))


@pytest.mark.parametrize("index", range(24))
def test_random_config_reference_vs_oracle(reference, oracle, index):
    from _fixtures import assert_step_matches
    from test_oracle_golden import _state

    cfg = G.cfg_fuzz(BASE % 100000 + index)
    K, seeds = 30, (BASE + 7 * index, BASE + 7 * index + 1)
    rec = G.Recorder(cfg, len(seeds), K)
    rng = np.random.default_rng(BASE + index)
    for e, seed in enumerate(seeds):
        env = rec.envs[e]
        env.reset(seed=int(seed))
        rec.snapshot_init(e)
        for s in range(K):
            ids = list(rec.ids)
            if index & 1:
                ids = [ids[i] for i in rng.permutation(len(ids))]
            rec.step(s, e, {a: int(rng.integers(0, 5)) for a in ids if rng.random() >= 0.1})
    g = _Mem(rec, cfg)
    b = oracle.OracleBatch(g.params, g.E)
    b.set_state(**g.init_state())
    for s in range(K):
        obs, rew, af, ef = b.step(g["actions"][s], g["order"][s])
        assert_step_matches(g, s, obs, rew, af, ef, _state(b), label=f"cfg {cfg}")


@pytest.mark.parametrize("index,policy", [(i, p) for i in range(8) for p in ("greedy", "waiting")])
def test_random_config_reference_policy_vs_oracle(reference, oracle, index, policy):
    from baseline_policies import GreedyPolicy, WaitingPolicy

    cfg = dict(G.cfg_fuzz(BASE % 100000 + 500 + index),
               truncated_config=dict(truncated_function="max_steps", max_steps=60))
    K = 40
    rec = G.Recorder(cfg, 1, K)
    env = rec.envs[0]
    env.reset(seed=BASE + index)
    rec.snapshot_init(0)
    pol = (GreedyPolicy if policy == "greedy" else WaitingPolicy)(randomness_factor=0.0, seed=42)
    g = _Mem(rec, cfg)
    b = oracle.OracleBatch(g.params, 1)
    b.set_state(**g.init_state())
    for s in range(K):
        acts = {a: int(pol.get_action(a, None, env)) for a in env.agents}
        mine = b.policy_actions(policy)[0]
        want = np.full(g.N, 255, np.uint8)
        for a, v in acts.items():
            want[rec.ids.index(a)] = v
        np.testing.assert_array_equal(mine, want, err_msg=f"{policy} step {s} cfg {cfg}")
        rec.step(s, 0, acts)
        b.step(rec.a["actions"][s], rec.a["order"][s], want_obs=False)


@pytest.mark.parametrize("index,policy,eps", [(0, "greedy", 0.3), (1, "waiting", 0.2), (2, "greedy", 0.6),
                                              (3, "waiting", 0.5), (4, "greedy", 0.1), (5, "waiting", 0.9)])
def test_host_policy_classes_vs_reference_policies_with_epsilon(reference, index, policy, eps):
    """collectivecrossing_amd.baseline_policies next to the reference's classes, both seeded with 42, on
    the same live reference env (mirrored into a host view each step): same action every call."""
    from _fixtures import config_from_dict
    from baseline_policies import GreedyPolicy, WaitingPolicy

    from collectivecrossing_amd import baseline_policies as bp
    from collectivecrossing_amd.env import CollectiveCrossingEnv

    cfg = dict(G.cfg_fuzz(BASE % 100000 + 900 + index),
               truncated_config=dict(truncated_function="max_steps", max_steps=50))
    from collectivecrossing import CollectiveCrossingEnv as RefEnv

    ref_env = RefEnv(config=G.build_ref_config(cfg))
    ref_env.reset(seed=BASE + index)
    ours = CollectiveCrossingEnv.host_view(config_from_dict(cfg))
    ref_pol = (GreedyPolicy if policy == "greedy" else WaitingPolicy)(randomness_factor=eps, seed=42)
    our_pol = (bp.GreedyPolicy if policy == "greedy" else bp.WaitingPolicy)(randomness_factor=eps, seed=42)
    ids = G.ids_of(cfg)
    for s in range(45):
        m = ours._mirror
        for i, a in enumerate(ids):
            ag = ref_env._agents[a]
            m.x[i], m.y[i] = int(ag.position[0]), int(ag.position[1])
            m.active[i], m.terminated[i], m.truncated[i] = ag.active, ag.terminated, ag.truncated
        m.touch()
        assert ours.agents == list(ref_env.agents)
        acts = {}
        for a in ref_env.agents:
            want = int(ref_pol.get_action(a, None, ref_env))
            got = int(our_pol.get_action(a, None, ours))
            assert got == want, f"{policy} eps {eps} step {s} agent {a} cfg {cfg}"
            acts[a] = want
        _, _, term, trunc, _ = ref_env.step(acts)
        if term["__all__"] or trunc["__all__"]:
            break
