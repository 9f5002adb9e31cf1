"""The short-launch kernel (csrc/ccx_step.hip): CollectiveCrossingEnv.step itself (collectivecrossing.py:161-261) for
launches of 1..16 env-steps -- one workgroup per tile (a sim wave + row waves, one LDS barrier per step), no ring / pacing.

ccx_step and short ccx_rollout calls take it by default (with or without a move order, no in-kernel policy); the tunable
`step_kernel` = 0 forces the rollout kernel.  Parity: (a) EVERY reference-recorded step fixture step by step, bit for bit, with
its recorded dict order (and, where that is the slot order, also without an order array); (b) the oracle, over launch lengths x batch sizes x agent counts (odd, 1, 50, 64) with
auto-reset; (c) the rollout kernel on the same inputs (the two kernels are independent implementations of the step);
(d) the step kernel is really the one that ran (its launch shape is reported, and a grid whose tables exceed the LDS
falls back)."""

import numpy as np
import pytest
from _fixtures import STEP_NPZ, Golden, assert_step_matches

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ccx():
    import torch

    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    from collectivecrossing_amd.batched import BatchedCollectiveCrossing

    return BatchedCollectiveCrossing


def _np(t):
    return None if t is None else t.cpu().numpy()


def _identity_order(g: Golden) -> bool:
    return bool((g["order"] == np.arange(g.N, dtype=np.uint8)).all())


IDENTITY_NPZ = [n for n in STEP_NPZ if _identity_order(Golden(n))]


def test_enough_fixtures_reach_the_step_kernel():
    assert len(IDENTITY_NPZ) >= 20, IDENTITY_NPZ


@pytest.mark.parametrize("name,with_order", [(n, True) for n in STEP_NPZ if not n.startswith("g14_")] + [(n, False) for n in IDENTITY_NPZ])
def test_step_kernel_replays_the_reference_step_by_step(ccx, name, with_order):
    """ccx_step = the short-launch kernel (with the recorded dict order as the move order, or without an order array where the
    recording used the slot order), against every reference-recorded step."""
    g = Golden(name)
    env = ccx(g.config, g.E)
    assert env.step_shape()["ok"] == 1, "these grids fit the step kernel's LDS"
    env.set_state(**g.init_state())
    for s in range(g.K):
        r = env.step(g["actions"][s], g["order"][s] if with_order else None)
        assert_step_matches(g, s, _np(r.obs), _np(r.reward), _np(r.agent_flags), _np(r.env_flags), env.get_state())
    c = env.counters()
    assert c["env_steps"] == g.K * g.E and c["agent_steps"] == g.K * g.E * g.N
    assert c["live_agent_steps"] == int(((g["agent_flags"] & 4) != 0).sum())
    env.close()


def _cfg(name):
    import bench
    if name == "c1":
        return bench.c2_config(max_steps=23)
    if name == "n1":
        from collectivecrossing_amd import configs as C
        return C.CollectiveCrossingConfig(width=6, height=16, division_y=8, tram_door_left=1, tram_door_right=3, tram_length=4,
                                          num_boarding_agents=1, num_exiting_agents=0, exiting_destination_area_y=0,
                                          boarding_destination_area_y=16, truncated_config=C.MaxStepsTruncatedConfig(max_steps=9))
    cfg, _ = bench.workload_config(name)
    return cfg


@pytest.mark.parametrize("cfg_name,E,K,wpb", [
    ("c1", 4096, 1, 0), ("c1", 4096, 16, 0), ("c1", 37, 7, 0), ("c1", 1, 3, 0), ("c1", 1000, 2, 2), ("c1", 5000, 5, 3),
    ("c3", 300, 4, 0), ("c3", 2, 16, 2), ("c5_50", 40, 6, 0), ("c5_64", 33, 9, 0), ("c5_64", 1024, 1, 0), ("n1", 700, 12, 3), ("c5_50", 7, 3, 1), ("c3", 64, 2, 1),
])
def test_short_launches_equal_the_oracle_and_the_rollout_kernel(oracle, ccx, cfg_name, E, K, wpb):
    """Several short launches in a row with auto-reset (the pool cursor, restarts inside a launch, counters) against the
    oracle, then the same through the rollout kernel: three independent implementations, one trajectory."""
    import torch

    from collectivecrossing_amd.params import lower_config
    from collectivecrossing_amd.reset import build_reset_pool
    cfg = _cfg(cfg_name)
    params = lower_config(cfg)
    N = params.num_boarding + params.num_exiting
    rng = np.random.default_rng(E * 131 + K)
    launches = 4
    actions = rng.integers(0, 6, size=(launches, K, E, N), dtype=np.uint8)
    actions[actions == 5] = 255                                   # some agents are absent from the action dict
    # every other case drives the agents in shuffled dict orders (collectivecrossing.py:197)
    orders = np.argsort(rng.random((launches, K, E, N)), axis=-1).astype(np.uint8) if (E + K) % 2 else None
    pool = build_reset_pool(cfg, 99, 61)
    ob = oracle.OracleBatch(params, E)
    ob.set_reset_pool(pool)
    ob.reset_from_pool()
    outs = {}
    for step_kernel in (1, 0):
        env = ccx(cfg, E)
        env.set_tunable("step_kernel", step_kernel)
        if wpb:
            env.set_tunable("step_rows", wpb)
        env.set_reset_pool(pool)
        env.reset_from_pool()
        res = [env.rollout(actions[j], None if orders is None else orders[j], auto_reset=True) for j in range(launches)]
        outs[step_kernel] = ([(_np(r.obs).view(np.uint32), _np(r.reward).view(np.uint64), _np(r.agent_flags), _np(r.env_flags))
                              for r in res], env.get_state(), env.counters())
        if step_kernel:
            shape = env.step_shape()
            assert shape["ok"] == 1 and (not wpb or shape["row_waves"] == wpb)
        env.close()
        torch.cuda.empty_cache()
    for j in range(launches):
        o_obs, o_rew, o_af, o_ef = ob.rollout(actions[j], None if orders is None else orders[j], auto_reset=True)
        for which in (1, 0):
            obs, rew, af, ef = outs[which][0][j]
            tag = f"launch {j} step_kernel={which}"
            np.testing.assert_array_equal(af, o_af, err_msg=tag)
            np.testing.assert_array_equal(ef, o_ef, err_msg=tag)
            np.testing.assert_array_equal(rew, o_rew.view(np.uint64), err_msg=tag)
            np.testing.assert_array_equal(obs, o_obs.view(np.uint32), err_msg=tag)
    for which in (1, 0):
        st, c = outs[which][1], outs[which][2]
        for k in ("x", "y", "active", "terminated", "truncated", "step_count", "episode"):
            np.testing.assert_array_equal(st[k], getattr(ob, k), err_msg=f"{k} step_kernel={which}")
        assert c == ob.counters.as_dict(), which


def test_outputs_are_optional_and_compact_rows_match(ccx):
    """Every output pointer of ccx_step_out may be NULL; the compact rows of the step kernel expand to its own rows."""
    import torch

    import bench
    cfg = bench.c2_config()
    E = 777
    acts = torch.randint(0, 5, (6, E, 8), dtype=torch.uint8, device="cuda")
    ref = ccx(cfg, E)
    ref.make_reset_pool(3, 64)
    ref.reset_from_pool()
    env = ccx(cfg, E)
    env.make_reset_pool(3, 64)
    env.reset_from_pool()
    for s in range(6):
        full = ref.step(acts[s], want_obs=True, want_compact=True)
        part = env.step(acts[s], want_obs=(s % 2 == 0), want_compact=(s % 3 == 0))
        assert torch.equal(full.reward.view(torch.int64), part.reward.view(torch.int64))
        assert torch.equal(full.agent_flags, part.agent_flags) and torch.equal(full.env_flags, part.env_flags)
        if part.obs is not None:
            assert torch.equal(full.obs.view(torch.int32), part.obs.view(torch.int32))
        if part.obs_compact is not None:
            assert torch.equal(full.obs_compact.view(torch.int32), part.obs_compact.view(torch.int32))
        rows = ref.expand_observations(full.obs_compact)
        assert torch.equal(rows.view(torch.int32), full.obs.view(torch.int32))
    ref.close()
    env.close()


def test_grids_whose_tables_exceed_the_lds_fall_back_to_the_rollout_kernel(oracle, ccx):
    """100 x 100: the per-wave cell + occupancy tables do not fit; ccx_step still works (rollout kernel, all-pairs masks)."""
    from collectivecrossing_amd import configs as C
    from collectivecrossing_amd.params import lower_config
    cfg = C.CollectiveCrossingConfig(width=100, height=100, division_y=50, tram_door_left=10, tram_door_right=30, tram_length=60,
                                     num_boarding_agents=6, num_exiting_agents=6, exiting_destination_area_y=0,
                                     boarding_destination_area_y=100, truncated_config=C.MaxStepsTruncatedConfig(max_steps=50))
    E = 9
    env = ccx(cfg, E)
    assert env.step_shape()["ok"] == 0
    ob = oracle.OracleBatch(lower_config(cfg), E)
    seeds = np.arange(E, dtype=np.uint64) + 5
    env.reset(seeds)
    st = env.get_state()
    ob.set_state(**{k: st[k] for k in ("x", "y", "active", "terminated", "truncated", "step_count")})
    rng = np.random.default_rng(1)
    for _ in range(5):
        a = rng.integers(0, 5, size=(E, 12), dtype=np.uint8)
        r = env.step(a)
        o_obs, o_rew, o_af, o_ef = ob.rollout(a[None], auto_reset=False)
        np.testing.assert_array_equal(_np(r.agent_flags), o_af[0])
        np.testing.assert_array_equal(_np(r.obs).view(np.uint32), o_obs[0].view(np.uint32))
        np.testing.assert_array_equal(_np(r.reward).view(np.uint64), o_rew[0].view(np.uint64))
    env.close()


def test_a_graph_of_single_steps_replays_bit_exactly(oracle, ccx):
    """Policy-in-the-loop stepping captured into a HIP graph (what bench.py's secondary.step_k1_graph times): 20 captured
    ccx_step launches, replayed three times, equal the oracle's 60 steps."""
    import torch

    import bench
    from collectivecrossing_amd.params import lower_config
    cfg = bench.c2_config(max_steps=1000)
    E, N, S = 512, 8, 20
    env = ccx(cfg, E)
    seeds = np.arange(E, dtype=np.uint64)
    env.reset(seeds)
    st = env.get_state()
    ob = oracle.OracleBatch(lower_config(cfg), E)
    ob.set_state(**{k: st[k] for k in ("x", "y", "active", "terminated", "truncated", "step_count")})
    acts_h = np.random.default_rng(5).integers(0, 5, size=(S, E, N), dtype=np.uint8)
    acts = torch.from_numpy(acts_h).cuda()
    side = torch.cuda.Stream()
    env.use_stream(side)
    with torch.cuda.stream(side):
        graph = torch.cuda.CUDAGraph()
        last = None
        with torch.cuda.graph(graph, stream=side):
            for s in range(S):
                last = env.step(acts[s])
        for rep in range(3):
            graph.replay()
            side.synchronize()
            for s in range(S):
                o_obs, o_rew, o_af, o_ef = ob.rollout(acts_h[s][None], auto_reset=False)
            np.testing.assert_array_equal(_np(last.agent_flags), o_af[0], err_msg=f"replay {rep}")
            np.testing.assert_array_equal(_np(last.obs).view(np.uint32), o_obs[0].view(np.uint32), err_msg=f"replay {rep}")
            np.testing.assert_array_equal(_np(last.reward).view(np.uint64), o_rew[0].view(np.uint64), err_msg=f"replay {rep}")
    st = env.get_state()
    np.testing.assert_array_equal(st["x"], ob.x)
    np.testing.assert_array_equal(st["step_count"], ob.step_count)
    env.close()


def test_short_launches_of_a_shard_walk_the_pool_by_global_env_index(oracle, ccx):
    """A shard of a larger batch (env_offset / total_envs, collectivecrossing_amd/sharding.py): restarts inside short launches take
    the pool entry of the GLOBAL env index, so trajectories do not depend on the world size -- step kernel vs oracle."""
    import bench
    from collectivecrossing_amd.params import lower_config
    from collectivecrossing_amd.reset import build_reset_pool
    cfg = bench.c2_config(max_steps=7)
    E, off, total, K = 500, 1500, 4000, 5
    pool = build_reset_pool(cfg, 5, 97)
    ob = oracle.OracleBatch(lower_config(cfg), E, env_offset=off, total_envs=total)
    env = ccx(cfg, E, env_offset=off, total_envs=total)
    assert env.step_shape()["ok"] == 1
    for b in (ob, env):
        b.set_reset_pool(pool)
        b.reset_from_pool()
    rng = np.random.default_rng(3)
    for launch in range(6):
        a = rng.integers(0, 5, size=(K, E, 8), dtype=np.uint8)
        res = env.rollout(a, auto_reset=True)
        o_obs, o_rew, o_af, o_ef = ob.rollout(a, auto_reset=True)
        np.testing.assert_array_equal(_np(res.env_flags), o_ef)
        np.testing.assert_array_equal(_np(res.agent_flags), o_af)
        np.testing.assert_array_equal(_np(res.obs).view(np.uint32), o_obs.view(np.uint32))
    st = env.get_state()
    for k in ("x", "y", "episode", "step_count"):
        np.testing.assert_array_equal(st[k], getattr(ob, k), err_msg=k)
    assert env.counters() == ob.counters.as_dict() and ob.counters.episodes > 0
    env.close()
