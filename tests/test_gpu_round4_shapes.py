"""GPU parity of the launch shapes round 4 added late: each case runs the default shape of a configuration whose
shape comes from one of the new rules and compares the whole trajectory, the state and the counters with the oracle, bit for bit
(collectivecrossing.py:161-261 through ccx_rollout):

  * LDS occupancy tables given up where they cost rounds (all-pairs instantiations, full tiles): 24 x 16 and 64 x 48 grids,
    8 and 16 agents, ragged batch sizes; forced either way through the tunable as well;
  * grids whose CELL table limits the workgroups per CU (100 x 100): one writer, four tiles per workgroup;
  * slabs that are not a whole number of 128-byte lines with long rows (50 agents, batch size off its multiple of 8): the
    per-step row layout (OUTM 3), with and without a move order (the instantiations that are not PLAIN), fused greedy too;
  * tile pairs that give way to singles (4 agents, 17 777 envs), the two-writer class with one writer in pairs (20 agents,
    8200 tiles), single-agent envs without tables (in test_gpu_round2.py).
"""
from types import SimpleNamespace

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ccx():
    import torch

    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    from collectivecrossing_amd.batched import BatchedCollectiveCrossing

    return BatchedCollectiveCrossing


def _cfg(w, h, n, max_steps=60):
    from collectivecrossing_amd import configs as C
    nb = (n + 1) // 2
    if (w, h) == (12, 8):
        return C.CollectiveCrossingConfig(width=12, height=8, division_y=4, tram_door_left=5, tram_door_right=7, tram_length=9,
                                          num_boarding_agents=nb, num_exiting_agents=n - nb, exiting_destination_area_y=0,
                                          boarding_destination_area_y=8, truncated_config=C.MaxStepsTruncatedConfig(max_steps=max_steps))
    return C.CollectiveCrossingConfig(width=w, height=h, division_y=h // 2, tram_door_left=w // 2 - 2, tram_door_right=w // 2 + 2,
                                      tram_length=w - 4, num_boarding_agents=nb, num_exiting_agents=n - nb, exiting_destination_area_y=0,
                                      boarding_destination_area_y=h, truncated_config=C.MaxStepsTruncatedConfig(max_steps=max_steps))


def _case(oracle, ccx, cfg, E, K, seed, order=False, setup=None):
    from test_gpu_round2 import _against_oracle

    from collectivecrossing_amd.params import lower_config
    g = SimpleNamespace(config=cfg, params=lower_config(cfg), N=cfg.num_boarding_agents + cfg.num_exiting_agents)
    return _against_oracle(oracle, ccx, g, E=E, K=K, seed=seed, order=order, setup=setup)


@pytest.mark.parametrize("w,h,n,E,K", [(24, 16, 8, 9001, 24), (64, 48, 8, 9001, 16), (64, 48, 16, 4501, 12), (40, 30, 3, 9001, 24)])
def test_tables_given_up_for_residency_equal_the_oracle(oracle, ccx, w, h, n, E, K):
    c, shape, _, _ = _case(oracle, ccx, _cfg(w, h, n), E, K, seed=101 + n)
    # (with tables these tiles would have been halved until the tables fit: 16 lanes on 64 x 48, 32 on 40 x 30)
    assert shape["lanes_per_wave"] == 64, shape
    # forced either way: the same trajectory (the counters are part of the comparison)
    for occ in (0, 1):
        c2, _, _, _ = _case(oracle, ccx, _cfg(w, h, n), E // 3, K, seed=101 + n, setup=lambda env: env.set_tunable("occ_tables", occ))


def test_a_cell_table_that_fills_the_lds_takes_fuller_workgroups(oracle, ccx):
    c, shape, _, _ = _case(oracle, ccx, _cfg(100, 100, 8), 5001, 16, seed=111)
    assert shape["waves_per_block"] >= 2, shape                      # (one-tile workgroups: 256 tiles at a time)
    _case(oracle, ccx, _cfg(100, 100, 8), 601, 16, seed=112, order=True)


@pytest.mark.parametrize("order", [False, True])
def test_misaligned_slabs_of_long_rows_take_the_per_step_layout_and_equal_the_oracle(oracle, ccx, order):
    import bench
    cfg = bench.workload_config("c5_50")[0]
    for E in (1004, 131):                                             # (50 agents: a multiple of 8 is aligned; 131: one partial round of tiny batch)
        env_probe = ccx(cfg, E)
        assert E % env_probe.rows_alignment() != 0
        env_probe.close()
        _case(oracle, ccx, cfg, E, 14, seed=120 + E, order=order)


def test_misaligned_fused_greedy_rollout_equals_the_oracle(oracle, ccx):
    import bench

    from collectivecrossing_amd.params import lower_config
    from collectivecrossing_amd.reset import build_reset_pool
    cfg = bench.workload_config("c5_50")[0]
    E, K = 516, 12
    p = lower_config(cfg)
    pool = build_reset_pool(cfg, 77, 64)
    ob, env = oracle.OracleBatch(p, E), ccx(cfg, E)
    try:
        for b in (ob, env):
            b.set_reset_pool(pool)
            b.reset_from_pool()
        res, acts = env.rollout_greedy(K, auto_reset=True)
        o_act, o_obs, o_rew, o_af, o_ef = ob.rollout_greedy(K, auto_reset=True)
        np.testing.assert_array_equal(acts.cpu().numpy(), o_act)
        np.testing.assert_array_equal(res.obs.cpu().numpy().view(np.uint32), o_obs.view(np.uint32))
        np.testing.assert_array_equal(res.reward.cpu().numpy().view(np.uint64), o_rew.view(np.uint64))
        np.testing.assert_array_equal(res.agent_flags.cpu().numpy(), o_af)
        np.testing.assert_array_equal(res.env_flags.cpu().numpy(), o_ef)
    finally:
        env.close()


@pytest.mark.parametrize("w,h,n,E,K", [(12, 8, 4, 17777, 12), (24, 16, 20, 16401, 8)])
def test_singles_and_one_writer_pairs_equal_the_oracle(oracle, ccx, w, h, n, E, K):
    c, shape, _, _ = _case(oracle, ccx, _cfg(w, h, n), E, K, seed=130 + n)
    if n == 4:
        assert shape["waves_per_block"] == 1 and shape["writers_per_tile"] == 3, shape
    else:
        assert shape["waves_per_block"] == 2 and shape["writers_per_tile"] == 1, shape
