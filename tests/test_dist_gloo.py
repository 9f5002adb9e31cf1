"""The N>1 path on CPU: 2 ranks, gloo.  Each rank owns a contiguous env shard (sharding.shard_range),
steps it (with the CPU oracle standing in for the GPU so the test runs here) and the counters are
all-reduced exactly like bench.py does over RCCL.  Properties checked:
  * the union of the shards reproduces the single-process trajectory bit for bit (a global env's
    reset-pool cursor depends on its global index only, never on the world size);
  * all_reduce(SUM) of the 6 counters equals the single-process counters; allreduce_max works."""

import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, out_dir: str) -> None:
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from _fixtures import Golden

    from collectivecrossing_amd import sharding
    from collectivecrossing_amd.reset import build_reset_pool
    from oracle import oracle

    r, w, _ = sharding.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    g = Golden("g8_rollout_c1")
    total, K = 37, 70                       # uneven split on purpose
    off, n = sharding.shard_range(total, world, rank)
    pool = build_reset_pool(g.config, 900, 53)
    actions = np.random.default_rng(5).integers(0, 5, size=(K, total, g.N), dtype=np.uint8)
    b = oracle.OracleBatch(g.params, n, env_offset=off, total_envs=total)
    b.set_reset_pool(pool)
    b.reset_from_pool()
    obs, rew, af, ef = b.rollout(np.ascontiguousarray(actions[:, off:off + n]), auto_reset=True)
    sharding.barrier()
    summed = sharding.allreduce_counters(b.counters.as_dict())
    tmax = sharding.allreduce_max(float(rank + 1))
    np.savez(f"{out_dir}/rank{rank}.npz", obs=obs, rew=rew, af=af, ef=ef, off=off, n=n,
             summed=np.array([summed[k] for k in sorted(summed)]), tmax=tmax)
    import torch.distributed as dist

    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_sharding_is_invisible_and_counters_reduce(tmp_path, oracle):
    from _fixtures import Golden

    from collectivecrossing_amd.reset import build_reset_pool

    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    g = Golden("g8_rollout_c1")
    total, K = 37, 70
    pool = build_reset_pool(g.config, 900, 53)
    actions = np.random.default_rng(5).integers(0, 5, size=(K, total, g.N), dtype=np.uint8)
    b = oracle.OracleBatch(g.params, total)
    b.set_reset_pool(pool)
    b.reset_from_pool()
    obs, rew, af, ef = b.rollout(actions, auto_reset=True)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    assert sum(int(p["n"]) for p in parts) == total
    np.testing.assert_array_equal(np.concatenate([p["obs"] for p in parts], axis=1), obs)
    np.testing.assert_array_equal(np.concatenate([p["af"] for p in parts], axis=1), af)
    np.testing.assert_array_equal(np.concatenate([p["ef"] for p in parts], axis=1), ef)
    np.testing.assert_array_equal(np.concatenate([p["rew"] for p in parts], axis=1).view(np.uint64), rew.view(np.uint64))
    single = b.counters.as_dict()
    for p in parts:
        assert p["summed"].tolist() == [single[k] for k in sorted(single)]
        assert float(p["tmax"]) == 2.0
    assert single["episodes"] > 0
