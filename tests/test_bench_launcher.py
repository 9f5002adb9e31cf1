"""`python bench.py --gpus N` as the BARE command (no torch.distributed.run): the process becomes a
launcher that starts N rank processes before anything touches a GPU, relays rank 0's JSON line and
returns the worst exit code.  Rehearsed here on CPU with gloo through `--rehearse` (the N>1 plumbing
without env stepping: rendezvous, barriers, the bench's three collectives with known per-rank values).
The reference's only parallelism is one env per EnvRunner process (examples/training_script.py:84);
one shard per rank is the equivalent."""

import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _bench(*args, env=None, timeout=300):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], capture_output=True, text=True,
                          timeout=timeout, env=e)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_bare_command_spawns_its_ranks_and_reduces_over_gloo(world):
    p = _bench("--gpus", str(world), "--rehearse", env={"CCX_DIST_BACKEND": "gloo"})
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout                      # only rank 0 prints
    d = json.loads(lines[0])
    assert d["rehearsal"] is True and d["value"] is None   # can never be mistaken for a measurement
    assert d["n_gpus"] == world and d["launcher"] == "bench.py"
    assert d["collective_backend"] == "gloo" and d["rccl_ranks"] == 0
    tri = world * (world + 1) // 2                         # rank r contributes (r + 1) * (q + 1)
    assert list(d["counters"].values()) == [tri * (q + 1) for q in range(6)]
    assert d["per_rank"] == [float(r) for r in range(world)]
    assert world - 1 <= d["elapsed_max"] < world - 1 + 60  # MAX over ranks of r + epsilon


@pytest.mark.timeout(300)
def test_a_failing_rank_fails_the_launcher():
    # no GPU in this container: every rank refuses to run the hot path (exit 3) instead of falling back
    # to anything on the CPU; the launcher reports it
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without GPU")
    p = _bench("--gpus", "2", "--steps", "1", "--warmup", "0", env={"CCX_DIST_BACKEND": "gloo"})
    assert p.returncode == 3, (p.returncode, p.stderr[-1500:])
    assert "no CPU implementation" in p.stderr and not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_world_size_mismatch_is_rejected():
    p = _bench("--gpus", "2", "--rehearse", env={"WORLD_SIZE": "1", "RANK": "0"})
    assert p.returncode == 2 and "WORLD_SIZE=1" in p.stderr


def test_rccl_is_never_silently_replaced():
    """sharding.init_from_env has no nccl -> gloo fallback: on a box without GPU, backend nccl raises."""
    code = ("import os, sys; sys.path.insert(0, %r); os.environ.update(RANK='0', WORLD_SIZE='2', MASTER_PORT='29877');"
            "from collectivecrossing_amd import sharding; sharding.init_from_env(backend='nccl')" % str(ROOT))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "needs a GPU" in p.stderr


@pytest.mark.timeout(300)
def test_the_drivers_own_launcher_reaches_the_same_code():
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N` (how the driver starts
    N > 1): bench.py finds RANK / WORLD_SIZE in the environment, does not spawn anything itself, and rank 0 alone
    prints the line."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    e = dict(os.environ, CCX_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(ROOT / "bench.py"),
                        "--gpus", "2", "--rehearse"], capture_output=True, text=True, timeout=280, env=e, cwd=str(ROOT))
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["launcher"] == "torch.distributed.run" and d["n_gpus"] == 2 and d["per_rank"] == [0.0, 1.0]
    assert list(d["counters"].values()) == [3 * (q + 1) for q in range(6)]
