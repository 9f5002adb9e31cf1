"""bench.py prints ONE JSON line with the driver's contract fields (run as a child process, small)."""

import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _run(*args, env=None):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], capture_output=True, text=True, timeout=600, env=e)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_bench_line_has_the_contract_fields_and_adds_up():
    d = _run("--steps", "3", "--warmup", "2", "--no-cpu-baseline")
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str),
                     ("data", str), ("config", dict), ("roofline", dict)):
        assert isinstance(d[key], typ), key
    assert d["vs_baseline"] is None and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 2
    assert d["unit"] == "env-steps/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic"
    c, r = d["config"], d["roofline"]
    assert "workload" in c and "model" not in c
    per_step = c["env_steps_per_step"] * c["envs_per_gpu"]
    assert d["value"] == pytest.approx(per_step / (d["ms_per_step"] * 1e-3), rel=1e-9)
    assert d["counters"]["env_steps"] == 3 * per_step
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-12)
    assert r["achieved"] == pytest.approx(r["bytes_per_launch"] / (r["kernel_ms_per_launch"] * 1e-3) / 1e9, rel=1e-9)
    assert r["bytes_per_launch"] == r["bytes_per_agent_step"] * per_step * c["agents"]
    # PMC traffic cannot be measured inside a plain run: `traffic` stays null, the figure replayed from the
    # tracked rocprofv3 passes is labelled as such and names its file
    assert r["traffic"] is None
    t = r["traffic_from_profiles"]
    assert t is None or (t["file"].startswith("profiles/") and (ROOT / t["file"]).exists()
                         and 0.9 < t["hbm_bytes_per_launch"] / r["bytes_per_launch"] < 1.2)
    assert 0.2 < r["frac"] < 1.0 and "cpu_baseline" not in d
    # exactly --warmup launches precede the `cold` window; `value` is the steady state after the settle launches
    cold = d["cold"]
    assert c["settle_launches"] == 80 - 2 - 3 and cold["value"] > 0 and 0.2 < r["frac_cold"] < 1.0
    assert cold["ratio_to_value"] == pytest.approx(cold["value"] / d["value"], rel=1e-9)
    assert d["rccl_ranks"] == 0 and d["collective_backend"] is None and d["per_rank_env_steps_per_sec"][0] >= d["value"] * 0.999
    sec = d["secondary"]
    assert 0.1 < sec["no_obs"]["us_per_env_step"] < 5 and 1 < sec["step_k1"]["us_per_step"] < 100


def test_bench_cpu_baseline_leg():
    d = _run("--steps", "2", "--warmup", "1")
    b = d["cpu_baseline"]
    assert b["kind"] == "port" and b["unit"] == "env-steps/s" and b["cores"] >= 1 and b["value"] > 1e5
    assert "sample" in b and b["value"] > 0


def test_bare_command_with_two_ranks_sharing_the_gpu():
    """`python bench.py --gpus 2` without torch.distributed.run: the launcher path on the GPU box.  Two
    ranks on ONE GPU cannot form an RCCL group (one device per rank), so the collectives run over gloo
    here, explicitly (CCX_DIST_BACKEND); the env shards, offsets and the counter reduction are the real ones."""
    d = _run("--gpus", "2", "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--envs-per-gpu", "1024",
             env={"CCX_DIST_BACKEND": "gloo"})
    c = d["config"]
    assert d["n_gpus"] == 2 and d["launcher"] == "bench.py" and d["collective_backend"] == "gloo" and d["rccl_ranks"] == 0
    assert c["global_envs"] == 2048 and c["envs_per_gpu"] == 1024
    assert d["counters"]["env_steps"] == 3 * c["env_steps_per_step"] * 2048          # summed over both ranks
    assert len(d["per_rank_env_steps_per_sec"]) == 2 and d["cpu_baseline"] is None and "secondary" not in d
    assert d["value"] <= sum(d["per_rank_env_steps_per_sec"]) * 1.0001


def test_rccl_group_of_one_rank_per_gpu_only():
    """With backend nccl (the default on a GPU box) two ranks on one GPU are refused loudly -- never
    downgraded to another backend."""
    e = dict(os.environ)
    e.pop("CCX_DIST_BACKEND", None)
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=e)
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box can really run two RCCL ranks")
    assert p.returncode != 0 and "one GPU per" in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
