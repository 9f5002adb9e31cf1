"""bench.py prints ONE JSON line with the driver's contract fields (run as a child process, small)."""

import json
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _run(*args):
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], capture_output=True, text=True, timeout=600)
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
    assert r["traffic"] is None or 0.9 < r["traffic"] / r["bytes_per_launch"] < 1.2
    assert 0.2 < r["frac"] < 1.0 and "cpu_baseline" not in d


def test_bench_cpu_baseline_leg():
    d = _run("--steps", "2", "--warmup", "1")
    b = d["cpu_baseline"]
    assert b["kind"] == "port" and b["unit"] == "env-steps/s" and b["cores"] >= 1 and b["value"] > 1e5
    assert "sample" in b and d["value"] > 20 * b["value"]
