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
    # measurement hygiene: nothing outside the process seeds the pace controller -- its start value is
    # measured in-process -- the line says so, carries the wall-clock fraction next to the HIP-event one and names what
    # `value` is
    assert c["pace_start_source"] == "calibration" and c["pace_start_ns"] > 0 and 3000 < c["pace_probe_GBs"] < 9000
    assert cold["pace_start_source"] == "calibration" and d["cold_unseeded"]["value"] == cold["value"]
    assert "steady state" in d["value_protocol"]
    assert d["retimed"] is None or d["retimed"]["first_wall_over_kernel_time"] > 1.15      # (only after a host stall)
    assert r["frac_wall"] == pytest.approx(r["bytes_per_launch"] / (d["ms_per_step"] * 1e-3) / 1e9 / 8000.0, rel=1e-9)
    assert r["frac_wall"] <= r["frac"] * 1.001 and cold["frac_wall"] <= cold["frac"] * 1.001
    assert d["rccl_ranks"] == 0 and d["collective_backend"] is None and d["per_rank_env_steps_per_sec"][0] >= d["value"] * 0.999
    sec = d["secondary"]
    assert 0.1 < sec["no_obs"]["us_per_env_step"] < 5 and 1 < sec["step_k1"]["us_per_step"] < 100
    assert 1 < sec["step_k1_graph"]["us_per_step"] < 100          # (the same launches without the host in the loop)
    assert sec["no_obs"]["us_per_env_step"] * 0.8 < sec["compact_obs"]["us_per_env_step"] < 5
    # round 4: every workload / batch size / duration the builder quotes is in the driver's line
    wl = {(w["workload"], w["envs"]): w for w in sec["workloads"]}
    assert set(wl) == {("c3", 4096), ("c5_50", 1024), ("c5_64", 1024), ("c2", 1024), ("c2", 2048), ("c2", 16384), ("c2", 32768), ("c2", 65536),
                       ("c2", 10000), ("c2", 20000)}
    assert wl[("c2", 10000)]["frac"] > 0.7 and wl[("c2", 20000)]["frac"] > 0.7      # (round 4 found them at 0.49 / 0.67)
    for key, w in wl.items():
        assert "error" not in w, w
        assert 0.2 < w["frac_wall"] <= w["frac"] * 1.001 < 1.0 and w["timed_launches"] == 10 and w["kernel_ms_per_launch"] > 0, key
        assert w["frac"] == pytest.approx(w["bytes_per_launch"] / (w["kernel_ms_per_launch"] * 1e-3) / 1e9 / 8000.0, rel=1e-9)
    assert wl[("c3", 4096)]["agents"] == 32 and wl[("c5_64", 1024)]["policy"] == "greedy" and wl[("c5_64", 1024)]["agents"] == 64
    su = sec["sustained"]
    assert su["seconds"] >= 2.0 and su["launches"] >= 1000 and 0.2 < su["frac_wall"] <= su["frac_kernel_mean"] * 1.001 < 1.0
    assert len(su["buckets"]) >= 8 and all(b["ms_min"] <= b["ms_median"] <= b["ms_max"] for b in su["buckets"])
    sh = sec["short_launches"]
    assert {"k1", "k2", "k4", "k8", "k16", "k32", "k64"} <= set(sh) and "error" not in sh
    assert sh["k1"]["us_per_launch"] < sh["k16"]["us_per_launch"] < sh["k64"]["us_per_launch"] and sh["k16"]["frac"] > sh["k1"]["frac"]
    # the short-launch kernel is what serves K = 1 (ccx_step.hip): device-side latency of a captured step
    assert sh["k1"]["us_per_launch"] < 5.5 and sec["step_k1_graph"]["us_per_step"] < 5.5


def test_bench_with_a_pace_cache_says_so(tmp_path):
    """Opt-in pace memory (CCX_PACE_CACHE names the file; nothing is ever read from or written to $HOME): the first run
    leaves the learned pace there, the second starts from it and labels its numbers accordingly."""
    cache = tmp_path / "pace.json"
    d1 = _run("--steps", "20", "--warmup", "25", "--no-cpu-baseline", "--no-secondary", env={"CCX_PACE_CACHE": str(cache)})
    assert d1["config"]["pace_start_source"] == "calibration" and cache.exists()
    key, entry = next(iter(json.loads(cache.read_text()).items()))
    assert "E4096" in key and entry["pace_ns"] > 0
    d2 = _run("--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--no-secondary", env={"CCX_PACE_CACHE": str(cache)})
    assert d2["config"]["pace_start_source"] == "user_cache" and d2["cold_unseeded"] is None
    assert d2["config"]["pace_start_ns"] == pytest.approx(entry["pace_ns"] * 1.01, rel=1e-3)


def test_bench_cpu_baseline_leg():
    d = _run("--steps", "2", "--warmup", "1")
    b = d["cpu_baseline"]
    assert b["kind"] == "port" and b["unit"] == "env-steps/s" and b["cores"] >= 1 and b["value"] > 1e5
    assert "sample" in b and b["value"] > 0
    # (round 4) the leg says how many cores the box has, how many this process may use and how many threads it ran
    assert b["threads_used"] == b["cores"] <= b["cores_of_the_box"] and b["cores_in_affinity_mask"] >= b["cores"]
    assert b["cgroup_cpu_quota"] is None or b["cgroup_cpu_quota"] > 0


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


@pytest.mark.timeout(1000)
@pytest.mark.parametrize("direct", [False, True], ids=["torch_distributed", "direct_rccl"])
def test_two_rccl_ranks_when_the_box_has_two_gpus(direct):
    """Self-activating evidence of the N > 1 path: on a box with >= 2 GPUs the BARE command
    `bench.py --gpus 2` spawns two ranks that form a real RCCL group -- through torch.distributed and through the
    C-ABI's own ncclAllReduce (`ccx_rccl_allreduce_counters`) -- the 48-byte counter reduction sums both shards and
    the two ranks run at the same rate.  One-GPU boxes skip (the refusal test below covers them).  The reference's
    only parallelism is one env per EnvRunner process (examples/training_script.py:84); one shard per GPU is the
    equivalent."""
    import torch
    if torch.cuda.device_count() < 2:            # (counting devices does not initialise the GPU)
        pytest.skip("needs >= 2 GPUs: RCCL refuses two ranks on one device")
    e = {k: v for k, v in os.environ.items()
         if k not in ("CCX_DIST_BACKEND", "RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
                        "--no-cpu-baseline", *(["--direct-rccl"] if direct else [])],
                       capture_output=True, text=True, timeout=900, env=e)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    c = d["config"]
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["collective_backend"] == "nccl" and d["launcher"] == "bench.py"
    assert c["global_envs"] == 2 * c["envs_per_gpu"]
    assert d["counters"]["env_steps"] == 3 * c["env_steps_per_step"] * c["global_envs"]      # summed over both ranks
    assert d["counters"]["agent_steps"] == d["counters"]["env_steps"] * c["agents"]
    if direct:
        assert d["counters_allreduce"] == "ccx_rccl_allreduce_counters (2 RCCL rank(s))"
    else:
        assert d["counters_allreduce"] == "torch.distributed"
    rates = d["per_rank_env_steps_per_sec"]
    assert len(rates) == 2 and min(rates) > 0.85 * max(rates), rates                          # weak scaling: same work, same rate
    assert d["value"] <= sum(rates) * 1.0001 and d["value"] > 1.5 * min(rates) * 0.85
    assert d["cpu_baseline"] is None and "secondary" not in d


def test_rccl_group_of_one_rank_per_gpu_only():
    """With backend nccl (the default on a GPU box) two ranks on one GPU are refused loudly -- never
    downgraded to another backend."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box can really run two RCCL ranks (test_two_rccl_ranks_when_the_box_has_two_gpus)")
    e = dict(os.environ)
    e.pop("CCX_DIST_BACKEND", None)
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=e)
    assert p.returncode != 0 and "one GPU per" in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
