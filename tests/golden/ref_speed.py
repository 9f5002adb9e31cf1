#!/usr/bin/env python3
"""Time the reference's own ``CollectiveCrossingEnv.step`` (pure Python) in the BUILD container and
record the number as data (profiles/r01_reference_python_speed.json).  The reference cannot travel
to the GPU box, so ``bench.py`` quotes this recorded figure next to the C-port baseline it times
live.  Protocol of SURVEY 8d: BASELINE config 1 (12x8, 5+3 agents), uniform random actions for the
agents, auto-reset with seed + episode, warm-up 200 steps, median of 3 repeats, one process =
one core; only the time inside ``env.step`` counts.  No-op when /root/reference is absent.

usage: python tests/golden/ref_speed.py
"""
import json
import os
import platform
import statistics
import sys
import time
from pathlib import Path

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import gen_golden as G  # noqa: E402


def main() -> int:
    if not (G.REF / "src" / "collectivecrossing").is_dir():
        print("reference not found: nothing to do")
        return 0
    G.import_reference()
    import numpy as np
    from collectivecrossing import CollectiveCrossingEnv

    out = {}
    for name, cfg in (("C1_12x8_5+3", G.cfg_c1()), ("C3_20x12_16+16", G.cfg_c3())):
        env = CollectiveCrossingEnv(config=G.build_ref_config(cfg))
        rng = np.random.default_rng(0)
        rates = []
        ids = G.ids_of(cfg)
        pre = [{a: int(rng.integers(0, 5)) for a in ids} for _ in range(256)]   # action generation excluded
        for rep in range(3):
            env.reset(seed=rep)
            episode, steps, n = 0, 0, 200 + (2000 if name.startswith("C1") else 300)
            spent = 0.0
            while steps < n:
                acts = pre[steps & 255]
                t0 = time.perf_counter()
                _, _, term, trunc, _ = env.step(acts)
                if steps >= 200:
                    spent += time.perf_counter() - t0
                steps += 1
                if term["__all__"] or trunc["__all__"]:
                    episode += 1
                    env.reset(seed=rep + episode)                                # resets excluded
            rates.append((n - 200) / spent)
        out[name] = {"env_steps_per_sec_one_core": statistics.median(rates), "repeats": rates}
        print(name, out[name])
    out["host"] = {"cpu": platform.processor() or platform.machine(), "cores_used": 1,
                   "python": platform.python_version(), "nproc": os.cpu_count()}
    out["note"] = ("pure-Python reference timed by tests/golden/ref_speed.py in the build container: "
                   "time inside env.step only (action generation and resets excluded)")
    (HERE.parent.parent / "profiles" / "r01_reference_python_speed.json").write_text(json.dumps(out, indent=1) + "\n")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
