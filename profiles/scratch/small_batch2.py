"""C2 batches of 2100-4096 envs, paced: barrier per step (default) vs the sequence-word ring (hand2 = 2), two shapes."""
import sys
import time

sys.path.insert(0, ".")
sys.path.insert(0, "profiles/scratch")
import ragged  # noqa: E402


def cand(lanes, w, hand2):
    def prep(e):
        if lanes:
            e.set_launch_shape(lanes, 1)
            e.set_writers(w)
        e.set_tunable("hand2", hand2)
    return prep


if __name__ == "__main__":
    t0 = time.time()
    for E in (2048, 2176, 2304, 2500, 2816, 3072, 3584, 4096, 5000, 8192):
        row = {}
        for name, prep in (("default", None), ("ring", cand(0, 0, 2)), ("32w2", cand(32, 2, 1)), ("32w2_ring", cand(32, 2, 2)),
                           ("64w3_ring", cand(64, 3, 2))):
            try:
                row[name] = ragged.measure("c2", E, prep)
            except Exception as exc:
                row[name] = {"error": repr(exc)[:80]}
        d = row["default"]
        print(f"[{time.time() - t0:4.0f}s] E={E:5d}: default {d.get('frac', 0):.3f} {d.get('shape')} pace {d.get('pace_ns', 0):.0f} | " +
              " ".join((f"{k} {v['frac']:.3f} (pace {v['pace_ns']:.0f})" if "frac" in v else f"{k} err") for k, v in row.items() if k != "default"), flush=True)
