"""C2 batches of 1000-3000 envs (the unpaced / just-paced regime): default shape vs lanes x writers candidates, settled."""
import sys
import time

sys.path.insert(0, ".")
sys.path.insert(0, "profiles/scratch")
import ragged  # noqa: E402


def cand(lanes, w):
    return lambda e: (e.set_launch_shape(lanes, 1), e.set_writers(w))


if __name__ == "__main__":
    t0 = time.time()
    for E in (1024, 1536, 2048, 2112, 2176, 2240, 2304, 2368, 2432, 2500, 2816):
        row = {}
        for name, prep in (("default", None), ("64w2", cand(64, 2)), ("64w3", cand(64, 3)), ("64w4", cand(64, 4)),
                           ("32w2", cand(32, 2)), ("32w3", cand(32, 3)), ("32w4", cand(32, 4))):
            try:
                row[name] = ragged.measure("c2", E, prep)
            except Exception as exc:
                row[name] = {"error": repr(exc)[:80]}
        d = row["default"]
        print(f"[{time.time() - t0:4.0f}s] E={E:5d}: default {d.get('frac', 0):.3f} {d.get('shape')} pace {d.get('pace_ns', 0):.0f} | " +
              " ".join(f"{k} {v['frac']:.3f}" if "frac" in v else f"{k} err" for k, v in row.items() if k != "default"), flush=True)
