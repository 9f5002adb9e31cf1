"""C2 batches of several rounds: balanced rounds as launches (default where the last round would waste > 6 %) vs one launch."""
import sys
import time

sys.path.insert(0, ".")
sys.path.insert(0, "profiles/scratch")
import ragged  # noqa: E402

if __name__ == "__main__":
    t0 = time.time()
    wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
    Es = (17000, 17768, 19000, 20000, 22496, 24000, 28464, 32768, 36032, 40000, 45600, 51296, 57712, 65536, 100003) if wl == "c2" else (4500, 5000, 6001, 7000, 8192, 10000)
    for E in Es:
        row = {}
        for name, prep in (("default", None), ("one_launch", lambda e: e.set_tunable("round_launches", 0)),
                           ("by_rounds", lambda e: e.set_tunable("round_launches", 2))):
            try:
                row[name] = ragged.measure(wl, E, prep)
            except Exception as exc:
                row[name] = {"error": repr(exc)[:80]}
        d = row["default"]
        print(f"[{time.time() - t0:4.0f}s] {wl} E={E:6d} K={d.get('K')}: default {d.get('frac', 0):.3f} {d.get('shape')} pace {d.get('pace_ns', 0):.0f} | " +
              " ".join((f"{k} {v['frac']:.3f} (pace {v['pace_ns']:.0f})" if "frac" in v else f"{k} err") for k, v in row.items() if k != "default"), flush=True)
