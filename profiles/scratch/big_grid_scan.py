"""Larger grids (the LDS occupancy tables limit how many tiles are resident): default vs all-pairs (occ_tables = 0), rows and no rows."""
import sys
import time

sys.path.insert(0, ".")
sys.path.insert(0, "profiles/scratch")
import cliff_scan  # noqa: E402
from collectivecrossing_amd import configs as C  # noqa: E402
from collectivecrossing_amd.batched import BatchedCollectiveCrossing  # noqa: E402

_orig = BatchedCollectiveCrossing.__init__
_occ = [-1]


def _init(self, *a, **k):
    _orig(self, *a, **k)
    if _occ[0] >= 0:
        self.set_tunable("occ_tables", _occ[0])


BatchedCollectiveCrossing.__init__ = _init


def cfg(w, h, n):
    nb = n // 2
    return C.CollectiveCrossingConfig(width=w, height=h, division_y=h // 2, tram_door_left=w // 2 - 2, tram_door_right=w // 2 + 2,
                                      tram_length=w - 4, num_boarding_agents=nb, num_exiting_agents=n - nb, exiting_destination_area_y=0,
                                      boarding_destination_area_y=h, truncated_config=C.MaxStepsTruncatedConfig(max_steps=200))


if __name__ == "__main__":
    t0 = time.time()
    for (w, h, n) in ((24, 16, 8), (24, 16, 4), (40, 30, 8), (40, 30, 3), (64, 48, 8), (100, 100, 8)):
        c = cfg(w, h, n)
        for mode in ("rows", "noobs"):
            for E in (1024, 4096, 8192, 16384, 32768):
                res = {}
                for occ in (-1, 1, 0):
                    _occ[0] = occ
                    try:
                        r = cliff_scan.measure(c, E, n, mode)
                        res[occ] = (round(r["us_per_env_step"], 3), round(r["frac"], 3) if r["frac"] else None, tuple(r["shape"]))
                    except Exception as exc:
                        res[occ] = repr(exc)[:80]
                print(f"[{time.time() - t0:4.0f}s] {w}x{h} N={n} {mode} E={E}: default {res[-1]}  tables {res[1]}  all-pairs {res[0]}", flush=True)
