"""Grids whose CELL table alone fills the LDS (100 x 100: 85 KB): tiles per workgroup x writers."""
import sys
import time

sys.path.insert(0, ".")
sys.path.insert(0, "profiles/scratch")
import big_grid_scan as b  # noqa: E402
import cliff_scan  # noqa: E402
from collectivecrossing_amd.batched import BatchedCollectiveCrossing  # noqa: E402

_prep = [None]
_orig2 = BatchedCollectiveCrossing.__init__


def _init(self, *a, **k):
    _orig2(self, *a, **k)
    if _prep[0]:
        _prep[0](self)


BatchedCollectiveCrossing.__init__ = _init
GRIDS = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]] or [(100, 100, 8), (80, 60, 8), (100, 100, 20)]
for (w, h, n) in GRIDS:
    c = b.cfg(w, h, n)
    for mode in ("rows", "noobs"):
        for E in (4096, 16384, 32768):
            res = {}
            for name, wr, t in (("default", 0, 0), ("w1t2", 1, 2), ("w1t4", 1, 4), ("w2t2", 2, 2), ("w3t2", 3, 2), ("w3t1", 3, 1)):
                _prep[0] = (lambda e, wr=wr, t=t: (e.set_writers(wr), e.set_launch_shape(0, t))) if wr else None
                try:
                    r = cliff_scan.measure(c, E, n, mode)
                    res[name] = (round(r["us_per_env_step"], 3), round(r["frac"], 3) if r["frac"] else None, tuple(r["shape"][1:]))
                except Exception as exc:
                    res[name] = repr(exc)[:60]
            print(f"{w}x{h} N={n} {mode} E={E}: " + "  ".join(f"{k} {v}" for k, v in res.items()), flush=True)
