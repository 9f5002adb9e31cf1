"""Fused greedy rollouts on larger grids: default (tables dropped where they cost rounds) vs forced tables vs all-pairs."""
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "profiles/scratch")
import big_grid_scan as b  # noqa: E402  (patches the constructor: b._occ)
import cliff_scan2  # noqa: E402

for (w, h, n) in ((24, 16, 8), (40, 30, 8), (64, 48, 8), (100, 100, 8)):
    c = b.cfg(w, h, n)
    for E in (1024, 4096, 16384):
        res = {}
        for occ in (-1, 1, 0):
            b._occ[0] = occ
            try:
                r = cliff_scan2.other(c, E, n, "greedy")
                res[occ] = round(r["us_per_env_step"], 3)
            except Exception as exc:
                res[occ] = repr(exc)[:70]
        print(f"{w}x{h} N={n} greedy E={E}: default {res[-1]}  tables {res[1]}  all-pairs {res[0]}", flush=True)
