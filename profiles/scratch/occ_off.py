"""Small lane groups (1-4 agents per env): LDS occupancy tables vs all-pairs compares in the rollout kernel, by batch size."""
import sys
import time

sys.path.insert(0, ".")
sys.path.insert(0, "profiles/scratch")
import cliff_scan  # noqa: E402
import shape_sweep  # noqa: E402
from collectivecrossing_amd.batched import BatchedCollectiveCrossing  # noqa: E402

_orig = BatchedCollectiveCrossing.__init__
_occ = [-1]


def _init(self, *a, **k):
    _orig(self, *a, **k)
    if _occ[0] >= 0:
        self.set_tunable("occ_tables", _occ[0])


BatchedCollectiveCrossing.__init__ = _init

if __name__ == "__main__":
    t0 = time.time()
    for N in (1, 3, 8):
        cfg = shape_sweep.config_for(N)
        for mode in ("rows", "noobs"):
            for E in (1024, 4096, 8192, 14040, 17768, 32768, 65536):
                res = {}
                for occ in (-1, 0):
                    _occ[0] = occ
                    try:
                        r = cliff_scan.measure(cfg, E, N, mode)
                        res[occ] = (round(r["us_per_env_step"], 3), tuple(r["shape"]))
                    except Exception as exc:
                        res[occ] = repr(exc)[:80]
                print(f"[{time.time() - t0:4.0f}s] N={N} {mode} E={E}: tables {res[-1]}  all-pairs {res[0]}", flush=True)
