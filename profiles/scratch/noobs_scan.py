"""C2 without observation rows around 512 tiles: us per env-step by batch size and writer count (tiles per workgroup 1)."""
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "profiles/scratch")
import cliff_scan  # noqa: E402
import shape_sweep  # noqa: E402
from collectivecrossing_amd.batched import BatchedCollectiveCrossing  # noqa: E402

_orig = BatchedCollectiveCrossing.__init__
_prep = [None]


def _init(self, *a, **k):
    _orig(self, *a, **k)
    if _prep[0]:
        _prep[0](self)


BatchedCollectiveCrossing.__init__ = _init
NAG = int(sys.argv[2]) if len(sys.argv) > 2 else 8
import cliff_scan2  # noqa: E402
cfg = shape_sweep.config_for(NAG) if NAG in (1, 3, 8, 12, 32, 50, 64) else cliff_scan2.config_for(NAG)
MODE = sys.argv[3] if len(sys.argv) > 3 else "noobs"
for E in [int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else (3072, 3424, 3600, 3840, 4000, 4096, 4104, 4200, 4328, 4864, 6160, 8192):
    res = {}
    for name, prep in (("default", None), ("w1", lambda e: (e.set_writers(1), e.set_launch_shape(0, 1))), ("w2", lambda e: (e.set_writers(2), e.set_launch_shape(0, 1))),
                       ("w1t2", lambda e: (e.set_writers(1), e.set_launch_shape(0, 2))), ("w2t2", lambda e: (e.set_writers(2), e.set_launch_shape(0, 2))), ("w3", lambda e: (e.set_writers(3), e.set_launch_shape(0, 1))),
                       ("w4", lambda e: (e.set_writers(4), e.set_launch_shape(0, 1)))):
        _prep[0] = prep
        for mode in (MODE,):
            r = cliff_scan.measure(cfg, E, NAG, mode)
            res[name] = round(r["us_per_env_step"], 3)
    print(f"N={NAG} {MODE} E={E}: " + " ".join(f"{k} {v}" for k, v in res.items()), flush=True)
