import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import os
if os.environ.get("CCX_DIAG_LIB"):
    from collectivecrossing_amd import _lib
    _lib.LIB_PATH = Path(os.environ["CCX_DIAG_LIB"]).resolve()
from collectivecrossing_amd import CollectiveCrossingEnv as Env
from collectivecrossing_amd.configs import CollectiveCrossingConfig, MaxStepsTruncatedConfig
for cur, mx in ((10, 10), (100000, 100000), (70000, 100000), (65536, 65536), (65535, 65535), (40000, 40000), (32768, 32768)):
    d = dict(width=10, height=8, division_y=4, tram_door_left=4, tram_door_right=5, tram_length=8,
             num_boarding_agents=1, num_exiting_agents=1, exiting_destination_area_y=1,
             boarding_destination_area_y=7, truncated_config=MaxStepsTruncatedConfig(max_steps=mx))
    env = Env(config=CollectiveCrossingConfig(**d))
    env.reset(seed=1)
    env._step_count = cur - 1
    o, r, te, tr, inf = env.step({})
    print(cur, mx, "trunc", tr, "term", te, "rew", r, "step_count", env._step_count)
    env.close()
