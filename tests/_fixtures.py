"""Loading of the committed golden vectors (tests/golden/*.npz, made by gen_golden.py)."""

from __future__ import annotations

import json
from pathlib import Path

import numpy as np

from collectivecrossing_amd import configs as C
from collectivecrossing_amd.params import lower_config

GOLDEN = Path(__file__).resolve().parent / "golden"
PLUGIN_NPZ = sorted(p.stem for p in GOLDEN.glob("g13_*.npz"))          # need tests/golden/custom_strategies.py registered first
ALL_NPZ = sorted(p.stem for p in GOLDEN.glob("*.npz") if p.stem not in PLUGIN_NPZ)
ROLLOUT_NPZ = [n for n in ALL_NPZ if n.startswith("g8_")]
STEP_NPZ = [n for n in ALL_NPZ if not n.startswith("g8_")]


def config_from_dict(cfg: dict) -> C.CollectiveCrossingConfig:
    """Fixture config dict -> our CollectiveCrossingConfig (``_relaxed`` skips validation, the
    way the survey ran BASELINE config 5 through the reference with ``model_construct``)."""
    kw = {k: v for k, v in cfg.items() if not k.endswith("_config") and not k.startswith("_")}
    rc = dict(cfg.get("reward_config", {"reward_function": "default"}))
    tc = dict(cfg.get("terminated_config", {"terminated_function": "individual_at_destination"}))
    uc = dict(cfg.get("truncated_config", {"truncated_function": "max_steps"}))
    # (a name outside the config registry = a user-registered strategy class, g13: Custom*Config carries the name; the classes
    # themselves are registered by the tests that load these fixtures, tests/golden/custom_strategies.py)
    kw["reward_config"] = (C.get_reward_config(rc.pop("reward_function"), **rc) if rc["reward_function"] in C.REWARD_CONFIGS
                           else C.CustomRewardConfig(**rc))
    kw["terminated_config"] = (C.get_terminated_config(tc.pop("terminated_function"), **tc)
                               if tc["terminated_function"] in C.TERMINATED_CONFIGS else C.CustomTerminatedConfig(**tc))
    kw["truncated_config"] = C.get_truncated_config(uc.pop("truncated_function"), **uc)
    if cfg.get("_relaxed"):
        kw.setdefault("observation_config", C.DefaultObservationConfig())
        kw.setdefault("render_mode", None)
        return C.CollectiveCrossingConfig.model_construct(**kw)
    return C.CollectiveCrossingConfig(**kw)


class Golden:
    def __init__(self, name: str):
        self.name = name
        with np.load(GOLDEN / f"{name}.npz") as z:
            self.a = {k: z[k] for k in z.files}
        self.cfg_dict = json.loads(str(self.a["config_json"]))
        self.config = config_from_dict(self.cfg_dict)
        self.params = lower_config(self.config, allow_position_only=True)
        self.K, self.E, self.N = self.a["actions"].shape
        self.L = 6 + 4 * self.N

    def __getitem__(self, k):
        return self.a[k]

    def init_state(self) -> dict:
        return dict(x=self["init_x"], y=self["init_y"], active=self["init_active"],
                    terminated=self["init_terminated"], truncated=self["init_truncated"],
                    step_count=self["init_step_count"])


def assert_step_matches(g: Golden, s, obs, reward, af, ef, state, *, label="") -> None:
    """Bit-exact comparison of one step's outputs + post-step state with the golden."""
    tag = f"{g.name} step {s} {label}"
    np.testing.assert_array_equal(state["x"], g["x"][s], err_msg=f"x {tag}")
    np.testing.assert_array_equal(state["y"], g["y"][s], err_msg=f"y {tag}")
    for k in ("active", "terminated", "truncated"):
        np.testing.assert_array_equal(state[k], g[k][s], err_msg=f"{k} {tag}")
    np.testing.assert_array_equal(state["step_count"], g["step_count"][s], err_msg=f"step_count {tag}")
    np.testing.assert_array_equal(af, g["agent_flags"][s], err_msg=f"agent_flags {tag}")
    np.testing.assert_array_equal(ef & 3, g["env_flags"][s] & 3, err_msg=f"env_flags {tag}")
    # rewards: f64 bit patterns (incl. the sign of zero) where the agent was live
    live = (g["agent_flags"][s] & 4) != 0
    got = np.where(live, reward, 0.0).view(np.uint64)
    exp = np.where(live, g["reward"][s], 0.0).view(np.uint64)
    np.testing.assert_array_equal(got, exp, err_msg=f"reward bits {tag}")
    if obs is not None:
        np.testing.assert_array_equal(obs.view(np.uint32), g["obs"][s].view(np.uint32),
                                      err_msg=f"obs {tag}")
