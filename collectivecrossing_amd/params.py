"""Lowering of a validated config to the POD ``ccx_params`` of ``include/ccx.h``.

Geometry follows utils/geometry.py:20-47 of the reference; strategy selection follows the
registries (rewards.py:186-216, terminateds.py:86-114, truncateds.py:99-128,
observations.py:122-149): an unknown strategy name is a ``ValueError`` raised when the env is
built, with the reference's message prefix.
"""

from __future__ import annotations

from dataclasses import dataclass

from ._abi import REWARD_MODES, TERMINATED_MODES, TRUNCATED_MODES, CcxParams
from .configs import CollectiveCrossingConfig


@dataclass(frozen=True)
class TramBoundaries:
    """Absolute tram / door columns (utils/geometry.py:10-17)."""

    tram_door_left: int
    tram_door_right: int
    tram_left: int
    tram_right: int


def calculate_tram_boundaries(config: CollectiveCrossingConfig) -> TramBoundaries:
    centre = config.width // 2
    half = config.tram_length // 2
    left = centre - half
    return TramBoundaries(tram_door_left=left + config.tram_door_left,
                          tram_door_right=left + config.tram_door_right,
                          tram_left=left, tram_right=centre + half)


def _lookup(kind: str, name: str, table: dict[str, int]) -> int:
    if name not in table:
        shown = [k for k in table if k != "custom"] if kind != "truncation" else list(table)
        raise ValueError(f"Unknown {kind} function '{name}'. Available: {', '.join(shown)}")
    return table[name]


def lower_config(config: CollectiveCrossingConfig) -> CcxParams:
    """Config -> ``ccx_params`` (what ``CollectiveCrossingEnv.__init__`` resolves at :59-78)."""
    tb = calculate_tram_boundaries(config)
    obs_name = config.observation_config.get_observation_function_name()
    if obs_name != "default":
        raise ValueError(f"Unknown observation function '{obs_name}'. Available: default")
    rc, tc, uc = config.reward_config, config.terminated_config, config.truncated_config
    p = CcxParams()
    p.width, p.height, p.division_y = config.width, config.height, config.division_y
    p.tram_left, p.tram_right = tb.tram_left, tb.tram_right
    p.door_left, p.door_right = tb.tram_door_left, tb.tram_door_right
    p.num_boarding, p.num_exiting = config.num_boarding_agents, config.num_exiting_agents
    p.boarding_dest_y = config.boarding_destination_area_y
    p.exiting_dest_y = config.exiting_destination_area_y
    p.reward_mode = _lookup("reward", rc.get_reward_function_name(), REWARD_MODES)
    p.terminated_mode = _lookup("termination", tc.get_terminated_function_name(), TERMINATED_MODES)
    p.truncated_mode = _lookup("truncation", uc.get_truncated_function_name(), TRUNCATED_MODES)
    p.max_steps = int(getattr(uc, "max_steps"))
    # defaults of the strategy configs that the selected strategy does not read
    p.boarding_destination_reward = float(getattr(rc, "boarding_destination_reward", 15.0))
    p.tram_door_reward = float(getattr(rc, "tram_door_reward", 10.0))
    p.tram_area_reward = float(getattr(rc, "tram_area_reward", 5.0))
    p.distance_penalty_factor = float(getattr(rc, "distance_penalty_factor", 0.1))
    p.goal_reward = float(getattr(rc, "goal_reward", 1.0))
    p.no_goal_reward = float(getattr(rc, "no_goal_reward", 0.0))
    p.step_penalty = float(getattr(rc, "step_penalty", -1.0))
    return p


def agent_ids(config_or_params) -> list[str]:
    """``boarding_0.. exiting_0..`` in slot order (collectivecrossing.py:101-150)."""
    nb = getattr(config_or_params, "num_boarding_agents", None)
    if nb is None:
        nb, ne = config_or_params.num_boarding, config_or_params.num_exiting
    else:
        ne = config_or_params.num_exiting_agents
    return [f"boarding_{i}" for i in range(nb)] + [f"exiting_{j}" for j in range(ne)]
