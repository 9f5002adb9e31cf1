"""Host-side configuration models: same names, fields, defaults and bounds as the reference.

One module holds what the reference spreads over ``configs.py``, ``reward_configs.py``,
``terminated_configs.py``, ``truncated_configs.py``, ``observation_configs.py`` and
``utils/pydantic.py`` (the sibling modules of those names re-export from here so reference import
paths keep working).  Nothing here runs on the GPU: a validated config is lowered ONCE to the POD
``ccx_params`` of ``include/ccx.h`` by :mod:`collectivecrossing_amd.params`.

Parity notes (reference file:line):
  * field bounds  -- configs.py:39-48 (grid and agent counts), reward_configs.py:30-53,66-71,86-97,
    112-117, truncated_configs.py:37-42,56-73, terminated_configs.py:60-74
  * cross-field rules -- configs.py:89-195 (tram fits grid, door inside tram, destinations on the
    right side of the division line, total agents <= min(w*h//4, 50), per-area agent caps,
    render_mode in {human, rgb_array, None})
  * models are frozen, reject unknown fields and validate defaults -- utils/pydantic.py:9-30
"""

from __future__ import annotations

from typing import Any

from pydantic import BaseModel, ConfigDict, Field, model_validator

__all__ = [
    "ConfigClass", "CollectiveCrossingConfig",
    "RewardConfig", "DefaultRewardConfig", "SimpleDistanceRewardConfig", "BinaryRewardConfig",
    "ConstantNegativeRewardConfig", "CustomRewardConfig", "REWARD_CONFIGS", "get_reward_config",
    "TerminatedConfig", "AllAtDestinationTerminatedConfig",
    "IndividualAtDestinationTerminatedConfig", "CustomTerminatedConfig", "TERMINATED_CONFIGS",
    "get_terminated_config",
    "TruncatedConfig", "MaxStepsTruncatedConfig", "CustomTruncatedConfig", "TRUNCATED_CONFIGS",
    "get_truncated_config",
    "ObservationConfig", "DefaultObservationConfig", "OBSERVATION_CONFIGS",
    "get_observation_config",
]


class ConfigClass(BaseModel):
    """Immutable, strict pydantic base (utils/pydantic.py:9-30)."""

    model_config = ConfigDict(extra="forbid", frozen=True, validate_assignment=True,
                              validate_default=True, arbitrary_types_allowed=False,
                              use_enum_values=True, populate_by_name=True)


def _bounded(default: float, lo: float, hi: float, doc: str) -> Any:
    return Field(default=default, ge=lo, le=hi, description=doc)


def _make_factory(kind: str, field: str, table: dict[str, type]) -> Any:
    """``get_<kind>_config(name, **kw)``: registry lookup, ``ValueError`` on unknown names."""

    def factory(name: str, **kwargs: Any) -> Any:
        if name not in table:
            raise ValueError(f"Unknown {kind} function '{name}'. Available: {', '.join(table)}")
        kwargs.pop(field, None)  # tolerate the selector being passed twice
        return table[name](**{field: name}, **kwargs)

    factory.__name__ = f"get_{field.removesuffix('_function')}_config"
    return factory


# --------------------------------------------------------------------------- rewards
class RewardConfig(ConfigClass):
    reward_function: str = Field(description="registry name of the reward strategy")

    def get_reward_function_name(self) -> str:
        return self.reward_function


class DefaultRewardConfig(RewardConfig):
    reward_function: str = "default"
    boarding_destination_reward: float = _bounded(15.0, -100.0, 100.0, "agent is on its destination row")
    tram_door_reward: float = _bounded(10.0, -100.0, 100.0, "boarding agent next to the door")
    tram_area_reward: float = _bounded(5.0, -100.0, 100.0, "boarding agent inside / exiting agent outside the tram")
    distance_penalty_factor: float = _bounded(0.1, 0.0, 10.0, "multiplier of the manhattan distance term")

    def get_reward_function_name(self) -> str:
        return "default"


class SimpleDistanceRewardConfig(RewardConfig):
    reward_function: str = "simple_distance"
    distance_penalty_factor: float = _bounded(0.1, 0.0, 10.0, "multiplier of |y - destination_y|")

    def get_reward_function_name(self) -> str:
        return "simple_distance"


class BinaryRewardConfig(RewardConfig):
    reward_function: str = "binary"
    goal_reward: float = _bounded(1.0, 0.0, 100.0, "reward at the goal")
    no_goal_reward: float = _bounded(0.0, -100.0, 100.0, "reward elsewhere")

    def get_reward_function_name(self) -> str:
        return "binary"


class ConstantNegativeRewardConfig(RewardConfig):
    reward_function: str = "constant_negative"
    step_penalty: float = _bounded(-1.0, -100.0, 0.0, "reward handed out every step")

    def get_reward_function_name(self) -> str:
        return "constant_negative"


class CustomRewardConfig(RewardConfig):
    time_penalty: float = _bounded(0.0, -10.0, 0.0, "per-step penalty")
    goal_bonus: float = _bounded(0.0, 0.0, 100.0, "bonus at the goal")
    collision_penalty: float = _bounded(0.0, -100.0, 0.0, "penalty per collision")
    efficiency_bonus: float = _bounded(0.0, 0.0, 100.0, "bonus for short paths")


REWARD_CONFIGS: dict[str, type[RewardConfig]] = {
    "default": DefaultRewardConfig, "simple_distance": SimpleDistanceRewardConfig,
    "binary": BinaryRewardConfig, "constant_negative": ConstantNegativeRewardConfig,
    "custom": CustomRewardConfig,
}
get_reward_config = _make_factory("reward", "reward_function", REWARD_CONFIGS)


# --------------------------------------------------------------------------- termination
class TerminatedConfig(ConfigClass):
    terminated_function: str = Field(description="registry name of the termination strategy")

    def get_terminated_function_name(self) -> str:
        return self.terminated_function


class AllAtDestinationTerminatedConfig(TerminatedConfig):
    terminated_function: str = "all_at_destination"

    def get_terminated_function_name(self) -> str:
        return "all_at_destination"


class IndividualAtDestinationTerminatedConfig(TerminatedConfig):
    terminated_function: str = "individual_at_destination"

    def get_terminated_function_name(self) -> str:
        return "individual_at_destination"


class CustomTerminatedConfig(TerminatedConfig):
    max_steps_per_agent: int = Field(default=1000, ge=1, le=10000)
    require_all_completion: bool = False
    timeout_penalty: bool = False


TERMINATED_CONFIGS: dict[str, type[TerminatedConfig]] = {
    "all_at_destination": AllAtDestinationTerminatedConfig,
    "individual_at_destination": IndividualAtDestinationTerminatedConfig,
    "custom": CustomTerminatedConfig,
}
get_terminated_config = _make_factory("termination", "terminated_function", TERMINATED_CONFIGS)


# --------------------------------------------------------------------------- truncation
class TruncatedConfig(ConfigClass):
    truncated_function: str = Field(description="registry name of the truncation strategy")

    def get_truncated_function_name(self) -> str:
        return self.truncated_function


class MaxStepsTruncatedConfig(TruncatedConfig):
    truncated_function: str = "max_steps"
    max_steps: int = Field(default=1000, ge=1, le=100000)

    def get_truncated_function_name(self) -> str:
        return "max_steps"


class CustomTruncatedConfig(TruncatedConfig):
    max_steps: int = Field(default=1000, ge=1, le=100000)
    early_truncation_threshold: float = Field(default=0.0, ge=0.0, le=1.0)
    require_all_agents_active: bool = False


TRUNCATED_CONFIGS: dict[str, type[TruncatedConfig]] = {
    "max_steps": MaxStepsTruncatedConfig, "custom": CustomTruncatedConfig,
}
get_truncated_config = _make_factory("truncation", "truncated_function", TRUNCATED_CONFIGS)


# --------------------------------------------------------------------------- observation
class ObservationConfig(ConfigClass):
    observation_function: str = Field(description="registry name of the observation strategy")

    def get_observation_function_name(self) -> str:
        return self.observation_function


class DefaultObservationConfig(ObservationConfig):
    observation_function: str = "default"

    def get_observation_function_name(self) -> str:
        return "default"


OBSERVATION_CONFIGS: dict[str, type[ObservationConfig]] = {"default": DefaultObservationConfig}
get_observation_config = _make_factory("observation", "observation_function", OBSERVATION_CONFIGS)


# --------------------------------------------------------------------------- environment
_RENDER_MODES = ("human", "rgb_array", None)


class CollectiveCrossingConfig(ConfigClass):
    """Grid geometry + agent counts + the four strategy configs (configs.py:15-77)."""

    width: int = Field(ge=1, le=100)
    height: int = Field(ge=1, le=100)
    division_y: int = Field(ge=1, le=100)
    tram_door_left: int = Field(ge=0, le=100)   # relative to tram_left
    tram_door_right: int = Field(ge=0, le=100)  # relative to tram_left
    tram_length: int = Field(ge=1, le=100)
    num_boarding_agents: int = Field(ge=0, le=100)
    num_exiting_agents: int = Field(ge=0, le=100)
    render_mode: str | None = None
    exiting_destination_area_y: int
    boarding_destination_area_y: int
    observation_config: ObservationConfig = Field(default_factory=DefaultObservationConfig)
    reward_config: RewardConfig = Field(default_factory=DefaultRewardConfig)
    terminated_config: TerminatedConfig = Field(default_factory=IndividualAtDestinationTerminatedConfig)
    truncated_config: TruncatedConfig = Field(default_factory=MaxStepsTruncatedConfig)
    # Not a field of the reference: its validator caps the total at min(w*h//4, 50) agents (configs.py:166), so
    # BASELINE configs[4] (32 + 32 agents) cannot even be constructed there.  False lifts that one
    # cap to the library's own limit of 64 agents per env (one wavefront lane per agent); every other rule stays.
    strict_reference_limits: bool = True

    # ---- cross-field rules, one generator so get_validation_errors() can list them all -------
    def _violations(self):
        w, h, div, tl = self.width, self.height, self.division_y, self.tram_length
        if tl > w:
            yield "Tram parameter", f"Tram length ({tl}) cannot exceed grid width ({w})"
        for side, v in (("left", self.tram_door_left), ("right", self.tram_door_right)):
            if not 0 <= v < tl:
                yield "Tram parameter", (f"Tram door {side} boundary ({v}) must be within tram "
                                         f"boundaries (0 to {tl - 1})")
        if self.tram_door_left > self.tram_door_right:
            yield "Tram parameter", (f"Tram door left boundary ({self.tram_door_left}) cannot be "
                                     f"greater than right boundary ({self.tram_door_right})")
        if not 0 <= self.exiting_destination_area_y < div:
            yield "Destination area", (
                f"Exiting destination area y-coordinate ({self.exiting_destination_area_y}) must "
                f"be within waiting area (0 to {div - 1})")
        if not div <= self.boarding_destination_area_y <= h:
            yield "Destination area", (
                f"Boarding destination area y-coordinate ({self.boarding_destination_area_y}) "
                f"must be within tram area ({div} to {h})")
        if div >= h:
            yield "Environment bounds", (f"Division line y-coordinate ({div}) must be less than "
                                         f"environment height ({h})")
        for side, v in (("left", self.tram_door_left), ("right", self.tram_door_right)):
            if v >= w:
                yield "Environment bounds", (f"Tram door {side} boundary ({v}) must be less than "
                                             f"environment width ({w})")
        nb, ne = self.num_boarding_agents, self.num_exiting_agents
        cap = min(w * h // 4, 50) if self.strict_reference_limits else min(w * h // 4, 64)
        if nb + ne > cap:
            yield "Agent count", (f"Total number of agents ({nb + ne}) exceeds reasonable limit "
                                  f"({cap}) for environment size {w}x{h}")
        if ne > (w * div) // 2:
            yield "Agent count", (f"Number of exiting agents ({ne}) may be too high for waiting "
                                  f"area size ({w * div})")
        if nb > (w * (h - div)) // 2:
            yield "Agent count", (f"Number of boarding agents ({nb}) may be too high for tram "
                                  f"area size ({w * (h - div)})")
        if self.render_mode not in _RENDER_MODES:
            yield "Render mode", (f"Invalid render_mode: {self.render_mode}. Valid modes are: "
                                  f"{list(_RENDER_MODES)}")

    @model_validator(mode="after")
    def validate_config(self) -> "CollectiveCrossingConfig":
        for _, message in self._violations():
            raise ValueError(message)
        return self

    def get_validation_errors(self) -> list[str]:
        return [f"{group} error: {message}" for group, message in self._violations()]

    def is_valid(self) -> bool:
        return not self.get_validation_errors()
