"""Strategy plugin interface of the reference, host side (rewards.py, terminateds.py,
truncateds.py, observations.py: ABC + registry dict + ``get_*_function``).

The four built-in families are executed by libccx: each built-in class carries the enum value the
kernel understands (``kernel_mode``) and ``CollectiveCrossingEnv.step`` never calls its Python
method.  The methods exist because they are part of the reference's plugin surface: callers
(and the reference's own tests, e.g. test_rewards.py:476-527 with hand-rolled mock envs) invoke
``strategy.calculate_*(agent_id, env)`` directly; here they evaluate the same rule on the env's
host mirror of the device state.  A user-registered strategy (no ``kernel_mode``) is evaluated on
that mirror after the GPU step and overrides the corresponding output -- the slow, documented path.
"""

from __future__ import annotations

from abc import ABC, abstractmethod

import numpy as np

from . import _abi
from .spaces import Box


def _done(agent_id, env) -> bool:
    a = env._agents[agent_id]
    return bool(a.terminated or a.truncated)


# ------------------------------------------------------------------------------------- rewards
class RewardFunction(ABC):
    kernel_mode: int | None = None
    # A user class whose value depends on NOTHING but the agent's own type and cell (and is ``None`` once the agent is
    # terminated / truncated, like the built-in ones, rewards.py:64) may say so: the batch path then evaluates it once per
    # (type, cell) on the host and runs it inside the kernels as a table (params.position_only_tables, ccx_set_reward_table).
    position_only: bool = False

    def __init__(self, reward_config):
        self.reward_config = reward_config

    @abstractmethod
    def calculate_reward(self, agent_id, env) -> float | None:
        """Reward of ``agent_id`` in the env's current state; ``None`` once it is done."""


class DefaultRewardFunction(RewardFunction):
    """rewards.py:41-99: destination > door > tram-area constants, else a manhattan term whose
    sign is negative for boarding agents and POSITIVE for exiting agents inside the tram."""

    kernel_mode = _abi.REWARD_MODES["default"]

    def calculate_reward(self, agent_id, env):
        if _done(agent_id, env):
            return None
        cfg = self.reward_config
        x, y = (int(v) for v in env._get_agent_position(agent_id))
        centre = (env.tram_door_left + env.tram_door_right) // 2
        if env._agents[agent_id].is_boarding:
            if env.has_agent_reached_destination(agent_id):
                return cfg.boarding_destination_reward
            if env.is_at_tram_door(agent_id):
                return cfg.tram_door_reward
            if env.is_in_tram_area(agent_id):
                return cfg.tram_area_reward
            return np.float64(-(abs(x - centre) + (env.config.division_y - y))) * cfg.distance_penalty_factor
        if env.is_in_exiting_destination_area(agent_id):
            return cfg.boarding_destination_reward
        if not env.is_in_tram_area(agent_id):
            return cfg.tram_area_reward
        return np.float64(abs(x - centre) + (y - env.config.division_y)) * cfg.distance_penalty_factor


class SimpleDistanceRewardFunction(RewardFunction):
    """rewards.py:102-129: ``-|y - destination_y| * factor`` (the destination has no x)."""

    kernel_mode = _abi.REWARD_MODES["simple_distance"]

    def calculate_reward(self, agent_id, env):
        if _done(agent_id, env):
            return None
        pos = env._get_agent_position(agent_id)
        goal = env.get_agent_destination_position(agent_id)
        axis = 1 if goal[0] is None else 0
        return np.float64(-abs(int(pos[axis]) - int(goal[axis]))) * self.reward_config.distance_penalty_factor


class BinaryRewardFunction(RewardFunction):
    """rewards.py:132-159.  The reference compares the position with ``(None, y)`` via
    ``np.array_equal`` which is never true, so every live agent gets ``no_goal_reward``
    (asserted by the reference's test_rewards.py:95-96); reproduced on purpose."""

    kernel_mode = _abi.REWARD_MODES["binary"]

    def calculate_reward(self, agent_id, env):
        return None if _done(agent_id, env) else self.reward_config.no_goal_reward


class ConstantNegativeRewardFunction(RewardFunction):
    kernel_mode = _abi.REWARD_MODES["constant_negative"]

    def calculate_reward(self, agent_id, env):
        return None if _done(agent_id, env) else self.reward_config.step_penalty


REWARD_FUNCTIONS: dict[str, type[RewardFunction]] = {
    "default": DefaultRewardFunction, "simple_distance": SimpleDistanceRewardFunction,
    "binary": BinaryRewardFunction, "constant_negative": ConstantNegativeRewardFunction,
}


# ------------------------------------------------------------------------------------- termination
class TerminatedFunction(ABC):
    kernel_mode: int | None = None
    position_only: bool = False   # as for RewardFunction: terminateds[id] from the agent's own type and cell alone (ccx_set_terminated_table)

    def __init__(self, terminated_config):
        self.terminated_config = terminated_config

    @abstractmethod
    def calculate_terminated(self, agent_id, env) -> bool | None:
        ...


class AllAtDestinationTerminatedFunction(TerminatedFunction):
    kernel_mode = _abi.TERMINATED_MODES["all_at_destination"]

    def calculate_terminated(self, agent_id, env):
        return all(env.has_agent_reached_destination(a) for a in env._agents)


class IndividualAtDestinationTerminatedFunction(TerminatedFunction):
    kernel_mode = _abi.TERMINATED_MODES["individual_at_destination"]

    def calculate_terminated(self, agent_id, env):
        return env.has_agent_reached_destination(agent_id)


TERMINATED_FUNCTIONS: dict[str, type[TerminatedFunction]] = {
    "all_at_destination": AllAtDestinationTerminatedFunction,
    "individual_at_destination": IndividualAtDestinationTerminatedFunction,
}


# ------------------------------------------------------------------------------------- truncation
class TruncatedFunction(ABC):
    kernel_mode: int | None = None

    def __init__(self, truncated_config):
        self.truncated_config = truncated_config

    @abstractmethod
    def calculate_truncated(self, agent_id, env) -> bool | None:
        ...


class MaxStepsTruncatedFunction(TruncatedFunction):
    kernel_mode = _abi.TRUNCATED_MODES["max_steps"]

    def calculate_truncated(self, agent_id, env):
        if _done(agent_id, env):
            return None
        return env._step_count >= self.truncated_config.max_steps


class CustomTruncatedFunction(MaxStepsTruncatedFunction):
    """truncateds.py:64-95: the shipped "custom" strategy only implements the max-steps rule."""

    def calculate_truncated(self, agent_id, env):
        r = super().calculate_truncated(agent_id, env)
        return None if r is None else bool(r)


TRUNCATED_FUNCTIONS: dict[str, type[TruncatedFunction]] = {
    "max_steps": MaxStepsTruncatedFunction, "custom": CustomTruncatedFunction,
}


# ------------------------------------------------------------------------------------- observation
class ObservationFunction(ABC):
    kernel_mode: int | None = None

    def __init__(self, observation_config):
        self.observation_config = observation_config

    @abstractmethod
    def get_agent_observation(self, agent_id, env) -> np.ndarray:
        ...

    @abstractmethod
    def return_agent_observation_space(self, agent_id, env):
        ...


class DefaultObservationFunction(ObservationFunction):
    """observations.py:40-118: ``[x, y, door_centre, division_y, door_left, door_right]`` then a
    4-tuple ``(x, y, type, active)`` per agent in slot order, ``-1`` x4 in the observer's own slot."""

    kernel_mode = 0

    def get_agent_observation(self, agent_id, env):
        ids = list(env._agents)
        me = ids.index(agent_id)
        out = np.empty(6 + 4 * len(ids), np.float32)
        a = env._agents[agent_id]
        out[:6] = (a.x, a.y, (env.tram_door_left + env.tram_door_right) // 2, env.config.division_y,
                   env.tram_door_left, env.tram_door_right)
        for j, other in enumerate(env._agents.values()):
            out[6 + 4 * j:10 + 4 * j] = (-1, -1, -1, -1) if j == me else (
                other.x, other.y, 1 if other.is_exiting else 0, 1 if other.active else 0)
        return out

    def return_agent_observation_space(self, agent_id, env):
        return Box(low=-1, high=max(env.config.width, env.config.height) - 1,
                   shape=(2 + 4 + 4 * len(env._agents),), dtype=np.float32)


OBSERVATION_FUNCTIONS: dict[str, type[ObservationFunction]] = {"default": DefaultObservationFunction}


def _getter(kind: str, table: dict, name_attr: str):
    def get(config):
        name = getattr(config, name_attr)()
        if name not in table:
            raise ValueError(f"Unknown {kind} function '{name}'. Available: {', '.join(table)}")
        return table[name](config)

    return get


get_reward_function = _getter("reward", REWARD_FUNCTIONS, "get_reward_function_name")
get_terminated_function = _getter("termination", TERMINATED_FUNCTIONS, "get_terminated_function_name")
get_truncated_function = _getter("truncation", TRUNCATED_FUNCTIONS, "get_truncated_function_name")
get_observation_function = _getter("observation", OBSERVATION_FUNCTIONS, "get_observation_function_name")
