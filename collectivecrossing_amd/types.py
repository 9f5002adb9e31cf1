"""Agent types and the per-agent state view of the dict API (reference: types.py:9-83).

In the reference an ``Agent`` is a dataclass that OWNS its state.  Here the state of every agent
lives in the SoA arrays of the GPU batch; ``Agent`` is a thin view onto the env's host mirror of
those arrays.  Reads return the mirrored values, writes (``agent.position = ...``,
``agent.deactivate()``, ...) update the mirror and mark it dirty so the next ``step`` uploads it
(``ccx_set_state_host``) before launching -- the way the reference's own tests poke
``env._agents[...]`` (tests/collectivecrossing/envs/test_collective_crossing.py:139-143).
"""

from __future__ import annotations

from enum import Enum

import numpy as np


class AgentType(Enum):
    BOARDING = "boarding"
    EXITING = "exiting"


class Agent:
    """View of slot ``index`` of a one-env host state mirror (fields of types.py:16-26)."""

    __slots__ = ("_mirror", "_index", "id", "agent_type")

    def __init__(self, mirror, index: int, agent_id: str, agent_type: AgentType):
        self._mirror, self._index, self.id, self.agent_type = mirror, index, agent_id, agent_type

    # -- position ------------------------------------------------------------------------------
    @property
    def position(self) -> np.ndarray:
        m, i = self._mirror, self._index
        return np.array([m.x[i], m.y[i]])

    @position.setter
    def position(self, value) -> None:
        v = np.asarray(value).reshape(-1)
        self._mirror.write("x", self._index, int(v[0]))
        self._mirror.write("y", self._index, int(v[1]))

    def update_position(self, new_position) -> None:
        self.position = new_position

    @property
    def x(self) -> int:
        return int(self._mirror.x[self._index])

    @property
    def y(self) -> int:
        return int(self._mirror.y[self._index])

    # -- flags ---------------------------------------------------------------------------------
    def _flag(name):  # noqa: N805
        def get(self) -> bool:
            return bool(getattr(self._mirror, name)[self._index])

        def put(self, value) -> None:
            self._mirror.write(name, self._index, int(bool(value)))

        return property(get, put)

    active = _flag("active")
    terminated = _flag("terminated")
    truncated = _flag("truncated")
    del _flag

    def _once(self, name: str, target: bool, message: str) -> None:
        if bool(getattr(self, name)) == target:
            raise ValueError(message)           # types.py:46-73: double calls raise
        setattr(self, name, target)

    def deactivate(self) -> None:
        self._once("active", False, "Agent is already deactivated.")

    def terminate(self) -> None:
        self._once("terminated", True, "Agent is already terminated.")

    def truncate(self) -> None:
        self._once("truncated", True, "Agent is already truncated.")

    is_boarding = property(lambda self: self.agent_type == AgentType.BOARDING)
    is_exiting = property(lambda self: self.agent_type == AgentType.EXITING)
    is_terminated = property(lambda self: self.terminated)
    is_truncated = property(lambda self: self.truncated)

    def __repr__(self) -> str:
        return (f"Agent(id={self.id!r}, type={self.agent_type.value}, pos=({self.x},{self.y}), "
                f"active={self.active}, terminated={self.terminated}, truncated={self.truncated})")
