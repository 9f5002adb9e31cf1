"""Host-side scripted policies with the reference's interface (src/baseline_policies/).

``GreedyPolicy`` / ``WaitingPolicy`` answer ``get_action(agent_id, observation, env)`` for ONE agent of
ONE env through the env's host views (``env._get_agent``, ``env._is_move_valid``, ...), exactly the
surface the reference's policies use, so the reference's demo loops
(scripts/run_greedy_policy_demo.py:67-109) run unchanged.  With ``randomness_factor > 0`` the
epsilon draws consume ``numpy.random.RandomState(seed)`` in the reference's order (one ``random()``
per call, then ``choice(valid_actions)``), which makes whole epsilon-greedy episodes reproducible
against the reference (tests/golden/g11_*).

This is the per-call convenience layer.  The fast path for epsilon = 0 is on the device:
``BatchedCollectiveCrossing.policy_actions`` (all agents of all envs in one launch) and
``rollout_greedy`` (policy + step fused, include/ccx.h: ccx_rollout_policy); both implement the same
rule (csrc/ccx_greedy.h) and are pinned to the same reference recordings.
"""

from __future__ import annotations

import numpy as np

from .actions import ACTION_TO_DIRECTION

__all__ = ["GreedyPolicy", "WaitingPolicy", "create_greedy_policy", "create_waiting_policy"]

RIGHT, UP, LEFT, DOWN, WAIT = 0, 1, 2, 3, 4


class GreedyPolicy:
    """Head for the door-centre column on the row next to the door, cross, then head for the
    destination row; when the preferred move is blocked try the preference list, else wait
    (greedy_policy.py:33-449)."""

    def __init__(self, randomness_factor: float, seed: int) -> None:
        self.randomness_factor = randomness_factor
        self.random_state = np.random.RandomState(seed)

    # ------------------------------------------------------------------ the reference's entry point
    def get_action(self, agent_id: str, observation, env) -> int:
        if self.randomness_factor > 0.0 and self.random_state.random() < self.randomness_factor:
            valid = [a for a in range(5) if self._is_valid_action(agent_id, a, env)]   # :48-60
            return self.random_state.choice(valid) if valid else WAIT
        return self._scripted_action(agent_id, env)

    def _scripted_action(self, agent_id: str, env) -> int:
        for action in self._candidates(agent_id, env):
            if self._is_valid_action(agent_id, action, env):
                return action
        return WAIT

    # ------------------------------------------------------------------ the rule
    def _candidates(self, agent_id: str, env) -> list[int]:
        """Primary action, then the fallback preference list (same table as csrc/ccx_greedy.h)."""
        agent = env._get_agent(agent_id)
        boarding = agent.agent_type.value == "boarding"
        cx, cy = int(agent.position[0]), int(agent.position[1])
        div = env.config.division_y
        door_x = (env.tram_door_left + env.tram_door_right) // 2
        dest_y = env.get_agent_destination_position(agent_id)[1]
        before_door = cy < div if boarding else cy > div                       # :118, :139
        door_level = div - 1 if boarding else div + 1                          # :122, :143
        forward, back = (UP, DOWN) if boarding else (DOWN, UP)
        dx = dy = 0
        if before_door:
            if cy == door_level:
                if cx == door_x:
                    dy = 1 if boarding else -1                                 # :126-127, :148-149
                else:
                    dx = int(np.sign(door_x - cx))                             # :129-130
            else:
                dy = int(np.sign(door_level - cy))                             # :133-134, :155-156
        else:
            dy = int(np.sign(dest_y - cy))                                     # :137, :159
        primary = RIGHT if dx == 1 else UP if dy == 1 else LEFT if dx == -1 else DOWN if dy == -1 else WAIT
        if before_door and cx != door_x:                                       # :334-349, :396-411
            toward, away = (RIGHT, LEFT) if cx < door_x else (LEFT, RIGHT)
            prefs = [toward, forward, away]
        else:                                                                  # :350-389, :412-449
            prefs = [forward, RIGHT, LEFT]
        return [primary, *prefs, back, WAIT]

    def _is_valid_action(self, agent_id: str, action: int, env) -> bool:
        """wait is always fine; a move must pass the env's own validity check (:238-264)."""
        if action == WAIT:
            return True
        cur = env._get_agent_position(agent_id)
        return bool(env._is_move_valid(agent_id, cur, cur + ACTION_TO_DIRECTION[action]))


class WaitingPolicy(GreedyPolicy):
    """Greedy, except that boarding agents outside the tram area wait until every live exiting
    agent stands on its destination row (waiting_policy.py:33-131)."""

    def get_action(self, agent_id: str, observation, env) -> int:
        if self.randomness_factor > 0.0 and self.random_state.random() < self.randomness_factor:
            valid = [a for a in range(5) if self._is_valid_action(agent_id, a, env)]
            return self.random_state.choice(valid) if valid else WAIT
        if self._should_agent_wait(agent_id, env):
            return WAIT
        return self._scripted_action(agent_id, env)

    def _should_agent_wait(self, agent_id: str, env) -> bool:
        agent = env._get_agent(agent_id)
        if agent.agent_type.value != "boarding" or env.is_in_tram_area(agent_id):
            return False
        for other_id, other in env._agents.items():                            # :102-131
            if other.terminated or other.truncated or other.agent_type.value != "exiting":
                continue
            if not env.has_agent_reached_destination(other_id):
                return True
        return False


def create_greedy_policy(epsilon: float = 0.1) -> GreedyPolicy:
    """greedy_policy.py:452-465 (the shared seed 42 is part of the reference's behaviour)."""
    return GreedyPolicy(randomness_factor=epsilon, seed=42)


def create_waiting_policy(epsilon: float = 0.1) -> WaitingPolicy:
    """waiting_policy.py:539-552."""
    return WaitingPolicy(randomness_factor=epsilon, seed=42)
