"""Action codes of the agents (reference: actions.py:8-24)."""

from enum import Enum

import numpy as np


class Actions(Enum):
    right = 0
    up = 1
    left = 2
    down = 3
    wait = 4


_DELTAS = {Actions.right: (1, 0), Actions.up: (0, 1), Actions.left: (-1, 0), Actions.down: (0, -1),
           Actions.wait: (0, 0)}
ACTION_TO_DIRECTION = {a.value: np.array(d) for a, d in _DELTAS.items()}
