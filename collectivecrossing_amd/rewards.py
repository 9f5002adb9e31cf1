"""Reference import path ``collectivecrossing.rewards``; the classes live in ``strategies``."""

from .strategies import (  # noqa: F401
    RewardFunction,
    DefaultRewardFunction,
    SimpleDistanceRewardFunction,
    BinaryRewardFunction,
    ConstantNegativeRewardFunction,
    REWARD_FUNCTIONS,
    get_reward_function,
)

__all__ = [
    "RewardFunction",
    "DefaultRewardFunction",
    "SimpleDistanceRewardFunction",
    "BinaryRewardFunction",
    "ConstantNegativeRewardFunction",
    "REWARD_FUNCTIONS",
    "get_reward_function",
]
