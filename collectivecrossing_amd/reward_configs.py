"""Reference import path ``collectivecrossing.reward_configs``; the models live in ``configs``."""

from .configs import (  # noqa: F401
    RewardConfig,
    DefaultRewardConfig,
    SimpleDistanceRewardConfig,
    BinaryRewardConfig,
    ConstantNegativeRewardConfig,
    CustomRewardConfig,
    REWARD_CONFIGS,
    get_reward_config,
)

__all__ = [
    "RewardConfig",
    "DefaultRewardConfig",
    "SimpleDistanceRewardConfig",
    "BinaryRewardConfig",
    "ConstantNegativeRewardConfig",
    "CustomRewardConfig",
    "REWARD_CONFIGS",
    "get_reward_config",
]
