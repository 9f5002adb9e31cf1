"""Reference import path ``collectivecrossing.truncated_configs``; the models live in ``configs``."""

from .configs import (  # noqa: F401
    TruncatedConfig,
    MaxStepsTruncatedConfig,
    CustomTruncatedConfig,
    TRUNCATED_CONFIGS,
    get_truncated_config,
)

__all__ = [
    "TruncatedConfig",
    "MaxStepsTruncatedConfig",
    "CustomTruncatedConfig",
    "TRUNCATED_CONFIGS",
    "get_truncated_config",
]
