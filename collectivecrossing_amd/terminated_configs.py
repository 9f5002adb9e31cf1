"""Reference import path ``collectivecrossing.terminated_configs``; the models live in ``configs``."""

from .configs import (  # noqa: F401
    TerminatedConfig,
    AllAtDestinationTerminatedConfig,
    IndividualAtDestinationTerminatedConfig,
    CustomTerminatedConfig,
    TERMINATED_CONFIGS,
    get_terminated_config,
)

__all__ = [
    "TerminatedConfig",
    "AllAtDestinationTerminatedConfig",
    "IndividualAtDestinationTerminatedConfig",
    "CustomTerminatedConfig",
    "TERMINATED_CONFIGS",
    "get_terminated_config",
]
