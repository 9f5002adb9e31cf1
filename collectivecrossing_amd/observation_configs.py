"""Reference import path ``collectivecrossing.observation_configs``; the models live in ``configs``."""

from .configs import (  # noqa: F401
    ObservationConfig,
    DefaultObservationConfig,
    OBSERVATION_CONFIGS,
    get_observation_config,
)

__all__ = [
    "ObservationConfig",
    "DefaultObservationConfig",
    "OBSERVATION_CONFIGS",
    "get_observation_config",
]
