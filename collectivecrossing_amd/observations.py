"""Reference import path ``collectivecrossing.observations``; the classes live in ``strategies``."""

from .strategies import (  # noqa: F401
    ObservationFunction,
    DefaultObservationFunction,
    OBSERVATION_FUNCTIONS,
    get_observation_function,
)

__all__ = [
    "ObservationFunction",
    "DefaultObservationFunction",
    "OBSERVATION_FUNCTIONS",
    "get_observation_function",
]
