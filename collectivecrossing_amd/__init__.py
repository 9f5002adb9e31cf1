"""collectivecrossing_amd -- MI355X-native batched CollectiveCrossing step.

``CollectiveCrossingEnv`` (dict API, drop-in for the reference) and ``BatchedCollectiveCrossing``
(array API) both run the step on the GPU through libccx (``include/ccx.h``); there is no CPU
implementation of the step path in this package.
"""

from .configs import CollectiveCrossingConfig  # noqa: F401

__version__ = "0.1.0"
__all__ = ["CollectiveCrossingConfig", "CollectiveCrossingEnv", "BatchedCollectiveCrossing"]


def __getattr__(name):  # lazy: importing the configs must not pull in torch
    if name == "CollectiveCrossingEnv":
        from .env import CollectiveCrossingEnv
        return CollectiveCrossingEnv
    if name == "BatchedCollectiveCrossing":
        from .batched import BatchedCollectiveCrossing
        return BatchedCollectiveCrossing
    raise AttributeError(name)
