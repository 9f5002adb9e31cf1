"""collectivecrossing_amd -- MI355X-native batched CollectiveCrossing step.

``CollectiveCrossingEnv`` (dict API, drop-in for the reference), ``BatchedCollectiveCrossing``
(array API), ``VectorCollectiveCrossing`` (many envs behind per-env dict views) and
``rllib.BatchedMultiAgentEnv`` (the batch as one RLlib ``MultiAgentEnv`` with flat agent ids) all run the step on
the GPU through libccx (``include/ccx.h``); there is no CPU implementation of the step path in this
package.
"""

from .configs import CollectiveCrossingConfig  # noqa: F401

__version__ = "0.4.0"
__all__ = ["CollectiveCrossingConfig", "CollectiveCrossingEnv", "BatchedCollectiveCrossing", "VectorCollectiveCrossing",
           "BatchedMultiAgentEnv"]


def __getattr__(name):  # lazy: importing the configs must not pull in torch
    if name == "CollectiveCrossingEnv":
        from .env import CollectiveCrossingEnv
        return CollectiveCrossingEnv
    if name == "BatchedCollectiveCrossing":
        from .batched import BatchedCollectiveCrossing
        return BatchedCollectiveCrossing
    if name == "VectorCollectiveCrossing":
        from .vector import VectorCollectiveCrossing
        return VectorCollectiveCrossing
    if name == "BatchedMultiAgentEnv":
        from .rllib import BatchedMultiAgentEnv
        return BatchedMultiAgentEnv
    raise AttributeError(name)


def _register_with_gymnasium() -> None:
    """Same env id as the reference (collectivecrossing/__init__.py:7-10) when gymnasium exists."""
    try:
        from gymnasium.envs.registration import register, registry
    except Exception:
        return
    env_id = "collectivecrossing/CollectiveCrossing-v0"
    if env_id not in registry:
        register(id=env_id, entry_point="collectivecrossing_amd.env:CollectiveCrossingEnv")


_register_with_gymnasium()
