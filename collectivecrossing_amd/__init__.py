"""collectivecrossing_amd -- MI355X-native batched CollectiveCrossing step (libccx + host mirror)."""

from .configs import CollectiveCrossingConfig  # noqa: F401

__version__ = "0.1.0"
