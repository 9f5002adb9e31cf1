"""Reference import path ``collectivecrossing.truncateds``; the classes live in ``strategies``."""

from .strategies import (  # noqa: F401
    TruncatedFunction,
    MaxStepsTruncatedFunction,
    CustomTruncatedFunction,
    TRUNCATED_FUNCTIONS,
    get_truncated_function,
)

__all__ = [
    "TruncatedFunction",
    "MaxStepsTruncatedFunction",
    "CustomTruncatedFunction",
    "TRUNCATED_FUNCTIONS",
    "get_truncated_function",
]
