"""Reference import path ``collectivecrossing.terminateds``; the classes live in ``strategies``."""

from .strategies import (  # noqa: F401
    TerminatedFunction,
    AllAtDestinationTerminatedFunction,
    IndividualAtDestinationTerminatedFunction,
    TERMINATED_FUNCTIONS,
    get_terminated_function,
)

__all__ = [
    "TerminatedFunction",
    "AllAtDestinationTerminatedFunction",
    "IndividualAtDestinationTerminatedFunction",
    "TERMINATED_FUNCTIONS",
    "get_terminated_function",
]
