"""Loader of ``libccx.so`` (the gfx950 HIP library behind ``include/ccx.h``).

There is no CPU implementation of the step path in this package: if the shared library is missing
or does not export the full C-ABI the import fails loudly, and every call that returns a non-zero
``ccx_status`` raises :class:`CcxError`.
"""

from __future__ import annotations

import ctypes as C
from pathlib import Path

from . import _abi

LIB_PATH = Path(__file__).resolve().parent / "libccx.so"
_lib: C.CDLL | None = None


class CcxError(RuntimeError):
    """A libccx call failed (``status`` holds the ccx_status code)."""

    def __init__(self, status: int, message: str):
        super().__init__(f"libccx error {status}: {message}")
        self.status = status


class CcxInputError(CcxError, ValueError):
    """CCX_CHECK_INPUTS found action bytes / move orders the reference would have raised ``ValueError``
    for (collectivecrossing.py:685-711)."""


def load() -> C.CDLL:
    """dlopen libccx.so, bind and type every symbol of include/ccx.h, check the ABI version."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C collectivecrossing_amd/csrc`. collectivecrossing_amd has no "
            "CPU fallback for the step path.")
    lib = C.CDLL(str(LIB_PATH))
    missing = [name for name in _abi.PROTOTYPES if not hasattr(lib, name)]
    if missing:
        raise ImportError(f"{LIB_PATH} does not export {missing} (include/ccx.h)")
    for name, (restype, argtypes) in _abi.PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    got = lib.ccx_abi_version()
    if got != _abi.ABI_VERSION:
        raise ImportError(f"libccx ABI {got} != python bindings ABI {_abi.ABI_VERSION}")
    _lib = lib
    return lib


def check(status: int) -> None:
    if status != _abi.OK:
        msg = load().ccx_last_error()
        text = msg.decode() if msg else ""
        raise (CcxInputError if text.startswith("invalid inputs") else CcxError)(status, text)
