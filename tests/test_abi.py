"""The C-ABI library loads and exports every symbol include/ccx.h declares (no GPU compute)."""

import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _declared_symbols():
    text = (ROOT / "include" / "ccx.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ccx_[a-z_0-9]+)\s*\(", text)))


def test_header_and_bindings_agree():
    from collectivecrossing_amd import _abi

    assert _declared_symbols() == sorted(_abi.PROTOTYPES)


def test_library_exports_every_declared_symbol():
    from collectivecrossing_amd import _lib

    if not _lib.LIB_PATH.exists():
        pytest.skip("libccx.so not built (run __graft_entry__.build())")
    lib = _lib.load()
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    from collectivecrossing_amd import _abi

    assert lib.ccx_abi_version() == _abi.ABI_VERSION == 5
    assert lib.ccx_obs_len(8) == 38 and lib.ccx_obs_len(3) == 18
    assert b"gfx950" in lib.ccx_build_info()


def test_struct_layouts_match_the_header():
    """sizeof/offsets of the ctypes mirrors against the C compiler's view of include/ccx.h."""
    import subprocess
    import tempfile

    from collectivecrossing_amd import _abi

    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "ccx.h"
int main(void){
 printf("%zu %zu %zu %zu %zu\n", sizeof(ccx_params), offsetof(ccx_params, max_steps),
        offsetof(ccx_params, boarding_destination_reward), offsetof(ccx_params, step_penalty),
        sizeof(ccx_counters));
 printf("%zu %zu %zu\n", sizeof(ccx_state), sizeof(ccx_step_out), sizeof(ccx_rollout_out));
 return 0; }'''
    with tempfile.TemporaryDirectory() as d:
        (Path(d) / "t.c").write_text(src)
        subprocess.run(["gcc", "-I", str(ROOT / "include"), "-o", f"{d}/t", f"{d}/t.c"], check=True)
        out = subprocess.run([f"{d}/t"], check=True, capture_output=True, text=True).stdout.split()
    P = _abi.CcxParams
    exp = [ctypes.sizeof(P), P.max_steps.offset, P.boarding_destination_reward.offset,
           P.step_penalty.offset, ctypes.sizeof(_abi.CcxCounters), ctypes.sizeof(_abi.CcxState),
           ctypes.sizeof(_abi.CcxStepOut), ctypes.sizeof(_abi.CcxRolloutOut)]
    assert [int(v) for v in out] == exp


def test_no_gpu_means_loud_failure():
    """Without a GPU the product refuses to run (no CPU fallback)."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from _fixtures import Golden

    from collectivecrossing_amd.batched import BatchedCollectiveCrossing

    with pytest.raises((RuntimeError, ImportError)):
        BatchedCollectiveCrossing(Golden("g7_n3_small").config, 4)


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under collectivecrossing_amd/ may import, load,
    link or call it (imports, the library name, its ccxo_ symbols, its directory)."""
    needles = ("import oracle", "from oracle", "oracle/", "oracle.", "ccxo_", "libccx_oracle", "ccx_oracle")
    for p in (ROOT / "collectivecrossing_amd").rglob("*"):
        if p.suffix in (".py", ".hip", ".h", ".cpp") or p.name == "Makefile":
            text = p.read_text()
            for n in needles:
                assert n not in text, (p, n)
