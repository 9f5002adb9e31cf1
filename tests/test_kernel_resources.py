"""Register budget of the built rollout kernels, read from the code objects inside libccx.so (no GPU needed).

The launch shapes count on 16 resident wavefronts of `ccx::rollout_kernel` per CU (4 per SIMD), i.e. at most 128
VGPRs and no scratch: when the 64-lane policy instantiations drifted to 129-130 VGPRs in round 2, C5 fell from
0.87 to 0.49 of the HBM peak without a single test noticing.  Instantiations with the LDS occupancy tables (OCC,
every grid the BASELINE configs use) must stay within the budget; the all-pairs fallback for huge grids may not."""

import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
LLVM = Path("/opt/rocm/lib/llvm/bin")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _kernels(tmp_path):
    so = ROOT / "collectivecrossing_amd" / "libccx.so"
    if not so.exists() or not (LLVM / "clang-offload-bundler").exists() or not shutil.which("objcopy"):
        pytest.skip("libccx.so or the ROCm LLVM tools are not available")
    fat = tmp_path / "fat.bin"
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", str(so), str(fat)], check=True)
    blob = fat.read_bytes()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    out = {}
    for n, a in enumerate(starts):
        part = tmp_path / f"bundle{n}.bin"
        part.write_bytes(blob[a:starts[n + 1] if n + 1 < len(starts) else len(blob)])
        co = tmp_path / f"bundle{n}.co"
        r = subprocess.run([str(LLVM / "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={part}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], capture_output=True, text=True)
        if r.returncode or not co.exists() or co.stat().st_size == 0:
            continue
        notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(co)], capture_output=True, text=True).stdout
        for block in notes.split("- .agpr_count")[1:]:
            name = re.search(r"\.name:\s+(\S+)", block)
            vgpr = re.search(r"\.vgpr_count:\s+(\d+)", block)
            scratch = re.search(r"\.private_segment_fixed_size:\s+(\d+)", block)
            if name and vgpr:
                out[name.group(1)] = (int(vgpr.group(1)), int(scratch.group(1)) if scratch else 0)
    return out


def test_kernarg_tail_offsets_match_the_code_objects(tmp_path):
    """The rollout kernels' epilogue re-reads the state / counter pointers from the kernel-argument segment
    (ccx_kernels.hip: KernargTail: declaration order, natural alignment).  The `.args` metadata of the built code objects
    says where every argument really lies: KState directly behind KParams, `counters` where the struct puts it."""
    so = ROOT / "collectivecrossing_amd" / "libccx.so"
    if not so.exists() or not (LLVM / "clang-offload-bundler").exists() or not shutil.which("objcopy"):
        pytest.skip("libccx.so or the ROCm LLVM tools are not available")
    fat = tmp_path / "fat.bin"
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", str(so), str(fat)], check=True)
    blob = fat.read_bytes()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    checked = 0
    for n, a in enumerate(starts):
        part = tmp_path / f"kb{n}.bin"
        part.write_bytes(blob[a:starts[n + 1] if n + 1 < len(starts) else len(blob)])
        co = tmp_path / f"kb{n}.co"
        r = subprocess.run([str(LLVM / "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={part}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], capture_output=True, text=True)
        if r.returncode or not co.exists() or co.stat().st_size == 0:
            continue
        notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(co)], capture_output=True, text=True).stdout
        for block in notes.split("  - .agpr_count")[1:]:
            name = re.search(r"\.name:\s+(\S+)", block)
            if not name or "rollout_kernel" not in name.group(1):
                continue
            args_txt = block.split(".group_segment_fixed_size")[0]
            args = [(int(o), int(z)) for o, z in re.findall(r"\.offset:\s+(\d+)\s+\.size:\s+(\d+)", args_txt)]
            explicit = [a_ for a_ in args][:12]
            assert len(explicit) == 12, (name.group(1), args)
            kp_size = explicit[0][1]
            assert explicit[0][0] == 0 and kp_size % 8 == 0
            assert explicit[1] == (kp_size, 56)                           # KState: 7 pointers, directly behind KParams
            # cell_info, actions, order (8 each), K, auto_reset (4 each), pool (8), KOut (5 pointers), counters
            assert explicit[9][0] == kp_size + 56 + 24 + 8 + 8 + 40 and explicit[9][1] == 8, (name.group(1), explicit)
            checked += 1
    assert checked >= 100


def test_rollout_kernels_fit_sixteen_wavefronts_per_cu(tmp_path):
    ks = {k: v for k, v in _kernels(tmp_path).items() if "rollout_kernel" in k}
    assert len(ks) >= 100, sorted(ks)[:5]          # 7 lane-group sizes x PAIR x OUTM x OCC x PLAIN
    # mangled template arguments: ILi<GLOG>ELb<PAIR>ELi<OUTM>ELb<OCC>ELb<PLAIN>E
    occ = {k: v for k, v in ks.items() if re.search(r"ILi\dELb[01]ELi[012]ELb1ELb[01]E", k)}
    assert len(occ) >= 80
    over = {k: v for k, v in occ.items() if v[0] > 128}
    assert not over, f"instantiations over 128 VGPRs (12 instead of 16 wavefronts per CU): {over}"
    # no scratch in the plain instantiations (the bench line, RL stepping) nor in the 64-lane policy ones (C5); the
    # register-bounded policy / move-order instantiations of smaller lane groups may spill a few dwords
    spills = {k: v for k, v in occ.items() if v[1] > (16 if ("v128" in k and "ILi6E" not in k) else 0)}
    assert not spills, f"scratch spills: {spills}"
    hot = [v for k, v in occ.items() if "ILi3ELb1ELi1ELb1ELb1E" in k and "v128" not in k]     # C2's kernel
    assert hot and hot[0][0] <= 128
