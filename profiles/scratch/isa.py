#!/usr/bin/env python3
"""ISA inspection helper for the rollout kernels (CPU-only).  usage:
  isa.py build [GLOG]         compile ccx_rollout_g.hip (one lane-group size, default 3: seconds) with -save-temps into /tmp/asm
  isa.py res [pattern]        VGPR / SGPR / spills / scratch per kernel from the last build
  isa.py loop <mangled-substring> [out.s]   the sim step loop (from the occupancy atomics' block to its back edge)"""
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
SRC = ROOT / "collectivecrossing_amd" / "csrc"
TMP = Path("/tmp/asm")
FLAGS = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function".split()


def build(glog):
    TMP.mkdir(exist_ok=True)
    glog = 3 if glog is None else glog
    defs = [f"-DCCX_GLOG={glog}"]
    tag = f"g{glog}"
    r = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *defs, "-save-temps=obj", "-Rpass-analysis=kernel-resource-usage",
                        "-c", "ccx_rollout_g.hip", "-o", str(TMP / f"{tag}.o")], cwd=SRC, capture_output=True, text=True)
    (TMP / f"{tag}_res.txt").write_text(r.stderr)
    errs = [ln for ln in r.stderr.splitlines() if not ln.startswith("remark")]
    print("\n".join(errs[:60]))
    src = TMP / "ccx_rollout_g-hip-amdgcn-amd-amdhsa-gfx950.s"
    if src.exists():
        src.replace(TMP / f"{tag}.s")
    (TMP / "last").write_text(tag)
    return r.returncode


def res(pattern=""):
    tag = (TMP / "last").read_text().strip()
    txt = (TMP / f"{tag}_res.txt").read_text()
    for blk in txt.split("Function Name: ")[1:]:
        name = blk.split()[0]
        if "rollout_kernel" not in name or pattern not in name:
            continue
        g = lambda k: int(re.search(k + r": (\d+)", blk).group(1))  # noqa: E731
        m = re.search(r"rollout_kernel(_v128)?ILi(\d)ELb(\d)ELi(\d)ELb(\d)ELb(\d)E", name)
        short = f"{'v128' if m.group(1) else '    '} G{m.group(2)} pair{m.group(3)} out{m.group(4)} occ{m.group(5)} plain{m.group(6)}"
        scr, occ = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")
        print(f"{short}: VGPR {g('VGPRs'):3d} SGPR {g('TotalSGPRs'):3d} sgpr-spill {g('SGPRs Spill'):3d} "
              f"vgpr-spill {g('VGPRs Spill'):3d} scratch {scr:3d} occ {occ}")


def func(sub):
    tag = (TMP / "last").read_text().strip()
    lines = (TMP / f"{tag}.s").read_text().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN3ccx") and sub in l and l.rstrip().endswith(":") is False and ":" in l.split(";")[0])
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return lines[start:end]


def loop(sub, out=None):
    f = func(sub)
    code = [l for l in f if l.strip() and not l.strip().startswith(";")]
    # the sim step loop: the block that holds the first ds_or after s_setprio, up to the branch back to it
    sp = next(i for i, l in enumerate(code) if "s_setprio" in l)
    o = next(i for i in range(sp, len(code)) if "ds_or_b" in l_(code[i]))
    # loop header = nearest label above `o` that some later branch targets
    labels = {l.split(":")[0]: i for i, l in enumerate(code) if re.match(r"^\.LBB\d+_\d+:", l)}
    best = None
    for i in range(o, len(code)):
        m = re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", code[i])
        if m and m.group(1) in labels and labels[m.group(1)] <= o:
            best = (labels[m.group(1)], i)
            break
    a, b = best
    body = code[a:b + 1]
    n_valu = sum(1 for l in body if re.match(r"\s+v_", l))
    n_salu = sum(1 for l in body if re.match(r"\s+s_", l) and "s_waitcnt" not in l and "s_nop" not in l)
    n_lds = sum(1 for l in body if re.match(r"\s+ds_", l))
    n_br = sum(1 for l in body if "branch" in l)
    print(f"step loop: {len(body)} lines, VALU {n_valu}, SALU {n_salu}, LDS {n_lds}, branches {n_br}, "
          f"readlane/writelane {sum(1 for l in body if 'v_readlane' in l or 'v_writelane' in l)}")
    text = "\n".join(body)
    if out:
        Path(out).write_text(text)
    else:
        print(text)


def l_(s):
    return s


if __name__ == "__main__":
    cmd = sys.argv[1]
    if cmd == "build":
        sys.exit(build(int(sys.argv[2]) if len(sys.argv) > 2 else None))
    elif cmd == "res":
        res(sys.argv[2] if len(sys.argv) > 2 else "")
    elif cmd == "loop":
        loop(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
