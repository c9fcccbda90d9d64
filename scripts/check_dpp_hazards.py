#!/usr/bin/env python
"""Build-time check of the gfx9 data hazard "VALU writes a VGPR -> a DPP instruction reads it as its
shuffled source: 2 wait states" on the library hipcc actually produced.

The kernels move neighbour rows with `v_mov_b32_dpp` / `v_add_f32_dpp` written as inline asm (a wave
shift folded into the add, a persistent destination that keeps the PDE boundary value).  hipcc pads
hazards for its own instructions but does not look inside an asm statement, so whether the two wait
states exist depends on its schedule.  The asm strings that can sit right behind a compiler write carry
their own `s_nop`; this script verifies the rest on the disassembly: for every DPP instruction of every
gfx950 code object in the shared library, no VALU instruction among the preceding two wait states may
write the DPP source register.  Textual predecessors are used (a branch target in the window is treated
as if both paths fell through, which is the conservative reading for straight-line sweep code).

usage: python scripts/check_dpp_hazards.py [path/to/libsigsvgd_hip.so]   -> exit 1 and a listing on a violation
"""
from __future__ import annotations

import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM_BIN = "/opt/rocm/lib/llvm/bin"
_REG = re.compile(r"^v(\d+)$|^v\[(\d+):(\d+)\]$")
_DPP_CTRL = ("wave_shr", "wave_shl", "wave_rol", "wave_ror", "row_shr", "row_shl", "row_ror", "row_mirror",
             "row_half_mirror", "row_bcast", "quad_perm", "row_newbcast", "row_share", "row_xmask")


def _regs(op: str):
    m = _REG.match(op.strip().lstrip("-|").rstrip("|"))
    if not m:
        return set()
    if m.group(1) is not None:
        return {int(m.group(1))}
    return set(range(int(m.group(2)), int(m.group(3)) + 1))


def _split(line: str):
    """'v_add_f64 v[2:3], v[4:5], -v[0:1] ...' -> (mnemonic, [operands])"""
    code = line.split("//")[0].strip()
    if not code or code.endswith(":"):
        return None, []
    parts = code.split(None, 1)
    ops = []
    if len(parts) > 1:
        # operands are comma-separated up to the first modifier token (modifiers follow a space without a comma)
        for k, tok in enumerate(parts[1].split(",")):
            tok = tok.strip()
            ops.append(tok.split()[0] if tok else tok)
    return parts[0], ops


def _writes_vgprs(mn: str, ops):
    if not mn.startswith("v_") or not ops:
        return set()
    if mn.startswith(("v_cmp", "v_cmpx", "v_readlane", "v_readfirstlane")):
        return set()
    return _regs(ops[0])


def _wait_states(mn: str, ops) -> int:
    if mn == "s_nop" and ops:
        try:
            return int(ops[0], 0) + 1
        except ValueError:
            return 1
    return 1


def check_disassembly(text: str):
    insts = []
    for ln in text.splitlines():
        mn, ops = _split(ln)
        if mn and re.match(r"^[sv]_|^ds_|^global_|^buffer_|^flat_|^scratch_", mn):
            insts.append((mn, ops, ln.strip()))
    bad, ndpp = [], 0
    for k, (mn, ops, raw) in enumerate(insts):
        if not (mn.endswith("_dpp") or any(c in raw for c in _DPP_CTRL)):
            continue
        ndpp += 1
        src = _regs(ops[1]) if len(ops) > 1 else set()
        states, j = 0, k - 1
        while j >= 0 and states < 2:
            pmn, pops, praw = insts[j]
            if _writes_vgprs(pmn, pops) & src:
                bad.append((praw, raw, states))
            states += _wait_states(pmn, pops)
            j -= 1
    return ndpp, bad


def disassemble(lib: str):
    tmp = tempfile.mkdtemp(prefix="dpp_hazard_")
    try:
        local = os.path.join(tmp, os.path.basename(lib))
        shutil.copy(lib, local)
        subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "--offloading", local], check=True, capture_output=True)
        outs = []
        for co in sorted(glob.glob(local + ".*gfx950*")):
            r = subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", "--no-show-raw-insn", co], check=True,
                               capture_output=True, text=True)
            outs.append(r.stdout)
        return outs
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main(argv) -> int:
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = argv[1] if len(argv) > 1 else os.path.join(root, "sigsvgd_amd", "libsigsvgd_hip.so")
    total, bad = 0, []
    for text in disassemble(lib):
        n, b = check_disassembly(text)
        total += n
        bad += b
    print(f"{lib}: {total} DPP instructions checked, {len(bad)} with a VALU write of the source inside 2 wait states")
    for w, r, s in bad[:40]:
        print(f"  HAZARD ({s} wait states): {w}   ->   {r}")
    return 1 if bad or total == 0 else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
