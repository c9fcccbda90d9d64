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


# ---- second check: the sweep statements' EXEC discipline ----------------------------------------------------------------
# The hand-written sweep statements (csrc/gram_fast.hip, csrc/quad_sweeps.h) move lane windows into EXEC and leave it at all
# ones.  EXEC is a reserved register for hipcc (a clobber on it is ignored with a warning), so the statements are only correct
# where the compiler's own EXEC is all ones: in wave-uniform control flow.  On the disassembly that reads: no statement -- an
# `s_mov_b64 exec, -1` directly followed by a wave_shr / wave_shl DPP move -- may lie
#   (a) between an `s_and_saveexec_b64 X` / `s_or_saveexec_b64 X` and the `s_or_b64 exec, exec, X` that closes it, or
#   (b) inside a divergent loop: one whose back edge is `s_cbranch_execnz`, or the innermost loop around an
#       `s_andn2_b64 exec, exec, X` (lanes leave the loop one by one: EXEC shrinks per trip).
# Textual nesting is used (hipcc lays structured regions out contiguously; a region moved out of line would escape this
# reading).  What this finds are the regions the COMPILER treats as divergent -- conditions it could not prove wave-uniform;
# they are harmless while every lane takes them at run time (the statement then restores exactly the mask the compiler
# expects), which is the call sites' contract and is checked at run time by the -DSIGSVGD_CHECK_EXEC build.  Informational.
_ADDR = re.compile(r"//\s*([0-9A-Fa-f]+):")


def check_exec_regions(text: str):
    """-> (statements found, [(function, address, reason)])"""
    found, bad = 0, []
    func, insts = None, []

    def flush():
        nonlocal found
        if not insts:
            return
        # loops = backward branches [target, branch]; a loop is DIVERGENT when its back edge is s_cbranch_execnz or when it is
        # the innermost loop around an `s_andn2_b64 exec, exec, X` (lanes leave the loop one by one)
        allloops, loops = [], []
        for addr, mn, ops, raw in insts:
            if (mn == "s_branch" or mn.startswith("s_cbranch_")) and ops and addr is not None:
                try:
                    off = int(ops[0], 0)
                except ValueError:
                    continue
                if off >= 0x8000:
                    off -= 0x10000
                tgt = addr + 4 + 4 * off
                if tgt <= addr:
                    allloops.append((tgt, addr))
                    if mn == "s_cbranch_execnz":
                        loops.append((tgt, addr))
        for addr, mn, ops, raw in insts:
            if mn in ("s_andn2_b64", "s_andn2_saveexec_b64") and ops and ops[0] == "exec" and addr is not None:
                inner = [l for l in allloops if l[0] <= addr <= l[1]]
                if inner:
                    loops.append(min(inner, key=lambda l: l[1] - l[0]))
        stack = []
        for k, (addr, mn, ops, raw) in enumerate(insts):
            if mn in ("s_and_saveexec_b64", "s_or_saveexec_b64") and ops:
                stack.append(ops[0])
            elif mn == "s_or_b64" and len(ops) >= 3 and ops[0] == "exec" and ops[1] == "exec" and ops[2] in stack:
                while stack and stack.pop() != ops[2]:
                    pass
            elif mn == "s_mov_b64" and len(ops) >= 2 and ops[0] == "exec" and ops[1] in stack:
                while stack and stack.pop() != ops[1]:
                    pass
            elif (mn == "s_mov_b64" and len(ops) >= 2 and ops[0] == "exec" and ops[1] == "-1" and k + 1 < len(insts)
                  and insts[k + 1][1].startswith("v_mov_b32_dpp") and ("wave_shr" in insts[k + 1][3] or "wave_shl" in insts[k + 1][3])):
                found += 1
                if stack:
                    bad.append((func, addr, f"inside the EXEC region saved in {stack[-1]}"))
                elif addr is not None and any(lo <= addr <= hi for lo, hi in loops):
                    bad.append((func, addr, "inside a divergent loop (s_cbranch_exec* back edge)"))

    for ln in text.splitlines():
        m = re.match(r"^[0-9a-fA-F]+ <(.+)>:\s*$", ln)
        if m:
            flush()
            func, insts = m.group(1), []
            continue
        mn, ops = _split(ln)
        if mn and re.match(r"^[sv]_|^ds_|^global_|^buffer_|^flat_|^scratch_", mn):
            a = _ADDR.search(ln)
            insts.append((int(a.group(1), 16) if a else None, mn, ops, ln.strip()))
    flush()
    return found, bad


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
    nst, ebad = 0, []
    for text in disassemble(lib):
        n, b = check_exec_regions(text)
        nst += n
        ebad += b
    # informational: a region the COMPILER treats as divergent (a condition it could not prove wave-uniform) is not a violation
    # when every lane takes it at run time, which is what the call sites guarantee; the run-time check of that guarantee is
    # scripts/dev/check_exec.py (-DSIGSVGD_CHECK_EXEC)
    print(f"{lib}: {nst} sweep statements (EXEC windows), {len(ebad)} of them inside a region hipcc treats as divergent "
          f"(dynamically uniform by construction; run-time check: scripts/dev/check_exec.py)")
    per = {}
    for f, a, why in ebad:
        per[f] = per.get(f, 0) + 1
    for f, c in sorted(per.items())[:60]:
        print(f"  EXEC: {c:4d} statements in {f}")
    return 1 if bad or total == 0 else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
