"""The inline-asm DPP instructions of the kernels need two wait states after a VALU write of their
shuffled source; hipcc does not pad inside asm statements, so the built library is checked on its
disassembly (scripts/check_dpp_hazards.py; `_lib.build()` runs the same check and rejects a bad build)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def test_checker_finds_planted_hazards():
    import check_dpp_hazards as c

    n, bad = c.check_disassembly("""
        v_fmac_f64_e32 v[2:3], v[4:5], v[6:7]
        v_mov_b32_dpp v0, v2 wave_shr:1 row_mask:0xf bank_mask:0xf
        v_fmac_f64_e32 v[2:3], v[4:5], v[6:7]
        s_nop 1
        v_mov_b32_dpp v0, v3 wave_shr:1 row_mask:0xf bank_mask:0xf
        v_cndmask_b32_e32 v19, v19, v23, vcc
        v_mov_b32_dpp v10, v18 wave_shr:1 row_mask:0xf bank_mask:0xf
        v_mov_b32_dpp v11, v19 wave_shr:1 row_mask:0xf bank_mask:0xf
        v_add_f32_e32 v7, v1, v2
        v_add_f32_dpp v5, v7, v9 wave_rol:1 row_mask:0xf bank_mask:0xf
        v_add_f32_e32 v9, v1, v2
        v_add_f32_dpp v5, v7, v9 wave_rol:1 row_mask:0xf bank_mask:0xf
    """)
    assert n == 6
    assert [(b[1].split()[1], b[2]) for b in bad] == [("v0,", 0), ("v11,", 1), ("v5,", 0)]


def test_built_library_has_no_dpp_hazard():
    import check_dpp_hazards as c

    lib = os.path.join(ROOT, "sigsvgd_amd", "libsigsvgd_hip.so")
    if not os.path.exists(lib):
        pytest.skip("library not built")
    total, bad = 0, []
    for text in c.disassemble(lib):
        n, b = c.check_disassembly(text)
        total += n
        bad += b
    assert total > 1000, "no DPP instructions found: the disassembly step is broken"
    assert not bad, bad[:5]


def test_exec_region_reader_finds_planted_statements():
    """the informational EXEC reader: a sweep statement (s_mov_b64 exec, -1 + wave shift) inside a saved-EXEC region or a
    divergent loop is reported, one in straight-line code is not"""
    import check_dpp_hazards as c

    text = """
0000000000001000 <kern_a>:
\ts_mov_b64 exec, -1                                         // 000000001000: BEFE01C1
\tv_mov_b32_dpp v1, v2 wave_shr:1 row_mask:0xf bank_mask:0xf   // 000000001004: 7E0202FA FF013802
\ts_and_saveexec_b64 s[4:5], vcc                              // 00000000100C: BE84206A
\ts_mov_b64 exec, -1                                         // 000000001010: BEFE01C1
\tv_mov_b32_dpp v1, v2 wave_shl:1 row_mask:0xf bank_mask:0xf   // 000000001014: 7E0202FA FF013002
\ts_or_b64 exec, exec, s[4:5]                                 // 00000000101C: 87FE047E
\ts_mov_b64 exec, -1                                         // 000000001020: BEFE01C1
\tv_mov_b32_dpp v1, v2 wave_shr:1 row_mask:0xf bank_mask:0xf   // 000000001024: 7E0202FA FF013802
0000000000002000 <kern_b>:
\ts_mov_b64 exec, -1                                         // 000000002000: BEFE01C1
\tv_mov_b32_dpp v1, v2 wave_shr:1 row_mask:0xf bank_mask:0xf   // 000000002004: 7E0202FA FF013802
\ts_andn2_b64 exec, exec, s[6:7]                              // 00000000200C: 89FE067E
\ts_cbranch_execnz 65531                                      // 000000002010: BF89FFFB
"""
    n, bad = c.check_exec_regions(text)
    assert n == 4
    assert [(f, a) for f, a, _ in bad] == [("kern_a", 0x1010), ("kern_b", 0x2000)]
