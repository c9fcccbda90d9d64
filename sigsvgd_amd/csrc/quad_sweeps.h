// Hand-written sweep statements of the 64 x 64-cell quadrants shared by the quadrant kernel (gram_quad.hip: long paths,
// dyadic order 0) and the refined-grid kernel (gram_dyad.hip: short paths with dyadic refinement): four PDE steps per asm
// statement, fp32 difference form, EXEC windows from one scalar shift per step, quadrant hand-over through LDS rows.
// Include inside namespace sigsvgd, in an anonymous namespace.
#pragma once

// ---- four steps of a sweep (cf. gram_fast.hip for the scheduling rules).  Per step: the EXEC window from one scalar
// shift (no table: a quadrant of n cell columns has lanes max(0, sigma - n + 1) .. min(sigma, 63) on anti-diagonal
// sigma, which is `wr` = the top n bits shifted right by 63 - sigma or left by sigma - 63), the DPP shift under full
// EXEC, the stencil under the window AND the quadrant's row mask, then -- still under the window -- the K_fwd slot
// store (forward) / S product (reverse), the boundary value of the NEXT step into the register the next shift
// writes (lane 0 / 63 keeps it: no DPP source), and the hand-over store.  The boundary values of a sweep sit one per
// lane in a VGPR (read from the hand-over row in LDS once, before the sweep) and reach the step through
// v_readlane with a compile-time lane: a sweep has no load and no s_waitcnt at all (the table + LDS version
// exposed an LDS and a scalar-load latency every four steps: 138 cycles per step against 59 in gram_fast).
// The trailing VALU instructions are also the wait states between the write of `cur` and the next DPP.
// (XTRA: between the slot store and the boundary move, which overwrites DIAG -- the condition-estimate accumulation of
//  few-channel forward-only launches, SIG_Q_SD)
#define SIG_Q_FWD_X(SH, KI, W, UP, DIAG, G, K, BNA, XTRA, BNB)                                \
    SH " %[tm], %[" W "], %[" KI "]\n\t"                                                      \
    "s_mov_b64 exec, -1\n\t"                                                                  \
    "v_mov_b32_dpp %[" UP "], %[cur] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" BNA           \
    "s_and_b64 exec, %[tm], %[rows]\n\t"                                                      \
    "v_add_f32 %[t], %[cur], %[" UP "]\n\t"                                                   \
    "v_mul_f32 %[y], %[r3], %[t]\n\t"                                                         \
    "v_add_f32 %[t], %[t], %[" DIAG "]\n\t"                                                   \
    "v_fmac_f32 %[y], %[t], %[" G "]\n\t"                                                     \
    "v_fmac_f32 %[V], %[" G "], %[y]\n\t"                                                     \
    "v_add_f32 %[cur], %[" UP "], %[V]\n\t"                                                   \
    "v_mov_b32 %[" K "], %[" DIAG "]\n\t" XTRA BNB                                            \
    "ds_write_b32 %[ha], %[cur]\n\t"                                                          \
    "v_add_u32 %[ha], %[hinc], %[ha]\n\t"
#define SIG_Q_FWD(SH, KI, W, UP, DIAG, G, K, BNA, BNB) SIG_Q_FWD_X(SH, KI, W, UP, DIAG, G, K, BNA, "", BNB)
// sum over the cells of |K[l][q] * gamma[l][q]|: the forward half of the condition estimate (gram_fast.hip, "conditioning")
#define SIG_Q_SD(DIAG, G) "v_fma_f32 %[sd], |%[" DIAG "]|, |%[" G "]|, %[sd]\n\t"
#define SIG_Q_REV(SH, KI, W, DN, DDIAG, G, K, BNA, BNB)                                       \
    SH " %[tm], %[" W "], %[" KI "]\n\t"                                                      \
    "s_mov_b64 exec, -1\n\t"                                                                  \
    "v_mov_b32_dpp %[" DN "], %[cur] wave_shl:1 row_mask:0xf bank_mask:0xf\n\t" BNA           \
    "s_and_b64 exec, %[tm], %[rows]\n\t"                                                      \
    "v_add_f32 %[t], %[cur], %[" DN "]\n\t"                                                   \
    "v_mul_f32 %[y], %[r3], %[t]\n\t"                                                         \
    "v_add_f32 %[t], %[t], %[" DDIAG "]\n\t"                                                  \
    "v_fmac_f32 %[y], %[t], %[" G "]\n\t"                                                     \
    "v_fmac_f32 %[V], %[" G "], %[y]\n\t"                                                     \
    "v_add_f32 %[cur], %[" DN "], %[V]\n\t"                                                   \
    "v_mul_f32 %[" K "], %[" K "], %[" DDIAG "]\n\t" BNB                                      \
    "ds_write_b32 %[ha], %[cur]\n\t"                                                          \
    "v_add_u32 %[ha], %[hinc], %[ha]\n\t"
// boundary value through a scalar: lane L of the per-sweep boundary register
#define SIG_Q_RL(L) "v_readlane_b32 %[sb], %[hb], %[" L "]\n\t"
#define SIG_Q_BMOV(DIAG) "v_mov_b32 %[" DIAG "], %[sb]\n\t"

// steps S0 .. S0+3 (S0 a multiple of 4).  hb: lane l holds the boundary value lane 0 needs after step l, i.e.
// K[64 b][64 h + l + 2]; lane 0 is outside the window from step 64 on, so the later steps carry none.
// MODE 0: as described; MODE 1 (forward-only launches of paths in <= 3 channels): the steps also accumulate sd += |K00 * gamma|
template <int S0, int MODE = 0>
__device__ __forceinline__ void quad_fwd4(float &cur, float &upA, float &upB, float &V, const float *g, float *ksl,
                                          const unsigned long long wr, const unsigned long long rows, const float hb,
                                          int &ha, const int hinc, const float r3, float &sd)
{
    float t, y;
    unsigned long long tm;
    SIG_EXEC_MUST_BE_FULL("quad_fwd4");
    if constexpr (S0 < 64 && MODE == 0) {
        int sb;
        asm volatile(SIG_Q_FWD("s_lshr_b64", "i0", "wr", "upA", "upB", "g0", "k0", SIG_Q_RL("l0"), SIG_Q_BMOV("upB"))
                     SIG_Q_FWD("s_lshr_b64", "i1", "wr", "upB", "upA", "g1", "k1", SIG_Q_RL("l1"), SIG_Q_BMOV("upA"))
                     SIG_Q_FWD("s_lshr_b64", "i2", "wr", "upA", "upB", "g2", "k2", SIG_Q_RL("l2"), SIG_Q_BMOV("upB"))
                     SIG_Q_FWD("s_lshr_b64", "i3", "wr", "upB", "upA", "g3", "k3", SIG_Q_RL("l3"), SIG_Q_BMOV("upA"))
                     "s_mov_b64 exec, -1\n\t"
                     : [cur] "+v"(cur), [upA] "+v"(upA), [upB] "+v"(upB), [V] "+v"(V), [ha] "+v"(ha), [t] "=&v"(t),
                       [y] "=&v"(y), [k0] "+v"(ksl[0]), [k1] "+v"(ksl[1]), [k2] "+v"(ksl[2]), [k3] "+v"(ksl[3]),
                       [tm] "=&s"(tm), [sb] "=&s"(sb)
                     : [g0] "v"(g[0]), [g1] "v"(g[1]), [g2] "v"(g[2]), [g3] "v"(g[3]), [hb] "v"(hb), [wr] "s"(wr),
                       [rows] "s"(rows), [hinc] "v"(hinc), [r3] "s"(r3), [i0] "n"(63 - S0), [i1] "n"(62 - S0),
                       [i2] "n"(61 - S0), [i3] "n"(60 - S0), [l0] "n"(S0), [l1] "n"(S0 + 1), [l2] "n"(S0 + 2),
                       [l3] "n"(S0 + 3)
                     : "scc");
    } else if constexpr (S0 < 64) {
        int sb;
        asm volatile(SIG_Q_FWD_X("s_lshr_b64", "i0", "wr", "upA", "upB", "g0", "k0", SIG_Q_RL("l0"), SIG_Q_SD("upB", "g0"), SIG_Q_BMOV("upB"))
                     SIG_Q_FWD_X("s_lshr_b64", "i1", "wr", "upB", "upA", "g1", "k1", SIG_Q_RL("l1"), SIG_Q_SD("upA", "g1"), SIG_Q_BMOV("upA"))
                     SIG_Q_FWD_X("s_lshr_b64", "i2", "wr", "upA", "upB", "g2", "k2", SIG_Q_RL("l2"), SIG_Q_SD("upB", "g2"), SIG_Q_BMOV("upB"))
                     SIG_Q_FWD_X("s_lshr_b64", "i3", "wr", "upB", "upA", "g3", "k3", SIG_Q_RL("l3"), SIG_Q_SD("upA", "g3"), SIG_Q_BMOV("upA"))
                     "s_mov_b64 exec, -1\n\t"
                     : [cur] "+v"(cur), [upA] "+v"(upA), [upB] "+v"(upB), [V] "+v"(V), [ha] "+v"(ha), [t] "=&v"(t),
                       [y] "=&v"(y), [k0] "+v"(ksl[0]), [k1] "+v"(ksl[1]), [k2] "+v"(ksl[2]), [k3] "+v"(ksl[3]),
                       [tm] "=&s"(tm), [sb] "=&s"(sb), [sd] "+v"(sd)
                     : [g0] "v"(g[0]), [g1] "v"(g[1]), [g2] "v"(g[2]), [g3] "v"(g[3]), [hb] "v"(hb), [wr] "s"(wr),
                       [rows] "s"(rows), [hinc] "v"(hinc), [r3] "s"(r3), [i0] "n"(63 - S0), [i1] "n"(62 - S0),
                       [i2] "n"(61 - S0), [i3] "n"(60 - S0), [l0] "n"(S0), [l1] "n"(S0 + 1), [l2] "n"(S0 + 2),
                       [l3] "n"(S0 + 3)
                     : "scc");
    } else if constexpr (MODE == 0) {
        const unsigned long long wl = (S0 + 3 == 127) ? 0ull : wr; // anti-diagonal 127 has no cell (a shift by 64 is one by 0)
        asm volatile(SIG_Q_FWD("s_lshl_b64", "i0", "wr", "upA", "upB", "g0", "k0", "", "")
                     SIG_Q_FWD("s_lshl_b64", "i1", "wr", "upB", "upA", "g1", "k1", "", "")
                     SIG_Q_FWD("s_lshl_b64", "i2", "wr", "upA", "upB", "g2", "k2", "", "")
                     SIG_Q_FWD("s_lshl_b64", "i3", "wl", "upB", "upA", "g3", "k3", "", "")
                     "s_mov_b64 exec, -1\n\t"
                     : [cur] "+v"(cur), [upA] "+v"(upA), [upB] "+v"(upB), [V] "+v"(V), [ha] "+v"(ha), [t] "=&v"(t),
                       [y] "=&v"(y), [k0] "+v"(ksl[0]), [k1] "+v"(ksl[1]), [k2] "+v"(ksl[2]), [k3] "+v"(ksl[3]),
                       [tm] "=&s"(tm)
                     : [g0] "v"(g[0]), [g1] "v"(g[1]), [g2] "v"(g[2]), [g3] "v"(g[3]), [wr] "s"(wr), [wl] "s"(wl),
                       [rows] "s"(rows), [hinc] "v"(hinc), [r3] "s"(r3), [i0] "n"(S0 - 63), [i1] "n"(S0 - 62),
                       [i2] "n"(S0 - 61), [i3] "n"((S0 - 60) & 63)
                     : "scc");
    } else {
        const unsigned long long wl = (S0 + 3 == 127) ? 0ull : wr;
        asm volatile(SIG_Q_FWD_X("s_lshl_b64", "i0", "wr", "upA", "upB", "g0", "k0", "", SIG_Q_SD("upB", "g0"), "")
                     SIG_Q_FWD_X("s_lshl_b64", "i1", "wr", "upB", "upA", "g1", "k1", "", SIG_Q_SD("upA", "g1"), "")
                     SIG_Q_FWD_X("s_lshl_b64", "i2", "wr", "upA", "upB", "g2", "k2", "", SIG_Q_SD("upB", "g2"), "")
                     SIG_Q_FWD_X("s_lshl_b64", "i3", "wl", "upB", "upA", "g3", "k3", "", SIG_Q_SD("upA", "g3"), "")
                     "s_mov_b64 exec, -1\n\t"
                     : [cur] "+v"(cur), [upA] "+v"(upA), [upB] "+v"(upB), [V] "+v"(V), [ha] "+v"(ha), [t] "=&v"(t),
                       [y] "=&v"(y), [k0] "+v"(ksl[0]), [k1] "+v"(ksl[1]), [k2] "+v"(ksl[2]), [k3] "+v"(ksl[3]),
                       [tm] "=&s"(tm), [sd] "+v"(sd)
                     : [g0] "v"(g[0]), [g1] "v"(g[1]), [g2] "v"(g[2]), [g3] "v"(g[3]), [wr] "s"(wr), [wl] "s"(wl),
                       [rows] "s"(rows), [hinc] "v"(hinc), [r3] "s"(r3), [i0] "n"(S0 - 63), [i1] "n"(S0 - 62),
                       [i2] "n"(S0 - 61), [i3] "n"((S0 - 60) & 63)
                     : "scc");
    }
}
// steps S0+3 .. S0 (descending).  hb: lane l holds the boundary value lane 63 needs after step l + 64, i.e.
// U[64 b + 64][64 h + l]; bm: the one after step 63 (a right quadrant hands lane 63 the first value of the left one);
// lane 63 is outside the window below step 63.
template <int S0>
__device__ __forceinline__ void quad_rev4(float &cur, float &dnA, float &dnB, float &V, const float *g, float *ksl,
                                          const unsigned long long wr, const unsigned long long rows, const float hb,
                                          const float bm, int &ha, const int hinc, const float r3)
{
    float t, y;
    unsigned long long tm;
    SIG_EXEC_MUST_BE_FULL("quad_rev4");
    if constexpr (S0 >= 64) {
        int sb;
        const unsigned long long wl = (S0 + 3 == 127) ? 0ull : wr;
        asm volatile(SIG_Q_REV("s_lshl_b64", "i3", "wl", "dnA", "dnB", "g3", "k3", SIG_Q_RL("l3"), SIG_Q_BMOV("dnB"))
                     SIG_Q_REV("s_lshl_b64", "i2", "wr", "dnB", "dnA", "g2", "k2", SIG_Q_RL("l2"), SIG_Q_BMOV("dnA"))
                     SIG_Q_REV("s_lshl_b64", "i1", "wr", "dnA", "dnB", "g1", "k1", SIG_Q_RL("l1"), SIG_Q_BMOV("dnB"))
                     SIG_Q_REV("s_lshl_b64", "i0", "wr", "dnB", "dnA", "g0", "k0", SIG_Q_RL("l0"), SIG_Q_BMOV("dnA"))
                     "s_mov_b64 exec, -1\n\t"
                     : [cur] "+v"(cur), [dnA] "+v"(dnA), [dnB] "+v"(dnB), [V] "+v"(V), [ha] "+v"(ha), [t] "=&v"(t),
                       [y] "=&v"(y), [k0] "+v"(ksl[0]), [k1] "+v"(ksl[1]), [k2] "+v"(ksl[2]), [k3] "+v"(ksl[3]),
                       [tm] "=&s"(tm), [sb] "=&s"(sb)
                     : [g0] "v"(g[0]), [g1] "v"(g[1]), [g2] "v"(g[2]), [g3] "v"(g[3]), [hb] "v"(hb), [wr] "s"(wr),
                       [wl] "s"(wl), [rows] "s"(rows), [hinc] "v"(hinc), [r3] "s"(r3), [i0] "n"(S0 - 63),
                       [i1] "n"(S0 - 62), [i2] "n"(S0 - 61), [i3] "n"((S0 - 60) & 63), [l0] "n"(S0 - 64),
                       [l1] "n"(S0 - 63), [l2] "n"(S0 - 62), [l3] "n"(S0 - 61)
                     : "scc");
    } else if constexpr (S0 == 60) {
        asm volatile(SIG_Q_REV("s_lshr_b64", "i3", "wr", "dnA", "dnB", "g3", "k3", "", "v_mov_b32 %[dnB], %[bm]\n\t")
                     SIG_Q_REV("s_lshr_b64", "i2", "wr", "dnB", "dnA", "g2", "k2", "", "")
                     SIG_Q_REV("s_lshr_b64", "i1", "wr", "dnA", "dnB", "g1", "k1", "", "")
                     SIG_Q_REV("s_lshr_b64", "i0", "wr", "dnB", "dnA", "g0", "k0", "", "")
                     "s_mov_b64 exec, -1\n\t"
                     : [cur] "+v"(cur), [dnA] "+v"(dnA), [dnB] "+v"(dnB), [V] "+v"(V), [ha] "+v"(ha), [t] "=&v"(t),
                       [y] "=&v"(y), [k0] "+v"(ksl[0]), [k1] "+v"(ksl[1]), [k2] "+v"(ksl[2]), [k3] "+v"(ksl[3]),
                       [tm] "=&s"(tm)
                     : [g0] "v"(g[0]), [g1] "v"(g[1]), [g2] "v"(g[2]), [g3] "v"(g[3]), [bm] "v"(bm), [wr] "s"(wr),
                       [rows] "s"(rows), [hinc] "v"(hinc), [r3] "s"(r3), [i0] "n"(63 - S0), [i1] "n"(62 - S0),
                       [i2] "n"(61 - S0), [i3] "n"(60 - S0)
                     : "scc");
    } else {
        asm volatile(SIG_Q_REV("s_lshr_b64", "i3", "wr", "dnA", "dnB", "g3", "k3", "", "")
                     SIG_Q_REV("s_lshr_b64", "i2", "wr", "dnB", "dnA", "g2", "k2", "", "")
                     SIG_Q_REV("s_lshr_b64", "i1", "wr", "dnA", "dnB", "g1", "k1", "", "")
                     SIG_Q_REV("s_lshr_b64", "i0", "wr", "dnB", "dnA", "g0", "k0", "", "")
                     "s_mov_b64 exec, -1\n\t"
                     : [cur] "+v"(cur), [dnA] "+v"(dnA), [dnB] "+v"(dnB), [V] "+v"(V), [ha] "+v"(ha), [t] "=&v"(t),
                       [y] "=&v"(y), [k0] "+v"(ksl[0]), [k1] "+v"(ksl[1]), [k2] "+v"(ksl[2]), [k3] "+v"(ksl[3]),
                       [tm] "=&s"(tm)
                     : [g0] "v"(g[0]), [g1] "v"(g[1]), [g2] "v"(g[2]), [g3] "v"(g[3]), [wr] "s"(wr), [rows] "s"(rows),
                       [hinc] "v"(hinc), [r3] "s"(r3), [i0] "n"(63 - S0), [i1] "n"(62 - S0), [i2] "n"(61 - S0),
                       [i3] "n"(60 - S0)
                     : "scc");
    }
}
// the unrolled sweeps (the group index has to be a compile-time constant for the shift and lane immediates).
// `slast` = last anti-diagonal with a cell + 2 (rows + columns of the quadrant): a quadrant of fewer than 64 x 64 cells has
// empty EXEC windows beyond it, and the groups of four steps that lie entirely beyond are skipped (uniform branches).  The two
// steps right after the last active one are kept: their DPP shifts refresh both neighbour registers with the lanes' final
// values, which the quadrant swept next in the same band takes as its first diagonal operands.
// EARLY = false compiles the tests out (quadrants that are always (nearly) full: the tests cost them 3 %).
template <int S0, bool EARLY, int MODE = 0>
__device__ __forceinline__ void quad_fwd_all(float &cur, float &upA, float &upB, float &V, const float *D, float *S,
                                             const unsigned long long wr, const unsigned long long rows, const float hb,
                                             int &ha, const int hinc, const float r3, const int slast, float &sd)
{
    quad_fwd4<S0, MODE>(cur, upA, upB, V, D + (S0 & 63), S + (S0 & 63), wr, rows, hb, ha, hinc, r3, sd);
    if constexpr (S0 + 4 < 128) { // (tested every 16 steps: a test per group of four costs a full quadrant 1.5 %)
        if (!EARLY || ((S0 + 4) & 15) != 0 || S0 + 4 <= slast)
            quad_fwd_all<S0 + 4, EARLY, MODE>(cur, upA, upB, V, D, S, wr, rows, hb, ha, hinc, r3, slast, sd);
    }
}
template <int S0, bool EARLY>
__device__ __forceinline__ void quad_rev_all(float &cur, float &dnA, float &dnB, float &V, const float *D, float *S,
                                             const unsigned long long wr, const unsigned long long rows, const float hb,
                                             const float bm, int &ha, const int hinc, const float r3, const int slast)
{
    if (!EARLY || (S0 & ~15) <= slast) quad_rev4<S0>(cur, dnA, dnB, V, D + (S0 & 63), S + (S0 & 63), wr, rows, hb, bm, ha, hinc, r3);
    if constexpr (S0 >= 4) quad_rev_all<S0 - 4, EARLY>(cur, dnA, dnB, V, D, S, wr, rows, hb, bm, ha, hinc, r3, slast);
}
