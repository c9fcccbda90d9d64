// Register-resident signature-kernel Gram forward/backward for the headline shapes:
// dyadic order 0, path length T <= 64, RBF static kernel, second-order stencil.
//
// Mapping (one wavefront per trajectory pair, as BASELINE.json north_star asks):
//   * lane l owns row l of the T x T static-kernel matrix G, row l of the (T-1)^2 increment matrix D,
//     row l+1 of the forward PDE solution K and row l of the reverse solution U;
//   * on anti-diagonal sigma (= row + column) every lane works on column sigma - l, so every
//     per-cell quantity a lane will need again (D, K_fwd / S, G) is filed under slot sigma & 63:
//     the slot index is a compile-time constant of the (fully unrolled) step, the same for all
//     lanes, so D and K_fwd live in 2 x 64 VGPRs and G in a [slot][lane] LDS image;
//   * neighbour rows are reached with wave-wide DPP shifts (no LDS round trip in the recurrence);
//   * the static kernel and the 4-corner increments run in fp64 (the increments cancel ~1e-2 of G); the two PDE
//     sweeps run in fp32 in DIFFERENCE FORM (a lane carries V = K[l+1][q] - K[l][q] along its row, see phase 2:
//     1.5e-7 on K, 1.9e-7 on the gradient against the fp64 oracle, where the plain fp32 stencil loses 1e-5);
//     everything that is only stored or contracted (D, K_fwd, G, S = K_fwd*U, R, gradient sums per pair) is fp32,
//     the reduction over pairs is fp64 (row side in registers, column side in grad_reduce_kernel).
//
// Per pair: phase 1 static kernel + increments (wrap-around skew, all lanes busy, 66 iterations);
// phase 2 forward sweep (K_fwd into the slots); phase 3 reverse sweep (U recurrence only, S = K_fwd*U
// overwrites K_fwd in its slot); phase 4 scatter R, R*G and both contractions in a second wrap-around
// pass (66 iterations, all lanes busy).  The two sweeps have on average half of their lanes outside the
// grid: their steps are hand-written (8 VALU + 2 SALU) and the idle lanes are switched off through EXEC windows
// read from a constant table.  The kernel is bound by vector-instruction issue at two waves per SIMD, i.e. by
// its instruction count (5,430 per pair at C4): see DESIGN.md 5.1 / 5.1.1.
//
// A workgroup is NW wavefronts = NW consecutive rows i of X against a chunk of columns j; the
// column trajectory (centred on its first point, fp64 + fp32 copies) is staged once per j in LDS
// and shared by the NW pairs.  With Y == X each unordered pair {i<j} is solved once: the row-side
// contraction gives d k(x_i,x_j)/d x_i, the column-side sums (travelling accumulators, one wave
// rotation per sum and iteration) give d k(x_j,x_i)/d x_j.  Row-side gradients stay in per-lane fp64
// registers while a workgroup works on a row tile of its item range; column-side results of the NW waves
// are summed in fixed order through LDS.  Both leave the kernel through plain stores into per-segment /
// per-item slabs that grad_reduce_kernel adds up in a fixed order: no atomics, bit-reproducible results.
//
// Reference semantics: sigkernel _SigKernelGram.forward/backward [RECALLED, SURVEY.md App. A];
// static kernel src/kernels/_traj_kernels.py:176-195; callers src/inference/score.py:68-69.
#include <atomic>

#include "sig_common.h"
#ifdef SIGSVGD_PHASE_STAMPS
#include <cstdio>
#endif

namespace sigsvgd {

// A wave's priority falls as it advances through its pair (static kernel 3, forward sweep 2, reverse sweep 1, gradient
// pass 0), so that the wave that is behind on a SIMD gets the issue slots and the waves stay in complementary phases.
// At C4: ordered gradient launches 8.47 -> 8.12 ms, forward-only 5.51 -> 5.12 ordered and 2.76 -> 2.59 symmetric.
// Symmetric gradient launches lose 1 % with it at two waves per SIMD (their gradient pass is twice as long) and keep the
// default arbitration there; at three waves per SIMD (paths of <= 32 points) they gain 3-8 % and use it; other priority patterns and a fixed priority for the younger half of the
// workgroup changed them by less than 0.5 %.
#define SIG_PRIO(n)                                                          \
    if (!GRAD || !SYM || (RING == 32 && NW == 4)) __builtin_amdgcn_s_setprio(n);

struct FastArgs {
    const void *X, *Y, *go;
    void *K;
    // Gradient partial sums leave the kernel through plain stores into slabs that the reduction kernel below adds up in
    // a FIXED order (no atomics: the result is bit-reproducible run to run, and nothing needs zeroing):
    double *rseg; // [owned tiles + workgroups][NW][T*d] fp64: row-side sums of one (workgroup, row tile) segment
    float *cslab; // [items][T*d] fp32: column-side sums of one (row tile, column) item (symmetric launches)
    int io64, A, B, T, d, symw;
    TileMap tm;                   // row tiles owned by this launch, in the order of the enumeration (multi-GPU sharding)
    long long nitems;             // (owned row tile, column) items of the launch
    double inv_h;
    unsigned char *kflag;         // [A][B] (d <= 4 only): 1 where the pair's fp32 solution cancelled or K is ill-conditioned in
                                  // the increments: the launcher lets the coverage kernel solve those pairs exactly (fp64)
#ifdef SIGSVGD_PHASE_STAMPS
    unsigned long long *stamps; // diagnostic build only: [8] shader-clock totals per phase, summed over waves
#endif
};

// Diagnostic build (-DSIGSVGD_PHASE_STAMPS, scripts/dev/phase_stamps.py): s_memtime around the phases of a pair,
// summed per wave in scalar registers and added to a buffer no output depends on.  Compiled out of the product.
#ifdef SIGSVGD_PHASE_STAMPS
#define SIG_STAMP(i)                                                         \
    {                                                                        \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();        \
        ph_[i] += now_ - tlast_;                                             \
        tlast_ = now_;                                                       \
    }
#else
#define SIG_STAMP(i)
#endif

// ---- wave-wide shifts on the DPP path ----------------------------------------------------------
// wave_shr:1 (0x138): lane l reads lane l-1; wave_shl:1 (0x130): lane l reads lane l+1.
// With bound_ctrl = 0 a lane without a source keeps `old`, which carries the PDE boundary value.
__device__ __forceinline__ int dpp_shr1_i(int v, int bound)
{
    return __builtin_amdgcn_update_dpp(bound, v, 0x138, 0xF, 0xF, false);
}
__device__ __forceinline__ int dpp_shl1_i(int v, int bound)
{
    return __builtin_amdgcn_update_dpp(bound, v, 0x130, 0xF, 0xF, false);
}
// bound_ctrl = 1 forms: a lane without a source gets 0 and no `old` register has to be initialised.
__device__ __forceinline__ int dpp_shr1_z(int v) { return __builtin_amdgcn_mov_dpp(v, 0x138, 0xF, 0xF, true); }
__device__ __forceinline__ int dpp_shl1_z(int v) { return __builtin_amdgcn_mov_dpp(v, 0x130, 0xF, 0xF, true); }
// shifts carrying the PDE boundary value 1.0 = {hi 0x3FF00000, lo 0}: only the high word needs `old`
__device__ __forceinline__ double dpp_shr1_one(double v)
{
    const int lo = dpp_shr1_z(__double2loint(v));
    const int hi = dpp_shr1_i(__double2hiint(v), 0x3FF00000);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_shl1_one(double v)
{
    const int lo = dpp_shl1_z(__double2loint(v));
    const int hi = dpp_shl1_i(__double2hiint(v), 0x3FF00000);
    return __hiloint2double(hi, lo);
}
// Same shifts with the 1.0 boundary kept in a PERSISTENT destination: the lane without a source
// (0 for shr, 63 for shl) is never written, so once `dst` holds 1.0 there it stays -- no per-step
// re-initialisation of the DPP `old` operand.  (asm: hipcc would re-materialise `old` every step.)
// Hazard note (gfx9: VALU write of a VGPR -> DPP read of it needs 2 wait states; hipcc pads its own
// instructions, not the inside of an asm statement): the DPP source is the stencil result of the step
// before, and whether two instructions separate them is hipcc's scheduling decision, not a property of
// this source.  `scripts/check_dpp_hazards.py` therefore checks every DPP instruction of the built
// library on its disassembly; `_lib.build()` runs it and refuses a library that violates the rule, and
// tests/test_dpp_hazards.py runs it again in the CPU suite.  (An `s_nop 1` inside the asm string would
// also do, at ~5 % of the launch: measured.)
__device__ __forceinline__ void dpp_shr1_keep(double &dst, double v)
{
    int dlo = __double2loint(dst), dhi = __double2hiint(dst);
    asm("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(dlo) : "v"(__double2loint(v)));
    asm("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(dhi) : "v"(__double2hiint(v)));
    dst = __hiloint2double(dhi, dlo);
}
__device__ __forceinline__ void dpp_shl1_keep(double &dst, double v)
{
    int dlo = __double2loint(dst), dhi = __double2hiint(dst);
    asm("v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(dlo) : "v"(__double2loint(v)));
    asm("v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(dhi) : "v"(__double2hiint(v)));
    dst = __hiloint2double(dhi, dlo);
}
// fp32 forms of the persistent-boundary shifts (the PDE sweeps run in fp32, see the kernel header)
__device__ __forceinline__ void dpp_shr1_keep(float &dst, float v)
{
    asm("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(dst) : "v"(v));
}
__device__ __forceinline__ void dpp_shl1_keep(float &dst, float v)
{
    asm("v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(dst) : "v"(v));
}
__device__ __forceinline__ double dpp_shl1_zero(double v)
{
    return __hiloint2double(dpp_shl1_z(__double2hiint(v)), dpp_shl1_z(__double2loint(v)));
}
// wave_rol:1 (0x134): lane l reads lane l+1, lane 63 reads lane 0 (full-wave rotation)
__device__ __forceinline__ float dpp_rol1(float v)
{
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x134, 0xF, 0xF, false));
}
// rotate-and-add in ONE VALU instruction: returns acc[lane+1] + v.  (hipcc does not fold a wave_rol
// v_mov_b32_dpp into the add.)  `acc` is the shuffled source: same hazard, same build-time check as above.
__device__ __forceinline__ float add_rol1(float acc, float v)
{
    float out;
    asm("v_add_f32_dpp %0, %1, %2 wave_rol:1 row_mask:0xf bank_mask:0xf" : "=v"(out) : "v"(acc), "v"(v));
    return out;
}
__device__ __forceinline__ float dpp_shr1_zero(float v) { return __int_as_float(dpp_shr1_z(__float_as_int(v))); }

// ---- one step of a PDE sweep, hand-scheduled (fp32 difference form, see phase 2 in the kernel) ------------------
// Every instruction of a step is listed here because the properties that make it fast are scheduling properties.
// Lanes outside the grid on a step (about half of them, averaged over a sweep) are switched off through EXEC, not
// through v_cmp / v_cndmask: the window of active lanes of anti-diagonal sigma, rows max(0, sigma-P+1) ..
// min(sigma, P-1), is a 64-bit constant per (P, sigma), read from a table in constant memory with scalar loads
// (SWEEP_MASK below) and moved into EXEC with one s_mov_b64.  Measured on MI355X at the kernel's occupancy
// (scripts/micro/exec_step.hip, two waves per SIMD): the v_cmp + 2 x v_cndmask step (11 VALU) 20.2 ns per step and
// wave, the EXEC step (8 VALU + 2 SALU) 10.4-11.0 ns -- the scalar moves issue in the shadow of the other wave's
// vector instructions, while the VCC round trip of the compare form stalls the chain far beyond its three
// instructions.  Per step:
//   s_mov exec,-1 ; DPP shift of `cur` into UP for EVERY lane (lane l reads l-1; a lane that has not started reads the
//     1.0 its neighbour still holds, so its diagonal operand is right when it starts; lane 0 keeps the boundary 1.0)
//   s_mov exec,window ; the stencil in difference form (5 VALU), the K_fwd slot store, only for lanes inside the grid:
//     a lane before its row starts keeps cur = 1, a finished lane keeps its last value (lane P-1: K[P,P]), and a
//     slot without a grid cell is never written (phase 4 relies on their zeros).
// The instruction after `v_add cur` and the s_mov are the 2 wait states of the gfx9 VALU-write -> DPP-read hazard
// (hipcc does not pad inside asm; scripts/check_dpp_hazards.py checks the built library).
// Eight steps form ONE asm statement (hipcc pads every asm statement with an `s_nop 0`, an issue slot of its own);
// EXEC is all ones again when the statement ends (the code around it may reload spilled registers).
struct SweepMasks {
    unsigned long long m[64][128]; // [P][sigma]
    constexpr SweepMasks() : m()
    {
        for (int P = 1; P < 64; ++P)
            for (int s = 0; s <= 2 * P - 2; ++s) {
                const int lo = s - P + 1 > 0 ? s - P + 1 : 0, hi = s < P - 1 ? s : P - 1;
                unsigned long long w = 0;
                for (int l = lo; l <= hi; ++l) w |= 1ull << l;
                m[P][s] = w;
            }
    }
};
__constant__ const SweepMasks SWEEP_MASK = SweepMasks();
// The same for paths of <= 32 points on the 32-slot ring (RING = 32 below): lanes 32..63 mirror lanes 0..31 (they own
// the same rows), so every window appears in both halves.
struct SweepMasks32 {
    unsigned long long m[32][64]; // [P][sigma]
    constexpr SweepMasks32() : m()
    {
        for (int P = 1; P < 32; ++P)
            for (int s = 0; s <= 2 * P - 2; ++s) {
                const int lo = s - P + 1 > 0 ? s - P + 1 : 0, hi = s < P - 1 ? s : P - 1;
                unsigned long long w = 0;
                for (int l = lo; l <= hi; ++l) w |= (1ull << l) | (1ull << (l + 32));
                m[P][s] = w;
            }
    }
};
__constant__ const SweepMasks32 SWEEP_MASK32 = SweepMasks32();

#define SIG_FWD_STEP(UP, DIAG, G, KSL, M)                                                     \
    "s_mov_b64 exec, -1\n\t"                                                                  \
    "v_mov_b32_dpp %[" UP "], %[cur] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"               \
    "s_mov_b64 exec, %[" M "]\n\t"                                                            \
    "v_add_f32 %[t], %[cur], %[" UP "]\n\t"                                                   \
    "v_mul_f32 %[y], %[r3], %[t]\n\t"                                                         \
    "v_add_f32 %[t], %[t], %[" DIAG "]\n\t"                                                   \
    "v_fmac_f32 %[y], %[t], %[" G "]\n\t"                                                     \
    "v_fmac_f32 %[V], %[" G "], %[y]\n\t"                                                     \
    "v_add_f32 %[cur], %[" UP "], %[V]\n\t" KSL
#define SIG_FWD_KSL(DIAG, K) "v_mov_b32 %[" K "], %[" DIAG "]\n\t"
#define SIG_FWD_NOKSL "v_max_f32 %[km], |%[cur]|, %[km]\n\t" // (forward-only launches: the largest |K| of the grid, in the hazard slot)
// few-channel forward-only launches also sum |K[l][q] * gamma[l][q]| over the cells: the forward half of the condition
// estimate sum |S * D| / |K| that decides whether the pair goes to the exact fp64 pass (see the kernel, "conditioning")
#define SIG_FWD_NOKSL_SD(DIAG, G) "v_max_f32 %[km], |%[cur]|, %[km]\n\tv_fma_f32 %[sd], |%[" DIAG "]|, |%[" G "]|, %[sd]\n\t"
#define SIG_REV_STEP(DN, DDIAG, G, K, M)                                                      \
    "s_mov_b64 exec, -1\n\t"                                                                  \
    "v_mov_b32_dpp %[" DN "], %[cur] wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"               \
    "s_mov_b64 exec, %[" M "]\n\t"                                                            \
    "v_add_f32 %[t], %[cur], %[" DN "]\n\t"                                                   \
    "v_mul_f32 %[y], %[r3], %[t]\n\t"                                                         \
    "v_add_f32 %[t], %[t], %[" DDIAG "]\n\t"                                                  \
    "v_fmac_f32 %[y], %[t], %[" G "]\n\t"                                                     \
    "v_fmac_f32 %[V], %[" G "], %[y]\n\t"                                                     \
    "v_add_f32 %[cur], %[" DN "], %[V]\n\t"                                                   \
    "v_mul_f32 %[" K "], %[" K "], %[" DDIAG "]\n\t"

// steps sigma0 .. sigma0+7 of the forward sweep (sigma0 even): g, ksl point at slots sigma0 & 63 ..; mk at the
// windows of sigma0 ..
// MODE 0: K_fwd into the slots (gradient launches); 1: forward only, row maximum; 2: forward only, row maximum and the
// sum of |K00 * gamma| (few-channel launches)
template <int MODE>
__device__ __forceinline__ void sweep_fwd8(float &cur, float &upA, float &upB, float &V, const float *g, float *ksl,
                                           const unsigned long long *mk, const float r3, float &km, float &sd)
{
    float t, y;
    SIG_EXEC_MUST_BE_FULL("sweep_fwd8");
    const unsigned long long m0 = mk[0], m1 = mk[1], m2 = mk[2], m3 = mk[3], m4 = mk[4], m5 = mk[5], m6 = mk[6], m7 = mk[7];
    if constexpr (MODE == 2)
        asm volatile(SIG_FWD_STEP("upA", "upB", "g0", SIG_FWD_NOKSL_SD("upB", "g0"), "m0") SIG_FWD_STEP("upB", "upA", "g1", SIG_FWD_NOKSL_SD("upA", "g1"), "m1")
                     SIG_FWD_STEP("upA", "upB", "g2", SIG_FWD_NOKSL_SD("upB", "g2"), "m2") SIG_FWD_STEP("upB", "upA", "g3", SIG_FWD_NOKSL_SD("upA", "g3"), "m3")
                     SIG_FWD_STEP("upA", "upB", "g4", SIG_FWD_NOKSL_SD("upB", "g4"), "m4") SIG_FWD_STEP("upB", "upA", "g5", SIG_FWD_NOKSL_SD("upA", "g5"), "m5")
                     SIG_FWD_STEP("upA", "upB", "g6", SIG_FWD_NOKSL_SD("upB", "g6"), "m6") SIG_FWD_STEP("upB", "upA", "g7", SIG_FWD_NOKSL_SD("upA", "g7"), "m7")
                     "s_mov_b64 exec, -1\n\t"
                     : [cur] "+v"(cur), [upA] "+v"(upA), [upB] "+v"(upB), [V] "+v"(V), [t] "=&v"(t), [y] "=&v"(y), [km] "+v"(km), [sd] "+v"(sd)
                     : [g0] "v"(g[0]), [g1] "v"(g[1]), [g2] "v"(g[2]), [g3] "v"(g[3]), [g4] "v"(g[4]), [g5] "v"(g[5]),
                       [g6] "v"(g[6]), [g7] "v"(g[7]), [r3] "s"(r3), [m0] "s"(m0), [m1] "s"(m1), [m2] "s"(m2),
                       [m3] "s"(m3), [m4] "s"(m4), [m5] "s"(m5), [m6] "s"(m6), [m7] "s"(m7));
    else if constexpr (MODE == 0)
        asm volatile(SIG_FWD_STEP("upA", "upB", "g0", SIG_FWD_KSL("upB", "k0"), "m0")
                     SIG_FWD_STEP("upB", "upA", "g1", SIG_FWD_KSL("upA", "k1"), "m1")
                     SIG_FWD_STEP("upA", "upB", "g2", SIG_FWD_KSL("upB", "k2"), "m2")
                     SIG_FWD_STEP("upB", "upA", "g3", SIG_FWD_KSL("upA", "k3"), "m3")
                     SIG_FWD_STEP("upA", "upB", "g4", SIG_FWD_KSL("upB", "k4"), "m4")
                     SIG_FWD_STEP("upB", "upA", "g5", SIG_FWD_KSL("upA", "k5"), "m5")
                     SIG_FWD_STEP("upA", "upB", "g6", SIG_FWD_KSL("upB", "k6"), "m6")
                     SIG_FWD_STEP("upB", "upA", "g7", SIG_FWD_KSL("upA", "k7"), "m7")
                     "s_mov_b64 exec, -1\n\t"
                     : [cur] "+v"(cur), [upA] "+v"(upA), [upB] "+v"(upB), [V] "+v"(V), [t] "=&v"(t), [y] "=&v"(y),
                       [k0] "+v"(ksl[0]), [k1] "+v"(ksl[1]), [k2] "+v"(ksl[2]), [k3] "+v"(ksl[3]), [k4] "+v"(ksl[4]),
                       [k5] "+v"(ksl[5]), [k6] "+v"(ksl[6]), [k7] "+v"(ksl[7])
                     : [g0] "v"(g[0]), [g1] "v"(g[1]), [g2] "v"(g[2]), [g3] "v"(g[3]), [g4] "v"(g[4]), [g5] "v"(g[5]),
                       [g6] "v"(g[6]), [g7] "v"(g[7]), [r3] "s"(r3), [m0] "s"(m0), [m1] "s"(m1), [m2] "s"(m2),
                       [m3] "s"(m3), [m4] "s"(m4), [m5] "s"(m5), [m6] "s"(m6), [m7] "s"(m7));
    else
        asm volatile(SIG_FWD_STEP("upA", "upB", "g0", SIG_FWD_NOKSL, "m0") SIG_FWD_STEP("upB", "upA", "g1", SIG_FWD_NOKSL, "m1")
                     SIG_FWD_STEP("upA", "upB", "g2", SIG_FWD_NOKSL, "m2") SIG_FWD_STEP("upB", "upA", "g3", SIG_FWD_NOKSL, "m3")
                     SIG_FWD_STEP("upA", "upB", "g4", SIG_FWD_NOKSL, "m4") SIG_FWD_STEP("upB", "upA", "g5", SIG_FWD_NOKSL, "m5")
                     SIG_FWD_STEP("upA", "upB", "g6", SIG_FWD_NOKSL, "m6") SIG_FWD_STEP("upB", "upA", "g7", SIG_FWD_NOKSL, "m7")
                     "s_mov_b64 exec, -1\n\t"
                     : [cur] "+v"(cur), [upA] "+v"(upA), [upB] "+v"(upB), [V] "+v"(V), [t] "=&v"(t), [y] "=&v"(y), [km] "+v"(km)
                     : [g0] "v"(g[0]), [g1] "v"(g[1]), [g2] "v"(g[2]), [g3] "v"(g[3]), [g4] "v"(g[4]), [g5] "v"(g[5]),
                       [g6] "v"(g[6]), [g7] "v"(g[7]), [r3] "s"(r3), [m0] "s"(m0), [m1] "s"(m1), [m2] "s"(m2),
                       [m3] "s"(m3), [m4] "s"(m4), [m5] "s"(m5), [m6] "s"(m6), [m7] "s"(m7));
}
// steps sigma0+7 .. sigma0 of the reverse sweep (descending; sigma0 even): lower neighbour through wave_shl,
// S = K_fwd * U[l+1][q+1] replaces K_fwd in its slot
__device__ __forceinline__ void sweep_rev8(float &cur, float &dnA, float &dnB, float &V, const float *g, float *ksl,
                                           const unsigned long long *mk, const float r3)
{
    float t, y;
    SIG_EXEC_MUST_BE_FULL("sweep_rev8");
    const unsigned long long m0 = mk[0], m1 = mk[1], m2 = mk[2], m3 = mk[3], m4 = mk[4], m5 = mk[5], m6 = mk[6], m7 = mk[7];
    asm volatile(SIG_REV_STEP("dnA", "dnB", "g7", "k7", "m7") SIG_REV_STEP("dnB", "dnA", "g6", "k6", "m6")
                 SIG_REV_STEP("dnA", "dnB", "g5", "k5", "m5") SIG_REV_STEP("dnB", "dnA", "g4", "k4", "m4")
                 SIG_REV_STEP("dnA", "dnB", "g3", "k3", "m3") SIG_REV_STEP("dnB", "dnA", "g2", "k2", "m2")
                 SIG_REV_STEP("dnA", "dnB", "g1", "k1", "m1") SIG_REV_STEP("dnB", "dnA", "g0", "k0", "m0")
                 "s_mov_b64 exec, -1\n\t"
                 : [cur] "+v"(cur), [dnA] "+v"(dnA), [dnB] "+v"(dnB), [V] "+v"(V), [t] "=&v"(t), [y] "=&v"(y),
                   [k0] "+v"(ksl[0]), [k1] "+v"(ksl[1]), [k2] "+v"(ksl[2]), [k3] "+v"(ksl[3]), [k4] "+v"(ksl[4]),
                   [k5] "+v"(ksl[5]), [k6] "+v"(ksl[6]), [k7] "+v"(ksl[7])
                 : [g0] "v"(g[0]), [g1] "v"(g[1]), [g2] "v"(g[2]), [g3] "v"(g[3]), [g4] "v"(g[4]), [g5] "v"(g[5]),
                   [g6] "v"(g[6]), [g7] "v"(g[7]), [r3] "s"(r3), [m0] "s"(m0), [m1] "s"(m1), [m2] "s"(m2), [m3] "s"(m3),
                   [m4] "s"(m4), [m5] "s"(m5), [m6] "s"(m6), [m7] "s"(m7));
}

// ---- fp64 forward sweep of a pair whose fp32 solution cancelled ----------------------------------------------------
// The fp32 sweeps resolve K to ~1e-7 of the LARGEST value on the pair's grid.  Where the discrete solution oscillates
// (rough paths in few channels against a narrow static kernel: K(x, y) passes through zero, turns negative), the value at
// the far corner can be a small remainder of much larger intermediate values and its relative error grows by their ratio
// (measured: up to 1.9e-5 per entry at T = 64, d = 2, against 1e-6 without cancellation).  A wavefront that sees
// max |K_grid| > 4 * max(|K[P][P]|, 0.1) in one of its pairs therefore repeats the forward sweep in fp64 from the
// increments it still holds (fp32 storage of fp64 differences: 2e-8) and stores that value over the first; the gradient
// keeps the fp32 solution (its error is relative to the largest gradient entry: 1.4e-6 in those regimes).  Fully unrolled
// (the slots are registers), plain selects instead of EXEC windows: it runs on a few pairs of a rough launch and on none of
// a smooth one.  Same indexing as the fp32 step: lane l computes K[l+1][q+1] on step sigma = l + q from cur = K[l+1][q],
// up = K[l][q+1] (lane l-1 before its own step) and the up of the step before = K[l][q].
// What the block costs the pairs that do not take it (same-box A/B against the build without it): C4 4.90 -> 4.88 ms,
// N=1024 T=32 1.36 -> 1.34 ms, forward-only launches +2 %, and +4 us on the 41 us launch of N=128, T=32 -- there the
// 168-register kernel now reloads ~20 loop invariants per pair from scratch, whatever shape the block takes (a rolled
// re-solve from the staged coordinates, a call, parking live values in memory around it: all measured, all worse).
template <int RING>
__device__ __forceinline__ double resweep_fwd_fp64(const float (&Dsl)[RING], int P, int lrow)
{
    double cur = 1.0, upprev = 1.0;
    // (opaque copies: the compiler otherwise hoists the 126 lane-window predicates out of the kernel's pair loop -- they
    //  depend on nothing a pair changes -- and keeps them in spilled SGPRs for the whole kernel)
    asm volatile("" : "+v"(lrow), "+s"(P));
    const bool rowv = lrow < P;
#pragma unroll
    for (int s0 = 0; s0 < 2 * RING - 3; s0 += 8) {
        if (s0 <= 2 * P - 2) { // (uniform; steps past 2P-2 have no lane inside the grid)
#pragma unroll
            for (int s = s0; s < s0 + 8 && s < 2 * RING - 3; ++s) {
                const double up = dpp_shr1_one(cur); // lane 0 (and, two rows per wavefront, the lane after the idle row 31): 1.0
                float gf = Dsl[s & (RING - 1)];
                asm volatile("" : "+v"(gf)); // (a slot serves steps s and s + RING: converted twice, not kept as a double in between)
                const double g = (double)gf;
                const double t = cur + up;
                const double nw = (t - upprev) + g * __builtin_fma(g, t + upprev, 1.7320508075688772 * t);
                cur = (rowv && (unsigned)(s - lrow) < (unsigned)P) ? nw : cur;
                upprev = up;
            }
        }
    }
    return cur;
}
// Sum over the lanes in DPP adds (no LDS round trip).  Lane 31 ends with the total of lanes 0 .. 31; lane 63 with the
// total of ALL 64 lanes if `whole`, else of lanes 32 .. 63 (two pairs per wavefront).
template <bool WHOLE>
__device__ __forceinline__ float wave_sum_dpp(float v)
{
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xF, 0xF, true)); // row_shr:1
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xF, 0xF, true)); // row_shr:2
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xF, 0xF, true)); // row_shr:4
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xF, 0xF, true)); // row_shr:8 (lane 15 of a row: its total)
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, true)); // row_bcast:15 into rows 1, 3
    if (WHOLE) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xC, 0xF, true)); // row_bcast:31 into rows 2, 3
    return v;
}
__device__ __forceinline__ float max3_abs(float m, float a, float b)
{
    asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(m) : "v"(a), "v"(b));
    return m;
}

// [slot][lane] image of G (then R*G) with row stride 65 floats: every in-sweep access is
// lane*4 + constant (one ds instruction with an immediate offset, nothing to keep in registers),
// and the transposed read of the symmetric pass ((m+n)&63)*65 + m hits 32 distinct banks.
constexpr int GS_STRIDE = 65;
constexpr int GS_WAVE = 64 * GS_STRIDE;
using f32x2 = __attribute__((ext_vector_type(2))) float;

__device__ __forceinline__ int gs_index(int slot, int lane) { return slot * GS_STRIDE + lane; }

template <typename IO>
__device__ __forceinline__ double load_io(const void *base, size_t idx)
{
    return (double)static_cast<const IO *>(base)[idx];
}
__device__ __forceinline__ double load_any(const void *base, size_t idx, int io64)
{
    return io64 ? load_io<double>(base, idx) : load_io<float>(base, idx);
}
__device__ __forceinline__ void store_any(void *base, size_t idx, double v, int io64)
{
    if (io64)
        static_cast<double *>(base)[idx] = v;
    else
        static_cast<float *>(base)[idx] = (float)v;
}

// LP: the last of the DPAD channels is padding (d == DPAD - 1, e.g. the 7-DoF arm at DPAD = 8): its FMA in the
// static kernel and its travelling column sum are dropped.
// RING: slots per lane = length of the column ring.  64 in general; 32 for paths of <= 32 points (T <= 32), where the
// skewed phases 1 and 4 take 34 iterations instead of 66 and a wavefront solves TWO pairs at once: lanes 0..31 own the
// points of trajectory i, lanes 32..63 those of trajectory i + 1, against the SAME column trajectory j (time index =
// lane & 31).  The wave-wide machinery stays intact: the lane next to the seam, row 31, has no cells for T <= 32 and
// therefore holds the boundary value 1 that row 0 of the second pair needs from its DPP shift (and S = 0 for the
// scatter); each of the 64 travelling column sums collects one of the two rows at every time index on its way through 32
// consecutive lanes, and the two sums of a column (they end 32 lanes apart) are added when the workgroup's waves are --
// exactly the sum over the tile's rows the column side wants.  (Round 2 let lanes 32..63 mirror lanes 0..31: the same
// work twice.)
template <int DPAD, int NW, bool GRAD, bool SYM, bool LP, int RING = 64>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(
    GRAD ? ((RING == 32 && NW == 4 && DPAD <= 8) ? 3 : 1) : ((DPAD <= 8) ? 3 : 2),
    GRAD ? ((RING == 32 && NW == 4 && DPAD <= 8) ? 3 : 2) : ((DPAD <= 8) ? 3 : 2)))) void gram_fast_kernel(FastArgs a)
{
    constexpr int DC = LP ? DPAD - 1 : DPAD; // channels that can be non-zero
    constexpr int NT = NW * 64;
    constexpr int RM = RING - 1;
    constexpr int RPW = (RING == 32) ? 2 : 1; // rows (trajectories i) per wavefront
    constexpr int NWR = NW * RPW;             // rows per tile
    // y rows are stored twice (row r and r + 64) so that the skewed row (t - lane) & 63 becomes
    // (64 - lane) + t: a per-lane base plus a compile-time offset.
    // Row strides are padded (YDS doubles / YFS floats) so that the 16 lanes a ds_read_b128 services per
    // cycle -- consecutive rows, one per lane -- start on 16 distinct 16-byte bank groups: with the
    // natural 64-B (fp64, d<=8) / 32-B (fp32) rows they alias 4-way / 2-way (measured: 54 % of all LDS
    // cycles were bank-conflict cycles).
    constexpr int YDS = DPAD + 2; // doubles per fp64 row: 80 B at DPAD=8 -> 80*l mod 256 distinct for 16 l
    constexpr int YFS = (DPAD == 4) ? 12 : DPAD + 4; // floats per fp32 row: 48 B at DPAD<=8, 80 B at 16
    // (sized by the ring: with 32 slots a 4-wave workgroup needs 47 KB instead of 84, three of them fit a CU)
    constexpr int GSW = RING * GS_STRIDE; // floats of a wavefront's [slot][lane] image
    __shared__ float Gs_all[GRAD ? NW * GSW : 1];
    __shared__ __align__(16) double yd[2 * RING * YDS];
    __shared__ double ynd[2 * RING];
    __shared__ double yref[DPAD];
    __shared__ __align__(16) float yf[GRAD ? 2 * RING * YFS : 1];

    const int tid = threadIdx.x, lane = tid & 63;
    // (scalar: row index, row pointers and the per-wave LDS bases then live in SGPRs; as a vector value hipcc hoists the
    //  row's element addresses out of the column loop as 64-bit VGPR pairs and spills them)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & RM; // the row this lane owns
    const int T = a.T, d = a.d, P = T - 1, io64 = a.io64;
    // Work distribution: the items of a launch -- (owned row tile, column), tile-major; symmetric launches only the
    // columns from the tile's first row on -- all cost the same (the NW waves of a workgroup meet at a barrier per
    // column), so they are split into contiguous equal ranges, one per workgroup of a grid that fills the chip once.
    // A range touches few row tiles: the per-lane row-side gradient accumulators live in registers across the columns of
    // a tile and are stored to the (workgroup, tile) segment's slot when the range leaves it; and every next column is known
    // in advance, so its loads are in flight during the current pair (the round-1/2 work queue exposed an atomic and a
    // load round trip per chunk: 63 % of the wave cycles at N=128, T=32 with its single-column chunks).
#ifdef SIGSVGD_PHASE_STAMPS
    unsigned long long ph_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast_ = __builtin_amdgcn_s_memtime();
#endif
    int j0 = 0, j1 = 0;
    float *Gs = Gs_all + (GRAD ? wave * GSW : 0);
    const double inv_h = a.inv_h;
    const float m2h = (float)(-2.0 * inv_h * 3.46410161513775459); // the G image holds G / sqrt(12)

    // K_fwd, then S = K_fwd * U, of this lane's row, one slot per anti-diagonal (slot = (column + lane) & 63).
    // Zeroed per pair right before the forward sweep (64 moves, 0.5 % of a pair): the sweeps write a slot only while
    // its cell is inside the grid, so the slots without a grid cell (column >= P or lane >= P) read 0 in the phase-4
    // pass, which takes all 64 slots unmasked -- and the 64 registers are free during staging and phase 1.
    float Ksl[RING];
    double gacc[DPAD]; // per-lane fp64 accumulators of d sum_j w_ij k(x_i, y_j) / d x_i[lane, c]
    double xraw[DPAD]; // raw row of x_i owned by this lane (fp64 copy of the fp32/fp64 input; exact)

    // ---- staging of column trajectory y_j: element e -> (t = e / DPAD, c = e % DPAD) -------------
    constexpr int EPT = (64 * DPAD + NT - 1) / NT; // elements per thread
    double stage_v[EPT], stage_r[EPT];
    auto stage_load = [&](int j) {
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int e = tid + k * NT;
            const int t = e / DPAD, c = e % DPAD;
            const bool ok = e < RING * DPAD && t < T && c < d && j < a.B;
            stage_v[k] = ok ? load_any(a.Y, ((size_t)j * T + t) * d + c, io64) : 0.0;
            stage_r[k] = ok ? load_any(a.Y, (size_t)j * T * d + c, io64) : 0.0;
        }
    };
    const double nscale = -inv_h * 1.4426950408889634074; // exponent scale: exp(-d/h) = 2^(nscale*d)
    auto stage_store = [&]() {
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int e = tid + k * NT;
            if (e < RING * DPAD) {
                const int t = e / DPAD, c = e % DPAD;
                const double v = stage_v[k] - stage_r[k];
                if (!(LP && c == DPAD - 1)) { // (LP: that slot receives the norm below)
                    yd[t * YDS + c] = v;
                    yd[(t + RING) * YDS + c] = v;
                }
                if (GRAD) {
                    yf[t * YFS + c] = (float)v;
                    yf[(t + RING) * YFS + c] = (float)v;
                }
                if (t == 0) yref[c] = stage_r[k];
                double s = v * v * nscale; // -log2(e)/h * |y~_t|^2 via xor-reduction over the DPAD lanes of a row
#pragma unroll
                for (int off = 1; off < DPAD; off <<= 1) s += __shfl_xor(s, off, 64);
                if (c == 0) {
                    ynd[t] = s;
                    ynd[t + RING] = s;
                    if (LP) { // the padded channel of the fp64 row carries the norm: phase 1 needs no separate read
                        yd[t * YDS + DPAD - 1] = s;
                        yd[(t + RING) * YDS + DPAD - 1] = s;
                    }
                }
            }
        }
    };

    // this workgroup's range of items, and the tile / column it starts in
    const long long it0 = a.nitems * blockIdx.x / gridDim.x, it1 = a.nitems * (blockIdx.x + 1) / gridDim.x;
    int remaining = (int)(it1 - it0);
    long long item = it0; // index of the (row tile, column) item in work
    int kq = 0, cstart = 0;
    {
        long long rem = it0;
        for (;; ++kq) {
            const int cn = a.B - (SYM ? a.tm.tile_of(kq) * NWR : 0);
            if (rem < cn) break;
            rem -= cn;
        }
        cstart = (int)rem;
    }
    bool staged = false;
    while (remaining > 0) {
    const int itile = a.tm.tile_of(kq);
    const int cfirst = SYM ? itile * NWR : 0; // first column that touches or crosses the diagonal
    const int ncol = min(a.B - cfirst - cstart, remaining);
    const int jnext_tile = SYM ? a.tm.tile_of(kq + 1) * NWR : 0; // where the range goes on, on the next owned tile
    const bool more_tiles = remaining > ncol;
    const int i0 = itile * NWR + wave * RPW;                // first (or only) row of this wavefront: scalar
    const int i = (RING == 32) ? i0 + (lane >> 5) : i0;     // the row this LANE works for
    const bool row_ok = i < a.A;
#pragma unroll
    for (int c = 0; c < DPAD; ++c) {
        gacc[c] = 0.0;
        xraw[c] = (row_ok && lrow < T && c < d) ? load_any(a.X, ((size_t)i * T + lrow) * d + c, io64) : 0.0;
    }
    j0 = cfirst + cstart;
    j1 = j0 + ncol;
    if (!staged) { // the range's first column: the only exposed load round trip
        staged = true;
        stage_load(j0);
        stage_store();
        __syncthreads();
    }

    for (int j = j0; j < j1; ++j) {
        // (two rows per wavefront: the second one is the later row, so the first decides whether there is work at all;
        //  a half without a pair of its own -- row beyond A, or left of the diagonal -- computes along and is dropped)
        const bool pair_ok = i0 < a.A && (!SYM || j >= i0);
        const bool mine = row_ok && (!SYM || j >= i); // this lane's pair exists
        const bool last = j + 1 == j1;
        const bool more = !last || more_tiles;
        if (more) stage_load(last ? jnext_tile : j + 1); // in flight during the pair

        float Dsl[RING]; // increments / sqrt(12) (the scale the difference-form stencil wants), one slot per anti-diagonal

        SIG_STAMP(0)
        SIG_PRIO(3)
        if (pair_ok) {
            // ---- phase 0: centre x_i on y_j[0] ----------------------------------------------------
            // x~ pre-scaled so that the exponent argument in base 2 is one fused dot product:
            //   log2(e) * (-|x~ - y~|^2 / h) = xn + ynd[q] + sum_c xs[c] * y~[q][c]
            double xs[DPAD];
            double xn = 0.0;
#pragma unroll
            for (int c = 0; c < DPAD; ++c) {
                const double xc = (lrow < T && c < d) ? xraw[c] - yref[c] : 0.0;
                xn = __builtin_fma(xc, xc, xn);
                xs[c] = xc * (-2.0 * nscale);
            }
            // everything downstream of the static kernel works with G / sqrt(12): the increments become
            // gamma = D / sqrt(12) (stencil below), and the gradient pass multiplies the scale back (m2h)
            xn = __builtin_fma(xn, nscale, -1.79248125036057809); // - log2(sqrt(12))

            // ---- phase 1: G rows (skewed: column (t - lane) & 63 on iteration t) -> D slots --------
            {
                double g0 = 0.0, g1 = 0.0, gprev = 0.0, rdprev = 0.0;
                const double *ybase = yd + (RING - lrow) * YDS; // row (t - lrow) & RM == ybase + t*YDS
                const double *nbase = ynd + (RING - lrow);
                // the y~ row of column t+1 is fetched while column t is computed (one s_waitcnt per
                // column instead of one per ds_read, and the LDS latency is off the exp chain)
                double ynx[DPAD], nnx;
#pragma unroll
                for (int c = 0; c < DPAD; ++c) ynx[c] = ybase[c];
                nnx = LP ? 0.0 : nbase[0];
#pragma unroll
                for (int t = 0; t < RING + 2; ++t) {
                    double g;
                    if (t < RING) {
                        double ycu[DPAD];
#pragma unroll
                        for (int c = 0; c < DPAD; ++c) ycu[c] = ynx[c];
                        double e2 = LP ? xn + ycu[DPAD - 1] : xn + nnx; // LP: the row's last slot is the norm
                        if (t < RING - 1) {
                            const double *yr = ybase + (t + 1) * YDS;
#pragma unroll
                            for (int c = 0; c < DPAD; ++c) ynx[c] = yr[c];
                            if (!LP) nnx = nbase[t + 1];
                        }
#pragma unroll
                        for (int c = 0; c < DC; ++c) e2 = __builtin_fma(xs[c], ycu[c], e2);
                        g = exp2_p7(e2);
                        if (GRAD) Gs[gs_index(t, lane)] = (float)g;
                        if (t == 0) g0 = g;
                        if (t == 1) g1 = g;
                    } else {
                        g = (t == RING) ? g0 : g1;
                    }
                    const double rd = g - gprev; // G[l, q] - G[l, q-1]
                    gprev = g;
                    if (t >= 2) {
                        // lane l+1 holds the same column difference one iteration later
                        const double nb = dpp_shl1_zero(rd);
                        Dsl[(t - 2) & RM] = (float)(nb - rdprev);
                        asm volatile("" : "+v"(Dsl[(t - 2) & RM])); // formed HERE: hipcc otherwise sinks the fp64 difference to the sweep
                    }
                    rdprev = rd;
                    __builtin_amdgcn_sched_barrier(0); // one column per scheduling region: bounds live ranges
                }
            }

            SIG_STAMP(1)
            SIG_PRIO(2)
            float kf_keep = 0.f;                   // K[P][P] of this lane's pair and the cancellation verdicts of phase 2, for the
            bool x0_keep = false, x1_keep = false; // conditioning check after the reverse sweep (4-channel gradient kernels)
            // ---- phase 2: forward sweep, anti-diagonal sigma = 0 .. 2P-2, in fp32 DIFFERENCE FORM ----------
            // With gamma = g / sqrt(12) the second-order stencil reads
            //     K11 - K01 = (K10 - K00) + F,   F = gamma * (sqrt(3) * t + gamma * (t + K00)),  t = K10 + K01,
            // so the lane carries V = K[l+1][q] - K[l][q] along its row (V += F: a lane-local recurrence whose
            // rounding errors are relative to |V| << |K|) and forms K11 = K01 + V with ONE full-magnitude add
            // that never feeds back into V.  In fp32 this is within 2e-7 of the fp64 recurrence for K and for
            // the gradient (scripts/dev/precision_vform.py: 1.5e-7 / 1.9e-7 at the C4 shape, where the plain
            // fp32 stencil loses 4e-6..2e-5), at 10 fp32-rate instructions per step instead of 12.5, seven of
            // them fp64-rate (measured cost per SIMD at two waves: fp64 2.4 ns, fp32 1.4 ns, DPP move 1.9 ns).
            {
                // `up` persists (lane 0 keeps the boundary K[0][.] = 1) in TWO registers used on alternate steps:
                // the diagonal neighbour K[l][q] of a step is the upper neighbour of the step before, whether or
                // not this lane was active then (a lane's value is 1.0 until its row starts and frozen after it
                // ends), so it needs no copy.
                float cur = 1.f, upA = 1.f, upB = 1.f, V = 0.f, km = 1.f, sd = 0.f;
                const int smax = 2 * P - 2;
                if (GRAD) {
#pragma unroll
                    for (int k = 0; k < RING; ++k) Ksl[k] = 0.f;
                }
                for (int rnd = 0; rnd < (RING == 64 ? 2 : 1); ++rnd) { // (RING = 32: 2P-2 <= 60, one round of 64 steps)
                    if (rnd * 64 > smax) break;
                    float r3 = 1.7320508075688772f;
                    asm volatile("" : "+s"(r3)); // opaque per round: nothing of the step is round-invariant
                    const unsigned long long *mk = (RING == 64 ? SWEEP_MASK.m[P] : SWEEP_MASK32.m[P]) + rnd * 64; // EXEC windows
#pragma unroll
                    for (int k0 = 0; k0 < 64; k0 += 8) // Ksl[k] <- K[l, q]
                        sweep_fwd8<GRAD ? 0 : (DPAD == 4 ? 2 : 1)>(cur, upA, upB, V, &Dsl[k0 & RM], &Ksl[k0 & RM], mk + k0, r3, km, sd);
                }
                // largest |K| on this lane's row against the pair's result: cancellation -> fp64 (resweep_fwd_fp64 above)
                if (GRAD) {
                    km = fabsf(cur); // K[l+1][P]; the slots: K[l][q], q < P
#pragma unroll
                    for (int k = 0; k < RING; k += 2) km = max3_abs(km, Ksl[k], Ksl[k + 1]);
                }
                float kf = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cur), P - 1));
                if (RING == 32) {
                    const float kf1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cur), 32 + P - 1));
                    kf = (lane >> 5) ? kf1 : kf;
                }
                // Paths in one to four channels (the 4-channel instantiations; the launcher passes a flag array for d <= 4 only --
                // d = 4 since a soak case with T = 33, d = 4 showed an entry of |K| ~ 0.005 off by 1.7e-5 RELATIVE after the fp64
                // re-sweep on fp32 increments: tiny entries need the exact pass whatever their conditioning):
                // the fp64 re-sweep below runs on fp32 increments, which is not enough where the discrete solution is
                // ill-conditioned, so such pairs are flagged for the EXACT fp64 pass of the coverage kernel that follows the
                // launch (fp64 static kernel, increments and sweeps: 6e-8).  Two rules, either one flags the pair:
                //   * cancellation of magnitudes (round 3): the grid maximum above 4 max(|K|, 0.1) (2 in one channel) -- the fp32
                //     sweeps resolve ~1e-6 of the largest value on the grid, the boundary value 1 included;
                //   * CONDITIONING (round 4): K[P][P] as a function of the increments has the first-order condition number
                //     c1 = sum |S * D| / |K| (S = K_fwd * U is dK/dD up to the stencil's second-order terms), and the fp32
                //     STORAGE of the increments (6e-8 each) costs K up to 2.7e-8 c1 whatever the precision of the sweeps
                //     (measured over 5,000 pairs of 25 roughness regimes, scripts/dev/cond_study.py: error <= 2.7e-8 c1, every
                //     pair beyond 3e-6 has c1 > 230; smooth paths -- the bench inputs at d <= 3 -- stay below 100 at T <= 128).
                //     Gradient launches have S in the slots after the reverse sweep and flag c1 > 150 there (below, "conditioning");
                //     forward-only launches have no U and use the bound  sum |K_fwd * D| * max(grid maximum, 1) / |K| > 300
                //     (>= c1 up to 15 % in every pair of the study, within a factor 7 of it for three channels; no pair beyond
                //     2.5e-6 stays below it).
                // A pair that came out NaN (a NaN in its inputs) is not marked: the coverage kernel's clamped exponential would
                // turn it into a number.  Compiled into the 4-channel kernels only: in the 8-channel ones the ballot and the byte
                // store cost the headline launch 1.7 % (same-box A/B), and no pair of >= 4 channels needs it (worst entry 9e-7 in
                // the roughest regimes of the study).
                bool x0 = false, x1 = false;
                if constexpr (DPAD == 4) {
                    if (a.kflag) {
                        const float kden = fmaxf(fabsf(kf), 0.1f);
                        bool fl = km > (d == 1 ? 2.f : 4.f) * kden;
                        if constexpr (!GRAD) { // sum over the pair's lanes of sum_q |K[l][q] gamma[l][q]|, times sqrt(12): in units of D
                            const float wsum = wave_sum_dpp<RING == 64>(sd);
                            float sds = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wsum), 63));
                            if (RING == 32) {
                                const float s0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wsum), 31));
                                sds = (lane >> 5) ? sds : s0;
                            }
                            fl = fl || sds * 3.46410161513775459f * fmaxf(km, 1.f) > 300.f * kden;
                        }
                        const unsigned long long xbal = __builtin_amdgcn_ballot_w64(mine && kf == kf && fl);
                        x0 = (RING == 64) ? xbal != 0 : (unsigned)xbal != 0u;
                        x1 = (xbal >> 32) != 0;
                    }
                }
                if (lrow == P - 1 && mine) { // this lane's last value is K[P, P]
                    store_any(a.K, (size_t)i * a.B + j, (double)cur, io64);
                    if (SYM && j != i) store_any(a.K, (size_t)j * a.B + i, (double)cur, io64);
                    if constexpr (DPAD == 4 && !GRAD) { // (two rows per wavefront: each half of the ballot is one pair)
                        if (a.kflag) a.kflag[(size_t)i * a.B + j] = (RING == 32 && (lane >> 5)) ? x1 : x0;
                    }
                }
                kf_keep = kf;
                x0_keep = x0;
                x1_keep = x1;
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(mine && km > 4.f * fmaxf(fabsf(kf), 0.1f)) != 0, 0)) {
                    const double k64 = resweep_fwd_fp64<RING>(Dsl, P, lrow);
                    if (lrow == P - 1 && mine) { // (same lane, same addresses as the first store: the later one stands)
                        store_any(a.K, (size_t)i * a.B + j, k64, io64);
                        if (SYM && j != i) store_any(a.K, (size_t)j * a.B + i, k64, io64);
                    }
                }
            }

            SIG_STAMP(2)
            SIG_PRIO(1)
            if (GRAD) {
                // weights of this lane's pair: row side w_ij, column side w_ji (per lane with two rows per wavefront,
                // uniform otherwise; fetched here so that the loads are long back when the gradient pass needs them)
                float w_ij = 1.f, w_ji = 1.f; // (per lane with two rows per wavefront; uniform otherwise)
                if (!mine) {
                    w_ij = 0.f; w_ji = 0.f;
                } else if (a.go) {
                    w_ij = (float)load_any(a.go, (size_t)i * a.B + j, io64);
                    if (SYM || a.symw) w_ji = (float)load_any(a.go, (size_t)j * a.B + i, io64);
                    if (a.symw) { w_ij += w_ji; w_ji = w_ij; }
                } else if (a.symw) {
                    w_ij = 2.f; w_ji = 2.f;
                }
                const float wcol = (mine && j != i) ? w_ji : 0.f; // the diagonal pair has no column side
                // ---- phase 3: reverse sweep (U recurrence; S replaces K_fwd slot by slot) -----------------
                float cur = 1.f, downA = 1.f, downB = 1.f, V = 0.f; // `down` persists (lane 63 keeps U[P][.] = 1), alternating as above
                float Sb = 0.f, eprev = 0.f;
                f32x2 acc[DPAD / 2]; // packed pairs: the contraction runs on v_pk_fma_f32
#pragma unroll
                for (int c = 0; c < DPAD / 2; ++c) acc[c] = f32x2{0.f, 0.f};
                const int smax = 2 * P - 2;

                // x~ of this lane's row as packed fp32 pairs (row-side closing formula; column-side products)
                f32x2 xc2[DPAD / 2];
#pragma unroll
                for (int c = 0; c < DPAD / 2; ++c) {
                    const bool ok0 = lrow < T && 2 * c < d, ok1 = lrow < T && 2 * c + 1 < d;
                    xc2[c] = f32x2{ok0 ? (float)(xraw[2 * c] - yref[2 * c]) : 0.f,
                                   ok1 ? (float)(xraw[2 * c + 1] - yref[2 * c + 1]) : 0.f};
                }
                // column-side (SYM): accumulators that TRAVEL one lane down per step.  On step sigma lane m
                // holds R*G[m, n] with n = sigma+2-m; on the next step lane m-1 holds the same column n, so
                // sum_m R*G[m,n] * x~_m rides a wave rotation (v_add_f32 with a wave_rol:1 DPP source) and
                // never needs a transposed pass: finished columns wrap past lane 0 through lanes that
                // are already idle, and after the last step column n sits in lane (64 - n) & 63.
                float tacc[DPAD];
#pragma unroll
                for (int c = 0; c < DPAD; ++c) tacc[c] = 0.f;

                int yfrow = RING - lrow;                        // row (sigma + 2 - lrow) & RM == yfrow + k2
                int gsoff = (GRAD ? wave * GSW : 0) + lane; // this wave's [slot][lane] image
                // one iteration of the phase-4 pass (below); `gcu`, `ycu`: G[l][n] and the y~_n row, fetched one
                // iteration ahead so that the LDS latency is off the chain and one s_waitcnt serves the iteration
                auto grad_part = [&](float Snew, bool accumulate, float gcu, const f32x2 *ycu) {
                    // R[l][q+2] = (S[l][q+2] - S[l][q+1]) - (S[l-1][q+2] - S[l-1][q+1]): the column difference
                    // e = S[l][q+1] - S[l][q] of this iteration serves this lane on the NEXT iteration and, through the wave
                    // shift, lane l+1 on this one (its column is one behind) -- each difference is formed once (rounds 1-2
                    // shifted S and formed the neighbour's difference a second time: one instruction more per iteration,
                    // 5.38 -> 5.25 ms per C4 launch in a same-box A/B; the result is the same bit for bit)
                    const float e = Sb - Snew;
                    const float up = dpp_shr1_zero(e); // lane l-1: S[l-1][q+2] - S[l-1][q+1]; lane 0: no row above
                    const float R = eprev - up;
                    eprev = e;
                    Sb = Snew;
                    if (!accumulate) return; // warm-up iteration: only the two-deep history is filled
                    const float rg = R * gcu;
                    const f32x2 rg2 = {rg, rg};
                    // Both contractions take the DIFFERENCE x~_m - y~_n (one packed subtraction per channel pair, shared
                    // by the two sides).  Rounds 1-2 accumulated sum R G y~_n and sum R G separately and closed with
                    // x~_m * sum - sum: when consecutive points lie more than a bandwidth apart, G is concentrated where
                    // x~_m - y~_n is smallest and that closing cancels catastrophically (gradient error 1e-3 relative to
                    // its maximum at |step|^2 / h ~ 30, profiles/r03_precision_sweep.md; 3e-6 in this form).
                    f32x2 df[DPAD / 2];
#pragma unroll
                    for (int c = 0; c < DPAD / 2; ++c) df[c] = xc2[c] - ycu[c];
#pragma unroll
                    for (int c = 0; c < DPAD / 2; ++c) acc[c] = __builtin_elementwise_fma(rg2, df[c], acc[c]);
                    // pin the running sums here: the contraction must stay inside its step
#pragma unroll
                    for (int c = 0; c < DPAD / 2; ++c) asm volatile("" : "+v"(acc[c]));
                    if (SYM) {
                        // (two rows per wavefront: a travelling sum collects BOTH rows on its way, so each contribution
                        //  carries its own pair's weight; with one row the weight is applied once, when the sums are parked)
                        const float rgc = (RING == 32) ? rg * wcol : rg;
                        const f32x2 rgc2 = {rgc, rgc};
#pragma unroll
                        for (int c = 0; c < DPAD / 2; ++c) {
                            const f32x2 pr = rgc2 * df[c];
                            tacc[2 * c] = add_rol1(tacc[2 * c], pr[0]);
                            if (2 * c + 1 < DC) tacc[2 * c + 1] = add_rol1(tacc[2 * c + 1], pr[1]);
                        }
#pragma unroll
                        for (int c = 0; c < DPAD; ++c) asm volatile("" : "+v"(tacc[c]));
                    }
                };

                for (int rnd = (RING == 64 ? 1 : 0); rnd >= 0; --rnd) {
                    if (rnd * 64 > smax) continue;
                    // (keeps the phase-4 LDS addresses out of this loop's invariants: without the pin hipcc
                    //  rearranges the pass that follows and the C4 launch goes from 6.9 to 10.9 ms)
                    asm volatile("" : "+v"(yfrow), "+v"(gsoff));
                    float r3 = 1.7320508075688772f;
                    asm volatile("" : "+s"(r3));
                    const unsigned long long *mk = (RING == 64 ? SWEEP_MASK.m[P] : SWEEP_MASK32.m[P]) + rnd * 64;
                    // same difference form, V = U[l][q] - U[l+1][q] carried towards smaller q; `down` alternates between
                    // two registers like `up` (the diagonal neighbour U[l+1][q+1] is the lower neighbour one step ago)
#pragma unroll
                    for (int k0 = 56; k0 >= 0; k0 -= 8) sweep_rev8(cur, downA, downB, V, &Dsl[k0 & RM], &Ksl[k0 & RM], mk + k0, r3);
                }

                // ---- conditioning (4-channel kernels with a flag array, i.e. d <= 4): c1 = sum |S * D| / max(|K|, 0.1) ------
                // The slots hold S = K_fwd * U (0 where there is no cell) and gamma = D / sqrt(12): 64 multiply-adds per lane, a
                // wave sum, one byte per pair (see phase 2 for the rule and its measurement).  C3 (d = 3): +1.2 % instructions.
                if constexpr (DPAD == 4) {
                    if (a.kflag) {
                        float cs = 0.f;
#pragma unroll
                        for (int k = 0; k < RING; ++k) cs = __builtin_fmaf(fabsf(Ksl[k]), fabsf(Dsl[k]), cs);
                        const float wsum = wave_sum_dpp<RING == 64>(cs);
                        float c1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wsum), 63));
                        if (RING == 32) {
                            const float c10 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wsum), 31));
                            c1 = (lane >> 5) ? c1 : c10;
                        }
                        const bool ill = kf_keep == kf_keep && c1 * 3.46410161513775459f > 150.f * fmaxf(fabsf(kf_keep), 0.1f);
                        if (lrow == P - 1 && mine)
                            a.kflag[(size_t)i * a.B + j] = (((RING == 32 && (lane >> 5)) ? x1_keep : x0_keep) || ill) ? 1 : 0;
                    }
                }
                SIG_STAMP(3)
                SIG_PRIO(0)
                // ---- phase 4: 4-corner scatter R and both contractions, one column per lane and iteration ----
                // The reverse sweep has only half of its lanes inside the grid at any step, so nothing but the
                // recurrence is left in it.  Here every lane is busy on every iteration: lane l takes the slots in
                // descending order, i.e. column q = (63 - it - l) & 63 with wrap-around; slots without a grid cell
                // (column >= P, row >= P) hold S = 0 (see Ksl), which is also exactly what the scatter needs at the
                // wrap (S[.][-1] = S[.][63] = 0).  Two warm-up iterations fill the history, the next 64 visit
                // every column n = q + 2 once; the travelling sums rotate as before and end in lane (64 - n) & 63.
                float gnx = 0.f, Svn = Ksl[RM];
                f32x2 ynx[DPAD / 2];
#pragma unroll
                for (int c = 0; c < DPAD / 2; ++c) ynx[c] = f32x2{0.f, 0.f};
#pragma unroll
                for (int it = 0; it < RING + 2; ++it) {
                    const float Sv = Svn, gcu = gnx;
                    f32x2 ycu[DPAD / 2];
#pragma unroll
                    for (int c = 0; c < DPAD / 2; ++c) ycu[c] = ynx[c];
                    if (it + 1 < RING + 2) { // next iteration's operands: slot (RING-2 - it) & RM, column slot (RING - it) & RM
                        Svn = Ksl[(RING - 2 - it) & RM];
                        if (it + 1 >= 2) {
                            const int k2n = (RING - it) & RM;
                            gnx = Gs_all[gsoff + k2n * GS_STRIDE];
                            const f32x2 *yr = reinterpret_cast<const f32x2 *>(yf + (yfrow + k2n) * YFS);
#pragma unroll
                            for (int c = 0; c < DPAD / 2; ++c) ynx[c] = yr[c];
                        }
                    }
                    grad_part(Sv, it >= 2, gcu, ycu);
                    __builtin_amdgcn_sched_barrier(0);
                }

                SIG_STAMP(4)
                // row-side gradient of this pair: -(2/h) * sum_n R G (x~_m - y~_n); column side: the same sum over m, negated
#pragma unroll
                for (int c = 0; c < DPAD; ++c) {
                    gacc[c] += (double)(w_ij * m2h * acc[c / 2][c % 2]);
                }

                if (SYM) {
                    // park the column-side result in this wave's own G region ([half][column][c]) for the block sum; the
                    // diagonal pair has no column side, a half without a pair contributes nothing
                    const int ncol = (RING - lrow) & RM; // the column whose finished sums this lane ended up with
                    const float wpark = (ncol <= P) ? ((RING == 32) ? -m2h : -(wcol * m2h)) : 0.f;
                    float *pk = Gs + ((RING == 32 ? (lane >> 5) * RING : 0) + ncol) * DPAD;
#pragma unroll
                    for (int c = 0; c < DPAD; ++c) pk[c] = wpark * tacc[c];
                }
            }
        } else if (GRAD && SYM) {
#pragma unroll
            for (int c = 0; c < DPAD; ++c) Gs[lane * DPAD + c] = 0.f; // idle wave contributes nothing
        }
        SIG_STAMP(5)
#ifndef SIG_EXPERIMENT_NO_PAIR_BARRIER // timing experiment only (results are garbage without it)
        __syncthreads(); // every wave is done with y_j (and has parked its column-side result)
#endif
        SIG_STAMP(6)
        if (GRAD && SYM) {
            // thread e takes element e of the column's [T][d] block: the NW waves' sums are added in wave order and the
            // item's partial goes to its own row of the slab with one coalesced store per element (grad_reduce_kernel
            // adds the rows of a column in tile order).  Rounds 1-2 sent it with one atomic per element into a shared
            // accumulator: 29.6 M memory-side atomics per C4 launch, and a result that depended on their order.
            const float inv_d = 1.0f / (float)d;
            float *dst = a.cslab + (size_t)item * (T * d);
            for (int e = tid; e < T * d; e += NT) {
                const int n = (int)(((float)e + 0.5f) * inv_d), c = e - n * d; // exact for e < 2^20
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    s += Gs_all[w * GSW + n * DPAD + c];
                    if (RING == 32) s += Gs_all[w * GSW + (RING + n) * DPAD + c]; // the wavefront's second row
                }
                dst[e] = s;
            }
        }
        ++item;
        if (more) stage_store();
#ifndef SIG_EXPERIMENT_NO_PAIR_BARRIER
        __syncthreads();
#endif
    }

    if (GRAD && row_ok && lrow < T) { // this (workgroup, row tile) segment's own slot: segments are numbered kq + workgroup
        const int rslot = wave * RPW + (RING == 32 ? (lane >> 5) : 0);
        double *dst = a.rseg + (((size_t)(kq + (int)blockIdx.x)) * NWR + rslot) * (size_t)(T * d) + lrow * d;
#pragma unroll
        for (int c = 0; c < DPAD; ++c)
            if (c < d) dst[c] = gacc[c];
    }
    remaining -= ncol;
    ++kq;
    cstart = 0;
    } // row tiles of the range
#ifdef SIGSVGD_PHASE_STAMPS
    SIG_STAMP(0)
    if (lane == 0 && a.stamps)
        for (int k = 0; k < 8; ++k) atomicAdd(&a.stamps[k], ph_[k]);
#endif
}

// ---- fixed-order reduction of the gradient partials -----------------------------------------------------------------
// One thread per output element (i, t, c).  Row side: the segments of row i's tile, one per workgroup whose item range
// met the tile, in workgroup order.  Column side (symmetric launches): the items (tile, column i) of every owned tile
// whose first row is <= i, in tile order.  fp64 sums; the order is a function of the launch geometry only, so two
// launches on the same input give the same bits.
struct GradReduceArgs {
    const double *rseg;
    const float *cslab;
    void *out; // [A][T*d]: the I/O type, or fp64 for the sharded partial solve
    int out64, A, B, TD, NW, sym, grid;
    TileMap tm;
    long long nitems;
};
// grid (ceil(T*d / 64), A), 4 wavefronts: a workgroup serves 64 elements of one row i, so everything that depends on the
// row only -- its tile, the tile's item range, the workgroups that met it, the slab row of column i in every owned
// tile -- is wave-uniform SCALAR arithmetic, and the slab rows are walked incrementally (a tile's items follow the
// previous tile's: one 64-bit add per tile; evaluating the closed forms per load made the kernel scalar-bound: 27.5 us
// at N=128, T=32, d=7 -- a third of the iteration -- and 100-137 us per C4 launch whatever the thread layout).  An
// element is a sum over up to N / NW slab rows (column side) and one segment per workgroup that met the row's tile (row
// side: a handful in large launches, up to ~100 in small ones, where a workgroup owns one or two columns): wavefront w
// adds the w-th quarter of each list, sixteen loads in flight, and the four partial sums are joined in wavefront order.
constexpr int RED_W = 4; // (8 wavefronts of half the share each: 78 us instead of 66 at C4)
__global__ __launch_bounds__(RED_W * 64) void grad_reduce_kernel(GradReduceArgs r)
{
    __shared__ double part[RED_W][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int e = blockIdx.x * 64 + lane;
    const int i = r.A - 1 - (int)blockIdx.y; // rows with the longest column sums first (66 -> 60 us at C4)
    const bool live = e < r.TD;
    double s = 0.0;
    if (live) {
        // row side: the segments of the row's tile, one per workgroup whose item range met it
        const int ti = i / r.NW, wv = i % r.NW;
        const int kqr = r.tm.kq_of_tile(ti);
        if (kqr >= 0) {
            const long long S0 = r.tm.start(kqr, r.B, r.NW, r.sym);
            const long long cn = r.sym ? r.B - ti * r.NW : r.B;
            // workgroup w works on the items [nitems*w/grid, nitems*(w+1)/grid): the one holding item x is
            const int wlo = (int)(((S0 + 1) * r.grid - 1) / r.nitems), whi = (int)(((S0 + cn) * r.grid - 1) / r.nitems);
            const int nseg = whi - wlo + 1, per = (nseg + RED_W - 1) / RED_W;
            const int w0 = wlo + wave * per, w1 = min(whi + 1, w0 + per);
            const size_t sstride = (size_t)r.NW * r.TD;
            const double *base = r.rseg + ((size_t)kqr * r.NW + wv) * r.TD + e + (size_t)w0 * sstride;
            for (int w = w0; w < w1; w += 8, base += 8 * sstride) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = (w + u < w1) ? base[u * sstride] : 0.0;
#pragma unroll
                for (int u = 0; u < 8; ++u) s += v[u];
            }
        }
        // column side: the items (owned tile, column i) of the tiles whose first row is <= i, in the order of the enumeration
        if (r.sym) {
            const int per = (r.tm.owned + RED_W - 1) / RED_W;
            const int k0 = wave * per, k1 = min(r.tm.owned, k0 + per);
            // slab row of item (tile kq, column i) = start(kq) + i - tile * NW; start(kq + 1) = start(kq) + B - tile * NW
            const float *row = r.cslab + (size_t)(k0 < k1 ? r.tm.start(k0, r.B, r.NW, 1) : 0) * r.TD + e;
            for (int kq = k0; kq < k1; kq += 16) {
                float v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const bool have = kq + u < k1;
                    const int first = have ? r.tm.tile_of(kq + u) * r.NW : r.B; // first row of the tile
                    const bool in = first <= i;
                    v[u] = in ? row[(size_t)(in ? i - first : 0) * r.TD] : 0.f;
                    row += have ? (size_t)(r.B - first) * r.TD : 0;
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) s += (double)v[u];
            }
        }
    }
    part[wave][lane] = s;
    __syncthreads();
    if (wave != 0 || !live) return;
    s = part[0][lane];
#pragma unroll
    for (int w = 1; w < RED_W; ++w) s += part[w][lane];
    const size_t idx = (size_t)i * r.TD + e;
    if (r.out64)
        static_cast<double *>(r.out)[idx] = s;
    else
        static_cast<float *>(r.out)[idx] = (float)s;
}

// ---- host side ---------------------------------------------------------------------------------
bool fast_supported(int A, int B, int T, int d, int n, int kind, unsigned flags)
{
    (void)A; (void)B;
    if (n != 0 || T < 3 || T > 64 || d > 16) return false;
    if (kind != SIGSVGD_STATIC_RBF) return false;
    if (flags & SIGSVGD_FLAG_NAIVE_SOLVER) return false;
    return true;
}

// compute units of the current device (256 on MI355X); the persistent grids are sized from it
int device_cu_count()
{
    static std::atomic<int> cache[64]; // per device ordinal (zero-initialised: not queried yet)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev >= 0 && dev < 64) {
        const int c = cache[dev].load(std::memory_order_relaxed);
        if (c > 0) return c;
    }
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    if (dev >= 0 && dev < 64) cache[dev].store(v, std::memory_order_relaxed);
    return v;
}

TileMap make_tilemap(int ntile, int off, int stride, bool fold)
{
    TileMap t;
    t.off = off; t.stride = stride; t.ntile = ntile; t.fold = fold ? 1 : 0;
    auto count_upto = [&](int hi) { return hi >= off ? (hi - off) / stride + 1 : 0; }; // tiles off + k*stride <= hi
    if (!fold) {
        t.m0 = t.owned = count_upto(ntile - 1);
    } else {
        t.m0 = count_upto((ntile - 1) / 2);                        // p <= ntile-1-p
        t.owned = t.m0 + (ntile >= 2 ? count_upto((ntile - 2) / 2) : 0); // mirror images of the p < ntile-1-p
    }
    return t;
}

// Geometry of a gradient launch of the register-resident / quadrant kernels: NW rows per tile, `resident` workgroups on
// the chip; the kernels and grad_reduce_kernel derive items, ranges and segments from the same numbers.
GradGeom grad_geometry(int A, int B, int TD, bool sym, int off, int stride, bool fold, int NW, long long resident)
{
    GradGeom g;
    g.NW = NW;
    g.tm = make_tilemap((A + NW - 1) / NW, off, stride, fold);
    g.nitems = g.tm.start(g.tm.owned, B, NW, sym ? 1 : 0);
    g.grid = (int)(g.nitems < resident ? g.nitems : resident);
    g.rseg_bytes = (((size_t)(g.tm.owned + g.grid) * NW * TD * sizeof(double)) + 255) & ~(size_t)255;
    g.cslab_bytes = sym ? (((size_t)g.nitems * TD * sizeof(float)) + 255) & ~(size_t)255 : 0;
    return g;
}

int grad_reduce_launch(const GradGeom &g, const double *rseg, const float *cslab, void *out, int out64, int A, int B, int TD,
                       bool sym, hipStream_t stream)
{
    GradReduceArgs r;
    r.rseg = rseg; r.cslab = cslab; r.out = out; r.out64 = out64;
    r.A = A; r.B = B; r.TD = TD; r.NW = g.NW; r.tm = g.tm;
    r.sym = sym ? 1 : 0; r.grid = g.grid > 0 ? g.grid : 1; r.nitems = g.nitems > 0 ? g.nitems : 1;
    if (A > 65535) {
        set_error("gradient reduction: %d rows exceed the grid limit", A);
        return SIGSVGD_E_UNSUPPORTED;
    }
    hipLaunchKernelGGL(grad_reduce_kernel, dim3((unsigned)((TD + 63) / 64), (unsigned)A), dim3(RED_W * 64), 0, stream, r);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "launch grad_reduce_kernel");
    return SIGSVGD_OK;
}

namespace {
inline int cu_count() { return device_cu_count(); }
// geometry of a GRADIENT launch (the forward-only launches keep no partial sums): rows per tile, workgroups a CU holds
inline int grad_nw(int T, int d) { return (d <= 8) ? 8 : 4; } // (T <= 32: 4 wavefronts of two rows each)
inline int grad_wg_per_cu(int T, int d) { return (d <= 8 && T <= 32) ? 3 : 1; }
using FastGeom = GradGeom;
FastGeom fast_geometry(int A, int B, int T, int d, bool sym, const TileMap &tm)
{
    return grad_geometry(A, B, T * d, sym, tm.off, tm.stride, tm.fold != 0, grad_nw(T, d),
                         (long long)cu_count() * grad_wg_per_cu(T, d));
}
FastGeom fast_geometry(int A, int B, int T, int d, bool sym)
{
    return grad_geometry(A, B, T * d, sym, 0, 1, false, grad_nw(T, d), (long long)cu_count() * grad_wg_per_cu(T, d));
}
} // namespace

int sym_tile_rows_fast(int T, int d) { return grad_nw(T, d); }

namespace {
// (the flag array of the exact fp64 pass: the 4-channel instantiations, i.e. paths in one to four channels; see the kernel)
inline size_t fast_flag_bytes(int A, int B, int d)
{
    return d <= 4 ? (((size_t)A * B + 255) & ~(size_t)255) + generic_repair_bytes() : 0;
}
} // namespace

int fast_workspace_bytes(int A, int B, int T, int d, int want_grad, unsigned flags, size_t *bytes)
{
    (void)flags;
    *bytes = 512 + fast_flag_bytes(A, B, d);
    if (want_grad) { // the larger of the ordered and the symmetric launch (the query carries no Y_IS_X promise)
        const FastGeom o = fast_geometry(A, B, T, d, false);
        size_t need = o.rseg_bytes;
        if (A == B) {
            const FastGeom y = fast_geometry(A, B, T, d, true);
            if (y.rseg_bytes + y.cslab_bytes > need) need = y.rseg_bytes + y.cslab_bytes;
        }
        *bytes += need + 256;
    }
    return SIGSVGD_OK;
}

namespace {
template <int DPAD, int NW, int RING = 64>
int launch_variant(const GramProblem &p, FastArgs &a, bool grad, bool sym)
{
    // items of this launch: (owned row tile, column); symmetric launches only the columns from the tile's first row on
    constexpr int NWR = NW * (RING == 32 ? 2 : 1); // rows per tile: the 32-slot ring solves two rows per wavefront
    const TileMap tm = make_tilemap((p.A + NWR - 1) / NWR, a.tm.off, a.tm.stride, a.tm.fold != 0);
    if (tm.owned <= 0) return SIGSVGD_OK;
    const long long total = tm.start(tm.owned, p.B, NWR, sym ? 1 : 0);
    if (total <= 0) return SIGSVGD_OK;
    const int ncu = cu_count();
    a.tm = tm;
    a.nitems = total;
#ifdef SIGSVGD_PHASE_STAMPS
    {
        static unsigned long long *dbg = nullptr;
        if (!dbg) (void)hipMalloc(&dbg, 8 * sizeof(unsigned long long));
        (void)hipMemsetAsync(dbg, 0, 8 * sizeof(unsigned long long), p.stream);
        a.stamps = dbg;
    }
#endif
    const long long resident = (long long)ncu * (grad ? ((RING == 32 && NW == 4 && DPAD <= 8) ? 3 : 1) : (NW == 4 ? ((DPAD <= 8) ? 3 : 2) : 1)); // workgroups the chip holds at once (LDS / VGPR bound)
    dim3 grid((unsigned)(total < resident ? total : resident), 1);
    dim3 block(NW * 64);
    if (grad) { // the reduction kernel re-derives the segments from this geometry: it must be the one the workspace was cut for
        const FastGeom g = fast_geometry(p.A, p.B, p.T, p.d, sym, a.tm);
        if (g.NW != NWR || g.grid != (int)grid.x || g.nitems != total) {
            set_error("fast: launch geometry mismatch (rows per tile %d/%d grid %d/%u items %lld/%lld)", g.NW, NWR, g.grid, grid.x,
                      g.nitems, total);
            return SIGSVGD_E_BADARG;
        }
    }
    constexpr bool HAS_LP = DPAD <= 8; // the d == DPAD - 1 instantiations exist for the 4- and 8-channel layouts
    const bool lp = HAS_LP && grad && p.d == DPAD - 1;
    if (!grad && sym)
        hipLaunchKernelGGL((gram_fast_kernel<DPAD, NW, false, true, false, RING>), grid, block, 0, p.stream, a);
    else if (!grad)
        hipLaunchKernelGGL((gram_fast_kernel<DPAD, NW, false, false, false, RING>), grid, block, 0, p.stream, a);
    else if (sym && lp)
        hipLaunchKernelGGL((gram_fast_kernel<DPAD, NW, true, true, HAS_LP, RING>), grid, block, 0, p.stream, a);
    else if (sym)
        hipLaunchKernelGGL((gram_fast_kernel<DPAD, NW, true, true, false, RING>), grid, block, 0, p.stream, a);
    else if (lp)
        hipLaunchKernelGGL((gram_fast_kernel<DPAD, NW, true, false, HAS_LP, RING>), grid, block, 0, p.stream, a);
    else
        hipLaunchKernelGGL((gram_fast_kernel<DPAD, NW, true, false, false, RING>), grid, block, 0, p.stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "launch gram_fast_kernel");
    if (a.kflag) { // exact fp64 pass of the coverage kernel over the flagged pairs (a few microseconds when there are none)
        const int rc = generic_repair_launch(p, a.kflag, nullptr, sym, a.tm, NWR);
        if (rc) return rc;
    }
#ifdef SIGSVGD_PHASE_STAMPS
    {
        unsigned long long h[8];
        (void)hipStreamSynchronize(p.stream);
        (void)hipMemcpy(h, a.stamps, sizeof(h), hipMemcpyDeviceToHost);
        double tot = 0;
        for (int k = 0; k < 8; ++k) tot += (double)h[k];
        static const char *nm[8] = {"staging/other", "phase 0+1 static kernel", "phase 2 forward sweep",
                                    "phase 3 reverse sweep", "phase 4 gradient pass", "pair epilogue",
                                    "barrier after the pair", "-"};
        fprintf(stderr, "[phase stamps] A=%d T=%d d=%d grad=%d sym=%d: ", p.A, p.T, p.d, (int)grad, (int)sym);
        for (int k = 0; k < 7; ++k) fprintf(stderr, "%s %.1f%% | ", nm[k], 100.0 * (double)h[k] / tot);
        fprintf(stderr, "total %.3e wave-cycles\n", tot);
    }
#endif
    return SIGSVGD_OK;
}

int dispatch_variant(const GramProblem &p, FastArgs &a, bool grad, bool sym)
{
    if (!grad && p.d <= 4) // forward only: no G image, <=168 VGPRs -> 4-wave workgroups pack 3 waves per SIMD
        return p.T <= 32 ? launch_variant<4, 4, 32>(p, a, false, sym) : launch_variant<4, 4>(p, a, false, sym);
    if (!grad && p.d <= 8)
        return p.T <= 32 ? launch_variant<8, 4, 32>(p, a, false, sym) : launch_variant<8, 4>(p, a, false, sym);
    if (p.d <= 4) // (paths of <= 32 points: the 32-slot ring on 4-wave workgroups, three per CU = 3 waves per SIMD)
        return p.T <= 32 ? launch_variant<4, 4, 32>(p, a, grad, sym) : launch_variant<4, 8>(p, a, grad, sym);
    if (p.d <= 8)
        return p.T <= 32 ? launch_variant<8, 4, 32>(p, a, grad, sym) : launch_variant<8, 8>(p, a, grad, sym);
    return launch_variant<16, 4>(p, a, grad, sym); // 1 wave per SIMD: 512-VGPR budget, no spills
}

// cut the two slabs out of the caller's workspace and enqueue kernel + reduction
int run_grad(const GramProblem &p, FastArgs &a, bool sym, void *out, int out64)
{
    const FastGeom g = fast_geometry(p.A, p.B, p.T, p.d, sym, a.tm);
    const size_t need = fast_flag_bytes(p.A, p.B, p.d) + g.rseg_bytes + g.cslab_bytes + 256;
    if (!p.ws || p.ws_bytes < need) {
        set_error("fast: workspace %zu B < required %zu B", p.ws_bytes, need);
        return SIGSVGD_E_WORKSPACE;
    }
    unsigned char *base = reinterpret_cast<unsigned char *>((reinterpret_cast<uintptr_t>(p.ws) + 255) & ~(uintptr_t)255);
    a.kflag = p.d <= 4 ? base : nullptr;
    base += fast_flag_bytes(p.A, p.B, p.d);
    a.rseg = reinterpret_cast<double *>(base);
    a.cslab = sym ? reinterpret_cast<float *>(base + g.rseg_bytes) : nullptr;
    int rc = dispatch_variant(p, a, true, sym);
    if (rc) return rc;
    return grad_reduce_launch(g, a.rseg, a.cslab, out, out64, p.A, p.B, p.T * p.d, sym, p.stream);
}

void fill_args(const GramProblem &p, FastArgs &a)
{
    a.X = p.X; a.Y = p.Y; a.go = p.grad_out; a.K = p.K_out;
    a.io64 = p.dtype == SIGSVGD_F64; a.A = p.A; a.B = p.B; a.T = p.T; a.d = p.d;
    a.symw = (p.flags & SIGSVGD_FLAG_SYM) ? 1 : 0; a.inv_h = p.inv_h;
    a.tm = make_tilemap(1, 0, 1, false); a.nitems = 0; // (off / stride / fold are what launch_variant reads: a full launch)
    a.rseg = nullptr;
    a.cslab = nullptr;
    a.kflag = nullptr;
}
} // namespace

int fast_launch(const GramProblem &p)
{
    const bool grad = p.gradX_out != nullptr;
    const bool sym = (p.flags & SIGSVGD_FLAG_Y_IS_X) && p.A == p.B; // Y is X: each unordered pair once, K mirrored
    FastArgs a;
    fill_args(p, a);
    if (a.symw && p.A != p.B) {
        set_error("sym backward needs A == B");
        return SIGSVGD_E_BADARG;
    }
    if (!grad) {
        if (p.d <= 4) {
            if (!p.ws || p.ws_bytes < fast_flag_bytes(p.A, p.B, p.d) + 256) {
                set_error("fast: workspace %zu B < required %zu B", p.ws_bytes, fast_flag_bytes(p.A, p.B, p.d) + 256);
                return SIGSVGD_E_WORKSPACE;
            }
            a.kflag = reinterpret_cast<unsigned char *>((reinterpret_cast<uintptr_t>(p.ws) + 255) & ~(uintptr_t)255);
        }
        return dispatch_variant(p, a, false, sym);
    }
    return run_grad(p, a, sym, p.gradX_out, p.dtype == SIGSVGD_F64);
}

// Symmetric partial solve for particle sharding: this launch owns the row tiles tile_offset + k*tile_stride of the upper
// triangle of Gram(X, X).  K_partial[N,N] (caller-zeroed) receives both orientations of every owned pair;
// grad_partial[N,T,d] (fp64) is OVERWRITTEN with this launch's share of the gradient (rows it does not touch get 0).
int fast_sym_partial(const GramProblem &p, int tile_offset, int tile_stride, bool fold, double *grad_partial)
{
    if (!fast_supported(p.A, p.B, p.T, p.d, p.n, p.kind, p.flags) || p.A != p.B) {
        set_error("sym_partial: shape/kernel outside the register-resident path (need n=0, 3<=T<=64, d<=16, RBF)");
        return SIGSVGD_E_UNSUPPORTED;
    }
    if (tile_stride < 1 || tile_offset < 0 || tile_offset >= tile_stride) {
        set_error("sym_partial: bad tile_offset/stride %d/%d", tile_offset, tile_stride);
        return SIGSVGD_E_BADARG;
    }
    FastArgs a;
    fill_args(p, a);
    a.Y = p.X;
    a.tm = make_tilemap((p.A + grad_nw(p.T, p.d) - 1) / grad_nw(p.T, p.d), tile_offset, tile_stride, fold);
    return run_grad(p, a, true, grad_partial, 1);
}

} // namespace sigsvgd

SIG_EXEC_DEBUG_GETTER(fast)
