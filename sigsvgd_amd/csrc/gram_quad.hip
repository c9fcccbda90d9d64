// Signature-kernel Gram forward/backward for LONG paths: dyadic order 0, 65 <= T <= 128, d <= 16, RBF,
// second-order stencil (BASELINE.json config C5: T = 128, d = 14).  STORED forward solution (nothing is regenerated
// backwards, so there is no limit on how rough the paths may be), at the register budget and occupancy of the
// 64-point kernel (gram_fast.hip): two wavefronts per SIMD, 64 + 64 slot registers.
//
// Mapping: one wavefront per trajectory pair.  The (T-1)^2 cell grid is cut into 2 x 2 QUADRANTS of <= 64 x 64 cells;
// in quadrant (b, h) lane l owns cell row 64 b + l and sweeps the cell columns 64 h .. 64 h + 63 anti-diagonal by
// anti-diagonal exactly as gram_fast.hip does (slot = (column + lane) & 63 for the increments D / sqrt(12) and for
// K_fwd -> S = K_fwd * U; fp32 difference form; lanes outside the grid switched off through EXEC windows).  What is
// new is the boundary of a quadrant:
//   * left / right: a row's running values (K, V) simply continue from quadrant (b, 0) into (b, 1) -- they live in
//     the lane; the neighbour row's corner value arrives through the same DPP shift that serves a lane that has not
//     started yet (its neighbour still holds the final value of the previous quadrant);
//   * top / bottom: lane 63 of band 0 files K[64][.] in LDS while it sweeps (one ds_write per step, through a
//     per-lane address: every other lane writes to a dummy cell), lane 0 of band 1 takes it from there: the
//     boundary value of the NEXT step is moved into the shift destination at the end of each step (it is `old` of
//     the next DPP shift, which has no source for lane 0).  Mirror image for U[64][.] in the reverse sweep.
// Only ONE quadrant's D and S are live at a time, so the forward solution of a quadrant must exist when its
// reverse sweep runs: the schedule recomputes (static kernel + forward sweep) instead of storing
//     band 0:  (0,0) (0,1)                                  forward only: leaves K[64][.]
//     band 1:  (1,0) (1,1)* (1,0)*                          * = reverse sweep + gradient pass follow the forward sweep
//     band 0:        (0,1)* (0,0)*                          (0,1) restarts from the lanes' state at the end of the first
//                                                           (0,0) pass: four registers per lane, kept across band 1
// i.e. 7 forward sweeps, 4 reverse sweeps, 4 gradient passes per pair (the minimum is 4 / 4 / 4; holding everything
// would take 4 x 128 slot registers or 100 KB of LDS per pair).  The static kernel (the expensive, LDS-bound part of
// a pass) runs only 4 times: the increments of the three quadrants that are visited twice are written to a per-wave
// scratch in global memory on the first visit (64 coalesced 256-B stores) and read back on the second -- 96 KB per
// pair that never leave L2 / MALL, because the grid is one workgroup per CU striding over the work items.
// The gradient pass is the 4-corner scatter of gram_fast.hip, with the static kernel re-evaluated in fp32 from the
// centred coordinates (no G image: LDS stays small enough for 8 wavefronts).  Points on the seams between
// quadrants need S from both sides and are done separately: point column 64 (and 0) per band from three captured
// values per lane, point row 64 per pair from the two S rows next to it (one dense pass).
// Column-side sums (Y is X) ride travelling accumulators (one v_add_f32_dpp wave_ror:1 per channel and step) and join a
// [column][channel] image in LDS once per quadrant.
//
// Reference semantics: sigkernel _SigKernelGram.forward/backward [RECALLED, SURVEY.md App. A];
// static kernel src/kernels/_traj_kernels.py:176-195.
#include "sig_common.h"

namespace sigsvgd {

struct QuadArgs {
    const void *X, *Y, *go;
    void *K;
    // gradient partial sums leave the kernel through plain stores, added up in a fixed order by grad_reduce_kernel
    // (gram_fast.hip): no atomics anywhere, bit-reproducible results
    double *rseg; // [owned tiles + workgroups][8][T*d]: row-side sums of one (workgroup, row tile) segment
    float *cslab; // [items][T*d]: column-side sums of one (row tile, column) item (symmetric launches)
    float *crec;  // [gridDim.x][8 waves][QREC]: the column-side sums of each wavefront's pair, joined after the pair's barrier
    float *rowg;  // [gridDim.x][8 waves][128 * 16]: row-side accumulator of a segment when it does not fit LDS (d = 15, 16)
    int io64, A, B, T, d, symw;
    TileMap tm;                   // row tiles owned by this launch, in the order of the enumeration (sharded partial solve)
    long long nitems;             // (owned row tile, column) items of the launch, tile-major; symmetric launches only the
                                  // columns from the tile's first row on; each workgroup takes one contiguous range
    float *dcache;                // [gridDim.x][8 waves][3 quadrants][64 slots][64 lanes] fp32: increments kept between
                                  // the forward-only and the full pass over a quadrant (gradient launches; may be NULL)
    double inv_h;
    unsigned char *kflag; // [A][B]: 1 where the fp32 solution of the pair cancelled (max |K_grid| > 4 max(|K|, 0.1)): the
                          // launcher lets the coverage kernel solve those pairs' K again in fp64 (generic_repair_launch)
#ifdef SIGSVGD_PHASE_STAMPS
    unsigned long long *stamps; // diagnostic build only (scripts/dev/phase_stamps.py): shader-clock totals per phase
#endif
};

#ifdef SIGSVGD_PHASE_STAMPS
#define SIG_QSTAMP(i)                                                        \
    {                                                                        \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();        \
        ph_[i] += now_ - tlast_;                                             \
        tlast_ = now_;                                                       \
    }
#else
#define SIG_QSTAMP(i)
#endif

namespace {
#ifndef SIGQ_NW // (-DSIGQ_NW=4: one wavefront per SIMD, a timing experiment of scripts/dev/ab_quad.py)
#define SIGQ_NW 8
#endif
constexpr int QNW = SIGQ_NW; // wavefronts (rows i) per workgroup
#ifndef SIGQ_CANCEL_RATIO
#define SIGQ_CANCEL_RATIO 8.f
#endif
// A pair is solved again in fp64 when max |K_grid| > ratio * max(|K[P][P]|, 0.1): the fp32 sweeps lose about 5e-7 (T = 64) ..
// 2e-6 (T = 128) of the LARGEST value on the grid (the boundary value 1 included: round 3 also asked for a grid maximum above
// 2, which left pairs that decay to K < 0.125 unchecked).  ratio = 8; 4 for paths in two channels and 2 in one.  Paths in <= 3
// channels are also checked for their CONDITIONING in the increments (sum |S D| / |K|, gram_fast.hip): that, not the grid
// maximum, is what the 9 soak cases of round 3 beyond 1e-5 had in common.
constexpr float QUAD_CANCEL_RATIO = SIGQ_CANCEL_RATIO;
// floats of a wavefront's column-side records of one pair: 4 quadrant passes + 2 halves of point row 64, each
// [DPAD + 1 values][64 lanes], + 4 seam-column records of DPAD + 1 values (sized for DPAD = 16)
constexpr int QREC = 6656;
using qf32x2 = __attribute__((ext_vector_type(2))) float;

__device__ __forceinline__ double q_ldany(const void *b, size_t i, int io64)
{
    return io64 ? static_cast<const double *>(b)[i] : (double)static_cast<const float *>(b)[i];
}
// n consecutive elements (stride 1) as doubles, ONE uniform branch on the I/O type and independent loads (a branch
// per element serialises the loads behind one s_waitcnt each)
template <int N>
__device__ __forceinline__ void q_ldrow(const void *b, size_t i0, int nvalid, int io64, double (&out)[N])
{
    // (clamped index + select instead of a guarded load: a guard is a branch per element, and hipcc reloads the
    //  spilled row pointer before each of them)
    if (io64) {
        const double *p = static_cast<const double *>(b) + i0;
        double tmp[N];
#pragma unroll
        for (int c = 0; c < N; ++c) tmp[c] = p[min(c, nvalid - 1)];
#pragma unroll
        for (int c = 0; c < N; ++c) out[c] = (c < nvalid) ? tmp[c] : 0.0;
    } else {
        const float *p = static_cast<const float *>(b) + i0;
        float tmp[N];
#pragma unroll
        for (int c = 0; c < N; ++c) tmp[c] = p[min(c, nvalid - 1)];
#pragma unroll
        for (int c = 0; c < N; ++c) out[c] = (c < nvalid) ? (double)tmp[c] : 0.0;
    }
}
__device__ __forceinline__ float q_max3_abs(float m, float a, float b)
{
    asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(m) : "v"(a), "v"(b));
    return m;
}
__device__ __forceinline__ void q_stany(void *b, size_t i, double v, int io64)
{
    if (io64)
        static_cast<double *>(b)[i] = v;
    else
        static_cast<float *>(b)[i] = (float)v;
}
// lane l <- lane l+1; lane 63 keeps `old` (compiler-visible DPP: hipcc pads its hazards)
__device__ __forceinline__ double q_shl_keep(double v, double old)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), 0x130, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), 0x130, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float q_shr_zero(float v) // lane l <- lane l-1, lane 0 gets 0
{
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x138, 0xF, 0xF, true));
}
// sum over the 64 lanes in six DPP adds (no LDS round trips); the total ends up in lane 63
__device__ __forceinline__ float q_wave_sum63(float v)
{
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xF, 0xF, true)); // row_shr:1
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xF, 0xF, true)); // row_shr:2
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xF, 0xF, true)); // row_shr:4
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xF, 0xF, true)); // row_shr:8 (inclusive scan of each row of 16)
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, true)); // row_bcast:15 into rows 1, 3
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xC, 0xF, true)); // row_bcast:31 into rows 2, 3
    return v;
}
// G[m][n] = 2^(-log2(e)/h * |x~_m - y~_n|^2) in fp32, from the DIFFERENCES (the expanded form loses 6e-8 of its largest
// term, 1e-5 of G for rough paths).  ONE expression for the gradient pass and both seam passes: the row-side sums
// telescope (constant column paths: zero gradient up to one fp32 rounding) only if every point sees the same bits.
// The differences are handed back: both gradient contractions take them (sum R G (x~_m - y~_n); the split form
// x~_m * sum R G - sum R G y~_n cancels catastrophically once consecutive points lie more than a bandwidth apart).
// (DC: channels that can be non-zero, even; the pairs beyond it are left out of the sum and their differences set to 0)
template <int DPAD, int DC = DPAD>
__device__ __forceinline__ float q_gval(const float (&xf)[DPAD], const qf32x2 (&y2)[DPAD / 2], float ns32,
                                        qf32x2 (&df)[DPAD / 2])
{
    qf32x2 e2 = qf32x2{0.f, 0.f};
#pragma unroll
    for (int c = DC / 2; c < DPAD / 2; ++c) df[c] = qf32x2{0.f, 0.f};
#pragma unroll
    for (int c = 0; c < DC / 2; ++c) {
        df[c] = qf32x2{xf[2 * c], xf[2 * c + 1]} - y2[c];
        e2 = __builtin_elementwise_fma(df[c], df[c], e2);
    }
    return __builtin_amdgcn_exp2f((e2[0] + e2[1]) * ns32);
}
// rotate-and-add in one VALU instruction: returns acc[lane-1] + v (lane 0 reads lane 63)
__device__ __forceinline__ float q_add_ror1(float acc, float v)
{
    float out;
    asm("v_add_f32_dpp %0, %1, %2 wave_ror:1 row_mask:0xf bank_mask:0xf" : "=v"(out) : "v"(acc), "v"(v));
    return out;
}

// 2^t as in sig_common.h (exp2_p7), with the coefficients in scalar registers: under this kernel's register pressure
// hipcc otherwise materialises them as VGPR pairs and spills those (16 scratch round trips per use site)
struct QExp7 {
    double c7, c6, c5, c4, c3, c2, c1, c0;
};
__device__ __forceinline__ QExp7 qexp7_coef()
{
    return QExp7{1.5303701161442145e-05, 1.5469729221575116e-04, 1.3333478471058548e-03, 9.618025613268967e-03,
                 5.5504109063307244e-02, 2.4022651213498578e-01, 6.931471805568296e-01,  0.9999999999595621};
}
__device__ __forceinline__ void qexp7_pin(QExp7 &k)
{
    asm volatile("" : "+s"(k.c7), "+s"(k.c6), "+s"(k.c5), "+s"(k.c4), "+s"(k.c3), "+s"(k.c2), "+s"(k.c1), "+s"(k.c0));
}
__device__ __forceinline__ double qexp2_p7(double t, const QExp7 &k)
{
    const double kf = __builtin_rint(t);
    const double f = t - kf;
    double p = __builtin_fma(k.c7, f, k.c6);
    p = __builtin_fma(p, f, k.c5);
    p = __builtin_fma(p, f, k.c4);
    p = __builtin_fma(p, f, k.c3);
    p = __builtin_fma(p, f, k.c2);
    p = __builtin_fma(p, f, k.c1);
    p = __builtin_fma(p, f, k.c0);
    return ldexp(p, (int)kf);
}

#include "quad_sweeps.h"
} // namespace

// ROWG: the row-side sums of a segment do not fit LDS next to the rest (d = 15, 16) and live in a per-wavefront global
// accumulator instead (a separate instantiation: its code costs the common one 40 spilled registers)
// EARLY: paths of <= 112 points (second band / half of <= 47 cells): the sweeps skip the steps beyond a quadrant's last
// anti-diagonal (T=100, d=7: 2.51 -> 2.40 ms symmetric); longer paths keep the unconditional 128-step statements
// FEW (forward-only launches of paths in <= 3 channels): the forward steps also accumulate sum |K_fwd * D|, the bound on the
// condition number that sends ill-conditioned pairs to the exact fp64 pass (gram_fast.hip, "conditioning"; gradient launches
// take the condition number itself from the S slots after each reverse sweep, a uniform branch on d)
template <int DPAD, bool GRAD, bool SYM, bool ROWG = false, bool EARLY = false, bool FEW = false>
__global__ __launch_bounds__(QNW * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void gram_quad_kernel(QuadArgs a)
{
    constexpr int NT = QNW * 64;
    // Channels the hot loops touch: the 16-channel gradient instantiation without ROWG serves d <= 14 only (d = 15, 16 take
    // the ROWG one), so the last channel pair is all zeros there -- the bimanual arm of BASELINE.json's stress config has
    // d = 14 -- and drops out of the static kernel's dot product, the fp32 kernel value and both contractions.
    constexpr int DC = (DPAD == 16 && GRAD && !ROWG) ? 14 : DPAD;
    constexpr int CS = DPAD + 1;  // values per point column of the column-side sums: DPAD channels and the weight sum
    constexpr int YDS = DPAD + 2; // fp64 row: coordinates, [DPAD] = -log2(e)/h * |y~|^2
    constexpr int YFS = (DPAD == 16) ? 18 : 12; // fp32 row (8-byte aligned; 18 l mod 64 visits 32 distinct even banks)
    // point column n = 64 h + c is stored at rows 128 h + c and 128 h + 64 + c: the skewed row (t - lane) & 63 of
    // half h is then base(h, lane) + t * stride
    __shared__ __align__(16) double yd[256 * YDS];
    __shared__ __align__(16) float yf[GRAD ? 256 * YFS : 4];
    __shared__ double yref[DPAD];
    constexpr int HN = 136; // hand-over rows: entries 0 .. 129 are read
    __shared__ float ones[HN];
    struct WaveLds { // everything a wavefront keeps for itself, behind ONE base address
        double g64[128], rdh[128];
        float hK[HN], hU[HN], hdummy[64], srow[256], x64[DPAD + 2];
    };
    __shared__ WaveLds wl_all[QNW];
    // row-side gradient of the wavefront's particle, summed over the columns of the work item before it goes to memory
    // (one coalesced flush per item instead of 128 d lane-strided fp64 atomics per pair: -8 % symmetric, -16 % ordered)
    constexpr int RS = (DPAD == 8) ? 9 : 15; // row stride (odd); d = 15, 16 do not fit next to the rest: ROWG
    __shared__ float rowacc_all[(GRAD && !ROWG) ? QNW * 128 * RS : 4];

    const int tid = threadIdx.x, lane = tid & 63;
    // (scalar: row index, row pointers and the per-wave LDS bases then live in SGPRs; as a vector value hipcc hoists the
    //  row's element addresses out of the column loop as 64-bit VGPR pairs and spills them)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = a.T, d = a.d, P = T - 1, io64 = a.io64;
    const int nrows1 = P - 64; // cell rows of band 1 = cell columns of half 1 (0 .. 63)
    const double inv_h = a.inv_h;
    const double nscale = -inv_h * 1.4426950408889634074;
    const float m2h = (float)(-2.0 * inv_h);
    WaveLds &wl = wl_all[wave];
    float *hK = wl.hK, *hU = wl.hU, *hdummy = wl.hdummy;
    double *g64 = wl.g64, *rdh = wl.rdh;
    float *srow63 = wl.srow, *srow64 = wl.srow + 128;
    float *x64 = wl.x64;
    float *rowacc = rowacc_all + ((GRAD && !ROWG) ? wave * 128 * RS : 0);
    constexpr bool rowlds = GRAD && !ROWG;
    for (int e = tid; e < HN; e += NT) ones[e] = 1.f;
    for (int e = lane; e < HN; e += 64) hK[e] = 1.f, hU[e] = 1.f; // (entries the sweeps do not write stay at the boundary value)

#ifdef SIGSVGD_PHASE_STAMPS
    unsigned long long ph_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast_ = __builtin_amdgcn_s_memtime();
#endif
    float *dcw = a.dcache ? a.dcache + ((size_t)blockIdx.x * QNW + wave) * (6 * 64 * 64) : nullptr;
    // KSTORE (8-channel layout): the forward solution of the three quadrants whose reverse sweep comes later waits in the
    // scratch next to the increments, and each quadrant is swept forwards once (4 / 4 / 4 sweeps and passes, the minimum)
    // instead of being re-swept before its reverse sweep (7 forward sweeps): T=100, d=7 symmetric 2.80 -> 2.70 ms, ordered
    // 4.19 -> 3.93 at N=256; T=128, d=14 ordered 5.52 -> 5.37.  The symmetric 16-channel instantiation keeps the re-sweep:
    // there the 64 extra loads in front of the reverse sweep and the registers they cost outweigh the sweep (3.83 -> 4.01 ms).
    constexpr bool KSTORE = DPAD == 8 || !SYM;

    // Work distribution as in gram_fast.hip: the items of a launch -- (owned row tile, column), tile-major; symmetric
    // launches only the columns from the tile's first row on -- all cost the same (the 8 waves meet at a barrier per
    // column), so a grid of at most one workgroup per CU cuts them into contiguous equal ranges.  A range touches few row
    // tiles: the row-side sums of a wavefront's particle stay in LDS across the columns of a tile and are stored to the
    // (workgroup, tile) segment's slot when the range leaves it.
    const long long it0 = a.nitems * blockIdx.x / gridDim.x, it1 = a.nitems * (blockIdx.x + 1) / gridDim.x;
    int remaining = (int)(it1 - it0);
    long long item = it0; // index of the (row tile, column) item in work
    int kq = 0, cstart = 0;
    {
        long long rem = it0;
        for (;; ++kq) {
            const int cn = a.B - (SYM ? a.tm.tile_of(kq) * QNW : 0);
            if (rem < cn) break;
            rem -= cn;
        }
        cstart = (int)rem;
    }
    // this wavefront's column-side records / global row accumulator (re-derived from scalars where they are used: as
    // values living across the pair they are spilled)
#define SIGQ_CRW (a.crec + ((size_t)blockIdx.x * QNW + wave) * QREC)
#define SIGQ_RGW (a.rowg + ((size_t)blockIdx.x * QNW + wave) * (128 * 16))
#pragma unroll 1
    while (remaining > 0) {
    const int itile = a.tm.tile_of(kq);
    const int cfirst = SYM ? itile * QNW : 0; // first column that touches or crosses the diagonal
    const int ncol = min(a.B - cfirst - cstart, remaining);
    const int i0 = itile * QNW;
    const int i = i0 + wave;
    const int j0 = cfirst + cstart, j1 = j0 + ncol;
    const bool row_ok = i < a.A;
    if (rowlds) {
        for (int e = lane; e < 128 * RS; e += 64) rowacc[e] = 0.f;
    } else if (GRAD) {
        for (int e = lane; e < 128 * 16; e += 64) SIGQ_RGW[e] = 0.f;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    for (int j = j0; j < j1; ++j, ++item) {
        // per-pair copies of the thread indices that the optimiser cannot see through: every index / address vector
        // built from them is recomputed inside the pair instead of becoming a loop invariant of the column loop that
        // is spilled and reloaded (one exposed scratch round trip each: measured 40 % of the kernel)
        int tidp = tid, lanep = lane;
        asm volatile("" : "+v"(tidp), "+v"(lanep));
        // ---- stage y_j (centred on its first point): fp64 rows + scaled norms, fp32 copy, both twice ----------
        __syncthreads();
        SIG_QSTAMP(10)
        constexpr int EPT = (128 * DPAD) / NT; // elements per thread: all loads of a thread are issued together
        double sv[EPT], sr[EPT];
        if (io64) {
#pragma unroll
            for (int k = 0; k < EPT; ++k) { // clamped indices + select below: no branch per element
                const int e = tidp + k * NT, t = min(e / DPAD, T - 1), c = min(e % DPAD, d - 1);
                sv[k] = static_cast<const double *>(a.Y)[((size_t)j * T + t) * d + c];
                sr[k] = static_cast<const double *>(a.Y)[(size_t)j * T * d + c];
            }
        } else {
            float fv[EPT], fr[EPT];
#pragma unroll
            for (int k = 0; k < EPT; ++k) {
                const int e = tidp + k * NT, t = min(e / DPAD, T - 1), c = min(e % DPAD, d - 1);
                fv[k] = static_cast<const float *>(a.Y)[((size_t)j * T + t) * d + c];
                fr[k] = static_cast<const float *>(a.Y)[(size_t)j * T * d + c];
            }
#pragma unroll
            for (int k = 0; k < EPT; ++k) sv[k] = (double)fv[k], sr[k] = (double)fr[k];
        }
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int e = tidp + k * NT;
            const bool ok = e / DPAD < T && e % DPAD < d;
            sv[k] = ok ? sv[k] : 0.0;
            sr[k] = ok ? sr[k] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int e = tidp + k * NT;
            const int t = e / DPAD, c = e % DPAD;
            const double r0 = sr[k];
            const double v = sv[k] - sr[k];
            const int r = 128 * (t >> 6) + (t & 63);
            yd[r * YDS + c] = v;
            yd[(r + 64) * YDS + c] = v;
            if (GRAD) {
                yf[r * YFS + c] = (float)v;
                yf[(r + 64) * YFS + c] = (float)v;
            }
            if (t == 0) yref[c] = r0;
            double s = v * v * nscale;
#pragma unroll
            for (int off = 1; off < DPAD; off <<= 1) s += __shfl_xor(s, off, 64);
            if (c == 0) {
                yd[r * YDS + DPAD] = s;
                yd[(r + 64) * YDS + DPAD] = s;
            }
        }
        __syncthreads();

        SIG_QSTAMP(7)
        if (row_ok && (!SYM || j >= i)) {
            float w_ij = 1.f, w_ji = 1.f; // row-side / column-side weights
            if (GRAD) {
                if (a.go) {
                    w_ij = (float)q_ldany(a.go, (size_t)i * a.B + j, io64);
                    if (SYM || a.symw) w_ji = (float)q_ldany(a.go, (size_t)j * a.B + i, io64);
                    if (a.symw) { w_ij += w_ji; w_ji = w_ij; }
                } else if (a.symw) {
                    w_ij = 2.f; w_ji = 2.f;
                }
                if (SYM && j == i) w_ji = 0.f; // diagonal pair: first-slot derivative only
            }

            // ---- G row 64 (the row beyond band 0): differences along the row for lane 63 of band 0 ------------
            {
                QExp7 ek = qexp7_coef();
                double xs2[DPAD], xn2 = 0.0, xr2[DPAD];
                q_ldrow<DPAD>(a.X, ((size_t)i * T + 64) * d, d, io64, xr2);
#pragma unroll
                for (int c = 0; c < DPAD; ++c) {
                    const double xc = (c < d) ? xr2[c] - yref[c] : 0.0;
                    xn2 = __builtin_fma(xc, xc, xn2);
                    xs2[c] = xc * (-2.0 * nscale);
                    if (GRAD && lanep == 0) x64[c] = (float)xc;
                }
                xn2 = __builtin_fma(xn2, nscale, -1.79248125036057809); // G / sqrt(12), as in the quadrant passes
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const double *yr = yd + (128 * hh + lanep) * YDS;
                    double e2 = xn2 + yr[DPAD];
#pragma unroll
                    for (int c = 0; c < DPAD; ++c) e2 = __builtin_fma(xs2[c], yr[c], e2);
                    g64[lanep + 64 * hh] = qexp2_p7(e2, ek);
                }
                if (GRAD) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) srow63[lanep + 64 * u] = 0.f; // (both seam rows: 256 floats)
                }
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): the row is in LDS before it is read back
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int n = lanep + 64 * hh;
                    rdh[n] = g64[n] - g64[(n - 1) & 127];
                }
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_s_waitcnt(0xc07f);
            }

            float Dsl[64]; // increments / sqrt(12) of the quadrant in work
            float Ssl[64]; // K_fwd, then S = K_fwd * U, of the quadrant in work
            float fc = 1.f, fuA = 1.f, fuB = 1.f, fV = 0.f; // forward chain of the band in work (K, neighbours, V)
            float svc = 1.f, svA = 1.f, svB = 1.f, svV = 0.f; // the same at the end of quadrant (0,0)
            int fwd_prev = -1;                               // quadrant (b + 2 h) of the previous forward sweep
            float rc = 1.f, rdA = 1.f, rdB = 1.f, rV = 0.f; // reverse chain (U)
            float cap0h0 = 0.f, cap63h0 = 0.f, cap0h1 = 0.f; // S[l][0], S[l][63] of half 0, S[l][64] (first of half 1)
            float xf[DPAD];
            // row-side contraction sums of the band in work: carried over its two reverse visits, sent when the band is done
            qf32x2 acc[DPAD / 2];
#pragma unroll
            for (int c = 0; c < DPAD / 2; ++c) acc[c] = qf32x2{0.f, 0.f};
            int rev_band = -1;
            bool kdone = false;
            float kmax = 1.f; // largest |K| this lane has seen on the pair's grid (boundary: 1)
            float cnd = 0.f;  // this lane's share of sum |S * gamma| (gradient launches, d <= 3) / sum |K_fwd * gamma| (FEW)
            float kfin_keep = 0.f;
            bool canc_keep = false;

            // visit list, 8 bits per visit: band | half << 1 | reverse << 2 | leave K[64][.] << 3 | increments << 4 (0 compute,
            // 1 compute and keep in the launch's scratch, 2 take from there) | scratch slot << 6
            unsigned long long vis = 0;
            int nv = 0;
            const int keep = a.dcache ? 1 : 0, take = a.dcache ? 2 : 0; // (from the uniform kernel argument: the visit codes
                                                                        //  must stay scalar, the EXEC windows are fetched by them)
            auto add = [&](int b, int h, int r, int hk, int dm, int ds) {
                if ((b ? nrows1 : 64) > 0 && (h ? nrows1 : 64) > 0) {
                    vis |= (unsigned long long)(b | (h << 1) | (r << 2) | (hk << 3) | (dm << 4) | (ds << 6)) << (8 * nv);
                    ++nv;
                }
            };
            if (GRAD) {
                if (nrows1 > 0) {
                    // band 0 forward (leaves K[64][.]; the lanes' state at the end of (0,0) is kept: sv*), band 1, then band 0
                    // again, where (0,1) restarts from the kept state instead of a third pass over (0,0).  The increments of
                    // the three quadrants that are visited twice make a round trip through L2 instead of being recomputed.
                    add(0, 0, 0, 1, keep, 0); add(0, 1, 0, 1, keep, 1); add(1, 0, 0, 0, keep, 2); add(1, 1, 1, 0, 0, 0);
                    add(1, 0, 1, 0, take, 2); add(0, 1, 1, 0, take, 1); add(0, 0, 1, 0, take, 0);
                } else {
                    add(0, 0, 1, 0, 0, 0);
                }
            } else {
                add(0, 0, 0, 1, 0, 0); add(0, 1, 0, 1, 0, 0); add(1, 0, 0, 0, 0, 0); add(1, 1, 0, 0, 0, 0);
            }
            const int b_last = nrows1 > 0 ? 1 : 0, h_last = nrows1 > 0 ? 1 : 0;

#pragma unroll 1
            for (int v = 0; v < nv; ++v) {
                const int code = (int)((vis >> (8 * v)) & 255);
                // a wave's priority falls as it advances through its visits: the wave that is behind on a SIMD gets the issue
                // slots, the two stay closer and wait less at the pair's closing barrier (5.31 -> 5.15 ms symmetric,
                // 6.67 -> 6.38 ordered at N=256, T=128, d=14)
                {
                    const int q4 = (4 * v) / nv;
                    if (q4 == 0) __builtin_amdgcn_s_setprio(3);
                    else if (q4 == 1) __builtin_amdgcn_s_setprio(2);
                    else if (q4 == 2) __builtin_amdgcn_s_setprio(1);
                    else __builtin_amdgcn_s_setprio(0);
                }
                const int b = code & 1, h = (code >> 1) & 1;
                const bool rev = (code & 4) != 0, leave_k = (code & 8) != 0;
                const int dmode = (code >> 4) & 3, dslot = code >> 6;
                const int nrows = b ? nrows1 : 64, ncols = h ? nrows1 : 64;
                QExp7 ek = qexp7_coef();
                int lv = lanep; // per-visit copy the optimiser cannot see through: address vectors built from it stay inside
                asm volatile("" : "+v"(lv)); // the visit instead of becoming spilled loop invariants
                const int m = 64 * b + lv; // point row of this lane
                const unsigned long long rows = nrows >= 64 ? ~0ull : ((1ull << nrows) - 1ull);
                const unsigned long long wr = ncols >= 64 ? ~0ull : (ncols > 0 ? ~0ull << (64 - ncols) : 0ull); // top ncols bits

                // ---- x_m, centred and pre-scaled (fp64 for the static kernel, fp32 for the gradient pass); re-read on
                // every visit (L2 hits) so that the fp64 copy is not live across the gradient pass
                double xs[DPAD], xn = 0.0;
                q_ldrow<DPAD>(a.X, ((size_t)i * T + min(m, P)) * d, d, io64, xs);
#pragma unroll
                for (int c = 0; c < DPAD; ++c) {
                    const double xc = (m <= P && c < d) ? xs[c] - yref[c] : 0.0;
                    xn = __builtin_fma(xc, xc, xn);
                    xs[c] = xc * (-2.0 * nscale);
                    xf[c] = (float)xc;
                }
                xn = __builtin_fma(xn, nscale, -1.79248125036057809); // - log2(sqrt(12)): D slots hold D / sqrt(12)

                SIG_QSTAMP(0)
                // ---- phase 1: G row (skewed: local column (t - lane) & 63 on iteration t) -> D slots --------------
                if (dmode == 2) { // second visit of the quadrant: the increments come back from the launch's scratch (L2)
                    const float *dp = dcw + dslot * 4096 + lv;
#pragma unroll
                    for (int k = 0; k < 64; ++k) Dsl[k] = dp[k * 64];
                    if (KSTORE) { // ... and the forward solution with them: no forward sweep on this visit
                        const float *kp = dcw + (3 + dslot) * 4096 + lv;
#pragma unroll
                        for (int k = 0; k < 64; ++k) Ssl[k] = kp[k * 64];
                    }
                } else {
                    const double *ybase = yd + (128 * h + 64 - lv) * YDS; // local column (t - lane) & 63 == ybase + t * YDS
                    // the point column that closes the last cell column of half 0 (column 64) is outside the ring
                    double g64v = 0.0;
                    if (h == 0) {
                        const double *yr = yd + 128 * YDS;
                        double e2 = xn + yr[DPAD];
#pragma unroll
                        for (int c = 0; c < DPAD; ++c) e2 = __builtin_fma(xs[c], yr[c], e2);
                        g64v = qexp2_p7(e2, ek);
                    }
                    const double *rdhh = rdh + 64 * h;
                    double g0 = 0.0, g1 = 0.0, gprev = 0.0, rdprev = 0.0;
                    // the y~ row of column t+1 is fetched into the SAME registers right after the dot product of column t:
                    // the exponential that follows covers the LDS latency, and there is no second row buffer to keep
                    double yrow[DPAD + 1];
#pragma unroll
                    for (int c = 0; c <= DPAD; ++c) yrow[c] = (c < DC || c == DPAD) ? ybase[c] : 0.0;
#pragma unroll
                    for (int t = 0; t < 66; ++t) {
                        double g;
                        if (t < 64) {
                            double e2 = xn + yrow[DPAD];
#pragma unroll
                            for (int c = 0; c < DC; ++c) e2 = __builtin_fma(xs[c], yrow[c], e2);
                            asm volatile("" : "+v"(e2));
                            if (t < 63) {
                                const double *yr = ybase + (t + 1) * YDS;
#pragma unroll
                                for (int c = 0; c <= DPAD; ++c)
                                    if (c < DC || c == DPAD) yrow[c] = yr[c];
                            }
                            qexp7_pin(ek);
                            g = qexp2_p7(e2, ek);
                            if (t == 0) g0 = g;
                            if (t == 1) g1 = g;
                        } else {
                            g = (t == 64) ? g0 : g1;
                        }
                        // a lane at local column 0 closes the previous row segment: G[m][64 h + 64] - G[m][64 h + 63]
                        const double gsel = (lv == (t & 63)) ? g64v : g;
                        const double rd = gsel - gprev; // G[m, c] - G[m, c-1]
                        gprev = g;
                        if (t >= 2) {
                            // lane l+1 holds the same column difference one iteration later; lane 63 of band 0 takes the
                            // row beyond the band (G row 64) from LDS: a virtual lane 64 is at local column t & 63
                            const int cc = (t & 63) ? (t & 63) : 64;
                            const double beyond = (64 * h + cc < 128) ? rdhh[cc] : 0.0;
                            const double nb = q_shl_keep(rd, beyond);
                            Dsl[(t - 2) & 63] = (float)(nb - rdprev);
                            asm volatile("" : "+v"(Dsl[(t - 2) & 63])); // formed here (hipcc otherwise sinks it to the sweep)
                        }
                        rdprev = rd;
                        __builtin_amdgcn_sched_barrier(0); // one column per scheduling region: bounds live ranges
                    }
                    if (dmode == 1) { // visited again later: 64 coalesced 256-B stores per wave
                        float *dp = dcw + dslot * 4096 + lv;
#pragma unroll
                        for (int k = 0; k < 64; ++k) dp[k * 64] = Dsl[k];
                    }
                }

                SIG_QSTAMP(1)
                // ---- phase 2: forward sweep of the quadrant -----------------------------------------------------
                if (!KSTORE || dmode != 2) {
                    const float *topb = (b ? hK : ones) + 64 * h; // K[64 b][64 h + q + 1] for lane 0 on step sigma = q
                    if (h == 0) { // a new band: left boundary column of ones
                        fc = 1.f;
                        fV = 0.f;
                        fuA = (lanep == 0) ? topb[1] : 1.f;
                        fuB = (lanep == 0) ? topb[0] : 1.f;
                    } else if (fwd_prev != b) { // right quadrant without its left one swept just before: kept state
                        fc = svc;
                        fV = svV;
                        fuA = svA;
                        fuB = svB;
                    }
                    fwd_prev = b + 2 * h;
                    // lane 63 leaves K[64 b + 64][64 h + q + 1] after step sigma = 63 + q
                    const unsigned ho = (unsigned)(size_t)(leave_k ? hK + 64 * h + 1 : hdummy + 63);
                    int haddr = (int)((lv == 63) ? ho : (unsigned)(size_t)(hdummy + lv));
                    const int hinc = (lanep == 63 && leave_k) ? 4 : 0;
                    float r3 = 1.7320508075688772f;
                    asm volatile("" : "+s"(r3));
#pragma unroll
                    for (int k = 0; k < 64; ++k) Ssl[k] = 0.f; // slots without a grid cell must read as S = 0
                    // The statements write the hand-over row in LDS but carry no "memory" clobber; what they write is read
                    // in LATER visits only, so one compiler barrier around the sweep is enough.
                    const float hbf = topb[lv + 2]; // lane l: the value lane 0 takes after step l
                    asm volatile("" ::: "memory");
                    quad_fwd_all<0, EARLY, FEW ? 1 : 0>(fc, fuA, fuB, fV, Dsl, Ssl, wr, rows, hbf, haddr, hinc, r3, nrows + ncols, cnd);
                    asm volatile("" ::: "memory");
                    if (!kdone) { // (the slots: K at the cells' upper left corners; fc: the row's last value so far)
                        kmax = q_max3_abs(kmax, fc, fc);
#pragma unroll
                        for (int k = 0; k < 64; k += 2) kmax = q_max3_abs(kmax, Ssl[k], Ssl[k + 1]);
                    }
                    if (KSTORE && dmode == 1) { // reverse sweep and gradient pass follow on a later visit
                        float *kp = dcw + (3 + dslot) * 4096 + lv;
#pragma unroll
                        for (int k = 0; k < 64; ++k) kp[k * 64] = Ssl[k];
                    }
                }
                if (b == 0 && h == 0) {
                    svc = fc;
                    svV = fV;
                    svA = fuA;
                    svB = fuB;
                }
                SIG_QSTAMP(2)
                if (!kdone && b == b_last && h == h_last) { // K[P][P]: last value of the last row with cells
                    kdone = true;
                    // a pair whose solution cancelled (see gram_fast.hip, resweep_fwd_fp64) is marked for the fp64 pass
                    const float kfin = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(fc), nrows - 1));
                    const float kden = fmaxf(fabsf(kfin), 0.1f);
                    // (round 4: without the floor "grid maximum > 2" -- a pair that decays from the boundary value 1 to K < 1 / ratio
                    //  loses the same ~2e-6 of the LARGEST value on its grid as one that oscillates)
                    bool fl = kmax > (d == 1 ? 2.f : d == 2 ? 4.f : QUAD_CANCEL_RATIO) * kden;
                    if constexpr (FEW) { // conditioning bound of a forward-only launch: sum |K_fwd D| max(grid maximum, 1) > 300 max(|K|, 0.1)
                        const float sds = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q_wave_sum63(cnd)), 63));
                        fl = fl || sds * 3.46410161513775459f * fmaxf(kmax, 1.f) > 300.f * kden;
                    }
                    const bool cancelled = __builtin_amdgcn_ballot_w64(kfin == kfin && fl) != 0;
                    if (lanep == nrows - 1) {
                        q_stany(a.K, (size_t)i * a.B + j, (double)fc, io64);
                        if (SYM && j != i) q_stany(a.K, (size_t)j * a.B + i, (double)fc, io64);
                        if (!GRAD && a.kflag) a.kflag[(size_t)i * a.B + j] = cancelled ? 1 : 0; // (gradient launches: after the last reverse sweep)
                    }
                    kfin_keep = kfin;
                    canc_keep = cancelled;
                }
                if (!GRAD || !rev) continue;

                // ---- phase 3: reverse sweep (S = K_fwd * U replaces K_fwd slot by slot) -------------------------
                {
                    const bool below = (b == 0) && nrows1 > 0;    // band 1 lies below: U[64][.] is in hU
                    const float *botb = (below ? hU : ones) + 64 * h; // U[64 b + 64][64 h + q] for lane 63 on step sigma = q + 63
                    if (rev_band != b) { // first reverse quadrant of the band: right boundary column of ones
                        rev_band = b;
#pragma unroll
                        for (int c = 0; c < DPAD / 2; ++c) acc[c] = qf32x2{0.f, 0.f};
                        rc = 1.f;
                        rV = 0.f;
                        // lane 63 starts on step 62 + ncols with the lower neighbour U[64 b + 64][64 h + ncols - 1] and the
                        // corner U[.][64 h + ncols]; odd steps shift into dnA, even steps into dnB
                        const float c1 = botb[ncols], c0 = botb[ncols - 1];
                        const bool odd = ((62 + ncols) & 1) != 0;
                        rdA = (lanep == 63) ? (odd ? c0 : c1) : 1.f;
                        rdB = (lanep == 63) ? (odd ? c1 : c0) : 1.f;
                    }
                    // lane 0 of band 1 leaves U[64][64 h + q] after step sigma = q
                    const bool leave_u = (b == 1);
                    const unsigned ho = (unsigned)(size_t)(leave_u ? hU + 64 * h + ncols - 1 : hdummy);
                    int haddr = (int)((lv == 0) ? ho : (unsigned)(size_t)(hdummy + lv));
                    const int hinc = (lanep == 0 && leave_u) ? -4 : 0;
                    float r3 = 1.7320508075688772f;
                    asm volatile("" : "+s"(r3));
                    // after step sigma >= 64 lane 63 takes the boundary value of step sigma - 1, U[.][64 h + sigma - 64];
                    // after step 63 a right quadrant hands it the first value of the left one (index -1)
                    const float hbr = botb[lv], bmr = botb[-h];
                    asm volatile("" ::: "memory");
                    quad_rev_all<124, EARLY>(rc, rdA, rdB, rV, Dsl, Ssl, wr, rows, hbr, bmr, haddr, hinc, r3, nrows + ncols);
                    asm volatile("" ::: "memory");
                }
                // conditioning, paths in <= 3 channels: this quadrant's share of sum |S * gamma| (slots without a cell hold S = 0)
                if (DPAD == 8 && d <= 3) {
                    float cs = 0.f;
#pragma unroll
                    for (int k = 0; k < 64; ++k) cs = __builtin_fmaf(fabsf(Ssl[k]), fabsf(Dsl[k]), cs);
                    cnd += cs;
                }
                SIG_QSTAMP(3)
                // ---- seam rows for the hand-over pass: S[63][.] (band 0, lane 63), S[64][.] (band 1, lane 0) ------
                // (slot k of lane l is local column (k - l) & 63; the index is formed from scalars: per-lane index vectors
                //  are loop invariants that hipcc hoists out of the pair loop and spills -- 64 serialised scratch loads)
                if (lanep == (b ? 0 : 63)) {
                    float *dst = (b ? srow64 : srow63) + 64 * h;
                    const int off = b ? 0 : 1;
#pragma unroll
                    for (int k = 0; k < 64; ++k) dst[(k + off) & 63] = Ssl[k];
                }

                // ---- phase 4: 4-corner scatter R, static kernel in fp32, both contractions -----------------------
                // iteration it: lane l is at local column n = (it - l) & 63; own row S[l][n] is slot it, the upper row
                // arrives through a wave shift one column ahead (lane l-1's slot it holds S[l-1][n+1]), hence the
                // two-deep history of the shifted values.  Local point column 0 needs the cell column left of the
                // quadrant: it is masked here and done per band from the values captured at the wrap.
                const float rowmask = (b == 1 && lanep == 0) ? 0.f : 1.f; // point row 64 is contracted in the seam pass
                const float ns32 = (float)nscale;
                {
                    float tacc[DPAD]; // column-side travelling sums (SYM)
#pragma unroll
                    for (int c = 0; c < DPAD; ++c) tacc[c] = 0.f;
                    float capA = 0.f, capB = 0.f;
                    float Nc = q_shr_zero(Ssl[62]); // S[l-1][n-1]
                    float Nb = q_shr_zero(Ssl[63]); // S[l-1][n]
                    float Sprev = Ssl[63];          // S[l][n-1]
                    const float *yfb = yf + (128 * h + 64 - lv) * YFS;
                    qf32x2 ynx[DPAD / 2]; // the y~ row of the next iteration (fetched one iteration ahead)
#pragma unroll
                    for (int c = 0; c < DPAD / 2; ++c) ynx[c] = c < DC / 2 ? reinterpret_cast<const qf32x2 *>(yfb)[c] : qf32x2{0.f, 0.f};
                    SIG_QSTAMP(6)
#pragma unroll
                    for (int it = 0; it < 64; ++it) {
                        qf32x2 yr2[DPAD / 2];
#pragma unroll
                        for (int c = 0; c < DPAD / 2; ++c) yr2[c] = ynx[c];
                        if (it < 63) {
                            const qf32x2 *yn = reinterpret_cast<const qf32x2 *>(yfb + (it + 1) * YFS);
#pragma unroll
                            for (int c = 0; c < DC / 2; ++c) ynx[c] = yn[c];
                        }
                        const float Scur = Ssl[it];
                        const float Na = q_shr_zero(Scur); // S[l-1][n+1]
                        const bool wrap = lv == it;        // local column 0
                        float R = ((Nc - Nb) + (Scur - Sprev)) * rowmask;
                        capA = wrap ? Scur : capA;
                        capB = wrap ? Sprev : capB;
                        R = wrap ? 0.f : R;
                        Nc = Nb;
                        Nb = Na;
                        Sprev = Scur;
                        qf32x2 df[DPAD / 2];
                        const float gv = q_gval<DPAD, DC>(xf, yr2, ns32, df);
                        const float rg = R * gv;
                        const qf32x2 rg2 = {rg, rg};
#pragma unroll
                        for (int c = 0; c < DC / 2; ++c) acc[c] = __builtin_elementwise_fma(rg2, df[c], acc[c]);
                        // pin the running sums here: the contraction must stay inside its iteration
#pragma unroll
                        for (int c = 0; c < DC / 2; ++c) asm volatile("" : "+v"(acc[c]));
                        if (SYM) {
                            const float rgw = rg * w_ji;
                            const qf32x2 rgw2 = {rgw, rgw};
#pragma unroll
                            for (int c = 0; c < DC / 2; ++c) { // (packed products: half the multiplies)
                                const qf32x2 pr = rgw2 * df[c];
                                tacc[2 * c] = q_add_ror1(tacc[2 * c], pr[0]);
                                tacc[2 * c + 1] = q_add_ror1(tacc[2 * c + 1], pr[1]);
                            }
#pragma unroll
                            for (int c = 0; c < DC; ++c) asm volatile("" : "+v"(tacc[c]));
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    SIG_QSTAMP(4)
                    if (h == 0) {
                        cap0h0 = capA;
                        cap63h0 = capB;
                    } else {
                        cap0h1 = capA;
                    }
                    if (SYM) { // the finished sums of local column (63 - lane) & 63: this pass's record, [value][lane]
                        float *dst = SIGQ_CRW + (2 * b + h) * (CS * 64) + lv;
#pragma unroll
                        for (int c = 0; c < DPAD; ++c) dst[c * 64] = tacc[c];
                    }
                }

                SIG_QSTAMP(8)
                // ---- the band's seam columns (after its left quadrant): point columns 0 and 64 ---------------------
                if (h == 0) {
#pragma unroll
                    for (int sc = 0; sc < 2; ++sc) {
                        // E_l = S[l][n-1] - S[l][n];  R[m][n] = E_{l-1} - E_l
                        const float E = sc ? (cap63h0 - cap0h1) : -cap0h0;
                        const float R = (q_shr_zero(E) - E) * rowmask;
                        qf32x2 ys2[DPAD / 2];
#pragma unroll
                        for (int c = 0; c < DPAD / 2; ++c) ys2[c] = reinterpret_cast<const qf32x2 *>(yf + (128 * sc) * YFS)[c];
                        qf32x2 dfs[DPAD / 2];
                        const float rg = R * q_gval<DPAD>(xf, ys2, ns32, dfs);
                        const qf32x2 rg2 = {rg, rg};
#pragma unroll
                        for (int c = 0; c < DPAD / 2; ++c) acc[c] = __builtin_elementwise_fma(rg2, dfs[c], acc[c]);
                        if (SYM) { // one column, 64 rows: wave sums, lane c adds channel c (all lanes on one address would
                                   // serialise 64-fold in LDS)
                            const float rgw = rg * w_ji;
#pragma unroll
                            for (int c = 0; c < DPAD; ++c) {
                                const float vsum = q_wave_sum63(rgw * dfs[c / 2][c % 2]);
                                if (lv == 63) SIGQ_CRW[6 * CS * 64 + (2 * b + sc) * CS + c] = vsum; // seam record [band][column 0 / 64]
                            }
                        }
                    }
                    cap0h0 = cap63h0 = cap0h1 = 0.f;
                    // the band is done: its row-side gradient d k(x_i, y_j) / d x_i[m] goes to the fp64 accumulation buffer
                    // (point row 64 comes from the seam pass below)
                    if (m <= P && !(b == 1 && lv == 0)) {
                        // (no-return adds so that nothing has to be loaded here; row m of this particle is touched by this
                        //  lane only, so they execute in program order and the sums do not depend on timing)
                        if (rowlds) {
                            float *dst = rowacc + m * RS;
#pragma unroll
                            for (int c = 0; c < RS; ++c)
                                if (c < d) atomicAdd(dst + c, w_ij * m2h * acc[c / 2][c % 2]);
                        } else {
                            float *dst = SIGQ_RGW + m * 16;
#pragma unroll
                            for (int c = 0; c < DPAD; ++c)
                                if (c < d) unsafeAtomicAdd(dst + c, w_ij * m2h * acc[c / 2][c % 2]);
                        }
                    }
                }
                SIG_QSTAMP(5)
            } // quadrant visits

            SIG_QSTAMP(0)
            if (GRAD && a.kflag) {
                // the pair's verdict for the exact fp64 pass: cancellation of magnitudes (forward sweep) or, in <= 3 channels,
                // the condition number c1 = sum |S D| / max(|K|, 0.1) > 150 (gram_fast.hip, "conditioning")
                bool ill = false;
                if (DPAD == 8 && d <= 3) {
                    const float c1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q_wave_sum63(cnd)), 63));
                    ill = kfin_keep == kfin_keep && c1 * 3.46410161513775459f > 150.f * fmaxf(fabsf(kfin_keep), 0.1f);
                }
                if (lanep == 0) a.kflag[(size_t)i * a.B + j] = (canc_keep || ill) ? 1 : 0;
            }
            if (GRAD) {
                // ---- seam: point row 64.  R[64][n] = (S[63][n-1] - S[63][n]) - (S[64][n-1] - S[64][n]), formed from both
                // bands' rows before the contraction; lanes take columns n = lane and lane + 64.
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_s_waitcnt(0xc07f);
                float part[DPAD], xm[DPAD];
#pragma unroll
                for (int c = 0; c < DPAD; ++c) {
                    part[c] = 0.f;
                    xm[c] = x64[c];
                }
                const float ns32 = (float)nscale;
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int n = lanep + 64 * hh;
                    const float Sa = n ? srow63[n - 1] : 0.f, Sz = srow63[n];
                    const float Ta = n ? srow64[n - 1] : 0.f, Tz = srow64[n];
                    const float *yr = yf + (128 * hh + lanep) * YFS;
                    float e2 = 0.f;
#pragma unroll
                    for (int c = 0; c < DPAD; ++c) e2 = __builtin_fmaf(xm[c] - yr[c], xm[c] - yr[c], e2);
                    const float rgn = (n <= P) ? ((Sa - Sz) - (Ta - Tz)) * __builtin_amdgcn_exp2f(e2 * ns32) : 0.f;
#pragma unroll
                    for (int c = 0; c < DPAD; ++c) part[c] = __builtin_fmaf(rgn, xm[c] - yr[c], part[c]);
                    if (SYM) { // record of point row 64: [half][value][lane], lane = local column
                        float *dst = SIGQ_CRW + (4 + hh) * (CS * 64) + lanep;
#pragma unroll
                        for (int c = 0; c < DPAD; ++c) dst[c * 64] = rgn * w_ji * (xm[c] - yr[c]);
                    }
                }
                // row 64 belongs to band 1's lane 0 accumulators
#pragma unroll
                for (int c = 0; c < DPAD; ++c) {
                    const float v = q_wave_sum63(w_ij * m2h * part[c]); // total in lane 63
                    if (lanep == 63 && c < d) { // (row 64 receives nothing else: band 1's lane 0 is masked out of the band flush)
                        if (rowlds) atomicAdd(rowacc + 64 * RS + min(c, RS - 1) + (lanep - 63), v);
                        else unsafeAtomicAdd(SIGQ_RGW + 64 * 16 + c + (lanep - 63), v);
                    }
                }
            }
            SIG_QSTAMP(6)
        } // this wavefront's pair

        if (GRAD && SYM) {
            // close the column-side sums of y_j over the rows of the tile:
            // d/dy_n = -(2/h) * sum_m w R G (y~_n - x~_m) = +(2/h) * (the recorded sums of w R G (x~_m - y~_n))
            // The eight wavefronts' records are added in wave order (rounds 1-2 joined them with LDS atomics, whose order
            // -- and with it the last bits of the gradient -- changed from run to run), and the item's sums go to their
            // own row of the column slab.
            __syncthreads(); // (also makes the other wavefronts' records visible: workgroup-scope release / acquire)
            float *dstc = a.cslab + (size_t)item * (T * d);
            // thread -> (channel c, point n) with n fastest: a wavefront reads 64 consecutive floats of a record per load
            // (with the channel fastest every lane touched its own 256-B line), all 24 loads of an element in flight;
            // plain loads: the records were written on this CU, whose L1 is coherent with its own stores
            for (int e = tidp; e < 128 * DPAD; e += NT) {
                const int c = e >> 7, n = e & 127;
                const int hq = n >> 6, q = n & 63, ln = (63 - q) & 63;
                float v1[QNW], v0[QNW], v64[QNW];
#pragma unroll
                for (int w = 0; w < QNW; ++w) {
                    const float *rb = a.crec + ((size_t)blockIdx.x * QNW + w) * QREC;
                    v1[w] = rb[((2 + hq) * CS + c) * 64 + ln]; // pass over quadrant (1, hq)
                    v0[w] = rb[(hq * CS + c) * 64 + ln];       // pass over quadrant (0, hq)
                    v64[w] = rb[((4 + hq) * CS + c) * 64 + q]; // point row 64
                }
                float sx = 0.f;
#pragma unroll
                for (int w = 0; w < QNW; ++w) {
                    const bool has = i0 + w < a.A && j >= i0 + w; // that wavefront had a pair
                    float t = 0.f;
                    if (nrows1 > 0) t += v1[w];
                    if (hq == 0 || nrows1 > 0) t += v0[w];
                    t += v64[w];
                    sx += has ? t : 0.f;
                }
                if (q == 0) { // seam columns 0 and 64, per band
#pragma unroll 1
                    for (int w = 0; w < QNW; ++w) {
                        if (i0 + w >= a.A || j < i0 + w) continue;
                        const float *rb = a.crec + ((size_t)blockIdx.x * QNW + w) * QREC + 6 * CS * 64;
                        sx += rb[hq * CS + c];
                        if (nrows1 > 0) sx += rb[(2 + hq) * CS + c];
                    }
                }
                if (n < T && c < d) dstc[n * d + c] = -m2h * sx;
            }
            SIG_QSTAMP(9)
        }
    }
    if (GRAD && row_ok) { // the segment's row-side sums: consecutive lanes on consecutive addresses
        int lf = lane;
        asm volatile("" : "+v"(lf));
        const int tot = T * d;
        double *dstr = a.rseg + (((size_t)(kq + (int)blockIdx.x)) * QNW + wave) * (size_t)tot;
        if (!rowlds) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wavefront's read-modify-writes have landed
        for (int e = lf; e < tot; e += 64) {
            const int m = e / d, c = e - m * d;
            // (the global accumulator was updated at the L2: read it there, not from a stale L1 line)
            dstr[e] = (double)(rowlds ? rowacc[m * RS + c]
                                      : __hip_atomic_load(SIGQ_RGW + m * 16 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
    }
    SIG_QSTAMP(11)
    remaining -= ncol;
    ++kq;
    cstart = 0;
    } // row tiles of the range

    SIG_QSTAMP(0)
#ifdef SIGSVGD_PHASE_STAMPS
    if (lane == 0 && a.stamps)
        for (int k = 0; k < 12; ++k) atomicAdd(&a.stamps[k], ph_[k]);
#endif
}

bool quad_supported(int A, int B, int T, int d, int n, int kind, unsigned flags)
{
    (void)A; (void)B;
    if (n != 0 || T < 65 || T > 128 || d > 16) return false;
    if (kind != SIGSVGD_STATIC_RBF) return false;
    if (flags & SIGSVGD_FLAG_NAIVE_SOLVER) return false;
    return true;
}

namespace {
inline int quad_cu_count() { return device_cu_count(); } // the grid is at most one workgroup per CU
constexpr size_t QUAD_DCACHE_PER_WG = (size_t)QNW * 6 * 64 * 64 * sizeof(float); // 768 KB: increments (+ forward solution, 8-channel layout) of 3 quadrants
constexpr size_t QUAD_CREC_PER_WG = (size_t)QNW * QREC * sizeof(float);          // 208 KB
constexpr size_t QUAD_ROWG_PER_WG = (size_t)QNW * 128 * 16 * sizeof(float);      // 64 KB (d = 15, 16 only)

inline GradGeom quad_geometry(int A, int B, int T, int d, bool sym, int off = 0, int stride = 1, bool fold = false)
{
    return grad_geometry(A, B, T * d, sym, off, stride, fold, QNW, (long long)quad_cu_count());
}
inline GradGeom quad_geometry(int A, int B, int T, int d, bool sym, const TileMap &tm)
{
    return quad_geometry(A, B, T, d, sym, tm.off, tm.stride, tm.fold != 0);
}
// workspace of a gradient launch: [row segments][column slab][column records][row accumulators (d >= 15)][increment scratch]
struct QuadCut {
    size_t rseg, cslab, crec, rowg, dcache, total;
};
inline QuadCut quad_cut(const GradGeom &g, int d, bool sym)
{
    QuadCut c;
    const size_t ncu = (size_t)quad_cu_count();
    c.rseg = 0;
    c.cslab = c.rseg + g.rseg_bytes;
    c.crec = c.cslab + g.cslab_bytes;
    c.rowg = c.crec + (sym ? ncu * QUAD_CREC_PER_WG : 0);
    c.dcache = c.rowg + (d > 14 ? ncu * QUAD_ROWG_PER_WG : 0);
    c.total = c.dcache + ncu * QUAD_DCACHE_PER_WG;
    return c;
}
} // namespace

// every launch: [A][B] bytes of cancellation flags + the (small) workspace of the fp64 pass over the flagged pairs, in
// front of the gradient launch's own areas
inline size_t quad_flag_bytes(int A, int B) { return (((size_t)A * B + 255) & ~(size_t)255) + generic_repair_bytes(); }

int quad_workspace_bytes(int A, int B, int T, int d, int want_grad, size_t *bytes)
{
    *bytes = quad_flag_bytes(A, B) + 512;
    if (!want_grad) return SIGSVGD_OK;
    // the larger of the ordered and the symmetric launch (the query carries no Y_IS_X promise); the increment scratch of a
    // full grid is 100 MB on 256 CUs and lives in L2 / MALL
    size_t need = quad_cut(quad_geometry(A, B, T, d, false), d, false).total;
    if (A == B) {
        const size_t y = quad_cut(quad_geometry(A, B, T, d, true), d, true).total;
        if (y > need) need = y;
    }
    *bytes = need + quad_flag_bytes(A, B) + 512;
    return SIGSVGD_OK;
}

namespace {
template <int DPAD>
int quad_launch_variant(const GramProblem &p, QuadArgs &a, bool grad, bool sym)
{
    const GradGeom g = quad_geometry(p.A, p.B, p.T, p.d, sym, a.tm);
    if (g.tm.owned <= 0 || g.nitems <= 0) return SIGSVGD_OK;
    a.tm = g.tm;
    a.nitems = g.nitems;
    dim3 grid((unsigned)g.grid), block(QNW * 64);
#ifdef SIGSVGD_PHASE_STAMPS
    {
        static unsigned long long *dbg = nullptr;
        if (!dbg) (void)hipMalloc(&dbg, 12 * sizeof(unsigned long long));
        (void)hipMemsetAsync(dbg, 0, 12 * sizeof(unsigned long long), p.stream);
        a.stamps = dbg;
    }
#endif
    constexpr bool HAS_ROWG = DPAD == 16;
    const bool rowg = HAS_ROWG && grad && p.d > 14;
    const bool early = p.T <= 112;
#define SIGQ_LAUNCH(G, S, R, E) hipLaunchKernelGGL((gram_quad_kernel<DPAD, G, S, R, E>), grid, block, 0, p.stream, a)
#define SIGQ_LAUNCH_FEW(S, E) hipLaunchKernelGGL((gram_quad_kernel<8, false, S, false, E, true>), grid, block, 0, p.stream, a)
    const bool few = DPAD == 8 && !grad && p.d <= 3;
    if (grad && sym && rowg)
        SIGQ_LAUNCH(true, true, HAS_ROWG, false);
    else if (grad && rowg)
        SIGQ_LAUNCH(true, false, HAS_ROWG, false);
    else if (grad && sym)
        { if (early) SIGQ_LAUNCH(true, true, false, true); else SIGQ_LAUNCH(true, true, false, false); }
    else if (grad)
        { if (early) SIGQ_LAUNCH(true, false, false, true); else SIGQ_LAUNCH(true, false, false, false); }
    else if (few && sym)
        { if (early) SIGQ_LAUNCH_FEW(true, true); else SIGQ_LAUNCH_FEW(true, false); }
    else if (few)
        { if (early) SIGQ_LAUNCH_FEW(false, true); else SIGQ_LAUNCH_FEW(false, false); }
    else if (sym)
        { if (early) SIGQ_LAUNCH(false, true, false, true); else SIGQ_LAUNCH(false, true, false, false); }
    else
        { if (early) SIGQ_LAUNCH(false, false, false, true); else SIGQ_LAUNCH(false, false, false, false); }
#undef SIGQ_LAUNCH
#undef SIGQ_LAUNCH_FEW
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "launch gram_quad_kernel");
#ifdef SIGSVGD_PHASE_STAMPS
    {
        unsigned long long hst[12];
        (void)hipStreamSynchronize(p.stream);
        (void)hipMemcpy(hst, a.stamps, sizeof(hst), hipMemcpyDeviceToHost);
        double tot = 0;
        for (int k = 0; k < 12; ++k) tot += (double)hst[k];
        static const char *nm[12] = {"staging/other", "phase 1 static kernel", "forward sweep", "reverse sweep",
                                     "gradient pass", "seams + row sums", "row seam (+ gradient-pass prologue)", "Y staging",
                                     "column-sum LDS adds", "closing barrier + column-side flush", "barrier before staging", "row-side flush"};
        fprintf(stderr, "[phase stamps quad] A=%d T=%d d=%d grad=%d sym=%d: ", p.A, p.T, p.d, (int)grad, (int)sym);
        for (int k = 0; k < 12; ++k) fprintf(stderr, "%s %.1f%% | ", nm[k], 100.0 * (double)hst[k] / tot);
        fprintf(stderr, "total %.3e wave-cycles\n", tot);
    }
#endif
    return SIGSVGD_OK;
}

int quad_dispatch(const GramProblem &p, QuadArgs &a, bool grad, bool sym)
{
    if (p.d <= 8) return quad_launch_variant<8>(p, a, grad, sym);
    return quad_launch_variant<16>(p, a, grad, sym);
}

void quad_fill_args(const GramProblem &p, QuadArgs &a)
{
    a.X = p.X; a.Y = p.Y; a.go = p.grad_out; a.K = p.K_out;
    a.rseg = nullptr; a.cslab = nullptr; a.crec = nullptr; a.rowg = nullptr;
    a.io64 = p.dtype == SIGSVGD_F64; a.A = p.A; a.B = p.B; a.T = p.T; a.d = p.d;
    a.symw = (p.flags & SIGSVGD_FLAG_SYM) ? 1 : 0; a.inv_h = p.inv_h;
    a.tm = make_tilemap(1, 0, 1, false); // (a full launch; quad_launch_variant derives the tile count)
    a.nitems = 0; a.dcache = nullptr; a.kflag = nullptr;
}

// the flag area at the head of the workspace (every launch has one), and the fp64 pass that follows the kernel
unsigned char *quad_ws_base(const GramProblem &p, size_t need)
{
    if (!p.ws || p.ws_bytes < need + quad_flag_bytes(p.A, p.B) + 256) {
        set_error("quad: workspace %zu B < required %zu B", p.ws_bytes, need + quad_flag_bytes(p.A, p.B) + 256);
        return nullptr;
    }
    return reinterpret_cast<unsigned char *>((reinterpret_cast<uintptr_t>(p.ws) + 255) & ~(uintptr_t)255);
}
int quad_repair(const GramProblem &p, const QuadArgs &a, bool sym)
{
    return generic_repair_launch(p, a.kflag, a.kflag + (((size_t)p.A * p.B + 255) & ~(size_t)255), sym, a.tm, QNW);
}

// cut the workspace, enqueue kernel + fixed-order reduction into `out` (the I/O type, or fp64 for the partial solve)
int quad_run_grad(const GramProblem &p, QuadArgs &a, bool sym, void *out, int out64)
{
    const GradGeom g = quad_geometry(p.A, p.B, p.T, p.d, sym, a.tm);
    const QuadCut c = quad_cut(g, p.d, sym);
    unsigned char *base = quad_ws_base(p, c.total);
    if (!base) return SIGSVGD_E_WORKSPACE;
    a.kflag = base;
    base += quad_flag_bytes(p.A, p.B);
    a.rseg = reinterpret_cast<double *>(base + c.rseg);
    a.cslab = sym ? reinterpret_cast<float *>(base + c.cslab) : nullptr;
    a.crec = sym ? reinterpret_cast<float *>(base + c.crec) : nullptr;
    a.rowg = p.d > 14 ? reinterpret_cast<float *>(base + c.rowg) : nullptr;
    a.dcache = reinterpret_cast<float *>(base + c.dcache);
    int rc = quad_dispatch(p, a, true, sym);
    if (rc) return rc;
    a.tm = g.tm;
    rc = quad_repair(p, a, sym);
    if (rc) return rc;
    return grad_reduce_launch(g, a.rseg, a.cslab, out, out64, p.A, p.B, p.T * p.d, sym, p.stream);
}
} // namespace

int quad_launch(const GramProblem &p)
{
    const bool grad = p.gradX_out != nullptr;
    const bool sym = (p.flags & SIGSVGD_FLAG_Y_IS_X) && p.A == p.B; // Y is X: each unordered pair once
    QuadArgs a;
    quad_fill_args(p, a);
    if (a.symw && p.A != p.B) {
        set_error("sym backward needs A == B");
        return SIGSVGD_E_BADARG;
    }
    if (!grad) {
        unsigned char *base = quad_ws_base(p, 0);
        if (!base) return SIGSVGD_E_WORKSPACE;
        a.kflag = base;
        int rc = quad_dispatch(p, a, false, sym);
        if (rc) return rc;
        a.tm = quad_geometry(p.A, p.B, p.T, p.d, sym, a.tm).tm;
        return quad_repair(p, a, sym);
    }
    return quad_run_grad(p, a, sym, p.gradX_out, p.dtype == SIGSVGD_F64);
}

// Sharded partial solve (sigsvgd_gram_sym_partial) for the long-path shapes: row tiles of 8 rows, tiles
// tile_offset + k * tile_stride, both orientations of K stored into the caller-zeroed K_partial; grad_partial (fp64) is
// OVERWRITTEN with this launch's share of the gradient.
int quad_sym_partial(const GramProblem &p, int tile_offset, int tile_stride, bool fold, double *grad_partial)
{
    if (tile_stride < 1 || tile_offset < 0 || tile_offset >= tile_stride) {
        set_error("bad tile_offset/tile_stride %d/%d", tile_offset, tile_stride);
        return SIGSVGD_E_BADARG;
    }
    QuadArgs a;
    quad_fill_args(p, a);
    a.tm = make_tilemap((p.A + QNW - 1) / QNW, tile_offset, tile_stride, fold);
    return quad_run_grad(p, a, true, grad_partial, 1);
}

} // namespace sigsvgd

SIG_EXEC_DEBUG_GETTER(quad)
