// Signature-kernel Gram forward/backward for LONG paths: dyadic order 0, 65 <= T <= 128, d <= 16, RBF,
// second-order stencil (BASELINE.json config C5: T = 128, d = 14).  Stored forward solution -- no regeneration,
// hence no limit on how rough the paths may be (the kernel this replaces re-ran the stencil backwards and had to
// refuse pairs whose static-kernel increments exceeded 0.4).
//
// Mapping: one wavefront per trajectory pair, ONE wavefront per SIMD (512-register budget).  The (T-1)^2 grid is
// cut into two bands of <= 64 rows; lane l owns row 64 b + l of band b.  Within a band everything is the
// register-resident scheme of gram_fast.hip on a ring of 128 columns: on anti-diagonal sigma = row + column lane l
// is at column sigma - l, so per-cell state is filed under slot sigma & 127, a compile-time constant of the
// unrolled step -- D (increments / sqrt(12)) in 128 VGPRs, K_fwd -> S = K_fwd * U in another 128.  Neighbour rows
// move with wave-wide DPP shifts; the sweeps run in the fp32 difference form (see gram_fast.hip, phase 2).
// A pair is solved in three band passes so that only ONE band's D and S are live at a time (2 x 128 registers):
//     pass 0  band 0: static kernel + increments, forward sweep; keeps only its last row K[64][.]   (LDS hand-over)
//     pass 1  band 1: static kernel + increments, forward sweep from K[64][.] (K stored), K[P][P] written;
//                     reverse sweep (S), U[64][.] handed over, gradient pass for rows 65..T-1
//     pass 2  band 0: static kernel + increments and forward sweep AGAIN (K stored), reverse sweep from U[64][.],
//                     gradient pass for rows 0..63
//     seam            row 64 takes the last S row of band 0 and the first of band 1: its 4-corner scatter is formed
//                     from both BEFORE the contraction (the halves are large and nearly cancel), one dense pass.
// Recomputing band 0 costs one extra static-kernel pass and forward sweep (~25 % of a pair); holding both bands
// would need 512 slot registers or a round trip of 100 KB per pair through L2.
// The static kernel G is NOT stored for the gradient pass: it is re-evaluated there in fp32 from the centred
// coordinates (8 packed FMAs + v_exp_f32 per cell at d = 14; it only weights the contraction, 1e-6 suffices).
// Column-side sums (Y is X: d k(x_j,x_i)/d x_j) go straight into a [column][channel] image in LDS shared by the
// four wavefronts of the workgroup (ds_add_f32; row stride 17 floats, so the 64 lanes of an instruction hit
// distinct banks), closed and flushed once per column trajectory.
//
// Reference semantics: sigkernel _SigKernelGram.forward/backward [RECALLED, SURVEY.md App. A];
// static kernel src/kernels/_traj_kernels.py:176-195.
#include "sig_common.h"

namespace sigsvgd {

struct BandArgs {
    const void *X, *Y, *go;
    void *K;
    double *gacc; // [A][T][d] fp64, zeroed by the launcher (or the caller's accumulating buffer: partial solve)
    int io64, A, B, T, d, JC, symw;
    int tile_offset, tile_stride; // row tiles tile_offset + k * tile_stride are solved (sharded partial solve)
    double inv_h;
};

namespace {
constexpr int BNW = 4;    // wavefronts (rows i) per workgroup
constexpr int RING = 128; // slots per band = longest supported path
constexpr int NSTEP = 192; // sweep steps executed per band (>= 64 + P - 1 for P <= 127), a multiple of 4
using bf32x2 = __attribute__((ext_vector_type(2))) float;

__device__ __forceinline__ double b_ldany(const void *b, size_t i, int io64)
{
    return io64 ? static_cast<const double *>(b)[i] : (double)static_cast<const float *>(b)[i];
}
__device__ __forceinline__ void b_stany(void *b, size_t i, double v, int io64)
{
    if (io64)
        static_cast<double *>(b)[i] = v;
    else
        static_cast<float *>(b)[i] = (float)v;
}
// lane l <- lane l+1; lane 63 keeps `old` (compiler-visible DPP: hipcc pads its hazards)
__device__ __forceinline__ double b_shl_keep(double v, double old)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), 0x130, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), 0x130, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float b_shr_zero(float v) // lane l <- lane l-1, lane 0 gets 0
{
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x138, 0xF, 0xF, true));
}

// ---- four steps of a sweep, hand-scheduled (cf. gram_fast.hip): fp32 difference form, no scalar instruction, the
// mask handled with v_cndmask.  The increments live in ACCUMULATOR registers ("a" operands, one v_accvgpr_read per
// step): with one wave per SIMD the 512-entry file is 256 arch + 256 acc registers, and D (128) + S (128) + the
// working set do not fit the arch half -- left to itself hipcc shuffles slots between the halves and scratch
// around every asm statement (measured: 684 spilled registers, 3x the time).  Additions for the banded grid: the lane without a DPP source (0 forward, 63 reverse)
// takes its neighbour row from a boundary value `bt` moved into the shift destination BEFORE the compare and the
// counter update (2 wait states ahead of the DPP instruction that reads it as `old`); the hand-over lane's new
// value goes to LDS through a per-lane address (every other lane writes to a scratch row).
#define SIG_B_FWD(UP, DIAG, G, BT, KSL)                                                       \
    "v_mov_b32 %[" UP "], %[" BT "]\n\t"                                                      \
    "v_cmp_gt_u32 vcc, %[P], %[cnt]\n\t"                                                      \
    "v_add_u32 %[cnt], 1, %[cnt]\n\t"                                                         \
    "v_mov_b32_dpp %[" UP "], %[cur] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"               \
    "v_add_f32 %[t], %[cur], %[" UP "]\n\t"                                                   \
    "v_mul_f32 %[y], %[r3], %[t]\n\t"                                                         \
    "v_accvgpr_read_b32 %[ge], %[" G "]\n\t"                                                  \
    "v_cndmask_b32 %[ge], 0, %[ge], vcc\n\t"                                                  \
    "v_add_f32 %[t], %[t], %[" DIAG "]\n\t"                                                   \
    "v_fmac_f32 %[y], %[t], %[ge]\n\t"                                                        \
    "v_fmac_f32 %[V], %[ge], %[y]\n\t"                                                        \
    "v_add_f32 %[cur], %[" UP "], %[V]\n\t"                                                   \
    "ds_write_b32 %[ha], %[cur]\n\t"                                                          \
    "v_add_u32 %[ha], 4, %[ha]\n\t" KSL
#define SIG_B_KSL(DIAG, K) "v_cndmask_b32 %[" K "], %[" K "], %[" DIAG "], vcc\n\t"
#define SIG_B_REV(DN, DDIAG, G, BT, K)                                                        \
    "v_mov_b32 %[" DN "], %[" BT "]\n\t"                                                      \
    "v_cmp_gt_u32 vcc, %[P], %[cnt]\n\t"                                                      \
    "v_add_u32 %[cnt], -1, %[cnt]\n\t"                                                        \
    "v_mov_b32_dpp %[" DN "], %[cur] wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"               \
    "v_add_f32 %[t], %[cur], %[" DN "]\n\t"                                                   \
    "v_mul_f32 %[y], %[r3], %[t]\n\t"                                                         \
    "v_accvgpr_read_b32 %[ge], %[" G "]\n\t"                                                  \
    "v_cndmask_b32 %[ge], 0, %[ge], vcc\n\t"                                                  \
    "v_add_f32 %[t], %[t], %[" DDIAG "]\n\t"                                                  \
    "v_fmac_f32 %[y], %[t], %[ge]\n\t"                                                        \
    "v_mul_f32 %[sv], %[" K "], %[" DDIAG "]\n\t"                                             \
    "v_fmac_f32 %[V], %[ge], %[y]\n\t"                                                        \
    "v_add_f32 %[cur], %[" DN "], %[V]\n\t"                                                   \
    "ds_write_b32 %[ha], %[cur]\n\t"                                                          \
    "v_add_u32 %[ha], -4, %[ha]\n\t"                                                          \
    "v_cndmask_b32 %[" K "], %[" K "], %[sv], vcc\n\t"

template <bool STORE>
__device__ __forceinline__ void band_fwd4(float &cur, float &upA, float &upB, float &V, const float *g, float *ksl,
                                          const float *bt, int &cnt, int &ha, const int P, const float r3)
{
    float ge, t, y;
    if (STORE)
        asm volatile(SIG_B_FWD("upA", "upB", "g0", "b0", SIG_B_KSL("upB", "k0"))
                     SIG_B_FWD("upB", "upA", "g1", "b1", SIG_B_KSL("upA", "k1"))
                     SIG_B_FWD("upA", "upB", "g2", "b2", SIG_B_KSL("upB", "k2"))
                     SIG_B_FWD("upB", "upA", "g3", "b3", SIG_B_KSL("upA", "k3"))
                     : [cur] "+v"(cur), [upA] "+v"(upA), [upB] "+v"(upB), [V] "+v"(V), [cnt] "+v"(cnt), [ha] "+v"(ha),
                       [ge] "=&v"(ge), [t] "=&v"(t), [y] "=&v"(y), [k0] "+v"(ksl[0]), [k1] "+v"(ksl[1]),
                       [k2] "+v"(ksl[2]), [k3] "+v"(ksl[3])
                     : [g0] "a"(g[0]), [g1] "a"(g[1]), [g2] "a"(g[2]), [g3] "a"(g[3]), [b0] "v"(bt[0]), [b1] "v"(bt[1]),
                       [b2] "v"(bt[2]), [b3] "v"(bt[3]), [P] "s"(P), [r3] "s"(r3)
                     : "vcc", "memory");
    else
        asm volatile(SIG_B_FWD("upA", "upB", "g0", "b0", "") SIG_B_FWD("upB", "upA", "g1", "b1", "")
                     SIG_B_FWD("upA", "upB", "g2", "b2", "") SIG_B_FWD("upB", "upA", "g3", "b3", "")
                     : [cur] "+v"(cur), [upA] "+v"(upA), [upB] "+v"(upB), [V] "+v"(V), [cnt] "+v"(cnt), [ha] "+v"(ha),
                       [ge] "=&v"(ge), [t] "=&v"(t), [y] "=&v"(y)
                     : [g0] "a"(g[0]), [g1] "a"(g[1]), [g2] "a"(g[2]), [g3] "a"(g[3]), [b0] "v"(bt[0]), [b1] "v"(bt[1]),
                       [b2] "v"(bt[2]), [b3] "v"(bt[3]), [P] "s"(P), [r3] "s"(r3)
                     : "vcc", "memory");
}
// steps k0+3 .. k0 (descending); bt[0] belongs to step k0+3
__device__ __forceinline__ void band_rev4(float &cur, float &dnA, float &dnB, float &V, const float *g, float *ksl,
                                          const float *bt, int &cnt, int &ha, const int P, const float r3)
{
    float ge, t, y, sv;
    asm volatile(SIG_B_REV("dnA", "dnB", "g3", "b0", "k3") SIG_B_REV("dnB", "dnA", "g2", "b1", "k2")
                 SIG_B_REV("dnA", "dnB", "g1", "b2", "k1") SIG_B_REV("dnB", "dnA", "g0", "b3", "k0")
                 : [cur] "+v"(cur), [dnA] "+v"(dnA), [dnB] "+v"(dnB), [V] "+v"(V), [cnt] "+v"(cnt), [ha] "+v"(ha),
                   [ge] "=&v"(ge), [t] "=&v"(t), [y] "=&v"(y), [sv] "=&v"(sv), [k0] "+v"(ksl[0]), [k1] "+v"(ksl[1]),
                   [k2] "+v"(ksl[2]), [k3] "+v"(ksl[3])
                 : [g0] "a"(g[0]), [g1] "a"(g[1]), [g2] "a"(g[2]), [g3] "a"(g[3]), [b0] "v"(bt[0]), [b1] "v"(bt[1]),
                   [b2] "v"(bt[2]), [b3] "v"(bt[3]), [P] "s"(P), [r3] "s"(r3)
                 : "vcc", "memory");
}
} // namespace

template <int DPAD, bool GRAD, bool SYM>
__global__ __launch_bounds__(BNW * 64, 1) void gram_band_kernel(BandArgs a)
{
    constexpr int NT = BNW * 64;
    constexpr int CS = DPAD + 1;  // row stride of the column-side image (odd: lanes on distinct banks)
    constexpr int YDS = DPAD + 2; // fp64 row: coordinates, [DPAD] = -log2(e)/h * |y~|^2
    constexpr int YFS = (DPAD == 4) ? 12 : DPAD + 4; // fp32 row: coordinates, [DPAD] = the same norm in fp32
    // rows are stored twice (r and r + 128): the skewed row (t - lane) & 127 is then base(lane) + t * stride
    __shared__ __align__(16) double yd[2 * RING * YDS];
    __shared__ __align__(16) float yf[GRAD ? 2 * RING * YFS : 4];
    __shared__ double yref[DPAD];
    __shared__ float colacc[(GRAD && SYM) ? 2 * RING * CS : 4];
    constexpr int HN = 2 * RING + 8;   // hand-over rows: entries 0 .. NSTEP + 65
    constexpr int HD = 2 * RING + 80;  // scratch row of the lanes that hand nothing over (one float per lane + the walk)
    __shared__ float ones[HN];
    __shared__ float hK_all[BNW * HN], hU_all[BNW * HN], hdummy_all[BNW * HD];
    __shared__ double g64_all[BNW * RING], rdh_all[BNW * RING];
    __shared__ float srow_all[GRAD ? BNW * 2 * RING : 4];
    __shared__ float x64_all[GRAD ? BNW * (DPAD + 2) : 4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int T = a.T, d = a.d, P = T - 1, io64 = a.io64;
    // grid decode as the kernel this replaces: ordered launches 2-D (column chunk, owned row tile); symmetric
    // launches 1-D over the chunks that reach the diagonal of their row tile
    int ty = blockIdx.y, cx = blockIdx.x;
    if (SYM) {
        const int nJ = (a.B + a.JC - 1) / a.JC;
        int rem = blockIdx.x;
        for (ty = 0;; ++ty) {
            const int first = ((a.tile_offset + ty * a.tile_stride) * BNW) / a.JC;
            const int cntc = nJ - first;
            if (rem < cntc) {
                cx = first + rem;
                break;
            }
            rem -= cntc;
        }
    }
    const int i0 = (a.tile_offset + ty * a.tile_stride) * BNW;
    const int i = i0 + wave;
    const int j0 = cx * a.JC, j1 = min(a.B, j0 + a.JC);
    const bool row_ok = i < a.A;
    const double inv_h = a.inv_h;
    const double nscale = -inv_h * 1.4426950408889634074;
    const float m2h = (float)(-2.0 * inv_h);
    float *hK = hK_all + wave * HN, *hU = hU_all + wave * HN, *hdummy = hdummy_all + wave * HD;
    double *g64 = g64_all + wave * RING, *rdh = rdh_all + wave * RING;
    float *srow63 = srow_all + (GRAD ? wave * 2 * RING : 0), *srow64 = srow63 + (GRAD ? RING : 0);
    float *x64 = x64_all + (GRAD ? wave * (DPAD + 2) : 0);
    for (int e = tid; e < HN; e += NT) ones[e] = 1.f;
    for (int e = lane; e < HN; e += 64) hK[e] = 1.f, hU[e] = 1.f; // (entries the sweeps do not reach stay at the boundary value)

    float gacc[2][DPAD]; // row-side gradient of (band, channel), summed over the column chunk in fp32
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int c = 0; c < DPAD; ++c) gacc[b][c] = 0.f;

    for (int j = j0; j < j1; ++j) {
        // ---- stage y_j (centred on its first point): fp64 rows + scaled norms, fp32 copy, both twice ----------
        __syncthreads();
        for (int e = tid; e < RING * DPAD; e += NT) {
            const int t = e / DPAD, c = e % DPAD;
            const bool ok = t < T && c < d;
            const double r0 = ok ? b_ldany(a.Y, (size_t)j * T * d + c, io64) : 0.0;
            const double v = ok ? b_ldany(a.Y, ((size_t)j * T + t) * d + c, io64) - r0 : 0.0;
            yd[t * YDS + c] = v;
            yd[(t + RING) * YDS + c] = v;
            if (GRAD) {
                yf[t * YFS + c] = (float)v;
                yf[(t + RING) * YFS + c] = (float)v;
            }
            if (t == 0) yref[c] = r0;
            double s = v * v * nscale;
#pragma unroll
            for (int off = 1; off < DPAD; off <<= 1) s += __shfl_xor(s, off, 64);
            if (c == 0) {
                yd[t * YDS + DPAD] = s;
                yd[(t + RING) * YDS + DPAD] = s;
                if (GRAD) {
                    yf[t * YFS + DPAD] = (float)s;
                    yf[(t + RING) * YFS + DPAD] = (float)s;
                }
            }
        }
        if (GRAD && SYM)
            for (int e = tid; e < 2 * RING * CS; e += NT) colacc[e] = 0.f;
        __syncthreads();

        if (row_ok && (!SYM || j >= i)) {
            float w_ij = 1.f, w_ji = 1.f; // row-side / column-side weights
            if (GRAD) {
                if (a.go) {
                    w_ij = (float)b_ldany(a.go, (size_t)i * a.B + j, io64);
                    if (SYM || a.symw) w_ji = (float)b_ldany(a.go, (size_t)j * a.B + i, io64);
                    if (a.symw) { w_ij += w_ji; w_ji = w_ij; }
                } else if (a.symw) {
                    w_ij = 2.f; w_ji = 2.f;
                }
                if (SYM && j == i) w_ji = 0.f; // diagonal pair: first-slot derivative only
            }

            // ---- G row 64 (the row beyond band 0): differences along the row for lane 63 of band 0 ------------
            {
                double xs2[DPAD], xn2 = 0.0;
#pragma unroll
                for (int c = 0; c < DPAD; ++c) {
                    const double xc = (c < d) ? b_ldany(a.X, ((size_t)i * T + 64) * d + c, io64) - yref[c] : 0.0;
                    xn2 = __builtin_fma(xc, xc, xn2);
                    xs2[c] = xc * (-2.0 * nscale);
                    if (GRAD && lane == 0) x64[c] = (float)xc;
                }
                xn2 = __builtin_fma(xn2, nscale, -1.79248125036057809); // G / sqrt(12), as in the band passes
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const double *yr = yd + (lane + 64 * h) * YDS;
                    double e2 = xn2 + yr[DPAD];
#pragma unroll
                    for (int c = 0; c < DPAD; ++c) e2 = __builtin_fma(xs2[c], yr[c], e2);
                    g64[lane + 64 * h] = exp2_p7(e2);
                }
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): the row is in LDS before it is read back
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int n = lane + 64 * h;
                    rdh[n] = g64[n] - g64[(n - 1) & (RING - 1)];
                }
            }

            float Dsl[RING]; // increments / sqrt(12) of the band in work
            float Ssl[RING]; // K_fwd, then S = K_fwd * U, of the band in work
            const int nrows1 = P - 64; // cell rows of band 1 (0 .. 63)

#pragma unroll 1
            for (int pass = 0; pass < 3; ++pass) {
                const int b = (pass == 1) ? 1 : 0;
                const bool full = GRAD && pass >= 1; // store K, run the reverse sweep and the gradient pass
                if (pass == 1 && nrows1 <= 0) continue;
                if (!GRAD && pass == 2) break;
                const int nrows = b ? nrows1 : 64;
                const int m = 64 * b + lane; // point row of this lane
                const int lane_q = (lane < nrows) ? lane : 0x40000000;

                // ---- x_m, centred and pre-scaled (fp64 for the static kernel, fp32 for the gradient pass) -------
                double xs[DPAD], xn = 0.0;
                float xf[DPAD];
#pragma unroll
                for (int c = 0; c < DPAD; ++c) {
                    const double xc = (m <= P && c < d) ? b_ldany(a.X, ((size_t)i * T + m) * d + c, io64) - yref[c] : 0.0;
                    xn = __builtin_fma(xc, xc, xn);
                    xs[c] = xc * (-2.0 * nscale);
                    xf[c] = (float)xc;
                }
                xn = __builtin_fma(xn, nscale, -1.79248125036057809); // - log2(sqrt(12)): D slots hold D / sqrt(12)

                // ---- phase 1: G row (skewed: column (t - lane) & 127 on iteration t) -> D slots --------------
                {
                    double g0 = 0.0, g1 = 0.0, gprev = 0.0, rdprev = 0.0;
                    const double *ybase = yd + (RING - lane) * YDS; // row (t - lane) & 127 == ybase + t * YDS
#pragma unroll
                    for (int t = 0; t < RING + 2; ++t) {
                        double g;
                        if (t < RING) {
                            const double *yr = ybase + t * YDS;
                            double e2 = xn + yr[DPAD];
#pragma unroll
                            for (int c = 0; c < DPAD; ++c) e2 = __builtin_fma(xs[c], yr[c], e2);
                            g = exp2_p7(e2);
                            if (t == 0) g0 = g;
                            if (t == 1) g1 = g;
                        } else {
                            g = (t == RING) ? g0 : g1;
                        }
                        const double rd = g - gprev; // G[m, c] - G[m, c-1]
                        gprev = g;
                        if (t >= 2) {
                            // lane l+1 holds the same column difference one iteration later; lane 63 of band 0 takes
                            // the row beyond the band (G row 64) from LDS
                            const double beyond = rdh[(t - 64) & (RING - 1)];
                            const double nb = b_shl_keep(rd, beyond);
                            Dsl[(t - 2) & (RING - 1)] = (float)(nb - rdprev);
                        }
                        rdprev = rd;
                        __builtin_amdgcn_sched_barrier(0); // one column per scheduling region: bounds live ranges
                    }
                }

                // ---- phase 2: forward sweep of the band ------------------------------------------------------
                float kfinal;
                {
                    const float *topb = b ? hK : ones;           // K[64 b][q + 1] for lane 0 on step sigma = q
                    float *hout = (b == 0) ? hK : hdummy;        // lane 63 of band 0 leaves K[64][.]
                    // LDS byte addresses of the hand-over targets (the low half of a flat LDS address is the LDS offset)
                    const unsigned hk_off = (unsigned)(size_t)hout;
                    const unsigned hd_off = (unsigned)(size_t)(hdummy + lane);
                    int haddr = (int)((lane == 63 ? hk_off : hd_off) + 4u * 2u); // entry sigma + 2 after step sigma
                    float cur = 1.f, upA = 1.f, upB = 1.f, V = 0.f;
                    float r3 = 1.7320508075688772f;
                    asm volatile("" : "+s"(r3));
                    int cnt = -lane_q;
                    if (full) {
#pragma unroll
                        for (int k = 0; k < RING; ++k) Ssl[k] = 0.f; // slots without a grid cell must read as S = 0
                    }
#pragma unroll
                    for (int s0 = 0; s0 < NSTEP; s0 += 4) {
                        float bt[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) bt[u] = topb[s0 + u + 65];
                        if (full)
                            band_fwd4<true>(cur, upA, upB, V, &Dsl[s0 & (RING - 1)], &Ssl[s0 & (RING - 1)], bt, cnt, haddr, P, r3);
                        else
                            band_fwd4<false>(cur, upA, upB, V, &Dsl[s0 & (RING - 1)], &Ssl[s0 & (RING - 1)], bt, cnt, haddr, P, r3);
                    }
                    kfinal = cur;
                }
                if (b == 0) {
                    // band 1 walks NSTEP steps as well and reads K[64][q + 1] up to q = NSTEP - 1: beyond the row's end
                    // the hand-over holds the row's last value, K[64][P] (a finished lane keeps its value only while
                    // its upper neighbour does)
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    const float fin = hK[NSTEP + 1];
                    for (int e = NSTEP + 2 + lane; e < HN; e += 64) hK[e] = fin;
                }
                // K[P][P]: last row of the last band with cells (band 1 if it has any, else band 0's first pass)
                {
                    const bool wr = (nrows1 > 0) ? (pass == 1) : (pass == 0);
                    const int lf = (nrows1 > 0) ? nrows1 - 1 : 63;
                    if (wr && lane == lf) {
                        b_stany(a.K, (size_t)i * a.B + j, (double)kfinal, io64);
                        if (SYM && j != i) b_stany(a.K, (size_t)j * a.B + i, (double)kfinal, io64);
                    }
                }
                if (!full || !GRAD) continue;

                // ---- phase 3: reverse sweep (S = K_fwd * U replaces K_fwd slot by slot) -------------------------
                {
                    const bool last_band = (b == 1) || (nrows1 <= 0);
                    const float *botb = last_band ? ones : hU;   // U[64 (b+1)][q] for lane 63 on step sigma = q + 63
                    float *hout = (b == 1) ? hU : hdummy;        // lane 0 of band 1 leaves U[64][.]
                    const unsigned ho_off = (unsigned)(size_t)hout;
                    const unsigned hd_off = (unsigned)(size_t)(hdummy + lane);
                    int haddr = (int)((lane == 0 ? ho_off : hd_off) + 4u * (unsigned)(NSTEP - 1 + 64)); // entry sigma + 64
                    float cur = 1.f, dnA = 1.f, dnB = 1.f, V = 0.f;
                    float r3 = 1.7320508075688772f;
                    asm volatile("" : "+s"(r3));
                    int cnt = NSTEP - 1 - lane_q;
#pragma unroll
                    for (int s0 = NSTEP - 4; s0 >= 0; s0 -= 4) {
                        float bt[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) bt[u] = botb[(s0 + 3 - u) + 1];
                        band_rev4(cur, dnA, dnB, V, &Dsl[s0 & (RING - 1)], &Ssl[s0 & (RING - 1)], bt, cnt, haddr, P, r3);
                    }
                }
                // ---- seam rows for the hand-over pass: S[63][.] (band 0, lane 63), S[64][.] (band 1, lane 0) ------
                if (lane == (b ? 0 : 63)) {
                    float *dst = b ? srow64 : srow63;
#pragma unroll
                    for (int k = 0; k < RING; ++k) dst[(k - lane) & (RING - 1)] = Ssl[k];
                }

                // ---- phase 4: 4-corner scatter R, static kernel in fp32, both contractions -----------------------
                // iteration it: lane l is at column n = (it - l) & 127 (the skew of phase 1); own row S[l][n] is slot it,
                // the upper row arrives through a wave shift one column ahead (lane l-1's slot it holds S[l-1][n+1]),
                // hence the two-deep history of the shifted values.
                {
                    const float rowmask = (b == 1 && lane == 0) ? 0.f : 1.f; // row 64 is contracted in the seam pass
                    float s0 = 0.f;
                    bf32x2 acc[DPAD / 2];
#pragma unroll
                    for (int c = 0; c < DPAD / 2; ++c) acc[c] = bf32x2{0.f, 0.f};
                    float xw[DPAD];
#pragma unroll
                    for (int c = 0; c < DPAD; ++c) xw[c] = xf[c] * w_ji;
                    const float ns32 = (float)nscale;
                    // history: at iteration 0 the previous columns are slots 127 (n-1) and 126 (n-2)
                    float Nc = b_shr_zero(Ssl[RING - 2]); // S[l-1][n-1]
                    float Nb = b_shr_zero(Ssl[RING - 1]); // S[l-1][n]
                    float Sprev = Ssl[RING - 1];          // S[l][n-1]
                    const float *yfb = yf + (RING - lane) * YFS;
                    float *cab = colacc + ((GRAD && SYM) ? (RING - lane) * CS : 0);
#pragma unroll
                    for (int it = 0; it < RING; ++it) {
                        const float Scur = Ssl[it];
                        const float Na = b_shr_zero(Scur); // S[l-1][n+1]
                        const float R = ((Nc - Nb) + (Scur - Sprev)) * rowmask;
                        Nc = Nb;
                        Nb = Na;
                        Sprev = Scur;
                        const float *yr = yfb + it * YFS;
                        const bf32x2 *yr2 = reinterpret_cast<const bf32x2 *>(yr);
                        // G[m][n] = 2^(-log2(e)/h * |x~_m - y~_n|^2) in fp32, from the DIFFERENCES: the expanded form
                        // |x|^2 + |y|^2 - 2<x,y> loses 6e-8 of its largest term, which for rough paths (|x~|^2 ~ 40) is
                        // 5e-6 of G -- measured 1.3e-5 on the gradient of a path against itself with K = 9e13
                        bf32x2 e2 = bf32x2{0.f, 0.f};
#pragma unroll
                        for (int c = 0; c < DPAD / 2; ++c) {
                            const bf32x2 df = bf32x2{xf[2 * c], xf[2 * c + 1]} - yr2[c];
                            e2 = __builtin_elementwise_fma(df, df, e2);
                        }
                        const float gv = __builtin_amdgcn_exp2f((e2[0] + e2[1]) * ns32);
                        const float rg = R * gv;
                        const bf32x2 rg2 = {rg, rg};
                        s0 += rg;
#pragma unroll
                        for (int c = 0; c < DPAD / 2; ++c) acc[c] = __builtin_elementwise_fma(rg2, yr2[c], acc[c]);
                        if (SYM) {
                            float *dst = cab + it * CS;
#pragma unroll
                            for (int c = 0; c < DPAD; ++c) atomicAdd(dst + c, rg * xw[c]);
                            atomicAdd(dst + DPAD, rg * w_ji);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int c = 0; c < DPAD; ++c) gacc[b][c] += w_ij * m2h * (xf[c] * s0 - acc[c / 2][c % 2]);
                }
            } // band passes

            if (GRAD) {
                // ---- seam: point row 64.  R[64][n] = (S[63][n-1] - S[63][n]) - (S[64][n-1] - S[64][n]), formed from both
                // bands' rows before the contraction; lanes take columns n = lane and lane + 64.
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_s_waitcnt(0xc07f);
                float ps0 = 0.f, part[DPAD], xm[DPAD];
#pragma unroll
                for (int c = 0; c < DPAD; ++c) {
                    part[c] = 0.f;
                    xm[c] = x64[c];
                }
                const float ns32 = (float)nscale;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int n = lane + 64 * h;
                    const float Sa = srow63[(n - 1) & (RING - 1)], Sz = srow63[n];
                    const float Ta = (nrows1 > 0) ? srow64[(n - 1) & (RING - 1)] : 0.f;
                    const float Tz = (nrows1 > 0) ? srow64[n] : 0.f;
                    const float *yr = yf + n * YFS;
                    float e2 = 0.f;
#pragma unroll
                    for (int c = 0; c < DPAD; ++c) e2 = __builtin_fmaf(xm[c] - yr[c], xm[c] - yr[c], e2);
                    const float rgn = (n <= P) ? ((Sa - Sz) - (Ta - Tz)) * __builtin_amdgcn_exp2f(e2 * ns32) : 0.f;
                    ps0 += rgn;
#pragma unroll
                    for (int c = 0; c < DPAD; ++c) part[c] = __builtin_fmaf(rgn, yr[c], part[c]);
                    if (SYM) {
                        float *dst = colacc + n * CS;
#pragma unroll
                        for (int c = 0; c < DPAD; ++c) atomicAdd(dst + c, rgn * w_ji * xm[c]);
                        atomicAdd(dst + DPAD, rgn * w_ji);
                    }
                }
                // row 64 belongs to band 1's lane 0 accumulators
#pragma unroll
                for (int c = 0; c < DPAD; ++c) {
                    float v = w_ij * m2h * (xm[c] * ps0 - part[c]);
#pragma unroll
                    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
                    if (lane == 0) gacc[1][c] += v;
                }
            }
        } // this wavefront's pair

        if (GRAD && SYM) {
            // close the column-side sums of y_j over the four rows of the tile:
            // d/dy_n = -(2/h) * (y~_n * sum_m w R G - sum_m w R G x~_m); the image is stored twice (rows n, n + 128)
            __syncthreads();
            for (int e = tid; e < T * DPAD; e += NT) {
                const int n = e / DPAD, c = e % DPAD;
                const float sw = colacc[n * CS + DPAD] + colacc[(n + RING) * CS + DPAD];
                const float sx = colacc[n * CS + c] + colacc[(n + RING) * CS + c];
                const float v = m2h * (yf[n * YFS + c] * sw - sx);
                if (c < d && v != 0.f) unsafeAtomicAdd(&a.gacc[((size_t)j * T + n) * d + c], (double)v);
            }
        }
    }

    if (GRAD && row_ok) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int p = kb * 64 + lane;
            if (p < T)
#pragma unroll
                for (int c = 0; c < DPAD; ++c)
                    if (c < d) unsafeAtomicAdd(&a.gacc[((size_t)i * T + p) * d + c], (double)gacc[kb][c]);
        }
    }
}

template <typename IO>
__global__ void band_finalize_kernel(const double *gacc, IO *gradX, size_t n)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n) gradX[idx] = (IO)gacc[idx];
}

bool band_supported(int A, int B, int T, int d, int n, int kind, unsigned flags)
{
    (void)A; (void)B;
    if (n != 0 || T < 65 || T > RING || d > 16) return false;
    if (kind != SIGSVGD_STATIC_RBF) return false;
    if (flags & SIGSVGD_FLAG_NAIVE_SOLVER) return false;
    return true;
}

int band_workspace_bytes(int A, int T, int d, int want_grad, size_t *bytes)
{
    *bytes = want_grad ? (size_t)A * T * d * sizeof(double) + 256 : 0;
    return SIGSVGD_OK;
}

namespace {
template <int DPAD>
int band_launch_variant(const GramProblem &p, BandArgs &a, bool grad, bool sym)
{
    const int ntile = (p.A + BNW - 1) / BNW;
    const int owned = (ntile - a.tile_offset + a.tile_stride - 1) / a.tile_stride;
    if (owned <= 0) return SIGSVGD_OK;
    int JC = 8;
    while (JC > 1 && (long long)owned * ((p.B + JC - 1) / JC) < (sym ? 2048 : 1024)) JC >>= 1;
    a.JC = JC;
    dim3 grid((p.B + JC - 1) / JC, owned), block(BNW * 64);
    if (sym) { // count the chunks on or right of the diagonal of every owned tile
        const int nJ = (p.B + JC - 1) / JC;
        long long total = 0;
        for (int k = 0; k < owned; ++k) {
            const int first = ((a.tile_offset + k * a.tile_stride) * BNW) / JC;
            if (first < nJ) total += nJ - first;
        }
        if (total <= 0) return SIGSVGD_OK;
        grid = dim3((unsigned)total, 1);
    }
    if (grad && sym)
        hipLaunchKernelGGL((gram_band_kernel<DPAD, true, true>), grid, block, 0, p.stream, a);
    else if (grad)
        hipLaunchKernelGGL((gram_band_kernel<DPAD, true, false>), grid, block, 0, p.stream, a);
    else if (sym)
        hipLaunchKernelGGL((gram_band_kernel<DPAD, false, true>), grid, block, 0, p.stream, a);
    else
        hipLaunchKernelGGL((gram_band_kernel<DPAD, false, false>), grid, block, 0, p.stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "launch gram_band_kernel");
    return SIGSVGD_OK;
}

int band_dispatch(const GramProblem &p, BandArgs &a, bool grad, bool sym)
{
    if (p.d <= 8) return band_launch_variant<8>(p, a, grad, sym);
    return band_launch_variant<16>(p, a, grad, sym);
}

void band_fill_args(const GramProblem &p, BandArgs &a)
{
    a.X = p.X; a.Y = p.Y; a.go = p.grad_out; a.K = p.K_out; a.gacc = nullptr;
    a.io64 = p.dtype == SIGSVGD_F64; a.A = p.A; a.B = p.B; a.T = p.T; a.d = p.d; a.JC = 1;
    a.symw = (p.flags & SIGSVGD_FLAG_SYM) ? 1 : 0; a.inv_h = p.inv_h;
    a.tile_offset = 0; a.tile_stride = 1;
}
} // namespace

int band_launch(const GramProblem &p)
{
    const bool grad = p.gradX_out != nullptr;
    const bool sym = (p.flags & SIGSVGD_FLAG_Y_IS_X) && p.A == p.B; // Y is X: each unordered pair once
    BandArgs a;
    band_fill_args(p, a);
    if (a.symw && p.A != p.B) {
        set_error("sym backward needs A == B");
        return SIGSVGD_E_BADARG;
    }
    const size_t nacc = (size_t)p.A * p.T * p.d;
    if (grad) {
        const size_t need = nacc * sizeof(double) + 256;
        if (!p.ws || p.ws_bytes < need) {
            set_error("band: workspace %zu B < required %zu B", p.ws_bytes, need);
            return SIGSVGD_E_WORKSPACE;
        }
        a.gacc = reinterpret_cast<double *>((reinterpret_cast<uintptr_t>(p.ws) + 255) & ~(uintptr_t)255);
        hipError_t e = hipMemsetAsync(a.gacc, 0, nacc * sizeof(double), p.stream);
        if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(gacc)");
    }
    int rc = band_dispatch(p, a, grad, sym);
    if (rc) return rc;
    if (grad) {
        const int bs = 256;
        const unsigned gs = (unsigned)((nacc + bs - 1) / bs);
        if (p.dtype == SIGSVGD_F64)
            hipLaunchKernelGGL(band_finalize_kernel<double>, dim3(gs), dim3(bs), 0, p.stream, a.gacc,
                               static_cast<double *>(p.gradX_out), nacc);
        else
            hipLaunchKernelGGL(band_finalize_kernel<float>, dim3(gs), dim3(bs), 0, p.stream, a.gacc,
                               static_cast<float *>(p.gradX_out), nacc);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "launch band_finalize_kernel");
    }
    return SIGSVGD_OK;
}

// Sharded partial solve (sigsvgd_gram_sym_partial) for the long-path shapes: row tiles of 4 rows,
// tiles tile_offset + k * tile_stride, both orientations of K stored into the caller-zeroed K_partial,
// gradient shares accumulated (fp64 atomics) straight into the caller-zeroed grad_partial.
int band_sym_partial(const GramProblem &p, int tile_offset, int tile_stride, double *grad_partial)
{
    if (tile_stride < 1 || tile_offset < 0 || tile_offset >= tile_stride) {
        set_error("bad tile_offset/tile_stride %d/%d", tile_offset, tile_stride);
        return SIGSVGD_E_BADARG;
    }
    BandArgs a;
    band_fill_args(p, a);
    a.gacc = grad_partial;
    a.tile_offset = tile_offset;
    a.tile_stride = tile_stride;
    return band_dispatch(p, a, true, true);
}

} // namespace sigsvgd
