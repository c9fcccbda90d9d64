// Signature-kernel Gram forward/backward for short paths whose REFINED grid has 129 .. 256 cells per side -- the
// reference's remaining call shapes: examples/script_sequential_distribution.ipynb (10 points, dyadic order 4: 144 cells)
// and examples/script_control_particle_maze.py:43-44 (30 points, order 3: 232 cells).  Until round 3 they ran on the
// coverage kernel (gram_generic.hip: fp64 sweeps, 53 instructions per step of which 8 are the stencil).
//
// Same frame as gram_dyad.hip -- one wavefront per trajectory pair, eight per workgroup sharing the staged column
// trajectory, everything around the sweeps on the COARSE grid (static kernel T x T in fp64, increment table
// D_coarse / (r^2 sqrt(12)) in LDS, block sums of S = K_fwd * U in fp64 LDS, 4-corner scatter and both contractions once
// per pair in fp32 on the differences x_m - y_n), gradient partial sums through the segment / item slabs of
// grad_reduce_kernel (no atomics between wavefronts, bit-reproducible) -- but the sweeps run over BANDS of 64 cell rows
// and all P columns, like the coverage kernel's: a quadrant decomposition of a 144-cell grid spends 60 % of its lane-steps
// outside the grid and needs the forward solution three times.  Per step: the lane's increment is read from the table
// (one LDS read, fetched a step ahead), the stencil is the fp32 difference form of gram_fast.hip (V = K[p+1][q] - K[p][q]
// carried along the row, one full-magnitude add per cell that never feeds back), lane 0 takes the band's upper boundary
// value through v_readlane and a select from a register refilled every 64 steps, lane 63 hands its row over through a
// lane-selected LDS address, and the forward solution goes to a per-wavefront scratch in [band][step][lane] order
// (coalesced 256-B rows, read back by the same wavefront through an eight-deep register ring; measured on the coverage
// kernel: that round trip is not what limits the step).
//
// Reference semantics: sigkernel _SigKernelGram.forward/backward [RECALLED, SURVEY.md App. A]; static kernel
// src/kernels/_traj_kernels.py:176-195; callers src/inference/score.py:68-69.
#include "sig_common.h"

namespace sigsvgd {

struct BandArgs {
    const void *X, *Y, *go;
    void *K;
    double *rseg; // [owned tiles + workgroups][8][T*d]
    float *cslab; // [items][T*d] (symmetric launches)
    float *wsk;   // [gridDim.x][8 waves][bands][steps][64]: forward solution of the pair in work (gradient launches)
    size_t wsk_per_wave;
    unsigned char *kflag; // [A][B]: 1 where the fp32 solution of the pair cancelled (max |K_grid| > r max(|K|, 0.1), r = 2 / 4 / 8) or is ill-conditioned (d <= 3), as in
                          // gram_quad.hip): the launcher lets the coverage kernel solve those pairs' K again in fp64
    int io64, A, B, T, d, n, symw;
    TileMap tm;
    long long nitems;
    double inv_h;
};

namespace {
// wavefronts (rows i) per workgroup: 8 (two per SIMD) when there are pairs to fill the chip, 4 (one per SIMD: a wavefront's
// dependent chain has the SIMD to itself) for launches of at most 4 pairs per CU, as in gram_dyad.hip
constexpr int BTMAX = 33;  // coarse points per path
constexpr int BPMAX = 256; // refined cells per side
constexpr int BPAD = 64;   // boundary rows: entry e lives at [BPAD + e]; lanes outside the grid write into the padding
constexpr int BHN = BPMAX + 2 + 2 * BPAD;

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
// lane l <- lane l-1 (lane 0 keeps its own) / lane l <- lane l+1 (lane 63 keeps its own)
__device__ __forceinline__ float b_shr(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x138, 0xF, 0xF, false));
}
__device__ __forceinline__ float b_shl(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x130, 0xF, 0xF, false));
}
// The forward neighbour shift of a sweep step: lane 0 of `dst` takes the value lane `src` of `from` holds (v_readlane ->
// v_writelane), every other lane its upper neighbour's `v` (DPP move; lane 0 has no source and keeps what v_writelane put
// there).  One asm statement because the hazards between its instructions are not hipcc's to pad: a VALU-written SGPR
// wants wait states before the next VALU instruction reads it (s_nop), and the three instructions in front of the DPP move
// are also the wait states between the stencil's write of `v` and the DPP read.
__device__ __forceinline__ void b_shr_take(float &dst, float v, float from, int src)
{
    int sb;
    asm("v_readlane_b32 %1, %2, %3\n\ts_nop 3\n\tv_writelane_b32 %0, %1, 0\n\t"
        "v_mov_b32_dpp %0, %4 wave_shr:1 row_mask:0xf bank_mask:0xf"
        : "+v"(dst), "=&s"(sb)
        : "v"(from), "s"(src), "v"(v));
}
// (the reverse shift's boundary lane is the band's last row, known at run time only: a second scalar operand is one too
//  many for v_writelane, so that lane takes its value through a select on a mask that does not change over the sweep.
//  `shifted` is an argument on purpose: written as `here ? sb : b_shl(v)` the DPP move is evaluated where `here` is false
//  only, i.e. with the boundary lane switched off in EXEC -- and a DPP move does not write a lane whose SOURCE lane is
//  switched off: the lane next to the boundary kept a stale value and every gradient was wrong by O(1))
__device__ __forceinline__ float b_shl_take(float shifted, float from, int src, bool here)
{
    const float sb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(from), src));
    return here ? sb : shifted;
}
} // namespace

// COMP: the forward sweep's full-magnitude add in two floats (see `clo` below): dyadic order >= 5, where the refined
// increments are so uniform that its rounding drifts (order 6, 256 cells, smooth paths: 1.2e-5 without, 1e-7 with); at the
// reference's orders 3 and 4 the plain add stays inside 5e-6 and the six instructions per step (+8 % / +19 % forward-only)
// are left out.
template <int DPAD, bool GRAD, bool SYM, bool COMP, int BNW>
__global__ __launch_bounds__(BNW * 64) __attribute__((amdgpu_waves_per_eu(BNW == 8 ? 2 : 1, 2))) void gram_band_kernel(BandArgs a)
{
    constexpr int NT = BNW * 64;
    constexpr int TM = BTMAX - 1; // coarse cells per side at most
    __shared__ __align__(16) double yd[BTMAX * (DPAD + 1)]; // y~_n in fp64 (centred on y[0]), [DPAD] = -|y~_n|^2 / h
    __shared__ __align__(16) float yf[BTMAX * DPAD];        // the same in fp32 for the coarse contraction
    __shared__ double yref[DPAD];
    struct WaveLds {
        double Sc[TM * TM];    // block sums of S = K_fwd * U over the fine cells of every coarse cell
        float Dc[TM * TM];     // coarse increments / (r^2 sqrt(12)); after the sweeps: the parked column-side sums
        float hK[BHN], hU[BHN]; // K[64 b][.] left by band b - 1 for band b; U[64 b][.] left by band b for band b - 1
        float dump[64];
        double dumpd[64];           // where the lanes without a finished run add their zero (reverse sweep)
        float rowacc[BTMAX * DPAD]; // row-side gradient of the wavefront's particle over the columns of a segment
    };
    __shared__ WaveLds wl_all[BNW];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = a.T, d = a.d, n = a.n, io64 = a.io64, Tm = T - 1, r = 1 << n;
    const int P = Tm << n, nb = (P + 63) >> 6, nsteps = P + 63;
    const double inv_h = a.inv_h;
    const float m2h = (float)(-2.0 * inv_h);
    const double dscale = 1.0 / ((double)r * (double)r * 3.46410161513775459); // 1 / (r^2 sqrt(12))
    const double inv_r2 = 1.0 / ((double)r * (double)r);
    WaveLds &wl = wl_all[wave];
    float *wsw = GRAD ? a.wsk + ((size_t)blockIdx.x * BNW + wave) * a.wsk_per_wave + 16 * 64 : nullptr; // (16 rows of padding in front)

    // static item ranges: (owned row tile, column), tile-major; symmetric launches only the columns from the tile's first row
    const long long it0 = a.nitems * blockIdx.x / gridDim.x, it1 = a.nitems * (blockIdx.x + 1) / gridDim.x;
    int remaining = (int)(it1 - it0);
    long long item = it0;
    int kq = 0, cstart = 0;
    {
        long long rem = it0;
        for (;; ++kq) {
            const int cn = a.B - (SYM ? a.tm.tile_of(kq) * BNW : 0);
            if (rem < cn) break;
            rem -= cn;
        }
        cstart = (int)rem;
    }
#pragma unroll 1
    while (remaining > 0) {
    const int itile = a.tm.tile_of(kq);
    const int cfirst = SYM ? itile * BNW : 0;
    const int ncolr = min(a.B - cfirst - cstart, remaining);
    const int i0 = itile * BNW, i = i0 + wave;
    const int j0 = cfirst + cstart, j1 = j0 + ncolr;
    const bool row_ok = i < a.A;
    if (GRAD)
        for (int e = lane; e < BTMAX * DPAD; e += 64) wl.rowacc[e] = 0.f;

#pragma unroll 1
    for (int j = j0; j < j1; ++j, ++item) {
        int lanep = lane;
        asm volatile("" : "+v"(lanep));
        // ---- stage y_j (coarse points, centred on its first point) ------------------------------------------------
        __syncthreads();
        for (int e = tid; e < T * DPAD; e += NT) {
            const int t = e / DPAD, c = e % DPAD;
            const double r0 = c < d ? b_ldany(a.Y, (size_t)j * T * d + c, io64) : 0.0;
            const double v = c < d ? b_ldany(a.Y, ((size_t)j * T + t) * d + c, io64) - r0 : 0.0;
            yd[t * (DPAD + 1) + c] = v;
            yf[t * DPAD + c] = (float)v;
            if (t == 0) yref[c] = r0;
            double s = v * v;
#pragma unroll
            for (int off = 1; off < DPAD; off <<= 1) s += __shfl_xor(s, off, 64);
            if (c == 0) yd[t * (DPAD + 1) + DPAD] = -s * inv_h;
        }
        __syncthreads();

        if (row_ok && (!SYM || j >= i)) {
            float w_ij = 1.f, w_ji = 1.f;
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
            // ---- coarse static kernel: lane m = point row m; G[m][b] in fp64, row differences, 4-corner increments --------
            float xf[DPAD]; // x~_m in fp32 for the coarse contraction
            {
                const int m = min(lanep, T - 1);
                double xs[DPAD], xn = 0.0;
#pragma unroll
                for (int c = 0; c < DPAD; ++c) {
                    const double xc = c < d ? b_ldany(a.X, ((size_t)i * T + m) * d + c, io64) - yref[c] : 0.0;
                    xn = __builtin_fma(xc, xc, xn);
                    xs[c] = xc * (2.0 * inv_h);
                    xf[c] = (float)xc;
                }
                xn = -xn * inv_h;
                double gprev = 0.0;
                for (int b = 0; b < T; ++b) {
                    const double *yr = yd + b * (DPAD + 1);
                    double e2 = xn + yr[DPAD];
#pragma unroll
                    for (int c = 0; c < DPAD; ++c) e2 = __builtin_fma(xs[c], yr[c], e2);
                    const double g = exp64(e2);
                    const double rd = g - gprev; // G[m][b] - G[m][b-1]
                    gprev = g;
                    const double nbr = shfl_down_f64(rd); // row m + 1
                    if (b >= 1 && lanep < Tm) wl.Dc[lanep * Tm + (b - 1)] = (float)((nbr - rd) * dscale);
                }
                if (GRAD)
                    for (int e = lanep; e < Tm * Tm; e += 64) wl.Sc[e] = 0.0;
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xc07f);

            // ---- forward sweep, band by band -------------------------------------------------------------------------
            double kfin = 1.0;
            float kmax = 1.f; // largest |K| this lane has seen on the pair's grid
            float kfin_keep = 0.f;
            bool canc_keep = false;
#pragma unroll 1
            for (int kb = 0; kb < nb; ++kb) {
                const int p = 64 * kb + lanep;
                const bool rowvalid = p < P;
                const float *dcrow = wl.Dc + (min(p, P - 1) >> n) * Tm;
                float *wb = GRAD ? wsw + (size_t)kb * nsteps * 64 : nullptr; // (uniform: the store takes it as a scalar base)
                float cur = 1.f, upprev = 1.f, V = 0.f, hbv = 1.f, up = 1.f;
                // K[p+1][q] = cur + clo: the one full-magnitude add of a step, K11 = K01 + V, is made in two floats.  On a
                // refined grid the increments of neighbouring cells are nearly identical, so its rounding has the same sign row
                // after row and K drifts by up to 6e-8 per ROW (1.2e-5 at 256 cells per side with smooth paths, K ~ 1);
                // the rounding error of each add travels down the rows as the low word (one more DPP move, three adds and a
                // select per step) and the drift is gone.  It feeds nothing else: the stencil takes the high words.
                float clo = 0.f;
                int q1 = 1 - lanep; // column + 1 of the cell in work
                // lane 63 hands K[64 kb + 64][q + 1] over through entry q + 1 of hK (entries below 1 and beyond P are padding:
                // a lane outside the grid writes whatever it computed there); the other lanes store into their own dump cell
                float *ho = (lanep == 63) ? wl.hK + BPAD + q1 : wl.dump + lanep;
                const int hinc = (lanep == 63) ? 1 : 0;
                const unsigned qlim = rowvalid ? (unsigned)P : 0u; // (no row: never inside the grid)
                float g = dcrow[(q1 - 1) >> n]; // (q < 0: a harmless read below the row; the lane is outside the grid)
#pragma unroll 1
                for (int s0 = 0; s0 < nsteps; s0 += 64) {
                    // lane 0's upper neighbour on step s is entry s + 1 of the row band kb - 1 left: 64 entries per refill
                    hbv = kb ? wl.hK[BPAD + s0 + lanep + 1] : 1.f;
                    const int send = __builtin_amdgcn_readfirstlane(min(64, nsteps - s0)); // (a scalar loop bound)
#pragma unroll 1
                    for (int u = 0; u < send; ++u) {
                        const bool active = (unsigned)(q1 - 1) < qlim;
                        const float gnx = dcrow[q1 >> n];
                        b_shr_take(up, cur, hbv, u);
                        // K11 - K01 = (K10 - K00) + F,  F = gamma (sqrt(3) t + gamma (t + K00)),  t = K10 + K01
                        const float t = cur + up;
                        float y = 1.7320508075688772f * t;
                        y = __builtin_fmaf(t + upprev, g, y);
                        const float Vn = __builtin_fmaf(g, y, V);
                        // (lane 0: the row the band above handed over carries its low word already -- 0 from the shift)
                        float Vt = Vn, nlo = 0.f;
                        if constexpr (COMP)
                            Vt += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(clo), 0x138, 0xF, 0xF, true));
                        const float nw = up + Vt;
                        if constexpr (COMP) nlo = Vt - (nw - up); // (exact while |K01| >= |V|; otherwise merely no better than before)
                        if (GRAD) { // K[p][q]: only the entries of grid cells are read back.  (asm: a scalar row base + the lane's
                                    //  constant offset instead of a 64-bit vector pointer bumped every step; the reverse sweep waits
                                    //  for these stores with s_waitcnt vmcnt(0) and a compiler barrier)
                            const float *rowp = wb + (size_t)(s0 + u) * 64;
                            asm volatile("global_store_dword %0, %1, %2" ::"v"(lanep * 4), "v"(upprev), "s"(rowp));
                        }
                        *ho = COMP ? nw + nlo : nw;
                        ho += hinc;
                        cur = active ? nw : cur;
                        asm("v_max_f32 %0, |%1|, %0" : "+v"(kmax) : "v"(cur)); // (fmaxf costs two canonicalising moves more)
                        if constexpr (COMP) clo = active ? nlo : clo;
                        V = active ? Vn : V;
                        upprev = active ? up : upprev;
                        g = gnx;
                        ++q1;
                    }
                }
                if (p == P - 1) kfin = (double)cur + (double)clo;
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_s_waitcnt(0xc07f); // the row is in LDS before the next band reads it
            }
            {
                const float kfv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int((float)kfin), (P - 1) & 63));
                // (round 4: no floor "grid maximum > 2" any more -- on a refined grid the full-magnitude add drifts by up to 6e-8 per
                //  row of the largest value, the boundary value 1 included, so a pair that merely decays to K = 0.15 is as exposed as
                //  one that oscillates (gram_dyad.hip, soak case 504); without the two-float add the ratio is at most 4)
                const bool cancelled = __builtin_amdgcn_ballot_w64(kfv == kfv && kmax > (d == 1 ? 2.f : (d == 2 || !COMP) ? 4.f : 8.f) * fmaxf(fabsf(kfv), 0.1f)) != 0;
                if (lanep == ((P - 1) & 63)) {
                    b_stany(a.K, (size_t)i * a.B + j, kfin, io64);
                    if (SYM && j != i) b_stany(a.K, (size_t)j * a.B + i, kfin, io64);
                    if (!GRAD) a.kflag[(size_t)i * a.B + j] = cancelled ? 1 : 0; // (gradient launches: with the condition number, below)
                }
                kfin_keep = kfv;
                canc_keep = cancelled;
            }

            if (GRAD) {
                // ---- reverse sweep: U towards smaller rows and columns; S = K_fwd[p][q] * U[p+1][q+1] block-summed ------------
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the forward solution has left the wavefront
#pragma unroll 1
                for (int kb = nb - 1; kb >= 0; --kb) {
                    const int p = 64 * kb + lanep;
                    const bool rowvalid = p < P;
                    const int L = min(64, P - 64 * kb);
                    const int arow = min(p, P - 1) >> n;
                    const float *dcrow = wl.Dc + arow * Tm;
                    double *scrow = wl.Sc + arow * Tm;
                    const bool lastband = kb == nb - 1;
                    float cur = 1.f, dprev = 1.f, V = 0.f, run = 0.f, hbv = 1.f;
                    const int nsp = P + L - 1;
                    int q = P - 1 + (L - 1 - lanep);
                    // K_fwd[p][q] was stored on forward step lane + q: row R = P + L - 2 - sp of the band's scratch on step sp
                    const float *wrow = wsw + (size_t)kb * nsteps * 64 + lanep;
                    int R = P + L - 2;
                    // (uniform row pointer of the ring's next load; the last groups of steps reach up to 15 rows below the band's
                    //  first: the 16 rows of padding in front of a wavefront's scratch, or the band below -- read, never used)
                    const float *rnext = wsw + ((size_t)kb * nsteps + (R - 8)) * 64;
                    // lane 0 hands U[64 kb][q] over through entry q of hU
                    float *ho = (lanep == 0) ? wl.hU + BPAD + q : wl.dump + lanep;
                    const int hinc = (lanep == 0) ? -1 : 0;
                    float g = dcrow[min(q, P - 1) >> n];
                    constexpr int KPF = 8;
                    float kfr[KPF];
#pragma unroll
                    for (int u = 0; u < KPF; ++u) kfr[u] = wrow[(size_t)max(R - u, 0) * 64];
#pragma unroll 1
                    for (int sp0 = 0; sp0 < nsp; sp0 += KPF) {
#pragma unroll
                        for (int u = 0; u < KPF; ++u, --q, --R) {
                            const int sp = sp0 + u;
                            // lane L-1's lower neighbour on step sp is entry P - 1 - sp of the row band kb + 1 left
                            if ((sp & 63) == 0) hbv = lastband ? 1.f : wl.hU[BPAD + max(P - 1 - sp - lanep, -BPAD)];
                            const bool active = rowvalid && (unsigned)q < (unsigned)P;
                            const float gnx = dcrow[(q - 1) >> n]; // (q < 1: a harmless read below the row)
                            const float kf = kfr[u];
                            kfr[u] = rnext[lanep];
                            rnext -= 64;
                            const float down = b_shl_take(b_shl(cur), hbv, sp & 63, lanep == L - 1);
                            // block sums without a branch: every lane adds every step -- its finished run to the coarse cell
                            // when it has just taken the cell's leftmost fine column, a zero to its own dump cell otherwise
                            // (two nested EXEC regions per step cost the unrolled loop more than the LDS add)
                            run = __builtin_fmaf(active ? kf : 0.f, dprev, run);
                            const bool fl = active && (q & (r - 1)) == 0;
                            float addend = fl ? run : 0.f;
                            asm volatile("" : "+v"(addend)); // (select, then convert: hipcc otherwise converts and selects both halves)
                            unsafeAtomicAdd(fl ? scrow + (q >> n) : wl.dumpd + lanep, (double)addend); // ds_add_f64
                            run = fl ? 0.f : run;
                            const float t = cur + down;
                            float y = 1.7320508075688772f * t;
                            y = __builtin_fmaf(t + dprev, g, y);
                            const float Vn = __builtin_fmaf(g, y, V);
                            const float nw = down + Vn;
                            *ho = nw; // (a lane outside the grid writes into the padding, or entries nobody reads)
                            ho += hinc;
                            cur = active ? nw : cur;
                            V = active ? Vn : V;
                            dprev = active ? down : dprev;
                            g = gnx;
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                }

                // the pair's verdict for the exact fp64 pass: the grid maximum (above) or, in <= 3 channels, the condition number of
                // K in the stored coarse increments, c1 = sqrt(12) sum |Sc * Dc| / max(|K|, 0.1) > 150 (gram_dyad.hip, gram_fast.hip)
                {
                    bool ill = false;
                    if (d <= 3) {
                        float cs = 0.f;
                        for (int e = lanep; e < Tm * Tm; e += 64) cs = __builtin_fmaf(fabsf((float)wl.Sc[e]), fabsf(wl.Dc[e]), cs);
#pragma unroll
                        for (int off = 1; off < 64; off <<= 1) cs += __shfl_xor(cs, off, 64);
                        ill = kfin_keep == kfin_keep && cs * 3.46410161513775459f > 150.f * fmaxf(fabsf(kfin_keep), 0.1f);
                    }
                    if (lanep == 0) a.kflag[(size_t)i * a.B + j] = (canc_keep || ill) ? 1 : 0;
                }
                // ---- coarse gradient: R = 4-corner scatter of S_coarse / r^2, RBF derivative, both contractions ----------
                const float ns32 = (float)(-inv_h * 1.4426950408889634074);
                auto Sat = [&](int aa, int bb) -> float {
                    return (aa >= 0 && aa < Tm && bb >= 0 && bb < Tm) ? (float)(wl.Sc[aa * Tm + bb] * inv_r2) : 0.f;
                };
                // row side: lane m sums over the columns n
                if (lanep < T) {
                    const int m = lanep;
                    float acc[DPAD];
#pragma unroll
                    for (int c = 0; c < DPAD; ++c) acc[c] = 0.f;
                    for (int nn = 0; nn < T; ++nn) {
                        const float Rv = (Sat(m - 1, nn - 1) + Sat(m, nn)) - (Sat(m - 1, nn) + Sat(m, nn - 1));
                        const float *yr = yf + nn * DPAD;
                        float df[DPAD], e2 = 0.f;
#pragma unroll
                        for (int c = 0; c < DPAD; ++c) {
                            df[c] = xf[c] - yr[c];
                            e2 = __builtin_fmaf(df[c], df[c], e2);
                        }
                        const float rg = Rv * __builtin_amdgcn_exp2f(e2 * ns32);
#pragma unroll
                        for (int c = 0; c < DPAD; ++c) acc[c] = __builtin_fmaf(rg, df[c], acc[c]);
                    }
#pragma unroll
                    for (int c = 0; c < DPAD; ++c)
                        if (c < d) wl.rowacc[m * DPAD + c] += w_ij * m2h * acc[c];
                }
                // column side (Y is X): lane n sums over the rows m; x~_m comes from the lanes through LDS (the parked area)
                if (SYM) {
                    float *xl = wl.Dc; // (the increments are not needed any more) x~ rows [T][DPAD], then the parked sums
                    if (lanep < T) {
#pragma unroll
                        for (int c = 0; c < DPAD; ++c) xl[lanep * DPAD + c] = xf[c];
                    }
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    float acc[DPAD];
#pragma unroll
                    for (int c = 0; c < DPAD; ++c) acc[c] = 0.f;
                    const int nn = min(lanep, T - 1);
                    const float *yr = yf + nn * DPAD;
                    for (int m = 0; m < T; ++m) {
                        const float Rv = (Sat(m - 1, nn - 1) + Sat(m, nn)) - (Sat(m - 1, nn) + Sat(m, nn - 1));
                        const float *xr = xl + m * DPAD;
                        float df[DPAD], e2 = 0.f;
#pragma unroll
                        for (int c = 0; c < DPAD; ++c) {
                            df[c] = xr[c] - yr[c];
                            e2 = __builtin_fmaf(df[c], df[c], e2);
                        }
                        const float rg = Rv * __builtin_amdgcn_exp2f(e2 * ns32);
#pragma unroll
                        for (int c = 0; c < DPAD; ++c) acc[c] = __builtin_fmaf(rg, df[c], acc[c]);
                    }
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    if (lanep < T) { // d k(x_j, x_i) / d y_n = -(2/h) sum_m R G (y~_n - x~_m)
#pragma unroll
                        for (int c = 0; c < DPAD; ++c) xl[lanep * DPAD + c] = -(w_ji * m2h) * acc[c];
                    }
                }
            }
        } else if (GRAD && SYM) {
            for (int e = lane; e < T * DPAD; e += 64) wl.Dc[e] = 0.f; // idle wavefront: nothing to add to the column
        }

        if (GRAD && SYM) {
            __syncthreads(); // every wavefront has parked its column-side sums
            float *dstc = a.cslab + (size_t)item * (T * d);
            for (int e = tid; e < T * d; e += NT) {
                const int nn = e / d, c = e - nn * d;
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < BNW; ++w) s += wl_all[w].Dc[nn * DPAD + c];
                dstc[e] = s;
            }
        }
    }
    if (GRAD && row_ok) { // the segment's row-side sums
        const int tot = T * d;
        double *dstr = a.rseg + (((size_t)(kq + (int)blockIdx.x)) * BNW + wave) * (size_t)tot;
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        for (int e = lane; e < tot; e += 64) {
            const int m = e / d, c = e - m * d;
            dstr[e] = (double)wl.rowacc[m * DPAD + c];
        }
    }
    remaining -= ncolr;
    ++kq;
    cstart = 0;
    } // row tiles of the range
}

bool band_supported(int A, int B, int T, int d, int n, int kind, unsigned flags)
{
    (void)A; (void)B;
    if (n < 1 || n > 7 || T < 3 || T > BTMAX || d > 16) return false;
    const int P = (T - 1) << n;
    if (P <= 128 || P > BPMAX) return false;
    if (kind != SIGSVGD_STATIC_RBF) return false;
    if (flags & SIGSVGD_FLAG_NAIVE_SOLVER) return false;
    return true;
}

namespace {
inline int band_nw(int A, int B, bool sym)
{
    const long long pairs = sym ? (long long)A * (A + 1) / 2 : (long long)A * B;
    return pairs <= 4ll * device_cu_count() ? 4 : 8;
}
inline GradGeom band_geometry(int A, int B, int T, int d, bool sym, int nw)
{
    return grad_geometry(A, B, T * d, sym, 0, 1, false, nw, (long long)device_cu_count());
}
inline size_t band_wsk_per_wave(int T, int n)
{
    const int P = (T - 1) << n;
    return (size_t)((P + 63) >> 6) * (size_t)(P + 63) * 64 + 16 * 64; // floats (+ 16 rows in front: the ring's loads need no clamp)
}
inline size_t band_wsk_bytes(int T, int n)
{
    return (((size_t)device_cu_count() * 8 * band_wsk_per_wave(T, n) * sizeof(float)) + 255) & ~(size_t)255;
}
} // namespace

namespace {
inline size_t band_flag_bytes(int A, int B) { return (((size_t)A * B + 255) & ~(size_t)255) + generic_repair_bytes(); }
} // namespace

int band_workspace_bytes(int A, int B, int T, int d, int n, int want_grad, size_t *bytes)
{
    *bytes = band_flag_bytes(A, B) + 512;
    if (!want_grad) return SIGSVGD_OK;
    const GradGeom o = band_geometry(A, B, T, d, false, band_nw(A, B, false));
    size_t need = o.rseg_bytes;
    if (A == B) {
        const GradGeom y = band_geometry(A, B, T, d, true, band_nw(A, B, true));
        if (y.rseg_bytes + y.cslab_bytes > need) need = y.rseg_bytes + y.cslab_bytes;
    }
    *bytes = need + band_wsk_bytes(T, n) + band_flag_bytes(A, B) + 1024;
    return SIGSVGD_OK;
}

namespace {
template <int DPAD, int BNW>
int band_launch_variant(const GramProblem &p, BandArgs &a, const GradGeom &g, bool grad, bool sym)
{
    if (g.tm.owned <= 0 || g.nitems <= 0) return SIGSVGD_OK;
    a.tm = g.tm;
    a.nitems = g.nitems;
    dim3 grid((unsigned)g.grid), block(BNW * 64);
    const bool comp = p.n >= 5;
#define SIGB_LAUNCH(G, S)                                                                                       \
    {                                                                                                           \
        if (comp) hipLaunchKernelGGL((gram_band_kernel<DPAD, G, S, true, BNW>), grid, block, 0, p.stream, a);   \
        else hipLaunchKernelGGL((gram_band_kernel<DPAD, G, S, false, BNW>), grid, block, 0, p.stream, a);       \
    }
    if (grad && sym)
        SIGB_LAUNCH(true, true)
    else if (grad)
        SIGB_LAUNCH(true, false)
    else if (sym)
        SIGB_LAUNCH(false, true)
    else
        SIGB_LAUNCH(false, false)
#undef SIGB_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "launch gram_band_kernel");
    return SIGSVGD_OK;
}
} // namespace

int band_launch(const GramProblem &p)
{
    const bool grad = p.gradX_out != nullptr;
    const bool sym = (p.flags & SIGSVGD_FLAG_Y_IS_X) && p.A == p.B;
    BandArgs a;
    a.X = p.X; a.Y = p.Y; a.go = p.grad_out; a.K = p.K_out; a.rseg = nullptr; a.cslab = nullptr; a.wsk = nullptr;
    a.wsk_per_wave = band_wsk_per_wave(p.T, p.n);
    a.io64 = p.dtype == SIGSVGD_F64; a.A = p.A; a.B = p.B; a.T = p.T; a.d = p.d; a.n = p.n;
    a.symw = (p.flags & SIGSVGD_FLAG_SYM) ? 1 : 0; a.inv_h = p.inv_h;
    a.nitems = 0;
    if (a.symw && p.A != p.B) {
        set_error("sym backward needs A == B");
        return SIGSVGD_E_BADARG;
    }
    const int nw = band_nw(p.A, p.B, sym);
    const GradGeom g = band_geometry(p.A, p.B, p.T, p.d, sym, nw);
    const size_t slabs = grad ? (g.rseg_bytes + g.cslab_bytes + 255) & ~(size_t)255 : 0;
    const size_t need = band_flag_bytes(p.A, p.B) + slabs + (grad ? band_wsk_bytes(p.T, p.n) : 0) + 256;
    if (!p.ws || p.ws_bytes < need) {
        set_error("band: workspace %zu B < required %zu B", p.ws_bytes, need);
        return SIGSVGD_E_WORKSPACE;
    }
    unsigned char *base = reinterpret_cast<unsigned char *>((reinterpret_cast<uintptr_t>(p.ws) + 255) & ~(uintptr_t)255);
    a.kflag = base;
    base += band_flag_bytes(p.A, p.B);
    if (grad) {
        a.rseg = reinterpret_cast<double *>(base);
        a.cslab = sym ? reinterpret_cast<float *>(base + g.rseg_bytes) : nullptr;
        a.wsk = reinterpret_cast<float *>(base + slabs);
    }
    int rc;
    if (nw == 4)
        rc = p.d <= 8 ? band_launch_variant<8, 4>(p, a, g, grad, sym) : band_launch_variant<16, 4>(p, a, g, grad, sym);
    else
        rc = p.d <= 8 ? band_launch_variant<8, 8>(p, a, g, grad, sym) : band_launch_variant<16, 8>(p, a, g, grad, sym);
    if (rc) return rc;
    // fp64 pass of the coverage kernel over the flagged pairs (a few microseconds when there are none)
    rc = generic_repair_launch(p, a.kflag, nullptr, sym, g.tm, nw);
    if (rc || !grad) return rc;
    return grad_reduce_launch(g, a.rseg, a.cslab, p.gradX_out, p.dtype == SIGSVGD_F64, p.A, p.B, p.T * p.d, sym, p.stream);
}

} // namespace sigsvgd
