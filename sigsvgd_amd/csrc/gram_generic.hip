// Generic signature-kernel Gram forward/backward kernel: any path length T, dyadic order n and
// channel count d whose per-pair state fits the CU's 160 KB LDS.  One wavefront per (i, j-chunk)
// work item; the refined P x P PDE grid is swept in bands of 64 rows, one grid row per lane,
// anti-diagonal by anti-diagonal (lane l is at column s-l on step s).  Neighbour values move
// between lanes with wave shifts; the band boundary row lives in LDS.  For the backward pass the
// forward solution is parked in a per-workgroup HBM scratch in [step][lane] order (coalesced
// 256-B rows, written and re-read by the same wavefront, so it stays in L2).
//
// With Y == X and enough pairs to fill the chip (>= 4096) each unordered pair is solved once: K is mirrored and
// the pair's gradient with respect to x_j comes from a second contraction of the same coarse scatter, added
// in fixed order by the reduction kernel from a per-pair slab; work items are then pulled from a counter.
//
// This is the coverage kernel (reference call sites use T=3..30 with n=2..6); the
// headline shapes (n=0, T<=64) take the register-resident kernel in gram_fast.hip.
//
// Reference semantics: sigkernel _SigKernelGram.forward/backward [RECALLED, SURVEY.md App. A];
// static kernel src/kernels/_traj_kernels.py:176-195.
#include <atomic>

#include "sig_common.h"
#ifdef SIGSVGD_PHASE_STAMPS
#include <cstdio>
#endif

namespace sigsvgd {

// Diagnostic build (-DSIGSVGD_PHASE_STAMPS, scripts/dev/phase_stamps.py N T d dyadic<n>): s_memtime per phase, summed
// over waves and printed after the launch.  Compiled out of the product.
#ifdef SIGSVGD_PHASE_STAMPS
#define SIG_GSTAMP(i)                                                        \
    {                                                                        \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();        \
        gph_[i] += now_ - gtl_;                                              \
        gtl_ = now_;                                                         \
    }
#else
#define SIG_GSTAMP(i)
#endif

struct GenericArgs {
    const void *X, *Y, *grad_out;
    void *K_out;
    double *partials; // [A][nchunks][T*d]
    double *colslab;  // [A][B][T*d] column-side gradients of the symmetric solve (yx): pair (i, j > i) writes its own block,
                      // reduce_partials_kernel adds the blocks of a column in row order (no atomics: reproducible bits)
    float *wsk;       // [grid][nbands*nsteps*64]
    int yx;           // Y is X: solve the pairs j >= i only, mirror K, add d k(x_j, x_i)/d x_j through colacc
    unsigned long long *next_item; // work counter (zeroed by the launcher): items are pulled, not assigned, because
                                   // with yx their cost varies from nothing to JC pairs
    int A, B, T, d, dp, n, r, P, Tm, TmS, nbands, nsteps, JC, nchunks, kind, naive, sym, want_grad;
    int big; // long paths (dyadic order 0 only): fp64 increments per band, S in the launch's scratch, no LDS gradient accumulator
    // fp64 pass over the pairs a fp32-sweep kernel flagged (generic_repair_launch; forward only): the pairs with
    // flags[i * B + j] != 0 among the rows of the tiles `tm` owns (tile_rows rows each), j >= i with yx
    const unsigned char *flags;
    TileMap tm;
    int tile_rows;
    double inv_h, inv_r2;
    long long total_items;
    size_t wsk_per_block;
#ifdef SIGSVGD_PHASE_STAMPS
    unsigned long long *stamps;
#endif
};

__host__ __device__ inline size_t generic_lds_bytes(int T, int d, int n, int want_grad, int big, int dd = 0)
{
    const int dp = (d % 2 == 0) ? d + 1 : d;
    const int Tm = T - 1, TmS = Tm | 1, P = (1 << n) * Tm;
    size_t dbl = (size_t)2 * T * dp + 2 * T + (P + 2) + 64; // (+64: per-lane dump cells of the sweeps' boundary stores)
    if (big) // long paths (dyadic order 0): fp64 increments of ONE band of 64 rows + the static-kernel row beyond it; S lives in
        return (dbl + (size_t)kWave * TmS + T) * sizeof(double); // the launch's scratch in global memory
    size_t flt = (size_t)Tm * TmS;
    if (want_grad) dbl += (size_t)Tm * Tm + (size_t)T * dp; // S fp64 + gradient accumulator
    if (dd) dbl += (size_t)Tm * TmS, flt -= (size_t)Tm * TmS; // increments kept in fp64 (see DT below)
    return dbl * sizeof(double) + flt * sizeof(float);
}

// NAIVE / GRAD / BIG are compile-time: tested per sweep step, each of them was a taken branch on the one wave's
// dependent chain.
// BIG (round 4): paths too long for the whole-grid tables (dyadic order 0).  The increments are formed PER BAND of 64 rows, in
// fp64, right before the band is swept (forwards and again backwards: the static kernel is evaluated twice), and S = K_fwd * U
// goes to the launch's scratch in the [step][lane] order of the stored forward solution (coalesced 256-B rows, fp32: each entry
// is written once).  Per-pair LDS: 64 (T-1) + 2 T d doubles -- T = 128, d = 16 takes 104 KB -- so this layout is fp64 end to
// end for every shape of the quadrant kernel (T <= 128) with and without the gradient; rounds 2-3 kept the whole table in
// fp32 there and the exact pass inherited 3e-5 from its rounding (soak case: T = 100, d = 2, gradient, forced coverage kernel).
// DT: storage type of the increment table.  float (rounds 1-2): 6e-8 per increment, invisible while K grows along the
// grid -- but where the discrete solution oscillates (rough paths in one or two channels) K[P][P] is
// a small remainder of much larger values and inherits ~3e-7 .. 3e-6 (T = 64 .. 128) of their ratio to it from the rounded
// increments, whatever the precision of the sweeps.  double wherever the table fits next to the rest (always for the
// reference's own shapes; the fp64 pass over flagged pairs and force_generic whenever 160 KB allow).
template <typename IO, bool NAIVE, bool GRAD, bool BIG, typename DT>
__global__ __launch_bounds__(64) void gram_generic_kernel(GenericArgs a)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int lane = threadIdx.x;
    const int T = a.T, d = a.d, dp = a.dp, Tm = a.Tm, TmS = a.TmS, P = a.P, n = a.n, r = a.r;
    constexpr bool naive = NAIVE;
    const bool rbf = a.kind == SIGSVGD_STATIC_RBF;

    double *xs = reinterpret_cast<double *>(smem_raw);
    double *ys = xs + (size_t)T * dp;
    double *xn = ys + (size_t)T * dp;
    double *yn = xn + T;
    double *rowbuf = yn + T;
    constexpr bool big = BIG;
    double *dump = rowbuf + (P + 2); // [64]: where the lanes that have nothing to hand over store
    double *Sm = dump + kWave;
    double *acc = Sm + ((GRAD && !big) ? (size_t)Tm * Tm : 0);
    DT *Dm = reinterpret_cast<DT *>(acc + ((GRAD && !big) ? (size_t)T * dp : 0)); // big: rows of the band in work only
    double *gnext = reinterpret_cast<double *>(Dm + (size_t)kWave * TmS);         // big: static-kernel row beyond the band [T]

    const IO *X = static_cast<const IO *>(a.X);
    const IO *Y = static_cast<const IO *>(a.Y);
    const IO *GO = static_cast<const IO *>(a.grad_out);
    IO *Kout = static_cast<IO *>(a.K_out);
    float *wsk = a.wsk + (size_t)blockIdx.x * a.wsk_per_block;
    float *wss = wsk + (size_t)a.nbands * a.nsteps * kWave; // big: S in the order of the stored forward solution
    // S[aa][bb] of the pair in work (big mode: row aa = band aa / 64, lane aa % 64, filed under step lane + bb)
    auto Sat = [&](int aa, int bb) -> double {
        if (big) return (double)wss[((size_t)(aa >> 6) * a.nsteps + (aa & 63) + bb) * kWave + (aa & 63)];
        return Sm[aa * Tm + bb];
    };
    // big: increments of the cell rows a0 .. a0 + nrows - 1 (one band), fp64, from static-kernel rows a0 .. a0 + nrows
    auto band_increments = [&](int a0, int nrows) {
        __syncthreads();
        for (int q = lane; q < T; q += kWave) { // the row beyond the band, one column per lane
            double dot = 0.0;
            const int pr = a0 + nrows; // <= T - 1
            for (int c = 0; c < d; ++c) dot = __builtin_fma(xs[pr * dp + c], ys[q * dp + c], dot);
            gnext[q] = rbf ? exp64((2.0 * dot - xn[pr] - yn[q]) * a.inv_h) : dot;
        }
        __syncthreads();
        const int p = a0 + lane;
        const bool valid = lane < nrows;
        double g_prev = 0.0, gn_prev = 0.0;
        for (int q = 0; q < T; ++q) {
            double gq = 0.0;
            if (valid) {
                double dot = 0.0;
                for (int c = 0; c < d; ++c) dot = __builtin_fma(xs[p * dp + c], ys[q * dp + c], dot);
                gq = rbf ? exp64((2.0 * dot - xn[p] - yn[q]) * a.inv_h) : dot;
            }
            const double rd = gq - g_prev;
            g_prev = gq;
            const double gn = gnext[q];
            double rdn = shfl_down_f64(rd);
            rdn = (lane == nrows - 1) ? gn - gn_prev : rdn;
            gn_prev = gn;
            if (q >= 1 && valid) Dm[lane * TmS + (q - 1)] = (DT)(rdn - rd);
        }
        __syncthreads();
    };

#ifdef SIGSVGD_PHASE_STAMPS
    unsigned long long gph_[6] = {0, 0, 0, 0, 0, 0}, gtl_ = __builtin_amdgcn_s_memtime();
#endif
    for (long long round = 0;; ++round) {
        long long item;
        if (a.next_item) { // symmetric solve: item costs vary from nothing to JC pairs -> pull from a counter
            unsigned long long pulled = 0;
            if (lane == 0) pulled = atomicAdd(a.next_item, 1ull);
            item = (long long)__shfl(pulled, 0, kWave);
        } else {
            item = (long long)blockIdx.x + round * gridDim.x;
        }
        if (item >= a.total_items) break;
        const int i = (int)(item / a.nchunks);
        const int chunk = (int)(item % a.nchunks);
        const int j0 = chunk * a.JC;
        const int j1 = min(a.B, j0 + a.JC);
        const IO *xi = X + (size_t)i * T * d;
        const bool empty = a.yx && j1 <= i; // chunk entirely left of the diagonal: solved from the other side
        if (a.flags) { // nothing flagged in this chunk (the usual case): next item, before anything is staged
            bool any = false;
            if (a.tm.kq_of_tile(i / a.tile_rows) >= 0)
                for (int j = max(j0, a.yx ? i : 0) + lane; j < j1; j += kWave) any |= a.flags[(size_t)i * a.B + j] != 0;
            if (__builtin_amdgcn_ballot_w64(any) == 0) continue; // (one wavefront per workgroup: uniform)
        }

        double *slab = GRAD ? a.partials + ((size_t)i * a.nchunks + chunk) * T * d : nullptr;
        if (GRAD && big)
            for (int e = lane; e < T * d; e += kWave) slab[e] = 0.0; // accumulated in place (L2 resident)
        // ---- stage x_i (centred on its first point for the translation-invariant RBF) ----------
        __syncthreads();
        for (int e = lane; e < T * dp; e += kWave) {
            const int t = e / dp, c = e % dp;
            double v = 0.0;
            if (c < d) v = (double)xi[t * d + c] - (rbf ? (double)xi[c] : 0.0);
            xs[e] = v;
            if (GRAD && !big) acc[e] = 0.0;
        }
        __syncthreads();
        for (int t = lane; t < T; t += kWave) {
            double s = 0.0;
            for (int c = 0; c < d; ++c) s = __builtin_fma(xs[t * dp + c], xs[t * dp + c], s);
            xn[t] = s;
        }

        unsigned long long fmask = 0; // fp64 pass: flagged columns among fbase .. fbase + 63 not visited yet
        int fbase = j0 - kWave;
        for (int j = empty ? j1 : j0; j < j1; ++j) {
            if (a.flags) { // the next flagged column (64 flags per load, the set bits taken in order)
                while (fmask == 0 && fbase + kWave < j1) {
                    fbase += kWave;
                    const int jc = fbase + lane;
                    const bool f = jc < j1 && !(a.yx && jc < i) && a.flags[(size_t)i * a.B + jc] != 0;
                    fmask = __builtin_amdgcn_ballot_w64(f);
                }
                if (fmask == 0) break;
                j = fbase + __builtin_ctzll(fmask);
                fmask &= fmask - 1;
            }
            if (a.yx && j < i) continue; // (one wavefront per workgroup: uniform)
            const IO *yj = Y + (size_t)j * T * d;
            __syncthreads();
            for (int e = lane; e < T * dp; e += kWave) {
                const int t = e / dp, c = e % dp;
                double v = 0.0;
                if (c < d) v = (double)yj[t * d + c] - (rbf ? (double)xi[c] : 0.0);
                ys[e] = v;
            }
            if (GRAD && !big)
                for (int e = lane; e < Tm * Tm; e += kWave) Sm[e] = 0.0;
            __syncthreads();
            for (int t = lane; t < T; t += kWave) {
                double s = 0.0;
                for (int c = 0; c < d; ++c) s = __builtin_fma(ys[t * dp + c], ys[t * dp + c], s);
                yn[t] = s;
            }
            __syncthreads();

            SIG_GSTAMP(0)
            // ---- phase 1: static kernel rows -> increments D (fp64 arithmetic, stored as DT) ----
            for (int rb = 0; rb < (big ? 0 : Tm); rb += kWave - 1) { // (big: per band, below)
                const int p = rb + lane;
                const bool valid = p < T;
                double g_prev = 0.0;
                for (int q = 0; q < T; ++q) {
                    double gq = 0.0;
                    if (valid) {
                        double dot = 0.0;
                        for (int c = 0; c < d; ++c) dot = __builtin_fma(xs[p * dp + c], ys[q * dp + c], dot);
                        gq = rbf ? exp64((2.0 * dot - xn[p] - yn[q]) * a.inv_h) : dot;
                    }
                    const double rd = gq - g_prev;
                    g_prev = gq;
                    const double rdn = shfl_down_f64(rd);
                    if (q >= 1 && lane < kWave - 1 && p + 1 < T) Dm[p * TmS + (q - 1)] = (DT)(rdn - rd);
                }
            }
            __syncthreads();

            SIG_GSTAMP(1)
            // ---- phase 2: forward Goursat sweep ------------------------------------------------
            // One wave, one dependent chain: a step costs what its instruction count costs (~5 cycles each), plus every
            // load it has to wait for.  So the step is branch-free (results of lanes outside the grid are computed and
            // dropped by selects, the K_fwd store is unconditional -- only entries of grid cells are ever read back) and
            // the two LDS operands of step s+1 (the lane's increment, lane 0's boundary value) are fetched during step s.
            double Kval = 1.0;
            for (int kb = 0; kb < a.nbands; ++kb) {
                const int p = kb * kWave + lane;
                const bool rowvalid = p < P;
                const bool first = kb == 0;
                if (big) band_increments(kb * kWave, min(kWave, P - kb * kWave));
                const DT *Drow = Dm + (size_t)(big ? min(lane, P - 1 - kb * kWave) : (min(p, P - 1) >> n)) * TmS;
                float *wp = wsk + (size_t)kb * a.nsteps * kWave + lane;
                double cur = 1.0, upprev = 1.0;
                int q = -lane;
                DT gf = Drow[0];                        // step 0 (only lane 0 is inside the grid)
                double rb = rowbuf[1];                  // lane 0's upper neighbour on step s: rowbuf[s + 1]
                rb = first ? 1.0 : rb;
                for (int s = 0; s < a.nsteps; ++s, ++q) {
                    const bool active = rowvalid && q >= 0 && q < P;
                    const DT gfn = Drow[min(max(q + 1, 0), P - 1) >> n];
                    const double rbr = rowbuf[min(s + 2, P)];
                    const double rbn = first ? 1.0 : rbr;
                    double up_in = shfl_up_f64(cur);
                    up_in = (lane == 0) ? rb : up_in;
                    const double g = (double)gf * a.inv_r2;
                    const double nw = stencil(cur, up_in, upprev, g, naive);
                    if (GRAD) { // (issued from inline asm: hipcc otherwise makes every step wait for the previous step's store)
                        const float kst = (float)upprev;
                        asm volatile("global_store_dword %0, %1, off" ::"v"(wp + (size_t)s * kWave), "v"(kst));
                    }
                    *((lane == kWave - 1 && active) ? rowbuf + (q + 1) : dump + lane) = nw; // (no branch: lane-selected address)
                    cur = active ? nw : cur;
                    upprev = active ? up_in : upprev;
                    gf = gfn;
                    rb = rbn;
                }
                if (p == P - 1) Kval = cur;
            }
            if (((P - 1) & (kWave - 1)) == lane) {
                Kout[(size_t)i * a.B + j] = (IO)Kval;
                if (a.yx && j != i) Kout[(size_t)j * a.B + i] = (IO)Kval;
            }

            SIG_GSTAMP(2)
            if (!GRAD) continue;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the forward solution is in L2 before it is read back
            __syncthreads();

            // ---- phase 3: reverse sweep, GG = K_fwd[p,q] * U[p+1,q+1], block-summed into S ------
            // (same shape as the forward sweep: branch-free step, the operands of the next steps fetched ahead -- the
            //  stored forward solution two steps ahead, it comes from L2)
            for (int kb = a.nbands - 1; kb >= 0; --kb) {
                const int p = kb * kWave + lane;
                const bool rowvalid = p < P;
                const int L = min(kWave, P - kb * kWave);
                const int arow = min(p, P - 1) >> n;
                if (big && kb != a.nbands - 1) band_increments(kb * kWave, L); // (last band: the forward sweep's table is still there)
                const DT *Drow = Dm + (size_t)(big ? min(lane, L - 1) : arow) * TmS;
                float *wsrow = wss + (size_t)kb * a.nsteps * kWave + lane;
                const bool lastband = kb == a.nbands - 1;
                const bool hands_over = lane == 0 && kb > 0;
                double cur = 1.0, dprev = 1.0, sb = 0.0;
                const int nsp = P + L - 1;
                int q = P - 1 + (L - 1 - lane);
                // K_fwd[p, q] was stored on forward step p_local + q = lane + q: one row of 64 per reverse step, descending
                const float *wrow = wsk + (size_t)kb * a.nsteps * kWave + lane;
                int R = P - 1 + L - 1; // row of step sp = 0
                DT gf = Drow[min(max(q, 0), P - 1) >> n];
                double rb = rowbuf[P - 1]; // lane L-1's lower neighbour on step sp: rowbuf[P - 1 - sp]
                rb = lastband ? 1.0 : rb;
                // ring of the next KPF rows of the stored forward solution (an L2 round trip is ~8 steps long); the loop
                // runs in groups of KPF steps, the steps past nsp have no lane inside the grid and change nothing
                constexpr int KPF = 8;
                float kfr[KPF];
#pragma unroll
                for (int u = 0; u < KPF; ++u) kfr[u] = wrow[(size_t)max(R - u, 0) * kWave];
                for (int sp0 = 0; sp0 < nsp; sp0 += KPF) {
#pragma unroll
                    for (int u = 0; u < KPF; ++u, --q, --R) {
                        const int sp = sp0 + u;
                        const bool active = rowvalid && q >= 0 && q < P;
                        const DT gfn = Drow[min(max(q - 1, 0), P - 1) >> n];
                        const double rbr = rowbuf[max(P - 2 - sp, 0)];
                        const double rbn = lastband ? 1.0 : rbr;
                        const double kf = (double)kfr[u];
                        kfr[u] = wrow[(size_t)max(R - KPF, 0) * kWave];
                        double down_in = shfl_down_f64(cur);
                        down_in = (lane == L - 1) ? rb : down_in;
                        const double g = (double)gf * a.inv_r2;
                        if (big) { // r == 1: the block is this cell; filed where K_fwd[p][q] is (row R of the band, this lane)
                            const float sst = active ? (float)(kf * dprev) : 0.f;
                            // (the padding steps past the last anti-diagonal, R < 0, go to a spare row behind the table)
                            asm volatile("global_store_dword %0, %1, off" ::"v"(R >= 0 ? wsrow + (size_t)R * kWave : wss + (size_t)a.nbands * a.nsteps * kWave + lane), "v"(sst));
                        } else {
                            sb = active ? __builtin_fma(kf, dprev, sb) : sb;
                            if (active && (q & (r - 1)) == 0) {
                                atomicAdd(&Sm[arow * Tm + (q >> n)], sb * a.inv_r2);
                                sb = 0.0;
                            }
                        }
                        const double nw = stencil(cur, down_in, dprev, g, naive);
                        *((hands_over && active) ? rowbuf + q : dump + lane) = nw;
                        cur = active ? nw : cur;
                        dprev = active ? down_in : dprev;
                        gf = gfn;
                        rb = rbn;
                    }
                }
            }
            if (big) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // S is in L2 before the assembly reads it
            __syncthreads();

            SIG_GSTAMP(3)
            // ---- phase 4: chain S -> R -> static-kernel derivative -> per-point gradient --------
            double w = 1.0;
            if (GO) {
                w = (double)GO[(size_t)i * a.B + j];
                if (a.sym) w += (double)GO[(size_t)j * a.B + i];
            } else if (a.sym) {
                w = 2.0;
            }
            // Short paths leave most lanes of the wave without a point row: `parts` lanes (a power of two, adjacent) then
            // share a row, each contracting a slice of the columns; their sums meet in a fixed-order butterfly.
            const int parts = (big || T > 32) ? 1 : (T <= 4 ? 16 : (T <= 8 ? 8 : (T <= 16 ? 4 : 2)));
            const int cslice = (T + parts - 1) / parts;
            const int niter = parts > 1 ? 1 : (T + kWave - 1) / kWave; // (T * parts <= 64: one pass)
            for (int it = 0; it < niter; ++it) {
                const int m = parts > 1 ? lane / parts : lane + it * kWave;
                const bool mvalid = m < T;
                const int nn0 = (lane % parts) * cslice, nn1 = mvalid ? min(T, nn0 + cslice) : nn0;
                for (int c0 = 0; c0 < d; c0 += 16) {
                    double accv[16];
#pragma unroll
                    for (int c = 0; c < 16; ++c) accv[c] = 0.0;
                    double s0 = 0.0;
                    for (int nn = nn0; nn < nn1; ++nn) {
                        double R = 0.0;
                        if (m >= 1 && nn >= 1) R += Sat(m - 1, nn - 1);
                        if (m < Tm && nn < Tm) R += Sat(m, nn);
                        if (m >= 1 && nn < Tm) R -= Sat(m - 1, nn);
                        if (m < Tm && nn >= 1) R -= Sat(m, nn - 1);
                        double rg = R;
                        if (rbf) {
                            double dot = 0.0;
                            for (int c = 0; c < d; ++c) dot = __builtin_fma(xs[m * dp + c], ys[nn * dp + c], dot);
                            rg = R * exp64((2.0 * dot - xn[m] - yn[nn]) * a.inv_h);
                            s0 += rg;
                        }
#pragma unroll
                        for (int c = 0; c < 16; ++c)
                            if (c0 + c < d) accv[c] = __builtin_fma(rg, ys[nn * dp + c0 + c], accv[c]);
                    }
                    for (int off = 1; off < parts; off <<= 1) { // (every lane of the wave takes part; idle ones add zeros)
                        s0 += __shfl_xor(s0, off, kWave);
#pragma unroll
                        for (int c = 0; c < 16; ++c)
                            if (c0 + c < d) accv[c] += __shfl_xor(accv[c], off, kWave);
                    }
                    if (mvalid && lane % parts == 0) {
#pragma unroll
                        for (int c = 0; c < 16; ++c) {
                            if (c0 + c < d) {
                                const double val = rbf ? (-2.0 * a.inv_h) * (xs[m * dp + c0 + c] * s0 - accv[c]) : accv[c];
                                if (big)
                                    slab[m * d + c0 + c] = __builtin_fma(w, val, slab[m * d + c0 + c]);
                                else
                                    acc[m * dp + c0 + c] = __builtin_fma(w, val, acc[m * dp + c0 + c]);
                            }
                        }
                    }
                }
            }
            SIG_GSTAMP(4)
            // ---- phase 4b (Y is X, j != i): the same pair seen from x_j, d k(x_j, x_i) / d x_j ---------
            if (a.yx && j != i) {
                double wc = 1.0;
                if (GO) {
                    wc = (double)GO[(size_t)j * a.B + i];
                    if (a.sym) wc += (double)GO[(size_t)i * a.B + j];
                } else if (a.sym) {
                    wc = 2.0;
                }
                for (int nn = lane; nn < T; nn += kWave) {
                    for (int c0 = 0; c0 < d; c0 += 16) {
                        double accv[16];
#pragma unroll
                        for (int c = 0; c < 16; ++c) accv[c] = 0.0;
                        double s0 = 0.0;
                        for (int m = 0; m < T; ++m) {
                            double R = 0.0;
                            if (m >= 1 && nn >= 1) R += Sat(m - 1, nn - 1);
                            if (m < Tm && nn < Tm) R += Sat(m, nn);
                            if (m >= 1 && nn < Tm) R -= Sat(m - 1, nn);
                            if (m < Tm && nn >= 1) R -= Sat(m, nn - 1);
                            double rg = R;
                            if (rbf) {
                                double dot = 0.0;
                                for (int c = 0; c < d; ++c) dot = __builtin_fma(xs[m * dp + c], ys[nn * dp + c], dot);
                                rg = R * exp64((2.0 * dot - xn[m] - yn[nn]) * a.inv_h);
                                s0 += rg;
                            }
#pragma unroll
                            for (int c = 0; c < 16; ++c)
                                if (c0 + c < d) accv[c] = __builtin_fma(rg, xs[m * dp + c0 + c], accv[c]);
                        }
#pragma unroll
                        for (int c = 0; c < 16; ++c) {
                            if (c0 + c < d) {
                                const double val = rbf ? (-2.0 * a.inv_h) * (ys[nn * dp + c0 + c] * s0 - accv[c]) : accv[c];
                                a.colslab[(((size_t)i * a.B + j) * T + nn) * d + c0 + c] = wc * val;
                            }
                        }
                    }
                }
            }
            SIG_GSTAMP(5)
        } // j

        if (GRAD && !big) {
            __syncthreads();
            for (int e = lane; e < T * d; e += kWave) slab[e] = acc[(e / d) * dp + (e % d)];
        }
    }
#ifdef SIGSVGD_PHASE_STAMPS
    SIG_GSTAMP(0)
    if (lane == 0 && a.stamps)
        for (int k = 0; k < 6; ++k) atomicAdd(&a.stamps[k], gph_[k]);
#endif
}

// gradX[i][e] = sum_chunk partials[i][chunk][e] + sum_{i' < i} colslab[i'][i][e]  (fixed order => deterministic)
template <typename IO>
__global__ void reduce_partials_kernel(const double *partials, const double *colslab, IO *gradX, int A, int nchunks,
                                       int TD)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)A * TD) return;
    const size_t i = idx / TD, e = idx % TD;
    double s = 0.0;
    for (int c = 0; c < nchunks; ++c) s += partials[(i * nchunks + c) * TD + e];
    if (colslab)
        for (size_t r = 0; r < i; ++r) s += colslab[(r * A + i) * TD + e];
    gradX[idx] = (IO)s;
}

namespace {
// Y is X: each unordered pair once -- when there are enough pairs to fill the chip (below that the launch is latency-bound
// and the second contraction pass of the symmetric solve only lengthens the critical path) and the per-pair slab of the
// column-side gradients stays below 1 GiB (beyond that: ordered pairs, twice the solves, no slab)
bool generic_solves_unordered(int A, int B, int T, int d, int want_grad, unsigned flags)
{
    if (!(flags & SIGSVGD_FLAG_Y_IS_X) || A != B || (long long)A * B < 4096) return false;
    if (want_grad && (size_t)A * B * T * d * sizeof(double) > ((size_t)1 << 30)) return false;
    return true;
}
struct GenericPlan {
    int dp, Tm, TmS, r, P, nbands, nsteps, JC, nchunks, grid, big, dd;
    long long items;
    size_t lds, partial_bytes, col_bytes, wsk_per_block, wsk_bytes;
};

// precise: keep the increments in fp64 whenever the table fits 160 KB (force_generic, the fp64 pass over flagged pairs);
// otherwise only where it costs no occupancy (<= 20 KB per pair: every shape of the reference's own calls)
int make_plan(int A, int B, int T, int d, int n, int want_grad, GenericPlan &pl, bool yx = false, bool precise = false)
{
    if (A < 1 || B < 1 || T < 2 || d < 1 || n < 0 || n > 10) {
        set_error("generic: bad shape A=%d B=%d T=%d d=%d n=%d", A, B, T, d, n);
        return SIGSVGD_E_BADARG;
    }
    pl.dp = (d % 2 == 0) ? d + 1 : d;
    pl.Tm = T - 1;
    pl.TmS = pl.Tm | 1;
    pl.r = 1 << n;
    const long long P64 = (long long)pl.r * pl.Tm;
    if (P64 > 16384) {
        set_error("generic: refined grid P=%lld too large", P64);
        return SIGSVGD_E_UNSUPPORTED;
    }
    pl.P = (int)P64;
    pl.nbands = (pl.P + kWave - 1) / kWave;
    pl.nsteps = pl.P + kWave - 1;
    pl.big = 0;
    pl.dd = 0;
    pl.lds = generic_lds_bytes(T, d, n, want_grad, 0);
    // increments in fp64: `precise` (force_generic, the fp64 pass over flagged pairs) whenever the whole-grid table fits 160 KB;
    // otherwise only where it costs no occupancy (<= 20 KB per pair: every shape of the reference's own calls)
    const size_t l2 = generic_lds_bytes(T, d, n, want_grad, 0, 1);
    if (l2 <= (precise ? (size_t)160 * 1024 : (size_t)20 * 1024)) {
        pl.dd = 1;
        pl.lds = l2;
    } else if (n == 0 && (precise || pl.lds > 160 * 1024)) { // long paths: per-band fp64 increments (fp64 end to end)
        pl.big = 1;
        pl.dd = 1;
        pl.lds = generic_lds_bytes(T, d, n, want_grad, 1);
    }
    if (pl.lds > 160 * 1024) {
        set_error("generic: per-pair state needs %zu B of LDS (> 160 KiB): T=%d d=%d n=%d", pl.lds, T, d, n);
        return SIGSVGD_E_UNSUPPORTED;
    }
    // j-chunk: enough work items to fill the chip, few enough partial slabs
    int JC = 32;
    while (JC > 1 && (long long)A * ((B + JC - 1) / JC) < (yx ? 16384 : 2048)) JC >>= 1; // yx: half the items are empty
    pl.JC = JC;
    pl.nchunks = (B + JC - 1) / JC;
    pl.items = (long long)A * pl.nchunks;
    const int per_cu = (int)((160 * 1024) / (pl.lds ? pl.lds : 1));
    int grid = device_cu_count() * (per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu));
    if ((long long)grid > pl.items) grid = (int)pl.items;
    pl.grid = grid;
    pl.partial_bytes = want_grad ? (size_t)A * pl.nchunks * T * d * sizeof(double) : 0;
    pl.col_bytes = (want_grad && yx) ? (((size_t)A * B * T * d * sizeof(double) + 255) & ~(size_t)255) : 0; // symmetric solve only
    // forward solution in [band][step][lane] order; long paths: S behind it in the same order + one spare row
    pl.wsk_per_block = want_grad ? (size_t)pl.nbands * pl.nsteps * kWave * (pl.big ? 2 : 1) + (pl.big ? kWave : 0) : 0;
    pl.wsk_bytes = pl.wsk_per_block * sizeof(float) * grid;
    return SIGSVGD_OK;
}
} // namespace

namespace {
inline size_t plan_bytes(const GenericPlan &pl) { return 512 + pl.partial_bytes + pl.col_bytes + pl.wsk_bytes; }
}

int generic_workspace_bytes(int A, int B, int T, int d, int n, int want_grad, bool precise, size_t *bytes)
{
    // the query carries no Y_IS_X promise: size for whichever of the ordered / symmetric plans needs more (the symmetric
    // one uses shorter column chunks, i.e. more partial slabs)
    GenericPlan pl;
    int rc = make_plan(A, B, T, d, n, want_grad, pl, false, precise);
    if (rc) return rc;
    *bytes = plan_bytes(pl);
    if (generic_solves_unordered(A, B, T, d, want_grad, SIGSVGD_FLAG_Y_IS_X)) {
        rc = make_plan(A, B, T, d, n, want_grad, pl, true, precise);
        if (rc) return rc;
        if (plan_bytes(pl) > *bytes) *bytes = plan_bytes(pl);
    }
    return SIGSVGD_OK;
}

namespace {
template <typename IO, bool NAIVE, bool GRAD, bool BIG, typename DT>
hipError_t generic_launch_one(const GenericPlan &pl, hipStream_t stream, const GenericArgs &a)
{
    // (once per instantiation, for the largest size any plan can ask for: the call costs ~10 us of host time, which a small
    //  launch -- the fp64 pass behind a 50-us kernel -- would pay every time
    //  -- per DEVICE: the attribute belongs to the current device's copy of the function, and a process may drive several)
    static std::atomic<unsigned long long> raised{0}; // bit = device ordinal (devices >= 64: raised every time)
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !((raised.load(std::memory_order_acquire) >> dev) & 1ull)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&gram_generic_kernel<IO, NAIVE, GRAD, BIG, DT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) raised.fetch_or(1ull << dev, std::memory_order_release);
    }
    hipLaunchKernelGGL((gram_generic_kernel<IO, NAIVE, GRAD, BIG, DT>), dim3(pl.grid), dim3(kWave), pl.lds, stream, a);
    return hipSuccess;
}
template <typename IO, bool NAIVE, typename DT>
hipError_t generic_dispatch2(bool grad, bool big, const GenericPlan &pl, hipStream_t stream, const GenericArgs &a)
{
    if constexpr (sizeof(DT) == 8) { // (the long-path layout is fp64 only: make_plan)
        if (big) return grad ? generic_launch_one<IO, NAIVE, true, true, DT>(pl, stream, a) : generic_launch_one<IO, NAIVE, false, true, DT>(pl, stream, a);
    }
    return grad ? generic_launch_one<IO, NAIVE, true, false, DT>(pl, stream, a) : generic_launch_one<IO, NAIVE, false, false, DT>(pl, stream, a);
}
template <typename IO>
hipError_t generic_dispatch1(bool naive, bool grad, bool big, const GenericPlan &pl, hipStream_t stream, const GenericArgs &a)
{
    if (pl.dd)
        return naive ? generic_dispatch2<IO, true, double>(grad, big, pl, stream, a) : generic_dispatch2<IO, false, double>(grad, big, pl, stream, a);
    return naive ? generic_dispatch2<IO, true, float>(grad, big, pl, stream, a) : generic_dispatch2<IO, false, float>(grad, big, pl, stream, a);
}
hipError_t generic_dispatch(bool f64, bool naive, bool grad, bool big, const GenericPlan &pl, hipStream_t stream,
                            const GenericArgs &a)
{
    return f64 ? generic_dispatch1<double>(naive, grad, big, pl, stream, a) : generic_dispatch1<float>(naive, grad, big, pl, stream, a);
}
} // namespace

int generic_launch(const GramProblem &p)
{
    const int want_grad = p.gradX_out != nullptr;
    const bool yx = generic_solves_unordered(p.A, p.B, p.T, p.d, want_grad, p.flags);
    GenericPlan pl;
    int rc = make_plan(p.A, p.B, p.T, p.d, p.n, want_grad, pl, yx, (p.flags & SIGSVGD_FLAG_FORCE_GENERIC) != 0);
    if (rc) return rc;
    const size_t need = plan_bytes(pl);
    if (p.ws == nullptr || p.ws_bytes < need) {
        set_error("generic: workspace %zu B < required %zu B", p.ws_bytes, need);
        return SIGSVGD_E_WORKSPACE;
    }
    const bool sym = (p.flags & SIGSVGD_FLAG_SYM) != 0;
    if (sym && p.A != p.B) {
        set_error("sym backward needs A == B");
        return SIGSVGD_E_BADARG;
    }
    GenericArgs a;
    a.X = p.X; a.Y = p.Y; a.grad_out = p.grad_out; a.K_out = p.K_out;
    unsigned char *base = reinterpret_cast<unsigned char *>((reinterpret_cast<uintptr_t>(p.ws) + 255) & ~(uintptr_t)255);
    a.next_item = nullptr;
    if (yx) {
        a.next_item = reinterpret_cast<unsigned long long *>(base);
        hipError_t ce = hipMemsetAsync(a.next_item, 0, sizeof(unsigned long long), p.stream);
        if (ce != hipSuccess) return hip_fail(ce, "hipMemsetAsync(work counter)");
    }
    a.partials = want_grad ? reinterpret_cast<double *>(base + 256) : nullptr;
    a.yx = yx ? 1 : 0;
    a.colslab = nullptr;
    unsigned char *after = want_grad ? reinterpret_cast<unsigned char *>(a.partials) + pl.partial_bytes : nullptr;
    if (want_grad && yx) a.colslab = reinterpret_cast<double *>(after);
    a.wsk = want_grad ? reinterpret_cast<float *>(after + pl.col_bytes) : nullptr;
    a.A = p.A; a.B = p.B; a.T = p.T; a.d = p.d; a.dp = pl.dp; a.n = p.n; a.r = pl.r; a.P = pl.P;
    a.Tm = pl.Tm; a.TmS = pl.TmS; a.nbands = pl.nbands; a.nsteps = pl.nsteps; a.JC = pl.JC;
    a.nchunks = pl.nchunks; a.kind = p.kind; a.naive = (p.flags & SIGSVGD_FLAG_NAIVE_SOLVER) ? 1 : 0;
    a.sym = sym ? 1 : 0; a.want_grad = want_grad; a.inv_h = p.inv_h;
    a.inv_r2 = 1.0 / ((double)pl.r * (double)pl.r);
    a.total_items = pl.items; a.wsk_per_block = pl.wsk_per_block; a.big = pl.big;
    a.flags = nullptr; a.tm = make_tilemap(1, 0, 1, false); a.tile_rows = 1;

#ifdef SIGSVGD_PHASE_STAMPS
    {
        static unsigned long long *dbg = nullptr;
        if (!dbg) (void)hipMalloc(&dbg, 6 * sizeof(unsigned long long));
        (void)hipMemsetAsync(dbg, 0, 6 * sizeof(unsigned long long), p.stream);
        a.stamps = dbg;
    }
#endif
    hipError_t e = generic_dispatch(p.dtype == SIGSVGD_F64, a.naive != 0, want_grad != 0, pl.big != 0, pl, p.stream, a);
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(generic)");
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "launch gram_generic_kernel");
#ifdef SIGSVGD_PHASE_STAMPS
    {
        unsigned long long hs[6];
        (void)hipStreamSynchronize(p.stream);
        (void)hipMemcpy(hs, a.stamps, sizeof(hs), hipMemcpyDeviceToHost);
        double tot = 0;
        for (int k = 0; k < 6; ++k) tot += (double)hs[k];
        static const char *nm[6] = {"staging/other", "phase 1 static kernel", "forward sweep", "reverse sweep",
                                    "phase 4 gradient", "phase 4b column side"};
        fprintf(stderr, "[phase stamps generic] A=%d T=%d d=%d n=%d grad=%d: ", p.A, p.T, p.d, p.n, want_grad);
        for (int k = 0; k < 6; ++k) fprintf(stderr, "%s %.1f%% | ", nm[k], 100.0 * (double)hs[k] / tot);
        fprintf(stderr, "total %.3e wave-cycles\n", tot);
    }
#endif
    if (want_grad) {
        const int TD = p.T * p.d;
        const size_t tot = (size_t)p.A * TD;
        const int bs = 256;
        const unsigned gs = (unsigned)((tot + bs - 1) / bs);
        if (p.dtype == SIGSVGD_F64)
            hipLaunchKernelGGL(reduce_partials_kernel<double>, dim3(gs), dim3(bs), 0, p.stream, a.partials,
                               static_cast<const double *>(a.colslab), static_cast<double *>(p.gradX_out), p.A,
                               pl.nchunks, TD);
        else
            hipLaunchKernelGGL(reduce_partials_kernel<float>, dim3(gs), dim3(bs), 0, p.stream, a.partials,
                               static_cast<const double *>(a.colslab), static_cast<float *>(p.gradX_out), p.A,
                               pl.nchunks, TD);
        e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "launch reduce_partials_kernel");
    }
    return SIGSVGD_OK;
}

// ---- fp64 pass over flagged pairs ------------------------------------------------------------------------------------
// The fp32-sweep kernels (gram_quad.hip) mark the pairs whose solution cancelled; this launch of the forward-only
// coverage kernel takes one row of K per work item, skips every row without a flag before staging anything (a few
// microseconds when nothing is flagged) and stores K of the flagged pairs again, from fp64 sweeps.  Items are assigned
// statically; no counter, no memset.
size_t generic_repair_bytes() { return 1024; }

int generic_repair_launch(const GramProblem &p, const unsigned char *flags, void *ws, bool sym, const TileMap &tm, int tile_rows)
{
    (void)ws;
    GenericPlan pl;
    int rc = make_plan(p.A, p.B, p.T, p.d, p.n, 0, pl, false, true);
    if (rc) return rc;
    GenericArgs a;
    a.X = p.X; a.Y = p.Y; a.grad_out = nullptr; a.K_out = p.K_out;
    a.next_item = nullptr; a.partials = nullptr; a.colslab = nullptr; a.wsk = nullptr;
    a.yx = sym ? 1 : 0;
    a.A = p.A; a.B = p.B; a.T = p.T; a.d = p.d; a.dp = pl.dp; a.n = p.n; a.r = pl.r; a.P = pl.P;
    a.Tm = pl.Tm; a.TmS = pl.TmS; a.nbands = pl.nbands; a.nsteps = pl.nsteps; a.JC = pl.JC;
    a.nchunks = pl.nchunks; a.kind = p.kind; a.naive = 0; a.sym = 0; a.want_grad = 0; a.inv_h = p.inv_h;
    a.inv_r2 = 1.0 / ((double)pl.r * (double)pl.r);
    // one item per row: its B flags are scanned 64 per load, and a row without a flag costs one pass over them
    a.JC = p.B; a.nchunks = 1;
    a.total_items = p.A; a.wsk_per_block = 0; a.big = pl.big;
    if ((long long)pl.grid > a.total_items) pl.grid = (int)a.total_items;
    a.flags = flags; a.tm = tm; a.tile_rows = tile_rows;
#ifdef SIGSVGD_PHASE_STAMPS
    a.stamps = nullptr;
#endif
    hipError_t e = generic_dispatch(p.dtype == SIGSVGD_F64, false, false, pl.big != 0, pl, p.stream, a);
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(generic, fp64 pass)");
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "launch gram_generic_kernel (fp64 pass)");
    return SIGSVGD_OK;
}

} // namespace sigsvgd
