// Backward of the truncated path signature (adjoint of sigsvgd_signature): given dL/dS for the signature of every path,
// dL/dX.  This is what makes `PathSigKernel` differentiable with respect to the PATH, which is how the reference uses it:
// `analytic_grad=False` (src/kernels/_traj_kernels.py:92) routes it to `autograd.grad(k_xx.sum(), x)`
// (src/inference/score.py:50-55, src/inference/svgd.py:41-43), i.e. through `signatory.signature` (:124-125).
//
// One workgroup per path, everything in fp64 in LDS.  The signature of the whole path is recomputed forwards (Chen's
// identity, as signature_kernel does), then the points are taken in REVERSE: with E = exp(D_t) the tensor exponential of
// the increment and S' = S (x) E the update of point t,
//   * the signature before the point comes back from the one after it, S = S' (x) exp(-D_t) (the group inverse: nothing
//     is stored per point);
//   * adjoint of the left factor:   G_j[w]  = sum_{m >= 0} (1/m!) sum_{v in C^m} G'_{j+m}[w v] D^v
//     -- m rounds of "contract the last letter with D";
//   * adjoint of the right factor:  dE_m[v] = sum_{j >= 0} sum_{w in C^j} S_j[w] G'_{j+m}[w v]          (S_0 = 1)
//   * adjoint of the exponential:   dD[a]   = sum_m (1/m!) sum_{r=1..m} sum_{v: v_r = a} dE_m[v] prod_{s != r} D[v_s]
//   * D_t = x_t - x_{t-1}:          dX[t]   = dD_t - dD_{t+1}.
// Every sum has one owner thread and a fixed order: no atomics, reproducible bits.
//
// Round 4: the index arithmetic is TABULATED once per workgroup, as the forward kernel does.  Round 3 decoded every element of
// every level at every point -- two integer divisions by runtime powers of C per Horner step, a while loop and more divisions
// per work unit of the exponential's adjoint, fp64 divisions by (k - r + 1) and m! -- and a point cost ~20,000 cycles whatever
// the signature's size: 0.56 ms for 1024 paths x 64 points x 2 channels at depth 3 against 25 us forwards.  Now a table in LDS
// holds, for element e of level k and Horner step r, the offset of the prefix (a_1 .. a_r) and the letter a_r (the monomial
// pass reads the same table), the reciprocals are multiplied, and each thread decodes its work units of the exponential's
// adjoint once, before the point loop.  Signatures whose table does not fit next to the six working copies keep the decoding
// per point (TAB = false).
#include <type_traits>

#include "sig_common.h"

namespace sigsvgd {

namespace {
constexpr int SB_MAX_DEPTH = 8;

struct SigLevels {
    int pw[SB_MAX_DEPTH + 1];  // C^k
    int off[SB_MAX_DEPTH + 2]; // offset of level k (1-based) in the concatenated signature; off[depth + 1] = total
};

constexpr int SB_UNITS = 4; // work units of the exponential's adjoint a thread can keep decoded (C depth (depth + 1) / 2 <= 4 threads)

// DEPTH > 0: the truncation depth as a compile-time constant -- every loop over levels unrolls and the level tables (C^k and
// the level offsets) are scalar registers; DEPTH = 0: any depth up to SB_MAX_DEPTH, tables in LDS (a load per loop bound).
template <typename T, bool TAB, int DEPTH>
__global__ __launch_bounds__(256) void signature_bwd_kernel(const T *__restrict__ X, const T *__restrict__ gsig, int L, int C,
                                                            int depth_arg, int basepoint, int sigdim, T *__restrict__ gX)
{
    const int depth = DEPTH > 0 ? DEPTH : depth_arg;
    extern __shared__ double sb_lds[];
    double *S = sb_lds, *Sn = S + sigdim, *G = Sn + sigdim, *Gn = G + sigdim, *A0 = Gn + sigdim, *A1 = A0 + sigdim;
    double *inc = A1 + sigdim, *gnext = inc + C, *parts = gnext + C; // parts: [C][depth (depth + 1) / 2]
    // TAB: tab[(r - 1) * sigdim + off[k] + e] = offset of the prefix (a_1 .. a_r) of element e of level k in the concatenated
    // signature, letter a_r in the low byte (r = 1 .. k)
    int *tab = reinterpret_cast<int *>(parts + (size_t)C * depth * (depth + 1) / 2);
    const int tid = threadIdx.x, nt = blockDim.x;
    const T *x = X + (size_t)blockIdx.x * L * C;
    T *gx = gX + (size_t)blockIdx.x * L * C;
    const int U = depth * (depth + 1) / 2;

    // (run-time depth: in LDS -- as private arrays indexed by run-time levels they lived in scratch memory, one scratch load
    //  per use; compile-time depth: private, every index is a constant after unrolling)
    __shared__ SigLevels lv_lds;
    __shared__ double rcp_lds[SB_MAX_DEPTH + 1]; // 1 / n
    SigLevels lv_reg;
    double rcp_reg[SB_MAX_DEPTH + 1];
    lv_reg.pw[0] = 1;
    lv_reg.off[0] = 0;
    lv_reg.off[1] = 0;
    rcp_reg[0] = 1.0;
#pragma unroll
    for (int k = 1; k <= SB_MAX_DEPTH; ++k) {
        lv_reg.pw[k] = lv_reg.pw[k - 1] * C;
        lv_reg.off[k + 1] = lv_reg.off[k] + lv_reg.pw[k];
        rcp_reg[k] = 1.0 / (double)k;
    }
    if (tid == 0) { // (the LDS copy serves every index that is not a constant after unrolling: the work units' (m, r))
        lv_lds = lv_reg;
        for (int n = 0; n <= SB_MAX_DEPTH; ++n) rcp_lds[n] = rcp_reg[n];
    }
    __syncthreads();
    struct Lv {
        const SigLevels &r, &l;
        const double *rr, *rl;
        __device__ __forceinline__ int pw_(int k) const { return (DEPTH > 0 && __builtin_constant_p(k)) ? r.pw[k] : l.pw[k]; }
        __device__ __forceinline__ int off_(int k) const { return (DEPTH > 0 && __builtin_constant_p(k)) ? r.off[k] : l.off[k]; }
        __device__ __forceinline__ double rcp_(int n) const { return (DEPTH > 0 && __builtin_constant_p(n)) ? rr[n] : rl[n]; }
    };
    const Lv lv{lv_reg, lv_lds, rcp_reg, rcp_lds};
    auto load_inc = [&](int t, double sign) { // D_t = x_t - x_{t-1} (x_{-1} = 0: the base point)
        if (tid < C)
            inc[tid] = sign * ((double)x[(size_t)t * C + tid] - (t > 0 ? (double)x[(size_t)(t - 1) * C + tid] : 0.0));
    };
    // dst = src (x) exp(inc): element (a_1 .. a_k) in Horner form, h_r = src_r[a_1..a_r] + h_{r-1} inc[a_r] / (k - r + 1)
    if (TAB) {
#pragma unroll
        for (int k = 1; k <= depth; ++k)
            for (int e = tid; e < lv.pw_(k); e += nt)
#pragma unroll
                for (int r = 1; r <= k; ++r) {
                    const int pr = e / lv.pw_(k - r);
                    tab[(r - 1) * sigdim + lv.off_(k) + e] = ((lv.off_(r) + pr) << 8) | (pr % C);
                }
    }
    auto chen = [&](const double *src, double *dst) {
#pragma unroll
        for (int k = 1; k <= depth; ++k)
            for (int e = tid; e < lv.pw_(k); e += nt) {
                double h = 1.0;
                if (TAB) {
                    const int *tp = tab + lv.off_(k) + e;
#pragma unroll
                    for (int r = 1; r <= k; ++r) {
                        const int code = tp[(r - 1) * sigdim];
                        h = __builtin_fma(h * inc[code & 255], lv.rcp_(k - r + 1), src[code >> 8]);
                    }
                } else {
                    for (int r = 1; r <= k; ++r) {
                        const int pr = e / lv.pw_(k - r);
                        h = __builtin_fma(h * inc[pr % C], lv.rcp_(k - r + 1), src[lv.off_(r) + pr]);
                    }
                }
                dst[lv.off_(k) + e] = h;
            }
    };
    // work units (a, m, r) of the exponential's adjoint, decoded once (TAB: C U <= SB_UNITS * threads)
    int un_a[SB_UNITS], un_m[SB_UNITS], un_r[SB_UNITS], un_phi[SB_UNITS], un_plo[SB_UNITS];
    double un_f[SB_UNITS];
    auto decode_unit = [&](int u, int &a, int &m, int &r, int &phi, int &plo, double &rf) {
        a = u / U;
        int q = u % U;
        m = 1;
        while (q >= m) { // unit q of the triangle -> (m, r)
            q -= m;
            ++m;
        }
        r = q + 1;
        phi = 0;
        plo = 0; // offsets of P_{r-1} and P_{m-r}
        for (int k = 0; k < r - 1; ++k) phi += lv.pw_(k);
        for (int k = 0; k < m - r; ++k) plo += lv.pw_(k);
        double fm = 1.0;
        for (int k = 2; k <= m; ++k) fm *= (double)k;
        rf = 1.0 / fm;
    };
    if (TAB) {
#pragma unroll
        for (int i = 0; i < SB_UNITS; ++i) {
            const int u = tid + i * nt;
            un_a[i] = un_m[i] = un_r[i] = 1, un_phi[i] = un_plo[i] = 0, un_f[i] = 0.0;
            if (u < C * U) decode_unit(u, un_a[i], un_m[i], un_r[i], un_phi[i], un_plo[i], un_f[i]);
        }
    }

    // ---- forward: the signature of the whole path ----------------------------------------------------------------
    for (int e = tid; e < sigdim; e += nt) {
        S[e] = 0.0;
        G[e] = (double)gsig[(size_t)blockIdx.x * sigdim + e];
    }
    if (tid < C) gnext[tid] = 0.0;
    const int t0 = basepoint ? 0 : 1;
    for (int t = t0; t < L; ++t) {
        __syncthreads();
        load_inc(t, 1.0);
        __syncthreads();
        chen(S, Sn);
        double *tmp = S;
        S = Sn;
        Sn = tmp;
    }

    // ---- reverse sweep over the points -------------------------------------------------------------------------------
    for (int t = L - 1; t >= t0; --t) {
        __syncthreads();
        load_inc(t, -1.0);
        __syncthreads();
        chen(S, Sn); // Sn = the signature before point t
        __syncthreads();
        if (tid < C) inc[tid] = -inc[tid]; // back to +D_t
        for (int e = tid; e < sigdim; e += nt) {
            Gn[e] = G[e];
            A0[e] = G[e];
        }
        __syncthreads();
        // adjoint of the left factor: m rounds of contracting the last letter with D
        double fact = 1.0, rfact = 1.0;
        double *a0 = A0, *a1 = A1;
#pragma unroll
        for (int m = 1; m < depth; ++m) {
            fact *= (double)m;
            rfact = 1.0 / fact; // (uniform, once per round)
#pragma unroll
            for (int j = 1; j <= depth - m; ++j)
                for (int w = tid; w < lv.pw_(j); w += nt) {
                    const double *src = a0 + lv.off_(j + 1) + (size_t)w * C;
                    double s = 0.0;
                    for (int a = 0; a < C; ++a) s = __builtin_fma(src[a], inc[a], s);
                    a1[lv.off_(j) + w] = s;
                    Gn[lv.off_(j) + w] += s * rfact;
                }
            __syncthreads();
            double *tmp = a0;
            a0 = a1;
            a1 = tmp;
        }
        // adjoint of the right factor, dE_m[v] (into a0), and the monomials P_k[w] = prod of D over the letters of w (into
        // a1: levels 0 .. depth-1 at offsets 0, 1, 1 + C, ...)
#pragma unroll
        for (int m = 1; m <= depth; ++m)
            for (int v = tid; v < lv.pw_(m); v += nt) {
                double s = G[lv.off_(m) + v];
#pragma unroll
                for (int j = 1; j <= depth - m; ++j) {
                    const double *sp = Sn + lv.off_(j);
                    const double *gp = G + lv.off_(j + m) + v;
                    for (int w = 0; w < lv.pw_(j); ++w) s = __builtin_fma(sp[w], gp[(size_t)w * lv.pw_(m)], s);
                }
                a0[lv.off_(m) + v] = s;
            }
        if (tid == 0) a1[0] = 1.0;
        __syncthreads();
        {
            int poff = 0; // offset of P_{k-1}
#pragma unroll
            for (int k = 1; k < depth; ++k) {
                const int noff = poff + lv.pw_(k - 1);
                for (int e = tid; e < lv.pw_(k); e += nt) {
                    if (TAB) { // prefix (a_1 .. a_{k-1}) and last letter of element e of level k: row k - 2 (offset) / k - 1 (letter)
                        const int last = tab[(k - 1) * sigdim + lv.off_(k) + e] & 255;
                        const int pre = k > 1 ? (tab[(k - 2) * sigdim + lv.off_(k) + e] >> 8) - lv.off_(k - 1) : 0;
                        a1[noff + e] = a1[poff + pre] * inc[last];
                    } else {
                        a1[noff + e] = a1[poff + e / C] * inc[e % C];
                    }
                }
                __syncthreads();
                poff = noff;
            }
        }
        // adjoint of the exponential: work unit (a, m, r) sums over the words with letter a at position r
        auto unit_sum = [&](int a, int m, int r, int phi, int plo, double rf) {
            const double *de = a0 + lv.off_(m);
            double s = 0.0;
            for (int hi = 0; hi < lv.pw_(r - 1); ++hi) {
                const double ph = a1[phi + hi];
                const double *row = de + (size_t)hi * lv.pw_(m - r + 1) + (size_t)a * lv.pw_(m - r);
                double sl = 0.0;
                for (int lo = 0; lo < lv.pw_(m - r); ++lo) sl = __builtin_fma(row[lo], a1[plo + lo], sl);
                s = __builtin_fma(ph, sl, s);
            }
            return s * rf;
        };
        if (TAB) {
#pragma unroll
            for (int i = 0; i < SB_UNITS; ++i) {
                const int u = tid + i * nt;
                if (u < C * U) parts[u] = unit_sum(un_a[i], un_m[i], un_r[i], un_phi[i], un_plo[i], un_f[i]);
            }
        } else {
            for (int u = tid; u < C * U; u += nt) {
                int a, m, r, phi, plo;
                double rf;
                decode_unit(u, a, m, r, phi, plo, rf);
                parts[u] = unit_sum(a, m, r, phi, plo, rf);
            }
        }
        __syncthreads();
        if (tid < C) {
            double gd = 0.0;
            for (int u = 0; u < U; ++u) gd += parts[tid * U + u];
            gx[(size_t)t * C + tid] = (T)(gd - gnext[tid]);
            gnext[tid] = gd;
        }
        // the adjoint and the signature move one point back
        double *tmp = S;
        S = Sn;
        Sn = tmp;
        tmp = G;
        G = Gn;
        Gn = tmp;
    }
    __syncthreads();
    if (!basepoint && tid < C) gx[tid] = (T)(-gnext[tid]); // x_0 enters D_1 only
}

// ---- small signatures: one THREAD per path, everything in registers -----------------------------------------------------
// For the signatures PathSigKernel meets on low-dimensional paths (C = 2 at depth 2-4, C = 3 at depth 2-3, C = 4 .. 6 at depth
// 2: at most 42 channels, 250 registers) the workgroup-per-path kernel above spends its time on ten dependent LDS stages per point with 14 of 64 lanes
// busy: 4.4 us per point, 0.31 ms for 1024 paths x 64 points x 2 channels at depth 3 (25 us forwards).  Here a thread owns a
// path: signature, adjoint and work arrays are private arrays whose every index is a compile-time constant after unrolling
// (C and the depth are template parameters), so they live in registers, there is no barrier and no LDS, and a point is a
// few hundred independent fp64 multiply-adds.  Same formulas, same order of every sum as above.
// (level sizes and offsets as constant tables: a recursive constexpr function in a loop bound stays a call until the outer loop
//  is unrolled, the inner loops are then not unrolled, and the private arrays are indexed at run time -- select chains)
template <int C>
struct SbTab {
    static constexpr int pw[6] = {1, C, C * C, C * C * C, C * C * C * C, C * C * C * C * C};                 // C^k
    static constexpr int off[6] = {0, 0, C, C + C * C, C + C * C + C * C * C, C + C * C + C * C * C + C * C * C * C}; // level k (1-based)
    static constexpr int poff[5] = {0, 1, 1 + C, 1 + C + C * C, 1 + C + C * C + C * C * C};                  // monomials of degree k
};

template <typename T, int C, int DEPTH>
__global__ __launch_bounds__(64) void signature_bwd_small_kernel(const T *__restrict__ X, const T *__restrict__ gsig, int N, int L,
                                                                 int basepoint, T *__restrict__ gX)
{
    using TB = SbTab<C>;
    constexpr int SD = TB::off[DEPTH + 1];
    static_assert(DEPTH <= 4, "tables of SbTab");
    const int path = blockIdx.x * 64 + threadIdx.x;
    if (path >= N) return;
    const T *x = X + (size_t)path * L * C;
    T *gx = gX + (size_t)path * L * C;
    double S[SD], Sn[SD], G[SD], Gn[SD], A0[SD], A1[SD], inc[C], gnext[C];
    double rcp[DEPTH + 1], rfact[DEPTH + 1];
    rcp[0] = 1.0;
    rfact[0] = 1.0;
#pragma unroll
    for (int k = 1; k <= DEPTH; ++k) {
        rcp[k] = 1.0 / (double)k;
        rfact[k] = rfact[k - 1] * rcp[k];
    }
    auto load_inc = [&](int t, double sign) { // D_t = x_t - x_{t-1} (x_{-1} = 0: the base point)
#pragma unroll
        for (int a = 0; a < C; ++a)
            inc[a] = sign * ((double)x[(size_t)t * C + a] - (t > 0 ? (double)x[(size_t)(t - 1) * C + a] : 0.0));
    };
    // dst = src (x) exp(inc): element (a_1 .. a_k) in Horner form, h_r = src_r[a_1..a_r] + h_{r-1} inc[a_r] / (k - r + 1)
    auto chen = [&](const double (&src)[SD], double (&dst)[SD]) {
#pragma unroll
        for (int k = 1; k <= DEPTH; ++k)
#pragma unroll
            for (int e = 0; e < TB::pw[k]; ++e) {
                double h = 1.0;
#pragma unroll
                for (int r = 1; r <= k; ++r) {
                    const int pr = e / TB::pw[k - r];
                    h = __builtin_fma(h * inc[pr % C], rcp[k - r + 1], src[TB::off[r] + pr]);
                }
                dst[TB::off[k] + e] = h;
            }
    };
#pragma unroll
    for (int e = 0; e < SD; ++e) {
        S[e] = 0.0;
        G[e] = (double)gsig[(size_t)path * SD + e];
    }
#pragma unroll
    for (int a = 0; a < C; ++a) gnext[a] = 0.0;
    const int t0 = basepoint ? 0 : 1;
    // ---- forward: the signature of the whole path ----
    for (int t = t0; t < L; ++t) {
        load_inc(t, 1.0);
        chen(S, Sn);
#pragma unroll
        for (int e = 0; e < SD; ++e) S[e] = Sn[e];
    }
    // ---- reverse sweep over the points ----
    for (int t = L - 1; t >= t0; --t) {
        load_inc(t, -1.0);
        chen(S, Sn); // Sn = the signature before point t
#pragma unroll
        for (int a = 0; a < C; ++a) inc[a] = -inc[a]; // back to +D_t
#pragma unroll
        for (int e = 0; e < SD; ++e) {
            Gn[e] = G[e];
            A0[e] = G[e];
        }
        // adjoint of the left factor: m rounds of contracting the last letter with D (A0 -> A1 -> A0 ...)
        auto contract = [&](const double (&src)[SD], double (&dst)[SD], int m) {
#pragma unroll
            for (int j = 1; j <= DEPTH - m; ++j)
#pragma unroll
                for (int w = 0; w < TB::pw[j]; ++w) {
                    double s = 0.0;
#pragma unroll
                    for (int a = 0; a < C; ++a) s = __builtin_fma(src[TB::off[j + 1] + w * C + a], inc[a], s);
                    dst[TB::off[j] + w] = s;
                    Gn[TB::off[j] + w] += s * rfact[m];
                }
        };
#pragma unroll
        for (int m = 1; m < DEPTH; ++m) {
            if (m & 1)
                contract(A0, A1, m);
            else
                contract(A1, A0, m);
        }
        // adjoint of the right factor dE_m[v] (into dE) and the monomials P_k[w] = prod of D over the letters of w (into Pm)
        double dE[SD], Pm[TB::poff[DEPTH]];
#pragma unroll
        for (int m = 1; m <= DEPTH; ++m)
#pragma unroll
            for (int v = 0; v < TB::pw[m]; ++v) {
                double s = G[TB::off[m] + v];
#pragma unroll
                for (int j = 1; j <= DEPTH - m; ++j)
#pragma unroll
                    for (int w = 0; w < TB::pw[j]; ++w)
                        s = __builtin_fma(Sn[TB::off[j] + w], G[TB::off[j + m] + w * TB::pw[m] + v], s);
                dE[TB::off[m] + v] = s;
            }
        Pm[0] = 1.0;
#pragma unroll
        for (int k = 1; k < DEPTH; ++k)
#pragma unroll
            for (int e = 0; e < TB::pw[k]; ++e) Pm[TB::poff[k] + e] = Pm[TB::poff[k - 1] + e / C] * inc[e % C];
        // adjoint of the exponential: unit (a, m, r) sums over the words with letter a at position r; units in the order of the
        // workgroup kernel (a major, then m, then r), summed per channel in that order
#pragma unroll
        for (int a = 0; a < C; ++a) {
            double gd = 0.0;
#pragma unroll
            for (int m = 1; m <= DEPTH; ++m)
#pragma unroll
                for (int r = 1; r <= m; ++r) {
                    double s = 0.0;
#pragma unroll
                    for (int hi = 0; hi < TB::pw[r - 1]; ++hi) {
                        double sl = 0.0;
#pragma unroll
                        for (int lo = 0; lo < TB::pw[m - r]; ++lo)
                            sl = __builtin_fma(dE[TB::off[m] + hi * TB::pw[m - r + 1] + a * TB::pw[m - r] + lo],
                                               Pm[TB::poff[m - r] + lo], sl);
                        s = __builtin_fma(Pm[TB::poff[r - 1] + hi], sl, s);
                    }
                    gd += s * rfact[m];
                }
            gx[(size_t)t * C + a] = (T)(gd - gnext[a]);
            gnext[a] = gd;
        }
#pragma unroll
        for (int e = 0; e < SD; ++e) {
            S[e] = Sn[e];
            G[e] = Gn[e];
        }
    }
    if (!basepoint) {
#pragma unroll
        for (int a = 0; a < C; ++a) gx[a] = (T)(-gnext[a]); // x_0 enters D_1 only
    }
}
} // namespace

int signature_bwd_launch(const void *X, const void *gsig, int N, int L, int C, int depth, int basepoint, int dtype, void *gX,
                         long long sigdim, hipStream_t stream)
{
    if (depth > SB_MAX_DEPTH || C > 255) {
        set_error("signature_backward: depth %d > %d or C=%d > 255", depth, SB_MAX_DEPTH, C);
        return SIGSVGD_E_UNSUPPORTED;
    }
    const size_t lds0 = ((size_t)6 * sigdim + 2 * (size_t)C + (size_t)C * depth * (depth + 1) / 2) * sizeof(double);
    if (sigdim < 0 || lds0 > 150 * 1024) {
        set_error("signature_backward: %lld channels (C=%d, depth=%d) need %zu B of LDS, more than the 150 KB this kernel uses",
                  sigdim, C, depth, lds0);
        return SIGSVGD_E_UNSUPPORTED;
    }
    if (!basepoint && L < 2) { // no increment at all: the signature is constant
        hipError_t e0 = hipMemsetAsync(gX, 0, (size_t)N * L * C * (dtype == SIGSVGD_F64 ? 8 : 4), stream);
        return e0 == hipSuccess ? SIGSVGD_OK : hip_fail(e0, "hipMemsetAsync(signature_backward)");
    }
    { // small signatures: one thread per path, registers only
        hipError_t es = hipSuccess;
        bool done = true;
        auto small = [&](auto kern, auto *Xp, auto *gp, auto *op) {
            hipLaunchKernelGGL(kern, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, stream, Xp, gp, N, L, basepoint, op);
            es = hipGetLastError();
        };
        auto pick = [&](auto *Xp, auto *gp, auto *op) {
            using TT = std::remove_cv_t<std::remove_pointer_t<decltype(op)>>;
            if (C == 2 && depth == 2) small(&signature_bwd_small_kernel<TT, 2, 2>, Xp, gp, op);
            else if (C == 2 && depth == 3) small(&signature_bwd_small_kernel<TT, 2, 3>, Xp, gp, op);
            else if (C == 3 && depth == 2) small(&signature_bwd_small_kernel<TT, 3, 2>, Xp, gp, op);
            else if (C == 4 && depth == 2) small(&signature_bwd_small_kernel<TT, 4, 2>, Xp, gp, op);
            else if (C == 2 && depth == 4) small(&signature_bwd_small_kernel<TT, 2, 4>, Xp, gp, op);
            else if (C == 5 && depth == 2) small(&signature_bwd_small_kernel<TT, 5, 2>, Xp, gp, op);
            else if (C == 3 && depth == 3) small(&signature_bwd_small_kernel<TT, 3, 3>, Xp, gp, op);
            else if (C == 6 && depth == 2) small(&signature_bwd_small_kernel<TT, 6, 2>, Xp, gp, op);
            else done = false;
        };
        if (dtype == SIGSVGD_F64)
            pick(static_cast<const double *>(X), static_cast<const double *>(gsig), static_cast<double *>(gX));
        else
            pick(static_cast<const float *>(X), static_cast<const float *>(gsig), static_cast<float *>(gX));
        if (done) return es == hipSuccess ? SIGSVGD_OK : hip_fail(es, "launch signature_bwd_small_kernel");
    }
    const int threads = sigdim <= 64 ? 64 : (sigdim <= 128 ? 128 : 256);
    // the index table: depth rows of sigdim ints behind the working copies; offsets must fit 23 bits next to the letter byte
    const size_t ldst = lds0 + (size_t)depth * sigdim * sizeof(int);
    const bool tab = ldst <= 150 * 1024 && sigdim < (1 << 23) && C * depth * (depth + 1) / 2 <= SB_UNITS * threads;
    const size_t lds = tab ? ldst : lds0;
    hipError_t e = hipSuccess;
    auto launch = [&](auto kern, auto *Xp, auto *gp, auto *op) {
        if (lds > 64 * 1024) {
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return;
        }
        hipLaunchKernelGGL(kern, dim3(N), dim3(threads), lds, stream, Xp, gp, L, C, depth, basepoint, (int)sigdim, op);
    };
    auto dispatch = [&](auto *Xp, auto *gp, auto *op) {
        using TT = std::remove_cv_t<std::remove_pointer_t<decltype(op)>>;
        if (!tab) return launch(&signature_bwd_kernel<TT, false, 0>, Xp, gp, op);
        switch (depth) { // (the depths the reference's kernels use, unrolled; others through the run-time form)
        case 2: return launch(&signature_bwd_kernel<TT, true, 2>, Xp, gp, op);
        case 3: return launch(&signature_bwd_kernel<TT, true, 3>, Xp, gp, op);
        case 4: return launch(&signature_bwd_kernel<TT, true, 4>, Xp, gp, op);
        default: return launch(&signature_bwd_kernel<TT, true, 0>, Xp, gp, op);
        }
    };
    if (dtype == SIGSVGD_F64)
        dispatch(static_cast<const double *>(X), static_cast<const double *>(gsig), static_cast<double *>(gX));
    else
        dispatch(static_cast<const float *>(X), static_cast<const float *>(gsig), static_cast<float *>(gX));
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(signature_bwd_kernel)");
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "launch signature_bwd_kernel");
    return SIGSVGD_OK;
}

} // namespace sigsvgd
