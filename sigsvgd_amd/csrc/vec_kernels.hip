// Vector ("particle") kernels with analytic first-slot gradients, and the truncated path signature:
// the rows SURVEY.md §8(f) ranks next to the signature-kernel path.
//
//   reference src/kernels/_kernels.py:64-299  GaussianKernel / ScaledGaussianKernel / IMQKernel / ScaledIMQKernel
//             src/utils/math.py:69-86,116-144 pw_dist_sq / scaled_pw_dist_sq
//             src/kernels/_traj_kernels.py:72-144  PathSigKernel = static kernel on signatory.signature(X, depth, basepoint=True)
//
// The reference builds the [A, B, D] difference tensor and reduces it; here the same sums are formed
// tile by tile in registers:
//   vec_sqdist_kernel   sq[i,j] = max(0, sum_c (xm_ic - ym_jc) (x_ic - y_jc))         (xm = x M, ym = y M; M = I: |x - y|^2)
//   vec_kgrad_kernel    K[i,j] = f(sq[i,j]);   dK[i,c] = s * sum_j go[i,j] w(sq[i,j]) (xm_ic - ym_jc)
//                       Gaussian: f = exp(-sq/(2h^2)), w = f;  IMQ: f = (1 + sq/(2h^2))^(-1/2), w = f^3
// Differences are taken directly (no |x|^2 + |y|^2 - 2xy expansion), so there is no cancellation; the
// arithmetic type is the I/O type (fp32 or fp64), like the reference's.  Two launches with sq in HBM
// between them because the default bandwidth is the median of sq (src/utils/math.py:28-34).
#include "sig_common.h"

namespace sigsvgd {

namespace {

constexpr int VT = 64;      // output tile edge of vec_sqdist_kernel
constexpr int VC = 16;      // channels staged per step
constexpr int VS = VT + 4;  // LDS row stride (elements): 16-B aligned rows, 2-way conflicts at worst

template <typename T, bool METRIC>
__global__ __launch_bounds__(256) void vec_sqdist_kernel(const T *__restrict__ X, const T *__restrict__ Y,
                                                         const T *__restrict__ XM, const T *__restrict__ YM, int A,
                                                         int B, int D, T *__restrict__ sq)
{
    __shared__ __align__(16) T xs[VC * VS], ys[VC * VS];
    __shared__ __align__(16) T xms[METRIC ? VC * VS : 4], yms[METRIC ? VC * VS : 4];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int row0 = blockIdx.y * VT, col0 = blockIdx.x * VT;
    T acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = T(0);

    for (int c0 = 0; c0 < D; c0 += VC) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + u * 256, r = e / VC, c = e % VC;
            const bool cok = c0 + c < D;
            const bool xok = cok && row0 + r < A, yok = cok && col0 + r < B;
            const size_t xi = (size_t)(row0 + r) * D + c0 + c, yi = (size_t)(col0 + r) * D + c0 + c;
            xs[c * VS + r] = xok ? X[xi] : T(0);
            ys[c * VS + r] = yok ? Y[yi] : T(0);
            if (METRIC) {
                xms[c * VS + r] = xok ? XM[xi] : T(0);
                yms[c * VS + r] = yok ? YM[yi] : T(0);
            }
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < VC; ++c) {
            T xv[4], yv[4], xmv[4], ymv[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                xv[a] = xs[c * VS + ty * 4 + a];
                yv[a] = ys[c * VS + tx * 4 + a];
                if (METRIC) {
                    xmv[a] = xms[c * VS + ty * 4 + a];
                    ymv[a] = yms[c * VS + tx * 4 + a];
                }
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const T dx = xv[a] - yv[b];
                    const T dm = METRIC ? xmv[a] - ymv[b] : dx;
                    acc[a][b] += dx * dm;
                }
        }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int gr = row0 + ty * 4 + a, gc = col0 + tx * 4 + b;
            if (gr < A && gc < B) sq[(size_t)gr * B + gc] = acc[a][b] > T(0) ? acc[a][b] : T(0); // clamp(min=0)
        }
}

constexpr int SIG_MAX_DEPTH = 16;
__device__ const double RECIP[SIG_MAX_DEPTH + 1] = {0.0, 1.0, 1.0 / 2, 1.0 / 3, 1.0 / 4, 1.0 / 5, 1.0 / 6, 1.0 / 7, 1.0 / 8,
                                                    1.0 / 9, 1.0 / 10, 1.0 / 11, 1.0 / 12, 1.0 / 13, 1.0 / 14, 1.0 / 15, 1.0 / 16};

constexpr int GR = 16; // rows per workgroup of vec_kgrad_kernel
constexpr int GJ = 64; // partners staged per step
constexpr int GC = 64; // channels per workgroup (4 per thread)

template <typename T>
__device__ __forceinline__ void kernel_fn(int kind, T sq, T half_inv_h2, T &k, T &w)
{
    if (kind == SIGSVGD_VEC_GAUSSIAN) {
        k = exp(-half_inv_h2 * sq);
        w = k;
    } else if (kind == SIGSVGD_VEC_IMQ) {
        const T den = T(1) + half_inv_h2 * sq;
        k = T(1) / sqrt(den);
        w = k / den; // den^(-3/2)
    } else { // SIGSVGD_VEC_UNIT: plain weighted differences (backward of vec_sqdist)
        k = sq;
        w = T(1);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void vec_kgrad_kernel(const T *__restrict__ sq, const T *__restrict__ XM,
                                                        const T *__restrict__ YM, const T *__restrict__ go, int A,
                                                        int B, int D, int kind, T half_inv_h2, T grad_scale,
                                                        T *__restrict__ Kout, T *__restrict__ dK)
{
    __shared__ __align__(16) T wt[GR * (GJ + 1)];
    __shared__ __align__(16) T yt[GJ * (GC + 4)];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int row0 = blockIdx.y * GR, ch0 = blockIdx.x * GC;
    const bool want_grad = dK != nullptr;
    const int gi = row0 + ty;
    T xm[4], acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int gc = ch0 + tx * 4 + c;
        xm[c] = (want_grad && gi < A && gc < D) ? XM[(size_t)gi * D + gc] : T(0);
        acc[c] = T(0);
    }
    for (int j0 = 0; j0 < B; j0 += GJ) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) { // weights of this 16 x 64 block; K is written by the first channel block
            const int e = tid + u * 256, r = e / GJ, jj = e % GJ;
            const int i = row0 + r, j = j0 + jj;
            T k = T(0), w = T(0);
            if (i < A && j < B) {
                const size_t idx = (size_t)i * B + j;
                kernel_fn<T>(kind, sq[idx], half_inv_h2, k, w);
                if (blockIdx.x == 0 && Kout) Kout[idx] = k;
                if (go) w *= go[idx];
            }
            wt[r * (GJ + 1) + jj] = w;
        }
        if (want_grad) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int e = tid + u * 256, jj = e / GC, c = e % GC;
                const int j = j0 + jj, gc = ch0 + c;
                yt[jj * (GC + 4) + c] = (j < B && gc < D) ? YM[(size_t)j * D + gc] : T(0);
            }
        }
        __syncthreads();
        if (want_grad) {
#pragma unroll 8
            for (int jj = 0; jj < GJ; ++jj) {
                const T w = wt[ty * (GJ + 1) + jj]; // zero for partners beyond B
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] += w * (xm[c] - yt[jj * (GC + 4) + tx * 4 + c]);
            }
        }
    }
    if (want_grad && gi < A) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int gc = ch0 + tx * 4 + c;
            if (gc < D) dK[(size_t)gi * D + gc] = grad_scale * acc[c];
        }
    }
}

// Truncated signature by Chen's identity, one workgroup per path.  Level k of the signature after
// appending the increment D is  S_k + sum_{m=1..k} S_{k-m} (x) D^{(x)m} / m!,  evaluated per element
// (a_1 .. a_k) in Horner form:  h_0 = 1,  h_r = S_r[a_1..a_r] + h_{r-1} * D[a_r] / (k - r + 1).
// Accumulated in fp64 in LDS (ping-pong buffers, one barrier per point).  The prefix positions and letters
// of every element do not depend on the point, so they are tabulated once ((position << 8) | letter, one
// int per element and Horner step): the integer divisions that produce them cost more than the rest.
template <typename T>
__global__ __launch_bounds__(256) void signature_kernel(const T *__restrict__ X, int L, int C, int depth,
                                                        int basepoint, int sigdim, int staged, T *__restrict__ out)
{
    extern __shared__ double sig_lds[]; // 2 * sigdim doubles, increments ([L][C] if staged, else [C]), table
    double *buf0 = sig_lds, *buf1 = sig_lds + sigdim, *incs = sig_lds + 2 * sigdim;
    int *tab = reinterpret_cast<int *>(incs + (staged ? (size_t)L * C : (size_t)C)); // [sigdim][depth]
    const int tid = threadIdx.x, nt = blockDim.x;
    const T *x = X + (size_t)blockIdx.x * L * C;
    for (int e = tid; e < sigdim; e += nt) buf0[e] = 0.0;
    if (staged) // the whole path in one coalesced pass instead of a global load (and its latency) per point
        for (int e = tid; e < L * C; e += nt) incs[e] = (double)x[e] - (e >= C ? (double)x[e - C] : 0.0);
    {
        int off = 0, len = C; // level k occupies [off, off + C^k)
        for (int k = 1; k <= depth; ++k) {
            for (int e = tid; e < len; e += nt) {
                int div = len / C, loff = 0, llen = C; // C^(k-1); offset / length of level r
                for (int r = 1; r <= k; ++r) {
                    const int pr = e / div; // a_1..a_r as a level-r flat index
                    tab[(off + e) * depth + (r - 1)] = ((loff + pr) << 8) | (pr % C);
                    loff += llen;
                    llen *= C;
                    div = div > 1 ? div / C : 1;
                }
            }
            off += len;
            len *= C;
        }
    }
    __syncthreads();
    // this thread's first element (all of them when sigdim <= blockDim): level and Horner table in registers
    const bool small = depth <= 4 && sigdim <= nt;
    int myk = 0, pk[4] = {0, 0, 0, 0};
    double rc[4] = {0.0, 0.0, 0.0, 0.0};
    if (small && tid < sigdim) {
        int off = 0, len = C;
        for (int k = 1; k <= depth; ++k) {
            if (tid >= off && tid < off + len) myk = k;
            off += len;
            len *= C;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r < myk) {
                pk[r] = tab[tid * depth + r];
                rc[r] = RECIP[myk - r];
            }
    }
    double *cur = buf0, *nxt = buf1;
    for (int t = basepoint ? 0 : 1; t < L; ++t) {
        __syncthreads();
        const double *inc = incs;
        if (staged) {
            inc = incs + (size_t)t * C;
        } else {
            if (tid < C)
                incs[tid] = (double)x[(size_t)t * C + tid] - (t > 0 ? (double)x[(size_t)(t - 1) * C + tid] : 0.0);
            __syncthreads();
        }
        if (small) { // one element per thread, every level in the same pass
            if (tid < sigdim) {
                double h = 1.0;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (r < myk) h = cur[pk[r] >> 8] + h * inc[pk[r] & 255] * rc[r];
                nxt[tid] = h;
            }
        } else {
            int off = 0, len = C;
            for (int k = 1; k <= depth; ++k) {
                for (int e = tid; e < len; e += nt) {
                    const int *tb = tab + (off + e) * depth;
                    double h = 1.0;
                    for (int r = 1; r <= k; ++r) {
                        const int pkk = tb[r - 1];
                        h = cur[pkk >> 8] + h * inc[pkk & 255] * RECIP[k - r + 1];
                    }
                    nxt[off + e] = h;
                }
                off += len;
                len *= C;
            }
        }
        double *tmp = cur;
        cur = nxt;
        nxt = tmp;
    }
    __syncthreads();
    for (int e = tid; e < sigdim; e += nt) out[(size_t)blockIdx.x * sigdim + e] = (T)cur[e];
}

template <typename T>
int vec_sqdist_t(const void *X, const void *Y, const void *XM, const void *YM, int A, int B, int D, void *sq,
                 hipStream_t stream)
{
    dim3 grid((B + VT - 1) / VT, (A + VT - 1) / VT), block(256);
    if (XM)
        hipLaunchKernelGGL((vec_sqdist_kernel<T, true>), grid, block, 0, stream, static_cast<const T *>(X),
                           static_cast<const T *>(Y), static_cast<const T *>(XM), static_cast<const T *>(YM), A, B, D,
                           static_cast<T *>(sq));
    else
        hipLaunchKernelGGL((vec_sqdist_kernel<T, false>), grid, block, 0, stream, static_cast<const T *>(X),
                           static_cast<const T *>(Y), static_cast<const T *>(nullptr), static_cast<const T *>(nullptr),
                           A, B, D, static_cast<T *>(sq));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "launch vec_sqdist_kernel");
    return SIGSVGD_OK;
}

template <typename T>
int vec_kgrad_t(const void *sq, const void *XM, const void *YM, const void *go, int A, int B, int D, int kind,
                double inv_h2, double grad_scale, void *K, void *dK, hipStream_t stream)
{
    dim3 grid(dK ? (D + GC - 1) / GC : 1, (A + GR - 1) / GR), block(256);
    hipLaunchKernelGGL((vec_kgrad_kernel<T>), grid, block, 0, stream, static_cast<const T *>(sq),
                       static_cast<const T *>(XM), static_cast<const T *>(YM), static_cast<const T *>(go), A, B, D, kind,
                       (T)(0.5 * inv_h2), (T)grad_scale, static_cast<T *>(K), static_cast<T *>(dK));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "launch vec_kgrad_kernel");
    return SIGSVGD_OK;
}

} // namespace

int vec_sqdist_launch(const void *X, const void *Y, const void *XM, const void *YM, int A, int B, int D, int dtype,
                      void *sq, hipStream_t stream)
{
    return dtype == SIGSVGD_F64 ? vec_sqdist_t<double>(X, Y, XM, YM, A, B, D, sq, stream)
                                : vec_sqdist_t<float>(X, Y, XM, YM, A, B, D, sq, stream);
}

int vec_kgrad_launch(const void *sq, const void *XM, const void *YM, const void *go, int A, int B, int D, int dtype,
                     int kind, double inv_h2, double grad_scale, void *K, void *dK, hipStream_t stream)
{
    return dtype == SIGSVGD_F64 ? vec_kgrad_t<double>(sq, XM, YM, go, A, B, D, kind, inv_h2, grad_scale, K, dK, stream)
                                : vec_kgrad_t<float>(sq, XM, YM, go, A, B, D, kind, inv_h2, grad_scale, K, dK, stream);
}

long long signature_channels(int C, int depth)
{
    long long total = 0, len = 1;
    for (int k = 1; k <= depth; ++k) {
        len *= C;
        total += len;
        if (total > (1LL << 30)) return -1;
    }
    return total;
}

int signature_launch(const void *X, int N, int L, int C, int depth, int basepoint, int dtype, void *out,
                     hipStream_t stream)
{
    const long long sigdim = signature_channels(C, depth);
    if (depth > SIG_MAX_DEPTH || C > 255) {
        set_error("signature: depth %d > %d or C=%d > 255", depth, SIG_MAX_DEPTH, C);
        return SIGSVGD_E_UNSUPPORTED;
    }
    const size_t tab_bytes = sigdim > 0 ? (size_t)sigdim * depth * sizeof(int) : 0;
    const size_t lds_min = (size_t)(2 * sigdim + C) * sizeof(double) + tab_bytes;
    if (sigdim < 0 || lds_min > 150 * 1024) {
        set_error("signature: %lld channels (C=%d, depth=%d) need %zu B of LDS, more than the 150 KB this kernel uses",
                  sigdim, C, depth, lds_min);
        return SIGSVGD_E_UNSUPPORTED;
    }
    // stage the whole path in LDS when it fits next to the two signature buffers
    const size_t lds_staged = (size_t)(2 * sigdim + (long long)L * C) * sizeof(double) + tab_bytes;
    const int staged = lds_staged <= 150 * 1024;
    const size_t lds = staged ? lds_staged : lds_min;
    // one wavefront per path while the signature has no more elements than lanes (PathSigKernel's own
    // use: C = 2, depth 3 -> 14): the per-point barriers of a single-wave workgroup cost nothing
    const int threads = sigdim <= 64 ? 64 : (sigdim <= 128 ? 128 : 256);
    hipError_t e;
    if (dtype == SIGSVGD_F64) {
        if (lds > 64 * 1024) {
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(&signature_kernel<double>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(signature_kernel)");
        }
        hipLaunchKernelGGL((signature_kernel<double>), dim3(N), dim3(threads), lds, stream, static_cast<const double *>(X),
                           L, C, depth, basepoint, (int)sigdim, staged, static_cast<double *>(out));
    } else {
        if (lds > 64 * 1024) {
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(&signature_kernel<float>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(signature_kernel)");
        }
        hipLaunchKernelGGL((signature_kernel<float>), dim3(N), dim3(threads), lds, stream, static_cast<const float *>(X), L,
                           C, depth, basepoint, (int)sigdim, staged, static_cast<float *>(out));
    }
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "launch signature_kernel");
    return SIGSVGD_OK;
}

} // namespace sigsvgd
