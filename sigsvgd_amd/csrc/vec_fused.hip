// Fused vector kernel with analytic gradient on the fp32 matrix cores, for a GIVEN bandwidth:
//   K[i,j] = f(sq[i,j]),   sq[i,j] = sum_c (xm_ic - ym_jc)(x_ic - y_jc)                 (xm = x M, ym = y M; M = I: |x - y|^2)
//   dK[i,c] = s * sum_j go[i,j] w(sq[i,j]) (xm_ic - ym_jc)
//   reference src/kernels/_kernels.py:64-299 (Gaussian / IMQ and their scaled variants), src/utils/math.py:69-86,116-144
// The two-launch path of vec_kernels.hip writes sq[A,B] to HBM and reads it back (the bandwidth may be the median of
// sq), and forms both sums on the vector ALUs.  When h is known this kernel does everything in one pass with both
// GEMM-shaped sums on v_mfma_f32_16x16x4_f32 -- the structure of a fused attention kernel without the softmax
// normalisation:
//   S = XM~ Y~^T (+ X~ YM~^T with a metric)     over the channel dimension, 64 x 64 tile per workgroup and column tile
//   sq = a_i + b_j - S,  K = f(sq) -> HBM,  W = go * w(sq)                     elementwise on the accumulators
//   O += W YM~,  wsum += row sums of W          W goes through LDS into the A-operand layout
//   dK = s * (xm~_i * wsum_i - O_i)             (xm_i - ym_j = xm~_i - ym~_j)
// ~ : every operand is centred on the first row of Y / YM (differences are unchanged), so the expansion
// |x|^2 + |y|^2 - 2 x.y loses bits relative to the spread of the particles, not to their distance from the origin
// (the reference uses the uncentred expansion for the unscaled kernels, math.py:69-86).  a_i, b_j are accumulated
// from the staged tiles.  The columns are split over blockIdx.x (partial O and wsum are added with fp32 atomics into
// the zeroed dK -- or, given a workspace, write them to per-split blocks that vec_join_kernel adds in split order: reproducible
// bits), so a launch has (A/64) x splits workgroups.  fp32 only, D <= 512.
#include "sig_common.h"

namespace sigsvgd {
namespace {
using ff32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int FM = 64;        // rows per workgroup (16 per wavefront)
constexpr int FN = 64;        // columns per tile
constexpr int FK_METRIC = 32; // channels per S stage with a metric (four staged tiles) ...
constexpr int FK_PLAIN = 64;  // ... and without (two): half the stages, half the barriers
constexpr int FC = 64;        // channels per O stage
constexpr int FCS = FC + 4;   // LDS row stride of the W tiles (= 4 mod 32: the 16 rows x 4 columns of an A read are conflict-free)
constexpr int FYS = FC + 16;  // LDS row stride of the YM~ tile (= 16 mod 32: the two k rows a 32-lane B read touches are disjoint)

template <typename T>
__device__ __forceinline__ void f_kernel_fn(int kind, T sq, T half_inv_h2, T &k, T &w)
{
    if (kind == SIGSVGD_VEC_GAUSSIAN) {
        k = __expf(-half_inv_h2 * sq);
        w = k;
    } else if (kind == SIGSVGD_VEC_IMQ) {
        const T den = T(1) + half_inv_h2 * sq;
        k = rsqrtf(den);
        w = k / den; // den^(-3/2)
    } else { // SIGSVGD_VEC_UNIT: plain weighted differences (backward of the distance)
        k = sq;
        w = T(1);
    }
}

// NT: 16-channel accumulator tiles of the O product (D <= 16 * NT).  VEC4: D % 4 == 0 and 16-B aligned matrices
// (global loads are float4).  Every stage is fetched into registers while the previous one is multiplied.
// Round 4: EIGHT wavefronts per workgroup.  Round 2-3's four left one wavefront per SIMD at the shapes that matter (N = 1024:
// 16 row tiles x 16 column splits = 256 workgroups) and 59 % of the wave time waiting (rocprofv3: MFMA pipe busy 15 %).  The two
// halves of the workgroup share the staged tiles and split the MFMA work of every stage: columns 0-31 / 32-63 of the S tile
// (and its elementwise part), channel tiles 0-1 / 2-3 of every O stage -- nothing to combine but the row sums of W.
template <int NT, bool METRIC, bool VEC4>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void vec_fused_kernel(const float *__restrict__ X, const float *__restrict__ Y,
                                                        const float *__restrict__ XM, const float *__restrict__ YM,
                                                        const float *__restrict__ go, int A, int B, int D, int kind,
                                                        float half_inv_h2, float grad_scale, int tiles_per_split,
                                                        float *__restrict__ Kout, float *__restrict__ dK,
                                                        float *__restrict__ part)
{
    constexpr int FK = METRIC ? FK_METRIC : FK_PLAIN; // channels per S stage
    constexpr int FKS = FK + 4;                       // LDS row stride of the S-stage tiles (floats)
    // (two buffers per staged tile, used alternately: one workgroup barrier per stage -- a stage's writes go to the buffer
    //  that was read two stages ago, and the barrier of the stage in between lies between the two)
    __shared__ __align__(16) float xs_[2][FM * FKS], ys_[2][FN * FKS];             // XM~ / Y~ stage  [row][k]
    __shared__ __align__(16) float xs2_[2][METRIC ? FM * FKS : 4], ys2_[2][METRIC ? FN * FKS : 4]; // X~ / YM~ stage
    __shared__ __align__(16) float wt[4 * 16 * FCS];                        // W tile of each wavefront [row][col]
    __shared__ __align__(16) float ymt_[2][FN * FYS];                       // YM~ stage       [col][channel]
    __shared__ __align__(16) float cyl[16 * NT], cyml[METRIC ? 16 * NT : 4]; // centres: row 0 of Y / YM
    __shared__ float an[FM], bn[FN], wsl[2][FM];
    const int tid = threadIdx.x, lane = tid & 63, wave8 = tid >> 6;
    const int wave = wave8 & 3, hf = wave8 >> 2; // row block of 16, half of the column / channel work
    const int row0 = blockIdx.y * FM;
    const int tile_lo = blockIdx.x * tiles_per_split;
    const int ntile = (B + FN - 1) / FN;
    const int tile_hi = min(ntile, tile_lo + tiles_per_split);
    const float *YMm = METRIC ? YM : Y;   // the matrix whose rows are subtracted in the gradient (ym)
    const float *XMm = METRIC ? XM : X;
    const float *cymp = METRIC ? cyml : cyl;
    // staging maps: S stage 64 rows x 32 channels: thread -> row tid / 8, channels (tid % 8) * 4 .. + 3;
    //               O stage 64 columns x 64 channels: thread -> column tid / 8, channels (tid % 8) * 8 .. + 7
    const int sr = tid >> 3, sk = (tid & 7) * (FK / 8), so = (tid & 7) * 8;
    const int ri = lane & 15, rk = lane >> 4; // MFMA operand indices of this lane

    for (int c = tid; c < 16 * NT; c += 512) {
        cyl[c] = c < D ? Y[c] : 0.f;
        if (METRIC) cyml[c] = c < D ? YM[c] : 0.f;
    }

    // n floats of row `g` of `P` starting at channel c0 (zeros beyond the matrix)
    auto load_row = [&](const float *P, int g, int rows, int c0, float *out, int n) {
        if (g >= rows) {
            for (int u = 0; u < n; ++u) out[u] = 0.f;
            return;
        }
        const float *p = P + (size_t)g * D + c0;
        if (VEC4) {
            for (int u = 0; u < n; u += 4) {
                ff32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (c0 + u < D) v = *reinterpret_cast<const ff32x4 *>(p + u);
                out[u] = v[0]; out[u + 1] = v[1]; out[u + 2] = v[2]; out[u + 3] = v[3];
            }
        } else {
            for (int u = 0; u < n; ++u) out[u] = (c0 + u < D) ? p[u] : 0.f;
        }
    };

    ff32x4 O[NT / 2]; // this half's channel tiles: tiles 2 hf, 2 hf + 1 of every stage of four
#pragma unroll
    for (int n = 0; n < NT / 2; ++n) O[n] = ff32x4{0.f, 0.f, 0.f, 0.f};
    float wsum[4] = {0.f, 0.f, 0.f, 0.f}; // partial row sums of W: rows 16 wave + 4 rk + r, this lane's columns
    float a_part = 0.f;                   // this thread's share of a_i for row sr (first tile only)
    bool have_a = false;
    __syncthreads();

    for (int tile = tile_lo; tile < tile_hi; ++tile) {
        const int col0 = tile * FN;
        ff32x4 S[2]; // this half's column blocks: 2 hf, 2 hf + 1
#pragma unroll
        for (int b = 0; b < 2; ++b) S[b] = ff32x4{0.f, 0.f, 0.f, 0.f};
        float b_part = 0.f;
        // ---- S = XM~ Y~^T (+ X~ YM~^T) over the channels, FK at a time ---------------------------------------------
        constexpr int SE = FK / 8; // elements per thread and matrix in an S stage
        float xv[SE], yv[SE], xv2[METRIC ? SE : 1], yv2[METRIC ? SE : 1];
        auto fetchS = [&](int k0) {
            load_row(XMm, row0 + sr, A, k0 + sk, xv, SE);
            load_row(Y, col0 + sr, B, k0 + sk, yv, SE);
            if (METRIC) {
                load_row(X, row0 + sr, A, k0 + sk, xv2, SE);
                load_row(YM, col0 + sr, B, k0 + sk, yv2, SE);
            }
        };
        fetchS(0);
        int pb = 0;
        for (int k0 = 0; k0 < D; k0 += FK, pb ^= 1) {
            float *xs = xs_[pb], *ys = ys_[pb], *xs2 = xs2_[pb], *ys2 = ys2_[pb];
            const bool xin = row0 + sr < A, yin = col0 + sr < B;
#pragma unroll
            for (int u = 0; u < SE; ++u) {
                const int c = k0 + sk + u;
                const bool ok = c < D;
                const float cy = cyl[ok ? c : 0], cym = cymp[ok ? c : 0];
                const float xm = (ok && xin) ? xv[u] - cym : 0.f;
                const float yy = (ok && yin) ? yv[u] - cy : 0.f;
                float x1 = xm, ym1 = yy;
                xs[sr * FKS + sk + u] = xm;
                ys[sr * FKS + sk + u] = yy;
                if (METRIC) {
                    x1 = (ok && xin) ? xv2[u] - cy : 0.f;
                    ym1 = (ok && yin) ? yv2[u] - cym : 0.f;
                    xs2[sr * FKS + sk + u] = x1;
                    ys2[sr * FKS + sk + u] = ym1;
                }
                if (!have_a) a_part = __builtin_fmaf(xm, x1, a_part);
                b_part = __builtin_fmaf(ym1, yy, b_part);
            }
            __syncthreads();
            if (k0 + FK < D) fetchS(k0 + FK); // in flight during the MFMAs
#pragma unroll
            for (int ks = 0; ks < FK; ks += 4) {
                const float av = xs[(16 * wave + ri) * FKS + ks + rk];
                const float av2 = METRIC ? xs2[(16 * wave + ri) * FKS + ks + rk] : 0.f;
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int cb = 2 * hf + b;
                    S[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, ys[(16 * cb + ri) * FKS + ks + rk], S[b], 0, 0, 0);
                    if (METRIC)
                        S[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av2, ys2[(16 * cb + ri) * FKS + ks + rk], S[b], 0, 0, 0);
                }
            }
        }
        // first YM~ stage of the O product: in flight during the elementwise part
        float yo[8];
        if (dK) load_row(YMm, col0 + sr, B, so, yo, 8);
        // norms: the eight threads of a staging row hold its partial sums
        if (!have_a) {
            a_part += __shfl_xor(a_part, 1, 64);
            a_part += __shfl_xor(a_part, 2, 64);
            a_part += __shfl_xor(a_part, 4, 64);
            if ((tid & 7) == 0) an[sr] = a_part;
            have_a = true;
        }
        b_part += __shfl_xor(b_part, 1, 64);
        b_part += __shfl_xor(b_part, 2, 64);
        b_part += __shfl_xor(b_part, 4, 64);
        __syncthreads(); // (also: the last S stage is consumed)
        if ((tid & 7) == 0) bn[sr] = b_part;
        __syncthreads();
        // ---- elementwise: sq -> K, W; C layout: column 16 b + ri, rows 16 wave + 4 rk + r ---------------------------
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int jl = 16 * (2 * hf + b) + ri, gj = col0 + jl;
            const float bj = bn[jl];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int il = 16 * wave + 4 * rk + r, gi = row0 + il;
                float w = 0.f;
                if (gi < A && gj < B) {
                    const float cross = METRIC ? S[b][r] : 2.f * S[b][r];
                    const float sq = fmaxf(an[il] + bj - cross, 0.f); // clamp(min=0), as the reference
                    float k;
                    f_kernel_fn<float>(kind, sq, half_inv_h2, k, w);
                    const size_t idx = (size_t)gi * B + gj;
                    if (Kout) Kout[idx] = k;
                    if (go) w *= go[idx];
                }
                wsum[r] += w;
                wt[(16 * wave + 4 * rk + r) * FCS + jl] = w;
            }
        }
        if (!dK) continue; // (uniform)
        // ---- O += W YM~ over the channels, FC at a time -------------------------------------------------------------
#pragma unroll
        for (int nb = 0; nb < NT / 4; ++nb) { // (fully unrolled: the accumulator tiles must be indexed statically)
            const int n0 = nb * FC;
            if (n0 >= D) break;
            float *ymt = ymt_[nb & 1];
            if (nb == 0) __syncthreads(); // the W tiles are written (later stages: the other YM~ buffer, see above)
            {
                const bool yin = col0 + sr < B;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = n0 + so + u;
                    ymt[sr * FYS + so + u] = (c < D && yin) ? yo[u] - cymp[c < D ? c : 0] : 0.f;
                }
            }
            __syncthreads();
            if (n0 + FC < D) load_row(YMm, col0 + sr, B, n0 + FC + so, yo, 8); // next stage, in flight during the MFMAs
#pragma unroll
            for (int ks = 0; ks < FN; ks += 4) {
                const float av = wt[(16 * wave + ri) * FCS + ks + rk]; // A[i = ri][k = column ks + rk]
#pragma unroll
                for (int nt = 0; nt < FC / 32; ++nt)
                    O[nb * 2 + nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, ymt[(ks + rk) * FYS + 16 * (2 * hf + nt) + ri],
                                                                          O[nb * 2 + nt], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    if (!dK) return;
    // row sums: add up the 16 lanes that share the rows (ri varies), then the two halves of the workgroup (columns 0-31 / 32-63)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) wsum[r] += __shfl_xor(wsum[r], off, 64);
        if (ri == 0) wsl[hf][16 * wave + 4 * rk + r] = wsum[r];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) wsum[r] = wsl[0][16 * wave + 4 * rk + r] + wsl[1][16 * wave + 4 * rk + r];
    // dK[i, c] += s * (xm~_i[c] * wsum_i - O_i[c]);  O layout: channel 16 n + ri, rows 16 wave + 4 rk + r
#pragma unroll
    for (int n = 0; n < NT / 2; ++n) {
        const int c = 16 * (4 * (n >> 1) + 2 * hf + (n & 1)) + ri; // tile (n & 1) of this half in stage n / 2
        if (c < D) {
            const float cym = cymp[c];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gi = row0 + 16 * wave + 4 * rk + r;
                if (gi < A) {
                    const float xm = XMm[(size_t)gi * D + c] - cym;
                    const float v = grad_scale * (xm * wsum[r] - O[n][r]);
                    // reproducible route: this column split's partial sum goes to its own [A][D] block of the workspace (every
                    // element of the block has exactly one writer) and vec_join_kernel adds the blocks in split order; without a
                    // workspace the splits meet in fp32 atomics on the zeroed output (last bits depend on their order)
                    if (part) part[((size_t)blockIdx.x * A + gi) * D + c] = v;
                    else unsafeAtomicAdd(&dK[(size_t)gi * D + c], v);
                }
            }
        }
    }
}

// dK[e] = sum over the column splits of part[s][e], in split order (fixed: reproducible bits)
__global__ __launch_bounds__(256) void vec_join_kernel(const float *__restrict__ part, int splits, size_t n, float *__restrict__ dK)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += part[(size_t)k * n + e];
    dK[e] = s;
}

// column splits of a launch: enough workgroups for two per compute unit
inline void fused_splits(int A, int B, bool grad, int &splits, int &per)
{
    const int ntile = (B + FN - 1) / FN, rows = (A + FM - 1) / FM;
    splits = grad ? (512 + rows - 1) / rows : ntile;
    if (splits > ntile) splits = ntile;
    if (splits < 1) splits = 1;
    per = (ntile + splits - 1) / splits;
    splits = (ntile + per - 1) / per;
}

template <int NT>
int fused_launch_nt(const float *X, const float *Y, const float *XM, const float *YM, const float *go, int A, int B, int D,
                    int kind, float half_inv_h2, float grad_scale, float *K, float *dK, float *ws, size_t ws_bytes,
                    hipStream_t stream)
{
    const int rows = (A + FM - 1) / FM;
    int splits, per;
    fused_splits(A, B, dK != nullptr, splits, per);
    // the reproducible route needs one [A][D] block per split (none with a single split: it writes dK itself)
    float *part = nullptr;
    if (dK && splits == 1) {
        part = dK;
    } else if (dK && ws) {
        if (ws_bytes < (size_t)splits * A * D * sizeof(float)) {
            set_error("vec_kernel_fused: workspace %zu B < required %zu B", ws_bytes, (size_t)splits * A * D * sizeof(float));
            return SIGSVGD_E_WORKSPACE;
        }
        part = ws;
    } else if (dK) {
        hipError_t e = hipMemsetAsync(dK, 0, (size_t)A * D * sizeof(float), stream);
        if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(dK)");
    }
    dim3 grid(splits, rows), block(512);
    auto al = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    const bool vec4 = (D % 4) == 0 && al(X) && al(Y) && al(XM) && al(YM);
#define SIG_VF_LAUNCH(M, V)                                                                                           \
    hipLaunchKernelGGL((vec_fused_kernel<NT, M, V>), grid, block, 0, stream, X, Y, XM, YM, go, A, B, D, kind, half_inv_h2, \
                       grad_scale, per, K, dK, part)
    if (XM && vec4)
        SIG_VF_LAUNCH(true, true);
    else if (XM)
        SIG_VF_LAUNCH(true, false);
    else if (vec4)
        SIG_VF_LAUNCH(false, true);
    else
        SIG_VF_LAUNCH(false, false);
#undef SIG_VF_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "launch vec_fused_kernel");
    if (part && part != dK) {
        const size_t n = (size_t)A * D;
        hipLaunchKernelGGL(vec_join_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, part, splits, n, dK);
        e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "launch vec_join_kernel");
    }
    return SIGSVGD_OK;
}
} // namespace

size_t vec_fused_workspace_bytes(int A, int B, int D)
{
    int splits, per;
    fused_splits(A, B, true, splits, per);
    return splits > 1 ? (size_t)splits * A * D * sizeof(float) : 0;
}

bool vec_fused_supported(int D, int dtype) { return dtype == SIGSVGD_F32 && D >= 1 && D <= 512; }

int vec_fused_launch(const void *X, const void *Y, const void *XM, const void *YM, const void *go, int A, int B, int D,
                     int kind, double inv_h2, double grad_scale, void *K, void *dK, void *ws, size_t ws_bytes, hipStream_t stream)
{
    float *w = static_cast<float *>(ws);
    const float *x = static_cast<const float *>(X), *y = static_cast<const float *>(Y);
    const float *xm = static_cast<const float *>(XM), *ym = static_cast<const float *>(YM);
    const float *g = static_cast<const float *>(go);
    float *k = static_cast<float *>(K), *dk = static_cast<float *>(dK);
    const float hh = (float)(0.5 * inv_h2), gs = (float)grad_scale;
    if (D <= 64) return fused_launch_nt<4>(x, y, xm, ym, g, A, B, D, kind, hh, gs, k, dk, w, ws_bytes, stream);
    if (D <= 128) return fused_launch_nt<8>(x, y, xm, ym, g, A, B, D, kind, hh, gs, k, dk, w, ws_bytes, stream);
    if (D <= 256) return fused_launch_nt<16>(x, y, xm, ym, g, A, B, D, kind, hh, gs, k, dk, w, ws_bytes, stream);
    return fused_launch_nt<32>(x, y, xm, ym, g, A, B, D, kind, hh, gs, k, dk, w, ws_bytes, stream);
}

} // namespace sigsvgd
