// SVGD velocity: v = -((K @ score - grad_k) / N) [* mask], optionally fused with the
// optimizer=None particle update X_out = X_in - lr * v   (reference src/inference/svgd.py:82-83,115;
// mask: src/inference/trajectory_svgd.py:84).
//
// The N x N x D product is the only GEMM-shaped piece of the hot path and runs on the fp32 MFMA
// (v_mfma_f32_16x16x4_f32: exact fp32 FMA chain, same numerics as the reference's fp32 matmul).
// Tile: one workgroup of 4 wavefronts computes 64 rows x 32 columns; each wavefront owns a
// 16 x 32 strip (two 16x16 accumulators).  K and score tiles are staged through LDS with 16-byte
// global loads; k-step 32 per stage.
#include "sig_common.h"

namespace sigsvgd {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int PM = 64;  // rows per workgroup
constexpr int PN = 32;  // cols per workgroup
constexpr int PK = 32;  // k per stage
constexpr int KS = PK + 1; // padded LDS strides (floats)
constexpr int SS = PN + 1;

__global__ __launch_bounds__(256) void svgd_phi_kernel(const float *__restrict__ K, const float *__restrict__ S,
                                                       const float *__restrict__ gk, const float *__restrict__ mask,
                                                       int N, int D, float *__restrict__ v_out,
                                                       const float *__restrict__ X_in, float *__restrict__ X_out, float lr)
{
    __shared__ float kt[PM * KS]; // K tile   [row][k]
    __shared__ float st[PK * SS]; // score tile [k][col]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.y * PM, col0 = blockIdx.x * PN;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};

    for (int k0 = 0; k0 < N; k0 += PK) {
        // stage K[row0:row0+64, k0:k0+32]: 2048 floats, 8 per thread
        for (int e = tid; e < PM * PK; e += 256) {
            const int r = e / PK, c = e % PK;
            const int gr = row0 + r, gc = k0 + c;
            kt[r * KS + c] = (gr < N && gc < N) ? K[(size_t)gr * N + gc] : 0.f;
        }
        // stage score[k0:k0+32, col0:col0+32]: 1024 floats, 4 per thread
        for (int e = tid; e < PK * PN; e += 256) {
            const int r = e / PN, c = e % PN;
            const int gr = k0 + r, gc = col0 + c;
            st[r * SS + c] = (gr < N && gc < D) ? S[(size_t)gr * D + gc] : 0.f;
        }
        __syncthreads();
        // A operand lane map (16x16x4): A[i = lane&15][k = lane>>4]; B[k = lane>>4][j = lane&15]
        const int ai = wave * 16 + (lane & 15), kk = lane >> 4, bj = lane & 15;
#pragma unroll
        for (int ks = 0; ks < PK; ks += 4) {
            const float av = kt[ai * KS + ks + kk];
            const float b0 = st[(ks + kk) * SS + bj];
            const float b1 = st[(ks + kk) * SS + 16 + bj];
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1, acc1, 0, 0, 0);
        }
        __syncthreads();
    }
    // C/D map (16x16): col = lane&15, row = (lane>>4)*4 + reg
    const float invN = 1.0f / (float)N;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const f32x4 a = half ? acc1 : acc0;
#pragma unroll
        for (int rgi = 0; rgi < 4; ++rgi) {
            const int gr = row0 + wave * 16 + (lane >> 4) * 4 + rgi;
            const int gc = col0 + half * 16 + (lane & 15);
            if (gr < N && gc < D) {
                const size_t idx = (size_t)gr * D + gc;
                float v = -((a[rgi] - gk[idx]) * invN);
                if (mask) v *= mask[idx];
                v_out[idx] = v;
                if (X_out) X_out[idx] = X_in[idx] - lr * v;
            }
        }
    }
}

int phi_launch(const float *K, const float *score, const float *grad_k, const float *mask, int N, int D,
               float *v_out, const float *X_in, float *X_out, float lr, hipStream_t stream)
{
    if (N < 1 || D < 1 || !K || !score || !grad_k || !v_out) {
        set_error("svgd_phi: bad arguments N=%d D=%d", N, D);
        return SIGSVGD_E_BADARG;
    }
    if ((X_in == nullptr) != (X_out == nullptr)) {
        set_error("svgd_phi: X_in and X_out must both be given or both be NULL");
        return SIGSVGD_E_BADARG;
    }
    dim3 grid((D + PN - 1) / PN, (N + PM - 1) / PM);
    hipLaunchKernelGGL(svgd_phi_kernel, grid, dim3(256), 0, stream, K, score, grad_k, mask, N, D, v_out, X_in, X_out, lr);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "launch svgd_phi_kernel");
    return SIGSVGD_OK;
}

} // namespace sigsvgd
