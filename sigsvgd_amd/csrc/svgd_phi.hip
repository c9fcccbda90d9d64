// SVGD velocity: v = -((K @ score - grad_k) / N) [* mask], optionally fused with the simple Adagrad
// scaling of the reference (svgd.py:110-113) and the optimizer=None particle update X_out = X_in - lr * v
// (reference src/inference/svgd.py:82-83,115; mask: src/inference/trajectory_svgd.py:84), or with the update of
// the reference's DEFAULT optimizer, torch.optim.Adam driven through a closure that sets X.grad = v
// (svgd.py:20,100-107): exp_avg / exp_avg_sq updated in place, bias corrections from a step counter that lives
// on the device (so the launch can be replayed from a captured graph), X_out = X_in - lr/bc1 * m / (sqrt(v2)/sqrt(bc2) + eps).
//
// The N x N x D product is the only GEMM-shaped piece of the hot path and runs on the fp32 MFMA
// (v_mfma_f32_16x16x4_f32: exact fp32 FMA chain, same numerics as the reference's fp32 matmul).
// It is small (0.94 GFLOP at N=1024, D=448) and latency-bound, so the kernel is organised for
// overlap rather than tile size: a workgroup of 4 wavefronts owns a 32 x 64 output tile (224
// workgroups at C4: one per CU), every wavefront computes the whole tile over ITS quarter of each
// 64-deep k stage (split-K inside the workgroup, combined through LDS at the end), global loads are
// 16 B per lane and the next stage is fetched into registers while the current one is multiplied.
#include "sig_common.h"

namespace sigsvgd {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int PM = 32;  // rows per workgroup
constexpr int PN = 64;  // cols per workgroup
constexpr int PK = 64;  // k per stage (16 per wavefront)
constexpr int KS = PK + 4; // LDS row strides (floats), padded: 16-B aligned, conflict-light
constexpr int SS = PN + 16; // = 16 mod 32: the two k rows a 32-lane read touches land on disjoint banks

struct AdamArgs {
    float *exp_avg, *exp_avg_sq; // [N, D] fp32 state, updated in place; exp_avg == NULL: no Adam
    const int *step;             // device counter: updates done so far (this launch is update *step + 1)
    double lr, beta1, beta2;     // doubles as torch holds them: 1 - beta and the bias corrections are formed in fp64
    float eps;
};

__global__ void counter_inc_kernel(int *ctr) { *ctr += 1; }

__global__ __launch_bounds__(256) void svgd_phi_kernel(const float *__restrict__ K, const float *__restrict__ S,
                                                       const float *__restrict__ gk, const float *__restrict__ mask,
                                                       int N, int D, float *__restrict__ v_out,
                                                       // (no __restrict__ on the particles: the in-place update passes the
                                                       //  same buffer as X_in and X_out; every element is read and then
                                                       //  written by the same thread)
                                                       const float *X_in, float *X_out, float lr,
                                                       float *__restrict__ adagrad, AdamArgs adam)
{
    __shared__ __align__(16) float kt[PM * KS];     // K tile      [row][k]
    __shared__ __align__(16) float st[PK * SS];     // score tile  [k][col]
    __shared__ __align__(16) float red[4 * PM * PN]; // partial tiles of the four wavefronts (k quarters)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.y * PM, col0 = blockIdx.x * PN;
    const bool vecK = (N % 4) == 0 && (reinterpret_cast<uintptr_t>(K) & 15) == 0;
    const bool vecS = (D % 4) == 0 && (reinterpret_cast<uintptr_t>(S) & 15) == 0;

    // this thread's share of a stage: K tile 32x64 = 512 float4 (2 per thread), score tile 64x64 = 1024 float4 (4)
    f32x4 ka[2], sa[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = tid + u * 256, r = e / (PK / 4), c4 = (e % (PK / 4)) * 4;
            const int gr = row0 + r, gc = k0 + c4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (gr < N) {
                if (vecK && gc + 3 < N)
                    v = *reinterpret_cast<const f32x4 *>(K + (size_t)gr * N + gc);
                else
#pragma unroll
                    for (int x = 0; x < 4; ++x)
                        if (gc + x < N) v[x] = K[(size_t)gr * N + gc + x];
            }
            ka[u] = v;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + u * 256, r = e / (PN / 4), c4 = (e % (PN / 4)) * 4;
            const int gr = k0 + r, gc = col0 + c4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (gr < N) {
                if (vecS && gc + 3 < D)
                    v = *reinterpret_cast<const f32x4 *>(S + (size_t)gr * D + gc);
                else
#pragma unroll
                    for (int x = 0; x < 4; ++x)
                        if (gc + x < D) v[x] = S[(size_t)gr * D + gc + x];
            }
            sa[u] = v;
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = tid + u * 256, r = e / (PK / 4), c4 = (e % (PK / 4)) * 4;
            *reinterpret_cast<f32x4 *>(&kt[r * KS + c4]) = ka[u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + u * 256, r = e / (PN / 4), c4 = (e % (PN / 4)) * 4;
            *reinterpret_cast<f32x4 *>(&st[r * SS + c4]) = sa[u];
        }
    };

    f32x4 acc[2][4]; // [row half][col quarter] 16x16 accumulators of this wavefront's k quarter
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    fetch(0);
    for (int k0 = 0; k0 < N; k0 += PK) {
        __syncthreads(); // previous stage fully consumed
        stash();
        __syncthreads();
        if (k0 + PK < N) fetch(k0 + PK); // in flight during the MFMAs below
        // A operand (16x16x4): A[i = lane&15][k = lane>>4]; B[k = lane>>4][j = lane&15]
        const int kk = lane >> 4, ij = lane & 15, kw = wave * 16;
#pragma unroll
        for (int ks = 0; ks < 16; ks += 4) {
            const float a0 = kt[ij * KS + kw + ks + kk];
            const float a1 = kt[(16 + ij) * KS + kw + ks + kk];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const float bv = st[(kw + ks + kk) * SS + b * 16 + ij];
                acc[0][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv, acc[0][b], 0, 0, 0);
                acc[1][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv, acc[1][b], 0, 0, 0);
            }
        }
    }
    // combine the four k-quarters: C/D map (16x16): col = lane&15, row = (lane>>4)*4 + reg.  Every wavefront parks its
    // partial tile, then ALL 256 threads run the epilogue, 8 outputs each with consecutive threads on consecutive columns
    // (round 2 left it to wavefront 0: 32 outputs per lane, one load-compute-store chain after the other -- 46 of the
    // kernel's 47 us at C4 were this tail)
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                red[wave * PM * PN + (a * 16 + (lane >> 4) * 4 + r) * PN + b * 16 + (lane & 15)] = acc[a][b][r];
    __syncthreads();
    const float invN = 1.0f / (float)N;
    float step_size = lr, inv_sqrt_bc2 = 1.f, omb1 = 0.f, omb2 = 0.f, b2 = 0.f;
    if (adam.exp_avg) { // torch.optim.Adam (amsgrad=False, weight_decay=0, maximize=False): scalars in fp64 as torch
        const double t = (double)(*adam.step + 1);
        step_size = (float)(adam.lr / (1.0 - pow(adam.beta1, t)));
        inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow(adam.beta2, t)));
        omb1 = (float)(1.0 - adam.beta1);
        omb2 = (float)(1.0 - adam.beta2);
        b2 = (float)adam.beta2;
    }
    constexpr int EPT = PM * PN / 256; // outputs per thread
    float sv[EPT], gv[EPT], xv[EPT], mv[EPT], av[EPT], e1[EPT], e2[EPT];
    bool ok[EPT];
#pragma unroll
    for (int u = 0; u < EPT; ++u) { // all loads first: one round trip for the lot
        const int el = tid + u * 256, lr_ = el / PN, lc = el % PN;
        const int gr = row0 + lr_, gc = col0 + lc;
        ok[u] = gr < N && gc < D;
        const size_t idx = ok[u] ? (size_t)gr * D + gc : 0;
        sv[u] = ((red[lr_ * PN + lc] + red[PM * PN + lr_ * PN + lc]) + red[2 * PM * PN + lr_ * PN + lc]) +
                red[3 * PM * PN + lr_ * PN + lc];
        gv[u] = gk[idx];
        mv[u] = mask ? mask[idx] : 1.f;
        av[u] = adagrad ? adagrad[idx] : 0.f;
        xv[u] = X_in ? X_in[idx] : 0.f;
        e1[u] = adam.exp_avg ? adam.exp_avg[idx] : 0.f;
        e2[u] = adam.exp_avg ? adam.exp_avg_sq[idx] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
        if (!ok[u]) continue;
        const int el = tid + u * 256;
        const size_t idx = (size_t)(row0 + el / PN) * D + col0 + el % PN;
        float v = -((sv[u] - gv[u]) * invN) * mv[u];
        if (adagrad) { // reference svgd.py:110-113: running sum of squared gradients, g / sqrt(sum + 1e-12)
            const float acc2 = av[u] + v * v;
            adagrad[idx] = acc2;
            v = v / sqrtf(acc2 + 1e-12f);
        }
        v_out[idx] = v;
        if (adam.exp_avg) {
            const float m = e1[u] + omb1 * (v - e1[u]); // lerp
            const float q = b2 * e2[u] + omb2 * (v * v);
            adam.exp_avg[idx] = m;
            adam.exp_avg_sq[idx] = q;
            X_out[idx] = xv[u] - step_size * (m / (sqrtf(q) * inv_sqrt_bc2 + adam.eps));
        } else if (X_out)
            X_out[idx] = xv[u] - lr * v;
    }
}

int phi_launch(const float *K, const float *score, const float *grad_k, const float *mask, int N, int D,
               float *v_out, const float *X_in, float *X_out, float lr, float *adagrad, hipStream_t stream,
               float *exp_avg, float *exp_avg_sq, int *step_dev, double lr_adam, double beta1, double beta2, float eps)
{
    if (N < 1 || D < 1 || !K || !score || !grad_k || !v_out) {
        set_error("svgd_phi: bad arguments N=%d D=%d", N, D);
        return SIGSVGD_E_BADARG;
    }
    if ((X_in == nullptr) != (X_out == nullptr)) {
        set_error("svgd_phi: X_in and X_out must both be given or both be NULL");
        return SIGSVGD_E_BADARG;
    }
    AdamArgs adam{exp_avg, exp_avg_sq, step_dev, lr_adam, beta1, beta2, eps};
    if (exp_avg && (!exp_avg_sq || !step_dev || !X_in || adagrad)) {
        set_error("svgd_adam_step: needs exp_avg, exp_avg_sq, step, X_in/X_out (and no Adagrad state)");
        return SIGSVGD_E_BADARG;
    }
    dim3 grid((D + PN - 1) / PN, (N + PM - 1) / PM);
    hipLaunchKernelGGL(svgd_phi_kernel, grid, dim3(256), 0, stream, K, score, grad_k, mask, N, D, v_out, X_in, X_out, lr,
                       adagrad, adam);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "launch svgd_phi_kernel");
    if (exp_avg) { // the counter moves only after every workgroup of the update has read it (stream order)
        hipLaunchKernelGGL(counter_inc_kernel, dim3(1), dim3(1), 0, stream, step_dev);
        e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "launch counter_inc_kernel");
    }
    return SIGSVGD_OK;
}

} // namespace sigsvgd
