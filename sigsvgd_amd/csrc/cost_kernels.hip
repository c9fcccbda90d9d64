// Trajectory cost of the reference's planning script in front of the signature-kernel path, on the device with its
// analytic gradient: knots -> natural cubic spline samples -> obstacle cost + path-length cost,
// and d cost / d knots in the same launch (the reference differentiates through torch autograd).
//
//   reference: examples/script_planning_obstacle_field.py:113-126 (batch_cost_fn) with
//              :18-23 (create_spline_trajectory: uniform knot times, `timesteps` uniform samples) and
//              :363-370 (the obstacle field: MixtureSameFamily(Categorical(w), Independent(Normal(mean, std), 1)))
//
//   knots_i   = [start, x_i[0], ..., x_i[Kx-1], target]                      [K = Kx + 2, d]
//   traj_i    = B @ knots_i                                                   [Tt, d]   (B: spline basis, or I)
//   obst_i    = w_obst * sum_t p(traj_i[t]),  p(z) = sum_m pi_m prod_c N(z_c; mu_mc, sigma_mc)
//   len_i     = || w_len * (traj_i[1:] - traj_i[:-1]) ||_F
//   cost_i    = obst_i + len_i;          grad_x[i] = d cost_i / d x_i   (the caller negates it for grad log p)
//
// One workgroup of 128 threads per particle, threads over the Tt samples; the spline is a [Tt, K] basis matrix
// (built once on the host for the fixed knot times) applied from LDS.  fp32 throughout (the reference runs this in
// fp32 on its device).  Tiny and latency-bound: the point is to keep the whole planning iteration on the GPU.
#include "sig_common.h"

namespace sigsvgd {

namespace {
constexpr int CT = 128;      // threads per workgroup
constexpr int C_KMAX = 64;   // knots including the two end poses
constexpr int C_DMAX = 16;
constexpr int C_TMAX = 1024; // trajectory samples

__device__ __forceinline__ float block_sum(float v, float *red, int tid)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return red[0] + red[1];
}
} // namespace

__global__ __launch_bounds__(CT) void obstacle_cost_kernel(const float *__restrict__ x, int N, int Kx, int d,
                                                          const float *__restrict__ start, const float *__restrict__ target,
                                                          const float *__restrict__ basis, int Tt,
                                                          const float *__restrict__ logw, const float *__restrict__ mean,
                                                          const float *__restrict__ stdv, int M, float w_obst, float w_len,
                                                          float *__restrict__ cost, float *__restrict__ traj,
                                                          float *__restrict__ grad_x)
{
    extern __shared__ float smem[];
    const int K = Kx + 2, tid = threadIdx.x, i = blockIdx.x;
    float *knots = smem;                 // [K][d]
    float *zs = knots + K * d;           // [Tt][d] samples, then d cost / d sample
    float *red = zs + (size_t)Tt * d;    // [2]
    __shared__ float s_invL;

    for (int e = tid; e < K * d; e += CT) {
        const int k = e / d, c = e % d;
        knots[e] = (k == 0) ? start[c] : (k == K - 1) ? target[c] : x[((size_t)i * Kx + (k - 1)) * d + c];
    }
    __syncthreads();
    // samples
    for (int t = tid; t < Tt; t += CT) {
        const float *b = basis + (size_t)t * K;
        for (int c = 0; c < d; ++c) {
            float z = 0.f;
            for (int k = 0; k < K; ++k) z = __builtin_fmaf(b[k], knots[k * d + c], z);
            zs[t * d + c] = z;
            if (traj) traj[((size_t)i * Tt + t) * d + c] = z;
        }
    }
    __syncthreads();
    // path length
    float l2 = 0.f;
    for (int t = tid; t + 1 < Tt; t += CT)
        for (int c = 0; c < d; ++c) {
            const float dz = zs[(t + 1) * d + c] - zs[t * d + c];
            l2 = __builtin_fmaf(dz, dz, l2);
        }
    l2 = block_sum(l2, red, tid);
    const float L = sqrtf(l2);
    if (tid == 0) s_invL = (L > 0.f) ? 1.f / L : 0.f;
    __syncthreads();
    const float invL = s_invL;
    // obstacle field and d cost / d sample (kept in registers until every thread has read its neighbours)
    float psum = 0.f;
    float gz[C_DMAX];
    const int reps = (Tt + CT - 1) / CT;
    for (int r = 0; r < reps; ++r) {
        const int t = tid + r * CT;
        const bool ok = t < Tt;
        float z[C_DMAX];
#pragma unroll
        for (int c = 0; c < C_DMAX; ++c) {
            z[c] = (ok && c < d) ? zs[t * d + c] : 0.f;
            gz[c] = 0.f;
        }
        if (ok) {
            for (int m = 0; m < M; ++m) {
                float q = 0.f, lognorm = logw[m];
                for (int c = 0; c < d; ++c) {
                    const float s = stdv[m * d + c], u = (z[c] - mean[m * d + c]) / s;
                    q = __builtin_fmaf(u, u, q);
                    lognorm -= __logf(s) + 0.91893853320467274f; // log sigma + log sqrt(2 pi)
                }
                const float pm = __expf(lognorm - 0.5f * q);
                psum += pm;
                for (int c = 0; c < d; ++c) {
                    const float s = stdv[m * d + c];
                    gz[c] -= pm * (z[c] - mean[m * d + c]) / (s * s);
                }
            }
#pragma unroll
            for (int c = 0; c < C_DMAX; ++c) {
                if (c < d) {
                    const float dl = (t > 0 ? z[c] - zs[(t - 1) * d + c] : 0.f) - (t + 1 < Tt ? zs[(t + 1) * d + c] - z[c] : 0.f);
                    gz[c] = w_obst * gz[c] + w_len * dl * invL; // d len / d z_t = w (Delta[t-1] - Delta[t]) / L
                }
            }
        }
        __syncthreads(); // every thread of this round has read its neighbours' samples
        if (ok)
#pragma unroll
            for (int c = 0; c < C_DMAX; ++c)
                if (c < d) zs[t * d + c] = gz[c];
        __syncthreads();
    }
    // NOTE: with more than one round (Tt > 128) a later round reads neighbours that an earlier round already turned
    // into gradients; the host restricts Tt <= 128 for the gradient output (the reference uses 100).
    psum = block_sum(psum, red, tid);
    if (tid == 0) cost[i] = w_obst * psum + w_len * L;
    // d cost / d knots = B^T (d cost / d samples): one thread per interior (knot, channel)
    if (grad_x)
        for (int e = tid; e < Kx * d; e += CT) {
            const int k = e / d + 1, c = e % d;
            float g = 0.f;
            for (int t = 0; t < Tt; ++t) g = __builtin_fmaf(basis[(size_t)t * K + k], zs[t * d + c], g);
            grad_x[((size_t)i * Kx + (k - 1)) * d + c] = g;
        }
}

int obstacle_cost_launch(const float *x, int N, int Kx, int d, const float *start, const float *target, const float *basis,
                         int Tt, const float *logw, const float *mean, const float *stdv, int M, float w_obst, float w_len,
                         float *cost, float *traj, float *grad_x, hipStream_t stream)
{
    if (N < 1 || Kx < 0 || d < 1 || d > C_DMAX || Kx + 2 > C_KMAX || Tt < 2 || Tt > C_TMAX || M < 1) {
        set_error("obstacle_cost: unsupported shape N=%d knots=%d d=%d samples=%d components=%d (need d <= %d, knots + 2 <= %d, "
                  "2 <= samples <= %d)", N, Kx, d, Tt, M, C_DMAX, C_KMAX, C_TMAX);
        return SIGSVGD_E_UNSUPPORTED;
    }
    if (grad_x && Tt > CT) {
        set_error("obstacle_cost: the gradient output supports up to %d trajectory samples (got %d)", CT, Tt);
        return SIGSVGD_E_UNSUPPORTED;
    }
    if ((!x && Kx > 0) || !start || !target || !basis || !logw || !mean || !stdv || !cost) {
        set_error("obstacle_cost: null pointer argument");
        return SIGSVGD_E_BADARG;
    }
    const size_t shmem = ((size_t)(Kx + 2) * d + (size_t)Tt * d + 4) * sizeof(float);
    if (shmem > 64 * 1024) { // (dynamic LDS beyond 64 KB would need an opt-in attribute and fail at launch, not here)
        set_error("obstacle_cost: knots and samples need %zu B of LDS per particle (> 64 KiB): knots=%d samples=%d d=%d", shmem,
                  Kx, Tt, d);
        return SIGSVGD_E_UNSUPPORTED;
    }
    hipLaunchKernelGGL(obstacle_cost_kernel, dim3(N), dim3(CT), shmem, stream, x, N, Kx, d, start, target, basis, Tt, logw,
                       mean, stdv, M, w_obst, w_len, cost, traj, grad_x);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "launch obstacle_cost_kernel");
    return SIGSVGD_OK;
}

} // namespace sigsvgd
