// Does a wave64 VALU instruction cost fewer cycles when whole 16-lane passes are masked off by EXEC?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k(double *out, int active_lanes, int iters)
{
    const int lane = threadIdx.x & 63;
    double a = 1.0 + lane * 1e-9, b = 1.0000001, c = 1e-9;
    double a2 = a + 1, a3 = a + 2, a4 = a + 3;
    if (lane < active_lanes) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                a = __builtin_fma(a, b, c);
                a2 = __builtin_fma(a2, b, c);
                a3 = __builtin_fma(a3, b, c);
                a4 = __builtin_fma(a4, b, c);
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + a2 + a3 + a4;
}
int main()
{
    double *d;
    hipMalloc(&d, 256 * 1024 * 64 * sizeof(double));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int waves_per_simd : {1, 2}) {
        for (int act : {64, 48, 32, 16, 8, 1}) {
            const int blocks = 256 * 4 * waves_per_simd; // one wave per block
            hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d, act, 2000);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d, act, 20000);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("waves/SIMD %d active lanes %2d: %.3f ms  (%.2f cycles per wave-instruction at 2.4 GHz)\n", waves_per_simd, act, ms,
                   ms * 1e-3 * 2.4e9 / (20000.0 * 64 * waves_per_simd));
        }
    }
    return 0;
}
