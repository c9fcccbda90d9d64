// Issue/pipe model of the gfx950 vector unit for the instruction classes the signature-kernel sweeps
// are made of: cycles per wave-instruction (s_memtime, shader clock) at 1, 2, 3 and 4 waves per SIMD,
// independent and dependent streams.  Decides between "fewer instructions", "more waves" and "ILP".
//   hipcc --offload-arch=gfx950 -O3 -o valu_model valu_model.hip && ./valu_model
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

enum Kind {
    F64_IND, F64_DEP, F32_IND, F32_DEP, PK_IND, PK_DEP, DPP_IND, DPP_DEP, CVT64_IND, CVT32_IND, ADD64_IND,
    MIX_64_32, MIX_64_DPP, F64_SALU, F32_SALU, EXP32_IND, RNDNE64, LDEXP64, F32_CMPX, STEP_LIKE, STEP_PAIR, NKIND
};
static const char *names[NKIND] = {
    "v_fma_f64 independent", "v_fma_f64 dependent", "v_fma_f32 independent", "v_fma_f32 dependent",
    "v_pk_fma_f32 independent", "v_pk_fma_f32 dependent", "v_mov_b32_dpp wave_shr independent",
    "v_mov_b32_dpp dependent chain", "v_cvt_f64_f32 independent", "v_cvt_f32_f64 independent",
    "v_add_f64 independent", "4 x fma_f64 + 4 x fma_f32 interleaved", "4 x fma_f64 + 4 x dpp mov interleaved",
    "4 x fma_f64 + 4 x s_add_u32", "4 x fma_f32 + 4 x s_add_u32", "v_exp_f32 independent", "v_rndne_f64 independent",
    "v_ldexp_f64 independent", "4 x fma_f32 + 4 x (v_cmp + s_and_saveexec/s_mov exec)",
    "sweep-step-like dependent chain (2 dpp, 2 cvt, add, sub, 2 fma f64, cmp, 3 masked mov)",
    "two interleaved sweep-step-like chains"};

template <int KIND>
__global__ void k(float *out, int iters, long long *cyc)
{
    const int lane = threadIdx.x & 63;
    double a[8], b = 1.0000001, c = 1e-9;
    float f[8], fb = 1.0000001f, fc = 1e-9f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[8], pb = {1.0000001f, 1.0000001f}, pc = {1e-9f, 1e-9f};
    int s[4] = {1, 2, 3, 4};
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        a[u] = 1.0 + lane * 1e-9 + u;
        f[u] = 1.0f + lane * 1e-6f + u;
        p[u] = f2{f[u], f[u] + 1};
    }
    double cur = a[0], up = a[1], diag = a[2], cur2 = a[3], up2 = a[4], diag2 = a[5];
    float g = f[0], g2 = f[1];
    int thr = 63, sig = 0;
    float ksl = 0.f, c12 = 1.0f / 12.0f, chalf = 0.5f;
    asm volatile("" : "+s"(c12), "+s"(chalf));
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (KIND == F64_IND) {
#define X(u) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[u]) : "v"(b), "v"(c));
                REP8(X)
#undef X
            } else if (KIND == F64_DEP) {
#define X(u) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
                REP8(X)
#undef X
            } else if (KIND == F32_IND) {
#define X(u) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[u]) : "v"(fb), "v"(fc));
                REP8(X)
#undef X
            } else if (KIND == F32_DEP) {
#define X(u) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[0]) : "v"(fb), "v"(fc));
                REP8(X)
#undef X
            } else if (KIND == PK_IND) {
#define X(u) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[u]) : "v"(pb), "v"(pc));
                REP8(X)
#undef X
            } else if (KIND == PK_DEP) {
#define X(u) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[0]) : "v"(pb), "v"(pc));
                REP8(X)
#undef X
            } else if (KIND == DPP_IND) {
#define X(u) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(f[u]) : "v"(fb));
                REP8(X)
#undef X
            } else if (KIND == DPP_DEP) {
#define X(u) asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(f[0]));
                REP8(X)
#undef X
            } else if (KIND == CVT64_IND) {
#define X(u) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[u]) : "v"(f[u]));
                REP8(X)
#undef X
            } else if (KIND == CVT32_IND) {
#define X(u) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[u]) : "v"(a[u]));
                REP8(X)
#undef X
            } else if (KIND == ADD64_IND) {
#define X(u) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[u]) : "v"(c));
                REP8(X)
#undef X
            } else if (KIND == MIX_64_32) {
#define X(u) if (u & 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[u]) : "v"(fb), "v"(fc)); \
             else asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[u]) : "v"(b), "v"(c));
                REP8(X)
#undef X
            } else if (KIND == MIX_64_DPP) {
#define X(u) if (u & 1) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(f[u]) : "v"(fb)); \
             else asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[u]) : "v"(b), "v"(c));
                REP8(X)
#undef X
            } else if (KIND == F64_SALU) {
#define X(u) if (u & 1) asm volatile("s_add_u32 %0, %0, 1" : "+s"(s[u >> 1]) : : "scc"); \
             else asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[u]) : "v"(b), "v"(c));
                REP8(X)
#undef X
            } else if (KIND == F32_SALU) {
#define X(u) if (u & 1) asm volatile("s_add_u32 %0, %0, 1" : "+s"(s[u >> 1]) : : "scc"); \
             else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[u]) : "v"(fb), "v"(fc));
                REP8(X)
#undef X
            } else if (KIND == EXP32_IND) {
#define X(u) asm volatile("v_exp_f32 %0, %0" : "+v"(f[u]));
                REP8(X)
#undef X
            } else if (KIND == RNDNE64) {
#define X(u) asm volatile("v_rndne_f64 %0, %0" : "+v"(a[u]));
                REP8(X)
#undef X
            } else if (KIND == LDEXP64) {
#define X(u) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a[u]) : "v"(s[0]));
                REP8(X)
#undef X
            } else if (KIND == F32_CMPX) {
#define X(u) if (u & 1) asm volatile("v_cmp_lt_u32 vcc, %1, %2\n\ts_and_saveexec_b64 s[20:21], vcc\n\tv_mov_b32 %0, %3\n\ts_mov_b64 exec, s[20:21]" \
                                     : "+v"(f[u]) : "v"(lane), "v"(thr), "v"(fb) : "vcc", "s20", "s21"); \
             else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[u]) : "v"(fb), "v"(fc));
                REP8(X)
#undef X
            } else if (KIND == STEP_LIKE || KIND == STEP_PAIR) {
                // one forward-sweep step as in gram_fast_kernel: dpp x2, cvt x2, add, sub, fma, fma, cmp, masked movs
#define STEP(CUR, UP, DIAG, G)                                                                                   \
    {                                                                                                             \
        int dlo = __double2loint(UP), dhi = __double2hiint(UP);                                                  \
        asm("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(dlo) : "v"(__double2loint(CUR))); \
        asm("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(dhi) : "v"(__double2hiint(CUR))); \
        UP = __hiloint2double(dhi, dlo);                                                                          \
        const float bb = G * (G * c12);                                                                           \
        const float aa = __builtin_fmaf(G, chalf, bb);                                                            \
        const double t = CUR + UP;                                                                                \
        double uu = t - DIAG;                                                                                     \
        uu = __builtin_fma(t, (double)aa, uu);                                                                    \
        const double nw = __builtin_fma(DIAG, (double)bb, uu);                                                    \
        if ((unsigned)(sig - lane) < (unsigned)thr) {                                                             \
            ksl = (float)DIAG;                                                                                    \
            CUR = nw;                                                                                             \
        }                                                                                                         \
        sig++;                                                                                                    \
    }
#define STEP2 STEP
                // (the pair variant interleaves at statement granularity only: hipcc keeps asm volatile order,
                //  so it shows what a second independent chain buys without fine interleaving)
                STEP(cur, up, diag, g)
                if (KIND == STEP_PAIR) { STEP2(cur2, up2, diag2, g2) }
                __builtin_amdgcn_sched_barrier(0);
                STEP(cur, diag, up, g)
                if (KIND == STEP_PAIR) { STEP2(cur2, diag2, up2, g2) }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += (float)a[u] + f[u] + p[u][0] + p[u][1];
    acc += (float)(cur + up + diag + cur2 + up2 + diag2) + g + g2 + ksl + sig + s[0] + s[1] + s[2] + s[3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
void run(float *d, long long *dc, int ncu)
{
    const int iters = 2000;
    // wave-instructions per loop iteration (VALU + SALU counted separately below)
    int valu = 32, other = 0;
    if (KIND == F64_SALU || KIND == F32_SALU) { valu = 16; other = 16; }
    if (KIND == F32_CMPX) { valu = 16 + 32; other = 32; } // 4 fma + 4 x (cmp + mov) VALU, 4 x 2 SALU per 8-group
    if (KIND == STEP_LIKE) { valu = 2 * 12; other = 0; } // as compiled: 2 dpp, add, 2 add_f64, 2 fmac_f64, cmp, cvt, 3 cndmask
    if (KIND == STEP_PAIR) { valu = 4 * 12; other = 0; }
    if (KIND == STEP_LIKE || KIND == STEP_PAIR) { valu *= 4; other *= 4; }
    printf("%-88s", names[KIND]);
    for (int w : {1, 2, 3, 4}) {
        const int nw = 4 * w; // waves per workgroup: w per SIMD
        hipLaunchKernelGGL(k<KIND>, dim3(ncu), dim3(64 * nw), 0, 0, d, 200, dc);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(ncu), dim3(64 * nw), 0, 0, d, iters, dc);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(ncu * nw);
        hipMemcpy(h.data(), dc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
        double s = 0;
        for (auto v : h) s += (double)v;
        s /= h.size();
        const double per_wave = s / ((double)iters * valu); // cycles per VALU wave-instruction as one wave sees it
        printf("  w=%d: %6.2f cyc/VALU/wave -> %5.2f cyc/VALU/SIMD (%.2f GHz)", w, per_wave, per_wave / w,
               s / (ms * 1e-3) / 1e9);
        (void)other;
    }
    printf("\n");
}

int main(int argc, char **)
{
    setvbuf(stdout, nullptr, _IOLBF, 0);
    const bool rest_only = argc > 1;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount;
    printf("%s, %d CUs\n", prop.name, ncu);
    float *d;
    long long *dc;
    hipMalloc(&d, (size_t)ncu * 1024 * sizeof(float));
    hipMalloc(&dc, (size_t)ncu * 16 * sizeof(long long));
    if (!rest_only) {
    run<F64_IND>(d, dc, ncu);
    run<F64_DEP>(d, dc, ncu);
    run<F32_IND>(d, dc, ncu);
    run<F32_DEP>(d, dc, ncu);
    run<PK_IND>(d, dc, ncu);
    run<PK_DEP>(d, dc, ncu);
    run<DPP_IND>(d, dc, ncu);
    run<DPP_DEP>(d, dc, ncu);
    run<CVT64_IND>(d, dc, ncu);
    run<CVT32_IND>(d, dc, ncu);
    run<ADD64_IND>(d, dc, ncu);
    run<MIX_64_32>(d, dc, ncu);
    }
    run<MIX_64_DPP>(d, dc, ncu);
    run<F64_SALU>(d, dc, ncu);
    run<F32_SALU>(d, dc, ncu);
    run<EXP32_IND>(d, dc, ncu);
    run<RNDNE64>(d, dc, ncu);
    run<LDEXP64>(d, dc, ncu);
    run<F32_CMPX>(d, dc, ncu);
    run<STEP_LIKE>(d, dc, ncu);
    run<STEP_PAIR>(d, dc, ncu);
    return 0;
}
