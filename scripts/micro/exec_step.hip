// A/B of the forward-sweep step of gram_fast.hip at the kernel's occupancy (8 waves per workgroup, one workgroup per CU):
//   A  the shipped step: activity through v_cmp + two v_cndmask, no scalar instruction (10 / 11 VALU)
//   B  activity through EXEC: full EXEC for the DPP shift, s_bfm_b64 EXEC window for the arithmetic (7 / 8 VALU + 2 SALU)
//   C  as B with the EXEC window taken from an SGPR pair shifted by SALU each step (general P): 7 / 8 VALU + 4 SALU
// Reports ns per step per SIMD-wave pair.   hipcc --offload-arch=gfx950 -O3 -o exec_step exec_step.hip && ./exec_step
#include <hip/hip_runtime.h>
#include <cstdio>

#define STEP_A(UP, DIAG, G, K)                                                                \
    "v_cmp_gt_u32 vcc, %[P], %[cnt]\n\t"                                                      \
    "v_add_u32 %[cnt], 1, %[cnt]\n\t"                                                         \
    "v_mov_b32_dpp %[" UP "], %[cur] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"               \
    "v_cndmask_b32 %[ge], 0, %[" G "], vcc\n\t"                                               \
    "v_add_f32 %[t], %[cur], %[" UP "]\n\t"                                                   \
    "v_mul_f32 %[y], %[r3], %[t]\n\t"                                                         \
    "v_add_f32 %[t], %[t], %[" DIAG "]\n\t"                                                   \
    "v_fmac_f32 %[y], %[t], %[ge]\n\t"                                                        \
    "v_cndmask_b32 %[" K "], %[" K "], %[" DIAG "], vcc\n\t"                                  \
    "v_fmac_f32 %[V], %[ge], %[y]\n\t"                                                        \
    "v_add_f32 %[cur], %[" UP "], %[V]\n\t"

#define STEP_B(UP, DIAG, G, K, W, O)                                                          \
    "s_mov_b64 exec, -1\n\t"                                                                  \
    "v_mov_b32_dpp %[" UP "], %[cur] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"               \
    "s_bfm_b64 exec, " W ", " O "\n\t"                                                        \
    "v_add_f32 %[t], %[cur], %[" UP "]\n\t"                                                   \
    "v_mul_f32 %[y], %[r3], %[t]\n\t"                                                         \
    "v_add_f32 %[t], %[t], %[" DIAG "]\n\t"                                                   \
    "v_fmac_f32 %[y], %[t], %[" G "]\n\t"                                                     \
    "v_fmac_f32 %[V], %[" G "], %[y]\n\t"                                                     \
    "v_add_f32 %[cur], %[" UP "], %[V]\n\t"                                                   \
    "v_mov_b32 %[" K "], %[" DIAG "]\n\t"

#define STEP_C(UP, DIAG, G, K)                                                                \
    "s_mov_b64 exec, -1\n\t"                                                                  \
    "v_mov_b32_dpp %[" UP "], %[cur] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"               \
    "s_lshl_b64 %[m], %[m], 1\n\t"                                                            \
    "s_and_b64 exec, %[m], %[rows]\n\t"                                                       \
    "v_add_f32 %[t], %[cur], %[" UP "]\n\t"                                                   \
    "v_mul_f32 %[y], %[r3], %[t]\n\t"                                                         \
    "v_add_f32 %[t], %[t], %[" DIAG "]\n\t"                                                   \
    "v_fmac_f32 %[y], %[t], %[" G "]\n\t"                                                     \
    "v_fmac_f32 %[V], %[" G "], %[y]\n\t"                                                     \
    "v_add_f32 %[cur], %[" UP "], %[V]\n\t"                                                   \
    "v_mov_b32 %[" K "], %[" DIAG "]\n\t"

// ---- gram_quad.hip's step and what its extra instructions cost -----------------------------------------------------
#define STEP_Q(UP, DIAG, G, K, SHI, LN, EXTRA_A, EXTRA_B, EXTRA_C)                            \
    "s_lshr_b64 %[tm], %[wr], " SHI "\n\t"                                                    \
    "s_mov_b64 exec, -1\n\t"                                                                  \
    "v_mov_b32_dpp %[" UP "], %[cur] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" EXTRA_A       \
    "s_and_b64 exec, %[tm], %[rows]\n\t"                                                      \
    "v_add_f32 %[t], %[cur], %[" UP "]\n\t"                                                   \
    "v_mul_f32 %[y], %[r3], %[t]\n\t"                                                         \
    "v_add_f32 %[t], %[t], %[" DIAG "]\n\t"                                                   \
    "v_fmac_f32 %[y], %[t], %[" G "]\n\t"                                                     \
    "v_fmac_f32 %[V], %[" G "], %[y]\n\t"                                                     \
    "v_add_f32 %[cur], %[" UP "], %[V]\n\t"                                                   \
    "v_mov_b32 %[" K "], %[" DIAG "]\n\t" EXTRA_B EXTRA_C
#define QRL(LN) "v_readlane_b32 %[sb], %[hb], " LN "\n\t"
#define QMV(DIAG) "v_mov_b32 %[" DIAG "], %[sb]\n\t"
#define QDS "ds_write_b32 %[ha], %[cur]\n\t" "v_add_u32 %[ha], %[hinc], %[ha]\n\t"
#define QDS1 "ds_write_b32 %[ha], %[cur]\n\t"

template <int KIND>
__global__ __launch_bounds__(1024) void kq(float *out, int iters, int P)
{
    __shared__ float sink[1024 + 64];
    const int lane = threadIdx.x & 63;
    float g[8], ks[8];
    for (int u = 0; u < 8; ++u) {
        g[u] = 1e-3f * (u + 1) + lane * 1e-6f;
        ks[u] = 0.f;
    }
    float cur = 1.f, upA = 1.f, upB = 1.f, V = 0.f, r3 = 1.7320508f, t, y, hb = 1.f + lane;
    unsigned long long wr = ~0ull << 8, rows = ~0ull >> 1, tm;
    int sb, ha = (int)(size_t)(sink + threadIdx.x), hinc = 0;
    asm volatile("" : "+s"(r3), "+s"(wr), "+s"(rows), "+v"(hinc));
#define QARGS                                                                                                          \
    : [cur] "+v"(cur), [upA] "+v"(upA), [upB] "+v"(upB), [V] "+v"(V), [t] "=&v"(t), [y] "=&v"(y), [k0] "+v"(ks[0]),      \
      [k1] "+v"(ks[1]), [k2] "+v"(ks[2]), [k3] "+v"(ks[3]), [tm] "=&s"(tm), [sb] "=&s"(sb), [ha] "+v"(ha)                \
    : [g0] "v"(g[0]), [g1] "v"(g[1]), [g2] "v"(g[2]), [g3] "v"(g[3]), [r3] "s"(r3), [wr] "s"(wr), [rows] "s"(rows),      \
      [hb] "v"(hb), [hinc] "v"(hinc)                                                                                     \
    : "scc"
    for (int i = 0; i < iters; ++i) {
        for (int r = 0; r < 2; ++r) {
            if (KIND == 0) // window from a shifted constant, nothing else
                asm volatile(STEP_Q("upA", "upB", "g0", "k0", "20", "", "", "", "") STEP_Q("upB", "upA", "g1", "k1", "19", "", "", "", "")
                             STEP_Q("upA", "upB", "g2", "k2", "18", "", "", "", "") STEP_Q("upB", "upA", "g3", "k3", "17", "", "", "", "")
                             "s_mov_b64 exec, -1\n\t" QARGS);
            else if (KIND == 1) // + boundary value through v_readlane / v_mov
                asm volatile(STEP_Q("upA", "upB", "g0", "k0", "20", "", QRL("3"), QMV("upB"), "") STEP_Q("upB", "upA", "g1", "k1", "19", "", QRL("4"), QMV("upA"), "")
                             STEP_Q("upA", "upB", "g2", "k2", "18", "", QRL("5"), QMV("upB"), "") STEP_Q("upB", "upA", "g3", "k3", "17", "", QRL("6"), QMV("upA"), "")
                             "s_mov_b64 exec, -1\n\t" QARGS);
            else if (KIND == 2) // + hand-over store and its address update
                asm volatile(STEP_Q("upA", "upB", "g0", "k0", "20", "", "", "", QDS) STEP_Q("upB", "upA", "g1", "k1", "19", "", "", "", QDS)
                             STEP_Q("upA", "upB", "g2", "k2", "18", "", "", "", QDS) STEP_Q("upB", "upA", "g3", "k3", "17", "", "", "", QDS)
                             "s_mov_b64 exec, -1\n\t" QARGS);
            else if (KIND == 3) // the shipped step: both
                asm volatile(STEP_Q("upA", "upB", "g0", "k0", "20", "", QRL("3"), QMV("upB"), QDS) STEP_Q("upB", "upA", "g1", "k1", "19", "", QRL("4"), QMV("upA"), QDS)
                             STEP_Q("upA", "upB", "g2", "k2", "18", "", QRL("5"), QMV("upB"), QDS) STEP_Q("upB", "upA", "g3", "k3", "17", "", QRL("6"), QMV("upA"), QDS)
                             "s_mov_b64 exec, -1\n\t" QARGS);
            else // hand-over store without the address update
                asm volatile(STEP_Q("upA", "upB", "g0", "k0", "20", "", "", "", QDS1) STEP_Q("upB", "upA", "g1", "k1", "19", "", "", "", QDS1)
                             STEP_Q("upA", "upB", "g2", "k2", "18", "", "", "", QDS1) STEP_Q("upB", "upA", "g3", "k3", "17", "", "", "", QDS1)
                             "s_mov_b64 exec, -1\n\t" QARGS);
        }
    }
    float s = cur + V + sink[threadIdx.x];
    for (int u = 0; u < 4; ++u) s += ks[u];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int NT>
void runq(float *d, int ncu, const char *name)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    hipLaunchKernelGGL(kq<KIND>, dim3(ncu), dim3(NT), 0, 0, d, 100, 63);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kq<KIND>, dim3(ncu), dim3(NT), 0, 0, d, iters, 63);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-72s %.3f ms   %.2f ns per step of one wave (%d wave%s/SIMD)\n", name, ms, ms * 1e6 / (iters * 8.0) / (NT / 256),
           NT / 256, NT >= 512 ? "s" : "");
}

template <int KIND>
__global__ __launch_bounds__(512) void k(float *out, int iters, int P)
{
    const int lane = threadIdx.x & 63;
    float g[8], ks[8];
    for (int u = 0; u < 8; ++u) {
        g[u] = 1e-3f * (u + 1) + lane * 1e-6f;
        ks[u] = 0.f;
    }
    float cur = 1.f, upA = 1.f, upB = 1.f, V = 0.f, r3 = 1.7320508f, ge, t, y;
    int cnt = -lane;
    unsigned long long m = 1, rows = ~0ull >> 1;
    asm volatile("" : "+s"(r3), "+s"(m), "+s"(rows));
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0)
            asm volatile(STEP_A("upA", "upB", "g0", "k0") STEP_A("upB", "upA", "g1", "k1") STEP_A("upA", "upB", "g2", "k2")
                         STEP_A("upB", "upA", "g3", "k3") STEP_A("upA", "upB", "g4", "k4") STEP_A("upB", "upA", "g5", "k5")
                         STEP_A("upA", "upB", "g6", "k6") STEP_A("upB", "upA", "g7", "k7")
                         : [cur] "+v"(cur), [upA] "+v"(upA), [upB] "+v"(upB), [V] "+v"(V), [cnt] "+v"(cnt), [ge] "=&v"(ge),
                           [t] "=&v"(t), [y] "=&v"(y), [k0] "+v"(ks[0]), [k1] "+v"(ks[1]), [k2] "+v"(ks[2]), [k3] "+v"(ks[3]),
                           [k4] "+v"(ks[4]), [k5] "+v"(ks[5]), [k6] "+v"(ks[6]), [k7] "+v"(ks[7])
                         : [g0] "v"(g[0]), [g1] "v"(g[1]), [g2] "v"(g[2]), [g3] "v"(g[3]), [g4] "v"(g[4]), [g5] "v"(g[5]),
                           [g6] "v"(g[6]), [g7] "v"(g[7]), [P] "s"(P), [r3] "s"(r3)
                         : "vcc");
        else if (KIND == 1)
            asm volatile(STEP_B("upA", "upB", "g0", "k0", "9", "0") STEP_B("upB", "upA", "g1", "k1", "10", "0")
                         STEP_B("upA", "upB", "g2", "k2", "11", "0") STEP_B("upB", "upA", "g3", "k3", "12", "0")
                         STEP_B("upA", "upB", "g4", "k4", "13", "0") STEP_B("upB", "upA", "g5", "k5", "14", "0")
                         STEP_B("upA", "upB", "g6", "k6", "15", "0") STEP_B("upB", "upA", "g7", "k7", "16", "0")
                         "s_mov_b64 exec, -1\n\t"
                         : [cur] "+v"(cur), [upA] "+v"(upA), [upB] "+v"(upB), [V] "+v"(V), [t] "=&v"(t), [y] "=&v"(y),
                           [k0] "+v"(ks[0]), [k1] "+v"(ks[1]), [k2] "+v"(ks[2]), [k3] "+v"(ks[3]), [k4] "+v"(ks[4]),
                           [k5] "+v"(ks[5]), [k6] "+v"(ks[6]), [k7] "+v"(ks[7])
                         : [g0] "v"(g[0]), [g1] "v"(g[1]), [g2] "v"(g[2]), [g3] "v"(g[3]), [g4] "v"(g[4]), [g5] "v"(g[5]),
                           [g6] "v"(g[6]), [g7] "v"(g[7]), [r3] "s"(r3)
                         : "exec");
        else if (KIND == 3)
            asm volatile(STEP_B("upA", "upB", "g0", "k0", "56", "0") STEP_B("upB", "upA", "g1", "k1", "57", "0")
                         STEP_B("upA", "upB", "g2", "k2", "58", "0") STEP_B("upB", "upA", "g3", "k3", "59", "0")
                         STEP_B("upA", "upB", "g4", "k4", "60", "0") STEP_B("upB", "upA", "g5", "k5", "61", "0")
                         STEP_B("upA", "upB", "g6", "k6", "62", "0") STEP_B("upB", "upA", "g7", "k7", "63", "0")
                         "s_mov_b64 exec, -1\n\t"
                         : [cur] "+v"(cur), [upA] "+v"(upA), [upB] "+v"(upB), [V] "+v"(V), [t] "=&v"(t), [y] "=&v"(y),
                           [k0] "+v"(ks[0]), [k1] "+v"(ks[1]), [k2] "+v"(ks[2]), [k3] "+v"(ks[3]), [k4] "+v"(ks[4]),
                           [k5] "+v"(ks[5]), [k6] "+v"(ks[6]), [k7] "+v"(ks[7])
                         : [g0] "v"(g[0]), [g1] "v"(g[1]), [g2] "v"(g[2]), [g3] "v"(g[3]), [g4] "v"(g[4]), [g5] "v"(g[5]),
                           [g6] "v"(g[6]), [g7] "v"(g[7]), [r3] "s"(r3)
                         : "exec");
        else if (KIND == 4)
            asm volatile(STEP_B("upA", "upB", "g0", "k0", "24", "0") STEP_B("upB", "upA", "g1", "k1", "25", "0")
                         STEP_B("upA", "upB", "g2", "k2", "26", "0") STEP_B("upB", "upA", "g3", "k3", "27", "0")
                         STEP_B("upA", "upB", "g4", "k4", "28", "0") STEP_B("upB", "upA", "g5", "k5", "29", "0")
                         STEP_B("upA", "upB", "g6", "k6", "30", "0") STEP_B("upB", "upA", "g7", "k7", "31", "0")
                         "s_mov_b64 exec, -1\n\t"
                         : [cur] "+v"(cur), [upA] "+v"(upA), [upB] "+v"(upB), [V] "+v"(V), [t] "=&v"(t), [y] "=&v"(y),
                           [k0] "+v"(ks[0]), [k1] "+v"(ks[1]), [k2] "+v"(ks[2]), [k3] "+v"(ks[3]), [k4] "+v"(ks[4]),
                           [k5] "+v"(ks[5]), [k6] "+v"(ks[6]), [k7] "+v"(ks[7])
                         : [g0] "v"(g[0]), [g1] "v"(g[1]), [g2] "v"(g[2]), [g3] "v"(g[3]), [g4] "v"(g[4]), [g5] "v"(g[5]),
                           [g6] "v"(g[6]), [g7] "v"(g[7]), [r3] "s"(r3)
                         : "exec");
        else
            asm volatile(STEP_C("upA", "upB", "g0", "k0") STEP_C("upB", "upA", "g1", "k1") STEP_C("upA", "upB", "g2", "k2")
                         STEP_C("upB", "upA", "g3", "k3") STEP_C("upA", "upB", "g4", "k4") STEP_C("upB", "upA", "g5", "k5")
                         STEP_C("upA", "upB", "g6", "k6") STEP_C("upB", "upA", "g7", "k7")
                         "s_mov_b64 exec, -1\n\t"
                         : [cur] "+v"(cur), [upA] "+v"(upA), [upB] "+v"(upB), [V] "+v"(V), [t] "=&v"(t), [y] "=&v"(y),
                           [k0] "+v"(ks[0]), [k1] "+v"(ks[1]), [k2] "+v"(ks[2]), [k3] "+v"(ks[3]), [k4] "+v"(ks[4]),
                           [k5] "+v"(ks[5]), [k6] "+v"(ks[6]), [k7] "+v"(ks[7]), [m] "+s"(m)
                         : [g0] "v"(g[0]), [g1] "v"(g[1]), [g2] "v"(g[2]), [g3] "v"(g[3]), [g4] "v"(g[4]), [g5] "v"(g[5]),
                           [g6] "v"(g[6]), [g7] "v"(g[7]), [r3] "s"(r3), [rows] "s"(rows)
                         : "exec", "scc");
    }
    float s = cur + V;
    for (int u = 0; u < 8; ++u) s += ks[u];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
void run(float *d, int ncu, const char *name)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    hipLaunchKernelGGL(k<KIND>, dim3(ncu), dim3(512), 0, 0, d, 100, 63);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(ncu), dim3(512), 0, 0, d, iters, 63);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // each SIMD runs 2 waves; a "step" here is one step of BOTH waves
    printf("%-60s %.3f ms   %.2f ns per step of one wave (2 waves/SIMD)\n", name, ms, ms * 1e6 / (iters * 8.0) / 2.0);
}

int main()
{
    setvbuf(stdout, nullptr, _IOLBF, 0);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    float *d;
    hipMalloc(&d, (size_t)prop.multiProcessorCount * 1024 * sizeof(float));
    run<0>(d, prop.multiProcessorCount, "A: v_cmp + 2 cndmask (11 VALU)");
    run<1>(d, prop.multiProcessorCount, "B: EXEC window by s_bfm_b64 (8 VALU + 2 SALU)");
    run<2>(d, prop.multiProcessorCount, "C: EXEC window from a shifted SGPR pair (8 VALU + 3 SALU)");
    run<3>(d, prop.multiProcessorCount, "B with 56..63 lanes in the window");
    run<4>(d, prop.multiProcessorCount, "B with 24..31 lanes in the window");
    run<0>(d, prop.multiProcessorCount, "A again");
    runq<0, 512>(d, prop.multiProcessorCount, "Q0: window from s_lshr + s_and (8 VALU + 3 SALU)");
    runq<1, 512>(d, prop.multiProcessorCount, "Q1: Q0 + v_readlane / v_mov boundary value (10 VALU)");
    runq<2, 512>(d, prop.multiProcessorCount, "Q2: Q0 + ds_write hand-over + address update (9 VALU + 1 LDS)");
    runq<4, 512>(d, prop.multiProcessorCount, "Q4: Q0 + ds_write hand-over only (8 VALU + 1 LDS)");
    runq<3, 512>(d, prop.multiProcessorCount, "Q3: gram_quad's step (11 VALU + 3 SALU + 1 LDS)");
    runq<0, 768>(d, prop.multiProcessorCount, "Q0, three waves per SIMD");
    runq<0, 1024>(d, prop.multiProcessorCount, "Q0, four waves per SIMD");
    runq<3, 768>(d, prop.multiProcessorCount, "Q3, three waves per SIMD");
    runq<3, 1024>(d, prop.multiProcessorCount, "Q3, four waves per SIMD");
    runq<0, 256>(d, prop.multiProcessorCount, "Q0, one wave per SIMD");
    runq<3, 256>(d, prop.multiProcessorCount, "Q3, one wave per SIMD");
    return 0;
}
