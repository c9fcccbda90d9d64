// Shared device helpers for the signature-kernel HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sigsvgd_hip.h"

namespace sigsvgd {

constexpr int kWave = 64;

// ---- error plumbing (host) -------------------------------------------------------------------
void set_error(const char *fmt, ...);
int hip_fail(hipError_t e, const char *what);

// ---- fp64 exp -----------------------------------------------------------------------------
// exp(a) = 2^k * P(r), k = rint(a*log2e), r = a - k*ln2 (two-word ln2), P = degree-11 Taylor/Horner
// on |r| <= ln2/2.  Relative error < 3e-16 * few; enough for the 4-corner cancellation in the
// kernel increments (needs ~1e-10).  One v_rndne_f64 + v_cvt_i32_f64 + v_ldexp_f64 + 14 FMA/MUL.
__device__ __forceinline__ double exp64(double a0)
{
    double a = fmin(fmax(a0, -1000.0), 700.0); // (fmax / fmin drop a NaN: it is put back at the end)
    const double kf = __builtin_rint(a * 1.4426950408889634074);
    double r = __builtin_fma(kf, -6.93147180369123816490e-01, a);
    r = __builtin_fma(kf, -1.90821492927058770002e-10, r);
    double p = 2.50521083854417187751e-08;              // 1/11!
    p = __builtin_fma(p, r, 2.75573192239858906526e-07); // 1/10!
    p = __builtin_fma(p, r, 2.75573192239858906526e-06); // 1/9!
    p = __builtin_fma(p, r, 2.48015873015873015873e-05); // 1/8!
    p = __builtin_fma(p, r, 1.98412698412698412698e-04); // 1/7!
    p = __builtin_fma(p, r, 1.38888888888888888889e-03); // 1/6!
    p = __builtin_fma(p, r, 8.33333333333333333333e-03); // 1/5!
    p = __builtin_fma(p, r, 4.16666666666666666667e-02); // 1/4!
    p = __builtin_fma(p, r, 1.66666666666666666667e-01); // 1/3!
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    const double e = ldexp(p, (int)kf);
    return (a0 != a0) ? a0 : e; // a NaN in a path poisons its own row / column of K, as in the reference
}

// 2^t for the register-resident kernel: t arrives already scaled by log2(e) (the scale is folded into
// the particle coordinates), so the reduction is exact: k = rint(t), f = t - k in [-1/2, 1/2], and
// 2^f is a degree-8 Chebyshev-interpolant (max relative error 1.1e-12, measured on 2e4 points;
// the 4-corner increments need ~1e-10).  12 fp64-rate instructions.
__device__ __forceinline__ double exp2_p8(double t)
{
    const double kf = __builtin_rint(t);
    const double f = t - kf;
    double p = 1.3255197199834888e-06;
    p = __builtin_fma(p, f, 1.5310079063632544e-05);
    p = __builtin_fma(p, f, 1.5403455940423864e-04);
    p = __builtin_fma(p, f, 1.3333450569733936e-03);
    p = __builtin_fma(p, f, 9.618129159303683e-03);
    p = __builtin_fma(p, f, 5.550410941203932e-02);
    p = __builtin_fma(p, f, 2.4022650695813685e-01);
    p = __builtin_fma(p, f, 6.931471805459342e-01);
    p = __builtin_fma(p, f, 0.9999999999999997);
    return ldexp(p, (int)kf);
}

// Degree-7 variant (max relative error 5.5e-11 on [-1/2, 1/2]) for the register-resident kernel, whose
// increments are rounded to fp32 (6e-8 relative) right after: one Horner step fewer per static-kernel value.
__device__ __forceinline__ double exp2_p7(double t)
{
    const double kf = __builtin_rint(t);
    const double f = t - kf;
    double p = 1.5303701161442145e-05;
    p = __builtin_fma(p, f, 1.5469729221575116e-04);
    p = __builtin_fma(p, f, 1.3333478471058548e-03);
    p = __builtin_fma(p, f, 9.618025613268967e-03);
    p = __builtin_fma(p, f, 5.5504109063307244e-02);
    p = __builtin_fma(p, f, 2.4022651213498578e-01);
    p = __builtin_fma(p, f, 6.931471805568296e-01);
    p = __builtin_fma(p, f, 0.9999999999595621);
    return ldexp(p, (int)kf);
}

// Same polynomial with the coefficients handed in by the caller, who keeps them in scalar registers inside
// a rolled loop (empty `asm volatile("" : "+s"(c))` per iteration): hipcc otherwise hoists them into
// VGPRs and pays one v_mov_b64 per Horner step to feed v_fmac_f64.
struct Exp2Coef {
    double c8, c7, c6, c5, c4, c3, c2, c1, c0;
};
__device__ __forceinline__ Exp2Coef exp2_coef()
{
    return Exp2Coef{1.3255197199834888e-06, 1.5310079063632544e-05, 1.5403455940423864e-04,
                    1.3333450569733936e-03, 9.618129159303683e-03,  5.550410941203932e-02,
                    2.4022650695813685e-01, 6.931471805459342e-01,  0.9999999999999997};
}
__device__ __forceinline__ void exp2_coef_pin(Exp2Coef &k)
{
    asm volatile("" : "+s"(k.c8), "+s"(k.c7), "+s"(k.c6), "+s"(k.c5), "+s"(k.c4), "+s"(k.c3), "+s"(k.c2), "+s"(k.c1),
                 "+s"(k.c0));
}
__device__ __forceinline__ double exp2_p8(double t, const Exp2Coef &k)
{
    const double kf = __builtin_rint(t);
    const double f = t - kf;
    double p = __builtin_fma(k.c8, f, k.c7);
    p = __builtin_fma(p, f, k.c6);
    p = __builtin_fma(p, f, k.c5);
    p = __builtin_fma(p, f, k.c4);
    p = __builtin_fma(p, f, k.c3);
    p = __builtin_fma(p, f, k.c2);
    p = __builtin_fma(p, f, k.c1);
    p = __builtin_fma(p, f, k.c0);
    return ldexp(p, (int)kf);
}

// ---- Goursat stencils -------------------------------------------------------------------------
// default (second order):  K11 = (K10 + K01)*(1 + g/2 + g^2/12) - K00*(1 - g^2/12)
// written in delta form so that the O(1) parts cancel exactly in fp64:
//   K11 = (t - K00) + t*a + K00*b,  t = K10 + K01, a = g/2 + g^2/12, b = g^2/12
// naive (first order):     K11 = K10 + K01 + K00*(g - 1)
__device__ __forceinline__ double stencil(double k10, double k01, double k00, double g, bool naive)
{
    const double t = k10 + k01;
    if (naive) return __builtin_fma(k00, g, t - k00);
    const double b = g * g * (1.0 / 12.0);
    const double a = __builtin_fma(g, 0.5, b);
    double u = t - k00;
    u = __builtin_fma(t, a, u);
    return __builtin_fma(k00, b, u);
}

template <typename T>
__device__ __forceinline__ double ld_as_f64(const T *p, size_t i)
{
    return (double)p[i];
}

template <typename T>
__device__ __forceinline__ void st_from_f64(T *p, size_t i, double v)
{
    p[i] = (T)v;
}

// Neighbour-lane moves of a double as two DPP moves (wave_shr:1 / wave_shl:1; the lane without a source keeps its own
// value).  __shfl_up / __shfl_down lower to ds_bpermute: an LDS round trip (~100 cycles) on the dependent chain of
// every PDE step, against ~8 cycles here.
__device__ __forceinline__ double shfl_up_f64(double v)   // lane l <- lane l-1 (lane 0 keeps own)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), 0x138, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), 0x138, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shfl_down_f64(double v) // lane l <- lane l+1 (lane 63 keeps own)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), 0x130, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), 0x130, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

// ---- EXEC discipline of the hand-written sweep statements ----------------------------------------------------------------
// The sweep statements (gram_fast.hip sweep_fwd8 / sweep_rev8, quad_sweeps.h) move lane windows into EXEC and leave it at
// all ones.  EXEC is a reserved register for hipcc: a clobber on it is ignored with a warning ("inline asm clobber list
// contains reserved registers"), so the compiler cannot be told.  The statements are therefore only correct where the
// compiler's own EXEC is all ones too, i.e. in wave-uniform control flow -- which every call site is by construction (the
// conditions around them are functions of kernel arguments, block and wavefront indices).  -DSIGSVGD_CHECK_EXEC turns that
// into a run-time check: each statement group first compares EXEC with all ones and records a violation in a device-side
// sticky word per translation unit (scripts/dev/check_exec.py builds that variant, runs the kernel families through it and
// reads the words back; profiles/r04_exec_check.txt).
#ifdef SIGSVGD_CHECK_EXEC
static __device__ unsigned g_exec_violations; // one per translation unit; read back by sigsvgd_debug_exec_violations_<unit>()
#define SIG_EXEC_MUST_BE_FULL(what)                                                        \
    do {                                                                                   \
        if (__builtin_amdgcn_read_exec() != ~0ull) g_exec_violations = 1u;                 \
    } while (0)
#define SIG_EXEC_DEBUG_GETTER(unit)                                                        \
    extern "C" unsigned sigsvgd_debug_exec_violations_##unit(void)                         \
    {                                                                                      \
        unsigned v = 0xffffffffu;                                                          \
        (void)hipDeviceSynchronize();                                                      \
        (void)hipMemcpyFromSymbol(&v, HIP_SYMBOL(sigsvgd::g_exec_violations), sizeof(v));  \
        return v;                                                                          \
    }
#else
#define SIG_EXEC_MUST_BE_FULL(what)
#define SIG_EXEC_DEBUG_GETTER(unit)
#endif

// ---- launch descriptors shared by host code -----------------------------------------------------
struct GramProblem {
    const void *X, *Y;
    int A, B, T, d, dtype;
    double inv_h;
    int n;            // dyadic order
    int kind;         // SIGSVGD_STATIC_*
    unsigned flags;
    const void *grad_out; // nullable
    void *K_out;
    void *gradX_out;  // nullable => forward only
    void *ws;
    size_t ws_bytes;
    hipStream_t stream;
};

// generic (any T, n, d that fits LDS) -- gram_generic.hip
int generic_workspace_bytes(int A, int B, int T, int d, int n, int want_grad, bool precise, size_t *bytes); // precise: FORCE_GENERIC
int generic_launch(const GramProblem &p);

// fixed-order reduction of the gradient partial sums shared by the register-resident and the quadrant kernel -- gram_fast.hip
// Which row tiles a launch owns and the order it enumerates them in (kq = 0 .. owned-1).  A full launch owns all of
// them (off 0, stride 1).  The sharded partial solve of rank `off` of `stride` owns the tiles off + k*stride (cyclic), or
// -- SIGSVGD_FLAG_FOLD_TILES -- those AND their mirror images ntile-1 - (off + k*stride): in the upper triangle tile t
// has B - t*NW columns, so a tile and its mirror image together always cost the same and every rank gets the same
// number of items (cyclic ownership alone gives rank 0 5.4 % more than the mean at N=1024 on 8 ranks).
struct TileMap {
    int off, stride, ntile, owned, m0, fold; // m0 tiles of the first kind (off + k*stride), then owned - m0 mirror images
    __host__ __device__ int tile_of(int kq) const
    {
        return kq < m0 ? off + kq * stride : ntile - 1 - (off + (kq - m0) * stride);
    }
    __host__ __device__ int kq_of_tile(int ti) const // -1: not owned
    {
        if (ti >= off && (ti - off) % stride == 0 && (ti - off) / stride < m0) return (ti - off) / stride;
        const int p = ntile - 1 - ti;
        if (p >= off && (p - off) % stride == 0 && (p - off) / stride < owned - m0) return m0 + (p - off) / stride;
        return -1;
    }
    // items of the owned tiles 0 .. kq-1 in the tile-major enumeration of the kernels: symmetric launches count the
    // columns from the tile's first row on (B - tile*NW), ordered ones all B
    __host__ __device__ long long start(int kq, int B, int NW, int sym) const
    {
        if (!sym) return (long long)kq * B;
        const long long k1 = kq < m0 ? kq : m0, k2 = kq - k1;
        long long s = k1 * B - (long long)NW * ((long long)stride * k1 * (k1 - 1) / 2 + (long long)off * k1);
        s += k2 * ((long long)B - (long long)(ntile - 1 - off) * NW) + (long long)NW * stride * k2 * (k2 - 1) / 2;
        return s;
    }
};
TileMap make_tilemap(int ntile, int off, int stride, bool fold);

struct GradGeom {
    int NW, grid;
    TileMap tm;
    long long nitems;
    size_t rseg_bytes, cslab_bytes;
};
int device_cu_count();
GradGeom grad_geometry(int A, int B, int TD, bool sym, int off, int stride, bool fold, int NW, long long resident);
int grad_reduce_launch(const GradGeom &g, const double *rseg, const float *cslab, void *out, int out64, int A, int B, int TD,
                       bool sym, hipStream_t stream);

// fp64 pass of the coverage kernel over the pairs a fp32-sweep kernel flagged (flags [A][B] bytes; `sym`: the pairs j >= i,
// K mirrored; rows of the tiles `tm` owns, `tile_rows` rows each; ws: generic_repair_bytes() of scratch) -- gram_generic.hip
size_t generic_repair_bytes();
int generic_repair_launch(const GramProblem &p, const unsigned char *flags, void *ws, bool sym, const TileMap &tm, int tile_rows);

// register-resident fast path (n == 0, T <= 64, RBF/linear) -- gram_fast.hip
bool fast_supported(int A, int B, int T, int d, int n, int kind, unsigned flags);
int fast_workspace_bytes(int A, int B, int T, int d, int want_grad, unsigned flags, size_t *bytes);
int fast_launch(const GramProblem &p);
int fast_sym_partial(const GramProblem &p, int tile_offset, int tile_stride, bool fold, double *grad_partial);
int sym_tile_rows_fast(int T, int d); // rows per tile of the gradient launches (ownership unit of the partial solve)

// long paths, stored forward solution, 2 x 2 quadrants of 64 x 64 cells at two waves per SIMD -- gram_quad.hip
bool quad_supported(int A, int B, int T, int d, int n, int kind, unsigned flags);
int quad_workspace_bytes(int A, int B, int T, int d, int want_grad, size_t *bytes);
int quad_launch(const GramProblem &p);
int quad_sym_partial(const GramProblem &p, int tile_offset, int tile_stride, bool fold, double *grad_partial);

// short paths with dyadic refinement, refined grid of 64 .. 128 cells per side, on the quadrant sweep engine -- gram_dyad.hip
bool dyad_supported(int A, int B, int T, int d, int n, int kind, unsigned flags);
int dyad_workspace_bytes(int A, int B, int T, int d, int want_grad, size_t *bytes);
int dyad_launch(const GramProblem &p);

// the same with 129 .. 256 refined cells per side, swept in bands of 64 rows (fp32 difference form) -- gram_band.hip
bool band_supported(int A, int B, int T, int d, int n, int kind, unsigned flags);
int band_workspace_bytes(int A, int B, int T, int d, int n, int want_grad, unsigned flags, size_t *bytes);
int band_launch(const GramProblem &p);
bool band_takes_refined(const GramProblem &p); // 65 .. 128 cells, small launches: the band-parallel kernel instead of gram_dyad.hip
size_t band_refined_workspace_bytes(int A, int B, int T, int d, int n, int want_grad, unsigned flags);

} // namespace sigsvgd
