// extern "C" entry points of libsigsvgd_hip.so (declared in include/sigsvgd_hip.h).
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <dlfcn.h>

#include "sig_common.h"

namespace sigsvgd {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char *what)
{
    set_error("%s: %s", what, hipGetErrorString(e));
    return SIGSVGD_E_HIP;
}

int phi_launch(const float *K, const float *score, const float *grad_k, const float *mask, int N, int D,
               float *v_out, const float *X_in, float *X_out, float lr, float *adagrad, hipStream_t stream,
               float *exp_avg = nullptr, float *exp_avg_sq = nullptr, int *step_dev = nullptr, double lr_adam = 0.0,
               double beta1 = 0.0, double beta2 = 0.0, float eps = 0.f);

int vec_sqdist_launch(const void *X, const void *Y, const void *XM, const void *YM, int A, int B, int D, int dtype,
                      void *sq, hipStream_t stream);
int vec_kgrad_launch(const void *sq, const void *XM, const void *YM, const void *go, int A, int B, int D, int dtype,
                     int kind, double inv_h2, double grad_scale, void *K, void *dK, hipStream_t stream);
bool vec_fused_supported(int D, int dtype);
int vec_fused_launch(const void *X, const void *Y, const void *XM, const void *YM, const void *go, int A, int B, int D,
                     int kind, double inv_h2, double grad_scale, void *K, void *dK, void *ws, size_t ws_bytes, hipStream_t stream);
size_t vec_fused_workspace_bytes(int A, int B, int D);
long long signature_channels(int C, int depth);
int obstacle_cost_launch(const float *x, int N, int Kx, int d, const float *start, const float *target, const float *basis,
                         int Tt, const float *logw, const float *mean, const float *stdv, int M, float w_obst, float w_len,
                         float *cost, float *traj, float *grad_x, hipStream_t stream);
int signature_launch(const void *X, int N, int L, int C, int depth, int basepoint, int dtype, void *out,
                     hipStream_t stream);
int signature_bwd_launch(const void *X, const void *gsig, int N, int L, int C, int depth, int basepoint, int dtype, void *gX,
                         long long sigdim, hipStream_t stream);

static int check_common(const void *X, const void *Y, int A, int B, int T, int d, int dtype, double inv_h,
                        int n, int kind, const void *K_out)
{
    if (!X || !Y || !K_out) {
        set_error("null pointer argument");
        return SIGSVGD_E_BADARG;
    }
    if (A < 1 || B < 1 || T < 2 || d < 1) {
        set_error("bad shape A=%d B=%d T=%d d=%d (need A,B,d >= 1 and T >= 2)", A, B, T, d);
        return SIGSVGD_E_BADARG;
    }
    if (dtype != SIGSVGD_F32 && dtype != SIGSVGD_F64) {
        set_error("bad dtype %d", dtype);
        return SIGSVGD_E_BADARG;
    }
    if (kind != SIGSVGD_STATIC_RBF && kind != SIGSVGD_STATIC_LINEAR) {
        set_error("bad static kernel kind %d", kind);
        return SIGSVGD_E_BADARG;
    }
    if (n < 0 || n > 10) {
        set_error("bad dyadic order %d", n);
        return SIGSVGD_E_BADARG;
    }
    if (kind == SIGSVGD_STATIC_RBF && !(inv_h > 0.0)) {
        set_error("RBF static kernel needs inv_h > 0 (got %g)", inv_h);
        return SIGSVGD_E_BADARG;
    }
    return SIGSVGD_OK;
}

// ---- roctx ranges around the launches (the tracing hook of this path) ------------------------------
// Enabled with SIGSVGD_ROCTX=1: libroctx64.so is looked up at run time (no link-time dependency), every entry
// point that enqueues work brackets its launches with roctxRangePush/Pop, so `rocprofv3 --marker-trace` shows
// which API call a kernel belongs to.  Off by default: zero cost beyond one branch.
namespace {
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx()
    {
        const char *on = getenv("SIGSVGD_ROCTX");
        if (!on || on[0] == '0') return;
        void *h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("/opt/rocm/lib/libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (!push || !pop) push = nullptr, pop = nullptr;
    }
};
struct Range {
    const Roctx &r;
    explicit Range(const char *name) : r(instance())
    {
        if (r.push) r.push(name);
    }
    ~Range()
    {
        if (r.push) r.pop();
    }
    static const Roctx &instance()
    {
        static Roctx x;
        return x;
    }
};
} // namespace

static int dispatch(const GramProblem &p)
{
    const int want_grad = p.gradX_out != nullptr;
    (void)want_grad;
    if (!(p.flags & SIGSVGD_FLAG_FORCE_GENERIC) && fast_supported(p.A, p.B, p.T, p.d, p.n, p.kind, p.flags))
        return fast_launch(p);
    // long paths (65 <= T <= 128): the quadrant kernel (stored forward solution, any roughness)
    if (!(p.flags & SIGSVGD_FLAG_FORCE_GENERIC) && quad_supported(p.A, p.B, p.T, p.d, p.n, p.kind, p.flags))
        return quad_launch(p);
    // short paths with dyadic refinement whose refined grid has 64 .. 128 cells per side (the reference's own call shapes)
    if (!(p.flags & SIGSVGD_FLAG_FORCE_GENERIC) && dyad_supported(p.A, p.B, p.T, p.d, p.n, p.kind, p.flags))
        return band_takes_refined(p) ? band_launch(p) : dyad_launch(p); // (small launches of 65 .. 128 cells: one wavefront per band)
    if (!(p.flags & SIGSVGD_FLAG_FORCE_GENERIC) && band_supported(p.A, p.B, p.T, p.d, p.n, p.kind, p.flags))
        return band_launch(p);
    return generic_launch(p);
}

} // namespace sigsvgd

using namespace sigsvgd;

extern "C" {

int sigsvgd_abi_version(void) { return SIGSVGD_ABI_VERSION; }

const char *sigsvgd_last_error(void) { return g_err; }

int sigsvgd_gram_workspace_bytes(int A, int B, int T, int d, int dyadic_order, int static_kind, int want_grad,
                                 unsigned flags, size_t *bytes)
{
    if (!bytes) {
        set_error("bytes == NULL");
        return SIGSVGD_E_BADARG;
    }
    if (static_kind != SIGSVGD_STATIC_RBF && static_kind != SIGSVGD_STATIC_LINEAR) {
        set_error("bad static kernel kind %d", static_kind);
        return SIGSVGD_E_BADARG;
    }
    // size for the kernel dispatch() picks: the same predicates on the same arguments
    const bool forced = (flags & SIGSVGD_FLAG_FORCE_GENERIC) != 0;
    if (!forced && fast_supported(A, B, T, d, dyadic_order, static_kind, flags))
        return fast_workspace_bytes(A, B, T, d, want_grad, flags, bytes);
    if (!forced && quad_supported(A, B, T, d, dyadic_order, static_kind, flags))
        return quad_workspace_bytes(A, B, T, d, want_grad, bytes);
    if (!forced && dyad_supported(A, B, T, d, dyadic_order, static_kind, flags)) {
        const int rc = dyad_workspace_bytes(A, B, T, d, want_grad, bytes);
        if (rc == SIGSVGD_OK && ((T - 1) << dyadic_order) >= 64 && dyadic_order >= 2) { // either kernel may take the launch
            const size_t pb = band_refined_workspace_bytes(A, B, T, d, dyadic_order, want_grad, flags);
            if (pb > *bytes) *bytes = pb;
        }
        return rc;
    }
    if (!forced && band_supported(A, B, T, d, dyadic_order, static_kind, flags))
        return band_workspace_bytes(A, B, T, d, dyadic_order, want_grad, flags, bytes);
    return generic_workspace_bytes(A, B, T, d, dyadic_order, want_grad, forced, bytes);
}

int sigsvgd_gram_fwd(const void *X, const void *Y, int A, int B, int T, int d, int dtype, double inv_h,
                     int dyadic_order, int static_kind, unsigned flags, void *K_out, void *workspace,
                     size_t workspace_bytes, void *stream)
{
    int rc = check_common(X, Y, A, B, T, d, dtype, inv_h, dyadic_order, static_kind, K_out);
    if (rc) return rc;
    GramProblem p{X, Y, A, B, T, d, dtype, inv_h, dyadic_order, static_kind, flags, nullptr,
                  K_out, nullptr, workspace, workspace_bytes, static_cast<hipStream_t>(stream)};
    Range range("sigsvgd_gram_fwd");
    return dispatch(p);
}

int sigsvgd_gram_fwd_bwd(const void *X, const void *Y, int A, int B, int T, int d, int dtype, double inv_h,
                         int dyadic_order, int static_kind, unsigned flags, const void *grad_out, void *K_out,
                         void *gradX_out, void *workspace, size_t workspace_bytes, void *stream)
{
    int rc = check_common(X, Y, A, B, T, d, dtype, inv_h, dyadic_order, static_kind, K_out);
    if (rc) return rc;
    if (!gradX_out) {
        set_error("gradX_out == NULL (use sigsvgd_gram_fwd for forward only)");
        return SIGSVGD_E_BADARG;
    }
    if ((flags & SIGSVGD_FLAG_Y_IS_X) && A != B) {
        set_error("Y_IS_X needs A == B");
        return SIGSVGD_E_BADARG;
    }
    GramProblem p{X, Y, A, B, T, d, dtype, inv_h, dyadic_order, static_kind, flags, grad_out,
                  K_out, gradX_out, workspace, workspace_bytes, static_cast<hipStream_t>(stream)};
    Range range("sigsvgd_gram_fwd_bwd");
    return dispatch(p);
}

int sigsvgd_gram_sym_partial(const void *X, int N, int T, int d, int dtype, double inv_h, int static_kind,
                             unsigned flags, int tile_offset, int tile_stride, const void *grad_out,
                             void *K_partial, double *grad_partial, void *workspace, size_t workspace_bytes,
                             void *stream)
{
    int rc = check_common(X, X, N, N, T, d, dtype, inv_h, 0, static_kind, K_partial);
    if (rc) return rc;
    if (!grad_partial) {
        set_error("grad_partial == NULL");
        return SIGSVGD_E_BADARG;
    }
    GramProblem p{X, X, N, N, T, d, dtype, inv_h, 0, static_kind, flags | SIGSVGD_FLAG_Y_IS_X, grad_out,
                  K_partial, grad_partial, workspace, workspace_bytes, static_cast<hipStream_t>(stream)};
    Range range("sigsvgd_gram_sym_partial");
    if (!fast_supported(N, N, T, d, 0, static_kind, flags) && quad_supported(N, N, T, d, 0, static_kind, flags))
        return quad_sym_partial(p, tile_offset, tile_stride, (flags & SIGSVGD_FLAG_FOLD_TILES) != 0, grad_partial);
    return fast_sym_partial(p, tile_offset, tile_stride, (flags & SIGSVGD_FLAG_FOLD_TILES) != 0, grad_partial);
}

int sigsvgd_gram_sym_tile_rows(int T, int d)
{
    if (fast_supported(1, 1, T, d, 0, SIGSVGD_STATIC_RBF, 0)) return sym_tile_rows_fast(T, d);
    if (quad_supported(1, 1, T, d, 0, SIGSVGD_STATIC_RBF, 0)) return 8;
    return 0;
}

int sigsvgd_svgd_phi(const float *K, const float *score, const float *grad_k, const float *mask, int N, int D,
                     float *v_out, const float *X_in, float *X_out, float lr, void *stream)
{
    Range range("sigsvgd_svgd_phi");
    return phi_launch(K, score, grad_k, mask, N, D, v_out, X_in, X_out, lr, nullptr, static_cast<hipStream_t>(stream));
}

int sigsvgd_svgd_step(const float *K, const float *score, const float *grad_k, const float *mask, int N, int D,
                      float *v_out, const float *X_in, float *X_out, float lr, float *adagrad_state, void *stream)
{
    Range range("sigsvgd_svgd_step");
    return phi_launch(K, score, grad_k, mask, N, D, v_out, X_in, X_out, lr, adagrad_state,
                      static_cast<hipStream_t>(stream));
}

int sigsvgd_svgd_adam_step(const float *K, const float *score, const float *grad_k, const float *mask, int N, int D,
                           float *v_out, const float *X_in, float *X_out, double lr, double beta1, double beta2, double eps,
                           float *exp_avg, float *exp_avg_sq, int *step_dev, void *stream)
{
    if (!exp_avg || !exp_avg_sq || !step_dev || !X_in || !X_out) {
        set_error("svgd_adam_step: null state / particle pointer");
        return SIGSVGD_E_BADARG;
    }
    if (!(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0)) {
        set_error("svgd_adam_step: bad hyper-parameters beta1=%g beta2=%g eps=%g", beta1, beta2, eps);
        return SIGSVGD_E_BADARG;
    }
    Range range("sigsvgd_svgd_adam_step");
    return phi_launch(K, score, grad_k, mask, N, D, v_out, X_in, X_out, (float)lr, nullptr,
                      static_cast<hipStream_t>(stream), exp_avg, exp_avg_sq, step_dev, lr, beta1, beta2, (float)eps);
}

int sigsvgd_vec_sqdist(const void *X, const void *Y, const void *XM, const void *YM, int A, int B, int D, int dtype,
                       void *sq_out, void *stream)
{
    if (!X || !Y || !sq_out || (XM == nullptr) != (YM == nullptr)) {
        set_error("vec_sqdist: null pointer argument (XM and YM must both be given or both be NULL)");
        return SIGSVGD_E_BADARG;
    }
    if (A < 1 || B < 1 || D < 1 || (dtype != SIGSVGD_F32 && dtype != SIGSVGD_F64)) {
        set_error("vec_sqdist: bad arguments A=%d B=%d D=%d dtype=%d", A, B, D, dtype);
        return SIGSVGD_E_BADARG;
    }
    return vec_sqdist_launch(X, Y, XM, YM, A, B, D, dtype, sq_out, static_cast<hipStream_t>(stream));
}

int sigsvgd_vec_fused_workspace_bytes(int A, int B, int D, size_t *bytes)
{
    if (!bytes || A < 1 || B < 1 || D < 1) {
        set_error("vec_fused_workspace_bytes: bad arguments A=%d B=%d D=%d", A, B, D);
        return SIGSVGD_E_BADARG;
    }
    *bytes = vec_fused_workspace_bytes(A, B, D);
    return SIGSVGD_OK;
}

int sigsvgd_vec_kernel_fused(const void *X, const void *Y, const void *XM, const void *YM, const void *grad_out, int A,
                             int B, int D, int dtype, int kind, double inv_h2, double grad_scale, void *K_out,
                             void *dK_out, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!X || !Y || (!K_out && !dK_out) || (XM == nullptr) != (YM == nullptr)) {
        set_error("vec_kernel_fused: null pointer argument (XM and YM must both be given or both be NULL)");
        return SIGSVGD_E_BADARG;
    }
    if (A < 1 || B < 1 || D < 1 || (dtype != SIGSVGD_F32 && dtype != SIGSVGD_F64)) {
        set_error("vec_kernel_fused: bad arguments A=%d B=%d D=%d dtype=%d", A, B, D, dtype);
        return SIGSVGD_E_BADARG;
    }
    if (kind != SIGSVGD_VEC_GAUSSIAN && kind != SIGSVGD_VEC_IMQ && kind != SIGSVGD_VEC_UNIT) {
        set_error("vec_kernel_fused: bad kind %d", kind);
        return SIGSVGD_E_BADARG;
    }
    if (!vec_fused_supported(D, dtype)) {
        set_error("vec_kernel_fused: fp32 with D <= 512 only (got dtype=%d D=%d); use sigsvgd_vec_sqdist + sigsvgd_vec_kernel",
                  dtype, D);
        return SIGSVGD_E_UNSUPPORTED;
    }
    Range range("sigsvgd_vec_kernel_fused");
    return vec_fused_launch(X, Y, XM, YM, grad_out, A, B, D, kind, inv_h2, grad_scale, K_out, dK_out, workspace, workspace_bytes,
                            static_cast<hipStream_t>(stream));
}

int sigsvgd_vec_kernel(const void *sq, const void *XM, const void *YM, const void *grad_out, int A, int B, int D,
                       int dtype, int kind, double inv_h2, double grad_scale, void *K_out, void *dK_out, void *stream)
{
    if (!sq || (!K_out && !dK_out) || (dK_out && (!XM || !YM))) {
        set_error("vec_kernel: null pointer argument");
        return SIGSVGD_E_BADARG;
    }
    if (A < 1 || B < 1 || D < 1 || (dtype != SIGSVGD_F32 && dtype != SIGSVGD_F64)) {
        set_error("vec_kernel: bad arguments A=%d B=%d D=%d dtype=%d", A, B, D, dtype);
        return SIGSVGD_E_BADARG;
    }
    if (kind != SIGSVGD_VEC_GAUSSIAN && kind != SIGSVGD_VEC_IMQ && kind != SIGSVGD_VEC_UNIT) {
        set_error("vec_kernel: bad kind %d", kind);
        return SIGSVGD_E_BADARG;
    }
    if (kind == SIGSVGD_VEC_UNIT && (K_out || !dK_out)) {
        set_error("vec_kernel: SIGSVGD_VEC_UNIT computes only dK_out (K_out must be NULL)");
        return SIGSVGD_E_BADARG;
    }
    if (kind != SIGSVGD_VEC_UNIT && !(inv_h2 > 0.0)) {
        set_error("vec_kernel: needs 1/h^2 > 0 (got %g)", inv_h2);
        return SIGSVGD_E_BADARG;
    }
    return vec_kgrad_launch(sq, XM, YM, grad_out, A, B, D, dtype, kind, inv_h2, grad_scale, K_out, dK_out,
                            static_cast<hipStream_t>(stream));
}

int sigsvgd_obstacle_cost(const float *x, int N, int knots, int d, const float *start, const float *target,
                          const float *basis, int samples, const float *log_weights, const float *mean, const float *std,
                          int components, float w_obstacle, float w_length, float *cost, float *traj, float *grad_x,
                          void *stream)
{
    Range range("sigsvgd_obstacle_cost");
    return obstacle_cost_launch(x, N, knots, d, start, target, basis, samples, log_weights, mean, std, components,
                                w_obstacle, w_length, cost, traj, grad_x, static_cast<hipStream_t>(stream));
}

int sigsvgd_signature(const void *X, int N, int L, int C, int depth, int basepoint, int dtype, void *out,
                      long long *channels, void *stream)
{
    if (N < 1 || L < 1 || C < 1 || depth < 1 || (dtype != SIGSVGD_F32 && dtype != SIGSVGD_F64)) {
        set_error("signature: bad arguments N=%d L=%d C=%d depth=%d dtype=%d", N, L, C, depth, dtype);
        return SIGSVGD_E_BADARG;
    }
    const long long ch = signature_channels(C, depth);
    if (ch < 0) {
        set_error("signature: C=%d depth=%d overflows", C, depth);
        return SIGSVGD_E_UNSUPPORTED;
    }
    if (channels) *channels = ch;
    if (!out) {
        if (channels) return SIGSVGD_OK;
        set_error("signature: out == NULL and channels == NULL");
        return SIGSVGD_E_BADARG;
    }
    if (!X) {
        set_error("signature: X == NULL");
        return SIGSVGD_E_BADARG;
    }
    return signature_launch(X, N, L, C, depth, basepoint, dtype, out, static_cast<hipStream_t>(stream));
}

int sigsvgd_signature_backward(const void *X, const void *grad_sig, int N, int L, int C, int depth, int basepoint, int dtype,
                               void *grad_X, void *stream)
{
    if (N < 1 || L < 1 || C < 1 || depth < 1 || (dtype != SIGSVGD_F32 && dtype != SIGSVGD_F64)) {
        set_error("signature_backward: bad arguments N=%d L=%d C=%d depth=%d dtype=%d", N, L, C, depth, dtype);
        return SIGSVGD_E_BADARG;
    }
    if (!X || !grad_sig || !grad_X) {
        set_error("signature_backward: null pointer argument");
        return SIGSVGD_E_BADARG;
    }
    const long long ch = signature_channels(C, depth);
    if (ch < 0) {
        set_error("signature_backward: C=%d depth=%d overflows", C, depth);
        return SIGSVGD_E_UNSUPPORTED;
    }
    Range range("sigsvgd_signature_backward");
    return signature_bwd_launch(X, grad_sig, N, L, C, depth, basepoint, dtype, grad_X, ch, static_cast<hipStream_t>(stream));
}

} // extern "C"
