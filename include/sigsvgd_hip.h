/*
 * sigsvgd_hip.h -- C ABI of libsigsvgd_hip.so: the MI355X (gfx950) implementation of the
 * signature-kernel SVGD hot path of lubaroli/sigsvgd.
 *
 * What each entry point replaces in the reference (Python; there is no native code there):
 *
 *   sigsvgd_gram_fwd       sigkernel.SigKernel.compute_Gram(X, Y) forward
 *                          call sites: src/kernels/_traj_kernels.py:203-206,
 *                                      src/inference/trajectory_svgd.py:60-63
 *                          static kernel fused in: src/kernels/_traj_kernels.py:176-195
 *   sigsvgd_gram_fwd_bwd   the same forward + its autograd backward for grad_output
 *                          (d sum(grad_out*K) / dX, first argument only), triggered by
 *                          src/inference/score.py:68-69, src/inference/trajectory_svgd.py:65
 *   sigsvgd_svgd_phi       SVGD._velocity dense part: v = -((K @ score - grad_k)/N) [* mask]
 *                          src/inference/svgd.py:82-83, src/inference/trajectory_svgd.py:84
 *                          optionally fused with the optimizer=None update X - lr*v (svgd.py:115)
 *   sigsvgd_svgd_step      the same with the adaptive_gradient=True scaling fused (svgd.py:110-113)
 *   sigsvgd_svgd_adam_step the same with the update of the reference's DEFAULT optimizer fused: torch.optim.Adam
 *                          stepped through a closure that sets X.grad to the velocity (svgd.py:20,100-107)
 *   sigsvgd_vec_sqdist     src/utils/math.py:69-86 pw_dist_sq, :116-144 scaled_pw_dist_sq
 *   sigsvgd_vec_kernel     src/kernels/_kernels.py:64-299 GaussianKernel / ScaledGaussianKernel /
 *                          IMQKernel / ScaledIMQKernel: K and d_K.sum(1) without the [A,B,D] tensor
 *   sigsvgd_obstacle_cost  batch_cost_fn of examples/script_planning_obstacle_field.py:113-126 (spline samples,
 *                          obstacle field, path length) and its gradient w.r.t. the knots (torch autograd there)
 *   sigsvgd_vec_kernel_fused  the same classes with a fixed bandwidth: distance, kernel and summed gradient in one launch
 *   sigsvgd_signature      signatory.signature(path, depth, basepoint) [third-party, absent] as
 *                          called by PathSigKernel, src/kernels/_traj_kernels.py:124-125
 *   sigsvgd_signature_backward  its autograd backward: PathSigKernel has analytic_grad=False (:92), so the reference
 *                          differentiates K THROUGH the signature (src/inference/score.py:50-55, svgd.py:41-43)
 *
 * Conventions
 *   - all pointers are DEVICE pointers (HIP), row-major contiguous; the caller owns every buffer
 *     and keeps it alive until the work queued on `stream` has completed;
 *   - inputs are read-only; outputs are fully overwritten; nothing persistent is allocated;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls only enqueue work
 *     (no host synchronisation, graph-capturable);
 *   - return value 0 = ok, negative = error (see SIGSVGD_E_*); sigsvgd_last_error() gives text;
 *   - `dtype` selects the I/O element type of X, Y, grad_out, K_out, gradX_out:
 *     SIGSVGD_F32 or SIGSVGD_F64.  Arithmetic of the register-resident and quadrant kernels
 *     (dyadic order 0, T <= 128) and of the refined-grid kernels (T <= 33 with dyadic refinement to
 *     64 .. 256 cells per side: the reference's own call shapes): fp64 static kernel + 4-corner
 *     increments, fp32 PDE sweeps in difference form, fp32 storage of per-pair intermediates, fp32
 *     gradient contraction, fp64 reduction over pairs; the coverage kernel (other refinements, longer
 *     paths, linear kernel, naive solver, SIGSVGD_FLAG_FORCE_GENERIC) is fp64 end to end (DESIGN.md
 *     "precision plan").  The fp32 route is checked PER PAIR and a pair that fails a check has its K solved
 *     again by the coverage kernel in fp64 (static kernel, increments, sweeps) inside the same call:
 *       (1) cancellation -- the largest |K| on the pair's PDE grid exceeds 2x .. 8x max(|K|, 0.1)
 *           (oscillating discrete solutions of rough paths in few channels);
 *       (2) conditioning, paths in <= 3 channels, dyadic order 0 -- the first-order condition number of K
 *           in the increments, sum |K_fwd U D| / max(|K|, 0.1), exceeds 150 (forward-only launches: the bound
 *           sum |K_fwd D| max(grid maximum, 1) / max(|K|, 0.1) exceeds 300): there the fp32 STORAGE of the
 *           increments limits K whatever the precision of the sweeps.
 *     Measured bound (DESIGN.md sections 2, 3): every entry of K_out within 1e-5 RELATIVE of the fp64
 *     reference's (plain |K - K_ref| / |K_ref|, no floor) over the committed rough-path cases and 18,000
 *     random soak cases; unflagged pairs of the calibration study are within 2.5e-6.  (The 8- / 16-channel
 *     kernels for T <= 64 have no exact pass: their cancelled pairs repeat the sweep in fp64 on fp32
 *     increments, enough in every case seen with d >= 5.)  Only K is repaired: the gradient of a flagged pair keeps
 *     the fp32 solution (its error is relative to the largest gradient entry of the launch and stayed
 *     below 5e-6 of it in every regime measured).  SIGSVGD_FLAG_FORCE_GENERIC returns 6e-8 anywhere.
 *   - results are bit-reproducible: every reduction over pairs runs in an order fixed by the launch
 *     geometry (no floating-point atomics), so two calls on the same inputs return the same bits.
 *     sigsvgd_vec_kernel_fused is reproducible when it is given its workspace (sigsvgd_vec_fused_workspace_bytes);
 *     without one its column splits meet in fp32 atomics and the last bits of dK_out may differ between calls.
 */
#ifndef SIGSVGD_HIP_H
#define SIGSVGD_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SIGSVGD_ABI_VERSION 9

/* dtype */
#define SIGSVGD_F32 0
#define SIGSVGD_F64 1

/* static kernel kinds */
#define SIGSVGD_STATIC_RBF 0    /* k(x,y) = exp(-|x-y|^2 * inv_h)   (reference: exp(-dist/h)) */
#define SIGSVGD_STATIC_LINEAR 1 /* k(x,y) = <x,y>                                              */

/* vector kernels (sigsvgd_vec_kernel) */
#define SIGSVGD_VEC_GAUSSIAN 0 /* k = exp(-sq / (2 h^2))          src/kernels/_kernels.py:106 */
#define SIGSVGD_VEC_IMQ 1      /* k = (1 + sq / (2 h^2))^(-1/2)     src/kernels/_kernels.py:229-230 */
#define SIGSVGD_VEC_UNIT 2     /* k = sq, w = 1: dK = grad_scale * sum_j grad_out_ij (XM_i - YM_j),
                                  the adjoint of sigsvgd_vec_sqdist (autograd through the distance) */

/* flags */
#define SIGSVGD_FLAG_NAIVE_SOLVER 1u /* first-order stencil (sigkernel _naive_solver=True)      */
#define SIGSVGD_FLAG_SYM 2u          /* sigkernel sym=True backward weighting: go + go^T (A==B) */
#define SIGSVGD_FLAG_Y_IS_X 4u       /* caller guarantees Y aliases X (same values): lets the   */
                                     /* library solve each unordered pair once                  */
#define SIGSVGD_FLAG_STORED_FORWARD 32u /* accepted and ignored: every kernel of this library keeps the forward solution   */
                                       /* (ABI 4-6 used it to route long paths away from a kernel that regenerated it)     */
#define SIGSVGD_FLAG_WS_CLEAN 16u     /* accepted and ignored since ABI 8: no launch needs a zeroed workspace any more */
                                      /* (partial sums are stored, not accumulated), none issues a memset, and a        */
                                      /* captured graph of an iteration consists of kernel nodes only                   */
#define SIGSVGD_FLAG_FOLD_TILES 64u    /* sigsvgd_gram_sym_partial: this launch owns the row tiles tile_offset + k*tile_stride AND  */
                                      /* their mirror images ntile-1 - (tile_offset + k*tile_stride): a tile and its mirror image  */
                                      /* together always hold the same number of pairs of the upper triangle, so every rank of   */
                                      /* the sharded step gets the same share (cyclic ownership alone: +5.4 % on the first rank)    */
#define SIGSVGD_FLAG_FORCE_GENERIC 8u /* use the coverage kernel: fp64 end to end (every entry of K is the fp64 reference's up to */
                                      /* the store in `dtype`, in any regime): whole-grid fp64 tables where they fit LDS, per-band */
                                      /* fp64 increments beyond that (dyadic order 0, T <= ~200); one wavefront per pair; tests,   */
                                      /* and callers who want the gradient of rough few-channel paths from fp64 sweeps as well     */

/* errors */
#define SIGSVGD_OK 0
#define SIGSVGD_E_BADARG -1
#define SIGSVGD_E_UNSUPPORTED -2 /* shape does not fit the device limits (LDS) */
#define SIGSVGD_E_WORKSPACE -3   /* workspace too small */
#define SIGSVGD_E_HIP -4         /* a HIP runtime call failed */

int sigsvgd_abi_version(void);
const char *sigsvgd_last_error(void);

/* Bytes of scratch the two Gram entry points need for this problem.  Forward-only launches need some too
 * (accumulation buffers, scratch of the persistent grids), so always query; the size covers every value of
 * SIGSVGD_FLAG_Y_IS_X / SIGSVGD_FLAG_SYM for the given shape.  `static_kind` and `flags` are the ones of the launch
 * (ABI 9: the static kernel decides which solver runs -- the linear kernel always takes the coverage kernel --, so the
 * query needs it; ABI <= 8 sized for RBF whatever the launch asked for).
 * want_grad = 0 for sigsvgd_gram_fwd, 1 for sigsvgd_gram_fwd_bwd.  Returns 0 and sets *bytes. */
int sigsvgd_gram_workspace_bytes(int A, int B, int T, int d, int dyadic_order, int static_kind, int want_grad,
                                 unsigned flags, size_t *bytes);

/* K_out[A,B] = signature-kernel Gram matrix of paths X[A,T,d], Y[B,T,d]. */
int sigsvgd_gram_fwd(const void *X, const void *Y, int A, int B, int T, int d, int dtype,
                     double inv_h, int dyadic_order, int static_kind, unsigned flags,
                     void *K_out, void *workspace, size_t workspace_bytes, void *stream);

/* As above, plus gradX_out[A,T,d] = d( sum_ij grad_out[i,j] K[i,j] ) / dX  (first slot only;
 * Y receives no gradient, as in the reference).  grad_out == NULL means all ones (the only case
 * the reference produces: callers differentiate K.sum()). */
int sigsvgd_gram_fwd_bwd(const void *X, const void *Y, int A, int B, int T, int d, int dtype,
                         double inv_h, int dyadic_order, int static_kind, unsigned flags,
                         const void *grad_out, void *K_out, void *gradX_out, void *workspace,
                         size_t workspace_bytes, void *stream);

/* Multi-GPU building block (particles sharded over ranks; new design, the reference has no
 * distributed code -- SURVEY.md §8e).  Solves the unordered pairs {i <= j} whose row tile
 * (sigsvgd_gram_sym_tile_rows(T, d) consecutive rows i) has index tile_offset + k*tile_stride for some k >= 0
 * -- with SIGSVGD_FLAG_FOLD_TILES also the mirror images of those tiles, see the flag -- on the full gathered
 * particle tensor X[N,T,d]:
 *   K_partial[N,N]      (dtype)  caller-ZEROED; receives both orientations K[i,j], K[j,i] of every owned pair
 *   grad_partial[N,T,d] (fp64)   OVERWRITTEN with this launch's share of d sum(grad_out*K)/dX (row- and
 *                                column-side; rows the owned pairs do not touch get 0)
 * Summing the buffers over tile_offset = 0..tile_stride-1 gives sigsvgd_gram_fwd_bwd's outputs (K exactly, the
 * gradient up to the fp64 rounding of the sum).  Shapes of the register-resident and quadrant kernels
 * (dyadic_order 0, 3 <= T <= 128, d <= 16, RBF).
 * `workspace` as sized by sigsvgd_gram_workspace_bytes(N, N, T, d, 0, static_kind, 1, SIGSVGD_FLAG_Y_IS_X). */
int sigsvgd_gram_sym_partial(const void *X, int N, int T, int d, int dtype, double inv_h,
                             int static_kind, unsigned flags, int tile_offset, int tile_stride,
                             const void *grad_out, void *K_partial, double *grad_partial,
                             void *workspace, size_t workspace_bytes, void *stream);

/* Rows per tile of the symmetric / partial solve for paths of T points in d channels: 4 for T <= 64 with d > 8, else 8
 * (0 for shapes sigsvgd_gram_sym_partial does not take).  The unit of ownership of the sharded solve. */
int sigsvgd_gram_sym_tile_rows(int T, int d);

/* v_out[N,D] = -((K[N,N] @ score[N,D] - grad_k[N,D]) / N) * (mask ? mask[N,D] : 1)   (fp32)
 * If X_in and X_out are non-NULL additionally X_out = X_in - lr * v_out (optimizer=None update).
 * The N x N x D product runs on the fp32 MFMA (exact fp32 FMA chain). */
int sigsvgd_svgd_phi(const float *K, const float *score, const float *grad_k, const float *mask,
                     int N, int D, float *v_out, const float *X_in, float *X_out, float lr,
                     void *stream);

/* The same launch with the reference's "simple Adagrad" (SVGD(adaptive_gradient=True), svgd.py:110-113)
 * fused into the epilogue when adagrad_state[N,D] is non-NULL:
 *   state += v^2;  v_out = v / sqrt(state + 1e-12);  X_out = X_in - lr * v_out
 * (v already multiplied by the mask).  adagrad_state is read and written in place; zero it before the
 * first iteration.  With adagrad_state == NULL this is sigsvgd_svgd_phi. */
int sigsvgd_svgd_step(const float *K, const float *score, const float *grad_k, const float *mask,
                      int N, int D, float *v_out, const float *X_in, float *X_out, float lr,
                      float *adagrad_state, void *stream);

/* The same launch with torch.optim.Adam's update (amsgrad=False, weight_decay=0, maximize=False) in the epilogue:
 *   t = *step_dev + 1;  m = m + (1-beta1)(v - m);  q = beta2 q + (1-beta2) v^2;
 *   X_out = X_in - lr/(1-beta1^t) * m / (sqrt(q)/sqrt(1-beta2^t) + eps);   then *step_dev = t
 * exp_avg (m) and exp_avg_sq (q) are [N,D] fp32, updated in place; step_dev is an int on the DEVICE (the bias
 * corrections are formed in the kernel, so the launch can be replayed from a captured HIP graph); it is
 * incremented by a one-thread launch behind the update.  lr, beta1, beta2 and eps are doubles as torch holds them
 * (1 - beta and the bias corrections are formed in fp64 and rounded once, as torch does).  v_out receives the velocity (what the reference stores
 * in X.grad and in iter_dict["grad"]). */
int sigsvgd_svgd_adam_step(const float *K, const float *score, const float *grad_k, const float *mask,
                           int N, int D, float *v_out, const float *X_in, float *X_out, double lr, double beta1,
                           double beta2, double eps, float *exp_avg, float *exp_avg_sq, int *step_dev, void *stream);

/* ---- vector kernels on particles X[A,D], Y[B,D] ---------------------------------
 * sq[i,j] = max(0, sum_c (XM[i,c] - YM[j,c]) * (X[i,c] - Y[j,c])).  XM = X @ M, YM = Y @ M for a metric
 * M[D,D] (scaled_pw_dist_sq); XM = YM = NULL means M = I, i.e. |x_i - y_j|^2 (pw_dist_sq).  Arithmetic
 * in `dtype`.  Differences are formed directly, so sq >= 0 up to rounding and the clamp is a no-op. */
int sigsvgd_vec_sqdist(const void *X, const void *Y, const void *XM, const void *YM, int A, int B, int D,
                       int dtype, void *sq_out, void *stream);

/* From sq[A,B]:  K_out[i,j] = f(sq[i,j])  (kind: SIGSVGD_VEC_GAUSSIAN / SIGSVGD_VEC_IMQ, inv_h2 = 1/h^2)
 * and, if dK_out != NULL,
 *   dK_out[i,c] = grad_scale * sum_j (grad_out ? grad_out[i,j] : 1) * w(sq[i,j]) * (XM[i,c] - YM[j,c]),
 * w = f for the Gaussian, f^3 = (1 + sq/(2h^2))^(-3/2) for the IMQ: the reference's `d_K.sum(1)`
 * with grad_scale = -1/h^2 (Gaussian kernels), +1/(2h^2) (IMQKernel: its (Y - X) sign convention,
 * _kernels.py:232) or -1/(2h^2) (ScaledIMQKernel, _kernels.py:297).  For M = I pass XM = X, YM = Y.
 * K_out may be NULL when only the gradient is wanted (it must be with SIGSVGD_VEC_UNIT). */
int sigsvgd_vec_kernel(const void *sq, const void *XM, const void *YM, const void *grad_out, int A, int B,
                       int D, int dtype, int kind, double inv_h2, double grad_scale, void *K_out,
                       void *dK_out, void *stream);

/* The same in ONE launch when the bandwidth is known in advance (no sq[A,B] round trip through HBM; both
 * GEMM-shaped sums on the fp32 matrix cores).  X, Y [.,D]; XM, YM = X M, Y M or both NULL (M = I); K_out / dK_out
 * nullable (not both).  fp32 and D <= 512 only (else SIGSVGD_E_UNSUPPORTED: use the two calls above).  Operands
 * are centred on the first row of Y (differences are unchanged); dK_out is fully overwritten. */
int sigsvgd_vec_kernel_fused(const void *X, const void *Y, const void *XM, const void *YM, const void *grad_out, int A,
                             int B, int D, int dtype, int kind, double inv_h2, double grad_scale, void *K_out,
                             void *dK_out, void *workspace, size_t workspace_bytes, void *stream);
/* Scratch for the reproducible route of sigsvgd_vec_kernel_fused (ABI 9): the launch splits the columns over the grid, and
 * with a workspace of this size every split stores its partial dK in a block of its own, added in split order by a second
 * small launch -- bits that depend on the launch geometry only.  workspace = NULL keeps the one-launch route whose splits
 * meet in fp32 atomics (last bits of dK_out may differ between calls).  0 bytes: a single split, nothing to join. */
int sigsvgd_vec_fused_workspace_bytes(int A, int B, int D, size_t *bytes);

/* ---- trajectory cost in front of the path -----------------------------------------------
 * The reference's planning cost, examples/script_planning_obstacle_field.py:113-126, with its analytic gradient:
 *   knots_i = [start, x_i[0..knots-1], target]  ->  traj_i = basis[samples, knots+2] @ knots_i   (natural cubic
 *   spline samples for fixed knot times, :18-23; pass the identity for use_splines=False)
 *   cost_i  = w_obstacle * sum_t p(traj_i[t]) + || w_length * (traj_i[1:] - traj_i[:-1]) ||_F
 *   p(z)    = sum_m exp(log_weights[m]) prod_c Normal(z_c; mean[m,c], std[m,c])   (the script's obstacle field, :363-370)
 * Outputs: cost[N], traj[N,samples,d] (nullable), grad_x[N,knots,d] = d cost_i / d x_i (nullable; samples <= 128).
 * All fp32; d <= 16, knots + 2 <= 64, samples <= 1024. */
int sigsvgd_obstacle_cost(const float *x, int N, int knots, int d, const float *start, const float *target,
                          const float *basis, int samples, const float *log_weights, const float *mean, const float *std,
                          int components, float w_obstacle, float w_length, float *cost, float *traj, float *grad_x,
                          void *stream);

/* ---- truncated path signature ----------------------------------------------------
 * out[N, C + C^2 + ... + C^depth] = signature of the piecewise-linear path X[N,L,C] (levels
 * concatenated, each level row-major), with a zero point prepended when basepoint != 0.  Chen's
 * identity, fp64 accumulation, I/O in `dtype`.  *channels (if non-NULL) receives the output width;
 * with out == NULL only that query is performed. */
int sigsvgd_signature(const void *X, int N, int L, int C, int depth, int basepoint, int dtype, void *out,
                      long long *channels, void *stream);

/* grad_X[N,L,C] = d( sum_{n,e} grad_sig[n,e] * signature(X)[n,e] ) / dX: the adjoint of sigsvgd_signature for the same
 * (depth, basepoint).  grad_sig is [N, channels]; fp64 arithmetic (the signature is rebuilt forwards, then unwound point by
 * point with the group inverse exp(-increment): nothing per point is stored), I/O in `dtype`.  depth <= 8 and
 * 6 * channels doubles of LDS (channels <= ~3000), else SIGSVGD_E_UNSUPPORTED. */
int sigsvgd_signature_backward(const void *X, const void *grad_sig, int N, int L, int C, int depth, int basepoint, int dtype,
                               void *grad_X, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SIGSVGD_HIP_H */
