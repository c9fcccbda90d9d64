/*
 * CPU ORACLE in plain C (test infrastructure + bench.py cpu_baseline "port"; NOT product code).
 *
 * PARITY STATUS: parity unpinned for the PDE arithmetic (third-party `sigkernel`,
 * crispitagorico/sigkernel@3b2373982e12b3d499a80228311a04debcc1bea1, /root/reference/setup.py:71,
 * absent from /root/reference) -- see oracle/sigkernel_oracle.py header.  This file is the same
 * restatement as sigkernel_oracle.py, streamed pair by pair so that the headline size
 * (N=1024, T=64, d=7) runs in O(P^2) memory per thread instead of the reference's O(N^2 P^2).
 *
 * Follows:
 *   static kernel  exp(-dist/h)  ...... /root/reference/src/kernels/_traj_kernels.py:176-195
 *   fp64 upcast of fp32 particles ..... /root/reference/src/kernels/_traj_kernels.py:204-205
 *   Gram + d(sum grad_out*K)/dX ....... /root/reference/src/inference/score.py:68-69
 *   [RECALLED] increments, dyadic tiling, second-order stencil, K_fwd*K_rev gradient:
 *              sigkernel _SigKernelGram.forward/backward.
 *
 * Build:  gcc -O3 -fopenmp -shared -fPIC -o oracle/_build/liboracle.so oracle/sigkernel_c.c -lm
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define KIND_RBF 0
#define KIND_LINEAR 1

/* one pair: X_i [T,d], Y_j [T,d] (fp32, upcast) -> K (scalar) and, if pts != NULL,
 * pts[m,c] = sum_n R[m,n] * dk(X_im,Y_jn)/dX_im[c]  (gradient of this pair's K wrt X_i).
 * scratch: G T*T, D (T-1)^2, Kf (P+1)^2, Kr (P+1)^2, S (T-1)^2 */
static double pair_solve(const float *xi, const float *yj, int T, int d, double inv_h, int n,
                         int naive, int kind, double *G, double *D, double *Kf, double *Kr,
                         double *S, double *pts)
{
    const int r = 1 << n, Tm = T - 1, P = r * Tm, W = P + 1;
    const double inv_r2 = 1.0 / ((double)r * (double)r);
    /* static Gram, same operation order as the reference: |x|^2 + |y|^2 - 2<x,y>, exp(-dist/h) */
    for (int p = 0; p < T; ++p) {
        double xs = 0.0;
        for (int c = 0; c < d; ++c) xs += (double)xi[p * d + c] * (double)xi[p * d + c];
        for (int q = 0; q < T; ++q) {
            double ys = 0.0, dot = 0.0;
            for (int c = 0; c < d; ++c) {
                double yv = (double)yj[q * d + c];
                ys += yv * yv;
                dot += (double)xi[p * d + c] * yv;
            }
            G[p * T + q] = (kind == KIND_LINEAR) ? dot : exp(-(-2.0 * dot + xs + ys) * inv_h);
        }
    }
    for (int a = 0; a < Tm; ++a)
        for (int b = 0; b < Tm; ++b)
            D[a * Tm + b] = G[(a + 1) * T + b + 1] + G[a * T + b] - G[(a + 1) * T + b] - G[a * T + b + 1];
    /* forward sweep */
    for (int q = 0; q <= P; ++q) Kf[q] = 1.0;
    for (int p = 0; p < P; ++p) {
        Kf[(p + 1) * W] = 1.0;
        const double *Drow = D + (p / r) * Tm;
        for (int q = 0; q < P; ++q) {
            double g = Drow[q / r] * inv_r2;
            double k10 = Kf[(p + 1) * W + q], k01 = Kf[p * W + q + 1], k00 = Kf[p * W + q];
            if (naive)
                Kf[(p + 1) * W + q + 1] = k10 + k01 + k00 * (g - 1.0);
            else {
                double g2 = g * g / 12.0;
                Kf[(p + 1) * W + q + 1] = (k10 + k01) * (1.0 + 0.5 * g + g2) - k00 * (1.0 - g2);
            }
        }
    }
    const double Kval = Kf[P * W + P];
    if (!pts) return Kval;
    /* reverse sweep, written in un-flipped coordinates: U[P,:]=U[:,P]=1,
     * U[p,q] = (U[p+1,q]+U[p,q+1])*A(g[p,q]) - U[p+1,q+1]*B(g[p,q])   (== flip(sweep(flip g))) */
    for (int q = 0; q <= P; ++q) Kr[P * W + q] = 1.0;
    for (int p = P - 1; p >= 0; --p) {
        Kr[p * W + P] = 1.0;
        const double *Drow = D + (p / r) * Tm;
        for (int q = P - 1; q >= 0; --q) {
            double g = Drow[q / r] * inv_r2;
            double k10 = Kr[(p + 1) * W + q], k01 = Kr[p * W + q + 1], k11 = Kr[(p + 1) * W + q + 1];
            if (naive)
                Kr[p * W + q] = k10 + k01 + k11 * (g - 1.0);
            else {
                double g2 = g * g / 12.0;
                Kr[p * W + q] = (k10 + k01) * (1.0 + 0.5 * g + g2) - k11 * (1.0 - g2);
            }
        }
    }
    /* S[a,b] = r^-2 * sum_{block} Kf[p,q]*Kr[p+1,q+1] */
    memset(S, 0, sizeof(double) * (size_t)Tm * Tm);
    for (int p = 0; p < P; ++p)
        for (int q = 0; q < P; ++q)
            S[(p / r) * Tm + q / r] += Kf[p * W + q] * Kr[(p + 1) * W + q + 1];
    for (int a = 0; a < Tm * Tm; ++a) S[a] *= inv_r2;
    /* pts[m,c] = sum_n R[m,n] V[m,n,c],  R = 4-corner scatter of S */
    for (int m = 0; m < T; ++m) {
        for (int c = 0; c < d; ++c) pts[m * d + c] = 0.0;
        for (int nn = 0; nn < T; ++nn) {
            double R = 0.0;
            if (m >= 1 && nn >= 1) R += S[(m - 1) * Tm + nn - 1];
            if (m < Tm && nn < Tm) R += S[m * Tm + nn];
            if (m >= 1 && nn < Tm) R -= S[(m - 1) * Tm + nn];
            if (m < Tm && nn >= 1) R -= S[m * Tm + nn - 1];
            if (kind == KIND_LINEAR) {
                for (int c = 0; c < d; ++c) pts[m * d + c] += R * (double)yj[nn * d + c];
            } else {
                double w = -2.0 * inv_h * R * G[m * T + nn];
                for (int c = 0; c < d; ++c)
                    pts[m * d + c] += w * ((double)xi[m * d + c] - (double)yj[nn * d + c]);
            }
        }
    }
    return Kval;
}

/* K_out [(i1-i0), B] and gradX_out [(i1-i0), T, d] (may be NULL) for rows i0..i1-1 of X.
 * grad_out [(i1-i0), B] or NULL (= ones).  Gradient flows to the first argument only. */
int oracle_gram_fwd_bwd(const float *X, const float *Y, int A, int B, int T, int d, double inv_h,
                        int n, int naive, int kind, const double *grad_out, int i0, int i1,
                        double *K_out, double *gradX_out, int nthreads)
{
    if (T < 2 || d < 1 || n < 0 || i0 < 0 || i1 > A || i0 > i1) return -1;
    const int r = 1 << n, Tm = T - 1, P = r * Tm, W = P + 1;
    (void)A;
    int status = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel
    {
        double *G = (double *)malloc(sizeof(double) * (size_t)T * T);
        double *D = (double *)malloc(sizeof(double) * (size_t)Tm * Tm);
        double *Kf = (double *)malloc(sizeof(double) * (size_t)W * W);
        double *Kr = (double *)malloc(sizeof(double) * (size_t)W * W);
        double *S = (double *)malloc(sizeof(double) * (size_t)Tm * Tm);
        double *pts = (double *)malloc(sizeof(double) * (size_t)T * d);
        double *acc = (double *)malloc(sizeof(double) * (size_t)T * d);
        if (!G || !D || !Kf || !Kr || !S || !pts || !acc) {
#pragma omp atomic write
            status = -2;
        } else {
#pragma omp for schedule(dynamic, 1)
            for (int i = i0; i < i1; ++i) {
                memset(acc, 0, sizeof(double) * (size_t)T * d);
                for (int j = 0; j < B; ++j) {
                    double k = pair_solve(X + (size_t)i * T * d, Y + (size_t)j * T * d, T, d, inv_h, n,
                                          naive, kind, G, D, Kf, Kr, S, gradX_out ? pts : NULL);
                    K_out[(size_t)(i - i0) * B + j] = k;
                    if (gradX_out) {
                        double w = grad_out ? grad_out[(size_t)(i - i0) * B + j] : 1.0;
                        for (int t = 0; t < T * d; ++t) acc[t] += w * pts[t];
                    }
                }
                if (gradX_out) memcpy(gradX_out + (size_t)(i - i0) * T * d, acc, sizeof(double) * (size_t)T * d);
            }
        }
        free(G); free(D); free(Kf); free(Kr); free(S); free(pts); free(acc);
    }
    return status;
}

/* phi = (K @ score - grad_k)/N ; X_new = X + lr*phi   (svgd.py:82-83,115 with v = -phi). fp64. */
int oracle_svgd_update(const double *K, const double *score, const double *grad_k, const double *X,
                       int N, int Dflat, double lr, double *phi_out, double *X_new)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) {
        for (int c = 0; c < Dflat; ++c) {
            double s = 0.0;
            for (int j = 0; j < N; ++j) s += K[(size_t)i * N + j] * score[(size_t)j * Dflat + c];
            double ph = (s - grad_k[(size_t)i * Dflat + c]) / (double)N;
            phi_out[(size_t)i * Dflat + c] = ph;
            if (X_new) X_new[(size_t)i * Dflat + c] = X[(size_t)i * Dflat + c] + lr * ph;
        }
    }
    return 0;
}

int oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
