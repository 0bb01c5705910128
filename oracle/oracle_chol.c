/*
 * oracle_chol.c -- dense part of the CPU oracle (see gpcc_oracle.h: TEST INFRASTRUCTURE,
 * PARITY UNPINNED): LAPACK-dpotrf-style blocked lower Cholesky, forward substitution and the
 * Gaussian log-density that Distributions.logpdf(MvNormal(mu, K), Y) evaluates at
 * /root/reference/src/gpccfixdelay_marginaliseb.jl:139.
 */
#include "gpcc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NB 64

/* unblocked lower Cholesky of the n x n block at A (dpotf2 'L'); returns 0 or 1-based pivot */
static int potf2_lower(int n, double *A, int lda)
{
    for (int j = 0; j < n; ++j) {
        double d = A[j + (long)j * lda];
        for (int k = 0; k < j; ++k) { double v = A[j + (long)k * lda]; d -= v * v; }
        if (!(d > 0.0)) return j + 1; /* catches NaN as LAPACK's disnan test does */
        d = sqrt(d);
        A[j + (long)j * lda] = d;
        for (int k = 0; k < j; ++k) {
            double ajk = A[j + (long)k * lda];
            const double *ck = A + (long)k * lda;
            double *cj = A + (long)j * lda;
            for (int i = j + 1; i < n; ++i) cj[i] -= ck[i] * ajk;
        }
        double inv = 1.0 / d;
        double *cj = A + (long)j * lda;
        for (int i = j + 1; i < n; ++i) cj[i] *= inv;
    }
    return 0;
}

/* B (m x nb) <- B * L^-T, L = nb x nb lower block (dtrsm Right/Lower/Trans/NonUnit) */
static void trsm_rlt(int m, int nb, const double *Lb, int ldl, double *B, int ldb)
{
    for (int j = 0; j < nb; ++j) {
        double *bj = B + (long)j * ldb;
        for (int k = 0; k < j; ++k) {
            double ljk = Lb[j + (long)k * ldl];
            const double *bk = B + (long)k * ldb;
            for (int i = 0; i < m; ++i) bj[i] -= bk[i] * ljk;
        }
        double inv = 1.0 / Lb[j + (long)j * ldl];
        for (int i = 0; i < m; ++i) bj[i] *= inv;
    }
}

/* C (lower part of the m x m trailing block) -= P P^T, P = m x nb panel (dsyrk/dgemm) */
static void syrk_lower(int m, int nb, const double *P, int ldp, double *C, int ldc)
{
    for (int jb = 0; jb < m; jb += NB) {
        int jn = (m - jb < NB) ? m - jb : NB;
        for (int j = jb; j < jb + jn; ++j) {
            double *cj = C + (long)j * ldc;
            for (int k = 0; k < nb; ++k) {
                double pjk = P[j + (long)k * ldp];
                const double *pk = P + (long)k * ldp;
                for (int i = j; i < m; ++i) cj[i] -= pk[i] * pjk;
            }
        }
    }
}

int gpcc_oracle_potrf_lower(int n, double *A, int lda)
{
    for (int k = 0; k < n; k += NB) {
        int nb = (n - k < NB) ? n - k : NB;
        double *Akk = A + k + (long)k * lda;
        int info = potf2_lower(nb, Akk, lda);
        if (info) return k + info;
        int m = n - k - nb;
        if (m > 0) {
            double *Pk = A + (k + nb) + (long)k * lda;
            trsm_rlt(m, nb, Akk, lda, Pk, lda);
            syrk_lower(m, nb, Pk, lda, A + (k + nb) + (long)(k + nb) * lda, lda);
        }
    }
    return 0;
}

/* one objective(alpha, rho) evaluation; Kbuf (N*N) and zbuf (N) are scratch */
static int loglik_one(int kernel_id, int L, const int *Nl, long N, const double *t, const double *y,
                      const double *sigma, int marginalise_b, const double *delays,
                      const double *alpha, double rho, double *Kbuf, double *zbuf, double *out)
{
    *out = NAN;
    int rc = gpcc_oracle_model_matrix(kernel_id, L, Nl, t, y, sigma, marginalise_b, delays, alpha, rho,
                                      Kbuf, zbuf);
    if (rc) return rc;
    /* MvNormal(bbar, K) -> PDMat(K) -> cholesky(K): PosDefException <=> info > 0 */
    int info = gpcc_oracle_potrf_lower((int)N, Kbuf, (int)N);
    if (info) return info;
    /* logdet(K) = 2 sum log L_ii ; sqmahal = |L^-1 (Y - bbar)|^2 */
    double logdet = 0.0;
    for (long i = 0; i < N; ++i) logdet += log(Kbuf[i + i * N]);
    logdet *= 2.0;
    for (long j = 0; j < N; ++j) { /* column-oriented forward substitution */
        double zj = zbuf[j] / Kbuf[j + j * N];
        zbuf[j] = zj;
        const double *cj = Kbuf + j * N;
        for (long i = j + 1; i < N; ++i) zbuf[i] -= cj[i] * zj;
    }
    double q = 0.0;
    for (long i = 0; i < N; ++i) q += zbuf[i] * zbuf[i];
    const double log2pi = 1.8378770664093454835606594728112;
    /* Distributions: mvnormal_c0 - sqmahal/2, mvnormal_c0 = -(N log 2pi + logdetcov)/2 */
    *out = -((double)N * log2pi + logdet) / 2.0 - q / 2.0;
    return 0;
}

int gpcc_oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

int gpcc_oracle_loglik_batch(int kernel_id, int L, const int *Nl, const double *t, const double *y,
                             const double *sigma, int marginalise_b, int M, const double *delays,
                             const double *alpha, const double *rho, double *loglik, int *info,
                             int nthreads)
{
    if (kernel_id < 0 || kernel_id > 3) return -3;
    if (L <= 0 || M < 0) return -4;
    long N = 0;
    for (int l = 0; l < L; ++l) { if (Nl[l] <= 0) return -4; N += Nl[l]; }
    if (nthreads < 1) nthreads = 1;
    int failed = 0;
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
#endif
    {
        double *Kbuf = (double *)malloc(sizeof(double) * (size_t)N * (size_t)N);
        double *zbuf = (double *)malloc(sizeof(double) * (size_t)N);
        if (!Kbuf || !zbuf) {
#ifdef _OPENMP
#pragma omp atomic write
#endif
            failed = 1;
        } else {
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
            for (int m = 0; m < M; ++m)
                info[m] = loglik_one(kernel_id, L, Nl, N, t, y, sigma, marginalise_b,
                                     delays + (long)m * L, alpha + (long)m * L, rho[m], Kbuf, zbuf,
                                     &loglik[m]);
        }
        free(Kbuf);
        free(zbuf);
    }
    return failed ? -5 : 0;
}
