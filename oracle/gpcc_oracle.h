/*
 * gpcc_oracle.h -- CPU restatement of GPCC.jl's marginal-log-likelihood hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under gpcc.jl_amd/ (the product) may link,
 * import or call this library.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and there only as the checker / the
 * reported CPU baseline.
 *
 * PARITY UNPINNED: the reference (pure Julia, /root/reference/src) cannot be run in
 * this image (no Julia toolchain) and its test-suite is empty
 * (/root/reference/test/runtests.jl:4-6), so there are no golden vectors to pin this
 * restatement against.  It is pinned instead by (i) an independent numpy/scipy
 * restatement (tests/golden/make_golden.py) that must agree to <=1e-12 relative and
 * (ii) analytic known-answer tests (tests/test_oracle_kat.py).
 *
 * Third-party arithmetic that is NOT under /root/reference (Project.toml:18-25 gives
 * compat ranges only; no Manifest.toml is committed):
 *   Distributions 0.25 / PDMats  -- MvNormal(mu, K), logpdf: Cholesky + triangular solve,
 *                                   logpdf = -(N log 2pi + logdet K)/2 - |L^-1 (Y-mu)|^2 / 2
 *   LinearAlgebra / OpenBLAS     -- dpotrf (lower/upper), PosDefException on pivot <= 0
 *   StatsFuns 0.9/1              -- logsumexp (max-shifted)
 *   MiscUtil (unversioned)       -- makematrixsymmetric! (no-op here: K is exactly symmetric)
 * Their published definitions are restated below.
 */
#ifndef GPCC_ORACLE_H
#define GPCC_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* kernel ids shared with include/gpcc_hip.h */
enum { GPCC_ORACLE_OU = 0, GPCC_ORACLE_RBF = 1, GPCC_ORACLE_MATERN32 = 2, GPCC_ORACLE_MATERN52 = 3 };

/* src/util.jl:15-52 -- scalar kernels, formulas verbatim (incl. rbf's exp(-0.5 r^2/(2 rho))). */
double gpcc_oracle_kernel(int kernel_id, double xi, double xj, double rho);

/* src/delayedCovariance.jl:1-38.  x, y are the L ragged vectors flattened in band order.
 * out is column-major (sum Nx) x (sum Ny), like Julia's Matrix.
 * returns 0, -1 (scale <= 0: the @assert at :3), -2 (rho <= 0: error at :5-7), -3 bad kernel id. */
int gpcc_oracle_covariance(int kernel_id, int L, const double *scale, const double *delays, double rho,
                           const int *Nx, const double *x, const int *Ny, const double *y, double *out);

/* src/gpccfixdelay_marginaliseb.jl:85-98 (marginalise_b=1) and src/gpccfixdelay.jl:85-96 (=0):
 * mean_b[l] = mean(y_l); Sigma_b[l] = 100*var(y_l) (n-1 variance; 0 when marginalise_b=0);
 * resid = Y - bbar (bbar = Q mu_b, resp. Q b with b = (Q'Q)\Q'Y = per-band means). */
int gpcc_oracle_precompute(int L, const int *Nl, const double *y, int marginalise_b,
                           double *mean_b, double *Sigma_b, double *resid);

/* K = delayedCovariance(kernel, alpha, tau, rho, t) + Sobs (+ B), column-major N x N
 * (marginaliseb.jl:135 / gpccfixdelay.jl:133), plus resid = Y - bbar. */
int gpcc_oracle_model_matrix(int kernel_id, int L, const int *Nl, const double *t, const double *y,
                             const double *sigma, int marginalise_b, const double *delays,
                             const double *alpha, double rho, double *K, double *resid);

/* In-place lower Cholesky of a column-major n x n matrix (dpotrf 'L' semantics):
 * returns 0 or the 1-based order of the first non-positive (or NaN) pivot. */
int gpcc_oracle_potrf_lower(int n, double *A, int lda);

/* objective(alpha, rho) for M independent (tau, alpha, rho) triples
 * (marginaliseb.jl:133-141 / gpccfixdelay.jl:131-139).  delays and alpha are row-major M x L.
 * info[m] = 0 or the potrf pivot index (loglik[m] = NaN then); argument errors as in
 * gpcc_oracle_covariance are returned per item as info[m] = -1 / -2.
 * nthreads > 1 evaluates items in parallel (OpenMP), one item per thread -- the shape of
 * README.md:181-211's pmap.  Returns 0, or <0 for errors that affect the whole call. */
int gpcc_oracle_loglik_batch(int kernel_id, int L, const int *Nl, const double *t, const double *y,
                             const double *sigma, int marginalise_b, int M, const double *delays,
                             const double *alpha, const double *rho, double *loglik, int *info,
                             int nthreads);

/* src/getprobabilities.jl:1-20.  logprior == NULL reproduces the 1-argument form
 * (log-prior = array of ones, :3). */
int gpcc_oracle_probabilities(int G, const double *loglik, const double *logprior, double *out);

int gpcc_oracle_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
