/*
 * oracle_model.c -- element-wise part of the CPU oracle (see gpcc_oracle.h: TEST INFRASTRUCTURE,
 * PARITY UNPINNED).  Compiled with -ffp-contract=off so that every product and sum is rounded
 * exactly as Julia's scalar code rounds it (Julia does not contract a*b+c unless asked).
 */
#include "gpcc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* src/util.jl:15-23 (OU), :28 (rbf), :32-40 (matern32), :44-52 (matern52). */
double gpcc_oracle_kernel(int kernel_id, double xi, double xj, double rho)
{
    switch (kernel_id) {
    case GPCC_ORACLE_OU: {
        double r = fabs(xi - xj);
        return exp(-r / rho);
    }
    case GPCC_ORACLE_RBF: {
        /* exp(-0.5*(xi-xj)^2/(2rho)): rho enters linearly -- reproduced verbatim. */
        double d = xi - xj;
        return exp(-0.5 * (d * d) / (2.0 * rho));
    }
    case GPCC_ORACLE_MATERN32: {
        double r = fabs(xi - xj);
        double s3 = sqrt(3.0);
        return (1.0 + s3 * r / rho) * exp(-s3 * r / rho);
    }
    case GPCC_ORACLE_MATERN52: {
        double r = fabs(xi - xj);
        double s5 = sqrt(5.0);
        return (1.0 + s5 * r / rho + (5.0 * (r * r)) / (3.0 * (rho * rho))) * exp(-s5 * r / rho);
    }
    default:
        return NAN;
    }
}

static int check_scale_rho(int L, const double *scale, double rho)
{
    for (int l = 0; l < L; ++l)
        if (!(scale[l] > 0.0)) return -1; /* delayedCovariance.jl:3 */
    if (rho <= 0.0) return -2;            /* delayedCovariance.jl:5-7 */
    return 0;
}

/* src/delayedCovariance.jl:23-31: block (i,j), element (n,m) =
 * scale[i]*scale[j]*kernel(x[i][n]-delays[i], y[j][m]-delays[j]; rho). */
int gpcc_oracle_covariance(int kernel_id, int L, const double *scale, const double *delays, double rho,
                           const int *Nx, const double *x, const int *Ny, const double *y, double *out)
{
    if (kernel_id < 0 || kernel_id > 3) return -3;
    int rc = check_scale_rho(L, scale, rho);
    if (rc) return rc;
    long sx = 0, sy = 0;
    for (int l = 0; l < L; ++l) { sx += Nx[l]; sy += Ny[l]; }
    long col = 0;
    for (int j = 0; j < L; ++j) {
        for (int m = 0; m < Ny[j]; ++m, ++col) {
            double yj = y[col] - delays[j];
            long row = 0;
            for (int i = 0; i < L; ++i) {
                double ss = scale[i] * scale[j];
                for (int n = 0; n < Nx[i]; ++n, ++row) {
                    double xi = x[row] - delays[i];
                    out[col * sx + row] = ss * gpcc_oracle_kernel(kernel_id, xi, yj, rho);
                }
            }
        }
    }
    (void)sy;
    return 0;
}

/* Statistics.mean / Statistics.var (corrected, n-1) as used at marginaliseb.jl:92-94. */
static double mean_of(const double *v, int n)
{
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += v[i];
    return s / n;
}

static double var_of(const double *v, int n, double m)
{
    double s = 0.0;
    for (int i = 0; i < n; ++i) { double d = v[i] - m; s += d * d; }
    return s / (n - 1);
}

int gpcc_oracle_precompute(int L, const int *Nl, const double *y, int marginalise_b,
                           double *mean_b, double *Sigma_b, double *resid)
{
    long off = 0;
    for (int l = 0; l < L; ++l) {
        if (Nl[l] <= 0) return -4;
        double m = mean_of(y + off, Nl[l]);
        mean_b[l] = m;
        /* marginaliseb.jl:94: Sigma_b = 100*diagm(var(y_l)); fixed-b variant has no B term. */
        Sigma_b[l] = marginalise_b ? 100.0 * var_of(y + off, Nl[l], m) : 0.0;
        for (int n = 0; n < Nl[l]; ++n) resid[off + n] = y[off + n] - m;
        off += Nl[l];
    }
    return 0;
}

int gpcc_oracle_model_matrix(int kernel_id, int L, const int *Nl, const double *t, const double *y,
                             const double *sigma, int marginalise_b, const double *delays,
                             const double *alpha, double rho, double *K, double *resid)
{
    long N = 0;
    for (int l = 0; l < L; ++l) N += Nl[l];
    int rc = gpcc_oracle_covariance(kernel_id, L, alpha, delays, rho, Nl, t, Nl, t, K);
    if (rc) return rc;
    double *mean_b = (double *)malloc(sizeof(double) * 2 * L);
    double *Sigma_b = mean_b + L;
    rc = gpcc_oracle_precompute(L, Nl, y, marginalise_b, mean_b, Sigma_b, resid);
    if (rc) { free(mean_b); return rc; }
    /* K = delayedCovariance + Sobs + B  (marginaliseb.jl:135): (k + sobs) + B, in that order. */
    long c0 = 0;
    for (int j = 0; j < L; ++j) {
        for (int m = 0; m < Nl[j]; ++m) {
            long col = c0 + m;
            long r0 = 0;
            for (int i = 0; i < L; ++i) {
                for (int n = 0; n < Nl[i]; ++n) {
                    long row = r0 + n;
                    double v = K[col * N + row];
                    v = v + ((row == col) ? sigma[row] * sigma[row] : 0.0);
                    if (marginalise_b) v = v + ((i == j) ? Sigma_b[i] : 0.0);
                    K[col * N + row] = v;
                }
                r0 += Nl[i];
            }
        }
        c0 += Nl[j];
    }
    free(mean_b);
    return 0;
}

/* src/getprobabilities.jl:10-20 with StatsFuns.logsumexp (max-shifted). */
int gpcc_oracle_probabilities(int G, const double *loglik, const double *logprior, double *out)
{
    if (G <= 0) return -1;
    double mx = -INFINITY;
    for (int g = 0; g < G; ++g) {
        double j = loglik[g] + (logprior ? logprior[g] : 1.0); /* getprobabilities.jl:3: ones */
        out[g] = j;
        if (j > mx) mx = j;
    }
    double s = 0.0;
    for (int g = 0; g < G; ++g) s += exp(out[g] - mx);
    double lse = mx + log(s);
    for (int g = 0; g < G; ++g) out[g] = exp(out[g] - lse);
    return 0;
}
