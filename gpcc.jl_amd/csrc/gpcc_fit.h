// Host side of gpcc_grid_loglik: the per-delay model fit of the reference, for a whole grid of candidate delays in
// lock-step.  Reference: src/gpccfixdelay_marginaliseb.jl:112-126 (unpack / makeα / makeρ), :160-176 (initial ρ),
// :188-196 (random candidates), :203-215 (getsolution: best random candidate, then Optim's NelderMead with
// Options(iterations, g_tol = 1e-6)), :222-226 (restarts), :351 (returned value = -minimum).
//
// The reference runs one strictly sequential optimiser per delay; here P = G x restarts independent minimisations
// advance together, so every optimiser round is ONE batch for gpcc_loglik_batch while each problem keeps its own
// trajectory.  The per-problem algorithm restates Optim.jl v1's NelderMead with its defaults: AdaptiveParameters
// (alpha = 1, beta = 1 + 2/n, gamma = 0.75 - 1/(2n), delta = 1 - 1/n), AffineSimplexer (a = 0.025, b = 0.5), one
// reflection per iteration followed by expansion / outside or inside contraction / shrink, convergence when
// sqrt(var(f_simplex) n/(n+1)) <= g_tol, and after the loop the centroid of the n best vertices is evaluated and
// returned if it beats the best vertex.  Optim.jl and MiscUtil.jl (makepositive, transformbetween) are not under
// /root/reference: makepositive is taken to be softplus, transformbetween(x, a, b) = a + (b - a) logistic(x).
// gpcc.jl_amd/neldermead.py holds the same algorithm in numpy; tests/ compare the two decision by decision.
#pragma once

#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

#include "gpcc_transforms.h"

namespace gpccfit {

// the forward transforms are shared with the device (bit-identical on both sides, gpcc_transforms.h); the inverses only
// prepare the random start candidates on the host
inline double makepositive(double x) { return gpcctf::makepositive(x); }
inline double invmakepositive(double y) { return y > 30.0 ? y : log(expm1(y)); }
inline double transformbetween(double x, double a, double b) { return gpcctf::transformbetween(x, a, b); }
inline double invtransformbetween(double y, double a, double b)
{
    const double u = (y - a) / (b - a);
    return log(u) - log1p(-u);
}

// xoshiro256++ seeded through splitmix64: the generator behind init_params == NULL.  (The reference draws from
// Julia's MersenneTwister(seed); a Julia caller reproduces those draws itself and passes them as init_params.)
struct Rng {
    uint64_t s[4];
    explicit Rng(uint64_t seed)
    {
        for (int i = 0; i < 4; ++i) {
            seed += 0x9E3779B97F4A7C15ull;
            uint64_t z = seed;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            s[i] = z ^ (z >> 31);
        }
    }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next()
    {
        const uint64_t r = rotl(s[0] + s[3], 23) + s[0], t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }   // [0, 1)
};

// candidates[R][C][L+1] in the optimiser's unconstrained coordinates; vary[l] = var(y_l) (n - 1)
inline void initial_params(int L, int R, int C, double rhomin, double rhomax, uint64_t seed, const double *vary, double *out)
{
    Rng rg(seed);
    std::vector<double> rho0(R);
    const double lo = rhomin + 1e-3, hi = rhomax - 1e-3;
    if (R <= 2) {                                                    // :160-176
        for (int i = 0; i < R; ++i) rho0[i] = lo + (hi - lo) * rg.uniform();
    } else {
        for (int i = 0; i < R; ++i) rho0[i] = exp(log(lo) + (log(hi) - log(lo)) * ((double)i / (R - 1)));
    }
    for (int i = 0; i < R; ++i)
        for (int c = 0; c < C; ++c) {
            double *x = out + ((long)i * C + c) * (L + 1);
            for (int l = 0; l < L; ++l) x[l] = invmakepositive(vary[l] * (rg.uniform() * (1.2 - 0.8) + 0.8));   // sampleα, :188
            x[L] = invtransformbetween(rho0[i], rhomin, rhomax);
        }
}

// f[i] = negative objective at row i of X (K x n), row i belonging to problem pidx[i]; +inf = rejected point
typedef int (*EvalFn)(void *ctx, long K, const long *pidx, const double *X, double *f);

enum Phase { REFLECT, EXPAND, OUTSIDE, INSIDE, SHRINK, FINAL, DONE };

struct BatchedNelderMead {
    long P;
    int n, iterations;
    double g_tol;
    long long f_calls = 0, rounds = 0;
    std::vector<int> it;
    // Speculative rounds (latency-bound fits: a README-size grid is 100-200 problems, a few per cent of the chip): the
    // expansion and both contraction points depend on the centroid and the reflected point only, not on f(reflected), so a
    // round of at most `speculate_max` requests evaluates all four candidates of an iteration at once and the decision
    // tree of Optim's loop is walked on the host afterwards -- ~1.6 dependent rounds per iteration become ~1.  Every problem
    // makes exactly the decisions it makes without speculation (same points, same values: needs an objective whose value
    // does not depend on the composition of the batch, which holds for the small-N kernels); f_calls counts the extra work.
    long speculate_max = 0;

    BatchedNelderMead(long P_, int n_, int iterations_, double g_tol_) : P(P_), n(n_), iterations(iterations_), g_tol(g_tol_) {}

    static double spread(const double *fs, int n1)   // sqrt(var(fs) (n1 - 1) / n1), var with n1 - 1
    {
        double sum = 0.0;
        for (int i = 0; i < n1; ++i) sum += fs[i];
        const double mean = sum / n1;
        double acc = 0.0;
        for (int i = 0; i < n1; ++i) {
            const double d = fs[i] - mean;
            acc += d * d;
        }
        return sqrt(acc / (n1 - 1) * ((double)(n1 - 1) / n1));
    }

    static void sortperm(const double *fs, int n1, int *order)   // stable, like Julia's sortperm
    {
        for (int i = 0; i < n1; ++i) order[i] = i;
        for (int i = 1; i < n1; ++i) {
            const int o = order[i];
            int j = i;
            while (j > 0 && fs[o] < fs[order[j - 1]]) {
                order[j] = order[j - 1];
                --j;
            }
            order[j] = o;
        }
    }

    int run(EvalFn eval, void *ctx, const double *x0, double *xmin, double *fmin)
    {
        const int n1 = n + 1;
        const double alpha = 1.0, beta = 1.0 + 2.0 / n, gamma = 0.75 - 1.0 / (2 * n), delta = 1.0 - 1.0 / n;
        std::vector<double> S((size_t)P * n1 * n), fs((size_t)P * n1), xr((size_t)P * n), fr(P), cen((size_t)P * n);
        std::vector<int> order((size_t)P * n1), phase(P), req(P);
        it.assign(P, 0);
        std::vector<long> pid;
        std::vector<double> X, fv;
        auto vert = [&](long p, int v) { return &S[((size_t)p * n1 + v) * n]; };
        auto evaluate = [&]() -> int {
            const long K = (long)pid.size();
            fv.resize(K);
            if (K == 0) return 0;
            const int rc = eval(ctx, K, pid.data(), X.data(), fv.data());
            if (rc) return rc;
            for (long i = 0; i < K; ++i)
                if (std::isnan(fv[i])) fv[i] = std::numeric_limits<double>::infinity();
            f_calls += K;
            rounds += 1;
            return 0;
        };
        auto push = [&](long p, const double *x) {
            pid.push_back(p);
            X.insert(X.end(), x, x + n);
        };
        auto centroid = [&](long p) {   // mean of the n best vertices
            double *c = &cen[(size_t)p * n];
            const int *o = &order[(size_t)p * n1];
            for (int d = 0; d < n; ++d) {
                double s = vert(p, o[0])[d];
                for (int j = 1; j < n; ++j) s += vert(p, o[j])[d];
                c[d] = s / n;
            }
        };

        // AffineSimplexer(a = 0.025, b = 0.5)
        for (long p = 0; p < P; ++p)
            for (int v = 0; v < n1; ++v) {
                double *x = vert(p, v);
                for (int d = 0; d < n; ++d) x[d] = x0[(size_t)p * n + d];
                if (v > 0) x[v - 1] = (1.0 + 0.5) * x0[(size_t)p * n + v - 1] + 0.025;
                push(p, x);
            }
        int rc = evaluate();
        if (rc) return rc;
        for (long p = 0; p < P; ++p) {
            for (int v = 0; v < n1; ++v) fs[(size_t)p * n1 + v] = fv[(size_t)p * n1 + v];
            sortperm(&fs[(size_t)p * n1], n1, &order[(size_t)p * n1]);
            phase[p] = (spread(&fs[(size_t)p * n1], n1) <= g_tol || iterations <= 0) ? FINAL : REFLECT;
        }

        std::vector<double> tmp(n);
        std::vector<char> finished(P);
        for (;;) {
            bool any = false;
            for (long p = 0; p < P; ++p) {
                req[p] = phase[p];
                any = any || phase[p] != DONE;
            }
            if (!any) break;
            pid.clear();
            X.clear();
            long nreflect = 0, nother = 0;
            for (long p = 0; p < P; ++p) {
                nreflect += req[p] == REFLECT;
                nother += (req[p] == SHRINK) ? n : (req[p] != REFLECT && req[p] != DONE);
            }
            const bool spec = nreflect > 0 && 4 * nreflect + nother <= speculate_max;
            // requests, grouped by phase (problems ascending inside a group)
            for (long p = 0; p < P; ++p)
                if (req[p] == REFLECT) {
                    centroid(p);
                    const double *c = &cen[(size_t)p * n], *xh = vert(p, order[(size_t)p * n1 + n]);
                    for (int d = 0; d < n; ++d) xr[(size_t)p * n + d] = c[d] + alpha * (c[d] - xh[d]);
                    push(p, &xr[(size_t)p * n]);
                }
            const long spec_base = (long)pid.size();   // speculative requests: 3 per reflecting problem (expand, outside, inside)
            if (spec)
                for (long p = 0; p < P; ++p)
                    if (req[p] == REFLECT) {
                        const double *c = &cen[(size_t)p * n], *r = &xr[(size_t)p * n];
                        for (int ph = EXPAND; ph <= INSIDE; ++ph) {
                            for (int d = 0; d < n; ++d)
                                tmp[d] = ph == EXPAND ? c[d] + beta * (r[d] - c[d])
                                       : ph == OUTSIDE ? c[d] + gamma * (r[d] - c[d]) : c[d] - gamma * (r[d] - c[d]);
                            push(p, tmp.data());
                        }
                    }
            for (int ph = EXPAND; ph <= INSIDE; ++ph)
                for (long p = 0; p < P; ++p)
                    if (req[p] == ph) {
                        const double *c = &cen[(size_t)p * n], *r = &xr[(size_t)p * n];
                        for (int d = 0; d < n; ++d)
                            tmp[d] = ph == EXPAND ? c[d] + beta * (r[d] - c[d])
                                   : ph == OUTSIDE ? c[d] + gamma * (r[d] - c[d]) : c[d] - gamma * (r[d] - c[d]);
                        push(p, tmp.data());
                    }
            for (int j = 1; j <= n; ++j)
                for (long p = 0; p < P; ++p)
                    if (req[p] == SHRINK) {
                        const int *o = &order[(size_t)p * n1];
                        const double *xl = vert(p, o[0]);
                        double *x = vert(p, o[j]);
                        for (int d = 0; d < n; ++d) x[d] = xl[d] + delta * (x[d] - xl[d]);
                        push(p, x);
                    }
            for (long p = 0; p < P; ++p)
                if (req[p] == FINAL) {
                    centroid(p);
                    push(p, &cen[(size_t)p * n]);
                }
            rc = evaluate();
            if (rc) return rc;

            // answers, in request order
            long pos = 0;
            std::fill(finished.begin(), finished.end(), 0);
            auto second = [&](long p, int ph, double v, const double *x) {   // answer to an expansion / contraction request
                int *o = &order[(size_t)p * n1];
                double *f = &fs[(size_t)p * n1];
                if (ph == EXPAND) {
                    const int hi = o[n];
                    const bool better = v < fr[p];
                    double *dst = vert(p, hi);
                    for (int d = 0; d < n; ++d) dst[d] = better ? x[d] : xr[(size_t)p * n + d];
                    f[hi] = better ? v : fr[p];
                    for (int j = n; j > 0; --j) o[j] = o[j - 1];   // the new point is the lowest
                    o[0] = hi;
                    phase[p] = REFLECT;
                    finished[p] = 1;
                } else {
                    const bool ok = ph == OUTSIDE ? v < fr[p] : v < f[o[n]];
                    if (ok) {
                        double *dst = vert(p, o[n]);
                        for (int d = 0; d < n; ++d) dst[d] = x[d];
                        f[o[n]] = v;
                        sortperm(f, n1, o);
                        phase[p] = REFLECT;
                        finished[p] = 1;
                    } else
                        phase[p] = SHRINK;
                }
            };
            long kspec = 0;
            for (long p = 0; p < P; ++p)
                if (req[p] == REFLECT) {
                    const double v = fv[pos++];
                    int *o = &order[(size_t)p * n1];
                    double *f = &fs[(size_t)p * n1];
                    fr[p] = v;
                    if (v < f[o[0]]) phase[p] = EXPAND;
                    else if (v < f[o[n - 1]]) {
                        double *x = vert(p, o[n]);
                        for (int d = 0; d < n; ++d) x[d] = xr[(size_t)p * n + d];
                        f[o[n]] = v;
                        sortperm(f, n1, o);
                        finished[p] = 1;
                    } else if (v < f[o[n]]) phase[p] = OUTSIDE;
                    else phase[p] = INSIDE;
                    if (spec) {   // the follow-up request was evaluated in this very round
                        const long q0 = spec_base + 3 * kspec++;
                        if (!finished[p]) {
                            const int ph = phase[p];
                            const long q = q0 + (ph - EXPAND);
                            second(p, ph, fv[q], &X[(size_t)q * n]);
                        }
                    }
                }
            if (spec) pos = spec_base + 3 * nreflect;
            for (int ph = EXPAND; ph <= INSIDE; ++ph)
                for (long p = 0; p < P; ++p)
                    if (req[p] == ph) {
                        const double v = fv[pos];
                        const double *x = &X[(size_t)pos * n];
                        ++pos;
                        second(p, ph, v, x);
                    }
            {
                const long base = pos;
                long ns = 0;
                for (long p = 0; p < P; ++p) ns += req[p] == SHRINK;
                long k = 0;
                for (long p = 0; p < P; ++p)
                    if (req[p] == SHRINK) {
                        int *o = &order[(size_t)p * n1];
                        double *f = &fs[(size_t)p * n1];
                        for (int j = 1; j <= n; ++j) f[o[j]] = fv[base + (long)(j - 1) * ns + k];
                        sortperm(f, n1, o);
                        phase[p] = REFLECT;
                        finished[p] = 1;
                        ++k;
                    }
                pos = base + (long)n * ns;
            }
            for (long p = 0; p < P; ++p)
                if (req[p] == FINAL) {
                    const double v = fv[pos];
                    const double *x = &X[(size_t)pos * n];
                    ++pos;
                    const int b = order[(size_t)p * n1];
                    const double fb = fs[(size_t)p * n1 + b];
                    const bool use_c = v < fb;
                    const double *src = use_c ? x : vert(p, b);
                    for (int d = 0; d < n; ++d) xmin[(size_t)p * n + d] = src[d];
                    fmin[p] = use_c ? v : fb;
                    phase[p] = DONE;
                }
            for (long p = 0; p < P; ++p)
                if (finished[p]) {
                    it[p] += 1;
                    if (spread(&fs[(size_t)p * n1], n1) <= g_tol || it[p] >= iterations) phase[p] = FINAL;
                }
        }
        return 0;
    }
};

// ------------------------------------------------------------------------------------------
// The deal of a delay grid over the devices of a multi-device handle (multi_grid_loglik, gpcc_hip.hip): device i fits the delays i,
// i + n, i + 2 n, ... (round-robin: iteration counts differ from delay to delay); its results travel as a block of `blk` =
// ceil(G / n) rows [loglik | info | iterations | rho | alpha(L)] (padding rows: NaN, 0, ...), the n blocks are gathered, and the
// caller's arrays are filled from the gathered buffer.  Pure index arithmetic, kept here so that the CPU suite pins it
// (tests/abi/fit_host_sanitize.cpp: G not divisible by n, G < n, G = 0).
// ------------------------------------------------------------------------------------------
inline long deal_count(long G, int n, int i) { return (G > i) ? (G - i + n - 1) / n : 0; }   // delays of device i
inline long deal_global(int n, int i, long j) { return j * n + i; }                          // grid index of device i's j-th delay
inline void deal_pack_rows(long blk, int L, long Gi, const double *ll, const int *info, const int *its, const double *rho, const double *alpha,
                           double *row /* blk x (L + 4) */)
{
    const int W = L + 4;
    for (long j = 0; j < blk; ++j) {
        double *q = row + (size_t)j * W;
        for (int e = 0; e < W; ++e) q[e] = 0.0;
        if (j < Gi) {
            q[0] = ll[j];
            q[1] = (double)info[j];
            q[2] = (double)its[j];
            q[3] = rho[j];
            for (int l = 0; l < L; ++l) q[4 + l] = alpha[(size_t)j * L + l];
        } else {
            q[0] = std::numeric_limits<double>::quiet_NaN();
        }
    }
}
inline void deal_scatter(long G, int n, int L, long blk, const double *gathered /* n x blk x (L + 4) */, double *loglik_out, int *info_out,
                         int *iterations_out /* may be NULL */, double *rho_out, double *alpha_out)
{
    const int W = L + 4;
    for (int i = 0; i < n; ++i) {
        const long Gi = deal_count(G, n, i);
        const double *src = gathered + (size_t)i * blk * W;
        for (long j = 0; j < Gi; ++j) {
            const long g = deal_global(n, i, j);
            const double *q = src + (size_t)j * W;
            loglik_out[g] = q[0];
            info_out[g] = (int)q[1];
            if (iterations_out) iterations_out[g] = (int)q[2];
            rho_out[g] = q[3];
            for (int l = 0; l < L; ++l) alpha_out[(size_t)g * L + l] = q[4 + l];
        }
    }
}

}   // namespace gpccfit
