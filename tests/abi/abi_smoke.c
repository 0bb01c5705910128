/* A plain C host over include/gpcc_hip.h -- no Python, no torch in the process: what a ccall / cgo / JNI binding sees.
 * Built and run by tests/test_gpu_parity.py::test_c_abi_from_plain_c (gcc, linked against libgpcc_hip.so).
 * Light curves come from closed formulas so that the Python side of the test can rebuild them exactly. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "gpcc_hip.h"

#define CHECK(call)                                                                     \
    do {                                                                                \
        int rc_ = (call);                                                               \
        if (rc_ != 0) {                                                                 \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, gpcc_last_error(h));          \
            return 1;                                                                   \
        }                                                                               \
    } while (0)

int main(void)
{
    enum { L = 2, N0 = 150, N1 = 131, N = N0 + N1, M = 5, G = 3 };
    static double t[N], y[N], sg[N];
    int Nl[L] = {N0, N1};
    gpcc_handle_t h = NULL;
    for (int i = 0; i < N; ++i) {   /* irregular, unsorted times; smooth signal + deterministic "noise" */
        const int band = i >= N0, j = band ? i - N0 : i;
        t[i] = fmod(7.3 * j + 0.37 * j * j, 50.0) + 0.01 * j;
        y[i] = (band ? 15.0 : 6.0) + (band ? 1.5 : 1.0) * sin(0.35 * (t[i] - (band ? 2.0 : 0.0))) + 0.3 * cos(12.9898 * i);
        sg[i] = 0.4 + 0.1 * fabs(sin(1.7 * i));
    }
    CHECK(gpcc_create(&h, L, Nl, t, y, sg, GPCC_KERNEL_MATERN32, 1, GPCC_PRECISION_FP64, 0));
    double delays[M * L], alpha[M * L], rho[M], ll[M], prob[M];
    int info[M];
    for (int m = 0; m < M; ++m) {
        delays[m * L] = 0.0;
        delays[m * L + 1] = 1.0 * m;
        alpha[m * L] = 1.0 + 0.1 * m;
        alpha[m * L + 1] = 1.5 - 0.1 * m;
        rho[m] = 3.0 + 0.5 * m;
    }
    CHECK(gpcc_loglik_batch(h, M, delays, alpha, rho, ll, info));
    CHECK(gpcc_probabilities(M, ll, NULL, prob, 0));
    for (int m = 0; m < M; ++m) printf("loglik %d %.17g %d %.17g\n", m, ll[m], info[m], prob[m]);
    /* the per-delay fit, library-side random candidates (seed 7) */
    double gd[G * L] = {0.0, 0.0, 0.0, 2.0, 0.0, 4.0}, fl[G], fa[G * L], fr[G];
    int fi[G], its[G];
    long long stats[2];
    CHECK(gpcc_grid_loglik(h, G, gd, 20, 1, 3, 0.1, 30.0, 7ull, NULL, fl, fa, fr, fi, its, stats));
    for (int g = 0; g < G; ++g) printf("fit %d %.17g %.17g %.17g %.17g %d %d\n", g, fl[g], fa[g * L], fa[g * L + 1], fr[g], fi[g], its[g]);
    printf("stats %lld %lld\n", stats[0], stats[1]);
    /* error reporting across the boundary: no exception, a code and a message */
    rho[0] = -1.0;
    CHECK(gpcc_loglik_batch(h, 1, delays, alpha, rho, ll, info));
    printf("badrho %d %d\n", info[0], isnan(ll[0]) ? 1 : 0);
    printf("nullcall %d\n", gpcc_loglik_batch(h, 1, NULL, alpha, rho, ll, info));
    CHECK(gpcc_destroy(h));
    printf("done\n");
    return 0;
}
