// potf2_bench.hip -- gpcc_potf2_core (the 16 x 16 Cholesky + inverse in the registers of one wave: the serial heart of every diagonal
// step) alone on a CU: core-clock cycles (s_memtime) and wall time (s_memrealtime, 100 MHz) per call, with 1 wave, and with the
// 8-wave workgroup of the diagonal kernels in which the other waves idle at a barrier or spin on MFMAs.
// hipcc --offload-arch=gfx950 -O3 -I gpcc.jl_amd/csrc tools/potf2_bench.hip -o tools/potf2_bench
#include "gpcc_kernels.hip.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(512) void bench(const double *D, double *out, unsigned long long *ticks, int reps, int mode)
{
    __shared__ double sD[16 * 17], sr[128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, q = lane >> 4;
    if (tid < 256) sD[(tid >> 4) * 17 + (tid & 15)] = D[tid];
    __syncthreads();
    if (wave == 0) {
        double acc = 0.0;
        const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
        for (int it = 0; it < reps; ++it) {
            double v[16];
#pragma unroll
            for (int cc = 0; cc < 16; ++cc) v[cc] = (q != 0) ? ((cc == lr) ? 1.0 : 0.0) : sD[lr * 17 + cc];
            double py = 1.0, quad = 0.0, rs = 0.0, rm = 0.0;
            int pe = 0;
            const int bad = gpcc_potf2_core<false, false>(v, sr, lr, q, lane, false, py, pe, quad, sr, rs, rm);
            acc += v[15] + py + pe + bad;
            asm volatile("" : "+v"(acc));
        }
        const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
        if (lane == 0) {
            ticks[0] = c1 - c0;
            ticks[1] = w1 - w0;
        }
        out[lane] = acc;
    } else if (mode == 1 && wave != 4) {   // the other SIMDs busy with MFMAs (not wave 0's SIMD: waves 0 and 4 share one)
        d4 a = {0, 0, 0, 0};
        for (int it = 0; it < reps * 40; ++it) a = __builtin_amdgcn_mfma_f64_16x16x4f64(1.0 + lane, 1.0, a, 0, 0, 0);
        out[64 + tid] = a[0];
    } else if (mode == 2) {                // every other wave busy, wave 4 (same SIMD) included
        d4 a = {0, 0, 0, 0};
        for (int it = 0; it < reps * 40; ++it) a = __builtin_amdgcn_mfma_f64_16x16x4f64(1.0 + lane, 1.0, a, 0, 0, 0);
        out[64 + tid] = a[0];
    }
}

// one call, results out: L rows (lanes 0-15), X = inv(L) columns (lanes 16-31), prod 1/sqrt(d) as mantissa / exponent, first bad pivot
__global__ __launch_bounds__(64) void check(const double *D, double *LX, double *scal)
{
    __shared__ double sD[16 * 17], sr[128];
    const int lane = threadIdx.x & 63, lr = lane & 15, q = lane >> 4;
    for (int e = lane; e < 256; e += 64) sD[(e >> 4) * 17 + (e & 15)] = D[e];
    __syncthreads();
    double v[16];
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) v[cc] = (q != 0) ? ((cc == lr) ? 1.0 : 0.0) : sD[lr * 17 + cc];
    double py = 1.0, quad = 0.0, rs = 0.0, rm = 0.0;
    int pe = 0;
    const int bad = gpcc_potf2_core<false, false>(v, sr, lr, q, lane, false, py, pe, quad, sr, rs, rm);
    if (lane < 32) {
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) LX[lane * 16 + cc] = v[cc];
    }
    if (lane == 0) { scal[0] = py; scal[1] = pe; scal[2] = bad; }
}

static void verify(const std::vector<double> &h, const char *what)
{
    double *D, *LX, *sc;
    hipMalloc(&D, 256 * 8); hipMalloc(&LX, 512 * 8); hipMalloc(&sc, 64);
    hipMemcpy(D, h.data(), 256 * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(check, 1, 64, 0, 0, D, LX, sc);
    hipDeviceSynchronize();
    std::vector<double> lx(512); double s3[3];
    hipMemcpy(lx.data(), LX, 512 * 8, hipMemcpyDeviceToHost); hipMemcpy(s3, sc, 24, hipMemcpyDeviceToHost);
    // lanes 0-15: v[cc] = L[l][cc] (cc <= l; the rest is what the updates left behind); lanes 16-31: v[cc] = X[cc][l]
    long double eL = 0, nD = 0, eX = 0, ld = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j <= i; ++j) {
            long double a = 0;
            for (int k = 0; k <= j; ++k) a += (long double)lx[i * 16 + k] * lx[j * 16 + k];
            eL = fmaxl(eL, fabsl(a - h[i * 16 + j])); nD = fmaxl(nD, fabsl((long double)h[i * 16 + j]));
        }
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {   // (L X)[i][j] = sum_k L[i][k] X[k][j], X[k][j] = lx[(16 + j) * 16 + k]
            long double a = 0;
            for (int k = 0; k <= i; ++k) a += (long double)lx[i * 16 + k] * lx[(16 + j) * 16 + k];
            eX = fmaxl(eX, fabsl(a - (i == j ? 1.0L : 0.0L)));
        }
    for (int i = 0; i < 16; ++i) ld += logl((long double)lx[i * 16 + i]);
    const long double ldg = -(logl((long double)s3[0]) + s3[1] * 0.69314718055994530942L);
    printf("check %-28s: max |L L' - D| / max |D| = %.2Le, max |L X - I| = %.2Le, sum log L_ii %.15Lg (from the running product %.15Lg, diff %.1Le), first bad pivot %d\n",
           what, eL / nD, eX, ld, ldg, ld - ldg, (int)s3[2]);
    hipFree(D); hipFree(LX); hipFree(sc);
}

int main()
{
    std::vector<double> h(256);
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) h[i * 16 + j] = (i == j ? 20.0 : 0.0) + 1.0 / (1.0 + (i > j ? i - j : j - i));
    double *D, *out; unsigned long long *t;
    hipMalloc(&D, 256 * 8); hipMalloc(&out, 1024 * 8); hipMalloc(&t, 64);
    hipMemcpy(D, h.data(), 256 * 8, hipMemcpyHostToDevice);
    verify(h, "well conditioned");
    {
        std::vector<double> g(256);   // an ill-conditioned one: a smooth kernel on close points plus a small nugget (cond ~ 1e9)
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) g[i * 16 + j] = 4.0e3 * exp(-0.5 * (i - j) * (i - j) / 36.0) + (i == j ? 1e-3 : 0.0);
        verify(g, "ill conditioned (rbf + 1e-3)");
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) g[i * 16 + j] = 1e-12 * ((i == j ? 3.0 : 0.0) + 1.0 / (1.0 + abs(i - j)));
        verify(g, "tiny scale (1e-12)");
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) g[i * 16 + j] = (i == j ? 1.0 : 0.0) + ((i == 9 && j == 9) ? -1.5 : 0.0) + 0.01 / (1.0 + abs(i - j));
        verify(g, "pivot 9 negative");
    }
    const int reps = 2000;
    const char *names[] = {"wave 0 alone (others at the end of the kernel)", "waves on the OTHER three SIMDs issue fp64 MFMAs", "all other waves issue fp64 MFMAs (wave 4 shares wave 0's SIMD)"};
    for (int threads : {64, 512})
        for (int mode = 0; mode < (threads == 64 ? 1 : 3); ++mode) {
            for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(bench, 1, threads, 0, 0, D, out, t, reps, mode);
            hipDeviceSynchronize();
            unsigned long long ht[2]; hipMemcpy(ht, t, 16, hipMemcpyDeviceToHost);
            printf("%3d threads, %-66s: %8.1f s_memtime ticks, %6.3f us per 16 x 16 potf2 + inverse (16 pivots): %5.1f ticks, %5.1f ns per pivot; ticks per us %.0f\n", threads,
                   names[mode], (double)ht[0] / reps, (double)ht[1] / reps / 100.0, (double)ht[0] / reps / 16, (double)ht[1] / reps / 100.0 / 16 * 1e3,
                   (double)ht[0] / ((double)ht[1] / 100.0));
        }
    return 0;
}
