// potf2_bench.hip -- gpcc_potf2_core (the 16 x 16 Cholesky + inverse in the registers of one wave: the serial heart of every diagonal
// step) alone on a CU: core-clock cycles (s_memtime) and wall time (s_memrealtime, 100 MHz) per call, with 1 wave, and with the
// 8-wave workgroup of the diagonal kernels in which the other waves idle at a barrier or spin on MFMAs.
// hipcc --offload-arch=gfx950 -O3 -I gpcc.jl_amd/csrc tools/potf2_bench.hip -o tools/potf2_bench
#include "gpcc_kernels.hip.h"
#include <cstdio>
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

int main()
{
    std::vector<double> h(256);
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) h[i * 16 + j] = (i == j ? 20.0 : 0.0) + 1.0 / (1.0 + (i > j ? i - j : j - i));
    double *D, *out; unsigned long long *t;
    hipMalloc(&D, 256 * 8); hipMalloc(&out, 1024 * 8); hipMalloc(&t, 64);
    hipMemcpy(D, h.data(), 256 * 8, hipMemcpyHostToDevice);
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
