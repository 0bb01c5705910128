// tools/microbench2.hip -- f64 MFMA issue rate vs waves per SIMD (4 independent accumulators per wave).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int THREADS, int NACC>
__global__ __launch_bounds__(THREADS) void probe(double *sink, int iters, double seed)
{
    double x = seed + threadIdx.x * 1e-9, y = 1.0 - threadIdx.x * 1e-9;
    d4 a[NACC];
    for (int i = 0; i < NACC; ++i) a[i] = (d4){seed * i, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) a[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += a[i][0] + a[i][1] + a[i][2] + a[i][3];
    if (s == 12345.6789) sink[0] = s;
}
template <int THREADS, int NACC>
void run(int blocks_per_cu, int iters)
{
    double *sink; hipMalloc(&sink, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int blocks = 256 * blocks_per_cu;
    for (int i = 0; i < 10; ++i) probe<THREADS, NACC><<<blocks, THREADS>>>(sink, iters, 1.000001);
    hipEventRecord(e0);
    probe<THREADS, NACC><<<blocks, THREADS>>>(sink, iters, 1.000001);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * (THREADS / 64) * iters * NACC * 2048.0;
    printf("threads %4d x %d blocks/CU (%2d waves/SIMD), %d acc: %7.3f ms  %6.1f TF\n", THREADS, blocks_per_cu,
           THREADS / 256 * blocks_per_cu, NACC, ms, flops / (ms * 1e-3) / 1e12);
    hipFree(sink);
}
int main()
{
    const int it = 20000;
    run<256, 4>(1, it); run<256, 8>(1, it);
    run<512, 4>(1, it); run<512, 8>(1, it);
    run<256, 4>(2, it); run<256, 8>(2, it);
    run<256, 4>(3, it); run<256, 8>(3, it);
    run<512, 4>(2, it); run<512, 8>(2, it);
    run<1024, 4>(1, it); run<1024, 8>(1, it);
    run<256, 4>(6, it); run<256, 4>(8, it); run<1024, 4>(2, it);
    return 0;
}
