// Dependent-chain latencies of the instructions on the pivot chain of gpcc_diag_factor (one wave, s_memtime).
// hipcc --offload-arch=gfx950 -O3 tools/latbench.hip -o tools/latbench && ./tools/latbench
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 256
__global__ void k(double *io, unsigned long long *t)
{
    double x = io[threadIdx.x], y = io[64 + threadIdx.x];
    unsigned long long t0, t1;
    // (a) dependent v_fma_f64
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < N; ++i) x = __builtin_fma(x, y, y);
    asm volatile("" ::"v"(x));
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) t[0] = t1 - t0;
    // (b) dependent v_rsq_f64
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < N; ++i) x = __builtin_amdgcn_rsq(x);
    asm volatile("" ::"v"(x));
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) t[1] = t1 - t0;
    // (c) readlane (2x) -> fma with SGPR operand -> readlane ...
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(x), i & 15);
        const int hi = __builtin_amdgcn_readlane(__double2hiint(x), i & 15);
        x = __builtin_fma(y, __hiloint2double(hi, lo), y);
    }
    asm volatile("" ::"v"(x));
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) t[2] = t1 - t0;
    // (d) independent fma stream (16 accumulators)
    double a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = x + i;
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < N; ++i) a[i & 15] = __builtin_fma(a[i & 15], y, y);
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("" ::"v"(a[i]));
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) t[3] = t1 - t0;
    // (e) dependent v_mul_f64
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < N; ++i) x = x * y;
    asm volatile("" ::"v"(x));
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) t[4] = t1 - t0;
    // (f) independent readlane pairs + fma into 16 accumulators (the trailing-row update pattern)
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(y), i & 15);
        const int hi = __builtin_amdgcn_readlane(__double2hiint(y), i & 15);
        a[i & 15] = __builtin_fma(x, __hiloint2double(hi, lo), a[i & 15]);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("" ::"v"(a[i]));
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) t[5] = t1 - t0;
    double s = x;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    io[threadIdx.x] = s;
}
int main()
{
    double *io; unsigned long long *t;
    hipMalloc(&io, 128 * 8); hipMalloc(&t, 64);
    double h[128]; for (int i = 0; i < 128; ++i) h[i] = 1.0 + 1e-3 * i;
    hipMemcpy(io, h, sizeof h, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, 1, 64, 0, 0, io, t);
    unsigned long long ht[8]; hipMemcpy(ht, t, 64, hipMemcpyDeviceToHost);
    const char *nm[] = {"dependent v_fma_f64", "dependent v_rsq_f64", "readlane x2 -> fma chain", "independent fma (16 acc)",
                        "dependent v_mul_f64", "independent readlane x2 + fma"};
    for (int i = 0; i < 6; ++i) printf("%-32s %6.1f cycles per step (memtime ticks / %d)\n", nm[i], (double)ht[i] / N, N);
    return 0;
}
