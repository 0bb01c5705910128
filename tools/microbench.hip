// tools/microbench.hip -- gfx950 fp64 rate probes (not part of the product).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench tools/microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));

struct Stamp { unsigned long long c0, c1, r0, r1; };

// MODE 0: mfma 16x16x4 (8 acc); 1: mfma 4x4x4 (8 acc); 2: v_fma_f64 (32 chains);
// 3: waves alternate roles (even: mfma, odd: valu); 4: one wave interleaves 8 mfma + VPM*8 fma per iter
template <int MODE, int THREADS, int VPM>
__global__ __launch_bounds__(THREADS) void probe(Stamp *st, double *sink, int iters, double seed)
{
    const int wave = threadIdx.x >> 6;
    double x = seed + threadIdx.x * 1e-9, y = 1.0 - threadIdx.x * 1e-9;
    d4 a[8];
    double s1[8];
    for (int i = 0; i < 8; ++i) { a[i] = (d4){seed * i, 0, 0, 0}; s1[i] = seed * i; }
    double f[32];
    for (int i = 0; i < 32; ++i) f[i] = seed * i;
    const bool do_mfma = (MODE == 0 || MODE == 1 || (MODE == 3 && (wave & 1) == 0));
    const bool do_valu = (MODE == 2 || (MODE == 3 && (wave & 1) == 1));
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (MODE == 4) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                a[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a[i], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < VPM; ++r) f[(i * VPM + r) & 31] = __builtin_fma(f[(i * VPM + r) & 31], x, y);
            }
        }
    } else {
        if (do_mfma) {
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (MODE == 1) s1[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, s1[i], 0, 0, 0);
                    else a[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a[i], 0, 0, 0);
                }
            }
        }
        if (do_valu) {
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 32; ++i) f[i] = __builtin_fma(f[i], x, y);
            }
        }
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += a[i][0] + a[i][1] + a[i][2] + a[i][3] + s1[i];
    for (int i = 0; i < 32; ++i) s += f[i];
    if (s == 12345.6789) sink[0] = s;
    if ((threadIdx.x & 63) == 0) st[blockIdx.x * (THREADS >> 6) + wave] = {c0, c1, r0, r1};
}

template <int MODE, int THREADS, int VPM = 0>
void run(const char *name, int blocks, int iters)
{
    Stamp *st; double *sink;
    int nw = blocks * THREADS / 64;
    hipMalloc(&st, sizeof(Stamp) * nw); hipMalloc(&sink, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 30; ++i) probe<MODE, THREADS, VPM><<<blocks, THREADS>>>(st, sink, iters, 1.000001);
    hipEventRecord(e0);
    probe<MODE, THREADS, VPM><<<blocks, THREADS>>>(st, sink, iters, 1.000001);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<Stamp> h(nw); hipMemcpy(h.data(), st, sizeof(Stamp) * nw, hipMemcpyDeviceToHost);
    std::vector<double> cm, cv, clk;
    for (int w = 0; w < nw; ++w) {
        auto &s = h[w];
        double c = (double)(s.c1 - s.c0);
        bool is_valu = (MODE == 2) || (MODE == 3 && (w & 1));
        (is_valu ? cv : cm).push_back(c);
        clk.push_back(c / (double)(s.r1 - s.r0) * 100.0);
    }
    std::sort(cm.begin(), cm.end()); std::sort(cv.begin(), cv.end()); std::sort(clk.begin(), clk.end());
    double medm = cm.empty() ? 0 : cm[cm.size() / 2], medv = cv.empty() ? 0 : cv[cv.size() / 2], medclk = clk[nw / 2];
    double mfma_flops = 0, valu_flops = 0;
    if (MODE == 0) mfma_flops = (double)nw * iters * 8 * 2048.0;
    if (MODE == 1) mfma_flops = (double)nw * iters * 8 * 512.0;
    if (MODE == 2) valu_flops = (double)nw * iters * 32 * 128.0;
    if (MODE == 3) { mfma_flops = (double)(nw / 2) * iters * 8 * 2048.0; valu_flops = (double)(nw / 2) * iters * 32 * 128.0; }
    if (MODE == 4) { mfma_flops = (double)nw * iters * 8 * 2048.0; valu_flops = (double)nw * iters * 8 * VPM * 128.0; }
    printf("%-36s thr=%4d blk=%4d wall %7.3f ms | mfma-wave cyc/iter %8.1f valu-wave cyc/iter %8.1f | clk %5.0f MHz | mfma %5.1f + valu %5.1f = %5.1f TF\n",
           name, THREADS, blocks, ms, medm / iters, medv / iters, medclk, mfma_flops / (ms * 1e-3) / 1e12,
           valu_flops / (ms * 1e-3) / 1e12, (mfma_flops + valu_flops) / (ms * 1e-3) / 1e12);
    hipFree(st); hipFree(sink);
}

int main()
{
    const int it = 20000;
    run<0, 256>("mfma16x16x4 1w/SIMD (8/iter)", 256, it);
    run<0, 512>("mfma16x16x4 2w/SIMD", 256, it);
    run<1, 256>("mfma4x4x4 1w/SIMD (8/iter)", 256, it);
    run<1, 512>("mfma4x4x4 2w/SIMD", 256, it);
    run<2, 256>("v_fma_f64 1w/SIMD (32/iter)", 256, it);
    run<2, 512>("v_fma_f64 2w/SIMD", 256, it);
    run<2, 1024>("v_fma_f64 4w/SIMD", 256, it);
    run<3, 512>("roles mfma|valu 2w/SIMD", 256, it);
    run<3, 1024>("roles mfma|valu 4w/SIMD", 256, it);
    run<4, 256, 1>("interleave 1 fma per mfma 1w", 256, it);
    run<4, 256, 2>("interleave 2 fma per mfma 1w", 256, it);
    run<4, 256, 4>("interleave 4 fma per mfma 1w", 256, it);
    run<4, 256, 8>("interleave 8 fma per mfma 1w", 256, it);
    run<4, 256, 16>("interleave 16 fma per mfma 1w", 256, it);
    run<4, 512, 8>("interleave 8 fma per mfma 2w", 256, it);
    run<4, 512, 16>("interleave 16 fma per mfma 2w", 256, it);
    return 0;
}
