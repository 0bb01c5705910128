// tools/oprate.hip -- gfx950 issue rates of the fp64 helper instructions the element code uses (not part of the product).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/oprate tools/oprate.hip ; run: tools/oprate
// One workgroup of 256 threads (one wave per SIMD) per CU, 16 independent chains per lane: cycles per wave-instruction.
// Result on MI355X (round 3; s_memtime counts shader clocks, so read the "ns" column / 10 as CYCLES): v_fma/add/mul_f64 6.0-6.3,
// v_rndne_f64 ~4, v_ldexp_f64 7.0, cvt_i32_f64 / cvt_f64_i32 / cvt_f32_f64 ~4.6 each, fmax(x, const) ~9 (canonicalise + max),
// exponent add through the integer pipe 7.3: none of the helpers of the element code is a slow-rate instruction -- there is
// nothing to gain from replacing rint / ldexp / the conversions by bit tricks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

template <int OP>
__global__ __launch_bounds__(256) void probe(unsigned long long *cyc, double *sink, int iters, double seed)
{
    double f[16];
    int e[16];
    for (int i = 0; i < 16; ++i) { f[i] = seed * (i + 1) + threadIdx.x * 1e-7; e[i] = i; }
    unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (OP == 0) f[i] = __builtin_fma(f[i], 1.0000001, 1e-9);
            else if (OP == 1) f[i] = f[i] + 1e-9;
            else if (OP == 2) f[i] = f[i] * 1.0000001;
            else if (OP == 3) f[i] = __builtin_rint(f[i] * 1.5) ;                       // mul + rndne
            else if (OP == 4) f[i] = __builtin_ldexp(f[i], (it & 1) ? 1 : -1);          // ldexp
            else if (OP == 5) { e[i] = (int)f[i]; f[i] = f[i] + (double)(e[i] & 1); }    // cvt_i32_f64 + cvt_f64_i32 + add
            else if (OP == 6) f[i] = __builtin_fmax(f[i], -745.0) + 1e-9;               // max + add
            else if (OP == 7) { unsigned long long b; memcpy(&b, &f[i], 8); b += (unsigned long long)((it & 1) ? 1 : -1) << 52; memcpy(&f[i], &b, 8); }   // exponent add
            else if (OP == 8) f[i] = (f[i] + 6755399441055744.0) - 6755399441055744.0;  // magic-number rounding (2 adds)
            else if (OP == 9) f[i] = (double)(float)f[i] + 1e-9;                         // cvt_f32_f64 + cvt_f64_f32 + add
            else if (OP == 10) f[i] = __builtin_fabs(f[i] - 1.25) * 1.0000001;          // sub|abs| + mul
        }
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < 16; ++i) s += f[i] + e[i];
    if (s == 12345.6789) sink[0] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = c1 - c0;
}

template <int OP>
static void run(const char *name, int instr_per_chain_step)
{
    unsigned long long *d_c; double *d_s;
    hipMalloc(&d_c, 256 * 8); hipMalloc(&d_s, 8);
    const int iters = 4000;
    probe<OP><<<256, 256>>>(d_c, d_s, iters, 1.0);
    probe<OP><<<256, 256>>>(d_c, d_s, iters, 1.0);
    unsigned long long h[256];
    hipMemcpy(h, d_c, sizeof h, hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < 256; ++i) avg += h[i]; avg /= 256;
    // s_memtime ticks at 100 MHz; report ns per 16-chain step per wave and, assuming 2.4 GHz, cycles per instruction
    const double ns = avg * 10.0 / iters / 16.0;
    printf("%-34s %7.2f ns per chain step  = %6.1f cycles @2.4GHz  (%d instr: %5.1f cycles each)\n", name, ns, ns * 2.4, instr_per_chain_step, ns * 2.4 / instr_per_chain_step);
    hipFree(d_c); hipFree(d_s);
}

int main()
{
    run<0>("v_fma_f64", 1);
    run<1>("v_add_f64", 1);
    run<2>("v_mul_f64", 1);
    run<3>("v_mul_f64 + v_rndne_f64", 2);
    run<4>("v_ldexp_f64", 1);
    run<5>("cvt_i32_f64 + and + cvt_f64_i32 + add", 4);
    run<6>("v_max_f64 + v_add_f64", 2);
    run<7>("exponent add (v_add_u32 hi)", 1);
    run<8>("magic rounding (2 v_add_f64)", 2);
    run<9>("cvt_f32_f64 + cvt_f64_f32 + add", 3);
    run<10>("sub + |abs| mul", 2);
    return 0;
}
