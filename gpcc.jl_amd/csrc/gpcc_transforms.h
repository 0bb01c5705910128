// gpcc_transforms.h -- the parameter transforms of `unpack` (src/gpccfixdelay_marginaliseb.jl:112-126) for BOTH sides of the
// boundary: the host optimiser (gpcc_fit.h) and the device kernels that unpack on the GPU (gpcc_small.hip.h).
//   makeα(x) = makepositive(x) + 1e-8,  makeρ(x) = transformbetween(x, ρmin, ρmax)          (:112, :114)
// MiscUtil.jl's source is not under /root/reference: makepositive is taken to be softplus, transformbetween(x, a, b) =
// a + (b - a) logistic(x) (parity unpinned).
// Why own exp / log here instead of libm / ocml: the optimiser's trajectory depends on every bit of alpha and rho, and the
// same parameter vector is unpacked on the host (returned alpha, rho; tile path) and on the device (small-N path).  These
// versions use only operations that are correctly rounded on both sides -- add, multiply, fma, divide, rint, frexp, ldexp --
// with contraction switched off, so host and device produce the SAME bits (tests/test_gpu_small_n.py checks it), within
// 4 ulp of libm (tests/test_neldermead_cpu.py).
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#define GPCC_HD __host__ __device__
#else
#define GPCC_HD
#endif

namespace gpcctf {

// exp(x): Cody-Waite reduction x = n ln2 + r, |r| <= ln2/2, degree-13 Taylor/Horner (truncation 4e-18), ldexp
GPCC_HD inline double rexp(double x)
{
#pragma clang fp contract(off)
    if (x != x) return x;
    if (x > 709.0) return __builtin_huge_val();
    if (x < -745.0) return 0.0;
    const double L2E = 1.4426950408889634074, LN2HI = 6.93147180369123816490e-01, LN2LO = 1.90821492927058770002e-10;
    const double n = __builtin_rint(x * L2E);
    double r = __builtin_fma(-n, LN2HI, x);
    r = __builtin_fma(-n, LN2LO, r);
    double p = 1.6059043836821614599e-10;  // 1/13!
    p = __builtin_fma(p, r, 2.0876756987868098979e-09);
    p = __builtin_fma(p, r, 2.5052108385441718775e-08);
    p = __builtin_fma(p, r, 2.7557319223985890653e-07);
    p = __builtin_fma(p, r, 2.7557319223985890653e-06);
    p = __builtin_fma(p, r, 2.4801587301587301587e-05);
    p = __builtin_fma(p, r, 1.9841269841269841270e-04);
    p = __builtin_fma(p, r, 1.3888888888888888889e-03);
    p = __builtin_fma(p, r, 8.3333333333333333333e-03);
    p = __builtin_fma(p, r, 4.1666666666666666667e-02);
    p = __builtin_fma(p, r, 1.6666666666666666667e-01);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    // two exact scalings: n may reach -1075 (subnormal results), beyond one ldexp's exact range only at the very end
    const int ni = (int)n, h = ni / 2;
    return __builtin_ldexp(__builtin_ldexp(p, h), ni - h);
}

// log(w), w > 0 finite and normal: w = m 2^e, m in [sqrt(1/2), sqrt(2)), log m = 2 atanh(s), s = (m-1)/(m+1), |s| <= 0.1716
GPCC_HD inline double rlog(double w)
{
#pragma clang fp contract(off)
    int e;
    double m = __builtin_frexp(w, &e);   // [0.5, 1)
    if (m < 0.70710678118654752440) {
        m = m * 2.0;
        e -= 1;
    }
    const double s = (m - 1.0) / (m + 1.0), z = s * s;
    double p = 1.0 / 25.0;
    p = __builtin_fma(p, z, 1.0 / 23.0);
    p = __builtin_fma(p, z, 1.0 / 21.0);
    p = __builtin_fma(p, z, 1.0 / 19.0);
    p = __builtin_fma(p, z, 1.0 / 17.0);
    p = __builtin_fma(p, z, 1.0 / 15.0);
    p = __builtin_fma(p, z, 1.0 / 13.0);
    p = __builtin_fma(p, z, 1.0 / 11.0);
    p = __builtin_fma(p, z, 1.0 / 9.0);
    p = __builtin_fma(p, z, 1.0 / 7.0);
    p = __builtin_fma(p, z, 1.0 / 5.0);
    p = __builtin_fma(p, z, 1.0 / 3.0);
    const double LN2HI = 6.93147180369123816490e-01, LN2LO = 1.90821492927058770002e-10;
    const double t = 2.0 * s, de = (double)e;
    // log w = e ln2 + t + t z p
    return __builtin_fma(de, LN2HI, t + __builtin_fma(t * z, p, de * LN2LO));
}

// makepositive(x) = softplus(x) = log(1 + exp(x))   (x > 30: x itself, as 1 + exp(-x) rounds to 1 far earlier)
GPCC_HD inline double makepositive(double x)
{
#pragma clang fp contract(off)
    if (x > 30.0) return x;
    const double u = rexp(x), w = 1.0 + u;
    if (w == 1.0) return u;                       // exp(x) below half an ulp of 1: log1p(u) = u
    const double c = (w - 1.0) - u;               // rounding error of 1 + u (exact)
    return rlog(w) - c / w;                       // log1p(u) = log(w) - c/w + O(c^2)
}

// transformbetween(x, a, b) = a + (b - a) / (1 + exp(-x))
GPCC_HD inline double transformbetween(double x, double a, double b)
{
#pragma clang fp contract(off)
    return a + (b - a) / (1.0 + rexp(-x));
}

}   // namespace gpcctf
