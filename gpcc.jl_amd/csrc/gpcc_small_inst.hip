// gpcc_small_inst.hip -- the instantiations of the small-N kernel families (gpcc_small.hip.h), compiled by build.py once per
// (family, kernel id): -DGPCC_INST_WIDE=0|1 -DGPCC_INST_KID=0..3 -> one object each, eight in parallel.
#include <atomic>
#include "gpcc_small.hip.h"

#ifndef GPCC_INST_KID
#error "compile with -DGPCC_INST_WIDE=0|1 -DGPCC_INST_KID=0..3"
#endif
#define GPCC_CAT2(a, b) a##b
#define GPCC_CAT(a, b) GPCC_CAT2(a, b)
constexpr int KID = GPCC_INST_KID;

#if !GPCC_INST_WIDE
// one wave per evaluation: every block count 1 .. 12 (N <= 191); four waves per SIMD up to NB = 3, three up to NB = 7, two up to NB = 10 (NB = 9, 10 with a
// few dozen scratch accesses: measured 14.7 -> 20.0 M evaluations/s at N = 128, 12.3 -> 15.4 M/s at N = 150; NB = 11, 12: no gain), one beyond
hipError_t GPCC_CAT(gpcc_small_launch_, GPCC_INST_KID)(int nb, const GpccCtx &c, const GpccGroup &g, hipStream_t s)
{
    switch (nb) {
    case 1: gpcc_small_eval<1, KID, 4><<<g.cnt, 64, 0, s>>>(c, g); break;
    case 2: gpcc_small_eval<2, KID, 4><<<g.cnt, 64, 0, s>>>(c, g); break;
    case 3: gpcc_small_eval<3, KID, 4><<<g.cnt, 64, 0, s>>>(c, g); break;
    case 4: gpcc_small_eval<4, KID, 2><<<g.cnt, 64, 0, s>>>(c, g); break;
#ifdef GPCC_AB_SMALL_WPE2
    case 5: gpcc_small_eval<5, KID, 2><<<g.cnt, 64, 0, s>>>(c, g); break;
    case 6: gpcc_small_eval<6, KID, 2><<<g.cnt, 64, 0, s>>>(c, g); break;
    case 7: gpcc_small_eval<7, KID, 2><<<g.cnt, 64, 0, s>>>(c, g); break;
#else
    // three waves per SIMD (round 4): the two waves of round 3 both sat in a pivot chain 0.36 of the time (small_pmc_summary.json);
    // <= 168 registers costs NB = 7 23 spilled registers (NB = 5, 6: none): N = 110 33.5 -> 37.1 M evaluations/s, N = 94 +14 %, N = 78 +14 %
    case 5: gpcc_small_eval<5, KID, 3><<<g.cnt, 64, 0, s>>>(c, g); break;
    case 6: gpcc_small_eval<6, KID, 3><<<g.cnt, 64, 0, s>>>(c, g); break;
    case 7: gpcc_small_eval<7, KID, 3><<<g.cnt, 64, 0, s>>>(c, g); break;
#endif
    case 8: gpcc_small_eval<8, KID, 2><<<g.cnt, 64, 0, s>>>(c, g); break;
    case 9: gpcc_small_eval<9, KID, 2><<<g.cnt, 64, 0, s>>>(c, g); break;
    case 10: gpcc_small_eval<10, KID, 2><<<g.cnt, 64, 0, s>>>(c, g); break;
    case 11: gpcc_small_eval<11, KID, 1><<<g.cnt, 64, 0, s>>>(c, g); break;
    default: gpcc_small_eval<12, KID, 1><<<g.cnt, 64, 0, s>>>(c, g); break;
    }
    return hipGetLastError();
}
#else
// four waves per evaluation: every block count 5 .. 12 (latency-bound batches of N <= 191: the SAME count as the one-wave kernel
// uses, so that both return the same bits) and the even ones up to 24 (N <= 383: a size in between takes the next one, identity
// padding)
template <int NB, int WGS>
static hipError_t launch_one(const GpccCtx &c, const GpccGroup &g, hipStream_t s)
{
    constexpr int bytes = GpccSmallWLds<NB, 4>::bytes;
    // per device (the attribute belongs to a device and a function); the fit's slice threads launch concurrently, so the flags are
    // atomics: two threads may both set the (idempotent) attribute, none reads a torn flag
    static std::atomic<bool> attr_done[64];
    int dev = 0;
    hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_done[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void *)gpcc_smallw_eval<NB, KID, 4, WGS>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return e;
        attr_done[dev].store(true, std::memory_order_release);
    }
    gpcc_smallw_eval<NB, KID, 4, WGS><<<g.cnt, 256, bytes, s>>>(c, g);
    return hipGetLastError();
}

hipError_t GPCC_CAT(gpcc_smallw_launch_, GPCC_INST_KID)(int nb, const GpccCtx &c, const GpccGroup &g, hipStream_t s)
{
    switch (nb) {
    case 5: return launch_one<5, 2>(c, g, s);
    case 6: return launch_one<6, 2>(c, g, s);
    case 7: return launch_one<7, 2>(c, g, s);
    case 8: return launch_one<8, 2>(c, g, s);
    case 9: return launch_one<9, 2>(c, g, s);
    case 10: return launch_one<10, 2>(c, g, s);
    case 11: return launch_one<11, 2>(c, g, s);
    case 12: return launch_one<12, 2>(c, g, s);
    default: break;
    }
    if (nb < 5) return hipErrorInvalidValue;   // (the caller sends such sizes to the one-wave kernels)
    if (nb <= 14) return launch_one<14, 2>(c, g, s);
    if (nb <= 16) return launch_one<16, 2>(c, g, s);
    if (nb <= 18) return launch_one<18, 1>(c, g, s);
    if (nb <= 20) return launch_one<20, 1>(c, g, s);
    if (nb <= 22) return launch_one<22, 1>(c, g, s);
    return launch_one<24, 1>(c, g, s);
}
#endif
