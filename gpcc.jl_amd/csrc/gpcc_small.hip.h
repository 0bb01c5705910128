// gpcc_small.hip.h -- the small-N family of libgpcc_hip.so (gfx950 / CDNA4 only): objective(alpha, rho) of
// /root/reference/src/gpccfixdelay_marginaliseb.jl:133-141 (src/gpccfixdelay.jl:131-139 without the B term) for the sizes
// the reference's own documentation runs -- N = 110 (60 + 50, README.md:156-211) and N = 150 (60 + 50 + 40, README.md:214-287;
// simulatedata.jl:119) -- in ONE launch per batch: assembly, Cholesky, forward substitution, log-determinant and the
// Gaussian log-density, with the matrix never leaving the register file.
//
// One WAVE per evaluation.  The N x N matrix K, bordered by the right-hand side r = Y - bbar as its LAST row/column (index
// 16 NB - 1; identity padding between N and it), is handled as the UPPER triangle of 16 x 16 blocks in MFMA accumulator
// layout: block (j, i), j <= i, element [a][b] = K[16 j + a][16 i + b] sits in lane (lr = b, q = a & 3), register a >> 2
// (v_mfma_f64_16x16x4_f64 C/D map: row = q + 4 reg, col = lane & 15).
// Why that layout: an accumulator block X, as it sits in registers, is at the same time a valid MFMA B operand (B[k][col] with
// k = its row index) and a valid A operand FOR ITS TRANSPOSE (A[row][k] = X[k][row]: lane (lr, q), k-step s holds X[q + 4 s][lr]
// = its own register s).  The upper Cholesky  K = U'U  then needs no data movement at all outside the 16 x 16 diagonal
// blocks -- 4 MFMAs per block product, operands straight from registers:
//     row j:  T[i] = -K[j][i]  (i >= j, assembled when the row is reached)     T[i] += sum_{m<j} U[m][j]' U[m][i]
//             16 x 16 step on -T[j]  ->  -inv(L_D)   (L_D = U[j][j]')           U[j][i] = (-inv(L_D)) T[i]   (i > j)
// (the "up-looking", row-by-row form; blocks are kept NEGATED so that the update is a plain accumulation, and the panel
// multiplies by -inv(L_D) instead).  What is live between rows is the rectangle U[0..j][j+1..NB-1] -- (j+1)(NB-1-j) <= NB^2/4
// blocks of 8 VGPRs instead of the whole triangle NB(NB+1)/2 a right-looking form keeps: 12 blocks (96 VGPRs) instead of 28 at
// N = 110, 25 (200) instead of 55 at N = 150 -- so N <= 111 runs TWO waves per SIMD (f64 MFMAs issue every 64 cycles only when
// two waves of a SIMD have some, profiles/r01/microbench_fp64.log; one wave alone: ~139) and N <= 191 fits the 512-entry unified
// VGPR/AGPR file of one wave without scratch.  gfx950's register file (512 KiB per CU) is its largest on-chip store; LDS only
// carries the staging of a row's elements, the point data and the 16 x 16 diagonal step.
// Only the diagonal block takes a detour through LDS: 16 x 16 potf2 + triangular inverse in "lane owns a row / a column"
// form (the register factorisation of gpcc_diag_body, gpcc_kernels.hip.h), software-pipelined by hand (the broadcast column
// of pivot j is applied to columns >= j+2 during the reciprocal square root of pivot j+1) and fenced per pivot.
// The right-hand side rides along as the last column: after the last real pivot its Schur complement is -w'w (w = L^-1 r)
// -- the quadratic form of Distributions.logpdf -- and the forward substitution is the same MFMAs as the factorisation.
// That pivot is skipped (set to 1), padding pivots are 1: sum log L_ii is unchanged.
// (Round 3's first version kept the whole triangle in registers, right-looking, with dynamic block loops: 16.5 M evaluations/s
// at N = 110 and 4.6 M/s with scratch spills at N = 150; this form: 30 and 15.4 M/s -- profiles/r03/.)
// N = 192 ... 383 (and batches too small to fill the chip with one wave each) run FOUR waves per evaluation, block column i owned
// by wave i % 4 -- gpcc_smallw_eval below; same arithmetic in the same order, so both forms return the same bits.
#pragma once
#include "gpcc_kernels.hip.h"
#include "gpcc_transforms.h"

#define GPCC_SMALL_MAXNB 12                      /* one wave per evaluation: bordered size N + 1 <= 192 */
#define GPCC_SMALL_MAXN (16 * GPCC_SMALL_MAXNB - 1)
#define GPCC_SMALLW_MAXNB 24                     /* four waves per evaluation: N + 1 <= 384 */
#define GPCC_SMALLW_MAXN (16 * GPCC_SMALLW_MAXNB - 1)
#define GPCC_SMALL_DLD 17
// blocks per staging pass of a row's elements (2 KiB each): half a row up to NB = 8, three blocks beyond -- NB = 9 and 10 then
// stay within 20 KiB of LDS per wave, i.e. eight waves (two per SIMD) per CU
// (round 4: ONE block per pass up to NB = 7 -- 13.1 KiB of LDS per wave, so that twelve waves (three per SIMD) fit a CU; the staging
// only exists to keep the element loop rolled, its depth changes no arithmetic)
#ifdef GPCC_AB_SMALL_WPE2   /* A/B builds only (tools/ab_small.sh): two waves per SIMD for NB = 5 .. 7, as in round 3 */
#define GPCC_SMALL_SB(NB) ((NB) <= 8 ? ((NB) + 1) / 2 : ((NB) <= 10 ? 3 : ((NB) + 1) / 2))
#else
#define GPCC_SMALL_SB(NB) ((NB) <= 7 ? 1 : (NB) <= 8 ? ((NB) + 1) / 2 : ((NB) <= 10 ? 3 : ((NB) + 1) / 2))
#endif

#ifdef GPCC_AB_SMALL_LEAN_OFF   /* A/B builds only */
#define GPCC_SMALL_LEAN(NB) false
#else
#define GPCC_SMALL_LEAN(NB) ((NB) == 9 || (NB) == 10)
#endif

// (Round 5: gpcc_potf2_core -- gpcc_kernels.hip.h, the same factorisation with its pivot-to-pivot chain taken off the lane broadcasts and
// the LDS round trip -- halves the time of ONE wave alone on a SIMD (the tile kernels' diagonal steps, the persistent launch) and returns
// the same bits; here, with two or three waves per SIMD filling each other's stalls, its extra broadcasts cost more issue slots than the
// shorter chain saves: N = 110 33.8 M instead of 37.1 M evaluations/s, N = 150 15.9 M instead of 17.3 M, same box.  So this family keeps
// the compact form.)
__device__ __forceinline__ void gpcc_small_potf2(const double *sD, double *sX, double *sr, const int lane, const bool last,
                                                 const int base, double &py, int &pe, int &bad, double &quad)
{
    constexpr int DLD = GPCC_SMALL_DLD;
    const int lr = lane & 15, q = lane >> 4;
    double v[16], cn[16];
    const double *row = sD + ((q != 0 ? 16 : 0) + lr) * DLD;   // L lanes: row lr of D; X lanes: row lr of the identity
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) { v[cc] = row[cc]; cn[cc] = 0.0; }
    double d = gpcc_bcast(v[0], 0), vp = 0.0;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (j == 15 && last) {   // the right-hand-side row: its Schur complement is -w'w; not a pivot
            quad = -d;
            d = 1.0;
        }
        if (!(d > 0.0) && bad == 0) bad = base + j + 1;   // also catches NaN
        const double y = gpcc_rsqrt(d);
        if (j >= 1) {   // the rest of pivot j-1's rank-1 update (column j had its share through the readlane below)
#pragma unroll
            for (int cc = j + 1; cc < 16; ++cc) v[cc] = __builtin_fma(-vp, cn[cc], v[cc]);
        }
        py *= __builtin_amdgcn_frexp_mant(y);
        pe += __builtin_amdgcn_frexp_exp(y);
        v[j] *= y;
        if (j < 15) {
            sr[q == 0 ? lr : 16 + lane] = v[j];   // column j of L_D -> LDS (the other lanes store to a dead area: no branch)
            const double lnx = gpcc_bcast(v[j], j + 1);
            v[j + 1] = __builtin_fma(-v[j], lnx, v[j + 1]);
            d = gpcc_bcast(v[j + 1], j + 1);
            vp = v[j];
#pragma unroll
            for (int cc = j + 2; cc < 16; ++cc) cn[cc] = sr[cc];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    pe += __builtin_amdgcn_frexp_exp(py);   // renormalise the running product once per block
    py = __builtin_amdgcn_frexp_mant(py);
    if (q == 1) {   // -inv(L_D), row-major: sX[row cc][col l]
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) sX[cc * DLD + lr] = -v[cc];
    }
}

// everything a row needs besides the register blocks (all scalarised after inlining)
struct GpccSmallState {
    const double *su, *sa, *ssb;   // shifted time; amplitude -- or, with `sep`, A_p = a_p exp(-s (u_p - c)); Sigma_b of the band
    const double *sB;              // sep: B_p = a_p exp(+s (u_p - c))   (gpcc_sep_point, gpcc_kernels.hip.h)
    bool sep;                      // the separable-exponential form is in use for this evaluation (uniform)
    const int *sbd;
    double *sstage, *sD, *sX, *sr;
    const double *sig2, *resid;
    GpccKernelConst kc;
    double kscale;   // gpcc_kernel_scale<KID>(kc)
    int N, lane;
    double py, quad;   // prod of the mantissas of 1 / sqrt(d_j); r' K^-1 r
    int pe, bad;       // sum of their exponents; order of the first non-positive pivot
};

// the four elements a lane holds of block (J, i), J <= i, of K bordered (row 16 J + q + 4 r, column 16 i + lr)
template <int KID, int NP>
__device__ __forceinline__ void gpcc_small_block(const GpccSmallState &st, const int J, const int i, double (&val)[4])
{
    // no contraction across these statements: whether "(a a') k" and the "+ Sobs" / "+ B" that follow it fuse into an fma would
    // otherwise be decided per instantiation, and the one-wave and four-wave kernels must return the same bits
#pragma clang fp contract(off)
    const int lane = st.lane, lr = lane & 15, q = lane >> 4, N = st.N;
    const int gc = 16 * i + lr;
    const double uc = st.su[gc], ac = st.sa[gc];
    const int bc = st.sbd[gc];
    const bool edge = 16 * i + 15 >= N;   // the block holds the right-hand side and/or padding (wave-uniform)
    int br[4];
    if (KID != 1 && st.sep) {   // (uniform) scale[i] scale[j] kernel from the separable factors: no exponential per element
        const double bcv = st.sB[gc];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gr = 16 * J + q + 4 * r;
            br[r] = st.sbd[gr];
            const double e = fmin(st.sa[gr] * bcv, ac * st.sB[gr]);
            if (KID == 0) val[r] = e;
            else {
                const double t = fabs(st.su[gr] - uc) * st.kscale;
                val[r] = (KID == 2) ? __builtin_fma(e, t, e) : e * __builtin_fma(t, __builtin_fma(t, 1.0 / 3.0, 1.0), 1.0);
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gr = 16 * J + q + 4 * r;
            br[r] = st.sbd[gr];
            const double kv = gpcc_kernel_eval_scaled<KID>(st.su[gr], uc, st.kscale);   // kernel(x - delays[i], y - delays[j]; rho)
            val[r] = (st.sa[gr] * ac) * kv;                                  // scale[i] scale[j] kernel, delayedCovariance.jl:27
        }
    }
    if (i == J) {   // (wave-uniform) + Sobs, marginaliseb.jl:89, :135; padding: 1, right-hand-side row: 0
        const double sgc = gc < N ? st.sig2[gc] : (gc == NP - 1 ? 0.0 : 1.0);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (q + 4 * r == lr) val[r] = val[r] + sgc;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)   // + B = Q Sigma_b Q' (same band), marginaliseb.jl:96, :135
        val[r] = val[r] + ((br[r] == bc && bc >= 0) ? st.ssb[gc] : 0.0);
    if (edge) {     // (wave-uniform) the last column = Y - bbar, and its mirror inside the last diagonal block
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gr = 16 * J + q + 4 * r;
            if (bc == -3 && br[r] >= 0) val[r] = st.resid[gr];
            if (br[r] == -3 && bc >= 0) val[r] = st.resid[gc];
        }
    }
}

// point p of the bordered problem -> the per-point LDS arrays; returns true if the separable form cannot hold this point.
// plain: store the amplitude itself in sa (the direct evaluation multiplies a a' k); else sa = A_p, sB = B_p (gpcc_sep_point).
// The same function serves the one-wave and the four-wave kernel: both must see the same bits.
template <int KID>
__device__ __forceinline__ bool gpcc_small_point(const GpccCtx &c, int p, int N, int NP, const double *delays, const double *sal, bool mb,
                                                 double kscale, double *su, double *sa, double *sB, double *ssb, int *sbd, bool plain)
{
    int b = -1;
    double u = 0.0, a = 0.0, sb = 0.0, A = 0.0, B = 0.0;
    bool outside = false;
    if (p < N) {
        b = c.band[p];
        u = c.t[p] - delays[b];            // x - delays[i], delayedCovariance.jl:27
        a = sal[b];
        sb = mb ? c.sigma_b[b] : 0.0;      // B = Q Sigma_b Q', marginaliseb.jl:96, :135
        if (KID != 1 && !plain) outside = !gpcc_sep_point(u, c.tmid, kscale, a, A, B);
    } else if (p == NP - 1) {              // the right-hand side: always the LAST row / column (local pivot 15 of the last block)
        b = -3;
    }
    sbd[p] = b; su[p] = u; ssb[p] = sb;
    sa[p] = (KID == 1 || plain) ? a : A;
    sB[p] = B;
    return outside;
}

// rows J .. NB-1 by compile-time recursion (a `#pragma unroll` of this loop is refused beyond ~16k IR instructions)
template <int NB, int KID, int J>
__device__ __forceinline__ void gpcc_small_rows(d4 (&U)[NB][NB], GpccSmallState &st)
{
    if constexpr (J < NB) {
        typedef GpccPrec<double> PD;
        constexpr int NP = 16 * NB, DLD = GPCC_SMALL_DLD;
        constexpr int SB = GPCC_SMALL_SB(NB);   // blocks per staging pass
        const int lane = st.lane, lr = lane & 15, q = lane >> 4;
        if constexpr (GPCC_SMALL_LEAN(NB)) {
            // Block-at-a-time form (round 4; NB = 9, 10, where the row-at-a-time form below spills 37 / 147 registers): the blocks of row J
            // stay in the LDS stage until they are consumed, ONE of them in registers at a time -- the diagonal block first (update,
            // 16 x 16 step), then block by block: update, times -inv(L_D).  Per block the same operations in the same order.
            double ax[4];
#pragma unroll
            for (int i0 = J; i0 < NB; i0 += SB) {
                const int i1 = (i0 + SB < NB) ? i0 + SB : NB;
#pragma nounroll
                for (int i = i0; i < i1; ++i) {
                    double val[4];
                    gpcc_small_block<KID, NP>(st, J, i, val);
#pragma unroll
                    for (int r = 0; r < 4; ++r) st.sstage[((i - i0) * 4 + r) * 64 + lane] = -val[r];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = i0; i < i1; ++i) {
                    d4 Ti;
#pragma unroll
                    for (int r = 0; r < 4; ++r) Ti[r] = st.sstage[((i - i0) * 4 + r) * 64 + lane];
#pragma unroll
                    for (int mm = 0; mm < J; ++mm)
#pragma unroll
                        for (int s = 0; s < 4; ++s) Ti = PD::mfma(U[mm][J][s], U[mm][i][s], Ti);
                    if (i == J) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) st.sD[(q + 4 * r) * DLD + lr] = -Ti[r];
                        __syncthreads();
                        __builtin_amdgcn_sched_barrier(0);
                        gpcc_small_potf2(st.sD, st.sX, st.sr, lane, J == NB - 1, 16 * J, st.py, st.pe, st.bad, st.quad);
                        __builtin_amdgcn_sched_barrier(0);
                        if (st.bad) return;
                        if constexpr (J < NB - 1) {
                            __syncthreads();
#pragma unroll
                            for (int s = 0; s < 4; ++s) ax[s] = st.sX[lr * DLD + q + 4 * s];
                        }
                    } else {
                        d4 o = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int s = 0; s < 4; ++s) o = PD::mfma(ax[s], Ti[s], o);
                        U[J][i] = o;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (J < NB - 1) {
                __syncthreads();   // sD / sX are rewritten by the next row
                __builtin_amdgcn_sched_barrier(0);
                gpcc_small_rows<NB, KID, J + 1>(U, st);
            }
            return;
        }
        d4 T[NB];
        // ---- (a) row J of S = -(K bordered): blocks (J, i), i >= J, through the LDS stage
#pragma unroll
        for (int i0 = J; i0 < NB; i0 += SB) {
            const int i1 = (i0 + SB < NB) ? i0 + SB : NB;
#pragma nounroll
            for (int i = i0; i < i1; ++i) {
                double val[4];
                gpcc_small_block<KID, NP>(st, J, i, val);
#pragma unroll
                for (int r = 0; r < 4; ++r) st.sstage[((i - i0) * 4 + r) * 64 + lane] = -val[r];
            }
#pragma unroll
            for (int i = i0; i < i1; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) T[i][r] = st.sstage[((i - i0) * 4 + r) * 64 + lane];
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- (b) T[i] += sum_{m<J} U[m][J]' U[m][i]
#pragma unroll
        for (int i = J; i < NB; ++i)
#pragma unroll
            for (int mm = 0; mm < J; ++mm)
#pragma unroll
                for (int s = 0; s < 4; ++s) T[i] = PD::mfma(U[mm][J][s], U[mm][i][s], T[i]);
        // ---- (c) the diagonal block: 16 x 16 potf2 + inverse
#pragma unroll
        for (int r = 0; r < 4; ++r) st.sD[(q + 4 * r) * DLD + lr] = -T[J][r];
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        gpcc_small_potf2(st.sD, st.sX, st.sr, lane, J == NB - 1, 16 * J, st.py, st.pe, st.bad, st.quad);
        __builtin_amdgcn_sched_barrier(0);
        if (st.bad) return;
        if constexpr (J < NB - 1) {
            // ---- (d) U[J][i] = (-inv(L_D)) T[i]
            __syncthreads();
            double ax[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) ax[s] = st.sX[lr * DLD + q + 4 * s];
#pragma unroll
            for (int i = J + 1; i < NB; ++i) {
                d4 o = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s = 0; s < 4; ++s) o = PD::mfma(ax[s], T[i][s], o);
                U[J][i] = o;
            }
            __syncthreads();   // sD / sX are rewritten by the next row
            __builtin_amdgcn_sched_barrier(0);
            gpcc_small_rows<NB, KID, J + 1>(U, st);
        }
    }
}

template <int NB, int KID, int WPE>
__global__ __launch_bounds__(64, WPE) void gpcc_small_eval(GpccCtx c, GpccGroup g)
{
    constexpr int NP = 16 * NB, DLD = GPCC_SMALL_DLD;
    constexpr int SB = GPCC_SMALL_SB(NB);
    const int m = blockIdx.x;
    if (m >= g.cnt) return;
    const int lane = threadIdx.x & 63;
    const int N = c.N;
    __shared__ double sal[GPCC_MAXL];   // the amplitudes of this evaluation
    const double *delays;
    double rho;
    if (g.xpar) {   // a request of the optimiser: unpack here (bit-identical to the host's gpcc_unpack_params)
        const double *x = g.xpar + (long)(g.first + m) * (c.L + 1);
        delays = g.delays + (long)g.xrow[g.first + m] * c.L;
        if (lane < c.L) sal[lane] = gpcctf::makepositive(x[lane]) + 1e-8;          // makeα, marginaliseb.jl:112
        rho = gpcctf::transformbetween(x[c.L], g.rhomin, g.rhomax);                // makeρ, :114
    } else {
        delays = g.delays + (long)(g.first + m) * c.L;
        if (lane < c.L) sal[lane] = g.alpha[(long)(g.first + m) * c.L + lane];
        rho = g.rho[g.first + m];
    }
    __syncthreads();
    {   // the reference's argument checks (delayedCovariance.jl:3, :5-7)
        int badarg = 0;
        for (int l = 0; l < c.L; ++l)
            if (!(sal[l] > 0.0)) badarg = -1;
        if (badarg == 0 && rho <= 0.0) badarg = -2;
        if (badarg) {
            if (lane == 0) {
                g.out_loglik[g.first + m] = __builtin_nan("");
                g.out_info[g.first + m] = badarg;
            }
            return;
        }
    }
    __shared__ double su[NP], sa[NP], sB[NP], ssb[NP];   // shifted time, amplitude (or A), B, Sigma_b of the band
    __shared__ int sbd[NP];                       // band id; -1 padding; -3 the right-hand-side row
    __shared__ double sstage[SB * 4 * 64];
    __shared__ double sD[32 * DLD], sX[16 * DLD], sr[96];   // sD rows 16..31: the identity

    const bool mb = c.marginalise_b != 0;
    const GpccKernelConst kc0 = gpcc_kernel_const<KID>(rho);
    const double kscale0 = gpcc_kernel_scale<KID>(kc0);
    bool outside = false;                         // some point's s (u - c) is beyond the range of the separable form
    for (int p = lane; p < NP; p += 64) outside |= gpcc_small_point<KID>(c, p, N, NP, delays, sal, mb, kscale0, su, sa, sB, ssb, sbd, false);
    const bool sep = KID != 1 && __builtin_amdgcn_ballot_w64(outside) == 0;
    if (KID != 1 && !sep) {                       // (uniform, rare: a huge time span over a tiny length scale) plain amplitudes instead
        for (int p = lane; p < NP; p += 64) gpcc_small_point<KID>(c, p, N, NP, delays, sal, mb, kscale0, su, sa, sB, ssb, sbd, true);
    }
    for (int e = lane; e < 16 * DLD; e += 64) sD[16 * DLD + e] = (e / DLD == e % DLD) ? 1.0 : 0.0;
    __syncthreads();

    GpccSmallState st;
    st.su = su; st.sa = sa; st.sB = sB; st.sep = sep; st.ssb = ssb; st.sbd = sbd; st.sstage = sstage; st.sD = sD; st.sX = sX; st.sr = sr;
    st.sig2 = c.sig2; st.resid = c.resid; st.kc = kc0; st.kscale = kscale0; st.N = N; st.lane = lane;
    st.py = 1.0; st.quad = 0.0; st.pe = 0; st.bad = 0;
    d4 U[NB][NB];   // finished rows: U[m][i], i > m
    gpcc_small_rows<NB, KID, 0>(U, st);
    if (lane == 0) {
        // logpdf(MvNormal(bbar, K), Y) = -(N log 2pi + logdet K) / 2 - (Y - bbar)' K^-1 (Y - bbar) / 2   (marginaliseb.jl:139)
        const double log2pi = 1.8378770664093454835606594728112;
        const double ld = -(log(st.py) + (double)st.pe * 0.69314718055994530942);   // sum log L_ii
        g.out_loglik[g.first + m] = st.bad ? __builtin_nan("") : -((double)N * log2pi + 2.0 * ld) / 2.0 - st.quad / 2.0;
        g.out_info[g.first + m] = st.bad;
    }
}


// ------------------------------------------------------------------------------------------
// gpcc_smallw_eval<NB, KID, W, WGS>: the same row-wise algorithm with ONE WORKGROUP OF W WAVES per evaluation, for
//  * 192 <= N <= 383 (NB <= 24): the live rectangle of the row-wise form -- up to NB^2/4 blocks -- no longer fits one wave's 512
//    registers but fits a CU's register file (4 waves x 512; rounds 1-2 and the first version of this round sent these sizes to the
//    tile kernels: 1.7 M evaluations/s at N = 192-256 against 8.3 M/s at N = 191, profiles/r03/size_sweep_64_to_1024.log);
//  * latency-bound batches of any smaller N (a Nelder-Mead round over a README-size grid is 100-800 evaluations: a fraction of the
//    chip's 1024 SIMDs): the assembly and the MFMAs of an evaluation run on four SIMDs instead of one.
// Block column i belongs to wave i % W: that wave assembles T[i], applies the finished rows to it, and keeps U[m][i] in its registers.
// Row J:   the owner of column J publishes U[0..J-1][J] (lane-private image: the A-operand of a transposed block is a lane's own
//          registers) in LDS  |  every wave: assemble its T[i], i >= J (own LDS stage)  || barrier ||
//          every wave: T[i] += sum_m U[m][J]' U[m][i] (A from LDS, B from registers) -- the owner: column J first, then the
//          16 x 16 step on -T[J] while the others work on their columns  || barrier ||  U[J][i] = (-inv(L_D)) T[i], i > J.
// Two barriers per row.  The running product behind sum log L_ii travels from owner to owner through LDS, so the result is the
// one-wave kernel's bit for bit (an evaluation must not depend on the batch it travels in: the optimiser relies on it); the first
// non-positive pivot stops every wave at the row's second barrier.
// LDS (dynamic): point data 36 NP bytes, column buffer (NB - 1) x 2 KiB, W stages of ceil(NB / W) x 2 KiB, the diagonal step.
// ------------------------------------------------------------------------------------------
template <int NB, int W>
struct GpccSmallWLds {
    static constexpr int NP = 16 * NB, NC = (NB + W - 1) / W, DLD = GPCC_SMALL_DLD;
    static constexpr int o_su = 0, o_sa = o_su + NP, o_sB = o_sa + NP, o_ssb = o_sB + NP, o_col = o_ssb + NP, o_stage = o_col + (NB - 1) * 256,
                         o_sD = o_stage + W * NC * 256, o_sX = o_sD + 32 * DLD, o_sr = o_sX + 16 * DLD, o_red = o_sr + 96,
                         o_sal = o_red + 2 * W + 2, o_end = o_sal + GPCC_MAXL;   // in doubles
    static constexpr int bytes = o_end * 8 + NP * 4 + 16;                        // + band ids (int) + flags
};

template <int NB, int KID, int W, int J>
__device__ __forceinline__ void gpcc_smallw_rows(d4 (&U)[NB][(NB + W - 1) / W], GpccSmallState &st, double *scol, double *sred, int *sflag, const int w)
{
    if constexpr (J < NB) {
        typedef GpccPrec<double> PD;
        constexpr int NP = 16 * NB, NC = (NB + W - 1) / W, DLD = GPCC_SMALL_DLD;
        constexpr int owner = J % W, cJ = J / W;          // wave and local column of block column J
        const int lane = st.lane, lr = lane & 15, q = lane >> 4;
        d4 T[NC];
        // ---- the owner publishes column J: U[m][J], m < J, as lane-private images
        if (w == owner) {
#pragma unroll
            for (int mm = 0; mm < J; ++mm)
#pragma unroll
                for (int s = 0; s < 4; ++s) scol[(mm * 4 + s) * 64 + lane] = U[mm][cJ][s];
        }
        // ---- (a) this wave's blocks of row J: columns i = c W + w >= J, through its own LDS stage
        {
            const int c0 = (J > w) ? (J - w + W - 1) / W : 0;
#pragma nounroll
            for (int c = c0; c < NC; ++c) {
                const int i = c * W + w;
                if (i >= NB) break;
                double val[4];
                gpcc_small_block<KID, NP>(st, J, i, val);
#pragma unroll
                for (int r = 0; r < 4; ++r) st.sstage[(c * 4 + r) * 64 + lane] = -val[r];
            }
#pragma unroll
            for (int c = 0; c < NC; ++c)
                if (c * W + w >= J && c * W + w < NB) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) T[c][r] = st.sstage[(c * 4 + r) * 64 + lane];
                }
        }
        __syncthreads();   // column J is published
        __builtin_amdgcn_sched_barrier(0);
        // ---- (b) T[i] += sum_{m<J} U[m][J]' U[m][i]; the owner takes column J first and factors it
        auto apply = [&](const bool only_cJ, const bool skip_cJ) {
#pragma unroll
            for (int mm = 0; mm < J; ++mm) {
                double a[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) a[s] = scol[(mm * 4 + s) * 64 + lane];
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    if (c * W + W - 1 < J) continue;                 // (compile time) no wave's column c reaches row J
                    if ((only_cJ && c != cJ) || (skip_cJ && c == cJ)) continue;
                    if (c * W + w >= J && c * W + w < NB) {
#pragma unroll
                        for (int s = 0; s < 4; ++s) T[c] = PD::mfma(a[s], U[mm][c][s], T[c]);
                    }
                }
            }
        };
        if (w == owner) {
            apply(true, false);
            // ---- (c) the diagonal block: 16 x 16 potf2 + inverse, while the other waves update their columns
#pragma unroll
            for (int r = 0; r < 4; ++r) st.sD[(q + 4 * r) * DLD + lr] = -T[cJ][r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_sched_barrier(0);
            // the running product of the pivots' reciprocal roots travels from owner to owner (sred[0..1]): the SAME sequence of
            // multiplications as in the one-wave kernel, so both kernels return the same bits
            st.py = sred[0];
            st.pe = (int)sred[1];
            gpcc_small_potf2(st.sD, st.sX, st.sr, lane, J == NB - 1, 16 * J, st.py, st.pe, st.bad, st.quad);
            __builtin_amdgcn_sched_barrier(0);
            if (lane == 0) {
                sred[0] = st.py;
                sred[1] = (double)st.pe;
                if (J == NB - 1) sred[2] = st.quad;
                if (st.bad) *sflag = st.bad;
            }
            if constexpr (J < NB - 1) apply(false, true);
        } else {
            if constexpr (J < NB - 1) apply(false, false);
        }
        __syncthreads();   // -inv(L_D) (and a failure, if any) is published; everybody is done with column J's images
        if (*sflag) return;
        if constexpr (J < NB - 1) {
            // ---- (d) U[J][i] = (-inv(L_D)) T[i], i > J
            double ax[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) ax[s] = st.sX[lr * DLD + q + 4 * s];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                if (c * W + W - 1 <= J) continue;                    // (compile time)
                if (c * W + w > J && c * W + w < NB) {
                    d4 o = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int s = 0; s < 4; ++s) o = PD::mfma(ax[s], T[c][s], o);
                    U[J][c] = o;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            gpcc_smallw_rows<NB, KID, W, J + 1>(U, st, scol, sred, sflag, w);
        }
    }
}

template <int NB, int KID, int W, int WGS>
__global__ __launch_bounds__(64 * W, WGS) void gpcc_smallw_eval(GpccCtx c, GpccGroup g)
{
    typedef GpccSmallWLds<NB, W> LD;
    constexpr int NP = LD::NP, NC = LD::NC, DLD = LD::DLD;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int m = blockIdx.x;
    if (m >= g.cnt) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = c.N;
    double *su = lds + LD::o_su, *sa = lds + LD::o_sa, *sB = lds + LD::o_sB, *ssb = lds + LD::o_ssb, *scol = lds + LD::o_col;
    double *sstage = lds + LD::o_stage + w * NC * 256, *sD = lds + LD::o_sD, *sX = lds + LD::o_sX, *sr = lds + LD::o_sr;
    double *sred = lds + LD::o_red, *sal = lds + LD::o_sal;
    int *sbd = (int *)(lds + LD::o_end), *sflag = sbd + NP;
    const double *delays;
    double rho;
    if (g.xpar) {   // a request of the optimiser: unpack here (bit-identical to the host's gpcc_unpack_params)
        const double *x = g.xpar + (long)(g.first + m) * (c.L + 1);
        delays = g.delays + (long)g.xrow[g.first + m] * c.L;
        if (tid < c.L) sal[tid] = gpcctf::makepositive(x[tid]) + 1e-8;             // makeα, marginaliseb.jl:112
        rho = gpcctf::transformbetween(x[c.L], g.rhomin, g.rhomax);                // makeρ, :114
    } else {
        delays = g.delays + (long)(g.first + m) * c.L;
        if (tid < c.L) sal[tid] = g.alpha[(long)(g.first + m) * c.L + tid];
        rho = g.rho[g.first + m];
    }
    if (tid == 0) {
        *sflag = 0;
        sflag[1] = 0;    // some point outside the range of the separable form
        sred[0] = 1.0;   // running product of the mantissas of 1 / sqrt(d_j) ...
        sred[1] = 0.0;   // ... and the sum of their exponents
        sred[2] = 0.0;   // r' K^-1 r
    }
    __syncthreads();
    {   // the reference's argument checks (delayedCovariance.jl:3, :5-7)
        int badarg = 0;
        for (int l = 0; l < c.L; ++l)
            if (!(sal[l] > 0.0)) badarg = -1;
        if (badarg == 0 && rho <= 0.0) badarg = -2;
        if (badarg) {
            if (tid == 0) {
                g.out_loglik[g.first + m] = __builtin_nan("");
                g.out_info[g.first + m] = badarg;
            }
            return;
        }
    }
    const bool mb = c.marginalise_b != 0;
    const GpccKernelConst kc0 = gpcc_kernel_const<KID>(rho);
    const double kscale0 = gpcc_kernel_scale<KID>(kc0);
    bool outside = false;
    for (int p = tid; p < NP; p += 64 * W) outside |= gpcc_small_point<KID>(c, p, N, NP, delays, sal, mb, kscale0, su, sa, sB, ssb, sbd, false);
    if (KID != 1 && __builtin_amdgcn_ballot_w64(outside) != 0 && lane == 0) sflag[1] = 1;
    for (int e = tid; e < 16 * DLD; e += 64 * W) sD[16 * DLD + e] = (e / DLD == e % DLD) ? 1.0 : 0.0;
    __syncthreads();
    const bool sep = KID != 1 && sflag[1] == 0;
    if (KID != 1 && !sep) {                       // (uniform, rare) plain amplitudes for the direct evaluation
        for (int p = tid; p < NP; p += 64 * W) gpcc_small_point<KID>(c, p, N, NP, delays, sal, mb, kscale0, su, sa, sB, ssb, sbd, true);
        __syncthreads();
    }

    GpccSmallState st;
    st.su = su; st.sa = sa; st.sB = sB; st.sep = sep; st.ssb = ssb; st.sbd = sbd; st.sstage = sstage; st.sD = sD; st.sX = sX; st.sr = sr;
    st.sig2 = c.sig2; st.resid = c.resid; st.kc = kc0; st.kscale = kscale0; st.N = N; st.lane = lane;
    st.py = 1.0; st.quad = 0.0; st.pe = 0; st.bad = 0;
    d4 U[NB][NC];
    gpcc_smallw_rows<NB, KID, W, 0>(U, st, scol, sred, sflag, w);
    __syncthreads();
    if (tid == 0) {
        const int bad = *sflag;
        const double log2pi = 1.8378770664093454835606594728112;
        const double ld = -(log(sred[0]) + (double)(int)sred[1] * 0.69314718055994530942);   // sum log L_ii
        g.out_loglik[g.first + m] = bad ? __builtin_nan("") : -((double)N * log2pi + 2.0 * ld) / 2.0 - sred[2] / 2.0;
        g.out_info[g.first + m] = bad;
    }
}

// ------------------------------------------------------------------------------------------
// Launchers.  The instantiations live in gpcc_small_inst.hip, which build.py compiles once per (family, kernel id) in parallel
// (8 objects; as one translation unit with gpcc_hip.hip the library took 7.5 minutes to build): gpcc_hip.hip only sees these.
// nb = number of 16 x 16 blocks of the bordered matrix.  hipSuccess, or the error of the attribute call / launch.
// ------------------------------------------------------------------------------------------
hipError_t gpcc_small_launch_0(int nb, const GpccCtx &c, const GpccGroup &g, hipStream_t s);
hipError_t gpcc_small_launch_1(int nb, const GpccCtx &c, const GpccGroup &g, hipStream_t s);
hipError_t gpcc_small_launch_2(int nb, const GpccCtx &c, const GpccGroup &g, hipStream_t s);
hipError_t gpcc_small_launch_3(int nb, const GpccCtx &c, const GpccGroup &g, hipStream_t s);
hipError_t gpcc_smallw_launch_0(int nb, const GpccCtx &c, const GpccGroup &g, hipStream_t s);
hipError_t gpcc_smallw_launch_1(int nb, const GpccCtx &c, const GpccGroup &g, hipStream_t s);
hipError_t gpcc_smallw_launch_2(int nb, const GpccCtx &c, const GpccGroup &g, hipStream_t s);
hipError_t gpcc_smallw_launch_3(int nb, const GpccCtx &c, const GpccGroup &g, hipStream_t s);
