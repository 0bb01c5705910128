// gpcc_small.hip.h -- the small-N family of libgpcc_hip.so (gfx950 / CDNA4 only): objective(alpha, rho) of
// /root/reference/src/gpccfixdelay_marginaliseb.jl:133-141 (src/gpccfixdelay.jl:131-139 without the B term) for the sizes
// the reference's own documentation runs -- N = 110 (60 + 50, README.md:156-211) and N = 150 (60 + 50 + 40, README.md:214-287;
// simulatedata.jl:119) -- in ONE launch per batch: assembly, Cholesky, forward substitution, log-determinant and the
// Gaussian log-density, with the matrix never leaving the register file.
//
// One WAVE per evaluation.  The N x N matrix K, bordered by the right-hand side r = Y - bbar as row/column N (and identity
// padding up to 16 NB), lives as the UPPER triangle of 16 x 16 blocks in MFMA accumulator layout: block (j, i), j <= i,
// element [a][b] = K[16 j + a][16 i + b] sits in lane (lr = b, q = a & 3), register a >> 2  (v_mfma_f64_16x16x4_f64 C/D map:
// row = q + 4 reg, col = lane & 15).  NB (NB + 1) / 2 blocks x 8 VGPRs: 224 registers at N = 110, 440 at N = 150 -- gfx950's
// 512-entry unified VGPR/AGPR file is the largest on-chip store of a CU (512 KiB against 160 KiB of LDS).
// Why that layout: an accumulator block X, as it sits in registers, is at the same time a valid MFMA B operand (B[k][col] with
// k = its row index) and a valid A operand FOR ITS TRANSPOSE (A[row][k] = X[k][row]: lane (lr, q), k-step s holds X[q + 4 s][lr]
// = its own register s).  The right-looking upper Cholesky  K = U'U  then needs no data movement at all outside the
// 16 x 16 diagonal blocks:
//     panel     U[jb][i]  = inv(L_D) K[jb][i]        (L_D = U[jb][jb]', lower)   A = inv(L_D) from LDS, B = block registers
//     trailing  K[j][i]  -= U[jb][j]' U[jb][i]                                    A = B-side registers of two blocks
// i.e. 4 MFMAs per block and step, operands straight from registers.  The blocks are kept NEGATED (S = -K) so that the
// trailing update is a plain accumulation (no negated operand copies); the panel multiplies by -inv(L_D) instead.
// Only the diagonal block takes the detour through LDS: 16 x 16 potf2 + triangular inverse in "lane owns a row / a column"
// form (the register factorisation of gpcc_diag_body, gpcc_kernels.hip.h).
// The right-hand side rides along as column N of the bordered matrix: after the last real pivot the Schur complement at
// (N, N) is -w'w (w = L^-1 r) -- the quadratic form of Distributions.logpdf -- and the forward substitution is the same
// MFMAs as the factorisation.  Pivot N is skipped (set to 1), padding pivots are 1: sum log L_ii is unchanged.
//
// Control flow: the block loops are DYNAMIC (one copy of the element code, one copy of the 16 x 16 factorisation) and
// dispatch through switch statements to statically indexed register blocks -- a fully unrolled NB = 10 instance would be
// ~80 KiB of code against a 64 KiB instruction cache.
#pragma once
#include "gpcc_kernels.hip.h"

#define GPCC_SMALL_MAXNB 10                      /* bordered size N + 1 <= 160 */
#define GPCC_SMALL_MAXN (16 * GPCC_SMALL_MAXNB - 1)
#define GPCC_SMALL_DLD 17
#define GPCC_SMALL_STAGE 4                       /* blocks per assembly group (LDS staging: 2 KiB per block) */

template <int NB>
__host__ __device__ constexpr int gpcc_sblk(int j, int i)   // index of block (j, i), j <= i, in row-major upper order
{
    return j * NB - j * (j - 1) / 2 + (i - j);
}

// panel + trailing update of step JB on the statically indexed register blocks
template <int NB, int JB>
__device__ __forceinline__ void gpcc_small_update(d4 (&acc)[NB * (NB + 1) / 2], const double (&ax)[4])
{
    typedef GpccPrec<double> PD;
#pragma unroll
    for (int i = JB + 1; i < NB; ++i) {   // U[JB][i] = (-inv(L_D)) S[JB][i]
        d4 o = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 4; ++s) o = PD::mfma(ax[s], acc[gpcc_sblk<NB>(JB, i)][s], o);
        acc[gpcc_sblk<NB>(JB, i)] = o;
    }
#pragma unroll
    for (int i = JB + 1; i < NB; ++i)
#pragma unroll
        for (int j = JB + 1; j <= i; ++j)   // S[j][i] += U[JB][j]' U[JB][i]
#pragma unroll
            for (int s = 0; s < 4; ++s)
                acc[gpcc_sblk<NB>(j, i)] = PD::mfma(acc[gpcc_sblk<NB>(JB, j)][s], acc[gpcc_sblk<NB>(JB, i)][s], acc[gpcc_sblk<NB>(j, i)]);
}

// run-time step / group index -> the statically indexed code for it (a compare chain; the register blocks stay scalars)
template <int NB, int JB>
__device__ __forceinline__ void gpcc_small_dispatch_update(int jb, d4 (&acc)[NB * (NB + 1) / 2], const double (&ax)[4])
{
    if constexpr (JB < NB - 1) {
        if (jb == JB) gpcc_small_update<NB, JB>(acc, ax);
        else gpcc_small_dispatch_update<NB, JB + 1>(jb, acc, ax);
    }
}
template <int NB, int JB>
__device__ __forceinline__ d4 gpcc_small_dispatch_diag(int jb, const d4 (&acc)[NB * (NB + 1) / 2])
{
    if constexpr (JB < NB - 1) {
        if (jb == JB) return acc[gpcc_sblk<NB>(JB, JB)];
        return gpcc_small_dispatch_diag<NB, JB + 1>(jb, acc);
    } else {
        return acc[gpcc_sblk<NB>(NB - 1, NB - 1)];
    }
}
// group G of the assembly: blocks 4 G .. 4 G + 3 from the LDS stage into their registers
template <int NB, int G>
__device__ __forceinline__ void gpcc_small_stage_load(int grp, d4 (&acc)[NB * (NB + 1) / 2], const double *stage, int lane)
{
    constexpr int NBLK = NB * (NB + 1) / 2, NGRP = (NBLK + GPCC_SMALL_STAGE - 1) / GPCC_SMALL_STAGE;
    if constexpr (G < NGRP) {
        if (grp == G) {
#pragma unroll
            for (int s = 0; s < GPCC_SMALL_STAGE; ++s)
                if (GPCC_SMALL_STAGE * G + s < NBLK) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[GPCC_SMALL_STAGE * G + s][r] = stage[(s * 4 + r) * 64 + lane];
                }
        } else {
            gpcc_small_stage_load<NB, G + 1>(grp, acc, stage, lane);
        }
    }
}

// ------------------------------------------------------------------------------------------
// gpcc_small_eval<NB, KID>: one wave = one evaluation (tau, alpha, rho) -> loglik, info.
// grid = number of evaluations, block = 64.  Needs no workspace: reads the handle's light curves, writes the outputs.
// ------------------------------------------------------------------------------------------
template <int NB, int KID>
__global__ __launch_bounds__(64) void gpcc_small_eval(GpccCtx c, GpccGroup g)
{
    typedef GpccPrec<double> PD;
    constexpr int NBLK = NB * (NB + 1) / 2, NP = 16 * NB, DLD = GPCC_SMALL_DLD;
    constexpr int NGRP = (NBLK + GPCC_SMALL_STAGE - 1) / GPCC_SMALL_STAGE;
    const int m = blockIdx.x;
    if (m >= g.cnt) return;
    const int lane = threadIdx.x & 63, lr = lane & 15, q = lane >> 4;
    const int N = c.N;
    const double *delays = g.delays + (long)(g.first + m) * c.L;
    const double *alpha = g.alpha + (long)(g.first + m) * c.L;
    const double rho = g.rho[g.first + m];
    {   // the reference's argument checks (delayedCovariance.jl:3, :5-7)
        int badarg = 0;
        for (int l = 0; l < c.L; ++l)
            if (!(alpha[l] > 0.0)) badarg = -1;
        if (badarg == 0 && rho <= 0.0) badarg = -2;
        if (badarg) {
            if (lane == 0) {
                g.out_loglik[g.first + m] = __builtin_nan("");
                g.out_info[g.first + m] = badarg;
            }
            return;
        }
    }
    const GpccKernelConst kc = gpcc_kernel_const<KID>(rho);

    __shared__ double su[NP], sa[NP], ssb[NP], ssg[NP], sres[NP];   // shifted time, amplitude, Sigma_b of the band, sigma^2 (diagonal add), Y - bbar
    __shared__ int sbd[NP];                                          // band id; -1 padding; -3 the right-hand-side row
    __shared__ double sstage[GPCC_SMALL_STAGE * 4 * 64];
    __shared__ double sD[32 * DLD], sX[16 * DLD], sr[96];   // sD rows 16..31: the identity (start values of the inverse's columns)

    const bool mb = c.marginalise_b != 0;
    for (int p = lane; p < NP; p += 64) {
        int b = -1;
        double u = 0.0, a = 0.0, sb = 0.0, sg = 1.0, rs = 0.0;   // padding: identity
        if (p < N) {
            b = c.band[p];
            u = c.t[p] - delays[b];            // x - delays[i], delayedCovariance.jl:27
            a = alpha[b];
            sb = mb ? c.sigma_b[b] : 0.0;      // B = Q Sigma_b Q', marginaliseb.jl:96, :135
            sg = c.sig2[p];                    // Sobs, :89
            rs = c.resid[p];                   // Y - bbar
        } else if (p == NP - 1) {   // the right-hand side: always the LAST row / column (local pivot 15 of the last block)
            b = -3;
            sg = 0.0;
        }
        sbd[p] = b; su[p] = u; sa[p] = a; ssb[p] = sb; ssg[p] = sg; sres[p] = rs;
    }
    for (int e = lane; e < 16 * DLD; e += 64) sD[16 * DLD + e] = (e / DLD == e % DLD) ? 1.0 : 0.0;
    __syncthreads();

    // ---- assembly: S = -(K bordered), block by block in row-major upper order, four blocks per group
    d4 acc[NBLK];
    {
        int j = 0, i = 0;
#pragma nounroll
        for (int grp = 0; grp < NGRP; ++grp) {
#pragma nounroll
            for (int s = 0; s < GPCC_SMALL_STAGE; ++s) {
                if (GPCC_SMALL_STAGE * grp + s >= NBLK) break;
                const int gc = 16 * i + lr;
                const double uc = su[gc], ac = sa[gc];
                const int bc = sbd[gc];
                const bool edge = 16 * i + 15 >= N;   // the block holds the right-hand side and/or padding (wave-uniform)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int gr = 16 * j + q + 4 * r;
                    const int br = sbd[gr];
                    const double kv = gpcc_kernel_eval<KID>(su[gr], uc, kc);   // kernel(x - delays[i], y - delays[j]; rho)
                    double val = (sa[gr] * ac) * kv;                           // scale[i] scale[j] kernel, delayedCovariance.jl:27
                    if (i == j && gr == gc) val = val + ssg[gr];               // + Sobs  (padding: 1, right-hand-side row: 0)
                    val = val + ((br == bc && br >= 0) ? ssb[gr] : 0.0);       // + B (same band)
                    if (edge) {
                        if (bc == -3 && br >= 0) val = sres[gr];               // last column = Y - bbar
                        if (br == -3 && bc >= 0) val = sres[gc];               // (its mirror inside the last diagonal block)
                    }
                    sstage[(s * 4 + r) * 64 + lane] = -val;
                }
                if (++i == NB) { ++j; i = j; }
            }
            gpcc_small_stage_load<NB, 0>(grp, acc, sstage, lane);
        }
    }

    // ---- right-looking blocked Cholesky (upper form) with the right-hand side as column N
    double py = 1.0, quad = 0.0;        // prod of the mantissas of 1 / sqrt(d_j), and r' K^-1 r
    int pe = 0, bad = 0;                // sum of their exponents; order of the first non-positive pivot
#pragma nounroll
    for (int jb = 0; jb < NB; ++jb) {
        const bool last = jb == NB - 1;
        {
            const d4 dg = gpcc_small_dispatch_diag<NB, 0>(jb, acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) sD[(q + 4 * r) * DLD + lr] = -dg[r];
        }
        __syncthreads();
        {
            // 16 x 16 potf2 + inverse in registers.  Lanes 0-15: lane l owns row l of D (v[cc] = D[l][cc]); lanes 16-31:
            // lane 16 + l owns column l of X = inv(L_D) (v[cc] = delta(cc, l) - sum_j L[cc][j] X[j][l]); lanes 32-63 shadow
            // them.  Right-looking, one instruction stream for both (see gpcc_diag_body).
            double v[16];
            const double *row = sD + ((q != 0 ? 16 : 0) + lr) * DLD;   // (X lanes: a row of the identity, through the same loads)
#pragma unroll
            for (int cc = 0; cc < 16; ++cc) v[cc] = row[cc];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                double d = gpcc_bcast(v[j], j);
                if (j == 15 && last) {   // the right-hand-side row: its Schur complement is -w'w; not a pivot
                    quad = -d;
                    d = 1.0;
                }
                if (!(d > 0.0) && bad == 0) bad = 16 * jb + j + 1;   // also catches NaN
                const double y = gpcc_rsqrt(d);
                py *= __builtin_amdgcn_frexp_mant(y);
                pe += __builtin_amdgcn_frexp_exp(y);
                v[j] *= y;
                if (j < 15) {
                    sr[q == 0 ? lr : 16 + lane] = v[j];   // column j of L_D -> LDS (the other lanes store to a dead area: no branch)
                    const double lnx = gpcc_bcast(v[j], j + 1);
                    v[j + 1] = __builtin_fma(-v[j], lnx, v[j + 1]);
#pragma unroll
                    for (int cc = j + 2; cc < 16; ++cc) v[cc] = __builtin_fma(-v[j], sr[cc], v[cc]);
                }
            }
            pe += __builtin_amdgcn_frexp_exp(py);   // renormalise the running product once per block
            py = __builtin_amdgcn_frexp_mant(py);
            if (q == 1) {   // -inv(L_D), row-major: sX[row cc][col l]
#pragma unroll
                for (int cc = 0; cc < 16; ++cc) sX[cc * DLD + lr] = -v[cc];
            }
        }
        if (bad || last) break;
        __syncthreads();
        double ax[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) ax[s] = sX[lr * DLD + q + 4 * s];
        gpcc_small_dispatch_update<NB, 0>(jb, acc, ax);
    }
    if (lane == 0) {
        // logpdf(MvNormal(bbar, K), Y) = -(N log 2pi + logdet K) / 2 - (Y - bbar)' K^-1 (Y - bbar) / 2   (marginaliseb.jl:139)
        const double log2pi = 1.8378770664093454835606594728112;
        const double ld = -(log(py) + (double)pe * 0.69314718055994530942);   // sum log L_ii
        g.out_loglik[g.first + m] = bad ? __builtin_nan("") : -((double)N * log2pi + 2.0 * ld) / 2.0 - quad / 2.0;
        g.out_info[g.first + m] = bad;
    }
}
