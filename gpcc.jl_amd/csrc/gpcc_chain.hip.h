// gpcc_chain.hip.h -- the FEW-EVALUATION path of libgpcc_hip.so (gfx950 / CDNA4 only): ONE persistent launch per group of at most
// `chain_max` evaluations (default 12) at N >= 384 -- call site 1 of the boundary, a single objective(alpha, rho)
// (/root/reference/src/gpccfixdelay_marginaliseb.jl:133-141, called one at a time by Optim's Nelder-Mead, :145-153, :209-211).
// Compiled as its own translation unit (gpcc_chain_inst.hip); the host side sees gpcc_chain_args.h only.
//
// Why: one evaluation's blocked Cholesky (cholesky(K), marginaliseb.jl:139) is a serial chain
//     diag(k) -> solve of tile (k+1,k) -> update of tile (k+1,k+1) -> diag(k+1) -> ...
// and as launches (gpcc_panel_trsm_rows + gpcc_small_step, 2 per step) every link waits for the previous one to END: 34 us diagonal
// step + 8 us solve + 8 us update per step, 1.8 ms at N = 4096 with >= 80 % of the CUs idle.  Kernel time, not launch overhead -- so
// moving the same links into one launch would buy nothing.  What this kernel changes is WHEN a link may start:
//   * the diagonal step publishes L_kk ROW BLOCK BY ROW BLOCK (16 rows each) and inv(D_f) right behind the 16 pivots of block f -- a
//     column solve (forward substitution, gpcc_chain_trsmq) finishes column block f the moment they exist, so when the diagonal step
//     ends only the last of eight pieces is left;
//   * the solve of tile (k+1,k) (four quarter-tile jobs on four CUs) publishes L(k+1,k) COLUMN BLOCK BY COLUMN BLOCK, and the workgroup
//     that will run diag(k+1) folds each one into tile (k+1,k+1) as it arrives (held in registers: 36 lower 16x16 blocks); the last
//     block it finishes itself (S7);
//   * two CHAIN workgroups per evaluation alternate: while A runs diag(k), B builds the image of tile (k+1,k+1); when A's last
//     inv(D_7) is out, B is two hand-offs (a few us) from its first pivot.
// Everything else -- the solves of the other tiles of column k and the right-looking trailing update -- is pulled as JOBS from an
// in-order queue by all other CUs ("workers"), sequenced by per-tile counters in global memory instead of kernel boundaries.
//
// Hand-offs follow /opt/skills/guides (MI355X_MICROARCH.md, inter-workgroup visibility; cdna_hip_programming.md Guideline 16): every
// byte another workgroup reads is stored `sc1` (write-through), every storing wave drains (`s_waitcnt vmcnt(0)`), the workgroup
// meets at a barrier (or at an LDS counter whose last adder signals), ONE lane signals (an `sc1` flag store or an agent-scope atomic
// add); a consumer polls with relaxed `sc1` loads from one lane, releases its workgroup through a barrier, and loads the bytes with
// `sc1` loads only -- register loads (buffer_load_dwordx4 / global_load_dwordx2 ... sc1) and LDS-DMA (global_load_lds_dwordx4 ... sc1).
// The LDS-DMA form is not in the guide's table; tools/xcd_probe.hip measured it on the box (consumer L1-warm, uneven load, every word
// checked: profiles/r05/xcd_probe_handoff_forms_and_hop_prices.log: 0 stale words of 157 M, the plain forms 31 %) -- measured, not an
// architectural guarantee.  A hop costs 0.6 us (flag) to 1.6 us (flag + 16 KiB).
//
// No deadlock, whatever is resident: chain workgroups are the lowest block indices (dispatched first); a worker takes jobs in queue
// order and a job only waits for jobs EARLIER in that order or for the chain, so the oldest unfinished job can always run.  Every
// spin is bounded (GPCC_CHAIN_SPIN_LIMIT polls, seconds): on expiry the waiter sets the abort word, every other waiter sees it and
// leaves, and the evaluation reports info = GPCC_INFO_TIMEOUT instead of hanging the GPU.
//
// Arithmetic: the same tile algorithm as the launch-per-step path (right-looking, 128 x 128 tiles, the 16-wide blocked diagonal
// step, fused forward substitution), other summation orders inside a tile: results agree to ~1e-11 relative in the log-likelihood.
// A non-positive pivot does not stop anything (NaNs flow through, every flag is still published); the first one is reported as
// `info`, LAPACK-style, like everywhere else.  fp64 handles only (nrhs = 1).
#pragma once
#include "gpcc_chain_args.h"

// (kernel-side sizes: not in gpcc_chain_args.h, so that tuning them does not rebuild the host object)
#define GPCC_CHAIN_THREADS 512
#define GPCC_CHAIN_LDS_BYTES (129 * 1024)  /* four 32 KiB operand stages + control words; > 80 KiB on purpose: ONE workgroup per CU -- the pivot chain runs 2-3x slower beside MFMA waves */
#define GPCC_CHAIN_SPIN_LIMIT (1u << 22)
#define GPCC_CHAIN_TLD 18
#define GPCC_CHAIN_TMP_OFF (GPCC_XIMG_ELEMS + GPCC_CHAIN_MAXRHS * GPCC_TILE + GPCC_TILE + 2)   /* doubles: gpcc_chain_diag's stmp */

typedef unsigned gpcc_u4 __attribute__((ext_vector_type(4)));
#ifndef GPCC_CHAIN_FN
#define GPCC_CHAIN_FN __device__ __forceinline__
#endif

// ---- agent-scope accesses (all hand-off traffic): relaxed atomics lower to global_load / global_store ... sc1
__device__ __forceinline__ unsigned gpcc_flag_ld(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void gpcc_flag_st(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned gpcc_flag_add(unsigned *p, unsigned v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double gpcc_ld_sc1(const double *p)
{
    return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void gpcc_st_sc1(double *p, double v)
{
    __hip_atomic_store((unsigned long long *)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// 16-byte sc1 accesses through a buffer descriptor built from a wave-uniform base (aux 16 = sc1); byte offsets per lane
__device__ __forceinline__ __amdgpu_buffer_rsrc_t gpcc_rsrc(const void *base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc((void *)gpcc_uniform_ptr(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ d2 gpcc_ld16_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off)
{
    return __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 16));
}
__device__ __forceinline__ void gpcc_st16_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off, d2 v)
{
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(gpcc_u4, v), r, (int)byte_off, 0, 16);
}
// two consecutive 1 KiB LDS-DMA pieces, sc1 (gpcc_dma_piece2 with the cache policy of a hand-off)
__device__ __forceinline__ void gpcc_dma_piece2_sc1(const void *gbase, unsigned voff, unsigned lds_addr)
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 sc1\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024 sc1"
                 :
                 : "s"(lds_addr), "v"(voff), "s"(gbase)
                 : "memory", "m0");
}
// one 1 KiB piece
__device__ __forceinline__ void gpcc_dma_piece1_sc1(const void *gbase, unsigned voff, unsigned lds_addr)
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 sc1" : : "s"(lds_addr), "v"(voff), "s"(gbase) : "memory", "m0");
}
__device__ __forceinline__ void gpcc_dma_chunk_sc1(const double *gA, const double *gB, unsigned stage_addr, int wave, int lane)
{
    const int uw = __builtin_amdgcn_readfirstlane(wave);
    const unsigned voff = (unsigned)lane * 16u;
    gpcc_dma_piece2_sc1(gpcc_uniform_ptr(gA + uw * 256), voff, stage_addr + uw * 2048);
    gpcc_dma_piece2_sc1(gpcc_uniform_ptr(gB + uw * 256), voff, stage_addr + GPCC_CHUNK_BYTES + uw * 2048);
}

// ONE lane polls *p until it is >= want; false = the launch is being abandoned (somebody's spin expired, or this one did)
__device__ __forceinline__ bool gpcc_wait_ge(const unsigned *p, unsigned want, unsigned *abortw, unsigned code)
{
    for (unsigned spins = 0;; ++spins) {
        if (gpcc_flag_ld(p) >= want) return true;
        __builtin_amdgcn_s_sleep(1);
        if ((spins & 31u) == 31u && gpcc_flag_ld(abortw) != 0u) return false;
        if (spins > GPCC_CHAIN_SPIN_LIMIT) {
            gpcc_flag_st(abortw, code);
            return false;
        }
    }
}
__device__ __forceinline__ int gpcc_tile_idx(int I, int J) { return I * (I + 1) / 2 + J; }
// the flag words of evaluation m
struct GpccChainFlags {
    unsigned *abortw, *xrow, *l7, *colflag, *lcnt, *ver;
};
__device__ __forceinline__ GpccChainFlags gpcc_chain_flags(const GpccChainArgs &a, int nt, int m)
{
    GpccChainFlags f;
    unsigned *ev = a.words + a.qbase + (long)m * a.ev_words;
    const int ntiles = nt * (nt + 1) / 2;
    f.abortw = a.words;
    f.xrow = ev;
    f.l7 = ev + nt;
    f.colflag = ev + 2 * nt;
    f.lcnt = ev + 10 * nt;
    f.ver = ev + 10 * nt + ntiles;
    return f;
}

// ------------------------------------------------------------------------------------------
// The diagonal step of the chain: gpcc_diag_body's algorithm (blocked dpotf2 by 16, explicit inverse, W_k = inv(L_kk) Z_k, sum log
// L_ii, W'W, first bad pivot) on a PACKED image -- the LOWER TRIANGLE only, 36 blocks of 16 x 16 doubles (72 KiB), the inverse built
// IN PLACE (LAPACK dtrtri's trick, block-wise):
//   * block (i, j), j <= i, at gpcc_bi(i, j); element (r, c) of a block at gpcc_be(r, c): physical row r ^ bit2(r), the sixteen
//     8-byte slots of a row XOR-ed with 2 (row >> 1) -- no padding, and both MFMA fragment patterns (A operand: lane (lr, q) reads
//     [lr][q + 4 s]; B operand / accumulator: [q + 4 s][lr]) are bank-conflict-free;
//   * a diagonal block holds D_b until wave 0 has factored it and inv(L_D)^T afterwards (L_D itself is read by nobody: the panel
//     multiplies by inv(L_D), sum log L_ii comes from the register factorisation);
//   * row i of inv(L) -- X[i][j] = -inv(D_i) sum_{m=j}^{i-1} L[i][m] X[m][j] -- reads ALL of L's row i and nothing else of row i is
//     read later, so after one barrier X[i][j] overwrites L[i][j].
// Eight waves: wave 0 runs the register factorisations (lanes 0-15 own the rows of D, lanes 16-31 the columns of inv(L_D), ONE
// right-looking instruction stream for both, software-pipelined), six workers the MFMA tasks, wave 4 (wave 0's SIMD mate) stays out
// of its way -- and PUBLISHES: row block i of inv(L) is final one barrier after it was built, and wave 4 copies it to global memory
// (`ximg`, row-major 16 x 16 blocks: a lane of a consumer finds its MFMA operand as 32 contiguous bytes) during the next block
// step, then sets xrow = i + 1.  The last row block goes out on all eight waves together with W_k and the step's scalars; xrow = 9
// says all of it is there.  (History: this image was built in round 4 for a half-CU diagonal kernel, measured beside
// MFMA-saturating waves, and rejected there -- LOG.md; alone on a CU it is what lets TWO workgroups per evaluation alternate.)
// ------------------------------------------------------------------------------------------
#define GPCC_OPAQUE_LANE(a, b) asm volatile("" : "+v"(a), "+v"(b))

// block (i, j) of the image -> ximg, row-major; one wave.  Diagonal blocks hold inv(L_D)^T in the image and leave untransposed.
__device__ __forceinline__ void gpcc_chain_publish_block(const double *sB, __amdgpu_buffer_rsrc_t xr, int i, int j, int lane)
{
    const int r = lane >> 2, c0 = 4 * (lane & 3);
    double v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (i == j) ? sB[gpcc_bi(i, i) + gpcc_be(c0 + e, r)] : sB[gpcc_bi(i, j) + gpcc_be(r, c0 + e)];
    const unsigned off = (unsigned)(gpcc_bi(i, j) + r * 16 + c0) * 8u;
    d2 lo = {v[0], v[1]}, hi = {v[2], v[3]};
    gpcc_st16_sc1(xr, off, lo);
    gpcc_st16_sc1(xr, off + 16u, hi);
}

// z_r2 -= L[r2][i] w_i (one wave; the right-hand sides along the lanes' n index, padded with zeros)
__device__ __forceinline__ void gpcc_chain_zupdate(const double *sB, double *sz, int r2, int i, int nrhs, int lrv, int qv)
{
    typedef GpccPrec<double> PD;
    const int zrow = (lrv < nrhs) ? lrv : 0;
    d4 S;
#pragma unroll
    for (int r = 0; r < 4; ++r) S[r] = sz[zrow * GPCC_TILE + r2 * 16 + qv + 4 * r];
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) {
        const double av = -sB[gpcc_bi(r2, i) + gpcc_be(lrv, qv + 4 * s2)];      // L[r2][i]
        const double wv = sz[zrow * GPCC_TILE + i * 16 + qv + 4 * s2];
        S = PD::mfma(av, (lrv < nrhs) ? wv : 0.0, S);
    }
    if (lrv < nrhs) {
#pragma unroll
        for (int r = 0; r < 4; ++r) sz[lrv * GPCC_TILE + r2 * 16 + qv + 4 * r] = S[r];
    }
}

// PRE: the image (36 blocks at smem) holds the updated lower triangle of tile (k,k), behind a barrier.  Returns false if abandoned.
GPCC_CHAIN_FN void gpcc_chain_diag(const GpccCtx &c, const GpccGroup &g, const GpccChainArgs &a, const GpccChainFlags &fl, const int k,
                                   const int m, double *smem, const int tid)
{
    typedef GpccPrec<double> PD;
    double *sB = smem;                                   // 36 blocks
    double *sz = sB + GPCC_XIMG_ELEMS;                   // nrhs x 128: Z_k, later W_k
    double *sr = sz + GPCC_CHAIN_MAXRHS * GPCC_TILE;     // [0, 80): wave 0's column scratch; [96, 112): per-block statistics
    int *sbad = (int *)(sr + GPCC_TILE);
    double *stmp = sr + GPCC_TILE + 2;                   // 16 x GPCC_CHAIN_TLD: the diagonal block about to be factored, ROW-MAJOR (a lane reads its row
                                                         // with eight 16-byte loads, conflict-free at this stride); block (0,0) arrives with the image

    const int lane = tid & 63, lr = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NWK = 6, NT = GPCC_CHAIN_THREADS;
    const int wk = (wave == 0 || wave == 4) ? -1 : (wave < 4 ? wave - 1 : wave - 2);
    const int slot = g.slot0 + m, nrhs = c.nrhs;
    const bool last = (k == c.nt - 1);
    double *xk = a.ximg + ((long)m * c.nt + k) * GPCC_XIMG_STRIDE;
    const __amdgpu_buffer_rsrc_t xr = gpcc_rsrc(xk, GPCC_XIMG_ELEMS * 8);

    if (tid == 0) *sbad = 0;
    unsigned long long *trd = a.trace ? a.trace + ((long)m * c.nt + k) * GPCC_CHAIN_TRACE_WORDS + 8 : nullptr;   // [jb][8]: (A) begins, (A) done, behind the first barrier, behind the second; wave 0: fold done, block loaded, factored, stored
    __syncthreads();
    for (int jb = 0; jb <= 8; ++jb) {
        if (trd && tid == 0) trd[8 * jb] = wall_clock64();
        const int r0 = jb * 16;
        int lrv = lr, qv = q;   // opaque per-phase copies: keeps the compiler from hoisting every lane-dependent address of all nine rounds out of the loop
        GPCC_OPAQUE_LANE(lrv, qv);
        if (wave == 6 && jb == 0) {
            // z_k: the right-hand side of this step's forward substitution.  Its last update -- z_k -= L(k,k-1) w_{k-1} -- is the END of the
            // four quarter solves of tile (k,k-1) (lcnt = 4), a hand-off later than the last column block this workgroup has just folded
            // in; it is first read by the W task of block step 1, two barriers from here: loaded now, beside the first 16 pivots
            bool okz = true;
            if (k > 0 && lane == 0) okz = gpcc_wait_ge(&fl.lcnt[gpcc_tile_idx(k, k - 1)], 4u, fl.abortw, 0x210u);
            (void)okz;   // (abandoned launch: the step finishes on stale data and the workgroup leaves at its next wait)
            for (int e = lane; e < nrhs * GPCC_TILE; e += 64)
                sz[e] = gpcc_ld_sc1(c.z + ((long)slot * nrhs + e / GPCC_TILE) * c.Np + k * GPCC_TILE + (e % GPCC_TILE));
        }
        if (wave == 7 && jb >= 1 && jb < 8) {
            // row block jb of L_kk (columns 0 .. jb-1) is final since the panel of column block jb - 1: out it goes, beside the 16 pivots of
            // D_jb; the flag follows with inv(D_jb) behind the barrier below.  (Wave 7: never has a panel task, and does not share wave
            // 0's SIMD -- whatever wave 4 issued would be taken from the pivot chain's issue slots.)
            for (int j = 0; j < jb; ++j) gpcc_chain_publish_block(sB, xr, jb, j, lane);
            if (jb == 7) {   // the last row block gets a flag of its own: S7 (gpcc_chain_trsmq) is built from it while the last 16 pivots run
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) gpcc_flag_st(&fl.l7[k], 1u);
            }
        }
        if (wave == 0) {
            if (jb > 0 && jb < 8) {   // C(jb-1) for the block the factorisation below needs: D_jb -= P_jb P_jb^T
                d4 x;
                double pa[4], pb[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) x[r] = sB[gpcc_bi(jb, jb) + gpcc_be(qv + 4 * r, lrv)];
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) {
                    pb[s2] = sB[gpcc_bi(jb, jb - 1) + gpcc_be(lrv, qv + 4 * s2)];
                    pa[s2] = -pb[s2];
                }
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) x = PD::mfma(pa[s2], pb[s2], x);
#pragma unroll
                for (int r = 0; r < 4; ++r) stmp[(qv + 4 * r) * GPCC_CHAIN_TLD + lrv] = x[r];   // (the image's block is overwritten by inv(L_D)^T below anyway)
            }
            if (trd && tid == 0) trd[8 * jb + 4] = wall_clock64();
            if (jb < 8) {
                // ---- (A) 16x16 potf2 + inverse in registers: lanes 0-15 own the rows of D, lanes 16-31 the columns of inv(L_D); ONE
                // right-looking instruction stream for both (gpcc_potf2_core)
                const bool xl = qv != 0;
                double v[16];
                double *blk = sB + gpcc_bi(jb, jb);
#pragma unroll
                for (int h2 = 0; h2 < 8; ++h2) {
                    const d2 rw = *(const d2 *)(stmp + lrv * GPCC_CHAIN_TLD + 2 * h2);
                    v[2 * h2] = xl ? ((2 * h2 == lrv) ? 1.0 : 0.0) : rw[0];
                    v[2 * h2 + 1] = xl ? ((2 * h2 + 1 == lrv) ? 1.0 : 0.0) : rw[1];
                }
                double py = 1.0;   // prod of 1/sqrt(d_j) as mantissa ...
                int pe = 0;        // ... and exponent: no overflow whatever the scale of K
                double rs_ = 0.0, rm_ = 0.0, quad_ = 0.0;
                if (trd) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (tid == 0) trd[8 * jb + 5] = wall_clock64();
                }
                const int bad = gpcc_potf2_core<false, false>(v, sr, lrv, qv, lane, false, py, pe, quad_, sr, rs_, rm_);   // (gpcc_kernels.hip.h)
                if (trd && tid == 0) trd[8 * jb + 6] = wall_clock64();
                if (lane < 32) {
                    if (xl) {   // column lrv of X = inv(L_D) as row lrv of the block: the block now holds inv(L_D)^T
#pragma unroll
                        for (int h2 = 0; h2 < 8; ++h2) {   // (gpcc_be keeps the pairs (2 h, 2 h + 1) adjacent: 16-byte stores)
                            const d2 pr2 = {v[2 * h2], v[2 * h2 + 1]};
                            *(d2 *)(blk + gpcc_be(lrv, 2 * h2)) = pr2;
                        }
                    }
                    if (lane == 0 && bad && *sbad == 0) *sbad = r0 + bad;
                    if (lane == 0) {
                        sr[96 + jb] = py;
                        sr[104 + jb] = (double)pe;
                    }
                }
                if (trd && tid == 0) trd[8 * jb + 7] = wall_clock64();
            }
        } else if (wk >= 0 && jb > 0) {
            const int jp = jb - 1;
            // ---- (C) rest of the trailing update of column block jp
            const int nb = 7 - jp, ntri = nb * (nb + 1) / 2;
            for (int t0 = 1 + wk; t0 < ntri; t0 += 2 * NWK) {
                int rfv[2], cfv[2];
                bool on[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int tt = t0 + NWK * u;
                    on[u] = tt < ntri;
                    const int te = on[u] ? tt : t0;
                    const int rr = (te >= 1) + (te >= 3) + (te >= 6) + (te >= 10) + (te >= 15) + (te >= 21);
                    rfv[u] = jp + 1 + rr;
                    cfv[u] = jp + 1 + te - rr * (rr + 1) / 2;
                }
                d4 x[2];
                double pa[2][4], pb[2][4];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) x[u][r] = sB[gpcc_bi(rfv[u], cfv[u]) + gpcc_be(qv + 4 * r, lrv)];
#pragma unroll
                    for (int s2 = 0; s2 < 4; ++s2) {
                        pa[u][s2] = -sB[gpcc_bi(rfv[u], jp) + gpcc_be(lrv, qv + 4 * s2)];
                        pb[u][s2] = sB[gpcc_bi(cfv[u], jp) + gpcc_be(lrv, qv + 4 * s2)];
                    }
                }
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) {
                    x[0] = PD::mfma(pa[0][s2], pb[0][s2], x[0]);
                    x[1] = PD::mfma(pa[1][s2], pb[1][s2], x[1]);
                }
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    if (on[u]) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) sB[gpcc_bi(rfv[u], cfv[u]) + gpcc_be(qv + 4 * r, lrv)] = x[u][r];
                    }
            }
        }
        GPCC_OPAQUE_LANE(lrv, qv);
        if (jb > 0 && wave == 1) {
            // ---- (W) w_i = inv(D_i) z_i for i = jb - 1: z_i has received every earlier w (phase 2 of the iterations before, below)
            const int i = jb - 1;
            const int zrow = (lrv < nrhs) ? lrv : 0;
            d4 Y = {0.0, 0.0, 0.0, 0.0}, Y1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double dv = sB[gpcc_bi(i, i) + gpcc_be(qv + 4 * r, lrv)];           // inv(D_i)[lrv][qv + 4 r]
                const double zv = sz[zrow * GPCC_TILE + i * 16 + qv + 4 * r];
                const double bv = (lrv < nrhs) ? zv : 0.0;
                if (r & 1) Y1 = PD::mfma(dv, bv, Y1);
                else Y = PD::mfma(dv, bv, Y);
            }
            if (lrv < nrhs) {
#pragma unroll
                for (int r = 0; r < 4; ++r) sz[lrv * GPCC_TILE + i * 16 + qv + 4 * r] = Y[r] + Y1[r];
            }
        }
        if (trd && tid == 0) trd[8 * jb + 1] = wall_clock64();
        __syncthreads();   // every reader of L's row jb-1 is done
        if (trd && tid == 0) trd[8 * jb + 2] = wall_clock64();
        if (jb < 8 && wave == 7) {   // inv(D_jb) is in the image: out at once; every store of this wave -- row block jb of L too -- has drained before the flag
            gpcc_chain_publish_block(sB, xr, jb, jb, lane);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) gpcc_flag_st(&fl.xrow[k], (unsigned)(jb + 1));
            if (trd && jb == 7 && lane == 0) trd[-5] = wall_clock64();   // (header word 3)
        }
        GPCC_OPAQUE_LANE(lrv, qv);
        if (jb > 0 && jb < 8) {
            // ---- (Z) z_r -= L[r][jb-1] w_{jb-1} for the rows below, one task per wave from wave 6 down (right-looking: the last w needs
            // ONE block product when its turn comes, not a chain of eight).  (Measured: only row jb here and the others beside the next
            // 16 pivots on one wave -- on wave 4, idle but on wave 0's SIMD, the pivots slow down by 0.2-0.9 us per block; on wave 5 the
            // six tasks in a row outlast the pivots.)
            const int r2 = jb + (6 - wave);
            if (wave <= 6 && r2 < 8) gpcc_chain_zupdate(sB, sz, r2, jb - 1, nrhs, lrv, qv);
        }
        if (jb == 8) break;
        GPCC_OPAQUE_LANE(lrv, qv);
        // ---- (B) panel of column block jb: P = A[rf, jb] inv(D_jb)^T for the row fragments below, in place
        for (int rf = jb + 1 + wave; rf < 8; rf += NT / 64) {
            double av[4], bv[4];
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                av[s2] = sB[gpcc_bi(rf, jb) + gpcc_be(lrv, qv + 4 * s2)];
                bv[s2] = sB[gpcc_bi(jb, jb) + gpcc_be(qv + 4 * s2, lrv)];   // B[k][c] = inv(D)[c][k]
            }
            d4 x = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) x = PD::mfma(av[s2], bv[s2], x);
#pragma unroll
            for (int r = 0; r < 4; ++r) sB[gpcc_bi(rf, jb) + gpcc_be(qv + 4 * r, lrv)] = x[r];
        }
        __syncthreads();
        if (trd && tid == 0) trd[8 * jb + 3] = wall_clock64();
    }
    __syncthreads();   // row block 7 of inv(L) is in the image, sz holds W_k
    // ---- the end of the step: W_k and the step's scalars -- then ONE flag value says "all of step k is there"
    for (int e = tid; e < nrhs * GPCC_TILE; e += NT)
        gpcc_st_sc1(c.w + ((long)slot * nrhs + e / GPCC_TILE) * c.Np + k * GPCC_TILE + (e % GPCC_TILE), sz[e]);
    double *sv = a.stepval + ((long)m * c.nt + k) * GPCC_CHAIN_STEPVALS;
    for (int e = wave; e < nrhs * nrhs; e += NT / 64) {   // this step's W'W: one wave per entry, fixed reduction tree
        const int ga = e / nrhs, gb = e % nrhs;
        double pr = sz[ga * GPCC_TILE + lane] * sz[gb * GPCC_TILE + lane] + sz[ga * GPCC_TILE + 64 + lane] * sz[gb * GPCC_TILE + 64 + lane];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) pr += __shfl_xor(pr, o);
        if (lane == 0) gpcc_st_sc1(sv + 2 + e, pr);
    }
    if (wave == 3) {
        double pr = 0.0;   // sum log L_jj per 16-block
        if (lane < 8) pr = -(log(sr[96 + lane]) + sr[104 + lane] * 0.69314718055994530942);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) pr += __shfl_xor(pr, o);
        if (lane == 0) {
            gpcc_st_sc1(sv, pr);
            gpcc_st_sc1(sv + 1, (double)*sbad);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains ...
    __syncthreads();
    if (tid == 0) gpcc_flag_st(&fl.xrow[k], 9u);        // ... ONE lane signals
    if (last && tid == 0) {
        // loglik = -(N log 2pi + logdet K)/2 - (Y-bbar)' K^-1 (Y-bbar)/2  (Distributions.logpdf, marginaliseb.jl:139): the steps' scalars in
        // step order, the same running sums the launch-per-step path keeps (steps of the OTHER chain workgroup were published before its
        // xrow = 9, which this workgroup has waited for, transitively, long ago)
        double ld = 0.0, *G = sr;   // (the Gram matrix in LDS: gpcc_loglik_from_gram indexes it dynamically)
        for (int e = 0; e < nrhs * nrhs; ++e) G[e] = 0.0;
        int bad = c.info[slot];   // (argument errors, set by the assembly before the launch)
        for (int kk = 0; kk < c.nt; ++kk) {
            const double *s2 = a.stepval + ((long)m * c.nt + kk) * GPCC_CHAIN_STEPVALS;
            ld = gpcc_ld_sc1(s2) + ld;
            const int b = (int)gpcc_ld_sc1(s2 + 1);
            if (bad == 0 && b != 0) bad = kk * GPCC_TILE + b;
            for (int e = 0; e < nrhs * nrhs; ++e) G[e] = gpcc_ld_sc1(s2 + 2 + e) + G[e];
        }
        c.logdet[slot] = ld;
        c.info[slot] = bad;
        int bad2 = bad;
        const double llv = gpcc_loglik_from_gram(c, G, nrhs, ld, bad2);
        g.out_loglik[g.first + m] = bad2 ? __builtin_nan("") : llv;
        g.out_info[g.first + m] = bad2;
    }
}

// ------------------------------------------------------------------------------------------
// The image of tile (k,k) for the chain: lower triangle of T(k,k) (every update of columns < k-1 applied by the workers) minus
// L(k,k-1) L(k,k-1)^T, the column blocks of L(k,k-1) folded in AS THE FOUR QUARTER SOLVES PUBLISH THEM (colflag).  36 blocks in the
// registers of eight waves, dealt 5/5/5/5/4/4/4/4 (gpcc_syrk_lower_wave's split); a column block is 16 KiB, copied by LDS-DMA (sc1)
// into one of two LDS stages.  Returns false if abandoned.
// ------------------------------------------------------------------------------------------
template <int RA, int CA, int NA, int RB, int CB, int NB>
GPCC_CHAIN_FN bool gpcc_chain_syrk_wave(const double *Tt, const double *Lt, const double *xprev, const unsigned *colflag, const unsigned *xrowp,
                                        unsigned *abortw, double *smem, int *ctl, int tid, int lane, unsigned long long *tr)
{
    typedef GpccPrec<double> P;
    const int lr = lane & 15, q = lane >> 4, sw = gpcc_sw(lr);
    d4 acc[NA + NB];
#pragma unroll
    for (int i = 0; i < NA + NB; ++i) {
        const int R = (i < NA) ? RA : RB, C = (i < NA) ? CA + i : CB + (i - NA);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][r] = -gpcc_ld_sc1(Tt + gpcc_elem_off<double>(16 * R + P::crow(q, r), 16 * C + lr));
    }
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned smem_addr = gpcc_lds_addr(smem);
    const double *p0 = smem + lr * 16 + (((2 * q) ^ sw) * 2);
    const double *p1 = smem + lr * 16 + (((2 * q + 1) ^ sw) * 2);
    d2 a70 = {0.0, 0.0}, a71 = {0.0, 0.0};
    int issued = -1;       // last column block whose copy this wave has issued
    unsigned peek = (lane == 0) ? gpcc_flag_ld(&colflag[1]) : 0u;    // colflag of the NEXT block as read a block earlier (lane 0): a workgroup that is behind -- its tile's other
                           // update came late, all of L(k,k-1) is there already -- copies block ch + 1 beside the products of block ch
    for (int ch = 0; ch < 8; ++ch) {
        // every wave polls for itself (one lane) and copies ITS 2 KiB of the column block into stage ch & 1 by LDS-DMA the moment the
        // fourth quarter has signalled it: one barrier per block (the stage written now was last read two blocks ago, and every wave
        // has passed the barrier in between)
        int okw = 1;
        if (ch == 7) {   // inv(D_7) of the step before: flagged well before S7 can be -- requested first, in flight while S7 is waited for and copied
            if (lane == 0) okw = gpcc_wait_ge(xrowp, 8u, abortw, 0x108u) ? 1 : 0;
            okw = __builtin_amdgcn_readfirstlane(okw);
            const double *xd = xprev + gpcc_bi(7, 7) + lr * 16 + 4 * q;
            a70 = d2{gpcc_ld_sc1(xd), gpcc_ld_sc1(xd + 1)};
            a71 = d2{gpcc_ld_sc1(xd + 2), gpcc_ld_sc1(xd + 3)};
        }
        if (issued < ch) {
            if (lane == 0 && okw) okw = gpcc_wait_ge(&colflag[ch], 4u, abortw, 0x100u + ch) ? 1 : 0;
            okw = __builtin_amdgcn_readfirstlane(okw);
            // (the last column block arrives as S7 -- before its product with inv(D_7)^T, gpcc_chain_trsmq -- in the step's published area)
            const double *src = (ch == 7) ? xprev + GPCC_XIMG_ELEMS : Lt + ch * 2048;
            if (okw) gpcc_dma_piece2_sc1(gpcc_uniform_ptr(src + wave * 256), (unsigned)lane * 16u, smem_addr + (unsigned)(((ch & 1) * 2048) * 8 + wave * 2048));
            issued = ch;
        }
        if (tr && ch == 7 && tid == 0) tr[7] = wall_clock64();   // header word 7: the last column block of L(k,k-1) seen
        const int so = (ch & 1) * 2048;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (ch == 7) {
            // L_7 = S7 inv(D_7)^T: every wave finishes the 16 rows it has copied itself (no barrier in between) the moment inv(D_7) of the
            // step before is there (requested above) -- V = inv(D_7) S7^T in the result layout [column][row], written back in the chunk's own layout
            const d2 a0 = a70, a1 = a71;
            const d2 b0 = *(const d2 *)(p0 + so + wave * 256), b1 = *(const d2 *)(p1 + so + wave * 256);
            d4 V = {0.0, 0.0, 0.0, 0.0};
            V = P::mfma(a0[0], b0[0], V);
            V = P::mfma(a0[1], b0[1], V);
            V = P::mfma(a1[0], b1[0], V);
            V = P::mfma(a1[1], b1[1], V);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cc = P::crow(q, r);
                smem[so + (wave * 16 + lr) * 16 + (((cc >> 1) ^ sw) * 2) + (cc & 1)] = V[r];
            }
        }
        if (!__syncthreads_and(okw)) return false;
        if (ch < 7) {
            const int nxt = __builtin_amdgcn_readfirstlane((int)peek);
            if (nxt >= 4) {   // (every reader of stage (ch + 1) & 1 -- block ch - 1 -- is behind the barrier above)
                const double *src = (ch + 1 == 7) ? xprev + GPCC_XIMG_ELEMS : Lt + (ch + 1) * 2048;
                gpcc_dma_piece2_sc1(gpcc_uniform_ptr(src + wave * 256), (unsigned)lane * 16u, smem_addr + (unsigned)((((ch + 1) & 1) * 2048) * 8 + wave * 2048));
                issued = ch + 1;
            }
            if (ch < 6 && lane == 0) peek = gpcc_flag_ld(&colflag[ch + 2]);
        }
        d2 aA[2], aB[2];
        aA[0] = *(const d2 *)(p0 + so + RA * 16 * 16);
        aA[1] = *(const d2 *)(p1 + so + RA * 16 * 16);
        if (NB > 0) {
            aB[0] = *(const d2 *)(p0 + so + RB * 16 * 16);
            aB[1] = *(const d2 *)(p1 + so + RB * 16 * 16);
        }
#pragma unroll
        for (int i = 0; i < NA + NB; ++i) {
            const int C = (i < NA) ? CA + i : CB + (i - NA);
            d2 b[2];
            b[0] = *(const d2 *)(p0 + so + C * 16 * 16);
            b[1] = *(const d2 *)(p1 + so + C * 16 * 16);
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[i] = P::mfma((i < NA) ? aA[s / 2][s % 2] : aB[s / 2][s % 2], b[s / 2][s % 2], acc[i]);
        }
    }
    __syncthreads();   // the stage is dead: the image takes its place
#pragma unroll
    for (int i = 0; i < NA + NB; ++i) {
        const int R = (i < NA) ? RA : RB, C = (i < NA) ? CA + i : CB + (i - NA);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            smem[gpcc_bi(R, C) + gpcc_be(P::crow(q, r), lr)] = -acc[i][r];
            if (R == 0 && C == 0) smem[GPCC_CHAIN_TMP_OFF + P::crow(q, r) * GPCC_CHAIN_TLD + lr] = -acc[i][r];   // the first block to be factored, row-major
        }
    }
    return true;
}

// one chain workgroup: role r of evaluation m runs the diagonal steps k = r, r + 2, ...
__device__ __forceinline__ void gpcc_chain_role(const GpccCtx &c, const GpccGroup &g, const GpccChainArgs &a, const int m, const int role, double *smem, int *ctl)
{
    const int tid0 = threadIdx.x;
    const int slot = g.slot0 + m;
    const GpccChainFlags fl = gpcc_chain_flags(a, c.nt, m);
    double *tiles = (double *)c.tiles + (long)slot * c.slot_stride;
    if (role == ((c.nt - 1) & 1) && tid0 == 0) {   // what the caller sees if the launch is abandoned (the last step overwrites it)
        g.out_loglik[g.first + m] = __builtin_nan("");
        g.out_info[g.first + m] = GPCC_INFO_TIMEOUT;
    }
    for (int k = role; k < c.nt; k += 2) {
        int tid = tid0;
        asm volatile("" : "+v"(tid));   // per-step opaque copy: keeps the compiler from hoisting every lane-dependent address of a step out of this loop (spills)
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        unsigned long long *tr = a.trace ? a.trace + ((long)m * c.nt + k) * GPCC_CHAIN_TRACE_WORDS : nullptr;
        if (tr && tid == 0) tr[0] = wall_clock64();
        const double *Tt = tiles + gpcc_tile_off(k, k);
        if (k == 0) {   // tile (0,0) as assembled (before the launch: plain loads): its lower 36 blocks -> image
            for (int p0 = tid; p0 < GPCC_TILE_ELEMS / 2; p0 += GPCC_CHAIN_THREADS) {
                const int e = p0 * 2;
                const int ch = e / (GPCC_TILE * 16), rem = e % (GPCC_TILE * 16), r = rem / 16, ks = rem % 16;
                const int col0 = ch * 16 + ((ks / 2) ^ gpcc_sw(r)) * 2;
                if ((col0 >> 4) <= (r >> 4)) {
                    const d2 v = *(const d2 *)(Tt + e);
                    smem[gpcc_bi(r >> 4, col0 >> 4) + gpcc_be(r & 15, col0 & 15)] = v[0];
                    smem[gpcc_bi(r >> 4, col0 >> 4) + gpcc_be(r & 15, (col0 & 15) + 1)] = v[1];
                    if (r < 16 && col0 < 16) *(d2 *)(smem + GPCC_CHAIN_TMP_OFF + r * GPCC_CHAIN_TLD + col0) = v;
                }
            }
        } else {
            if (tid == 0) ctl[0] = gpcc_wait_ge(&fl.ver[gpcc_tile_idx(k, k)], 4u * (unsigned)(k - 1), fl.abortw, 0x200u) ? 1 : 0;
            __syncthreads();
            if (!ctl[0]) return;
            const double *Lt = tiles + gpcc_tile_off(k, k - 1);
            const unsigned *cf = fl.colflag + 8 * (k - 1);
            const double *xprev = a.ximg + ((long)m * c.nt + k - 1) * GPCC_XIMG_STRIDE;   // what step k - 1 published
            bool ok;
            switch (wave) {
            case 0: ok = gpcc_chain_syrk_wave<7, 0, 5, 0, 0, 0>(Tt, Lt, xprev, cf, &fl.xrow[k - 1], fl.abortw, smem, ctl, tid, lane, tr); break;
            case 1: ok = gpcc_chain_syrk_wave<6, 0, 5, 0, 0, 0>(Tt, Lt, xprev, cf, &fl.xrow[k - 1], fl.abortw, smem, ctl, tid, lane, tr); break;
            case 2: ok = gpcc_chain_syrk_wave<5, 0, 5, 0, 0, 0>(Tt, Lt, xprev, cf, &fl.xrow[k - 1], fl.abortw, smem, ctl, tid, lane, tr); break;
            case 3: ok = gpcc_chain_syrk_wave<4, 0, 5, 0, 0, 0>(Tt, Lt, xprev, cf, &fl.xrow[k - 1], fl.abortw, smem, ctl, tid, lane, tr); break;
            case 4: ok = gpcc_chain_syrk_wave<7, 5, 3, 0, 0, 1>(Tt, Lt, xprev, cf, &fl.xrow[k - 1], fl.abortw, smem, ctl, tid, lane, tr); break;
            case 5: ok = gpcc_chain_syrk_wave<6, 5, 2, 1, 0, 2>(Tt, Lt, xprev, cf, &fl.xrow[k - 1], fl.abortw, smem, ctl, tid, lane, tr); break;
            case 6: ok = gpcc_chain_syrk_wave<5, 5, 1, 2, 0, 3>(Tt, Lt, xprev, cf, &fl.xrow[k - 1], fl.abortw, smem, ctl, tid, lane, tr); break;
            default: ok = gpcc_chain_syrk_wave<3, 0, 4, 0, 0, 0>(Tt, Lt, xprev, cf, &fl.xrow[k - 1], fl.abortw, smem, ctl, tid, lane, tr); break;
            }
            if (!ok) return;
        }
        if (tr && tid == 0) tr[1] = wall_clock64();
        gpcc_chain_diag(c, g, a, fl, k, m, smem, tid);   // (begins with a barrier behind its own loads: the image is complete for every wave)
        if (tr && tid == 0) tr[2] = wall_clock64();
    }
}

// ------------------------------------------------------------------------------------------
// Worker job TRSMQ(I, k, qr): rows 32 qr .. 32 qr + 31 of L(I,k) = T(I,k) inv(L_kk)^T, column block f as soon as row block f of
// inv(L_kk) is published (xrow[k] >= f + 1), fused with the forward substitution z_I -= L(I,k) w_k at the end (xrow[k] = 9).
// Wave (rh, kq): rows 16 rh .. of the quarter, K chunks kq and kq + 4 (the A operand -- the wave's rows of T(I,k) -- in registers);
// the four K partials of a block are summed through LDS in a fixed order by 256 threads that own (row, two columns) and write the
// result in the tile's own byte layout with 16-byte sc1 stores.  For the tile below the diagonal (I = k + 1) every column block is
// signalled separately (colflag[k][f]): the chain workgroup that builds tile (k+1,k+1) consumes them as they come.
// ------------------------------------------------------------------------------------------
GPCC_CHAIN_FN bool gpcc_chain_trsmq(const GpccCtx &c, const GpccChainArgs &a, const GpccChainFlags &fl, const int m, const int slot, const int k,
                                    const int I, const int qr, double *smem, int *ctl, const int tid, unsigned long long *wt)
{
    typedef GpccPrec<double> P;
    constexpr int PLD = 17, PW = 16 * PLD, LLD = GPCC_CHAIN_TLD, LB = 16 * LLD;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, q = lane >> 4;
    const int rh = wave & 1, kg = (wave >> 1) & 1;
    const bool worker = wave < 4;           // waves 0-3 multiply, waves 4-7 sum, store and signal
    const bool chain_tile = (I == k + 1);
    double *tiles = (double *)c.tiles + (long)slot * c.slot_stride;
    double *Tt = tiles + gpcc_tile_off(I, k);
    const __amdgpu_buffer_rsrc_t tres = gpcc_rsrc(Tt, GPCC_TILE_ELEMS * 8);
    double *xk = a.ximg + ((long)m * c.nt + k) * GPCC_XIMG_STRIDE;
    const __amdgpu_buffer_rsrc_t xres = gpcc_rsrc(xk, GPCC_XIMG_ELEMS * 8);
    const __amdgpu_buffer_rsrc_t s7res = gpcc_rsrc(xk + GPCC_XIMG_ELEMS, 16 * GPCC_TILE * 8);
    if (tid == 0) ctl[0] = gpcc_wait_ge(&fl.ver[gpcc_tile_idx(I, k)], 4u * (unsigned)k, fl.abortw, 0x300u) ? 1 : 0;
    __syncthreads();
    if (!ctl[0]) return false;
    if (wt && tid == 0) wt[2] = wall_clock64();
    // summing threads (waves 4-7): (rh2, row, sp) = 16-byte slot sp (columns 2 sp, 2 sp + 1) of row 16 rh2 + row of the quarter
    const int rt = tid & 255, rh2 = rt >> 7, rrow = (rt >> 3) & 15, sp = rt & 7;
    double lv[8][2];
    int *cdone = ctl + 8;       // [8]: summing waves that have drained their stores of column block f (LDS counters; the last one signals)
    if (tid < 8) cdone[tid] = 0;
    int seen = 0;               // row blocks of L_kk known to be published (uniform)
    double *part = smem;                    // [2][4 waves][16 x PLD]: a wave's contribution to -L_f^T, [column][row]
    double *lbuf = smem + 2 * 4 * PW;       // [8][2][16 x LLD]: the quarter's finished column blocks, [row][column]: the B operand of the later ones (slot 7: S7)
    // this wave's elements of T(I,k): row 32 qr + 16 rh + lr, columns 16 f + q + 4 r (only K group 0 starts from T)
    auto load_t = [&](int f, double (&t)[4]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] = gpcc_ld_sc1(Tt + gpcc_elem_off<double>(32 * qr + 16 * rh + lr, 16 * f + q + 4 * r));
    };
    // row block f of L_kk as the A operand: blocks (f, j) [column c = lr][k = 4 q ..] for j = kg, kg + 2, .. below f; inv(D_f) [c' = lr][c = q + 4 r]
    auto load_a = [&](int f, d2 (&av)[4][2], double (&dv)[4], bool with_d) {
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            const int j = kg + 2 * c4;
            if (j < f) {   // (wave-uniform)
                const unsigned off = (unsigned)((gpcc_bi(f, j) + lr * 16 + 4 * q) * 8);
                av[c4][0] = gpcc_ld16_sc1(xres, off);
                av[c4][1] = gpcc_ld16_sc1(xres, off + 16u);
            }
        }
        if (with_d) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dv[r] = gpcc_ld_sc1(xk + gpcc_bi(f, f) + lr * 16 + q + 4 * r);
        }
    };
    // column block f of L(I,k), published once its four storing waves have drained: each adds to an LDS counter behind its own wait, the
    // last one adds to the global counter the chain polls (no workgroup barrier in between)
    auto signal_block = [&](int f) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int last = 0;
        if (lane == 0) last = (__hip_atomic_fetch_add(&cdone[f], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 3) ? 1 : 0;
        if (last) gpcc_flag_add(&fl.colflag[8 * k + f], 1u);
    };
    unsigned long long *ht = (a.trace && chain_tile) ? a.trace + ((long)m * c.nt + k) * GPCC_CHAIN_TRACE_WORDS : nullptr;   // header words 4-6 of step k: the LAST of the four quarters of tile (k+1,k) (atomic max)
    d2 an[4][2];                // the next row block's operands, requested ahead when that row is known to be out already
    double dn[4], tn[4] = {0.0, 0.0, 0.0, 0.0};
    bool have_next = false;
    if (worker && kg == 0) load_t(0, tn);
#pragma unroll
    for (int f = 0; f < 8; ++f) {   // (unrolled: lv stays in registers)
        // ---- top of the block: row block f of L_kk and inv(D_f) are out (xrow >= f + 1); the barrier also makes the previous column
        // block visible in lbuf
        const bool s7path = (f == 7) && chain_tile;   // the chain tile's last column block leaves as S7, before inv(D_7) is known (below)
        if (tid == GPCC_CHAIN_THREADS - 64) {   // one lane polls
            if (s7path) ctl[0] = gpcc_wait_ge(&fl.l7[k], 1u, fl.abortw, 0x318u) ? 1 : 0;
            else ctl[0] = (seen >= f + 1) ? 1 : (gpcc_wait_ge(&fl.xrow[k], (unsigned)(f + 1), fl.abortw, 0x310u + f) ? 1 : 0);
            ctl[3] = (int)gpcc_flag_ld(&fl.xrow[k]);
        }
        __syncthreads();
        if (!ctl[0]) return false;
        seen = ctl[3] > seen ? ctl[3] : seen;
        if (ht && f == 7 && tid == 0) atomicMax(&ht[4], (unsigned long long)wall_clock64());
        const bool two = f >= 2;   // K group 1 (column blocks 1, 3, 5) has a share from row block 2 on
        if (worker) {
            d2 av[4][2];
            double dv[4], tv[4];
            const bool mine = kg == 0 || two;
            if (mine) {
                if (have_next) {
#pragma unroll
                    for (int c4 = 0; c4 < 4; ++c4) { av[c4][0] = an[c4][0]; av[c4][1] = an[c4][1]; }
#pragma unroll
                    for (int r = 0; r < 4; ++r) dv[r] = dn[r];
                } else {
                    load_a(f, av, dv, !s7path);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) tv[r] = tn[r];
            have_next = (f < 7) && (seen >= f + 2) && !(f == 6 && chain_tile);
            if (have_next && (kg == 0 || f + 1 >= 2)) load_a(f + 1, an, dn, true);
            if (f < 7 && kg == 0) load_t(f + 1, tn);
            if (mine) {
                // P = sum_j Lkk[f][j] L_j^T - T_f^T  ([column][row]: rows of the quarter along the lanes), then V = inv(D_f) P = -L_f^T:
                // the first product's result registers ARE the second one's B operand
                d4 acc, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = (kg == 0) ? -tv[r] : 0.0;
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    const int j = kg + 2 * c4;
                    if (j < f) {
                        const double *lb = lbuf + (j * 2 + rh) * LB + lr * LLD + 4 * q;
                        const d2 b0 = *(const d2 *)lb, b1 = *(const d2 *)(lb + 2);
                        if (c4 & 1) {
                            acc1 = P::mfma(av[c4][0][0], b0[0], acc1);
                            acc1 = P::mfma(av[c4][0][1], b0[1], acc1);
                            acc1 = P::mfma(av[c4][1][0], b1[0], acc1);
                            acc1 = P::mfma(av[c4][1][1], b1[1], acc1);
                        } else {
                            acc = P::mfma(av[c4][0][0], b0[0], acc);
                            acc = P::mfma(av[c4][0][1], b0[1], acc);
                            acc = P::mfma(av[c4][1][0], b1[0], acc);
                            acc = P::mfma(av[c4][1][1], b1[1], acc);
                        }
                    }
                }
                if (f > 2) acc += acc1;
                d4 V = {0.0, 0.0, 0.0, 0.0}, V1 = {0.0, 0.0, 0.0, 0.0};
                if (s7path) {
                    V = acc;   // (P itself: the product with inv(D_7) is the consumer's)
                } else {
                    V = P::mfma(dv[0], acc[0], V);
                    V1 = P::mfma(dv[1], acc[1], V1);
                    V = P::mfma(dv[2], acc[2], V);
                    V1 = P::mfma(dv[3], acc[3], V1);
                    V += V1;
                }
                double *pw = part + ((f & 1) * 4 + wave) * PW;
#pragma unroll
                for (int r = 0; r < 4; ++r) pw[P::crow(q, r) * PLD + lr] = V[r];
            }
        } else if (f > 0 && chain_tile) {
            signal_block(f - 1);   // (the stores were issued before the barrier above; nothing else for these waves to do until the next one)
            if (ht && f == 7 && tid == 256) atomicMax(&ht[6], (unsigned long long)wall_clock64());
        }
        __syncthreads();
        if (!worker) {
            const double *pp = part + ((f & 1) * 4 + rh2) * PW + (2 * sp) * PLD + rrow;
            d2 sm = {pp[0], pp[PLD]};                     // K group 0 (wave = 2 kg + rh)
            if (two) { sm[0] += pp[2 * PW]; sm[1] += pp[2 * PW + PLD]; }
            sm = -sm;
            const int row = 32 * qr + 16 * rh2 + rrow;
            if (s7path) {
                // S7 = T_7 - sum_{j<7} L_j Lkk[7][j]^T goes out in the place of the column block (colflag[7] = "S7 is out"): the chain workgroup
                // that folds it into tile (k+1,k+1) multiplies it with inv(D_7)^T itself the moment that block is published -- one hand-off
                // (inv(D_7) -> fold) instead of two (inv(D_7) -> this workgroup -> store, drain, flag -> fold): 3.3 us per diagonal step
                gpcc_st16_sc1(s7res, (unsigned)((row * 16 + ((sp ^ gpcc_sw(row)) * 2)) * 8), sm);
                *(d2 *)(lbuf + (7 * 2 + rh2) * LB + rrow * LLD + 2 * sp) = sm;
                signal_block(7);
            } else {
                lv[f][0] = sm[0];
                lv[f][1] = sm[1];
                if (f < 7) *(d2 *)(lbuf + (f * 2 + rh2) * LB + rrow * LLD + 2 * sp) = sm;
                gpcc_st16_sc1(tres, (unsigned)((f * 2048 + row * 16 + ((sp ^ gpcc_sw(row)) * 2)) * 8), sm);
            }
        }
        if (ht && f == 7 && tid == 256) atomicMax(&ht[5], (unsigned long long)wall_clock64());
        if (s7path) {
            // ... and this workgroup's own copy of the block, for the tile: L_7 = S7 inv(D_7)^T, off the chain's path
            if (tid == GPCC_CHAIN_THREADS - 64) ctl[0] = gpcc_wait_ge(&fl.xrow[k], 8u, fl.abortw, 0x319u) ? 1 : 0;
            __syncthreads();   // (S7 is complete in lbuf, too)
            if (!ctl[0]) return false;
            if (wave < 2) {    // one wave per row half: V = inv(D_7) S7^T = L_7^T
                const unsigned off = (unsigned)((gpcc_bi(7, 7) + lr * 16 + 4 * q) * 8);
                const d2 a0 = gpcc_ld16_sc1(xres, off), a1 = gpcc_ld16_sc1(xres, off + 16u);
                const double *lb = lbuf + (7 * 2 + wave) * LB + lr * LLD + 4 * q;
                const d2 b0 = *(const d2 *)lb, b1 = *(const d2 *)(lb + 2);
                d4 V = {0.0, 0.0, 0.0, 0.0};
                V = P::mfma(a0[0], b0[0], V);
                V = P::mfma(a0[1], b0[1], V);
                V = P::mfma(a1[0], b1[0], V);
                V = P::mfma(a1[1], b1[1], V);
                double *pw = part + wave * PW;   // (buffer 0: last read for column block 6, two barriers ago)
#pragma unroll
                for (int r = 0; r < 4; ++r) pw[P::crow(q, r) * PLD + lr] = V[r];
            }
            __syncthreads();
            if (!worker) {
                const double *pp = part + rh2 * PW + (2 * sp) * PLD + rrow;
                const d2 sm = {pp[0], pp[PLD]};
                lv[7][0] = sm[0];
                lv[7][1] = sm[1];
                const int row = 32 * qr + 16 * rh2 + rrow;
                gpcc_st16_sc1(tres, (unsigned)((7 * 2048 + row * 16 + ((sp ^ gpcc_sw(row)) * 2)) * 8), sm);
            }
        }
    }
    // w_k (xrow = 9) is what the forward substitution below still needs
    if (tid == GPCC_CHAIN_THREADS - 64) ctl[0] = gpcc_wait_ge(&fl.xrow[k], 9u, fl.abortw, 0x320u) ? 1 : 0;
    __syncthreads();
    if (!ctl[0]) return false;
    // forward substitution of logpdf's whitening: z_I[row] -= sum_c L(I,k)[row][c] w_k[c] -- AFTER the last column block is signalled
    // (the chain folds it into tile (k+1,k+1) meanwhile; z_{k+1} is wanted a block step later: gpcc_chain_diag)
    if (tid >= 256) {
        const double *wp = c.w + (long)slot * c.nrhs * c.Np + k * GPCC_TILE;
        double pr = 0.0;
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            pr = fma(lv[f][0], gpcc_ld_sc1(wp + 16 * f + 2 * sp), pr);
            pr = fma(lv[f][1], gpcc_ld_sc1(wp + 16 * f + 2 * sp + 1), pr);
        }
        pr += __shfl_xor(pr, 1);
        pr += __shfl_xor(pr, 2);
        pr += __shfl_xor(pr, 4);
        if (sp == 0) {
            double *zp = c.z + (long)slot * c.nrhs * c.Np + I * GPCC_TILE + 32 * qr + 16 * rh2 + rrow;
            gpcc_st_sc1(zp, gpcc_ld_sc1(zp) - pr);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) gpcc_flag_add(&fl.lcnt[gpcc_tile_idx(I, k)], 1u);
    return true;
}

// ------------------------------------------------------------------------------------------
// Worker job UPD(I, J, k): the right-looking trailing update T(I,J) -= L(I,k) L(J,k)^T of one tile (8 waves x (32 x 64), operands by
// LDS-DMA -- sc1 -- into a 4-deep ring, the result out of the registers as 16-byte sc1 stores, whole rows of a chunk per instruction), once both
// column tiles are complete (lcnt = 4 quarters) and the tile has received
// column k - 1 (ver = k).
// ------------------------------------------------------------------------------------------
// ncol > 1 (job kind 4, gpcc_chain_queue.h): the columns k .. k + ncol - 1 in one job -- the tiles of a tile row are contiguous, so
// L(I,k) | L(I,k+1) | ... and L(J,k) | L(J,k+1) | ... are simply operands of 8 ncol chunks instead of 8; the same sums in the same order as
// ncol jobs in a row (what a job stores and the next one loads are the same doubles), one fixed cost (polls, first copies, the tile in and
// out: ~6 us) instead of ncol.
GPCC_CHAIN_FN bool gpcc_chain_upd(const GpccCtx &c, const GpccChainFlags &fl, const int slot, const int k, const int I, const int J, const int ncol,
                                  double *smem, int *ctl, const int tid, unsigned long long *wt)
{
    typedef GpccPrec<double> P;
    constexpr int CH = 2048;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1, lr = lane & 15, q = lane >> 4, sw = gpcc_sw(lr);
    if (tid < 64) {   // the three inputs are polled by three lanes side by side (one after the other: three round trips of 0.6 us when all are there already)
        bool ok = true;
        if (tid < 2 * ncol + 1) {   // lane 0: the tile has received column k - 1; lanes 1 .. 2 ncol: the column tiles of rows I and J are solved
            const int cc = k + ((tid - 1) >> 1);
            const unsigned *p = (tid == 0) ? &fl.ver[gpcc_tile_idx(I, J)] : &fl.lcnt[gpcc_tile_idx((tid & 1) ? I : J, cc)];
            ok = gpcc_wait_ge(p, (tid == 0) ? 4u * (unsigned)k : 4u, fl.abortw, 0x400u + (unsigned)(tid > 2 ? 2 : tid));
        }
        const bool all = __all(ok);
        if (tid == 0) {
            ctl[0] = all ? 1 : 0;
            if (wt) wt[2] = wall_clock64();
        }
    }
    __syncthreads();
    if (!ctl[0]) return false;
    double *tiles = (double *)c.tiles + (long)slot * c.slot_stride;
    const double *gA = (const double *)gpcc_uniform_ptr(tiles + gpcc_tile_off(I, k)), *gB = (const double *)gpcc_uniform_ptr(tiles + gpcc_tile_off(J, k));
    double *Tt = tiles + gpcc_tile_off(I, J);
    const __amdgpu_buffer_rsrc_t tres = gpcc_rsrc(Tt, GPCC_TILE_ELEMS * 8);
    const unsigned smem_addr = gpcc_lds_addr(smem);
    // operands by LDS-DMA (sc1) into a ring of FOUR 32 KiB stages, three chunks ahead (one workgroup per CU: nobody else hides the
    // latency of a read that comes from another XCD's write-through).  The accumulators start at -T(I,J) like everywhere else in this
    // library: forming the product from zero and subtracting it once at the end (the tile prefetched as linear pieces) was measured --
    // no faster, and 20x less accurate on matrices with a large B term (1.3e-11 instead of 6e-13 at N = 4095): the running sum then
    // never shrinks towards the Schur complement
    gpcc_dma_chunk_sc1(gA, gB, smem_addr, wave, lane);
    gpcc_dma_chunk_sc1(gA + CH, gB + CH, smem_addr + 2 * GPCC_CHUNK_BYTES, wave, lane);
    gpcc_dma_chunk_sc1(gA + 2 * CH, gB + 2 * CH, smem_addr + 4 * GPCC_CHUNK_BYTES, wave, lane);
    d4 acc[2][4];
#pragma unroll
    for (int fm = 0; fm < 2; ++fm)
#pragma unroll
        for (int fn = 0; fn < 4; ++fn)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[fm][fn][r] = -gpcc_ld_sc1(Tt + gpcc_elem_off<double>(wr * 32 + fm * 16 + P::crow(q, r), wc * 64 + fn * 16 + lr));
    const double *pa0 = smem + (wr * 32 + lr) * 16 + (((2 * q) ^ sw) * 2);
    const double *pa1 = smem + (wr * 32 + lr) * 16 + (((2 * q + 1) ^ sw) * 2);
    const double *pb0 = smem + CH + (wc * 64 + lr) * 16 + (((2 * q) ^ sw) * 2);
    const double *pb1 = smem + CH + (wc * 64 + lr) * 16 + (((2 * q + 1) ^ sw) * 2);
    for (int cb = 0; cb < ncol; ++cb) {   // ncol tiles of K: the chunks of L(I,k) and L(J,k) and, contiguous behind them, of the columns that follow
      const bool lastcol = (cb == ncol - 1);
#pragma unroll
      for (int c8 = 0; c8 < 8; ++c8) {
        const int ch = cb * 8 + c8;
        // chunk ch has landed (4 DMA instructions per wave and chunk; chunk ch + 1 may still fly); behind the barrier every wave has
        // finished chunk ch - 1, whose stage takes chunk ch + 2
        if (c8 < 6 || !lastcol) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (c8 == 6) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (c8 < 5 || !lastcol) gpcc_dma_chunk_sc1(gA + (long)(ch + 3) * CH, gB + (long)(ch + 3) * CH, smem_addr + ((c8 + 3) % 4) * 2 * GPCC_CHUNK_BYTES, wave, lane);
        const int so = (c8 % 4) * 2 * CH;
        d2 a2[2][2];
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            a2[f][0] = *(const d2 *)(pa0 + so + f * 16 * 16);
            a2[f][1] = *(const d2 *)(pa1 + so + f * 16 * 16);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            d2 b[2][2];
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                b[f][0] = *(const d2 *)(pb0 + so + (2 * h + f) * 16 * 16);
                b[f][1] = *(const d2 *)(pb1 + so + (2 * h + f) * 16 * 16);
            }
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                for (int fm = 0; fm < 2; ++fm)
#pragma unroll
                    for (int f = 0; f < 2; ++f) acc[fm][2 * h + f] = P::mfma(a2[fm][s2 / 2][s2 % 2], b[f][s2 / 2][s2 % 2], acc[fm][2 * h + f]);
        }
      }
    }
    // out straight from the registers as 16-byte sc1 stores: a lane pair (columns lr, lr ^ 1) swaps half of its rows, after which the even
    // lane holds columns (lr, lr + 1) of rows q, q + 4 and the odd lane those of rows q + 8, q + 12 -- one store instruction then
    // writes eight whole 128-byte rows of a chunk (through LDS, in two halves with four barriers, this took 4.5 of the job's 22 us)
    {
        const bool odd = (lr & 1) != 0;
        const int cs = lr >> 1;
#pragma unroll
        for (int fm = 0; fm < 2; ++fm)
#pragma unroll
            for (int fn = 0; fn < 4; ++fn) {
                const int chn = 4 * wc + fn;
#pragma unroll
                for (int h = 0; h < 2; ++h) {   // rows r = h (even lane), r = h + 2 (odd lane)
                    const double mine = odd ? acc[fm][fn][h + 2] : acc[fm][fn][h];
                    const double give = odd ? acc[fm][fn][h] : acc[fm][fn][h + 2];
                    const double got = __shfl_xor(give, 1);
                    const int row = wr * 32 + fm * 16 + P::crow(q, odd ? h + 2 : h);
                    const d2 v = odd ? d2{-got, -mine} : d2{-mine, -got};
                    gpcc_st16_sc1(tres, (unsigned)((chn * CH + row * 16 + ((cs ^ gpcc_sw(row)) * 2)) * 8), v);
                }
            }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) gpcc_flag_st(&fl.ver[gpcc_tile_idx(I, J)], 4u * (unsigned)(k + ncol));
    return true;
}

// ------------------------------------------------------------------------------------------
// Worker job UPDQ(I, J, k, qr): rows 32 qr .. 32 qr + 31 of the update of tile (I,J) by column k, for the tiles the NEXT step needs at
// once (groups of one or two evaluations, GpccChainArgs::quarters): local column 0 -- J = k + 1, the tiles the next step solves -- and
// the diagonal tile (k+2,k+2), which a chain workgroup starts to build when step k is published.  A whole-tile update takes 22 us, and
// the solves of column k + 1 can only begin behind it: with one job per tile the launch could not step faster than update + solve (22 +
// 11 us) whatever the chain did.  As four jobs of a quarter of the work on four CUs the tile is there 7 us after column k is, and its
// solves run beside the diagonal step again.  (Four quarters cost 30 us of CU time instead of 22: with more evaluations in flight the
// whole-tile job is the better one.)  Wave w: columns 16 w .. 16 w + 15, two row fragments; per chunk of K the 64-row half of L(I,k)
// that holds the quarter (8 KiB) and all of L(J,k) (16 KiB) by LDS-DMA into a ring of three stages.  ver counts quarters: this job adds
// 1, a whole-tile update stores 4 (k + 1).
// ------------------------------------------------------------------------------------------
GPCC_CHAIN_FN bool gpcc_chain_updq(const GpccCtx &c, const GpccChainFlags &fl, const int slot, const int k, const int I, const int J, const int qr,
                                   double *smem, int *ctl, const int tid, unsigned long long *wt)
{
    typedef GpccPrec<double> P;
    constexpr int CH = 2048, ST = 3072;   // doubles per chunk of a tile; per ring stage (1024 of A, 2048 of B)
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, q = lane >> 4, sw = gpcc_sw(lr);
    if (tid < 64) {   // (three lanes poll the three inputs side by side: gpcc_chain_upd)
        bool ok = true;
        if (tid < 3) {
            const unsigned *p = (tid == 0) ? &fl.lcnt[gpcc_tile_idx(I, k)] : (tid == 1) ? &fl.lcnt[gpcc_tile_idx(J, k)] : &fl.ver[gpcc_tile_idx(I, J)];
            ok = gpcc_wait_ge(p, (tid == 2) ? 4u * (unsigned)k : 4u, fl.abortw, 0x500u + (unsigned)tid);
        }
        const bool all = __all(ok);
        if (tid == 0) {
            ctl[0] = all ? 1 : 0;
            if (wt) wt[2] = wall_clock64();
        }
    }
    __syncthreads();
    if (!ctl[0]) return false;
    double *tiles = (double *)c.tiles + (long)slot * c.slot_stride;
    const double *gA = (const double *)gpcc_uniform_ptr(tiles + gpcc_tile_off(I, k) + (qr >> 1) * 1024), *gB = (const double *)gpcc_uniform_ptr(tiles + gpcc_tile_off(J, k));
    double *Tt = tiles + gpcc_tile_off(I, J);
    const __amdgpu_buffer_rsrc_t tres = gpcc_rsrc(Tt, GPCC_TILE_ELEMS * 8);
    const unsigned smem_addr = gpcc_lds_addr(smem);
    const unsigned voff = (unsigned)lane * 16u;
    auto dma = [&](int ch) {   // three 1 KiB pieces per wave
        const unsigned st = smem_addr + (unsigned)((ch % 3) * ST * 8);
        gpcc_dma_piece1_sc1(gpcc_uniform_ptr(gA + (long)ch * CH + wave * 128), voff, st + wave * 1024);
        gpcc_dma_piece2_sc1(gpcc_uniform_ptr(gB + (long)ch * CH + wave * 256), voff, st + 8192 + wave * 2048);
    };
    dma(0);
    dma(1);
    d4 acc[2];
#pragma unroll
    for (int fm = 0; fm < 2; ++fm)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[fm][r] = -gpcc_ld_sc1(Tt + gpcc_elem_off<double>(32 * qr + fm * 16 + P::crow(q, r), 16 * wave + lr));
    const double *pa0 = smem + ((qr & 1) * 32 + lr) * 16 + (((2 * q) ^ sw) * 2);
    const double *pa1 = smem + ((qr & 1) * 32 + lr) * 16 + (((2 * q + 1) ^ sw) * 2);
    const double *pb0 = smem + 1024 + (16 * wave + lr) * 16 + (((2 * q) ^ sw) * 2);
    const double *pb1 = smem + 1024 + (16 * wave + lr) * 16 + (((2 * q + 1) ^ sw) * 2);
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) {
        // (the 8 loads of T above are older than every DMA piece still wanted: vmcnt counts them out first)
        if (ch < 7) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (ch + 2 < 8) dma(ch + 2);
        const int so = (ch % 3) * ST;
        const d2 b0 = *(const d2 *)(pb0 + so), b1 = *(const d2 *)(pb1 + so);
#pragma unroll
        for (int fm = 0; fm < 2; ++fm) {
            const d2 a0 = *(const d2 *)(pa0 + so + fm * 256), a1 = *(const d2 *)(pa1 + so + fm * 256);
            acc[fm] = P::mfma(a0[0], b0[0], acc[fm]);
            acc[fm] = P::mfma(a0[1], b0[1], acc[fm]);
            acc[fm] = P::mfma(a1[0], b1[0], acc[fm]);
            acc[fm] = P::mfma(a1[1], b1[1], acc[fm]);
        }
    }
    // out straight from the registers (gpcc_chain_upd's lane-pair swap): chunk = this wave's 16 columns, whole 128-byte rows per instruction
    {
        const bool odd = (lr & 1) != 0;
        const int cs = lr >> 1;
#pragma unroll
        for (int fm = 0; fm < 2; ++fm)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const double mine = odd ? acc[fm][h + 2] : acc[fm][h];
                const double give = odd ? acc[fm][h] : acc[fm][h + 2];
                const double got = __shfl_xor(give, 1);
                const int row = 32 * qr + fm * 16 + P::crow(q, odd ? h + 2 : h);
                const d2 v = odd ? d2{-got, -mine} : d2{-mine, -got};
                gpcc_st16_sc1(tres, (unsigned)((wave * CH + row * 16 + ((cs ^ gpcc_sw(row)) * 2)) * 8), v);
            }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) gpcc_flag_add(&fl.ver[gpcc_tile_idx(I, J)], 1u);
    return true;
}


// grid: the chain block range (16 per 8 evaluations: blocks b and b + 8 -- one XCD, as dispatched -- are the two roles of an
// evaluation) + workers; block 512; LDS GPCC_CHAIN_LDS_BYTES.
__global__ __launch_bounds__(GPCC_CHAIN_THREADS, 2) void gpcc_chain_kernel(GpccCtx c, GpccGroup g, GpccChainArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double *smem = (double *)smem_raw;
    int *ctl = (int *)(smem_raw + GPCC_CHAIN_LDS_BYTES - 128);   // 32 words: [0, 8) a job's control words, [8, 16) gpcc_chain_trsmq's counters, [16] a job's quarter
    const int b = blockIdx.x, tid0 = threadIdx.x;
    // the dedicated range: per 8 evaluations 16 blocks for the two chain roles (b, b + 8) and, with helpers, 32 more for the four
    // quarter solves of tile (k+1,k) -- all six workgroups of evaluation m on blocks = m (mod 8): one XCD as dispatched (a speed bonus
    // for their hand-offs, nothing depends on it)
    const int per8 = a.helpers ? 48 : 16;
    const int ncb = per8 * ((g.cnt + 7) / 8);
    if (b < ncb) {
        const int grp = b / per8, h = b % per8, m = grp * 8 + (h & 7);
        if (m < g.cnt) {
            if (h < 16) {
                gpcc_chain_role(c, g, a, m, (h >> 3) & 1, smem, ctl);
            } else {
                // chain helper: quarter q of L(k+1,k) for every step -- claimed before the diagonal step begins, so that it always runs
                // BESIDE it (as queue jobs these were claimed behind the previous step's bulk: in the first nt/4 steps of a single
                // evaluation only 2 us before the diagonal step ended, and the chain then waited 14 us for them)
                const int q = (h - 16) >> 3, slot = g.slot0 + m;
                const GpccChainFlags fl = gpcc_chain_flags(a, c.nt, m);
                for (int k = 0; k < c.nt - 1; ++k) {
                    int tid = tid0;
                    asm volatile("" : "+v"(tid));
                    if (!gpcc_chain_trsmq(c, a, fl, m, slot, k, k + 1, q, smem, ctl, tid, nullptr)) return;
                    __syncthreads();
                }
            }
            return;
        }
    }
    // ---- worker: jobs in ONE queue order (gpcc_chain_queue.h), claimed by a returning atomic add (wait-free); a claimed job waits for its
    // inputs.  Every job's inputs are produced by jobs EARLIER in this order or by the chain (the solves of step k need local column 0 of
    // step k-1; local column 0 of step k was local column 1 of step k-1 (NEAR1); FAR(k-1) needs step k-1's solves and FAR(k-2) /
    // NEAR1(k-2); NEAR1(k) was local column 2 of step k-1: FAR(k-1), just in front), so the oldest unfinished job can always run: no
    // deadlock whatever is resident (checked for every size by tests/abi/chain_queue_check.cpp).  Why the bulk is one step late: in plain
    // step order the urgent jobs of step k were claimed only after every bulk job of step k - 1 had been -- in the first nt/4 steps that is
    // two rounds of 23 us jobs on 254 CUs, and the chain of a single evaluation waited 16 us instead of 5 between its diagonal steps
    // (profiles/r05/chain_trace_first_version.log).
    int ks = 0;
    for (;;) {
        int tid = tid0;
        asm volatile("" : "+v"(tid));   // (per-job opaque copy, as in gpcc_chain_role)
        if (tid == 0) {
            int j = -1;
            while (ks < c.nt) {   // (the job order: gpcc_chain_queue.h)
                const int nj = g.cnt * gpcc_chain_list_len(c.nt, ks, a.helpers, a.quarters, a.batch);
                const int t = (nj > 0) ? (int)gpcc_flag_add(&a.words[16 + ks], 1u) : 0;
                if (t < nj) {
                    j = t;
                    break;
                }
                ++ks;
            }
            int kind = -1, jk = 0, jI = 0, jJ = 0, jq = 0;
            if (j >= 0) {
                const GpccChainJob jb = gpcc_chain_decode(c.nt, ks, j / g.cnt, a.helpers, a.quarters, a.batch);
                kind = jb.kind; jk = jb.k; jI = jb.I; jJ = jb.J; jq = jb.q;
            }
            if (kind == 1) jJ = jq;   // (a solve's quarter travels in the column slot)
            ctl[16] = jq;
            ctl[1] = kind; ctl[2] = jk; ctl[5] = (j >= 0) ? j % g.cnt : 0; ctl[6] = jI; ctl[7] = jJ; ctl[3] = ks;
        }
        __syncthreads();
        const int kind = ctl[1], k = ctl[2], m = ctl[5], jI = ctl[6], jJ = ctl[7], jq = ctl[16];
        ks = ctl[3];
        __syncthreads();   // (ctl is rewritten by the job's own waits)
        if (kind < 0) return;
        const int slot = g.slot0 + m;
        const GpccChainFlags fl = gpcc_chain_flags(a, c.nt, m);
        unsigned long long *wt = nullptr;
        if (a.wtrace) {
            if (tid == 0) ctl[4] = (int)gpcc_flag_add(&a.words[1], 1u);
            __syncthreads();
            if (ctl[4] < a.wtrace_cap) wt = a.wtrace + 4L * ctl[4];
            if (wt && tid == 0) {
                wt[0] = ((unsigned long long)kind << 56) | ((unsigned long long)k << 32) | (unsigned)(jI * 1024 + jJ);
                wt[1] = wall_clock64();
            }
        }
        bool ok;
        if (kind == 1) ok = gpcc_chain_trsmq(c, a, fl, m, slot, k, jI, jJ, smem, ctl, tid, wt);
        else if (kind == 3) ok = gpcc_chain_updq(c, fl, slot, k, jI, jJ, jq, smem, ctl, tid, wt);
        else ok = gpcc_chain_upd(c, fl, slot, k, jI, jJ, (kind == 4) ? jq : 1, smem, ctl, tid, wt);
        if (!ok) return;
        if (wt && tid == 0) wt[3] = wall_clock64();
        __syncthreads();
    }
}
