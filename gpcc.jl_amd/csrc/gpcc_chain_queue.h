// gpcc_chain_queue.h -- the job order of the persistent few-evaluation launch (gpcc_chain.hip.h) as plain integer code: which job a
// ticket of the list of "step" ks stands for.  Compiled into the kernel, and -- because the launch's freedom from deadlock rests on
// this order alone ("every job's inputs are produced by jobs EARLIER in the order, or by the chain") -- also into a host test that
// checks exactly that for every matrix size (tests/abi/chain_queue_check.cpp).
//
// The list of step k (n = nt - k - 1 tile rows below the diagonal tile; local tile coordinates a = I - k - 1 >= b = J - k - 1):
//     URGENT(k)    the quarter solves of column k (with helpers: without tile (k+1,k)'s); the updates by column k of what the NEXT step
//                  needs at once -- local column 0, (k+2.., k+1), the column it solves, then its diagonal tile (k+2,k+2): in quarters for
//                  small groups --; of the tiles next to the diagonal -- (k+3,k+2), (k+3,k+3), (k+4,k+3), ...;
//     FAR(k - 1)   the updates by column k - 1 of the other tiles from local column 2 on (the bulk: ~n^2/2 jobs), ONE STEP LATE;
//                  with wmax > 1, SEVERAL columns per job where a tile has the slack (kind 4: the tiles of a tile row are contiguous, so
//                  the operands of w columns are simply w times as long; one fixed cost of ~6 us per w x 13.7 us of MFMAs instead of per
//                  13.7) -- see gpcc_chain_far_hi below;
//     NEAR1(k)     the updates by column k of the rest of local column 1.
#pragma once
#if defined(__HIPCC__)
#define GPCC_HD __host__ __device__ __forceinline__
#else
#define GPCC_HD inline
#endif

struct GpccChainJob {
    int kind;   // -1: none; 1: quarter solve of tile (I,k), quarter q; 2: update of tile (I,J) by column k; 3: its row quarter q;
                // 4: update of tile (I,J) by the q columns k .. k + q - 1
    int k, I, J, q;
};

// The bulk ("FAR") jobs in the list of step ks: column blocks that END at column kf = ks - 1.  A tile's columns are cut into ALIGNED blocks
// of w = 1, 2, 4 ... wmax columns (k0 = 0 mod w): block (k0, w) "fits" tile column J if the tile is still at local column >= w + 1 when
// the block's last column has been solved (J - k0 - 1 >= 2 w: it has w steps of slack left when its job is queued), and a tile takes the
// MAXIMAL fitting blocks -- fitting is inherited by both halves of a fitting block, so the maximal ones partition a tile's columns; far
// from the diagonal that is wmax columns per job, towards the diagonal 4, 2, 1.  Jobs of one width in the list of step ks: local columns
// (of step kf) rb = w + 1 .. hi(w), rows ra = rb + 2 .. np - 1.  wmax = 1: every column its own job.
#define GPCC_CHAIN_MAX_BATCH 8
GPCC_HD int gpcc_chain_far_hi(int nt, int ks, int w, int wmax)   // last local column (of step ks - 1) of the jobs of width w; < w + 1: none
{
    const int kf = ks - 1, np = nt - kf - 1;
    if (ks < 1 || (kf + 1) % w != 0) return 0;
    const int k0 = kf + 1 - w;
    int hi = np - 3;
    if (2 * w <= wmax) {   // the block of twice the width takes the tiles it fits
        const int lim = (k0 % (2 * w) == 0) ? 3 * w : 2 * w;
        if (lim < hi) hi = lim;
    }
    return hi;
}
GPCC_HD int gpcc_chain_far_count(int nt, int ks, int w, int wmax)
{
    const int kf = ks - 1, np = nt - kf - 1, lo = w + 1, hi = gpcc_chain_far_hi(nt, ks, w, wmax);
    if (hi < lo) return 0;
    // sum_{rb = lo}^{hi} (np - 2 - rb)
    return (hi - lo + 1) * (np - 2) - (hi * (hi + 1) - (lo - 1) * lo) / 2;
}

// jobs per evaluation in the list of step ks
GPCC_HD int gpcc_chain_list_len(int nt, int ks, int helpers, int quarters, int wmax)
{
    const int n = nt - ks - 1;
    const int qd = quarters ? 4 : 1;
    const int nsol = n >= 1 ? 4 * (n - (helpers ? 1 : 0)) : 0;   // (with helpers the solves of tile (k+1,k) are not queue jobs)
    const int urgent = n >= 2 ? nsol + qd * n + (2 * n - 4) : nsol;
    int far = 0;
    for (int w = wmax; w >= 1; w >>= 1) far += gpcc_chain_far_count(nt, ks, w, wmax);
    const int near = n >= 4 ? n - 3 : 0;
    return urgent + far + near;
}

// job jj (0 <= jj < gpcc_chain_list_len) of the list of step ks
GPCC_HD GpccChainJob gpcc_chain_decode(int nt, int ks, int jj, int helpers, int quarters, int wmax)
{
    GpccChainJob jb;
    jb.kind = -1; jb.k = ks; jb.I = 0; jb.J = 0; jb.q = 0;
    const int qd = quarters ? 4 : 1;
    const int n = nt - ks - 1, np = n + 1;
    const int nsol = n >= 1 ? 4 * (n - (helpers ? 1 : 0)) : 0;
    const int nnext = n >= 2 ? qd * n : 0;
    const int urgent = n >= 2 ? nsol + nnext + (2 * n - 4) : nsol;
    int far = 0;
    for (int w = wmax; w >= 1; w >>= 1) far += gpcc_chain_far_count(nt, ks, w, wmax);
    int ra, rb;
    if (jj < nsol) {                  // quarter solve (I, ks, q)
        jb.kind = 1; jb.I = ks + 1 + (helpers ? 1 : 0) + jj / 4; jb.J = ks; jb.q = jj % 4;
    } else if (jj < nsol + nnext) {   // what the next step needs at once: local column 0, (k+2.., k+1), then the diagonal tile (k+2,k+2)
        const int u = (jj - nsol) / qd;
        jb.kind = quarters ? 3 : 2; jb.q = (jj - nsol) % qd;
        jb.I = (u < n - 1) ? ks + 2 + u : ks + 2;
        jb.J = (u < n - 1) ? ks + 1 : ks + 2;
    } else if (jj < urgent) {         // next to the diagonal: (k+3,k+2), (k+3,k+3), (k+4,k+3), ...
        const int bnd = jj - nsol - nnext + 2;
        ra = 1 + bnd / 2;
        rb = (bnd & 1) ? ra : ra - 1;
        jb.kind = 2; jb.I = ks + 1 + ra; jb.J = ks + 1 + rb;
    } else if (jj < urgent + far) {   // the bulk: column blocks that end at column ks - 1, the widest first (the longest jobs start first)
        int u = jj - urgent;
        for (int w = wmax; w >= 1; w >>= 1) {
            const int cnt = gpcc_chain_far_count(nt, ks, w, wmax);
            if (u >= cnt) {
                u -= cnt;
                continue;
            }
            rb = w + 1;
            while (u >= np - 2 - rb) {
                u -= np - 2 - rb;
                ++rb;
            }
            ra = rb + 2 + u;
            jb.kind = (w == 1) ? 2 : 4; jb.k = ks - w; jb.I = ks + ra; jb.J = ks + rb; jb.q = w;
            break;
        }
    } else {                          // the rest of local column 1 of this step: (k+4.., k+2)
        ra = 3 + (jj - urgent - far);
        jb.kind = 2; jb.I = ks + 1 + ra; jb.J = ks + 2;
    }
    return jb;
}
