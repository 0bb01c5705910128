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
//     NEAR1(k)     the updates by column k of the rest of local column 1.
#pragma once
#if defined(__HIPCC__)
#define GPCC_HD __host__ __device__ __forceinline__
#else
#define GPCC_HD inline
#endif

struct GpccChainJob {
    int kind;   // -1: none; 1: quarter solve of tile (I,k), quarter q; 2: update of tile (I,J) by column k; 3: its row quarter q
    int k, I, J, q;
};

// jobs per evaluation in the list of step ks
GPCC_HD int gpcc_chain_list_len(int nt, int ks, int helpers, int quarters)
{
    const int n = nt - ks - 1, np = n + 1;   // np: rows below the diagonal tile of step ks - 1
    const int qd = quarters ? 4 : 1;
    const int nsol = n >= 1 ? 4 * (n - (helpers ? 1 : 0)) : 0;   // (with helpers the solves of tile (k+1,k) are not queue jobs)
    const int urgent = n >= 2 ? nsol + qd * n + (2 * n - 4) : nsol;
    const int far = (ks >= 1 && np >= 5) ? (np - 4) * (np - 3) / 2 : 0;
    const int near = n >= 4 ? n - 3 : 0;
    return urgent + far + near;
}

// job jj (0 <= jj < gpcc_chain_list_len) of the list of step ks
GPCC_HD GpccChainJob gpcc_chain_decode(int nt, int ks, int jj, int helpers, int quarters)
{
    GpccChainJob jb;
    jb.kind = -1; jb.k = ks; jb.I = 0; jb.J = 0; jb.q = 0;
    const int qd = quarters ? 4 : 1;
    const int n = nt - ks - 1, np = n + 1;
    const int nsol = n >= 1 ? 4 * (n - (helpers ? 1 : 0)) : 0;
    const int nnext = n >= 2 ? qd * n : 0;
    const int urgent = n >= 2 ? nsol + nnext + (2 * n - 4) : nsol;
    const int far = (ks >= 1 && np >= 5) ? (np - 4) * (np - 3) / 2 : 0;
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
    } else if (jj < urgent + far) {   // the bulk of step ks - 1: local columns rb = 2 .. np - 3, rows ra = rb + 2 .. np - 1
        int u = jj - urgent;
        rb = 2;
        while (u >= np - 2 - rb) {
            u -= np - 2 - rb;
            ++rb;
        }
        ra = rb + 2 + u;
        jb.kind = 2; jb.k = ks - 1; jb.I = ks + ra; jb.J = ks + rb;
    } else {                          // the rest of local column 1 of this step: (k+4.., k+2)
        ra = 3 + (jj - urgent - far);
        jb.kind = 2; jb.I = ks + 1 + ra; jb.J = ks + 2;
    }
    return jb;
}
