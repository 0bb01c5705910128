// gpcc_chain_args.h -- what the host side (gpcc_hip.hip) and the kernel (gpcc_chain.hip.h, compiled in gpcc_chain_inst.hip) of the
// persistent few-evaluation launch share: sizes, the argument block, the launcher.
#pragma once
#include "gpcc_kernels.hip.h"
#include "gpcc_chain_queue.h"


#define GPCC_CHAIN_MAX_EVALS 32
#define GPCC_CHAIN_MAXRHS 4
#define GPCC_XIMG_STRIDE (GPCC_XIMG_ELEMS + 16 * GPCC_TILE)   /* doubles per (evaluation, step): the published blocks of L_kk, then S7 (gpcc_chain_trsmq) */
#define GPCC_XIMG_ELEMS (36 * 256)         /* the lower 36 blocks of inv(L_kk), 16 x 16 row-major each */
#define GPCC_INFO_TIMEOUT (-9)
#define GPCC_CHAIN_STEPVALS (2 + GPCC_CHAIN_MAXRHS * GPCC_CHAIN_MAXRHS)
#define GPCC_CHAIN_WTRACE_CAP 32768        /* job stamps kept per launch */
#define GPCC_CHAIN_TRACE_WORDS 80          /* stamps per diagonal step: 3 of the role + 8 per block step */


struct GpccChainArgs {
    unsigned *words;             // zeroed before every launch: [0] abort word; [16 + k] job counter of step k (k = 0 .. nt - 1); from qbase on, per
                                 // evaluation (ev_words each):
                                 //   xrow[nt] | l7[nt] | colflag[nt][8] | lcnt[ntiles] | ver[ntiles]
    double *ximg;                // evaluations x nt x GPCC_XIMG_STRIDE: row blocks of L_kk and the inv(D_f) as published (block (f, j) at gpcc_bi(f, j), row-major),
                                 // then S7: the last column block of L(k+1,k) before its product with inv(D_7)^T, in the tile's chunk layout
    double *stepval;             // evaluations x nt x GPCC_CHAIN_STEPVALS: per diagonal step [sum log L_ii of the block, first bad pivot, W'W]
    unsigned long long *trace;   // optional (NULL): evaluations x nt x GPCC_CHAIN_TRACE_WORDS wall-clock stamps of the chain (tools/chain_trace.py)
    unsigned long long *wtrace;  // optional (NULL): wtrace_cap x 4 stamps of the workers' jobs: [kind | step | tile, fetched, dependencies met, done]; words[1] counts
    int wtrace_cap;
    int ev_words;                // words per evaluation
    int qbase;                   // first per-evaluation word
    int helpers;                 // 1: four more dedicated workgroups per evaluation run the quarter solves of the tile below the diagonal
                                 //    (few evaluations: latency); 0: those solves are queue jobs like the others (more workers)
    int quarters;                // 1: the updates the next step needs at once are queue jobs of a quarter tile (gpcc_chain_updq); 0: whole tiles
    int batch;                   // 1, 2, 4 or 8: the widest column block of a bulk update job (job kind 4, gpcc_chain_queue.h); 1: one column per job
};

// per-evaluation jobs of step k (n = nt - k - 1 tile rows below the diagonal tile): 4 n quarter solves + n(n+1)/2 - 1 tile updates (tile
// (k+1,k+1) belongs to the chain) -- what the host sizes the grid by
__host__ __device__ __forceinline__ int gpcc_chain_jobs(int n) { return n <= 0 ? 0 : 4 * n + 3 * (n - 1) + n * (n + 1) / 2 - 1; }

// gpcc_chain_inst.hip: the kernel's dynamic LDS attribute (once per process and device), and one launch
hipError_t gpcc_chain_configure();
void gpcc_chain_launch(const GpccCtx &c, const GpccGroup &g, const GpccChainArgs &a, unsigned grid, hipStream_t s);
