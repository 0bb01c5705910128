/*
 * gpcc_hip.h -- C ABI of libgpcc_hip.so: the MI355X (gfx950) implementation of GPCC.jl's
 * marginal-log-likelihood hot path.
 *
 * The reference (pure Julia, /root/reference) has no FFI / plugin interface of its own
 * (SURVEY.md section 8(b)); this header IS the boundary a maintainer binds with `ccall`
 * (INTEGRATION.md shows the Julia shim) and the build's own host layer binds with ctypes
 * (gpcc.jl_amd/_capi.py).  Each entry point cites the reference code whose body it replaces.
 *
 * Conventions
 *   - plain pointers and sizes only; host pointers unless the name ends in _device.
 *   - every function returns int: 0 ok, <0 error (message via gpcc_last_error); never throws.
 *   - per-item status in info[] follows LAPACK potrf: 0 ok; >0 = order of the first
 *     non-positive pivot (the reference's PosDefException, swallowed by safewrapper at
 *     src/gpccfixdelay_marginaliseb.jl:153), loglik = NaN; -1 = some alpha <= 0 (the @assert at
 *     src/delayedCovariance.jl:3); -2 = rho <= 0 (error() at src/delayedCovariance.jl:5-7).
 *   - ragged light curves are passed flattened in band order, user order inside a band
 *     (Y = reduce(vcat, yarray), src/gpccfixdelay_marginaliseb.jl:85).
 *   - per-evaluation parameter blocks are ROW-major M x L (Julia passes an L x M Matrix).
 *   - no callbacks on the likelihood path (only gpcc_neldermead_batch, the optimiser on its own, takes one), no
 *     retained caller pointers, all device memory owned by the handle;
 *     one handle per thread/process; several handles (and processes) may share a GPU.
 *   - there is NO CPU fallback: without a HIP device every compute entry returns an error.
 */
#ifndef GPCC_HIP_H
#define GPCC_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gpcc_handle_s *gpcc_handle_t;

/* kernel ids: src/util.jl:15-23 (OU), :28 (rbf), :32-40 (matern32), :44-52 (matern52).
 * Any other Julia callable stays on the pure-Julia path. */
enum { GPCC_KERNEL_OU = 0, GPCC_KERNEL_RBF = 1, GPCC_KERNEL_MATERN32 = 2, GPCC_KERNEL_MATERN52 = 3 };
enum { GPCC_PRECISION_FP64 = 0, GPCC_PRECISION_FP32 = 1 };
enum { GPCC_MAX_BANDS = 8 };

enum {
    GPCC_OK = 0,
    GPCC_ERR_ARGUMENT = -1,      /* bad sizes / ids / NULL pointers */
    GPCC_ERR_HIP = -2,           /* a HIP runtime call failed (no device, OOM, launch failure) */
    GPCC_ERR_UNSUPPORTED = -3,   /* valid request this build does not implement yet */
    GPCC_ERR_STATE = -4
};

int gpcc_version(void);

/* What this library was built from: "src=<first 16 hex digits of the SHA-256 of its sources> defines=[<extra compiler defines>]".
 * The product library is always built without extra defines (gpcc.jl_amd/build.py refuses them for the default output path); A/B
 * libraries of tools/ carry theirs here.  bench.py prints it and quotes committed counter summaries only when it matches. */
const char *gpcc_build_info(void);

/* Last error text of a handle; handle == NULL gives the calling thread's last
 * handle-less error (gpcc_create / gpcc_covariance / gpcc_probabilities). */
const char *gpcc_last_error(gpcc_handle_t handle);

/* Replaces the per-call precompute of gpccfixdelay (src/gpccfixdelay_marginaliseb.jl:85-98;
 * src/gpccfixdelay.jl:85-96 when marginalise_b == 0): uploads t, sigma^2, Y - bbar once and keeps
 * mu_b[l] = mean(y_l), Sigma_b[l] = 100 var(y_l) (n-1).  Nl[l] >= 1 (>= 2 when marginalise_b). */
int gpcc_create(gpcc_handle_t *handle, int L, const int *Nl, const double *t, const double *y,
                const double *sigma, int kernel_id, int marginalise_b, int precision, int device_id);
int gpcc_destroy(gpcc_handle_t handle);

/* The same for a list of devices of ONE process: the reference parallelises the delay-grid map
 * with pmap workers (README.md:181-211, :258-287); here one handle owns a replica of the light curves on every listed
 * device.  gpcc_loglik_batch and gpcc_grid_loglik then cut every batch into contiguous blocks, one host thread and
 * one device per block, and collect [loglik | info] with ONE all-gather (RCCL over xGMI, ncclCommInitAll +
 * ncclAllGather; no torch, no MPI), after which every device holds the whole vector (gpcc_multi_gathered).
 * RCCL refuses a communicator with a repeated device: lists with duplicates (the one-GPU rehearsal {0, 0}) and
 * single-entry lists gather through host memory instead; gpcc_get_option(h, "gather_mode") says which.
 * A multi-device handle is accepted by every entry point that takes a handle, except gpcc_loglik_batch_device
 * (device pointers belong to one device): the single-matrix utilities (gpcc_predict, gpcc_posterior_offsets,
 * gpcc_model_matrix, gpcc_factor_dense, profiling) run on device_ids[0]; options apply to all devices. */
enum { GPCC_GATHER_NONE = 0, GPCC_GATHER_RCCL = 1, GPCC_GATHER_HOST = 2 };
int gpcc_create_multi(gpcc_handle_t *handle, int L, const int *Nl, const double *t, const double *y,
                      const double *sigma, int kernel_id, int marginalise_b, int precision,
                      const int *device_ids, int n_devices);

/* The gathered vector of the last gpcc_loglik_batch as it sits on device `which` of a multi-device handle:
 * n_devices blocks of [loglik(blk) | info-as-double(blk)], blk = ceil(M / n_devices) written to *blk_out
 * (out == NULL only queries blk).  After gpcc_grid_loglik on a multi-device handle (round 4: the fit is sharded BY DELAY, device i
 * fits the delays i, i + n, i + 2n, ... with its own lock-step optimiser, and the results are collected by ONE all-gather):
 * n_devices blocks of blk = ceil(G / n_devices) rows [loglik, info, iterations, rho, alpha(L)], i.e. gpcc_get_option("gather_width")
 * = L + 4 doubles per row instead of 2 x blk. */
int gpcc_multi_gathered(gpcc_handle_t handle, int which, long *blk_out, double *out, long capacity);

/* Timing of the last gpcc_loglik_batch on a multi-device handle: compute_ms[n_devices] = each device's share on its own
 * stream (HIP events), *gather_ms = the gather phase (RCCL all-gather + final copy; host wall clock), *total_ms = the whole
 * call.  Any pointer may be NULL.  (One persistent host thread per device runs the shares; none is created per batch.) */
int gpcc_multi_stats(gpcc_handle_t handle, double *compute_ms, double *gather_ms, double *total_ms);

/* Options (gpcc_set_option / gpcc_get_option).  Defaults are the measured best; none changes WHAT is computed, several select the
 * factorisation path by group size (results of different paths agree to ~1e-13 relative in fp64).  History of every option, and of the
 * variants that were measured and removed, is in LOG.md.
 *
 *   key                      default  meaning
 *   -----------------------  -------  ---------------------------------------------------------------------------------------------
 *   streams                  2        groups of a batch alternate between this many HIP streams
 *   slots_per_stream         256      evaluations resident per group (fewer where 256 slots exceed 55 % of the device's memory)
 *   chain_max                32       groups of at most this many evaluations at N >= 384 -- a single objective(alpha, rho),
 *                                     marginaliseb.jl:133-141 called from Optim's loop, :209-211 -- run as ONE persistent launch
 *                                     (gpcc_chain: two chain workgroups per evaluation carry diagonal step -> column solve -> next
 *                                     diagonal tile without leaving their CUs, all other CUs pull trailing-update jobs; fp64 handles);
 *                                     0 = the two-launches-per-step path below
 *   chain_work_max           4096     ... up to 12 evaluations: and evaluations x (N/128)^2 at most this (12 evaluations up to N = 2048, 4 at
 *                                     N = 4096: above, the path below is faster)
 *   chain_wide_work_max      1024     ... 13 .. chain_max evaluations: and evaluations x (N/128)^2 at most this (32 at N <= 512, 28 at N = 768, 16 at
 *                                     N = 1024: there the alternative,
 *                                     two halves on two streams, is slower even when the halves do overlap -- and they only do when the
 *                                     runtime maps the two streams onto different hardware queues)
 *   chain_workers_max        0        ... at most this many worker workgroups per launch (0 = as many as the widest step has jobs, up to the
 *                                     chip).  For P caller processes sharing a GPU at N >= 2048, where every launch would ask for all CUs and
 *                                     the launches queue: about n_cus / P - 6 (4 processes at N = 4096: 1210 instead of 800 evaluations/s in
 *                                     total; a single caller is 2.4x slower with it).  Changes no result.
 *   chain_batch              8        ... bulk tile updates take aligned blocks of up to this many columns (1, 2, 4, 8) per job where a tile has
 *                                     the slack -- far from the diagonal 8, towards it 4, 2, 1 (8 w chunks of K instead of 8: one fixed cost of
 *                                     ~6 us per w x 13.7 us of MFMAs; the same sums in the same order -- the same bits); 1 = one column per job
 *   chain_batch_min          40000    ... for groups of evaluations x (N/128)^3 >= this (the worker-bound ones: from 3 evaluations at N = 3072, 2 at
 *                                     N = 4096; -3 ... -12 % there, nothing or slightly worse where the chain is the bound)
 *   chain_helpers_max        6        ... with four more dedicated workgroups per evaluation (the solves of the tile below the diagonal run
 *                                     beside every diagonal step) for groups of at most this many evaluations
 *   chain_quarters_max       2        ... and the tile updates the next step needs at once as four quarter-tile jobs each, for groups of
 *                                     at most this many evaluations (latency for CU time: 4 x 7.4 us instead of 22 us per tile)
 *   fused_small_max          12       ... otherwise such groups run the kernels gpcc_panel_trsm_rows and gpcc_small_step: 2 launches per step
 *   right_looking_max        12       groups of at most this many evaluations factorise right-looking
 *   fused_solve              1        larger groups: panel solve inside the update kernel (gpcc_syrk_diag + gpcc_update_solve);
 *                                     0 = the three-kernel path (gpcc_panel_update, gpcc_diag_factor, gpcc_panel_trsm) everywhere
 *   fused_solve_min          112      ... from this group size on (smaller left-looking groups keep the three-kernel path)
 *   fused_solve_min_split    64       ... the same threshold for each half of a split group (follows fused_solve_min when that is
 *                                     set below 64 or above 112)
 *   fold_assembly            1        groups of more than fused_small_max evaluations do not write the off-diagonal tiles of
 *                                     delayedCovariance: the job that reads a tile first evaluates its elements (same bits)
 *   hybrid_tail              1        three-kernel groups finish right-looking once their trailing matrices fit hybrid_mall_mb
 *   hybrid_mall_mb           400      ... that budget (MB; the Infinity Cache is 256 MiB)
 *   hybrid_occ               384      ... and only steps with fewer left-looking jobs than this become right-looking
 *   split_min                24       a group of split_min .. split_max evaluations runs as two halves on two streams (0 = never)
 *   split_max                240      (see split_min)
 *   split_nt_min             12       ... at N > 128 (split_nt_min - 1)
 *   split_small              1        ... and the smaller groups for which that was measured to pay
 *   shared_prefix            1        0 off; 1: gpcc_loglik_batch detects a fixed-hyper-parameter delay sweep (README.md:172-174) and
 *                                     factorises the tile rows inside band 1 once per group (bitwise the same); 2: the caller asserts it
 *   small_n                  1        N <= 383 runs the small-N family (one launch per batch, the matrix in registers, always fp64:
 *                                     an fp32 handle's precision is ignored there); 0 = the tile kernels; env GPCC_SMALL_N = default
 *   small_wide_max           512      small-N batches of at most this many evaluations run four waves per evaluation
 *   fit_device_unpack        1        gpcc_grid_loglik, small-N path: the kernel unpacks the optimiser's vectors itself
 *   fit_speculate            1        ... latency-bound rounds evaluate all four candidate points of an iteration at once
 *   fit_threads              0        ... host threads (slices) of a large fit; 0 = by size
 *   fp32_refine              1        fp32 handles: fp64 refinement of the quadratic forms
 *   fp32_guard               1        fp32 handles: evaluations whose pivot ratios exceed the limits are repeated in fp64
 *   fp32_assemble            1        fp32 handles: tiles inside one band pair are evaluated in fp32
 *   fp32_chain               1        fp32 handles: a call that chain_max / chain_work_max would give to the persistent launch on an fp64
 *                                     handle (one objective(alpha, rho) at N >= 384: bound by latency, not by the matrix pipe) is
 *                                     evaluated in fp64 by it, on the handle's internal fp64 twin: the fp64 handle's bits, conditioning
 *                                     estimates 0; 0 = fp32 tiles on the launch-per-step path ("fp32_chain_count": evaluations so far)
 *
 * Read-only keys of gpcc_get_option: "N", "Np", "precision", "bytes_per_slot", "share_tiles", "n_devices", "gather_mode",
 * "gather_width", "small_n_max" (383), "small_n_active", "small_n_count", "chain_count" (evaluations that took the persistent
 * launch so far), "chain_last_grid" (workgroups of the last one), "fp32_guard_count", "fp32_chain_count", "workspace_streams" / "workspace_slots" (what the workspace really holds: smaller than "streams" /
 * "slots_per_stream" only if the device's memory was short when it was allocated -- then gpcc_last_error carries a note; the
 * options themselves are never rewritten). */
int gpcc_set_option(gpcc_handle_t handle, const char *key, long value);
long gpcc_get_option(gpcc_handle_t handle, const char *key);

/* mu_b[L], Sigma_b[L], resid[N] as precomputed at create (any pointer may be NULL). */
int gpcc_get_constants(gpcc_handle_t handle, double *mean_b, double *Sigma_b, double *resid);

/* fp32 handles (N >= 384; smaller problems are evaluated in fp64 by the small-N kernels and report zeros here):
 * [sum_i K_ii / d_i, max_i K_ii / d_i] over the Cholesky pivots d_i of each of the first M evaluations
 * of the last gpcc_loglik_batch / gpcc_loglik_batch_device call (2 M doubles) -- the conditioning measure behind the
 * fp32 accuracy guard.  An fp32 handle (a) refines the quadratic forms r'K^-1 r, Q'K^-1 Q, Q'K^-1 r in fp64 after the
 * fp32 factorisation (one backward solve + one pass over the fp64 elements of K regenerated on the fly: second-order
 * accurate, N^2 work; option "fp32_refine", default 1) and (b) repeats in fp64 -- internally, on a small fp64 workspace
 * it creates on first use -- every evaluation whose mean pivot ratio sum / N exceeds 300 (30 without refinement), whose LARGEST
 * ratio exceeds 5e3 (round 3: an adversarial search found evaluations below the mean limit with errors up to 0.6) or whose
 * fp32 factorisation met a non-positive pivot, so that results stay within the 1e-3 bar of fp32 also for
 * ill-conditioned hyper-parameters (calibration: DESIGN.md 4.7).  Options: "fp32_guard" (1 default, 0 = never repeat),
 * "fp32_guard_count" (read-only: evaluations repeated so far).  The guard reads the estimates back, so an fp32
 * handle synchronises the caller's stream once per call (also in the _device form).  An evaluation whose fp32 factorisation
 * broke down, or whose arguments were refused, reports +inf for both numbers.  Option "fp32_assemble" (1 default): the elements of
 * fp32 tiles that lie inside one band pair are evaluated in fp32 too (distance in fp64, rounded once; v_exp_f32) instead of in fp64
 * and rounded once -- the quadratic forms are refined from the exact fp64 elements either way. */
int gpcc_get_conditioning(gpcc_handle_t handle, int M, double *out);

/* THE HOT PATH.  objective(alpha, rho) of src/gpccfixdelay_marginaliseb.jl:133-141
 * (src/gpccfixdelay.jl:131-139 when marginalise_b == 0) for M independent (tau, alpha, rho):
 *   K = delayedCovariance(kernel, alpha, tau, rho, tarray) + Sobs + B ; logpdf(MvNormal(bbar, K), Y).
 * M = 1 serves the closure call site (:145-153, :209, :211); large M serves the delay-grid sweep
 * (README.md:172-174, :202-206, :231).  Blocking; caller-allocated outputs. */
int gpcc_loglik_batch(gpcc_handle_t handle, int M, const double *delays, const double *alpha,
                      const double *rho, double *loglik, int *info);

/* Same with DEVICE pointers, enqueued behind `stream` (a hipStream_t, NULL = default stream) and
 * joined back into it: asynchronous, outputs valid once `stream` has drained. */
int gpcc_loglik_batch_device(gpcc_handle_t handle, int M, const double *d_delays,
                             const double *d_alpha, const double *d_rho, double *d_loglik,
                             int *d_info, void *stream);

/* Dense K = delayedCovariance + Sobs + B of one (tau, alpha, rho), column-major N x N
 * (src/gpccfixdelay_marginaliseb.jl:135, :237-241) -- for prediction and tests. */
int gpcc_model_matrix(gpcc_handle_t handle, const double *delays, const double *alpha, double rho,
                      double *K_out);

/* Lower Cholesky factor of that K (PDMat's cholesky at marginaliseb.jl:139), column-major N x N,
 * strict upper triangle zero; *info as above. */
int gpcc_factor_dense(gpcc_handle_t handle, const double *delays, const double *alpha, double rho,
                      double *L_out, int *info);

/* predictTest(ttest::Vector{Vector}) of src/gpccfixdelay_marginaliseb.jl:259-289 at given (tau, alpha, rho):
 * joint predictive mean mu_out[sum Ntest] and covariance Sigma_out (column-major, + JITTER*I, :279) for
 * Ntest[l] test times per band (flattened in band order); also the training log-likelihood / info.
 * One augmented factorisation on the device: Sigma = cB - V'V, mu = V'w + Q* mu_b with V = L^-1 kB*. */
int gpcc_predict(gpcc_handle_t handle, const double *delays, const double *alpha, double rho,
                 const int *Ntest, const double *ttest, double *mu_out, double *Sigma_out,
                 double *loglik, int *info);

/* Posterior of the offsets b (src/gpccfixdelay_marginaliseb.jl:248-252): mu_postb[L], Sigma_postb[L x L]
 * (column-major, symmetrised).  The N x N solves (Sobs + K) \ [Q Y] run on the device as an augmented
 * factorisation; only the final L x L inverse is host arithmetic. */
int gpcc_posterior_offsets(gpcc_handle_t handle, const double *delays, const double *alpha, double rho,
                           double *mu_postb, double *Sigma_postb, int *info);

/* logpdf(MvNormal(mu, Sigma), x) for an explicit dense Sigma (column-major n x n; mu may be NULL = 0):
 * the test log-likelihood of predictTest(ttest, ytest, sigmatest), src/gpccfixdelay_marginaliseb.jl:311-343
 * (:325).  info > 0 is the PosDefException the reference catches at :327-341. */
int gpcc_mvnormal_logpdf(int n, const double *Sigma, const double *mu, const double *x, double *loglik,
                         int *info, int device_id);

/* delayedCovariance(kernel, scale, delays, rho, x, y) of src/delayedCovariance.jl:1-35 (pass
 * y == x, Ny == Nx for the 5-argument form, :38).  out is column-major (sum Nx) x (sum Ny).
 * Returns GPCC_ERR_ARGUMENT with the reference's message for scale <= 0 / rho <= 0. */
int gpcc_covariance(int kernel_id, int L, const double *scale, const double *delays, double rho,
                    const int *Nx, const double *x, const int *Ny, const double *y, double *out,
                    int device_id);

/* The per-delay model FIT for a whole grid of candidate delays -- what README.md:172-174 maps over
 * (`gpcc(...; delays = [0; d])[1]` for each d) -- as one call.  For each row of delays (G x L) and each
 * restart: the best of `initialrandom` random candidates (marginaliseb.jl:209) starts Nelder-Mead with
 * Optim's defaults and Options(iterations, g_tol = 1e-6) (:205-211) over the unconstrained parameters of
 * `unpack` (:112-126); the best restart wins (:222-226).  All G x numberofrestarts minimisations advance in
 * lock-step, every optimiser round being one gpcc_loglik_batch; each keeps its own trajectory.
 *   loglik_out[g] = -result.minimum (:351), alpha_out (G x L), rho_out[g] = the optimised hyper-parameters,
 *   info_out[g]   = 0, or 1 when no evaluated point was valid (loglik_out[g] = -Inf),
 *   iterations_out[g] (may be NULL) = Nelder-Mead iterations of the winning restart,
 *   stats_out (may be NULL) = {objective evaluations, batched rounds}.
 * init_params: numberofrestarts x initialrandom x (L+1) candidates in the optimiser's unconstrained
 * coordinates (the same for every delay: each reference gpcc call seeds its own generator with `seed`).  A
 * Julia caller draws them exactly as the reference does (MersenneTwister(seed), :160-196) and passes them;
 * NULL = drawn here with the same recipe from xoshiro256++(seed) (gpcc_initial_params shows them). */
int gpcc_grid_loglik(gpcc_handle_t h, int G, const double *delays, int iterations, int numberofrestarts,
                     int initialrandom, double rhomin, double rhomax, unsigned long long seed,
                     const double *init_params, double *loglik_out, double *alpha_out, double *rho_out,
                     int *info_out, int *iterations_out, long long *stats_out);

/* The candidates gpcc_grid_loglik uses when init_params == NULL (out: restarts x initialrandom x (L+1)). */
int gpcc_initial_params(gpcc_handle_t h, int numberofrestarts, int initialrandom, double rhomin, double rhomax,
                        unsigned long long seed, double *out);

/* The optimiser inside gpcc_grid_loglik on its own: P independent Nelder-Mead minimisations of dimension n in
 * lock-step (Optim.jl's NelderMead defaults as the reference uses them, marginaliseb.jl:205-211) over a batched
 * objective.  f(ctx, K, pidx, X, out) must write out[i] = objective of problem pidx[i] at row i of X (K x n);
 * NaN / +Inf = rejected point; non-zero return aborts with that code.  x0, xmin: P x n; fmin: P;
 * iterations_out (P) and stats_out ({evaluations, rounds}) may be NULL.  Host only; the tests pin this against the
 * numpy and the scalar restatements of the same algorithm. */
typedef int (*gpcc_batch_objective_t)(void *ctx, long K, const long *pidx, const double *X, double *out);
int gpcc_neldermead_batch(long P, int n, int iterations, double g_tol, const double *x0, gpcc_batch_objective_t f,
                          void *ctx, double *xmin, double *fmin, int *iterations_out, long long *stats_out);

/* `unpack` of marginaliseb.jl:112-126 for M parameter vectors X (M x (L+1)): alpha = makepositive(x[1:L]) + 1e-8,
 * rho = transformbetween(x[L+1], rhomin, rhomax).  MiscUtil.jl's source is not part of the reference tree:
 * makepositive is taken to be softplus, transformbetween(x, a, b) = a + (b - a) / (1 + exp(-x)).  Host only. */
int gpcc_unpack_params(int M, int L, const double *X, double rhomin, double rhomax, double *alpha, double *rho);

/* getprobabilities(loglikel[, logpriorpdfvalues]) of src/getprobabilities.jl:1-20;
 * logprior == NULL is the 1-argument form (log-prior of ones, :3). */
int gpcc_probabilities(int G, const double *loglik, const double *logprior, double *out,
                       int device_id);
int gpcc_probabilities_device(int G, const double *d_loglik, const double *d_logprior,
                              double *d_out, void *stream);

/* Per-kernel HIP-event timing (bench.py's roofline leg).  While enabled every launch is
 * bracketed by events on its own stream and groups run on ONE stream. */
enum {
    GPCC_PROF_ASSEMBLE = 0,     /* gpcc_assemble_tiles     -- HBM-bound */
    GPCC_PROF_PANEL_UPDATE = 1, /* gpcc_panel_update       -- fp64 MFMA, dominant */
    GPCC_PROF_DIAG = 2,         /* gpcc_diag_factor */
    GPCC_PROF_TRSM = 3,         /* gpcc_panel_trsm         -- fp64 MFMA */
    GPCC_PROF_REFINE = 4,       /* fp32 mode: backward solve + X' K0 X + final arithmetic */
    GPCC_PROF_SMALL_STEP = 5,   /* gpcc_small_step: trailing update + next diagonal step in one launch (a few evaluations) */
    GPCC_PROF_SMALL_EVAL = 6,   /* gpcc_small_eval / gpcc_smallw_eval: a whole evaluation (assembly + Cholesky + solve) of N <= 383 points in one or four waves */
    GPCC_PROF_COUNT = 7
};
int gpcc_profile_enable(gpcc_handle_t handle, int on);
int gpcc_profile_reset(gpcc_handle_t handle);
int gpcc_profile_get(gpcc_handle_t handle, int which, long *launches, double *total_ms);

/* On-device self-test of the f64 MFMA fragment maps and a timing probe of the fp64 MFMA rate:
 * returns 0 when the maps are as the kernels assume; *tflops (may be NULL) = measured rate. */
/* Measurement plumbing of the persistent few-evaluation launch (option "chain_max"): with option "chain_trace" = 1 its chain
 * workgroups stamp the device's wall clock per diagonal step k, 80 stamps each; out_us[80 k + 0 / 1 / 2] = microseconds (from the
 * first stamp of the evaluation) at which the workgroup of step k began to build tile (k,k), had it complete (first pivot next), had
 * published the whole step; [80 k + 8 + 8 jb + 0 .. 7] = block step jb of the diagonal step: begins, wave 0 done, behind its first
 * barrier, behind its second; wave 0: fold done, block loaded, factored, stored; -1 where nothing was stamped.  `evaluation` = index in the last group of at most
 * chain_max evaluations; capacity >= 80 nt doubles (nt = Np / 128).  tools/chain_trace.py prints the critical chain from it. */
int gpcc_chain_trace(gpcc_handle_t handle, int evaluation, double *out_us, int capacity);
/* ... and the workers' jobs of the same launch: rows of 6 doubles [kind (1 quarter-tile solve, 2 tile update), step k, index in the step,
 * fetched, dependencies met, done] (microseconds from the first fetch), at most capacity_rows (32768 are kept per launch). */
int gpcc_chain_jobs_trace(gpcc_handle_t handle, double *out, long capacity_rows, long *rows_out);

int gpcc_selftest(int device_id, double *mfma_f64_tflops);

#ifdef __cplusplus
}
#endif
#endif
