// gpcc_hip.hip -- host side of libgpcc_hip.so: the C ABI of include/gpcc_hip.h over the gfx950
// kernels of gpcc_kernels.hip.h.  No CPU fallback: every compute entry needs a HIP device.
#include "gpcc_kernels.hip.h"
#include "gpcc_small.hip.h"
#include "gpcc_chain_args.h"
#include "gpcc_fit.h"

#include "../../include/gpcc_hip.h"

#include <rccl/rccl.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <limits>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#define GPCC_VERSION_NUMBER 100
#define GPCC_MAX_STREAMS 8

static thread_local std::string g_err;

struct ProfRec { int which; hipEvent_t a, b; };

// small-N path, host-pointer entries: a LANE = one stream + pinned, device-mapped host buffers.  The kernel reads the
// parameters from them and writes loglik / info into them directly (zero-copy over PCIe: 8 (2L+1) bytes in, 12 bytes out per
// evaluation), so a call is pack + ONE launch + one stream synchronisation, without a single hipMemcpy.  Lane 0 serves
// gpcc_loglik_batch; a large per-delay fit (gpcc_grid_loglik) runs several lanes from as many host threads.
#define GPCC_MAX_LANES 4
struct SmallLane {
    hipStream_t stream = nullptr;
    double *par = nullptr, *ll = nullptr;
    int *info = nullptr, *row = nullptr;
    long cap = 0;
    void release()
    {
        if (par) hipHostFree(par);
        if (ll) hipHostFree(ll);
        if (info) hipHostFree(info);
        if (row) hipHostFree(row);
        if (stream) hipStreamDestroy(stream);
        par = ll = nullptr; info = row = nullptr; stream = nullptr; cap = 0;
    }
};

// one persistent host thread per device of a multi-device handle: it takes a job (this device's share of a batch), runs it
// with its device current, and reports back -- no thread is created or joined per batch
struct MultiWorker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, done = true, stop = false;
    int rc = 0;
    void loop()
    {
        for (;;) {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return has_job || stop; });
            if (stop) return;
            std::function<int()> f = std::move(job);
            has_job = false;
            lk.unlock();
            const int r = f();
            lk.lock();
            rc = r;
            done = true;
            cv.notify_all();
        }
    }
    void submit(std::function<int()> f)
    {
        std::lock_guard<std::mutex> lk(mu);
        job = std::move(f);
        has_job = true;
        done = false;
        cv.notify_all();
    }
    int wait()
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return done; });
        return rc;
    }
    void shutdown()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
            cv.notify_all();
        }
        if (th.joinable()) th.join();
    }
};

struct gpcc_handle_s {
    int device = 0;
    int L = 0, N = 0, Np = 0, nt = 0, kernel_id = 0, mb = 1, precision = 0;
    int nrhs = 1, woodbury = 0;   // fp32 + marginalised b: K0 in fp32, B through the capacitance matrix in fp64
    int Nl[GPCC_MAXL];
    double mean_b[GPCC_MAXL], sigma_b[GPCC_MAXL];
    std::vector<double> resid_host;
    double *d_t = nullptr, *d_sig2 = nullptr, *d_resid = nullptr, *d_yv = nullptr;
    int *d_band = nullptr;
    std::vector<double> t_host, y_host, sig2_host;
    double tmid = 0.0;       // midpoint of the observation times (GpccCtx::tmid)
    std::vector<int> band_host;
    // options
    int streams = 2;         // groups of a batch alternate between this many streams: the next group's assembly and first steps fill
                             // the idle CUs of the previous group's last steps (headline +0.9 %, 4096 delays at N = 2048 +5 %:
                             // profiles/r03/two_streams_ab.log); a batch of one group is unaffected
    int slots_per_stream = 256, right_looking_max = GPCC_RIGHT_LOOKING_MAX;
    int fused_small_max = 12;  // groups of at most this many evaluations run gpcc_small_step (update + next diagonal step in one launch)
    int chain_max = 32;        // option "chain_max": groups of at most this many evaluations run as ONE persistent launch (gpcc_chain.hip.h; fp64 handles, fp32 ones through their twin) where
                               // the policy below says it wins; 0 = never
    long chain_wide_work_max = 1024; // option "chain_wide_work_max": ... groups of 13 .. chain_max evaluations: evaluations x (N/128)^2 at most this (32 at N <= 512, 28 at N = 768,
                                     // 16 at N = 1024)
    std::atomic<long> chain_count{0};   // evaluations that took the persistent launch so far ("chain_count")
    long chain_work_max = 4096; // option "chain_work_max": ... and at most this many evaluations x tile-steps^2 (12 at N <= 2048, 4 at N = 4096: above, the
                                // launch-per-step path is the faster one -- profiles/r05/latency_small_batches.log)
    int chain_quarters_max = 2; // option "chain_quarters_max": groups of at most this many evaluations update the tiles the next step needs at once in quarter-tile jobs
    int chain_workers_max = 0;  // option "chain_workers_max": at most this many worker workgroups per persistent launch (0 = as many as the widest step has jobs, up to the
                                // chip): caller PROCESSES that share a GPU at N >= 2048 (each launch would otherwise ask for every CU and the launches queue)
    long chain_last_grid = 0;   // workgroups of the last persistent launch ("chain_last_grid")
    int chain_batch = 8;        // option "chain_batch": bulk tile updates of the persistent launch take aligned blocks of up to this many columns per job (1, 2, 4, 8)
                                // where the tile has the slack
    long chain_batch_min = 40000; // option "chain_batch_min": ... for groups with evaluations x (N/128)^3 >= this (3 evaluations at N = 3072, 2 at N = 4096, 10 at N = 2048)
    int chain_helpers_max = 6; // option "chain_helpers_max": groups of at most this many evaluations give each evaluation four more dedicated workgroups
                               // (the quarter solves of the tile below the diagonal run beside every diagonal step instead of being queue jobs)
    int chain_trace = 0;       // option "chain_trace": the chain workgroups stamp their phases (gpcc_chain_trace; tools/chain_trace.py)
    unsigned *d_chain_words = nullptr;             // per workspace stream: the launch's flag words (zeroed before every launch)
    double *d_ximg = nullptr, *d_stepval = nullptr;   // ... the published inverses of the diagonal blocks, the steps' scalars
    unsigned long long *d_chain_trace = nullptr;
    long chain_region_words = 0;                   // words per stream region
    int chain_ev_words = 0, chain_qbase = 0, chain_streams = 0, n_cus = 0;
    int shared_prefix = 1;   // 0 off, 1 auto (host-pointer API detects it), 2 the caller asserts it
    int fused_solve_min = 112;   // ... from this group size on (below it the diagonal tile's serial K-loop on ONE CU per evaluation costs
                                 // more than the fused solve saves: measured crossover 96-128 evaluations at N = 1024 and N = 4096)
    int fused_solve_min_split = 64;   // option "fused_solve_min_split": ... for the two halves of a split group (split_min): 128-160 evaluations at
                                 // N = 4096 +1.5 ... 2 %, at N = 2048 +4 ... 6 % (profiles/r04/fused_solve_min_sweep_after_fold.log)
    int fused_solve = 1;     // option "fused_solve": left-looking groups run gpcc_syrk_diag + gpcc_update_solve (2 launches per step)
    int fold_assembly = 1;   // option "fold_assembly": groups of more than fused_small_max evaluations do not assemble the off-diagonal tiles; the job
                             // of the factorisation that reads a tile first (gpcc_update_solve / gpcc_panel_update) evaluates it into its
                             // accumulators, bit for bit the assembled tile (GpccCtx::fold)
    int hybrid_tail = 1;     // option "hybrid_tail": left-looking groups of the three-kernel path finish RIGHT-looking once their
                             // trailing matrices fit the Infinity Cache and the left-looking steps would leave CUs idle
    int hybrid_mall_mb = 400;   // option "hybrid_mall_mb": budget for the trailing matrices of a group (256 MiB Infinity Cache; measured
                                // best at 400: profiles/r03/midsize_tail_budget_sweep.log)
    int hybrid_occ = 384;       // option "hybrid_occ": ... and only steps with fewer left-looking jobs than this become right-looking
    int split_min = 24;         // option "split_min": a group of at least this many evaluations (0 = never) runs as TWO halves on two
                                // streams, so that the update of one half hides the diagonal-step / panel-solve chain of the other ...
    int split_max = 240;        // option "split_max": ... up to this many (a full group of 256 has no idle chain left to hide: -1.3 % when split).  Round 3: 160
                                // (halves on the three-kernel path: 192 was -1 ... -3 %); with fused halves (fused_solve_min_split) 176-224 evaluations gain
                                // +0.7 ... 1.8 % at N = 4096 and nothing at N = 2048 (profiles/r04/fused_solve_min_sweep_after_fold.log) ...
    int split_nt_min = 12;      // option "split_nt_min": ... at N > 128 * (this - 1) (below, the halves only get in each other's way).
                                // Measured: profiles/r03/midsize_split_groups.log (+3 ... +8 % for 24-111 evaluations at N >= 2048)
    int split_small = 1;        // option "split_small": smaller groups too, where it was measured to pay (same log): 13-23 evaluations
                                // up to N = 2048 and 13-19 up to N = 3072 (two right-looking halves: +5 ... +35 %), 6-12 evaluations
                                // from N = 2945 on (+5 ... +10 %)
    int small_n = 1;         // option "small_n": N <= GPCC_SMALL_MAXN runs gpcc_small_eval (one launch per batch, one wave per evaluation,
                             // the matrix in registers; always fp64) instead of the tile kernels
    std::atomic<long> small_count{0};   // evaluations that took that path so far ("small_n_count")
    int small_wide_max = 512;           // option "small_wide_max": batches of at most this many evaluations (two workgroups per CU) run four waves per evaluation
    bool mixed_rows = false; // some tile row straddles two bands or holds padding (GpccCtx::fold_mixed)
    int share_tiles = 0;     // tile rows wholly inside band 1
    bool share_now = false;  // decision for the batch being enqueued
    // workspace
    bool ws_ready = false;
    int ws_streams = 0, ws_slots = 0;           // what the workspace holds (smaller than the options if the memory was short)
    int ws_req_streams = 0, ws_req_slots = 0;   // the option values it was built for
    double *d_tiles = nullptr, *d_linv = nullptr, *d_z = nullptr, *d_w = nullptr, *d_logdet = nullptr,
           *d_quad = nullptr;   // d_quad: Gram matrices (slots x MAXRHS^2)
    int *d_info = nullptr;
    double *d_kdiag = nullptr, *d_cond = nullptr;   // fp32 mode: diag(K) as assembled, pivot-ratio sums (GpccCtx)
    double *d_gpart = nullptr;                      // fp32 mode: per-tile partials of X' K0 X (refinement)
    double *d_sep = nullptr, *d_seps = nullptr;     // separable factors of the points, the distance scale (GpccCtx::sep, ::seps)
    int *d_sepflag = nullptr;                       // per-tile-row flags (GpccCtx::sepflag)
    int fp32_refine = 1;                            // option "fp32_refine": 0 = no refinement of the quadratic forms
    int fp32_assemble = 1;                          // option "fp32_assemble": fp32 tiles inside one band pair are EVALUATED in fp32 as well (0: in fp64,
                                                    // rounded once).  +3.6 % at N = 4096; soak and adversarial search unchanged (worst 2.9e-5 / 3.4e-4):
                                                    // profiles/r03/fp32_assembly_in_fp32.log
    long slot_stride = 0;
    hipStream_t str[GPCC_MAX_STREAMS] = {};
    hipEvent_t ev_done[GPCC_MAX_STREAMS] = {};
    hipStream_t str2[GPCC_MAX_STREAMS] = {};      // second half of a split group (option "split_min")
    hipEvent_t ev_fork[GPCC_MAX_STREAMS] = {}, ev_join[GPCC_MAX_STREAMS] = {};
    hipEvent_t ev_start = nullptr;
    hipStream_t main_stream = nullptr;
    // staging for the host-pointer API
    double *d_par = nullptr, *d_out = nullptr, *d_ocond = nullptr;
    int *d_oinfo = nullptr;
    long par_cap = 0;
    SmallLane lanes[GPCC_MAX_LANES];
    double *d_cand = nullptr;   // gpcc_grid_loglik: the candidate-delay table (G x L) on the device
    long cand_cap = 0;
    int fit_speculate = 1;      // option "fit_speculate": speculative optimiser rounds on the small-N path (gpcc_fit.h)
    int fit_device_unpack = 1;  // option "fit_device_unpack": the small-N kernel unpacks the optimiser's vectors itself
    int fit_threads = 0;        // option "fit_threads": host threads (lanes) of a small-N fit; 0 = by problem count
    // fp32 mode: a-posteriori accuracy guard.  Every evaluation reports the sum and the maximum of
    // K_ii / d_i over its pivots; where the error model built on them exceeds the budget, the evaluation is repeated
    // on an internal fp64 handle (`fb`, created on first use) and its result replaces the fp32 one.
    int fp32_guard = 1;              // option "fp32_guard": 0 = raw fp32 results
    int fp32_chain = 1;              // option "fp32_chain": a call the persistent launch would take on an fp64 handle is evaluated by the fp64 twin
    long fp32_chain_count = 0;       // evaluations that went that way ("fp32_chain_count")
    long cond_cap = 0, fb_cap = 0;
    int *d_fb_idx = nullptr, *d_fb_info = nullptr;
    double *d_fb_par = nullptr, *d_fb_out = nullptr;
    gpcc_handle_t fb = nullptr;
    long fb_count = 0;               // evaluations repeated in fp64 so far ("fp32_guard_count")
    long fb_slots = 0;               // slots of the fp64 workspace behind the guard
    std::vector<double> cond_host, ll_host, sigma_host;
    std::vector<int> info_host, fb_idx_host;
    // profiling
    bool prof = false;
    std::vector<ProfRec> recs;
    long prof_n[GPCC_PROF_COUNT] = {};
    double prof_ms[GPCC_PROF_COUNT] = {};
    std::string err;
    // multi-device flavour (gpcc_create_multi): this handle owns one single-device handle per entry of device_ids and
    // nothing else on a device; batches are cut into contiguous blocks, one worker thread per device, and the results
    // are collected with ONE all-gather (RCCL over xGMI; through host memory when device ids repeat)
    std::vector<gpcc_handle_t> subs;
    std::vector<ncclComm_t> comms;        // one per sub-handle when the gather is RCCL's
    std::vector<double *> d_send, d_recv; // per sub: [loglik(blk) | info(blk)] and n x that
    std::vector<double> h_gather;
    std::vector<std::unique_ptr<MultiWorker>> workers;   // one per sub-handle, started on the first batch
    bool workers_failed = false;          // thread creation failed once: shares run one after the other on the calling thread
    std::vector<hipEvent_t> ev_a, ev_b;   // per sub: around its share of the last batch (statistics)
    std::vector<double> stat_compute_ms;  // per device: its share of the last batch, on its stream (HIP events)
    double stat_gather_ms = 0.0, stat_total_ms = 0.0;   // the all-gather + final copies, and the whole call, host wall clock
    long gather_cap = 0;                  // doubles per device the gather buffers hold
    long gather_blk = 0;                  // block length of the last gathered batch / fit ...
    int gather_width = 2;                 // ... and doubles per evaluation in it (2: [loglik | info]; L + 4: a fitted delay)
    int gather_mode = 0;                  // GPCC_GATHER_*
    bool is_multi() const { return !subs.empty(); }
};

// the handle that holds the light curves and answers the single-matrix utilities (prediction, postb, dense exports)
static inline gpcc_handle_t primary(gpcc_handle_t h) { return (h && h->is_multi()) ? h->subs[0] : h; }

static int fail(gpcc_handle_t h, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) {
        static std::mutex mu;   // lanes of one handle may fail at the same time
        std::lock_guard<std::mutex> lk(mu);
        h->err = buf;
    }
    g_err = buf;
    return code;
}

#define HIPCHK(h, call)                                                                                  \
    do {                                                                                                 \
        hipError_t e_ = (call);                                                                          \
        if (e_ != hipSuccess)                                                                            \
            return fail(h, GPCC_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                        __LINE__);                                                                       \
    } while (0)

extern "C" int gpcc_version(void) { return GPCC_VERSION_NUMBER; }

// gpcc_build_info(): gpcc_buildinfo.hip (its own object: the record changes with every source, this file's object does not)

extern "C" const char *gpcc_last_error(gpcc_handle_t h) { return h ? h->err.c_str() : g_err.c_str(); }

// Every extern "C" entry that touches the GPU runs under a DeviceGuard: the calling thread's current device is
// switched to the handle's for the duration of the call and restored on return, so a host framework (Julia's
// AMDGPU.jl, torch) whose current device differs never finds its later allocations on another GPU.
struct DeviceGuard {
    int prev = -1, rc = 0;
    DeviceGuard(gpcc_handle_t h, int device)
    {
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess || n <= 0) {
            rc = fail(h, GPCC_ERR_HIP, "no HIP device available (%s); libgpcc_hip has no CPU fallback",
                      e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
            return;
        }
        if (device < 0 || device >= n) {
            rc = fail(h, GPCC_ERR_ARGUMENT, "device_id %d out of range [0,%d)", device, n);
            return;
        }
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        e = hipSetDevice(device);
        if (e != hipSuccess) rc = fail(h, GPCC_ERR_HIP, "hipSetDevice(%d) failed: %s", device, hipGetErrorString(e));
    }
    ~DeviceGuard()
    {
        if (prev >= 0) hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};
#define GPCC_ON_DEVICE(h, device)      \
    DeviceGuard guard_((h), (device)); \
    if (guard_.rc) return guard_.rc

static int multi_destroy(gpcc_handle_t h);
static int multi_grid_loglik(gpcc_handle_t h, int G, const double *delays, int iterations, int R, int C, double rhomin, double rhomax,
                            const double *cands, double *loglik_out, double *alpha_out, double *rho_out, int *info_out,
                            int *iterations_out, long long *stats_out);
static int multi_loglik_batch(gpcc_handle_t h, int M, const double *delays, const double *alpha, const double *rho,
                              double *loglik, int *info);

// a caller-supplied stream must live on the device the kernels are launched on (NULL = that device's default stream)
static int stream_on_device(gpcc_handle_t h, void *stream, int device)
{
    if (!stream) return 0;
    hipDevice_t sd = -1;
    hipError_t e = hipStreamGetDevice((hipStream_t)stream, &sd);
    if (e != hipSuccess) return fail(h, GPCC_ERR_ARGUMENT, "stream argument is not a valid hipStream_t (%s)", hipGetErrorString(e));
    if ((int)sd != device) return fail(h, GPCC_ERR_ARGUMENT, "stream belongs to device %d, the handle to device %d", (int)sd, device);
    return 0;
}

// ------------------------------------------------------------------------------------------
extern "C" int gpcc_create(gpcc_handle_t *out, int L, const int *Nl, const double *t, const double *y,
                           const double *sigma, int kernel_id, int marginalise_b, int precision, int device_id)
{
    if (!out) return fail(nullptr, GPCC_ERR_ARGUMENT, "handle pointer is NULL");
    *out = nullptr;
    if (L < 1 || L > GPCC_MAXL) return fail(nullptr, GPCC_ERR_ARGUMENT, "L=%d outside [1,%d]", L, GPCC_MAXL);
    if (!Nl || !t || !y || !sigma) return fail(nullptr, GPCC_ERR_ARGUMENT, "NULL light-curve pointer");
    if (kernel_id < 0 || kernel_id > 3) return fail(nullptr, GPCC_ERR_ARGUMENT, "unknown kernel_id %d", kernel_id);
    if (precision != GPCC_PRECISION_FP64 && precision != GPCC_PRECISION_FP32)
        return fail(nullptr, GPCC_ERR_ARGUMENT, "unknown precision %d", precision);
    long N = 0;
    for (int l = 0; l < L; ++l) {
        if (Nl[l] < 1 || (marginalise_b && Nl[l] < 2))
            return fail(nullptr, GPCC_ERR_ARGUMENT, "band %d has %d observations (need >= %d)", l + 1, Nl[l],
                        marginalise_b ? 2 : 1);
        N += Nl[l];
    }
    if (N > 65536) return fail(nullptr, GPCC_ERR_ARGUMENT, "N=%ld too large", N);
    GPCC_ON_DEVICE(nullptr, device_id);

    gpcc_handle_t h = new gpcc_handle_s();
    h->device = device_id;
    h->L = L;
    h->N = (int)N;
    h->nt = (int)((N + GPCC_TILE - 1) / GPCC_TILE);
    h->Np = h->nt * GPCC_TILE;
    h->kernel_id = kernel_id;
    h->mb = marginalise_b ? 1 : 0;
    h->precision = precision;
    h->woodbury = (precision == GPCC_PRECISION_FP32 && marginalise_b) ? 1 : 0;
    h->nrhs = h->woodbury ? L + 1 : 1;
    h->share_tiles = (L >= 2) ? Nl[0] / GPCC_TILE : 0;
    if (const char *e = getenv("GPCC_FP32_ASSEMBLE")) h->fp32_assemble = e[0] != '0';   // default of option "fp32_assemble" (A/B runs of the accuracy tools)
    if (const char *e = getenv("GPCC_SMALL_N")) h->small_n = e[0] != '0';   // default of option "small_n" (A/B runs, tests of the tile kernels at small N)
    {   // default group size: 256 evaluations resident (one per CU in the diagonal step: cfg5, N = 16384 in fp32, 0.55 GB per slot, runs
        // 88.1 evals/s with 256 slots against 84.8 with the 120 a 64 GiB cap allowed) wherever that fits 55 % of the device's TOTAL
        // memory -- not of what happens to be free: the group size selects the factorisation path, so the same handle must dispatch the
        // same way on every run whatever else occupies the GPU (if the memory is not there when the workspace is allocated, the
        // workspace shrinks and says so: ensure_workspace)
        double per_slot = ((double)h->nt * (h->nt + 1) / 2 + (precision ? h->nt : 1)) * GPCC_TILE_ELEMS * (precision ? 4.0 : 8.0) + 16.0 * h->Np * (L + 1) +
                          32.0 * h->Np + 4.0 * h->nt + 32.0;   // (+ the separable factors u, A, B, a of every point, the tile-row flags, the scale)
        if (precision)   // fp32 mode: diag(K) as assembled, the refinement's per-tile partial sums, the pivot-ratio statistics
            per_slot += 8.0 * h->Np + 8.0 * GPCC_MAXRHS * GPCC_MAXRHS * ((double)h->nt * (h->nt + 1) / 2) + 16.0;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || total_b == 0) total_b = (size_t)64 << 30;
        long cap = (long)(0.55 * (double)total_b / per_slot);
        if (cap < 8) cap = 8;
        if (h->slots_per_stream > cap) h->slots_per_stream = (int)(cap / 8 * 8);
        // the second stream doubles the workspace: only where both fit in half of the memory
        if (2.0 * h->slots_per_stream * per_slot > 0.5 * (double)total_b) h->streams = 1;
    }
    std::vector<double> ht(h->Np, 0.0), hs(h->Np, 0.0), hr(h->Np, 0.0), hy(h->Np, 0.0);
    std::vector<int> hb(h->Np, -1);
    long off = 0;
    for (int l = 0; l < L; ++l) {
        h->Nl[l] = Nl[l];
        // mu_b = mean(y_l), Sigma_b = 100 var(y_l) (n-1): marginaliseb.jl:92-94
        double s = 0.0;
        for (int n = 0; n < Nl[l]; ++n) s += y[off + n];
        const double mean = s / Nl[l];
        double v = 0.0;
        for (int n = 0; n < Nl[l]; ++n) { const double d = y[off + n] - mean; v += d * d; }
        h->mean_b[l] = mean;
        h->sigma_b[l] = marginalise_b ? 100.0 * (v / (Nl[l] - 1)) : 0.0;
        for (int n = 0; n < Nl[l]; ++n) {
            ht[off + n] = t[off + n];
            hs[off + n] = sigma[off + n] * sigma[off + n];  // Sobs = Diagonal(sigma.^2), :89
            hr[off + n] = y[off + n] - mean;                // Y - bbar (bbar = Q mu_b / Q b)
            hb[off + n] = l;
            hy[off + n] = y[off + n];
        }
        off += Nl[l];
    }
    {
        double lo = t[0], hi = t[0];
        for (long i = 1; i < N; ++i) { lo = std::min(lo, t[i]); hi = std::max(hi, t[i]); }
        h->tmid = 0.5 * (lo + hi);
    }
    h->resid_host.assign(hr.begin(), hr.begin() + N);
    h->t_host.assign(ht.begin(), ht.begin() + N);
    h->y_host.assign(hy.begin(), hy.begin() + N);
    h->sig2_host.assign(hs.begin(), hs.begin() + N);
    h->sigma_host.assign(sigma, sigma + N);
    h->band_host.assign(hb.begin(), hb.begin() + N);
    h->mixed_rows = (N % GPCC_TILE) != 0;   // (padding in the last tile row)
    for (long i = 0; i < N && !h->mixed_rows; ++i) h->mixed_rows = hb[i] != hb[i - i % GPCC_TILE];
#define CR(call)                                                                                             \
    do {                                                                                                     \
        hipError_t e_ = (call);                                                                              \
        if (e_ != hipSuccess) {                                                                              \
            int r_ = fail(nullptr, GPCC_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_));             \
            gpcc_destroy(h);                                                                                 \
            return r_;                                                                                       \
        }                                                                                                    \
    } while (0)
    const size_t nb = sizeof(double) * h->Np;
    CR(hipMalloc(&h->d_t, nb));
    CR(hipMalloc(&h->d_sig2, nb));
    CR(hipMalloc(&h->d_resid, nb));
    CR(hipMalloc(&h->d_yv, nb));
    CR(hipMalloc(&h->d_band, sizeof(int) * h->Np));
    CR(hipMemcpy(h->d_t, ht.data(), nb, hipMemcpyHostToDevice));
    CR(hipMemcpy(h->d_sig2, hs.data(), nb, hipMemcpyHostToDevice));
    CR(hipMemcpy(h->d_resid, hr.data(), nb, hipMemcpyHostToDevice));
    CR(hipMemcpy(h->d_yv, hy.data(), nb, hipMemcpyHostToDevice));
    CR(hipMemcpy(h->d_band, hb.data(), sizeof(int) * h->Np, hipMemcpyHostToDevice));
    CR(hipStreamCreateWithFlags(&h->main_stream, hipStreamNonBlocking));
    CR(hipEventCreateWithFlags(&h->ev_start, hipEventDisableTiming));
#undef CR
    *out = h;
    return 0;
}

static void free_workspace(gpcc_handle_t h)
{
    hipFree(h->d_tiles); hipFree(h->d_linv); hipFree(h->d_z); hipFree(h->d_w);
    hipFree(h->d_logdet); hipFree(h->d_quad); hipFree(h->d_info); hipFree(h->d_kdiag); hipFree(h->d_cond); hipFree(h->d_gpart);
    hipFree(h->d_sep); hipFree(h->d_seps); hipFree(h->d_sepflag);
    hipFree(h->d_chain_words); hipFree(h->d_ximg); hipFree(h->d_stepval); hipFree(h->d_chain_trace);
    h->d_chain_words = nullptr; h->d_ximg = h->d_stepval = nullptr; h->d_chain_trace = nullptr; h->chain_streams = 0;
    h->d_tiles = h->d_linv = h->d_z = h->d_w = h->d_logdet = h->d_quad = h->d_kdiag = h->d_cond = h->d_gpart = nullptr;
    h->d_sep = h->d_seps = nullptr;
    h->d_info = h->d_sepflag = nullptr;
    for (int s = 0; s < GPCC_MAX_STREAMS; ++s) {
        if (h->str[s]) { hipStreamDestroy(h->str[s]); h->str[s] = nullptr; }
        if (h->ev_done[s]) { hipEventDestroy(h->ev_done[s]); h->ev_done[s] = nullptr; }
        if (h->str2[s]) { hipStreamDestroy(h->str2[s]); h->str2[s] = nullptr; }
        if (h->ev_fork[s]) { hipEventDestroy(h->ev_fork[s]); h->ev_fork[s] = nullptr; }
        if (h->ev_join[s]) { hipEventDestroy(h->ev_join[s]); h->ev_join[s] = nullptr; }
    }
    h->ws_ready = false;
}

extern "C" int gpcc_destroy(gpcc_handle_t h)
{
    if (!h) return 0;
    if (h->is_multi()) return multi_destroy(h);
    DeviceGuard guard_(nullptr, h->device);   // restores the caller's current device
    hipDeviceSynchronize();
    for (auto &r : h->recs) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
    free_workspace(h);
    hipFree(h->d_t); hipFree(h->d_sig2); hipFree(h->d_resid); hipFree(h->d_band); hipFree(h->d_yv);
    hipFree(h->d_par); hipFree(h->d_out); hipFree(h->d_oinfo);
    for (auto &ln : h->lanes) ln.release();
    hipFree(h->d_cand);
    hipFree(h->d_ocond); hipFree(h->d_fb_idx); hipFree(h->d_fb_par); hipFree(h->d_fb_out); hipFree(h->d_fb_info);
    if (h->fb) gpcc_destroy(h->fb);
    if (h->main_stream) hipStreamDestroy(h->main_stream);
    if (h->ev_start) hipEventDestroy(h->ev_start);
    delete h;
    return 0;
}

extern "C" int gpcc_set_option(gpcc_handle_t h, const char *key, long v)
{
    if (!h || !key) return fail(h, GPCC_ERR_ARGUMENT, "NULL handle/key");
    if (h->is_multi()) {   // every device gets the same tunables
        for (gpcc_handle_t sub : h->subs) {
            const int rc = gpcc_set_option(sub, key, v);
            if (rc) return fail(h, rc, "%s", sub->err.c_str());
        }
        return 0;
    }
    if (!strcmp(key, "streams")) {
        if (v < 1 || v > GPCC_MAX_STREAMS) return fail(h, GPCC_ERR_ARGUMENT, "streams must be in [1,%d]", GPCC_MAX_STREAMS);
        h->streams = (int)v;
    } else if (!strcmp(key, "slots_per_stream")) {
        if (v < 1 || v > 4096) return fail(h, GPCC_ERR_ARGUMENT, "slots_per_stream must be in [1,4096]");
        h->slots_per_stream = (int)v;
    } else if (!strcmp(key, "right_looking_max")) {
        h->right_looking_max = (int)v;
    } else if (!strcmp(key, "fused_small_max")) {
        h->fused_small_max = (int)v;
    } else if (!strcmp(key, "chain_max")) {
        if (v < 0 || v > GPCC_CHAIN_MAX_EVALS) return fail(h, GPCC_ERR_ARGUMENT, "chain_max must be in [0,%d]", GPCC_CHAIN_MAX_EVALS);
        h->chain_max = (int)v;
    } else if (!strcmp(key, "chain_wide_work_max")) {
        if (v < 0) return fail(h, GPCC_ERR_ARGUMENT, "chain_wide_work_max must be >= 0");
        h->chain_wide_work_max = (long)v;
    } else if (!strcmp(key, "chain_work_max")) {
        if (v < 0) return fail(h, GPCC_ERR_ARGUMENT, "chain_work_max must be >= 0");
        h->chain_work_max = (long)v;
    } else if (!strcmp(key, "chain_workers_max")) {
        if (v < 0) return fail(h, GPCC_ERR_ARGUMENT, "chain_workers_max must be >= 0");
        h->chain_workers_max = (int)v;
    } else if (!strcmp(key, "chain_batch")) {
        if (v != 1 && v != 2 && v != 4 && v != GPCC_CHAIN_MAX_BATCH) return fail(h, GPCC_ERR_ARGUMENT, "chain_batch must be 1, 2, 4 or %d", GPCC_CHAIN_MAX_BATCH);
        h->chain_batch = (int)v;
    } else if (!strcmp(key, "chain_batch_min")) {
        if (v < 0) return fail(h, GPCC_ERR_ARGUMENT, "chain_batch_min must be >= 0");
        h->chain_batch_min = (long)v;
    } else if (!strcmp(key, "chain_quarters_max")) {
        if (v < 0 || v > GPCC_CHAIN_MAX_EVALS) return fail(h, GPCC_ERR_ARGUMENT, "chain_quarters_max must be in [0,%d]", GPCC_CHAIN_MAX_EVALS);
        h->chain_quarters_max = (int)v;
    } else if (!strcmp(key, "chain_helpers_max")) {
        if (v < 0 || v > GPCC_CHAIN_MAX_EVALS) return fail(h, GPCC_ERR_ARGUMENT, "chain_helpers_max must be in [0,%d]", GPCC_CHAIN_MAX_EVALS);
        h->chain_helpers_max = (int)v;
    } else if (!strcmp(key, "chain_trace")) {
        h->chain_trace = v != 0;
        if (h->chain_trace && !h->d_chain_trace) h->chain_streams = 0;   // (the buffers are rebuilt with a trace area on the next launch)
    } else if (!strcmp(key, "fused_solve")) {
        h->fused_solve = v != 0;
    } else if (!strcmp(key, "fused_solve_min_split")) {
        if (v < 1) return fail(h, GPCC_ERR_ARGUMENT, "fused_solve_min_split must be >= 1");
        h->fused_solve_min_split = (int)v;
    } else if (!strcmp(key, "fold_assembly")) {
        h->fold_assembly = v != 0;
    } else if (!strcmp(key, "fused_solve_min")) {
        h->fused_solve_min = (int)v;
        // (the halves of a split group follow: a caller who moves the threshold away from its default range -- to pin a path -- pins
        //  it for the halves too; "fused_solve_min_split" set afterwards overrides)
        h->fused_solve_min_split = (v < 64) ? (int)v : (v > 112 ? (int)v : 64);
    } else if (!strcmp(key, "small_n")) {
        h->small_n = v != 0;
    } else if (!strcmp(key, "small_wide_max")) {
        h->small_wide_max = (int)v;
    } else if (!strcmp(key, "hybrid_tail")) {
        h->hybrid_tail = v != 0;
    } else if (!strcmp(key, "hybrid_occ")) {
        h->hybrid_occ = (int)v;
    } else if (!strcmp(key, "split_max")) {
        h->split_max = (int)v;
    } else if (!strcmp(key, "split_nt_min")) {
        h->split_nt_min = (int)v;
    } else if (!strcmp(key, "split_small")) {
        h->split_small = v != 0;
    } else if (!strcmp(key, "split_min")) {
        if (v < 0) return fail(h, GPCC_ERR_ARGUMENT, "split_min must be >= 0");
        h->split_min = (int)v;
    } else if (!strcmp(key, "hybrid_mall_mb")) {
        if (v < 0 || v > 4096) return fail(h, GPCC_ERR_ARGUMENT, "hybrid_mall_mb must be in [0,4096]");
        h->hybrid_mall_mb = (int)v;
    } else if (!strcmp(key, "fit_speculate")) {
        h->fit_speculate = v != 0;
    } else if (!strcmp(key, "fit_device_unpack")) {
        h->fit_device_unpack = v != 0;
    } else if (!strcmp(key, "fit_threads")) {
        if (v < 0 || v > GPCC_MAX_LANES) return fail(h, GPCC_ERR_ARGUMENT, "fit_threads must be in [0,%d]", GPCC_MAX_LANES);
        h->fit_threads = (int)v;
    } else if (!strcmp(key, "fp32_guard")) {
        h->fp32_guard = v != 0;
    } else if (!strcmp(key, "fp32_chain")) {
        h->fp32_chain = v != 0;
    } else if (!strcmp(key, "fp32_refine")) {
        h->fp32_refine = v != 0;
    } else if (!strcmp(key, "fp32_assemble")) {
        h->fp32_assemble = v != 0;
    } else if (!strcmp(key, "shared_prefix")) {
        if (v < 0 || v > 2) return fail(h, GPCC_ERR_ARGUMENT, "shared_prefix must be 0, 1 or 2");
        h->shared_prefix = (int)v;
    } else {
        return fail(h, GPCC_ERR_ARGUMENT, "unknown option '%s'", key);
    }
    return 0;
}

extern "C" long gpcc_get_option(gpcc_handle_t h, const char *key)
{
    if (!h || !key) return -1;
    if (!strcmp(key, "n_devices")) return h->is_multi() ? (long)h->subs.size() : 1;
    if (!strcmp(key, "gather_mode")) return h->gather_mode;
    if (!strcmp(key, "gather_width")) return h->gather_width;
    h = primary(h);
    if (!strcmp(key, "streams")) return h->streams;
    if (!strcmp(key, "slots_per_stream")) return h->slots_per_stream;
    if (!strcmp(key, "workspace_streams")) return h->ws_ready ? h->ws_streams : h->streams;       // (differ from the options only if the
    if (!strcmp(key, "workspace_slots")) return h->ws_ready ? h->ws_slots : h->slots_per_stream;  //  memory was short at allocation)
    if (!strcmp(key, "right_looking_max")) return h->right_looking_max;
    if (!strcmp(key, "fused_small_max")) return h->fused_small_max;
    if (!strcmp(key, "chain_max")) return h->chain_max;
    if (!strcmp(key, "chain_count")) return h->chain_count;
    if (!strcmp(key, "chain_trace")) return h->chain_trace;
    if (!strcmp(key, "chain_helpers_max")) return h->chain_helpers_max;
    if (!strcmp(key, "chain_quarters_max")) return h->chain_quarters_max;
    if (!strcmp(key, "chain_work_max")) return h->chain_work_max;
    if (!strcmp(key, "chain_wide_work_max")) return h->chain_wide_work_max;
    if (!strcmp(key, "chain_workers_max")) return h->chain_workers_max;
    if (!strcmp(key, "chain_batch")) return h->chain_batch;
    if (!strcmp(key, "chain_batch_min")) return h->chain_batch_min;
    if (!strcmp(key, "chain_last_grid")) return h->chain_last_grid;
    if (!strcmp(key, "shared_prefix")) return h->shared_prefix;
    if (!strcmp(key, "share_tiles")) return h->share_tiles;
    if (!strcmp(key, "N")) return h->N;
    if (!strcmp(key, "Np")) return h->Np;
    if (!strcmp(key, "bytes_per_slot"))
        return (long)(((long)h->nt * (h->nt + 1) / 2 + (h->precision ? h->nt : 1)) * GPCC_TILE_ELEMS) * (h->precision ? 4 : 8) + 16L * h->Np * h->nrhs +
               32L * h->Np + 4L * h->nt + 32L +
               (h->precision ? 8L * h->Np + 8L * GPCC_MAXRHS * GPCC_MAXRHS * ((long)h->nt * (h->nt + 1) / 2) + 16L : 0L);
    if (!strcmp(key, "precision")) return h->precision;
    if (!strcmp(key, "fused_solve")) return h->fused_solve;
    if (!strcmp(key, "fused_solve_min")) return h->fused_solve_min;
    if (!strcmp(key, "fused_solve_min_split")) return h->fused_solve_min_split;
    if (!strcmp(key, "fold_assembly")) return h->fold_assembly;
    if (!strcmp(key, "small_n")) return h->small_n;
    if (!strcmp(key, "hybrid_tail")) return h->hybrid_tail;
    if (!strcmp(key, "hybrid_mall_mb")) return h->hybrid_mall_mb;
    if (!strcmp(key, "split_min")) return h->split_min;
    if (!strcmp(key, "split_max")) return h->split_max;
    if (!strcmp(key, "split_nt_min")) return h->split_nt_min;
    if (!strcmp(key, "split_small")) return h->split_small;
    if (!strcmp(key, "fit_speculate")) return h->fit_speculate;
    if (!strcmp(key, "fit_device_unpack")) return h->fit_device_unpack;
    if (!strcmp(key, "fit_threads")) return h->fit_threads;
    if (!strcmp(key, "small_n_max")) return GPCC_SMALLW_MAXN;
    if (!strcmp(key, "small_wide_max")) return h->small_wide_max;
    if (!strcmp(key, "small_n_active")) return (h->small_n && h->N <= GPCC_SMALLW_MAXN) ? 1 : 0;
    if (!strcmp(key, "small_n_count")) return h->small_count;
    if (!strcmp(key, "fp32_guard")) return h->fp32_guard;
    if (!strcmp(key, "fp32_chain")) return h->fp32_chain;
    if (!strcmp(key, "fp32_chain_count")) return h->fp32_chain_count;
    if (!strcmp(key, "fp32_refine")) return h->fp32_refine;
    if (!strcmp(key, "fp32_assemble")) return h->fp32_assemble;
    if (!strcmp(key, "fp32_guard_count")) return h->fb_count;
    return -1;
}

extern "C" int gpcc_get_constants(gpcc_handle_t h, double *mean_b, double *Sigma_b, double *resid)
{
    if (!h) return fail(h, GPCC_ERR_ARGUMENT, "NULL handle");
    h = primary(h);
    if (mean_b) memcpy(mean_b, h->mean_b, sizeof(double) * h->L);
    if (Sigma_b) memcpy(Sigma_b, h->sigma_b, sizeof(double) * h->L);
    if (resid) memcpy(resid, h->resid_host.data(), sizeof(double) * h->N);
    return 0;
}

extern "C" int gpcc_get_conditioning(gpcc_handle_t h, int M, double *out)
{
    if (!h || !out || M < 0) return fail(h, GPCC_ERR_ARGUMENT, "bad argument");
    h = primary(h);
    if (h->precision != GPCC_PRECISION_FP32) return fail(h, GPCC_ERR_ARGUMENT, "pivot ratios are tracked by fp32 handles only");
    if ((size_t)2 * M > h->cond_host.size()) return fail(h, GPCC_ERR_ARGUMENT, "the last batch had %ld evaluations", (long)(h->cond_host.size() / 2));
    memcpy(out, h->cond_host.data(), sizeof(double) * 2 * M);
    return 0;
}

// the diagonal kernel (and gpcc_small_step, which contains it) needs GPCC_DIAG_LDS_BYTES = 158.7 KiB of dynamic LDS (static_assert
// against the 160 KiB of a CU in gpcc_kernels.hip.h), the MFMA kernels 64 KiB, the quarter-tile solve 80 KiB (per device, idempotent)
static int set_kernel_attributes(gpcc_handle_t h)
{
    HIPCHK(h, hipFuncSetAttribute((const void *)gpcc_diag_factor<double>, hipFuncAttributeMaxDynamicSharedMemorySize, GPCC_DIAG_LDS_BYTES));
    HIPCHK(h, hipFuncSetAttribute((const void *)gpcc_diag_factor<float>, hipFuncAttributeMaxDynamicSharedMemorySize, GPCC_DIAG_LDS_BYTES));
    HIPCHK(h, hipFuncSetAttribute((const void *)gpcc_small_step<double>, hipFuncAttributeMaxDynamicSharedMemorySize, GPCC_DIAG_LDS_BYTES));
    HIPCHK(h, hipFuncSetAttribute((const void *)gpcc_syrk_diag<double>, hipFuncAttributeMaxDynamicSharedMemorySize, GPCC_DIAG_LDS_BYTES));
    HIPCHK(h, hipFuncSetAttribute((const void *)gpcc_syrk_diag<float>, hipFuncAttributeMaxDynamicSharedMemorySize, GPCC_DIAG_LDS_BYTES));
    HIPCHK(h, hipFuncSetAttribute((const void *)gpcc_small_step<float>, hipFuncAttributeMaxDynamicSharedMemorySize, GPCC_DIAG_LDS_BYTES));
    HIPCHK(h, hipFuncSetAttribute((const void *)gpcc_panel_update<double, false>, hipFuncAttributeMaxDynamicSharedMemorySize, GPCC_GEMM_LDS_BYTES));
    HIPCHK(h, hipFuncSetAttribute((const void *)gpcc_update_solve<double, false>, hipFuncAttributeMaxDynamicSharedMemorySize, GPCC_UPSOLVE_LDS_BYTES));
    HIPCHK(h, hipFuncSetAttribute((const void *)gpcc_update_solve<float, false>, hipFuncAttributeMaxDynamicSharedMemorySize, GPCC_UPSOLVE_LDS_BYTES));
    HIPCHK(h, hipFuncSetAttribute((const void *)gpcc_update_solve<double, true>, hipFuncAttributeMaxDynamicSharedMemorySize, GPCC_UPSOLVE_LDS_BYTES));
    HIPCHK(h, hipFuncSetAttribute((const void *)gpcc_update_solve<float, true>, hipFuncAttributeMaxDynamicSharedMemorySize, GPCC_UPSOLVE_LDS_BYTES));
    HIPCHK(h, hipFuncSetAttribute((const void *)gpcc_panel_update<double, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, GPCC_GEMM_LDS_BYTES));
    HIPCHK(h, hipFuncSetAttribute((const void *)gpcc_panel_update<double, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, GPCC_GEMM_LDS_BYTES));
    HIPCHK(h, hipFuncSetAttribute((const void *)gpcc_panel_update<float, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, GPCC_GEMM_LDS_BYTES));
    HIPCHK(h, hipFuncSetAttribute((const void *)gpcc_panel_update<float, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, GPCC_GEMM_LDS_BYTES));
    HIPCHK(h, hipFuncSetAttribute((const void *)gpcc_panel_update<double, true>, hipFuncAttributeMaxDynamicSharedMemorySize, GPCC_GEMM_LDS_BYTES));
    HIPCHK(h, hipFuncSetAttribute((const void *)gpcc_panel_update<float, false>, hipFuncAttributeMaxDynamicSharedMemorySize, GPCC_GEMM_LDS_BYTES));
    HIPCHK(h, hipFuncSetAttribute((const void *)gpcc_panel_update<float, true>, hipFuncAttributeMaxDynamicSharedMemorySize, GPCC_GEMM_LDS_BYTES));
    HIPCHK(h, hipFuncSetAttribute((const void *)gpcc_panel_trsm<double>, hipFuncAttributeMaxDynamicSharedMemorySize, GPCC_GEMM_LDS_BYTES));
    HIPCHK(h, hipFuncSetAttribute((const void *)gpcc_panel_trsm<float>, hipFuncAttributeMaxDynamicSharedMemorySize, GPCC_GEMM_LDS_BYTES));
    HIPCHK(h, hipFuncSetAttribute((const void *)gpcc_panel_trsm_rows<double>, hipFuncAttributeMaxDynamicSharedMemorySize, GPCC_TRSM_ROWS_LDS_BYTES));
    HIPCHK(h, hipFuncSetAttribute((const void *)gpcc_panel_trsm_rows<float>, hipFuncAttributeMaxDynamicSharedMemorySize, GPCC_TRSM_ROWS_LDS_BYTES));
    HIPCHK(h, gpcc_chain_configure());
    return 0;
}

// the buffers of the persistent few-evaluation launch (gpcc_chain.hip.h), one region per workspace stream; built on first use
static int ensure_chain(gpcc_handle_t h)
{
    if (h->d_chain_words && h->chain_streams == h->ws_streams) return 0;
    HIPCHK(h, hipDeviceSynchronize());
    hipFree(h->d_chain_words); hipFree(h->d_ximg); hipFree(h->d_stepval); hipFree(h->d_chain_trace);
    h->d_chain_words = nullptr; h->d_ximg = h->d_stepval = nullptr; h->d_chain_trace = nullptr; h->chain_streams = 0;
    const long ntiles = (long)h->nt * (h->nt + 1) / 2;
    h->chain_qbase = 16 + 2 * ((h->nt + 15) / 16 * 16);   // abort word + trace counter, then the urgent and the bulk queue's counter per step
    h->chain_ev_words = (int)((10L * h->nt + 2 * ntiles + 3) / 4 * 4);   // xrow, l7, colflag[8] per step; lcnt, ver per tile
    h->chain_region_words = ((long)h->chain_qbase + (long)GPCC_CHAIN_MAX_EVALS * h->chain_ev_words + 63) / 64 * 64;
    const long S = h->ws_streams, E = GPCC_CHAIN_MAX_EVALS;
    HIPCHK(h, hipMalloc(&h->d_chain_words, sizeof(unsigned) * h->chain_region_words * S));
    HIPCHK(h, hipMalloc(&h->d_ximg, sizeof(double) * S * E * h->nt * GPCC_XIMG_STRIDE));
    HIPCHK(h, hipMalloc(&h->d_stepval, sizeof(double) * S * E * h->nt * GPCC_CHAIN_STEPVALS));
    if (h->chain_trace) {
        const size_t tw = (size_t)S * E * h->nt * GPCC_CHAIN_TRACE_WORDS + (size_t)S * 4 * GPCC_CHAIN_WTRACE_CAP;   // chain stamps, then the workers' jobs
        HIPCHK(h, hipMalloc(&h->d_chain_trace, sizeof(unsigned long long) * tw));
        HIPCHK(h, hipMemset(h->d_chain_trace, 0, sizeof(unsigned long long) * tw));
    }
    if (!h->n_cus) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess || n <= 0) n = 256;
        h->n_cus = n;
    }
    h->chain_streams = h->ws_streams;
    return 0;
}

// a group of a few evaluations (one objective(alpha, rho)) on an fp64 handle's own workspace: the persistent launch
// which group sizes the persistent launch wins at (profiles/r05/latency_small_batches.log, chain_13_to_32_evaluations_ab.log): up to 12
// evaluations while evaluations x (N/128)^2 <= chain_work_max; 13 .. chain_max = 32 while <= chain_wide_work_max (32 evaluations at
// N <= 512, 28 at N = 768, 16 at N = 1024: there the alternative -- two halves on two streams -- is slower even when the halves overlap,
// and they only overlap when the runtime happens to map the two streams onto different hardware queues: 0.45 against 0.52 / 0.89 ms
// for 13 evaluations at N = 1024, 0.25 against 0.34 for 32 at N = 512)
static bool chain_policy(gpcc_handle_t h, int cnt, int nt)
{
    if (h->chain_max <= 0 || cnt > h->chain_max || cnt > GPCC_CHAIN_MAX_EVALS || nt <= 1) return false;
    // right_looking_max below its default switches the few-evaluation paths off for larger groups -- this one included (0: every group left-looking)
    if (cnt > h->right_looking_max && h->right_looking_max < GPCC_RIGHT_LOOKING_MAX) return false;
    return (long)cnt * nt * nt <= (cnt <= 12 ? h->chain_work_max : h->chain_wide_work_max);
}

static bool takes_chain(gpcc_handle_t h, const GpccCtx &c, int cnt, int concurrent = 1)
{
    // (a half of a split group runs beside the other half on a second stream: the persistent launch has ONE set of flag words per
    //  workspace stream, and wants the chip to itself)
    return concurrent <= 1 && chain_policy(h, cnt, c.nt) && c.nt_fact == c.nt &&
           !c.share_p && !c.store_l && c.nrhs == 1 && !c.woodbury && h->precision == GPCC_PRECISION_FP64 && c.tiles == (void *)h->d_tiles &&
           h->d_chain_words != nullptr && h->chain_streams == h->ws_streams;
}

// the buffers of the persistent launch for the group that starts at slot g.slot0 (one region per workspace stream)
static GpccChainArgs chain_args(gpcc_handle_t h, const GpccCtx &c, const GpccGroup &g)
{
    const int si = g.slot0 / (h->ws_slots > 0 ? h->ws_slots : 1);
    GpccChainArgs a;
    a.words = h->d_chain_words + (long)si * h->chain_region_words;
    a.ximg = h->d_ximg + (long)si * GPCC_CHAIN_MAX_EVALS * c.nt * GPCC_XIMG_STRIDE;
    a.stepval = h->d_stepval + (long)si * GPCC_CHAIN_MAX_EVALS * c.nt * GPCC_CHAIN_STEPVALS;
    a.trace = h->d_chain_trace ? h->d_chain_trace + (long)si * GPCC_CHAIN_MAX_EVALS * c.nt * GPCC_CHAIN_TRACE_WORDS : nullptr;
    a.wtrace = h->d_chain_trace ? h->d_chain_trace + (size_t)h->ws_streams * GPCC_CHAIN_MAX_EVALS * c.nt * GPCC_CHAIN_TRACE_WORDS + (size_t)si * 4 * GPCC_CHAIN_WTRACE_CAP : nullptr;
    a.wtrace_cap = GPCC_CHAIN_WTRACE_CAP;
    a.ev_words = h->chain_ev_words;
    a.qbase = h->chain_qbase;
    a.helpers = (g.cnt <= h->chain_helpers_max) ? 1 : 0;
    a.quarters = (g.cnt <= h->chain_quarters_max) ? 1 : 0;
    // (worker-bound launches only: a group whose chain is the bound gains nothing from longer jobs and loses a little balance --
    //  profiles/r05/chain_batch_ab.log; the results are the same bits either way)
    a.batch = ((long)g.cnt * c.nt * c.nt * c.nt >= h->chain_batch_min) ? h->chain_batch : 1;
    return a;
}

// ------------------------------------------------------------------------------------------
static int ensure_workspace(gpcc_handle_t h)
{
    if (h->ws_ready && h->ws_req_streams == h->streams && h->ws_req_slots == h->slots_per_stream) return 0;
    HIPCHK(h, hipDeviceSynchronize());
    h->slot_stride = ((long)h->nt * (h->nt + 1) / 2) * GPCC_TILE_ELEMS;
    const size_t esz = h->precision ? sizeof(float) : sizeof(double);
    int run_streams = h->streams, run_slots = h->slots_per_stream;   // what the workspace really gets: the OPTIONS stay what the caller set
    for (;;) {
        // The group size was chosen from the device's memory; if somebody else (another handle, another process sharing the GPU) holds
        // it, run smaller groups rather than fail: first one stream, then half the slots.  Reported, not silent: "workspace_streams" /
        // "workspace_slots" say what is in use and gpcc_last_error carries a note.
        free_workspace(h);
        const long slots = (long)run_streams * run_slots;
        hipError_t e = hipMalloc(&h->d_tiles, esz * h->slot_stride * slots);
        if (e == hipSuccess) e = hipMalloc(&h->d_linv, esz * GPCC_TILE_ELEMS * slots * (h->precision == GPCC_PRECISION_FP32 ? h->nt : 1));
        if (e == hipSuccess) e = hipMalloc(&h->d_z, sizeof(double) * h->Np * h->nrhs * slots);
        if (e == hipSuccess) e = hipMalloc(&h->d_w, sizeof(double) * h->Np * h->nrhs * slots);
        if (e == hipSuccess) e = hipMalloc(&h->d_logdet, sizeof(double) * slots);
        if (e == hipSuccess) e = hipMalloc(&h->d_quad, sizeof(double) * GPCC_MAXRHS * GPCC_MAXRHS * slots);
        if (e == hipSuccess) e = hipMalloc(&h->d_info, sizeof(int) * slots);
        if (e == hipSuccess) e = hipMalloc(&h->d_sep, sizeof(double) * 4 * h->Np * slots);
        if (e == hipSuccess) e = hipMalloc(&h->d_seps, sizeof(double) * 4 * slots);
        if (e == hipSuccess) e = hipMalloc(&h->d_sepflag, sizeof(int) * h->nt * slots);
        if (e == hipSuccess && h->precision == GPCC_PRECISION_FP32) {
            e = hipMalloc(&h->d_kdiag, sizeof(double) * h->Np * slots);
            if (e == hipSuccess) e = hipMalloc(&h->d_cond, sizeof(double) * 2 * slots);
            if (e == hipSuccess) e = hipMalloc(&h->d_gpart, sizeof(double) * GPCC_MAXRHS * GPCC_MAXRHS * ((long)h->nt * (h->nt + 1) / 2) * slots);
        }
        if (e == hipSuccess) break;
        (void)hipGetLastError();
        free_workspace(h);
        if (e != hipErrorOutOfMemory || (run_streams == 1 && run_slots <= 1))
            return fail(h, GPCC_ERR_HIP, "workspace of %ld slots: %s", slots, hipGetErrorString(e));
        if (run_streams > 1) run_streams = 1;
        else run_slots = run_slots > 16 ? (run_slots / 2 + 7) / 8 * 8 : run_slots / 2;
    }
    for (int s = 0; s < run_streams; ++s) {
        HIPCHK(h, hipStreamCreateWithFlags(&h->str[s], hipStreamNonBlocking));
        HIPCHK(h, hipEventCreateWithFlags(&h->ev_done[s], hipEventDisableTiming));
        HIPCHK(h, hipStreamCreateWithFlags(&h->str2[s], hipStreamNonBlocking));
        HIPCHK(h, hipEventCreateWithFlags(&h->ev_fork[s], hipEventDisableTiming));
        HIPCHK(h, hipEventCreateWithFlags(&h->ev_join[s], hipEventDisableTiming));
    }
    { int rc_ = set_kernel_attributes(h); if (rc_) return rc_; }
    h->ws_streams = run_streams;
    h->ws_slots = run_slots;
    h->ws_req_streams = h->streams;
    h->ws_req_slots = h->slots_per_stream;
    h->ws_ready = true;
    if (run_streams != h->streams || run_slots != h->slots_per_stream) {   // a note, not an error: the call goes on with smaller groups
        char buf[256];
        snprintf(buf, sizeof buf, "note: the device's memory did not hold %d x %d slots; the workspace runs %d x %d (options unchanged)",
                 h->streams, h->slots_per_stream, run_streams, run_slots);
        h->err = buf;
    }
    return 0;
}

static GpccCtx make_ctx(gpcc_handle_t h)
{
    GpccCtx c;
    c.tiles = h->d_tiles; c.linv = h->d_linv; c.z = h->d_z; c.w = h->d_w;
    c.logdet = h->d_logdet; c.gram = h->d_quad; c.info = h->d_info;
    c.kdiag = h->d_kdiag; c.cond = h->d_cond; c.gpart = h->d_gpart;
    c.linv_keep = (h->precision == GPCC_PRECISION_FP32) ? 1 : 0;
    c.t = h->d_t; c.sig2 = h->d_sig2; c.resid = h->d_resid; c.band = h->d_band; c.yv = h->d_yv;
    c.tmid = h->tmid;
    c.sep = h->d_sep; c.seps = h->d_seps; c.sepflag = h->d_sepflag; c.fold = 0; c.fold_mixed = 0;
    for (int l = 0; l < GPCC_MAXL; ++l) c.sigma_b[l] = (l < h->L) ? h->sigma_b[l] : 0.0;
    c.slot_stride = h->slot_stride;
    c.L = h->L; c.N = h->N; c.Np = h->Np; c.nt = h->nt; c.kernel_id = h->kernel_id; c.marginalise_b = h->mb;
    c.nt_fact = h->nt;
    c.nrhs = h->nrhs; c.woodbury = h->woodbury; c.share_p = 0; c.store_l = 0;
    c.chain_words = nullptr; c.chain_qbase = 0; c.chain_ev_words = 0;
    c.asm32 = (h->precision == GPCC_PRECISION_FP32 && h->fp32_assemble && h->fp32_refine) ? 1 : 0;
    return c;
}

struct ProfScope {
    gpcc_handle_t h; int which; hipStream_t s; hipEvent_t a = nullptr, b = nullptr;
    ProfScope(gpcc_handle_t h_, int w, hipStream_t s_) : h(h_), which(w), s(s_)
    {
        if (h->prof) { hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a, s); }
    }
    ~ProfScope()
    {
        if (h->prof) { hipEventRecord(b, s); h->recs.push_back({which, a, b}); }
    }
};

template <bool EXT, typename T>
static void launch_assemble_t(const GpccCtx &c, const GpccGroup &g, hipStream_t s)
{
    dim3 grid(c.nt * c.nt, g.cnt, c.chain_words ? 4 : 1);   // (in front of the persistent launch: a row quarter per workgroup)
    switch (c.kernel_id) {
    case 0: gpcc_assemble_tiles<0, EXT, T><<<grid, 256, 0, s>>>(c, g); break;
    case 1: gpcc_assemble_tiles<1, EXT, T><<<grid, 256, 0, s>>>(c, g); break;
    case 2: gpcc_assemble_tiles<2, EXT, T><<<grid, 256, 0, s>>>(c, g); break;
    default: gpcc_assemble_tiles<3, EXT, T><<<grid, 256, 0, s>>>(c, g); break;
    }
}

static void launch_assemble(gpcc_handle_t h, const GpccCtx &c, const GpccGroup &g, hipStream_t s, bool ext, bool f32)
{
    ProfScope p(h, GPCC_PROF_ASSEMBLE, s);
    if (f32) { if (ext) launch_assemble_t<true, float>(c, g, s); else launch_assemble_t<false, float>(c, g, s); }
    else { if (ext) launch_assemble_t<true, double>(c, g, s); else launch_assemble_t<false, double>(c, g, s); }
}

static int enqueue_factor(gpcc_handle_t h, const GpccCtx &c, const GpccGroup &g, hipStream_t s, bool f32, int concurrent);

// the group runs left-looking with the panel solve inside the update (gpcc_syrk_diag + gpcc_update_solve)
static bool takes_fused_solve(gpcc_handle_t h, const GpccCtx &c, int cnt, int concurrent)
{
    if (takes_chain(h, c, cnt, concurrent)) return false;   // (one path per group: the persistent launch wants its tiles assembled)
    const bool right = (cnt <= h->right_looking_max) && (c.nt_fact == c.nt) && !c.share_p;
    // (the halves of a split group hide each other's serial diagonal-tile loop: the fused path pays from 64 evaluations per half on)
    const int min_cnt = (concurrent >= 2) ? h->fused_solve_min_split : h->fused_solve_min;
    return !right && !c.share_p && c.nt_fact == c.nt && h->fused_solve && cnt >= min_cnt && !c.store_l;
}

static int enqueue_group(gpcc_handle_t h, const GpccCtx &c_in, const GpccGroup &g, hipStream_t s, bool factor = true,
                         bool ext = false, int f32 = -1, int concurrent = 1)
{
    const bool single = (f32 < 0) ? (h->precision == GPCC_PRECISION_FP32) : (f32 != 0);
    GpccCtx c = c_in;
    // fold: on the fused path every off-diagonal tile (I,k) is read exactly once, by the job that updates and solves it -- which can
    // evaluate the elements itself (gpcc_update_solve); only what the flags of gpcc_sep_points exclude is still assembled
    // (rbf is not separable: its tiles are folded with the direct evaluation -- or the fp32 one --, through the general instantiations)
    c.fold = 0;
    if (factor && !ext && h->fold_assembly && c.sep && c.nt > 1 && !c.share_p && c.nt_fact == c.nt && !c.store_l) {
        const bool right = g.cnt <= h->right_looking_max;
        if (takes_fused_solve(h, c, g.cnt, concurrent)) c.fold = 1;
        else if (!((right && g.cnt <= h->fused_small_max) || takes_chain(h, c, g.cnt, concurrent))) c.fold = 2;   // (not the few-evaluation paths)
        // tile rows that straddle two bands or hold padding: the MIXED instantiations (a handle without such rows keeps the leaner ones)
        c.fold_mixed = (c.fold && (h->mixed_rows || c.kernel_id == 1)) ? 1 : 0;
    }
    if (c.fold) {
        ProfScope p(h, GPCC_PROF_ASSEMBLE, s);
        dim3 grid(c.nt, g.cnt);
        switch (c.kernel_id) {
        case 0: gpcc_sep_points<0><<<grid, GPCC_TILE, 0, s>>>(c, g); break;
        case 1: gpcc_sep_points<1><<<grid, GPCC_TILE, 0, s>>>(c, g); break;
        case 2: gpcc_sep_points<2><<<grid, GPCC_TILE, 0, s>>>(c, g); break;
        default: gpcc_sep_points<3><<<grid, GPCC_TILE, 0, s>>>(c, g); break;
        }
    }
    if (factor && !ext && !single && takes_chain(h, c, g.cnt, concurrent)) {   // the assembly zeroes the flag words of the persistent launch that follows
        const GpccChainArgs ca = chain_args(h, c, g);
        c.chain_words = ca.words; c.chain_qbase = ca.qbase; c.chain_ev_words = ca.ev_words;
    }
    launch_assemble(h, c, g, s, ext, single);
    if (!factor) return 0;
    int rc = enqueue_factor(h, c, g, s, single, concurrent);
    if (rc || !single || !h->fp32_refine || !c.gpart || c.linv_keep != 1) return rc;
    {   // fp32: refine the quadratic forms in fp64 -- backward solve, X' K0 X on the fly, final arithmetic
        ProfScope pr(h, GPCC_PROF_REFINE, s);
        gpcc_back_solve<float><<<g.cnt, 512, 0, s>>>(c, g);
        dim3 grid(c.nt * c.nt, g.cnt);
#define GPCC_REFINE_NR(KID)                                                                        \
    switch (c.nrhs) {                                                                              \
    case 1: gpcc_refine_partials<KID, 1><<<grid, 256, 0, s>>>(c, g); break;                        \
    case 3: gpcc_refine_partials<KID, 3><<<grid, 256, 0, s>>>(c, g); break;                        \
    case 4: gpcc_refine_partials<KID, 4><<<grid, 256, 0, s>>>(c, g); break;                        \
    default: gpcc_refine_partials<KID, 0><<<grid, 256, 0, s>>>(c, g); break;                       \
    }
        switch (c.kernel_id) {
        case 0: GPCC_REFINE_NR(0) break;
        case 1: GPCC_REFINE_NR(1) break;
        case 2: GPCC_REFINE_NR(2) break;
        default: GPCC_REFINE_NR(3) break;
        }
#undef GPCC_REFINE_NR
        gpcc_refine_finish<<<g.cnt, 256, 0, s>>>(c, g);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(h, GPCC_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
    return 0;
}

template <typename T>
static void enqueue_factor_t(gpcc_handle_t h, const GpccCtx &c, const GpccGroup &g_in, hipStream_t s, int concurrent)
{
    // a handful of evaluations (one objective(alpha, rho), predictTest, postb): spread every evaluation's jobs over
    // the 8 XCDs; the usual mapping keeps an evaluation on the XCD blockIdx % 8 selects, 1/8 of the chip for one matrix
    GpccGroup g = g_in;
    g.spread = (g.cnt < 8 && !c.share_p) ? 1 : 0;
    const int cnt8 = g.spread ? g.cnt : 8 * ((g.cnt + 7) / 8);
    // small groups (the single objective(alpha, rho) call): right-looking, many short jobs per step
    const bool right = (g.cnt <= h->right_looking_max) && (c.nt_fact == c.nt) && !c.share_p;
    const int p = c.share_p;   // shared prefix: steps k < p only involve the leader's rows < p and everyone's rows >= p
    if (sizeof(T) == 8 && takes_chain(h, c, g.cnt, concurrent)) {
        // a few evaluations on an fp64 handle: ONE persistent launch -- two chain workgroups per evaluation, everybody else pulls jobs
        // (gpcc_chain.hip.h)
        ProfScope pr(h, GPCC_PROF_SMALL_STEP, s);
        const GpccChainArgs a = chain_args(h, c, g);   // (its flag words were zeroed by the assembly launch in front: GpccCtx::chain_words)
        const int dedicated = (a.helpers ? 6 : 2) * g.cnt;                 // chain roles (+ the four solve helpers) per evaluation
        const int ncb = (a.helpers ? 48 : 16) * ((g.cnt + 7) / 8);         // their block range
        long workers = (long)g.cnt * gpcc_chain_jobs(c.nt - 1);   // the widest step; more workgroups than that would only spin
        long room = (long)h->n_cus - dedicated;
        if (h->chain_workers_max > 0 && room > h->chain_workers_max) room = h->chain_workers_max;
        if (workers > room) workers = room;
        if (workers < 1) workers = 1;
        const long grid = (dedicated + workers > ncb) ? dedicated + workers : ncb;   // (blocks of the dedicated range without a role work too)
        gpcc_chain_launch(c, g, a, (unsigned)grid, s);
        h->chain_last_grid = grid;
        h->chain_count += g.cnt;
        return;
    }
    if (right && g.cnt <= h->fused_small_max && c.nt > 1) {
        // a few evaluations (the objective(alpha, rho) call site): diag(0), then per step the panel solve and ONE launch
        // that holds the trailing update of step k AND the diagonal step k+1 (gpcc_small_step): 2 nt - 1 launches
        {
            ProfScope pr(h, GPCC_PROF_DIAG, s);
            gpcc_diag_factor<T><<<g.cnt, GPCC_DIAG_THREADS, GPCC_DIAG_LDS_BYTES, s>>>(c, g, 0);
        }
        for (int k = 0; k < c.nt - 1; ++k) {
            {
                ProfScope pr(h, GPCC_PROF_TRSM, s);
                gpcc_panel_trsm_rows<T><<<g.cnt * (c.nt - k - 1) * 4, 512, GPCC_TRSM_ROWS_LDS_BYTES, s>>>(c, g, k);
            }
            ProfScope pr(h, GPCC_PROF_SMALL_STEP, s);
            const int n = c.nt - k - 1;
            gpcc_small_step<T><<<g.cnt * (n * (n + 1) / 2), 512, GPCC_DIAG_LDS_BYTES, s>>>(c, g, k);
        }
        return;
    }
    if (takes_fused_solve(h, c, g.cnt, concurrent)) {
        // left-looking, the panel solve inside the update (gpcc_update_solve): per step the diagonal tile first
        // (gpcc_syrk_diag: lower-triangle update + diagonal step in one workgroup per evaluation), then the rest of column k
        for (int k = 0; k < c.nt; ++k) {
            {
                ProfScope pr(h, GPCC_PROF_DIAG, s);
                gpcc_syrk_diag<T><<<g.cnt, 512, GPCC_DIAG_LDS_BYTES, s>>>(c, g, k);
            }
            if (k < c.nt - 1) {
                ProfScope pr(h, GPCC_PROF_PANEL_UPDATE, s);
                if (c.fold_mixed) gpcc_update_solve<T, true><<<cnt8 * (c.nt - k - 1), GPCC_GEMM_THREADS, GPCC_UPSOLVE_LDS_BYTES, s>>>(c, g, k);
                else gpcc_update_solve<T, false><<<cnt8 * (c.nt - k - 1), GPCC_GEMM_THREADS, GPCC_UPSOLVE_LDS_BYTES, s>>>(c, g, k);
            }
        }
        return;
    }
    // Hybrid tail (plain left-looking groups of a few dozen evaluations): a left-looking step is cnt (nt - k) jobs of k tile
    // products, so the LAST steps are a few long jobs on a mostly idle chip (32 evaluations at N = 4096: steps 25-31 run at
    // 75 ... 12 % of the 256 CUs, profiles/r03/midsize_update_per_step_plain_vs_splitk.log).  From step ks on the group therefore
    // runs right-looking: ONE catch-up launch gives every trailing tile (I,J), I >= J >= ks, its whole sum over the finished
    // columns < ks (cnt n(n+1)/2 equal jobs that stream the rows in lock-step through L2), then each step updates the trailing
    // matrix with its one new column -- many short jobs, affordable because ks is chosen so that the trailing matrices of the
    // whole group stay in the Infinity Cache.  Measured and dropped for the same purpose: split-K and stream-K decompositions
    // of the update with a last-arriver reduction (bitwise deterministic; no gain: DESIGN.md 4.3).
    int ks = right ? 0 : c.nt_fact;   // first right-looking step
    if (!right && !p && !g.spread && h->hybrid_tail && c.nt_fact == c.nt && c.nt >= 6) {
        const double tile_mb = GPCC_TILE_ELEMS * sizeof(T) / 1048576.0;
        int n = 0;
        while ((double)g.cnt * (n + 1) * (n + 2) / 2 * tile_mb * concurrent <= h->hybrid_mall_mb) ++n;   // trailing size the cache holds (shared by the halves of a split group)
        const int nocc = (h->hybrid_occ + g.cnt - 1) / g.cnt;                                // steps with fewer left-looking jobs than that
        if (n > nocc) n = nocc;
        if (n > c.nt - 1) n = c.nt - 1;                                                      // (right-looking from step 1 on)
        if (n >= 2) ks = c.nt - n;
    }
    auto launch_diag = [&](int k, hipStream_t st) {
        ProfScope pr(h, GPCC_PROF_DIAG, st);
        gpcc_diag_factor<T><<<(k < p) ? 1 : g.cnt, GPCC_DIAG_THREADS, GPCC_DIAG_LDS_BYTES, st>>>(c, g, k);
    };
    auto launch_trsm = [&](int k, hipStream_t st) {
        ProfScope pr(h, GPCC_PROF_TRSM, st);
        const int grid = (k < p) ? cnt8 * (c.nt - p) + (p - k - 1) : cnt8 * (c.nt - k - 1);
        if (grid > 0) gpcc_panel_trsm<T><<<grid, GPCC_GEMM_THREADS, GPCC_GEMM_LDS_BYTES, st>>>(c, g, k);
    };
    auto launch_right = [&](int grid, hipStream_t st, int k, int ktiles, int kcol) {   // (only a launch whose K loop starts at column 0 folds)
        if (c.fold_mixed && kcol == 0) gpcc_panel_update<T, true, true><<<grid, GPCC_GEMM_THREADS, GPCC_GEMM_LDS_BYTES, st>>>(c, g, k, ktiles, kcol);
        else gpcc_panel_update<T, true><<<grid, GPCC_GEMM_THREADS, GPCC_GEMM_LDS_BYTES, st>>>(c, g, k, ktiles, kcol);
    };
    for (int k = 0; k < c.nt_fact; ++k) {
        const bool rstep = k >= ks;
        if (k > 0 && !rstep) {
            ProfScope pr(h, GPCC_PROF_PANEL_UPDATE, s);
            const int grid = (k < p) ? cnt8 * (c.nt - p) + (p - k) : cnt8 * (c.nt - k);
            if (c.fold_mixed) gpcc_panel_update<T, false, true><<<grid, GPCC_GEMM_THREADS, GPCC_GEMM_LDS_BYTES, s>>>(c, g, k, k, 0);
            else gpcc_panel_update<T, false><<<grid, GPCC_GEMM_THREADS, GPCC_GEMM_LDS_BYTES, s>>>(c, g, k, k, 0);
        }
        if (k > 0 && k == ks) {   // catch-up: all trailing tiles (I,J), I >= J >= ks, minus their sums over columns 0 .. ks-1
            ProfScope pr(h, GPCC_PROF_PANEL_UPDATE, s);
            const int n = c.nt - k;
            launch_right(cnt8 * (n * (n + 1) / 2), s, k - 1, k, 0);
        }
        launch_diag(k, s);
        if (k < c.nt - 1) launch_trsm(k, s);
        if (rstep && k < c.nt - 1) {
            const int n = c.nt - k - 1;
            {
                ProfScope pr(h, GPCC_PROF_PANEL_UPDATE, s);
                launch_right(cnt8 * (n * (n + 1) / 2), s, k, 1, k);
            }
        }
    }
    // augmented systems: Schur complement of the rows beyond the factorised columns,
    // S = C - V^T V with V = L^-1 [cross block]  
    for (int k = c.nt_fact; k < c.nt; ++k)
        gpcc_panel_update<T, false><<<cnt8 * (c.nt - k), GPCC_GEMM_THREADS, GPCC_GEMM_LDS_BYTES, s>>>(c, g, k, c.nt_fact, 0);
}

// left-looking blocked Cholesky + fused forward solve (+ Schur complement of non-factorised rows)
static int enqueue_factor(gpcc_handle_t h, const GpccCtx &c, const GpccGroup &g, hipStream_t s, bool f32, int concurrent)
{
    if (f32) enqueue_factor_t<float>(h, c, g, s, concurrent);
    else enqueue_factor_t<double>(h, c, g, s, concurrent);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(h, GPCC_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
    return 0;
}

// ------------------------------------------------------------------------------------------
// The small-N family (gpcc_small.hip.h): N <= GPCC_SMALL_MAXN -- the sizes of the reference's own documentation (N = 110, 150,
// README.md:156-287) -- is ONE launch per batch on the caller's stream: no workspace, no slots, no groups, no events.
// ------------------------------------------------------------------------------------------
static inline bool small_path(const gpcc_handle_t h) { return h->small_n && h->N <= GPCC_SMALLW_MAXN; }

// d_xpar != NULL: requests of the optimiser (M x (L+1) unconstrained vectors, unpacked on the device; d_delays is then the
// candidate-delay table and d_xrow[i] the row of evaluation i)
static int enqueue_small(gpcc_handle_t h, int M, const double *d_delays, const double *d_alpha, const double *d_rho,
                         double *d_loglik, int *d_info, hipStream_t caller, const double *d_xpar = nullptr,
                         const int *d_xrow = nullptr, double rhomin = 0.0, double rhomax = 0.0)
{
    GpccCtx c = make_ctx(h);
    c.nrhs = 1; c.woodbury = 0;   // the literal fp64 model whatever the handle's precision
    GpccGroup g;
    g.delays = d_delays; g.alpha = d_alpha; g.rho = d_rho;
    g.out_loglik = d_loglik; g.out_info = d_info; g.out_cond = nullptr;
    g.first = 0; g.slot0 = 0; g.cnt = M; g.spread = 0;
    g.xpar = d_xpar; g.xrow = d_xrow; g.rhomin = rhomin; g.rhomax = rhomax;
    const int nb = (h->N + 1 + 15) / 16;   // the matrix bordered by the right-hand side, in 16 x 16 blocks
    // one wave per evaluation where that fits (N <= 191) and the batch is large enough to fill the SIMDs; four waves per
    // evaluation for N = 192 .. 383, and for batches of at most small_wide_max evaluations (latency-bound: optimiser rounds)
    const bool wide = nb > GPCC_SMALL_MAXNB || (M <= h->small_wide_max && nb >= 5);
    hipError_t e = hipSuccess;
    {
        ProfScope pr(h, GPCC_PROF_SMALL_EVAL, caller);
        typedef hipError_t (*launch_t)(int, const GpccCtx &, const GpccGroup &, hipStream_t);
        static const launch_t narrow[4] = {gpcc_small_launch_0, gpcc_small_launch_1, gpcc_small_launch_2, gpcc_small_launch_3};
        static const launch_t four[4] = {gpcc_smallw_launch_0, gpcc_smallw_launch_1, gpcc_smallw_launch_2, gpcc_smallw_launch_3};
        e = (wide ? four : narrow)[h->kernel_id & 3](nb, c, g, caller);
    }
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) return fail(h, GPCC_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
    h->small_count += M;
    return 0;
}

// M evaluations, device pointers, enqueued behind `caller` and joined back into it; d_cond (2 per evaluation, fp32
// handles) may be NULL.  The calling thread holds the device.
static int enqueue_batch(gpcc_handle_t h, int M, const double *d_delays, const double *d_alpha, const double *d_rho,
                         double *d_loglik, int *d_info, double *d_cond, hipStream_t caller)
{
    if (small_path(h)) return enqueue_small(h, M, d_delays, d_alpha, d_rho, d_loglik, d_info, caller);
    int rc = ensure_workspace(h);
    if (rc) return rc;
    if (!h->share_now) h->share_now = (h->shared_prefix == 2);   // device pointers cannot be inspected: only on assertion
    if (h->chain_max > 0 && h->precision == GPCC_PRECISION_FP64 && h->nt > 1) {
        const int tail = (M % h->ws_slots) ? M % h->ws_slots : h->ws_slots;   // (only a batch's last group can be this small)
        if (chain_policy(h, tail, h->nt)) {
            rc = ensure_chain(h);
            if (rc) return rc;
        }
    }
    const GpccCtx c = make_ctx(h);
    const int S = h->prof ? 1 : h->ws_streams;  // profiling serialises groups onto one stream
    const int cs = h->ws_slots;
    const int ngroups = (M + cs - 1) / cs;
    const int used = ngroups < S ? ngroups : S;
    HIPCHK(h, hipEventRecord(h->ev_start, caller));
    for (int s = 0; s < used; ++s) HIPCHK(h, hipStreamWaitEvent(h->str[s], h->ev_start, 0));
    for (int gi = 0; gi < ngroups; ++gi) {
        const int s = gi % S;
        GpccGroup g;
        g.delays = d_delays; g.alpha = d_alpha; g.rho = d_rho;
        g.out_loglik = d_loglik; g.out_info = d_info; g.out_cond = d_cond;
        g.first = gi * cs;
        g.slot0 = s * cs;
        g.cnt = (M - g.first < cs) ? (M - g.first) : cs;
        g.spread = 0;   // decided per launch sequence in enqueue_factor_t
        GpccCtx cg = c;
        if (h->share_now && h->share_tiles > 0 && h->share_tiles < h->nt && g.cnt > h->right_looking_max && g.cnt > 1)
            cg.share_p = h->share_tiles;
        // a group of a few dozen evaluations spends up to a fifth of its time in the serial chain diagonal step -> panel solve
        // with the chip nearly idle (16 evaluations at N = 4096: 2.1 of 8.9 ms): run it as two halves on two streams, so that
        // the update of one half overlaps the chain of the other (same slots; each half picks its own path by its size)
        bool split = h->split_min > 0 && g.cnt >= h->split_min && g.cnt <= h->split_max && h->nt >= h->split_nt_min;
        if (h->split_min > 0 && h->split_small && g.cnt < h->split_min)
            split = (g.cnt > h->right_looking_max && g.cnt < 24 && (h->nt <= 16 || (h->nt <= 24 && g.cnt < 20))) || (g.cnt >= 6 && g.cnt <= h->right_looking_max && h->nt >= 24);
        split = split && !h->prof && !cg.share_p && g.cnt >= 2 && h->nt > 1 && !takes_chain(h, cg, g.cnt);
        if (split) {
            GpccGroup ga = g, gb = g;
            ga.cnt = g.cnt >= 24 ? 8 * ((g.cnt + 15) / 16) : (g.cnt + 1) / 2;   // (multiples of 8 keep an evaluation's workgroups on one XCD)
            gb.first = g.first + ga.cnt; gb.slot0 = g.slot0 + ga.cnt; gb.cnt = g.cnt - ga.cnt;
            hipError_t he = hipEventRecord(h->ev_fork[s], h->str[s]);
            if (he == hipSuccess) he = hipStreamWaitEvent(h->str2[s], h->ev_fork[s], 0);
            if (he == hipSuccess) {
                rc = enqueue_group(h, cg, ga, h->str[s], true, false, -1, 2);
                if (!rc) rc = enqueue_group(h, cg, gb, h->str2[s], true, false, -1, 2);
            }
            if (he == hipSuccess && !rc) he = hipEventRecord(h->ev_join[s], h->str2[s]);
            if (he == hipSuccess && !rc) he = hipStreamWaitEvent(h->str[s], h->ev_join[s], 0);
            if (he != hipSuccess && !rc) rc = fail(h, GPCC_ERR_HIP, "split group: %s", hipGetErrorString(he));
            if (rc) break;
            continue;
        }
        rc = enqueue_group(h, cg, g, h->str[s]);
        if (rc) break;
    }
    if (rc) {
        // whatever was enqueued (possibly one half of a split group on its own stream, never joined) still reads the caller's
        // parameter arrays and writes its outputs: nothing may be left running when the error is returned
        const std::string msg = h->err;
        for (int s = 0; s < h->ws_streams; ++s) {
            if (h->str[s]) (void)hipStreamSynchronize(h->str[s]);
            if (h->str2[s]) (void)hipStreamSynchronize(h->str2[s]);
        }
        (void)hipGetLastError();
        h->err = msg;
        h->share_now = false;
        return rc;
    }
    for (int s = 0; s < used; ++s) {
        HIPCHK(h, hipEventRecord(h->ev_done[s], h->str[s]));
        HIPCHK(h, hipStreamWaitEvent(caller, h->ev_done[s], 0));
    }
    h->share_now = false;
    return 0;
}


// ------------------------------------------------------------------------------------------
// fp32 accuracy guard.  An fp32 blocked Cholesky (fp64 diagonal blocks, fp64 right-hand sides) computes
// L~ L~' = K0 + E with |E_ij| <= gamma sqrt(K_ii K_jj), gamma between u32 and N u32; to first order the log-likelihood
// moves by -(tr(K0^-1 E) - w' E w) / 2 (w = K0^-1 r).  The w' E w part (10-100x the other) is removed by the fp64
// refinement of the quadratic forms (gpcc_back_solve / gpcc_refine_partials / gpcc_refine_finish); what is left grows
// with the pivot ratios K_ii / d_i >= 1 (d_i = L_ii^2, the Schur complement a pivot sees; (K0^-1)_ii >= 1 / d_i), and
// once u32 times the condition number approaches 1 the factor is useless and the refinement with it.
// gpcc_diag_factor therefore accumulates S = sum_i K_ii / d_i per evaluation, and an evaluation whose MEAN PIVOT RATIO
// S / N exceeds a limit is repeated in fp64.  Calibration (tools/calibrate_fp32.py against the fp64 path, N 1..4096,
// 1-6 bands, all kernels, both b-modes, sigma 0.05..1, alpha 1e-2..1e2, rho 0.1..300; raw fp32 errors in that set reach
// 0.37, refined ones 23 where the factor is garbage):
//   refined   (profiles/r02/fp32_refined_calibration.log.gz, 43 484 evaluations): worst error among S/N <= 300 is
//             2.1e-5, among S/N <= 1000 1.3e-4, among S/N <= 2000 2.6e-3               -> limit 300 (50x below the bar)
//   unrefined (profiles/r02/fp32_guard_calibration.log.gz, 45 214 evaluations): worst error among S/N <= 50 is 3.6e-4,
//             among S/N <= 100 6.5e-4 (and a soak case at 1.06e-3 below 74)             -> limit 30
// The benchmark's own regime (sigma = 0.75, alpha = var(y)) has S/N ~ 15-45, up to ~150 with alpha scaled by 2.
// A measure fitted to a random sample, not a proof; "fp32_guard" = 0 switches it off.
// ------------------------------------------------------------------------------------------
// Round 3: the mean hides outliers.  An ADVERSARIAL search (tools/adversarial_fp32.py: evolutionary, every candidate evaluated on
// the GPU) found hyper-parameters that pass the mean test and are badly wrong -- tiny rho (0.1-0.16, a nearly diagonal K) with a
// huge amplitude in one band: S/N = 299.8 with an error of 0.61, 297 with 4e-2 (profiles/r03/fp32_guard_adversarial_*.log).
// In all of them a FEW pivots have ratios K_ii / d_i of 4e4 ... 1e6 -- at u32 times that the pivot itself is wrong in its leading
// digits and the first-order error model behind the mean does not apply -- while every survivor with a largest ratio below
// 7e3 stayed below 1e-3 as long as S/N <= 700.  So the guard also bounds the LARGEST ratio.  A second search against "mean <= 300 and
// max <= 1e4" (3.2 million evaluations) topped out at 4.8e-4, at a largest ratio of 9.6e3: the limit is set to 5e3 for margin.
#define GPCC_FP32_LIMIT_REFINED 300.0
#define GPCC_FP32_LIMIT_RAW 30.0
#define GPCC_FP32_LIMIT_MAX_RATIO 5.0e3
static inline bool fp32_needs_fp64(double S, double mx, int N, bool refined)
{
    return !(S <= (refined ? GPCC_FP32_LIMIT_REFINED : GPCC_FP32_LIMIT_RAW) * (double)(N > 0 ? N : 1)) ||   // NaN -> true
           !(mx <= GPCC_FP32_LIMIT_MAX_RATIO);
}

__global__ void gpcc_gather_params(int nf, int L, const int *idx, const double *delays, const double *alpha, const double *rho,
                                   double *out /* nf x L | nf x L | nf */)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nf) return;
    const long src = idx[i];
    for (int l = 0; l < L; ++l) {
        out[(long)i * L + l] = delays[src * L + l];
        out[(long)nf * L + (long)i * L + l] = alpha[src * L + l];
    }
    out[2L * nf * L + i] = rho[src];
}

__global__ void gpcc_scatter_results(int nf, const int *idx, const double *ll, const int *info, double *out_ll, int *out_info)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nf) return;
    out_ll[idx[i]] = ll[i];
    out_info[idx[i]] = info[i];
}

// the fp64 twin of an fp32 handle: the literal fp64 model on the same light curves, a small workspace of its own.  Behind the guard
// (flagged evaluations are repeated by it) and behind "fp32_chain" (few-evaluation calls are evaluated by it in the first place).
static int ensure_fb(gpcc_handle_t h, long want_slots)
{
    if (!h->fb) {
        int rc = gpcc_create(&h->fb, h->L, h->Nl, h->t_host.data(), h->y_host.data(), h->sigma_host.data(), h->kernel_id, h->mb,
                             GPCC_PRECISION_FP64, h->device);
        if (rc) return fail(h, rc, "fp32 handle: creating the fp64 handle failed: %s", g_err.c_str());
        gpcc_set_option(h->fb, "shared_prefix", 0);
        gpcc_set_option(h->fb, "streams", 1);   // (its workspace is sized for one stream's slots)
        h->fb_slots = 0;
    }
    if (want_slots > h->fb_slots) {   // it only ever grows
        gpcc_set_option(h->fb, "slots_per_stream", want_slots);
        h->fb_slots = want_slots;
    }
    return 0;
}

// fp32 handle, a call of a few evaluations (one objective(alpha, rho), marginaliseb.jl:133-141 as Optim calls it): on the
// launch-per-step path such a group is bound by the latency of its serial chain, not by the matrix pipe -- fp32 tiles buy nothing
// there -- and the persistent launch (gpcc_chain.hip.h) exists for fp64 tiles only.  The call is handed to the fp64 twin: its result
// is the fp64 handle's, bit for bit (well inside the fp32 mode's 1e-3), in 2/3 of the time at N = 4096.  Same policy as takes_chain.
static bool fp32_call_goes_to_fp64_chain(gpcc_handle_t h, int M)
{
    return h->precision == GPCC_PRECISION_FP32 && h->fp32_chain && !small_path(h) && !h->is_multi() && !h->prof &&
           chain_policy(h, M, h->nt);
}

static int fp32_prepare_fp64_chain(gpcc_handle_t h, int M)
{
    int rc = ensure_fb(h, M <= 16 ? 16 : (M + 7) / 8 * 8);   // (one group of the twin's workspace)
    if (rc) return rc;
    h->fb->chain_max = h->chain_max;   // the twin follows this handle's few-evaluation options
    h->fb->chain_work_max = h->chain_work_max;
    h->fb->chain_wide_work_max = h->chain_wide_work_max;
    h->fb->chain_helpers_max = h->chain_helpers_max;
    h->fb->chain_quarters_max = h->chain_quarters_max;
    h->fb->chain_batch = h->chain_batch;
    h->fb->chain_batch_min = h->chain_batch_min;
    h->fb->chain_workers_max = h->chain_workers_max;
    h->fb->right_looking_max = h->right_looking_max;
    h->cond_host.assign(2 * (size_t)M, 0.0);   // evaluated in fp64: nothing to guard
    h->fp32_chain_count += M;
    return 0;
}

static int fp32_guard_pass(gpcc_handle_t h, int M, const double *d_delays, const double *d_alpha, const double *d_rho,
                           double *d_loglik, int *d_info, hipStream_t caller)
{
    // read the fp32 results and their conditioning back (the one synchronisation an fp32 handle adds per call)
    h->cond_host.resize(2 * (size_t)M);
    h->ll_host.resize(M);
    h->info_host.resize(M);
    HIPCHK(h, hipMemcpyAsync(h->cond_host.data(), h->d_ocond, sizeof(double) * 2 * M, hipMemcpyDeviceToHost, caller));
    HIPCHK(h, hipMemcpyAsync(h->ll_host.data(), d_loglik, sizeof(double) * M, hipMemcpyDeviceToHost, caller));
    HIPCHK(h, hipMemcpyAsync(h->info_host.data(), d_info, sizeof(int) * M, hipMemcpyDeviceToHost, caller));
    HIPCHK(h, hipStreamSynchronize(caller));
    for (int i = 0; i < M; ++i)   // a factorisation that broke down has no pivot ratios (the sums stop being meaningful at the failing pivot)
        if (h->info_host[i] != 0) h->cond_host[2 * (size_t)i] = h->cond_host[2 * (size_t)i + 1] = INFINITY;
    if (!h->fp32_guard) return 0;
    h->fb_idx_host.clear();
    for (int i = 0; i < M; ++i)
        // a pivot that is non-positive in fp32 may only be lost to rounding: fp64 decides (argument errors < 0 stay)
        if (h->info_host[i] > 0 || (h->info_host[i] == 0 && fp32_needs_fp64(h->cond_host[2 * i], h->cond_host[2 * i + 1], h->N, h->fp32_refine != 0)))
            h->fb_idx_host.push_back(i);
    const int nf = (int)h->fb_idx_host.size();
    if (nf == 0) return 0;
    {   // workspace of the fp64 repeat: as many slots as evaluations to repeat (a batch that is mostly ill-conditioned then
        // runs as few large groups on the fused path instead of many 16-wide right-looking ones), between 16 and 128,
        // within a quarter of the memory that is free right now; it only ever grows
        int rc0 = ensure_fb(h, 0);
        if (rc0) return rc0;
        size_t free_b = 0, total_b = 0;
        long want = nf < 16 ? 16 : (nf > 128 ? 128 : nf);
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const long per_slot = gpcc_get_option(h->fb, "bytes_per_slot");
            const long fit = per_slot > 0 ? (long)(free_b / 4 / (size_t)per_slot) : want;
            if (want > fit) want = fit;
        }
        if (want < 1) want = 1;
        rc0 = ensure_fb(h, want);
        if (rc0) return rc0;
    }
    if (nf > h->fb_cap) {
        hipFree(h->d_fb_idx); hipFree(h->d_fb_par); hipFree(h->d_fb_out); hipFree(h->d_fb_info);
        h->d_fb_idx = h->d_fb_info = nullptr; h->d_fb_par = h->d_fb_out = nullptr; h->fb_cap = 0;
        const long cap = nf > 64 ? nf : 64;
        HIPCHK(h, hipMalloc(&h->d_fb_idx, sizeof(int) * cap));
        HIPCHK(h, hipMalloc(&h->d_fb_info, sizeof(int) * cap));
        HIPCHK(h, hipMalloc(&h->d_fb_par, sizeof(double) * cap * (2 * h->L + 1)));
        HIPCHK(h, hipMalloc(&h->d_fb_out, sizeof(double) * cap));
        h->fb_cap = cap;
    }
    HIPCHK(h, hipMemcpyAsync(h->d_fb_idx, h->fb_idx_host.data(), sizeof(int) * nf, hipMemcpyHostToDevice, caller));
    gpcc_gather_params<<<(nf + 127) / 128, 128, 0, caller>>>(nf, h->L, h->d_fb_idx, d_delays, d_alpha, d_rho, h->d_fb_par);
    int rc = gpcc_loglik_batch_device(h->fb, nf, h->d_fb_par, h->d_fb_par + (long)nf * h->L, h->d_fb_par + 2L * nf * h->L,
                                      h->d_fb_out, h->d_fb_info, caller);
    if (rc) return fail(h, rc, "fp32 guard: fp64 re-evaluation failed: %s", h->fb->err.c_str());
    gpcc_scatter_results<<<(nf + 127) / 128, 128, 0, caller>>>(nf, h->d_fb_idx, h->d_fb_out, h->d_fb_info, d_loglik, d_info);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(h, GPCC_ERR_HIP, "fp32 guard: %s", hipGetErrorString(e));
    HIPCHK(h, hipStreamSynchronize(caller));   // fb_idx_host / staging may be reused by the next call
    h->fb_count += nf;
    return 0;
}

extern "C" int gpcc_loglik_batch_device(gpcc_handle_t h, int M, const double *d_delays, const double *d_alpha,
                                        const double *d_rho, double *d_loglik, int *d_info, void *stream)
{
    if (!h) return fail(h, GPCC_ERR_ARGUMENT, "NULL handle");
    if (M < 0) return fail(h, GPCC_ERR_ARGUMENT, "M=%d < 0", M);
    if (M == 0) return 0;
    if (!d_delays || !d_alpha || !d_rho || !d_loglik || !d_info) return fail(h, GPCC_ERR_ARGUMENT, "NULL pointer");
    if (h->is_multi()) return fail(h, GPCC_ERR_UNSUPPORTED, "a multi-device handle takes host pointers (gpcc_loglik_batch): device pointers belong to one device");
    GPCC_ON_DEVICE(h, h->device);
    int rc = stream_on_device(h, stream, h->device);
    if (rc) return rc;
    hipStream_t caller = (hipStream_t)stream;
    if (fp32_call_goes_to_fp64_chain(h, M)) {
        rc = fp32_prepare_fp64_chain(h, M);
        if (rc) return rc;
        rc = gpcc_loglik_batch_device(h->fb, M, d_delays, d_alpha, d_rho, d_loglik, d_info, stream);
        if (rc) return fail(h, rc, "fp32 handle, few-evaluation call in fp64: %s", h->fb->err.c_str());
        return 0;
    }
    const bool small = small_path(h);
    if (small && h->precision == GPCC_PRECISION_FP32) h->cond_host.assign(2 * (size_t)M, 0.0);   // evaluated in fp64: nothing to guard
    const bool f32 = h->precision == GPCC_PRECISION_FP32 && !small;
    if (f32 && M > h->cond_cap) {
        hipFree(h->d_ocond);
        h->d_ocond = nullptr; h->cond_cap = 0;
        HIPCHK(h, hipMalloc(&h->d_ocond, sizeof(double) * 2 * M));
        h->cond_cap = M;
    }
    rc = enqueue_batch(h, M, d_delays, d_alpha, d_rho, d_loglik, d_info, f32 ? h->d_ocond : nullptr, caller);
    if (rc) return rc;
    if (f32) return fp32_guard_pass(h, M, d_delays, d_alpha, d_rho, d_loglik, d_info, caller);
    return 0;
}

static int ensure_staging(gpcc_handle_t h, long M)
{
    if (M <= h->par_cap) return 0;
    hipFree(h->d_par); hipFree(h->d_out); hipFree(h->d_oinfo);
    h->d_par = h->d_out = nullptr; h->d_oinfo = nullptr; h->par_cap = 0;
    HIPCHK(h, hipMalloc(&h->d_par, sizeof(double) * M * (2 * h->L + 1)));
    HIPCHK(h, hipMalloc(&h->d_out, sizeof(double) * M));
    HIPCHK(h, hipMalloc(&h->d_oinfo, sizeof(int) * M));
    h->par_cap = M;
    return 0;
}

// Host-pointer parameters of M evaluations -> the handle's staging buffers -> the device path, results left in
// d_loglik / d_info (device), everything ordered on h->main_stream, nothing synchronised.  The caller holds the device.
static int enqueue_host_batch(gpcc_handle_t h, int M, const double *delays, const double *alpha, const double *rho,
                              double *d_loglik, int *d_info)
{
    int rc = ensure_staging(h, M);
    if (rc) return rc;
    const long ML = (long)M * h->L;
    double *dd = h->d_par, *da = h->d_par + ML, *dr = h->d_par + 2 * ML;
    HIPCHK(h, hipMemcpyAsync(dd, delays, sizeof(double) * ML, hipMemcpyHostToDevice, h->main_stream));
    HIPCHK(h, hipMemcpyAsync(da, alpha, sizeof(double) * ML, hipMemcpyHostToDevice, h->main_stream));
    HIPCHK(h, hipMemcpyAsync(dr, rho, sizeof(double) * M, hipMemcpyHostToDevice, h->main_stream));
    if (h->shared_prefix == 1 && M > 1 && h->share_tiles > 0) {
        // fixed-hyper-parameter delay sweeps (README.md:172-174 with delays = [0; d]): every evaluation has the same
        // band-1 amplitude, delay and rho, so the leading tile rows of K are identical within a group
        bool same = rho[0] > 0.0;
        for (int i = 0; i < M && same; ++i) {
            same = alpha[(long)i * h->L] == alpha[0] && delays[(long)i * h->L] == delays[0] && rho[i] == rho[0];
            for (int l = 0; l < h->L && same; ++l) same = alpha[(long)i * h->L + l] > 0.0;   // no argument errors in the batch
        }
        h->share_now = same;
    }
    rc = gpcc_loglik_batch_device(h, M, dd, da, dr, d_loglik ? d_loglik : h->d_out, d_info ? d_info : h->d_oinfo, h->main_stream);
    h->share_now = false;
    return rc;
}

static int ensure_lane(gpcc_handle_t h, int li, long M)
{
    SmallLane &ln = h->lanes[li];
    if (!ln.stream) HIPCHK(h, hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking));
    if (M <= ln.cap) return 0;
    if (ln.par) hipHostFree(ln.par);
    if (ln.ll) hipHostFree(ln.ll);
    if (ln.info) hipHostFree(ln.info);
    if (ln.row) hipHostFree(ln.row);
    ln.par = ln.ll = nullptr; ln.info = ln.row = nullptr; ln.cap = 0;
    const long cap = M < 1024 ? 1024 : M + M / 2;
    HIPCHK(h, hipHostMalloc((void **)&ln.par, sizeof(double) * cap * (2 * h->L + 1), hipHostMallocDefault));
    HIPCHK(h, hipHostMalloc((void **)&ln.ll, sizeof(double) * cap, hipHostMallocDefault));
    HIPCHK(h, hipHostMalloc((void **)&ln.info, sizeof(int) * cap, hipHostMallocDefault));
    HIPCHK(h, hipHostMalloc((void **)&ln.row, sizeof(int) * cap, hipHostMallocDefault));
    ln.cap = cap;
    return 0;
}

extern "C" int gpcc_loglik_batch(gpcc_handle_t h, int M, const double *delays, const double *alpha,
                                 const double *rho, double *loglik, int *info)
{
    if (!h) return fail(h, GPCC_ERR_ARGUMENT, "NULL handle");
    if (M < 0) return fail(h, GPCC_ERR_ARGUMENT, "M=%d < 0", M);
    if (M == 0) return 0;
    if (!delays || !alpha || !rho || !loglik || !info) return fail(h, GPCC_ERR_ARGUMENT, "NULL pointer");
    if (h->is_multi()) return multi_loglik_batch(h, M, delays, alpha, rho, loglik, info);
    GPCC_ON_DEVICE(h, h->device);
    if (small_path(h)) {   // zero-copy: parameters read from, results written to pinned host memory by the kernel itself
        int rc = ensure_lane(h, 0, M);
        if (rc) return rc;
        SmallLane &ln = h->lanes[0];
        const long ML = (long)M * h->L;
        memcpy(ln.par, delays, sizeof(double) * ML);
        memcpy(ln.par + ML, alpha, sizeof(double) * ML);
        memcpy(ln.par + 2 * ML, rho, sizeof(double) * M);
        if (h->precision == GPCC_PRECISION_FP32) h->cond_host.assign(2 * (size_t)M, 0.0);   // evaluated in fp64: nothing to guard
        rc = enqueue_small(h, M, ln.par, ln.par + ML, ln.par + 2 * ML, ln.ll, ln.info, ln.stream);
        if (rc) return rc;
        HIPCHK(h, hipStreamSynchronize(ln.stream));
        memcpy(loglik, ln.ll, sizeof(double) * M);
        memcpy(info, ln.info, sizeof(int) * M);
        return 0;
    }
    if (fp32_call_goes_to_fp64_chain(h, M)) {
        int rc = fp32_prepare_fp64_chain(h, M);
        if (rc) return rc;
        rc = gpcc_loglik_batch(h->fb, M, delays, alpha, rho, loglik, info);
        if (rc) return fail(h, rc, "fp32 handle, few-evaluation call in fp64: %s", h->fb->err.c_str());
        return 0;
    }
    if (h->precision == GPCC_PRECISION_FP64 && chain_policy(h, M, h->nt) && !h->prof) {
        // One objective(alpha, rho) (marginaliseb.jl:133-141 as Optim calls it, :145-153): the whole call is pack -> two launches (assembly,
        // persistent factorisation) on workspace stream 0 -> one stream synchronisation.  Parameters are read from, results written to
        // pinned, device-mapped host memory by the kernels themselves: no copy calls, no events (the general path below costs ~50 us
        // more per call in runtime calls alone -- a quarter of an evaluation at N = 512)
        int rc = ensure_workspace(h);
        if (rc) return rc;
        if (M <= h->ws_slots) {   // (one group: a workspace of fewer slots takes the general path below, group by group)
        rc = ensure_chain(h);
        if (!rc) rc = ensure_lane(h, 0, M);
        if (rc) return rc;
        SmallLane &ln = h->lanes[0];
        const long ML = (long)M * h->L;
        memcpy(ln.par, delays, sizeof(double) * ML);
        memcpy(ln.par + ML, alpha, sizeof(double) * ML);
        memcpy(ln.par + 2 * ML, rho, sizeof(double) * M);
        const GpccCtx c = make_ctx(h);
        GpccGroup g;
        g.delays = ln.par; g.alpha = ln.par + ML; g.rho = ln.par + 2 * ML;
        g.out_loglik = ln.ll; g.out_info = ln.info; g.out_cond = nullptr;
        g.first = 0; g.slot0 = 0; g.cnt = M; g.spread = 0;
        if (takes_chain(h, c, M)) {
            rc = enqueue_group(h, c, g, h->str[0]);
            if (rc) { (void)hipStreamSynchronize(h->str[0]); return rc; }
            HIPCHK(h, hipStreamSynchronize(h->str[0]));
            memcpy(loglik, ln.ll, sizeof(double) * M);
            memcpy(info, ln.info, sizeof(int) * M);
            for (int i = 0; i < M; ++i)
                if (info[i] == GPCC_INFO_TIMEOUT)
                    return fail(h, GPCC_ERR_STATE, "the persistent few-evaluation launch was abandoned (a bounded wait expired: evaluation %d); "
                                                   "gpcc_set_option(handle, \"chain_max\", 0) selects the launch-per-step path", i);
            return 0;
        }
        }
    }
    int rc = enqueue_host_batch(h, M, delays, alpha, rho, nullptr, nullptr);
    if (rc) return rc;
    HIPCHK(h, hipMemcpyAsync(loglik, h->d_out, sizeof(double) * M, hipMemcpyDeviceToHost, h->main_stream));
    HIPCHK(h, hipMemcpyAsync(info, h->d_oinfo, sizeof(int) * M, hipMemcpyDeviceToHost, h->main_stream));
    HIPCHK(h, hipStreamSynchronize(h->main_stream));
    for (int i = 0; i < M; ++i)
        if (info[i] == GPCC_INFO_TIMEOUT)
            return fail(h, GPCC_ERR_STATE, "the persistent few-evaluation launch was abandoned (a bounded wait expired: evaluation %d); "
                                           "gpcc_set_option(handle, \"chain_max\", 0) selects the launch-per-step path", i);
    return 0;
}

// ------------------------------------------------------------------------------------------
// Augmented systems (prediction, posterior of the offsets).  The points of the handle are followed,
// from the next tile boundary on, by `next` extra rows (test points, or explicit rows Q / Y); only
// the first nt (training) tile columns are factorised.  What the unchanged kernels then leave behind:
//   tiles (extra, extra) = C - V^T V  (V = L^-1 [cross block]): the predictive covariance, resp.
//                          -R^T K^-1 R for explicit rows R;
//   z[extra]             = -V^T w = -(kB*)^T K^-1 (Y - bbar): minus the centred predictive mean.
// One temporary slot; not a hot path.
// ------------------------------------------------------------------------------------------
struct AugRun {
    GpccCtx c;
    double *d_pts = nullptr;   // t | sig2 | resid | yv (4 x Npa)
    int *d_band = nullptr;
    double *d_ws = nullptr;    // tiles | linv | z | w | logdet | quad
    int *d_info = nullptr;
    int off = 0, next = 0;
    void release() { hipFree(d_pts); hipFree(d_band); hipFree(d_ws); hipFree(d_info); d_pts = d_ws = nullptr; d_band = d_info = nullptr; }
};

static int run_augmented(gpcc_handle_t h, const double *delays, const double *alpha, double rho, int next,
                         const double *ext_t, const int *ext_band, int marginalise_b, AugRun &a, bool factor = true)
{
    GPCC_ON_DEVICE(h, h->device);
    int rc = 0;
    rc = set_kernel_attributes(h);
    if (rc) return rc;
    rc = ensure_staging(h, 1);
    if (rc) return rc;
    for (int l = 0; l < h->L; ++l)
        if (!(alpha[l] > 0.0)) return fail(h, GPCC_ERR_ARGUMENT, "AssertionError: all(scale .> 0)");
    if (rho <= 0.0) return fail(h, GPCC_ERR_ARGUMENT, "ρ=%.8f is <= 0", rho);
    const int off = h->Np;                                   // extra rows start on a tile boundary
    const int nta = h->nt + (next + GPCC_TILE - 1) / GPCC_TILE;
    const int Npa = nta * GPCC_TILE;
    std::vector<double> pts(4 * (size_t)Npa, 0.0);
    std::vector<int> band(Npa, -1);
    for (int i = 0; i < h->N; ++i) {
        pts[i] = h->t_host[i];
        pts[Npa + i] = h->sig2_host[i];
        pts[2 * (size_t)Npa + i] = h->resid_host[i];
        pts[3 * (size_t)Npa + i] = h->y_host[i];
        band[i] = h->band_host[i];
    }
    for (int i = 0; i < next; ++i) {
        pts[off + i] = ext_t ? ext_t[i] : 0.0;               // sig2 = 0, resid = 0 for extra rows
        band[off + i] = ext_band[i];
    }
    const long stride = ((long)nta * (nta + 1) / 2) * GPCC_TILE_ELEMS;
    const size_t wsz = (size_t)stride + GPCC_TILE_ELEMS + 2 * (size_t)Npa + 1 + GPCC_MAXRHS * GPCC_MAXRHS;
    HIPCHK(h, hipMalloc(&a.d_pts, sizeof(double) * pts.size()));
    hipError_t e = hipMalloc(&a.d_band, sizeof(int) * Npa);
    if (e == hipSuccess) e = hipMalloc(&a.d_ws, sizeof(double) * wsz);
    if (e == hipSuccess) e = hipMalloc(&a.d_info, sizeof(int));
    if (e == hipSuccess) e = hipMemcpy(a.d_pts, pts.data(), sizeof(double) * pts.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(a.d_band, band.data(), sizeof(int) * Npa, hipMemcpyHostToDevice);
    if (e != hipSuccess) { a.release(); return fail(h, GPCC_ERR_HIP, "augmented run: %s", hipGetErrorString(e)); }
    GpccCtx c = make_ctx(h);
    c.t = a.d_pts; c.sig2 = a.d_pts + Npa; c.resid = a.d_pts + 2 * (size_t)Npa; c.yv = a.d_pts + 3 * (size_t)Npa;
    c.band = a.d_band;
    c.tiles = a.d_ws; c.linv = a.d_ws + stride; c.z = a.d_ws + stride + GPCC_TILE_ELEMS; c.w = c.z + Npa;
    c.logdet = c.w + Npa; c.gram = c.logdet + 1; c.info = a.d_info;
    c.kdiag = nullptr; c.cond = nullptr; c.gpart = nullptr; c.linv_keep = 0;
    c.sep = nullptr; c.seps = nullptr; c.sepflag = nullptr;   // (sized for the handle's own N: never folded here)
    c.slot_stride = stride; c.Np = Npa; c.nt = nta; c.nt_fact = h->nt; c.marginalise_b = marginalise_b;
    c.nrhs = 1; c.woodbury = 0;   // the dense utilities always run the literal fp64 model
    c.store_l = 1;
    if (!marginalise_b) for (int l = 0; l < GPCC_MAXL; ++l) c.sigma_b[l] = 0.0;
    hipStream_t s = h->main_stream;
    double *dd = h->d_par, *da = h->d_par + h->L, *dr = h->d_par + 2 * h->L;
    e = hipMemcpyAsync(dd, delays, sizeof(double) * h->L, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(da, alpha, sizeof(double) * h->L, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(dr, &rho, sizeof(double), hipMemcpyHostToDevice, s);
    if (e != hipSuccess) { a.release(); return fail(h, GPCC_ERR_HIP, "augmented run: %s", hipGetErrorString(e)); }
    GpccGroup g;
    g.delays = dd; g.alpha = da; g.rho = dr; g.out_loglik = h->d_out; g.out_info = h->d_oinfo; g.out_cond = nullptr;
    g.first = 0; g.slot0 = 0; g.cnt = 1; g.spread = 0;
    const bool was_prof = h->prof;
    h->prof = false;
    rc = enqueue_group(h, c, g, s, factor, true, 0);
    h->prof = was_prof;
    if (rc) { a.release(); return rc; }
    a.c = c; a.off = off; a.next = next;
    return 0;
}

// copies the (extra x extra) block (symmetric, + jitter on the diagonal), z[extra], loglik and info back
static int fetch_augmented(gpcc_handle_t h, AugRun &a, double *block, double *zext, double *loglik, int *info, double jitter,
                           int symmetric = 1)
{
    // (run_augmented restored the caller's current device on return: allocate and launch on the handle's again)
    DeviceGuard guard_(h, h->device);
    if (guard_.rc) { a.release(); return guard_.rc; }
    hipStream_t s = h->main_stream;
    const long nn = (long)a.next * a.next;
    double *d_dense = nullptr;
    hipError_t e = hipMalloc(&d_dense, sizeof(double) * nn);
    if (e == hipSuccess) {
        gpcc_export_dense<double><<<(unsigned)((nn + 255) / 256), 256, 0, s>>>(a.c, 0, d_dense, symmetric, a.off, a.next, jitter);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(block, d_dense, sizeof(double) * nn, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && zext) e = hipMemcpyAsync(zext, a.c.z + a.off, sizeof(double) * a.next, hipMemcpyDeviceToHost, s);
    double ll = 0.0;
    int inf = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&ll, h->d_out, sizeof(double), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(&inf, h->d_oinfo, sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    hipFree(d_dense);
    a.release();
    if (e != hipSuccess) return fail(h, GPCC_ERR_HIP, "augmented fetch: %s", hipGetErrorString(e));
    if (loglik) *loglik = ll;
    if (info) *info = inf;
    return 0;
}

extern "C" int gpcc_model_matrix(gpcc_handle_t h, const double *delays, const double *alpha, double rho, double *K_out)
{
    if (!h || !delays || !alpha || !K_out) return fail(h, GPCC_ERR_ARGUMENT, "NULL pointer");
    h = primary(h);
    AugRun a;
    int rc = run_augmented(h, delays, alpha, rho, 0, nullptr, nullptr, h->mb, a, false);
    if (rc) return rc;
    a.off = 0; a.next = h->N;
    return fetch_augmented(h, a, K_out, nullptr, nullptr, nullptr, 0.0, 1);
}

extern "C" int gpcc_factor_dense(gpcc_handle_t h, const double *delays, const double *alpha, double rho,
                                 double *L_out, int *info)
{
    if (!h || !delays || !alpha || !L_out) return fail(h, GPCC_ERR_ARGUMENT, "NULL pointer");
    h = primary(h);
    AugRun a;
    int rc = run_augmented(h, delays, alpha, rho, 0, nullptr, nullptr, h->mb, a, true);
    if (rc) return rc;
    a.off = 0; a.next = h->N;
    return fetch_augmented(h, a, L_out, nullptr, nullptr, info, 0.0, 0);
}

extern "C" int gpcc_predict(gpcc_handle_t h, const double *delays, const double *alpha, double rho, const int *Ntest,
                            const double *ttest, double *mu_out, double *Sigma_out, double *loglik, int *info)
{
    if (!h || !delays || !alpha || !Ntest || !ttest || !mu_out || !Sigma_out) return fail(h, GPCC_ERR_ARGUMENT, "NULL pointer");
    h = primary(h);
    long nt = 0;
    for (int l = 0; l < h->L; ++l) { if (Ntest[l] < 0) return fail(h, GPCC_ERR_ARGUMENT, "negative Ntest"); nt += Ntest[l]; }
    if (nt <= 0 || nt > 32768) return fail(h, GPCC_ERR_ARGUMENT, "total number of test points %ld outside [1, 32768]", nt);
    std::vector<int> eb(nt);
    long o = 0;
    for (int l = 0; l < h->L; ++l) for (int n = 0; n < Ntest[l]; ++n) eb[o++] = l;
    AugRun a;
    int rc = run_augmented(h, delays, alpha, rho, (int)nt, ttest, eb.data(), h->mb, a);
    if (rc) return rc;
    std::vector<double> z(nt);
    // Σpred = cB - kB*' (KSobsB \ kB*) + JITTER*I  (marginaliseb.jl:275-279), JITTER = 1e-8 (:69)
    rc = fetch_augmented(h, a, Sigma_out, z.data(), loglik, info, 1e-8);
    if (rc) return rc;
    // μpred = kB*' (KSobsB \ (Y - b̄)) + Q* μb  (:283-285);  z[test] = -(kB*)' K^-1 (Y - b̄)
    for (long i = 0; i < nt; ++i) mu_out[i] = -z[i] + h->mean_b[eb[i]];
    return 0;
}

extern "C" int gpcc_posterior_offsets(gpcc_handle_t h, const double *delays, const double *alpha, double rho,
                                      double *mu_postb, double *Sigma_postb, int *info)
{
    if (!h || !delays || !alpha || !mu_postb || !Sigma_postb) return fail(h, GPCC_ERR_ARGUMENT, "NULL pointer");
    h = primary(h);
    if (!h->mb) return fail(h, GPCC_ERR_ARGUMENT, "the fixed-b variant (gpccfixdelay.jl) has no posterior over offsets");
    for (int l = 0; l < h->L; ++l)   // Sigma_b[l] = 100 var(y_l) = 0: inv(Sigma_b) of marginaliseb.jl:248 does not exist
        if (!(h->sigma_b[l] > 0.0)) return fail(h, GPCC_ERR_ARGUMENT, "band %d has zero flux variance: the prior of its offset is degenerate (Sigma_b singular)", l + 1);
    const int L = h->L, ne = L + 1;
    std::vector<int> eb(ne);
    for (int e = 0; e < ne; ++e) eb[e] = -2 - e;          // rows Q[:,0..L-1] and Y against (Sobs + K), no B term
    AugRun a;
    int rc = run_augmented(h, delays, alpha, rho, ne, nullptr, eb.data(), 0, a);
    if (rc) return rc;
    std::vector<double> S((size_t)ne * ne);
    int inf = 0;
    rc = fetch_augmented(h, a, S.data(), nullptr, nullptr, &inf, 0.0);
    if (rc) return rc;
    if (info) *info = inf;
    // S = -[Q Y]' (Sobs+K)^-1 [Q Y].   Σpostb = (Σb^-1 + Q'(Sobs+K)^-1 Q)^-1 ;
    // μpostb = Σpostb (Q'(Sobs+K)^-1 Y + Σb^-1 μb)      (marginaliseb.jl:248-250)
    std::vector<double> A((size_t)L * 2 * L, 0.0);        // [Σb^-1 + Q'K^-1Q | I] -> Gauss-Jordan
    for (int i = 0; i < L; ++i) {
        for (int j = 0; j < L; ++j) A[i * 2 * L + j] = -S[(size_t)j * ne + i] + (i == j ? 1.0 / h->sigma_b[i] : 0.0);
        A[i * 2 * L + L + i] = 1.0;
    }
    for (int p = 0; p < L; ++p) {                          // SPD: no pivoting needed
        const double piv = A[p * 2 * L + p];
        for (int j = 0; j < 2 * L; ++j) A[p * 2 * L + j] /= piv;
        for (int i = 0; i < L; ++i) {
            if (i == p) continue;
            const double f = A[i * 2 * L + p];
            for (int j = 0; j < 2 * L; ++j) A[i * 2 * L + j] -= f * A[p * 2 * L + j];
        }
    }
    for (int i = 0; i < L; ++i)
        for (int j = 0; j < L; ++j)
            Sigma_postb[(size_t)j * L + i] = 0.5 * (A[i * 2 * L + L + j] + A[j * 2 * L + L + i]);   // makematrixsymmetric, :252
    for (int i = 0; i < L; ++i) {
        double acc = 0.0;
        for (int j = 0; j < L; ++j) acc += A[i * 2 * L + L + j] * (-S[(size_t)j * ne + L] + h->mean_b[j] / h->sigma_b[j]);
        mu_postb[i] = acc;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------
extern "C" int gpcc_mvnormal_logpdf(int n, const double *Sigma, const double *mu, const double *x, double *loglik,
                                    int *info, int device_id)
{
    if (n < 1 || n > 65536 || !Sigma || !x || !loglik || !info) return fail(nullptr, GPCC_ERR_ARGUMENT, "bad argument");
    GPCC_ON_DEVICE(nullptr, device_id);
    int rc = 0;
    gpcc_handle_s tmp;   // stack handle: only err/prof fields are touched by the helpers
    rc = set_kernel_attributes(&tmp);
    if (rc) return fail(nullptr, rc, "%s", tmp.err.c_str());
    const int nt = (n + GPCC_TILE - 1) / GPCC_TILE, Np = nt * GPCC_TILE;
    const long stride = ((long)nt * (nt + 1) / 2) * GPCC_TILE_ELEMS;
    const size_t wsz = (size_t)stride + GPCC_TILE_ELEMS + 2 * (size_t)Np + 4 + GPCC_MAXRHS * GPCC_MAXRHS;
    std::vector<double> r(n);
    for (int i = 0; i < n; ++i) r[i] = x[i] - (mu ? mu[i] : 0.0);
    double *d_ws = nullptr, *d_in = nullptr;
    int *d_info = nullptr;
    hipError_t e = hipMalloc(&d_ws, sizeof(double) * wsz);
    if (e == hipSuccess) e = hipMalloc(&d_in, sizeof(double) * ((size_t)n * n + n));
    if (e == hipSuccess) e = hipMalloc(&d_info, 2 * sizeof(int));
    if (e == hipSuccess) e = hipMemcpy(d_in, Sigma, sizeof(double) * (size_t)n * n, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_in + (size_t)n * n, r.data(), sizeof(double) * n, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        GpccCtx c;
        memset(&c, 0, sizeof c);
        c.tiles = d_ws; c.linv = d_ws + stride; c.z = d_ws + stride + GPCC_TILE_ELEMS; c.w = c.z + Np;
        c.logdet = c.w + Np; c.gram = c.logdet + 1; c.info = d_info; c.nrhs = 1;
        c.slot_stride = stride; c.L = 1; c.N = n; c.Np = Np; c.nt = nt; c.nt_fact = nt;
        GpccGroup g;
        memset(&g, 0, sizeof g);
        g.out_loglik = c.gram + GPCC_MAXRHS * GPCC_MAXRHS; g.out_info = d_info + 1; g.cnt = 1;
        gpcc_load_dense<double><<<nt * nt, 256>>>(c, 0, d_in, n, d_in + (size_t)n * n);
        rc = enqueue_factor(&tmp, c, g, nullptr, false, 1);
        if (rc == 0) {
            e = hipMemcpy(loglik, g.out_loglik, sizeof(double), hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipMemcpy(info, g.out_info, sizeof(int), hipMemcpyDeviceToHost);
        }
    }
    hipFree(d_ws); hipFree(d_in); hipFree(d_info);
    if (rc) return fail(nullptr, rc, "%s", tmp.err.c_str());
    if (e != hipSuccess) return fail(nullptr, GPCC_ERR_HIP, "gpcc_mvnormal_logpdf: %s", hipGetErrorString(e));
    return 0;
}

// ------------------------------------------------------------------------------------------
extern "C" int gpcc_covariance(int kernel_id, int L, const double *scale, const double *delays, double rho,
                               const int *Nx, const double *x, const int *Ny, const double *y, double *out,
                               int device_id)
{
    if (kernel_id < 0 || kernel_id > 3) return fail(nullptr, GPCC_ERR_ARGUMENT, "unknown kernel_id %d", kernel_id);
    if (L < 1 || !scale || !delays || !Nx || !x || !Ny || !y || !out)
        return fail(nullptr, GPCC_ERR_ARGUMENT, "bad argument");
    for (int l = 0; l < L; ++l)
        if (!(scale[l] > 0.0)) return fail(nullptr, GPCC_ERR_ARGUMENT, "AssertionError: all(scale .> 0)");
    if (rho <= 0.0) return fail(nullptr, GPCC_ERR_ARGUMENT, "ρ=%.8f is <= 0", rho);
    GPCC_ON_DEVICE(nullptr, device_id);
    int rc = 0;
    long nx = 0, ny = 0;
    for (int l = 0; l < L; ++l) {
        if (Nx[l] < 0 || Ny[l] < 0) return fail(nullptr, GPCC_ERR_ARGUMENT, "negative band length");
        nx += Nx[l]; ny += Ny[l];
    }
    if (nx == 0 || ny == 0) return 0;
    // shifted times x - delays[band] and per-point scales (delayedCovariance.jl:27)
    std::vector<double> hx(2 * nx), hy(2 * ny);
    long o = 0;
    for (int l = 0; l < L; ++l)
        for (int n = 0; n < Nx[l]; ++n, ++o) { hx[o] = x[o] - delays[l]; hx[nx + o] = scale[l]; }
    o = 0;
    for (int l = 0; l < L; ++l)
        for (int n = 0; n < Ny[l]; ++n, ++o) { hy[o] = y[o] - delays[l]; hy[ny + o] = scale[l]; }
    double *dx = nullptr, *dy = nullptr, *dout = nullptr;
    hipError_t e = hipMalloc(&dx, sizeof(double) * 2 * nx);
    if (e == hipSuccess) e = hipMalloc(&dy, sizeof(double) * 2 * ny);
    if (e == hipSuccess) e = hipMalloc(&dout, sizeof(double) * nx * ny);
    if (e == hipSuccess) e = hipMemcpy(dx, hx.data(), sizeof(double) * 2 * nx, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dy, hy.data(), sizeof(double) * 2 * ny, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        const unsigned grid = (unsigned)((nx * ny + 255) / 256);
        switch (kernel_id) {
        case 0: gpcc_covariance_kernel<0><<<grid, 256>>>(nx, ny, dx, dx + nx, dy, dy + ny, rho, dout); break;
        case 1: gpcc_covariance_kernel<1><<<grid, 256>>>(nx, ny, dx, dx + nx, dy, dy + ny, rho, dout); break;
        case 2: gpcc_covariance_kernel<2><<<grid, 256>>>(nx, ny, dx, dx + nx, dy, dy + ny, rho, dout); break;
        default: gpcc_covariance_kernel<3><<<grid, 256>>>(nx, ny, dx, dx + nx, dy, dy + ny, rho, dout); break;
        }
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out, dout, sizeof(double) * nx * ny, hipMemcpyDeviceToHost);
    hipFree(dx); hipFree(dy); hipFree(dout);
    if (e != hipSuccess) return fail(nullptr, GPCC_ERR_HIP, "gpcc_covariance: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int gpcc_probabilities_device(int G, const double *d_loglik, const double *d_logprior, double *d_out,
                                         void *stream)
{
    if (G <= 0 || !d_loglik || !d_out) return fail(nullptr, GPCC_ERR_ARGUMENT, "bad argument");
    int dev = -1;
    if (stream) {   // the launch goes to the stream's device whatever the thread's current device is
        hipDevice_t sd = -1;
        if (hipStreamGetDevice((hipStream_t)stream, &sd) != hipSuccess) return fail(nullptr, GPCC_ERR_ARGUMENT, "stream argument is not a valid hipStream_t");
        dev = (int)sd;
    } else if (hipGetDevice(&dev) != hipSuccess) {
        return fail(nullptr, GPCC_ERR_HIP, "no HIP device available; libgpcc_hip has no CPU fallback");
    }
    GPCC_ON_DEVICE(nullptr, dev);
    gpcc_probabilities_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(G, d_loglik, d_logprior, d_out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, GPCC_ERR_HIP, "gpcc_probabilities: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int gpcc_probabilities(int G, const double *loglik, const double *logprior, double *out, int device_id)
{
    if (G <= 0 || !loglik || !out) return fail(nullptr, GPCC_ERR_ARGUMENT, "bad argument");
    GPCC_ON_DEVICE(nullptr, device_id);
    int rc = 0;
    double *d = nullptr;
    hipError_t e = hipMalloc(&d, sizeof(double) * 3 * (size_t)G);
    if (e == hipSuccess) e = hipMemcpy(d, loglik, sizeof(double) * G, hipMemcpyHostToDevice);
    if (e == hipSuccess && logprior) e = hipMemcpy(d + G, logprior, sizeof(double) * G, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        rc = gpcc_probabilities_device(G, d, logprior ? d + G : nullptr, d + 2 * (size_t)G, nullptr);
        if (rc) { hipFree(d); return rc; }
        e = hipMemcpy(out, d + 2 * (size_t)G, sizeof(double) * G, hipMemcpyDeviceToHost);
    }
    hipFree(d);
    if (e != hipSuccess) return fail(nullptr, GPCC_ERR_HIP, "gpcc_probabilities: %s", hipGetErrorString(e));
    return 0;
}

// ------------------------------------------------------------------------------------------
// gpcc_grid_loglik: the optimised per-delay log-likelihood (host logic in gpcc_fit.h, objective = gpcc_loglik_batch)
extern "C" int gpcc_unpack_params(int M, int L, const double *X, double rhomin, double rhomax, double *alpha, double *rho)
{
    if (M < 0 || L < 1 || L > GPCC_MAXL) return fail(nullptr, GPCC_ERR_ARGUMENT, "bad M=%d or L=%d", M, L);
    if (M == 0) return 0;
    if (!X || !alpha || !rho) return fail(nullptr, GPCC_ERR_ARGUMENT, "NULL pointer");
    for (long i = 0; i < M; ++i) {
        const double *x = X + i * (L + 1);
        for (int l = 0; l < L; ++l) alpha[i * L + l] = gpccfit::makepositive(x[l]) + 1e-8;    // makeα, marginaliseb.jl:112
        rho[i] = gpccfit::transformbetween(x[L], rhomin, rhomax);                             // makeρ, :114
    }
    return 0;
}

static void band_variances(gpcc_handle_t h, double *vary)
{
    long off = 0;
    for (int l = 0; l < h->L; ++l) {
        const int n = h->Nl[l];
        double m = 0.0, v = 0.0;
        for (int i = 0; i < n; ++i) m += h->y_host[off + i];
        m /= n;
        for (int i = 0; i < n; ++i) v += (h->y_host[off + i] - m) * (h->y_host[off + i] - m);
        vary[l] = n > 1 ? v / (n - 1) : 0.0;
        off += n;
    }
}

extern "C" int gpcc_initial_params(gpcc_handle_t h, int numberofrestarts, int initialrandom, double rhomin, double rhomax,
                                   unsigned long long seed, double *out)
{
    if (!h) return fail(h, GPCC_ERR_ARGUMENT, "NULL handle");
    if (numberofrestarts < 1 || initialrandom < 1 || !(rhomin > 0.0) || !(rhomax > rhomin + 2e-3) || !out)
        return fail(h, GPCC_ERR_ARGUMENT, "bad fit options (restarts %d, initialrandom %d, rho in (%g, %g))",
                    numberofrestarts, initialrandom, rhomin, rhomax);
    double vary[GPCC_MAXL];
    h = primary(h);
    band_variances(h, vary);
    gpccfit::initial_params(h->L, numberofrestarts, initialrandom, rhomin, rhomax, seed, vary, out);
    return 0;
}

extern "C" int gpcc_neldermead_batch(long P, int n, int iterations, double g_tol, const double *x0,
                                     gpcc_batch_objective_t f, void *ctx, double *xmin, double *fmin,
                                     int *iterations_out, long long *stats_out)
{
    if (P < 0 || n < 1 || iterations < 0) return fail(nullptr, GPCC_ERR_ARGUMENT, "bad P=%ld, n=%d or iterations=%d", P, n, iterations);
    if (P == 0) return 0;
    if (!x0 || !f || !xmin || !fmin) return fail(nullptr, GPCC_ERR_ARGUMENT, "NULL pointer");
    gpccfit::BatchedNelderMead nm(P, n, iterations, g_tol);
    const int rc = nm.run(f, ctx, x0, xmin, fmin);
    if (rc) return rc;
    if (iterations_out)
        for (long p = 0; p < P; ++p) iterations_out[p] = nm.it[p];
    if (stats_out) {
        stats_out[0] = nm.f_calls;
        stats_out[1] = nm.rounds;
    }
    return 0;
}

namespace {
struct FitEval {
    gpcc_handle_t h;
    const double *cand;   // G x L
    int R, L;
    double rhomin, rhomax;
    std::vector<double> delays, alpha, rho, ll;
    std::vector<int> info;
    bool device_unpack = false;   // single-device small-N handle: the kernel unpacks the optimiser's vectors (fit_eval_small)
    int lane = 0;                 // ... through this lane (stream + pinned buffers) of the handle
    long p0 = 0;                  // global index of this optimiser's problem 0 (a fit may be cut into slices, one per lane)
};

// the small-N route of an optimiser round: X and the delay rows go to pinned memory as they are, the kernel unpacks
static int fit_eval_small(gpcc_handle_t h, const FitEval &e, long K, const long *pidx, const double *X, double *f)
{
    GPCC_ON_DEVICE(h, h->device);
    int rc = ensure_lane(h, e.lane, K);
    if (rc) return rc;
    SmallLane &ln = h->lanes[e.lane];
    const int n = e.L + 1;
    memcpy(ln.par, X, sizeof(double) * K * n);
    for (long i = 0; i < K; ++i) ln.row[i] = (int)((e.p0 + pidx[i]) / e.R);
    rc = enqueue_small(h, (int)K, h->d_cand, nullptr, nullptr, ln.ll, ln.info, ln.stream, ln.par, ln.row, e.rhomin, e.rhomax);
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(ln.stream));
    for (long i = 0; i < K; ++i)   // safewrapper(negativeobjective), :149-153: a failed evaluation is +Inf
        f[i] = ln.info[i] == 0 ? -ln.ll[i] : std::numeric_limits<double>::infinity();
    return 0;
}

int fit_eval(void *vctx, long K, const long *pidx, const double *X, double *f)
{
    FitEval &e = *static_cast<FitEval *>(vctx);
    const int L = e.L;
    if (K > 0x7fffffffL) return fail(e.h, GPCC_ERR_ARGUMENT, "optimiser round of %ld evaluations", K);
    if (e.device_unpack) return fit_eval_small(e.h, e, K, pidx, X, f);
    e.delays.resize(K * L); e.alpha.resize(K * L); e.rho.resize(K); e.ll.resize(K); e.info.resize(K);
    for (long i = 0; i < K; ++i)
        for (int l = 0; l < L; ++l) e.delays[i * L + l] = e.cand[((e.p0 + pidx[i]) / e.R) * L + l];
    int rc = gpcc_unpack_params((int)K, L, X, e.rhomin, e.rhomax, e.alpha.data(), e.rho.data());
    if (rc) return rc;
    rc = gpcc_loglik_batch(e.h, (int)K, e.delays.data(), e.alpha.data(), e.rho.data(), e.ll.data(), e.info.data());
    if (rc) return rc;
    for (long i = 0; i < K; ++i)   // safewrapper(negativeobjective), :149-153: a failed evaluation is +Inf
        f[i] = e.info[i] == 0 ? -e.ll[i] : std::numeric_limits<double>::infinity();
    return 0;
}
}   // namespace

extern "C" int gpcc_grid_loglik(gpcc_handle_t h, int G, const double *delays, int iterations, int numberofrestarts,
                                int initialrandom, double rhomin, double rhomax, unsigned long long seed,
                                const double *init_params, double *loglik_out, double *alpha_out, double *rho_out,
                                int *info_out, int *iterations_out, long long *stats_out)
{
    if (!h) return fail(h, GPCC_ERR_ARGUMENT, "NULL handle");
    if (G < 0) return fail(h, GPCC_ERR_ARGUMENT, "G=%d < 0", G);
    if (G == 0) return 0;
    if (!delays || !loglik_out || !alpha_out || !rho_out || !info_out) return fail(h, GPCC_ERR_ARGUMENT, "NULL pointer");
    const int L = primary(h)->L, R = numberofrestarts, C = initialrandom, n = L + 1;
    if (R < 1 || C < 1 || iterations < 0 || !(rhomin > 0.0) || !(rhomax > rhomin + 2e-3))
        return fail(h, GPCC_ERR_ARGUMENT, "bad fit options (iterations %d, restarts %d, initialrandom %d, rho in (%g, %g))",
                    iterations, R, C, rhomin, rhomax);
    const long P = (long)G * R;
    if (P * (long)std::max(C, n + 1) > 0x7fffffffL) return fail(h, GPCC_ERR_ARGUMENT, "G x restarts = %ld is too large", P);
    std::vector<double> cands((size_t)R * C * n);
    if (init_params) memcpy(cands.data(), init_params, sizeof(double) * cands.size());
    else {
        double vary[GPCC_MAXL];
        band_variances(primary(h), vary);
        gpccfit::initial_params(L, R, C, rhomin, rhomax, seed, vary, cands.data());
    }
    // a multi-device handle shards the FIT BY DELAY (README.md:195-211, :258-287: pmap over candidate delays, one gpcc per worker):
    // every device runs its own lock-step optimiser over its delays, with every fast path of a single-device fit; one gather at the end
    if (h->is_multi())
        return multi_grid_loglik(h, G, delays, iterations, R, C, rhomin, rhomax, cands.data(), loglik_out, alpha_out, rho_out, info_out,
                                 iterations_out, stats_out);
    FitEval ev{h, delays, R, L, rhomin, rhomax, {}, {}, {}, {}, {}};
    const bool small_fit = small_path(h);
    if (small_fit && h->fit_device_unpack) {   // the candidate-delay table lives on the device for the duration of the fit
        GPCC_ON_DEVICE(h, h->device);
        if ((long)G * L > h->cand_cap) {
            hipFree(h->d_cand);
            h->d_cand = nullptr; h->cand_cap = 0;
            HIPCHK(h, hipMalloc(&h->d_cand, sizeof(double) * (size_t)G * L));
            h->cand_cap = (long)G * L;
        }
        HIPCHK(h, hipMemcpy(h->d_cand, delays, sizeof(double) * (size_t)G * L, hipMemcpyHostToDevice));
        ev.device_unpack = true;
    }

    // Problems p = (delay p / R, restart p % R), p in [lo, hi): the best of the random candidates starts each (:209; every
    // delay sees the same candidates -- each reference gpcc call seeds its own generator with `seed`), then the lock-step
    // Nelder-Mead.  A slice is self-contained: own optimiser, own evaluator state, own lane.
    std::vector<double> xmin((size_t)P * n), fmin(P);
    std::vector<int> its(P);
    auto fit_slice = [&](FitEval sev, long lo, long hi, long long *f_calls, long long *rounds) -> int {
        const long Ps = hi - lo;
        sev.p0 = lo;
        std::vector<long> pid((size_t)Ps * C);
        std::vector<double> X((size_t)Ps * C * n), f0((size_t)Ps * C), x0((size_t)Ps * n);
        for (long p = 0; p < Ps; ++p)
            for (int c = 0; c < C; ++c) {
                pid[p * C + c] = p;
                memcpy(&X[(p * C + c) * n], &cands[((size_t)((lo + p) % R) * C + c) * n], sizeof(double) * n);
            }
        int rc = fit_eval(&sev, Ps * C, pid.data(), X.data(), f0.data());
        if (rc) return rc;
        for (long p = 0; p < Ps; ++p) {
            int best = 0;
            for (int c = 1; c < C; ++c)
                if (f0[p * C + c] < f0[p * C + best]) best = c;
            memcpy(&x0[p * n], &X[(p * C + best) * n], sizeof(double) * n);
        }
        gpccfit::BatchedNelderMead nm(Ps, n, iterations, 1e-6);       // Optim.Options(iterations, g_tol = 1e-6), :205
        // small-N kernels: an evaluation's value does not depend on what else is in the batch, and up to ~4 waves per CU
        // cost no more time than one -- so latency-bound rounds evaluate the whole decision tree of an iteration at once
        if (small_fit && h->fit_speculate) nm.speculate_max = 1024;
        rc = nm.run(fit_eval, &sev, x0.data(), &xmin[(size_t)lo * n], &fmin[lo]);
        if (rc) return rc;
        for (long p = 0; p < Ps; ++p) its[lo + p] = nm.it[p];
        *f_calls = nm.f_calls + Ps * C;
        *rounds = nm.rounds + 1;
        return 0;
    };
    // Large grids on the small-N path (README.md:227: 111 x 111 delay pairs): the host side of a round -- simplex arithmetic
    // of thousands of problems, packing -- is as long as the kernel, so the problems are cut into slices that run on their
    // own host threads, streams and pinned buffers: one slice's host work overlaps the others' kernels.  Every problem
    // keeps its trajectory (its evaluations do not depend on the batch they travel in): same results as one slice.
    int T = 1;
    if (small_fit && ev.device_unpack && !h->prof) {
        T = h->fit_threads > 0 ? h->fit_threads : (int)std::min<long>(GPCC_MAX_LANES, P / 2048);
        if (T < 1) T = 1;
        if (T > P) T = (int)P;
    }
    long long f_calls = 0, rounds = 0;
    if (T == 1) {
        const int rc = fit_slice(ev, 0, P, &f_calls, &rounds);
        if (rc) return rc;
    } else {
        std::vector<int> rcs(T, 0);
        std::vector<long long> fc(T, 0), rd(T, 0);
        std::vector<std::thread> workers;
        try {
            for (int t = 0; t < T; ++t) {
                FitEval sev = ev;
                sev.lane = t;
                const long lo = P * t / T, hi = P * (t + 1) / T;
                workers.emplace_back([&, sev, lo, hi, t] { rcs[t] = fit_slice(sev, lo, hi, &fc[t], &rd[t]); });
            }
        } catch (...) {   // thread creation failed: run what is missing here (the library never throws across the C ABI)
            for (auto &w : workers) w.join();
            const int started = (int)workers.size();
            workers.clear();
            for (int t = started; t < T; ++t) {
                FitEval sev = ev;
                sev.lane = t;
                rcs[t] = fit_slice(sev, P * t / T, P * (t + 1) / T, &fc[t], &rd[t]);
            }
        }
        for (auto &w : workers) w.join();
        for (int t = 0; t < T; ++t) {
            if (rcs[t]) return rcs[t];
            f_calls += fc[t];
            rounds = std::max(rounds, rd[t]);   // slices advance side by side
        }
    }
    for (int g = 0; g < G; ++g) {
        long pick = (long)g * R;                                     // best restart, :224
        for (int r = 1; r < R; ++r)
            if (fmin[(long)g * R + r] < fmin[pick]) pick = (long)g * R + r;
        loglik_out[g] = -fmin[pick];                                 // :351
        gpcc_unpack_params(1, L, &xmin[pick * n], rhomin, rhomax, alpha_out + (long)g * L, rho_out + g);
        info_out[g] = std::isfinite(fmin[pick]) ? 0 : 1;             // 1: no candidate and no simplex vertex was valid
        if (iterations_out) iterations_out[g] = its[pick];
    }
    if (stats_out) {
        stats_out[0] = f_calls;
        stats_out[1] = rounds;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------
// Multi-device handles: the delay grid shards over the devices of ONE process -- the shape that fits
// a Julia host (README.md:181-211, :258-287 parallelise the same map with pmap workers).  Static contiguous block
// partition of every batch, light curves replicated at create, no data-path collective, and ONE all-gather of
// [loglik | info] per call (RCCL over xGMI: ncclCommInitAll + ncclAllGather inside a group), after which every device
// holds the whole vector.  RCCL refuses a communicator with a repeated device, so device lists with duplicates (the
// one-GPU rehearsal {0, 0}) and single-device lists gather through host memory instead (gather_mode says which).
// ------------------------------------------------------------------------------------------
__global__ void gpcc_pack_gather(int cnt, long blk, const int *info, double *send)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= blk) return;
    if (i >= cnt) send[i] = __builtin_nan("");                 // padding of the last block
    send[blk + i] = (i < cnt) ? (double)info[i] : 0.0;         // info travels as a double: one collective, one dtype
}

static int multi_fail(gpcc_handle_t h, gpcc_handle_t sub, int rc)
{
    return fail(h, rc, "device %d: %s", sub->device, sub->err.c_str());
}

static void multi_free_buffers(gpcc_handle_t h)
{
    for (size_t i = 0; i < h->subs.size(); ++i) {
        DeviceGuard g(nullptr, h->subs[i]->device);
        if (i < h->d_send.size()) hipFree(h->d_send[i]);
        if (i < h->d_recv.size()) hipFree(h->d_recv[i]);
    }
    h->d_send.clear();
    h->d_recv.clear();
    h->gather_cap = 0;
}

// gather buffers for blocks of `blk` evaluations of `width` doubles each (2: [loglik | info] of a batch; L + 4: a fitted delay)
static int multi_ensure_buffers(gpcc_handle_t h, long blk, int width = 2)
{
    const long need = blk * width;
    if (need <= h->gather_cap) return 0;
    multi_free_buffers(h);
    const size_t n = h->subs.size();
    h->d_send.assign(n, nullptr);
    h->d_recv.assign(n, nullptr);
    for (size_t i = 0; i < n; ++i) {
        GPCC_ON_DEVICE(h, h->subs[i]->device);
        HIPCHK(h, hipMalloc(&h->d_send[i], sizeof(double) * need));
        HIPCHK(h, hipMalloc(&h->d_recv[i], sizeof(double) * need * n));
    }
    h->h_gather.resize((size_t)need * n);
    h->gather_cap = need;
    return 0;
}

static int multi_destroy(gpcc_handle_t h)
{
    for (auto &w : h->workers) w->shutdown();
    h->workers.clear();
    for (size_t i = 0; i < h->ev_a.size(); ++i) {
        DeviceGuard g(nullptr, h->subs[i]->device);
        hipEventDestroy(h->ev_a[i]);
        hipEventDestroy(h->ev_b[i]);
    }
    for (size_t i = 0; i < h->comms.size(); ++i) {
        DeviceGuard g(nullptr, h->subs[i]->device);
        ncclCommDestroy(h->comms[i]);
    }
    multi_free_buffers(h);
    for (gpcc_handle_t sub : h->subs) gpcc_destroy(sub);
    delete h;
    return 0;
}

extern "C" int gpcc_create_multi(gpcc_handle_t *out, int L, const int *Nl, const double *t, const double *y,
                                 const double *sigma, int kernel_id, int marginalise_b, int precision,
                                 const int *device_ids, int n_devices)
{
    if (!out) return fail(nullptr, GPCC_ERR_ARGUMENT, "handle pointer is NULL");
    *out = nullptr;
    if (!device_ids || n_devices < 1 || n_devices > 64)
        return fail(nullptr, GPCC_ERR_ARGUMENT, "device_ids is NULL or n_devices=%d outside [1,64]", n_devices);
    gpcc_handle_t h = new gpcc_handle_s();
    bool distinct = true;
    for (int i = 0; i < n_devices; ++i)
        for (int j = 0; j < i; ++j) distinct = distinct && device_ids[i] != device_ids[j];
    for (int i = 0; i < n_devices; ++i) {
        gpcc_handle_t sub = nullptr;
        const int rc = gpcc_create(&sub, L, Nl, t, y, sigma, kernel_id, marginalise_b, precision, device_ids[i]);
        if (rc) {   // g_err holds gpcc_create's message
            const std::string msg = g_err;
            multi_destroy(h);
            return fail(nullptr, rc, "device_ids[%d] = %d: %s", i, device_ids[i], msg.c_str());
        }
        h->subs.push_back(sub);
    }
    h->device = device_ids[0];
    h->gather_mode = GPCC_GATHER_HOST;
    // (GPCC_MULTI_FORCE_RCCL=1 takes the RCCL route also for a single device: a one-rank communicator -- how the one-GPU box
    // exercises ncclCommInitAll / ncclAllGather / ncclCommDestroy of this library at all)
    const char *force = getenv("GPCC_MULTI_FORCE_RCCL");
    if (distinct && (n_devices > 1 || (force && force[0] == '1'))) {
        h->comms.assign(n_devices, nullptr);
        const ncclResult_t r = ncclCommInitAll(h->comms.data(), n_devices, device_ids);
        if (r != ncclSuccess) {
            h->comms.clear();
            multi_destroy(h);
            return fail(nullptr, GPCC_ERR_HIP, "ncclCommInitAll over %d devices failed: %s", n_devices, ncclGetErrorString(r));
        }
        h->gather_mode = GPCC_GATHER_RCCL;
    }
    *out = h;
    return 0;
}

// events around every device's share (statistics) and the persistent worker threads, one per device: created on first use
static int multi_start(gpcc_handle_t h)
{
    const int n = (int)h->subs.size();
    if (h->ev_a.empty()) {
        // created into locals and published only when ALL exist: a failure half-way must not leave null events behind for later calls
        std::vector<hipEvent_t> ea(n, nullptr), eb(n, nullptr);
        hipError_t ce = hipSuccess;
        for (int i = 0; i < n && ce == hipSuccess; ++i) {
            DeviceGuard dg(nullptr, h->subs[i]->device);
            ce = hipEventCreate(&ea[i]);
            if (ce == hipSuccess) ce = hipEventCreate(&eb[i]);
        }
        if (ce != hipSuccess) {
            for (int i = 0; i < n; ++i) {
                DeviceGuard dg(nullptr, h->subs[i]->device);
                if (ea[i]) hipEventDestroy(ea[i]);
                if (eb[i]) hipEventDestroy(eb[i]);
            }
            return fail(h, GPCC_ERR_HIP, "hipEventCreate: %s", hipGetErrorString(ce));
        }
        h->ev_a.swap(ea);
        h->ev_b.swap(eb);
        h->stat_compute_ms.assign(n, 0.0);
    }
    if (h->workers.empty() && !h->workers_failed && n > 1) {
        try {
            for (int i = 0; i < n; ++i) {
                h->workers.emplace_back(new MultiWorker());
                MultiWorker *w = h->workers.back().get();
                w->th = std::thread([w] { w->loop(); });
            }
        } catch (...) {   // no threads to be had: fall back to the calling thread (the library never throws across the C ABI)
            for (auto &w : h->workers) w->shutdown();
            h->workers.clear();
            h->workers_failed = true;
        }
    }
    return 0;
}

// one device's share of a batch: evaluations [lo, lo + cnt) -> d_send[i] = [loglik(blk) | info(blk)], stream-ordered
static int multi_worker(gpcc_handle_t h, int i, long blk, long lo, int cnt, const double *delays, const double *alpha,
                        const double *rho)
{
    gpcc_handle_t sub = h->subs[i];
    GPCC_ON_DEVICE(sub, sub->device);
    int rc = ensure_staging(sub, cnt > 0 ? cnt : 1);
    if (rc) return rc;
    HIPCHK(sub, hipEventRecord(h->ev_a[i], sub->main_stream));
    if (cnt > 0) {
        rc = enqueue_host_batch(sub, cnt, delays + lo * sub->L, alpha + lo * sub->L, rho + lo, h->d_send[i], sub->d_oinfo);
        if (rc) return rc;
    }
    gpcc_pack_gather<<<(unsigned)((blk + 255) / 256), 256, 0, sub->main_stream>>>(cnt, blk, sub->d_oinfo, h->d_send[i]);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(sub, GPCC_ERR_HIP, "gpcc_pack_gather: %s", hipGetErrorString(e));
    HIPCHK(sub, hipEventRecord(h->ev_b[i], sub->main_stream));
    if (h->gather_mode == GPCC_GATHER_HOST) {   // this device's block -> its place in the host gather buffer
        HIPCHK(sub, hipMemcpyAsync(h->h_gather.data() + (size_t)i * 2 * blk, h->d_send[i], sizeof(double) * 2 * blk,
                                   hipMemcpyDeviceToHost, sub->main_stream));
        HIPCHK(sub, hipStreamSynchronize(sub->main_stream));
    }
    return 0;
}

static int multi_loglik_batch(gpcc_handle_t h, int M, const double *delays, const double *alpha, const double *rho,
                              double *loglik, int *info)
{
    const int n = (int)h->subs.size();
    const long blk = ((long)M + n - 1) / n;
    int rc = multi_ensure_buffers(h, blk);
    if (rc) return rc;
    const auto t_begin = std::chrono::steady_clock::now();
    rc = multi_start(h);
    if (rc) return rc;
    std::vector<int> rcs(n, 0);
    for (int i = 0; i < n; ++i) {
        const long lo = (long)i * blk;
        const int cnt = (int)std::max(0L, std::min(blk, (long)M - lo));
        if (!h->workers.empty()) h->workers[i]->submit([=] { return multi_worker(h, i, blk, lo, cnt, delays, alpha, rho); });
        else rcs[i] = multi_worker(h, i, blk, lo, cnt, delays, alpha, rho);
    }
    if (!h->workers.empty())
        for (int i = 0; i < n; ++i) rcs[i] = h->workers[i]->wait();
    for (int i = 0; i < n; ++i)
        if (rcs[i]) {
            // a share failed: the other devices' streams still hold enqueued work that reads the caller's buffers and the
            // handle's staging -- drain them all before handing control back
            for (int j = 0; j < n; ++j) {
                DeviceGuard g(nullptr, h->subs[j]->device);
                hipStreamSynchronize(h->subs[j]->main_stream);
            }
            return multi_fail(h, h->subs[i], rcs[i]);
        }
    const auto t_gather = std::chrono::steady_clock::now();
    h->gather_blk = blk;
    h->gather_width = 2;
    const size_t nb = sizeof(double) * 2 * blk;
    if (h->gather_mode == GPCC_GATHER_RCCL) {
        // THE collective of the path: every device contributes its block and receives all of them
        ncclResult_t r = ncclGroupStart();
        for (int i = 0; i < n && r == ncclSuccess; ++i)
            r = ncclAllGather(h->d_send[i], h->d_recv[i], (size_t)2 * blk, ncclDouble, h->comms[i], h->subs[i]->main_stream);
        const ncclResult_t r2 = ncclGroupEnd();
        if (r == ncclSuccess) r = r2;
        if (r != ncclSuccess) return fail(h, GPCC_ERR_HIP, "ncclAllGather failed: %s", ncclGetErrorString(r));
        for (int i = 0; i < n; ++i) {
            GPCC_ON_DEVICE(h, h->subs[i]->device);
            HIPCHK(h, hipStreamSynchronize(h->subs[i]->main_stream));
        }
        GPCC_ON_DEVICE(h, h->subs[0]->device);
        HIPCHK(h, hipMemcpy(h->h_gather.data(), h->d_recv[0], nb * n, hipMemcpyDeviceToHost));
    } else {
        // same result through host memory: the blocks are already in h_gather; every device gets the whole vector
        for (int i = 0; i < n; ++i) {
            GPCC_ON_DEVICE(h, h->subs[i]->device);
            HIPCHK(h, hipMemcpyAsync(h->d_recv[i], h->h_gather.data(), nb * n, hipMemcpyHostToDevice, h->subs[i]->main_stream));
        }
        for (int i = 0; i < n; ++i) {
            GPCC_ON_DEVICE(h, h->subs[i]->device);
            HIPCHK(h, hipStreamSynchronize(h->subs[i]->main_stream));
        }
    }
    for (int i = 0; i < n; ++i) {
        const long lo = (long)i * blk;
        const long cnt = std::max(0L, std::min(blk, (long)M - lo));
        const double *src = h->h_gather.data() + (size_t)i * 2 * blk;
        for (long j = 0; j < cnt; ++j) {
            loglik[lo + j] = src[j];
            info[lo + j] = (int)src[blk + j];
        }
    }
    const auto t_end = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) {   // every stream has drained: the events are complete
        float ms = 0.f;
        GPCC_ON_DEVICE(h, h->subs[i]->device);
        h->stat_compute_ms[i] = hipEventElapsedTime(&ms, h->ev_a[i], h->ev_b[i]) == hipSuccess ? ms : -1.0;
    }
    h->stat_gather_ms = std::chrono::duration<double, std::milli>(t_end - t_gather).count();
    h->stat_total_ms = std::chrono::duration<double, std::milli>(t_end - t_begin).count();
    return 0;
}

// ------------------------------------------------------------------------------------------
// The per-delay FIT on a multi-device handle (round 4): sharded BY DELAY, one gather at the end.  The reference parallelises the
// fit over candidate delays (README.md:195-211, :258-287: pmap, one gpcc per worker), and SURVEY 8(e) asks for round-robin chunks
// because iteration counts differ from delay to delay: device i takes the delays g = i, i + n, i + 2n, ... and runs gpcc_grid_loglik
// on ITS single-device handle, on its persistent worker thread -- its own lock-step optimiser with every fast path of a
// single-device fit (device-side unpack, speculative rounds, threaded slices on the small-N path).  Every delay sees the same random
// candidates (each reference gpcc call seeds its own generator) and its evaluations do not depend on what else is in a batch, so a
// delay's result is the single-device fit's bit for bit (small-N path; to rounding where the tile kernels pick their path by group
// size).  Then ONE collective: every device contributes its block of [loglik | info | iterations | rho | alpha(L)] rows and receives
// all of them (RCCL all-gather over xGMI; through host memory when device ids repeat), as after a batch (gpcc_multi_gathered).
// Until round 3 every optimiser ROUND was a sharded batch: n thread hand-offs, n stream syncs and a gather 150-800 times per fit.
// ------------------------------------------------------------------------------------------
static int multi_grid_loglik(gpcc_handle_t h, int G, const double *delays, int iterations, int R, int C, double rhomin, double rhomax,
                            const double *cands, double *loglik_out, double *alpha_out, double *rho_out, int *info_out,
                            int *iterations_out, long long *stats_out)
{
    const int n = (int)h->subs.size(), L = h->subs[0]->L, W = L + 4;
    const long blk = ((long)G + n - 1) / n;
    int rc = multi_ensure_buffers(h, blk, W);
    if (rc) return rc;
    const auto t_begin = std::chrono::steady_clock::now();
    rc = multi_start(h);
    if (rc) return rc;
    struct Share {
        std::vector<double> delays, ll, alpha, rho, row;
        std::vector<int> info, its;
        long long stats[2] = {0, 0};
        double ms = 0.0;
    };
    std::vector<Share> sh(n);
    std::vector<int> rcs(n, 0);
    for (int i = 0; i < n; ++i) {
        Share &s = sh[i];
        const long Gi = gpccfit::deal_count(G, n, i);
        s.delays.resize((size_t)Gi * L); s.ll.resize(Gi); s.alpha.resize((size_t)Gi * L); s.rho.resize(Gi); s.info.resize(Gi); s.its.resize(Gi);
        for (long j = 0; j < Gi; ++j) memcpy(&s.delays[(size_t)j * L], delays + (size_t)gpccfit::deal_global(n, i, j) * L, sizeof(double) * L);
        auto job = [h, i, Gi, iterations, R, C, rhomin, rhomax, cands, blk, W, L, &sh]() -> int {
            Share &s = sh[i];
            gpcc_handle_t sub = h->subs[i];
            const auto t0 = std::chrono::steady_clock::now();
            int r = Gi > 0 ? gpcc_grid_loglik(sub, (int)Gi, s.delays.data(), iterations, R, C, rhomin, rhomax, 0, cands, s.ll.data(),
                                              s.alpha.data(), s.rho.data(), s.info.data(), s.its.data(), s.stats)
                           : 0;
            if (r) return r;
            s.ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            // this device's block of the gather: rows [loglik | info | iterations | rho | alpha(L)], padding NaN / 0
            s.row.assign((size_t)blk * W, 0.0);
            gpccfit::deal_pack_rows(blk, L, Gi, s.ll.data(), s.info.data(), s.its.data(), s.rho.data(), s.alpha.data(), s.row.data());
            GPCC_ON_DEVICE(sub, sub->device);
            HIPCHK(sub, hipMemcpyAsync(h->d_send[i], s.row.data(), sizeof(double) * blk * W, hipMemcpyHostToDevice, sub->main_stream));
            if (h->gather_mode == GPCC_GATHER_HOST) memcpy(h->h_gather.data() + (size_t)i * blk * W, s.row.data(), sizeof(double) * blk * W);
            HIPCHK(sub, hipStreamSynchronize(sub->main_stream));
            return 0;
        };
        if (!h->workers.empty()) h->workers[i]->submit(job);
        else rcs[i] = job();
    }
    if (!h->workers.empty())
        for (int i = 0; i < n; ++i) rcs[i] = h->workers[i]->wait();
    for (int i = 0; i < n; ++i)
        if (rcs[i]) return multi_fail(h, h->subs[i], rcs[i]);   // (gpcc_grid_loglik synchronises before it returns: nothing is left running)
    const auto t_gather = std::chrono::steady_clock::now();
    h->gather_blk = blk;
    h->gather_width = W;
    const size_t nb = sizeof(double) * blk * W;
    if (h->gather_mode == GPCC_GATHER_RCCL) {
        ncclResult_t r = ncclGroupStart();
        for (int i = 0; i < n && r == ncclSuccess; ++i)
            r = ncclAllGather(h->d_send[i], h->d_recv[i], (size_t)blk * W, ncclDouble, h->comms[i], h->subs[i]->main_stream);
        const ncclResult_t r2 = ncclGroupEnd();
        if (r == ncclSuccess) r = r2;
        if (r != ncclSuccess) return fail(h, GPCC_ERR_HIP, "ncclAllGather failed: %s", ncclGetErrorString(r));
        for (int i = 0; i < n; ++i) {
            GPCC_ON_DEVICE(h, h->subs[i]->device);
            HIPCHK(h, hipStreamSynchronize(h->subs[i]->main_stream));
        }
        GPCC_ON_DEVICE(h, h->subs[0]->device);
        HIPCHK(h, hipMemcpy(h->h_gather.data(), h->d_recv[0], nb * n, hipMemcpyDeviceToHost));
    } else {
        for (int i = 0; i < n; ++i) {
            GPCC_ON_DEVICE(h, h->subs[i]->device);
            HIPCHK(h, hipMemcpyAsync(h->d_recv[i], h->h_gather.data(), nb * n, hipMemcpyHostToDevice, h->subs[i]->main_stream));
        }
        for (int i = 0; i < n; ++i) {
            GPCC_ON_DEVICE(h, h->subs[i]->device);
            HIPCHK(h, hipStreamSynchronize(h->subs[i]->main_stream));
        }
    }
    long long f_calls = 0, rounds = 0;
    gpccfit::deal_scatter(G, n, L, blk, h->h_gather.data(), loglik_out, info_out, iterations_out, rho_out, alpha_out);
    for (int i = 0; i < n; ++i) {
        f_calls += sh[i].stats[0];
        rounds = std::max(rounds, sh[i].stats[1]);   // the devices advance side by side
        h->stat_compute_ms[i] = sh[i].ms;            // (host wall clock of the device's whole fit)
    }
    if (stats_out) {
        stats_out[0] = f_calls;
        stats_out[1] = rounds;
    }
    const auto t_end = std::chrono::steady_clock::now();
    h->stat_gather_ms = std::chrono::duration<double, std::milli>(t_end - t_gather).count();
    h->stat_total_ms = std::chrono::duration<double, std::milli>(t_end - t_begin).count();
    return 0;
}

// per-device time of the last gpcc_loglik_batch on a multi-device handle (HIP events on each device's stream around its
// share), the time of the gather phase (all-gather + final copies; in the host-gather mode the device-to-host copies run
// inside the shares) and of the whole call (host wall clock): one run of a multi-GPU benchmark then shows WHERE a scaling
// loss comes from -- an uneven share, a slow device, or the collective
extern "C" int gpcc_multi_stats(gpcc_handle_t h, double *compute_ms, double *gather_ms, double *total_ms)
{
    if (!h || !h->is_multi()) return fail(h, GPCC_ERR_ARGUMENT, "not a multi-device handle");
    if (h->stat_compute_ms.empty()) return fail(h, GPCC_ERR_STATE, "no batch has run on this handle yet");
    if (compute_ms) memcpy(compute_ms, h->stat_compute_ms.data(), sizeof(double) * h->stat_compute_ms.size());
    if (gather_ms) *gather_ms = h->stat_gather_ms;
    if (total_ms) *total_ms = h->stat_total_ms;
    return 0;
}

// the gathered vector as it sits on device `which` of the handle after the last gpcc_loglik_batch (tests; a host that
// wants getprobabilities on a device of its choice): [loglik(blk) | info(blk)] per device block, n blocks
extern "C" int gpcc_multi_gathered(gpcc_handle_t h, int which, long *blk_out, double *out, long capacity)
{
    if (!h || !h->is_multi()) return fail(h, GPCC_ERR_ARGUMENT, "not a multi-device handle");
    if (which < 0 || which >= (int)h->subs.size()) return fail(h, GPCC_ERR_ARGUMENT, "device index %d outside [0,%d)", which, (int)h->subs.size());
    const long total = h->gather_width * h->gather_blk * (long)h->subs.size();
    if (blk_out) *blk_out = h->gather_blk;
    if (!out) return 0;
    if (capacity < total) return fail(h, GPCC_ERR_ARGUMENT, "capacity %ld < %ld", capacity, total);
    GPCC_ON_DEVICE(h, h->subs[which]->device);
    HIPCHK(h, hipMemcpy(out, h->d_recv[which], sizeof(double) * total, hipMemcpyDeviceToHost));
    return 0;
}

// ------------------------------------------------------------------------------------------
static void prof_collect(gpcc_handle_t h)
{
    if (h->recs.empty()) return;
    DeviceGuard guard_(nullptr, h->device);
    hipDeviceSynchronize();
    for (auto &r : h->recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { h->prof_n[r.which] += 1; h->prof_ms[r.which] += ms; }
        hipEventDestroy(r.a);
        hipEventDestroy(r.b);
    }
    h->recs.clear();
}

// the stamps of the last persistent launch on workspace stream 0 (option "chain_trace"), microseconds from the first one
extern "C" int gpcc_chain_trace(gpcc_handle_t h, int evaluation, double *out_us, int capacity)
{
    if (!h || !out_us) return fail(h, GPCC_ERR_ARGUMENT, "NULL pointer");
    h = primary(h);
    if (!h->d_chain_trace) return fail(h, GPCC_ERR_STATE, "no trace: set option \"chain_trace\" to 1 before the evaluation");
    const int W = GPCC_CHAIN_TRACE_WORDS;
    if (evaluation < 0 || evaluation >= GPCC_CHAIN_MAX_EVALS || capacity < W * h->nt) return fail(h, GPCC_ERR_ARGUMENT, "evaluation %d / capacity %d (need %d nt = %d)", evaluation, capacity, W, W * h->nt);
    GPCC_ON_DEVICE(h, h->device);
    HIPCHK(h, hipDeviceSynchronize());
    std::vector<unsigned long long> st((size_t)W * h->nt);
    HIPCHK(h, hipMemcpy(st.data(), h->d_chain_trace + (long)evaluation * h->nt * W, sizeof(unsigned long long) * st.size(), hipMemcpyDeviceToHost));
    int khz = 100000;
    (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, h->device);
    unsigned long long t0 = ~0ull;
    for (unsigned long long v : st) if (v && v < t0) t0 = v;
    for (size_t i = 0; i < st.size(); ++i) out_us[i] = st[i] ? (double)(st[i] - t0) * 1e3 / khz : -1.0;
    return 0;
}

// the job stamps of the last persistent launch on workspace stream 0: rows [kind (1 solve, 2 update), step, index in the step, fetched, ready, done] (us)
extern "C" int gpcc_chain_jobs_trace(gpcc_handle_t h, double *out, long capacity_rows, long *rows_out)
{
    if (!h || !out || !rows_out) return fail(h, GPCC_ERR_ARGUMENT, "NULL pointer");
    h = primary(h);
    if (!h->d_chain_trace) return fail(h, GPCC_ERR_STATE, "no trace: set option \"chain_trace\" to 1 before the evaluation");
    GPCC_ON_DEVICE(h, h->device);
    HIPCHK(h, hipDeviceSynchronize());
    unsigned cnt = 0;
    HIPCHK(h, hipMemcpy(&cnt, h->d_chain_words + 1, sizeof(unsigned), hipMemcpyDeviceToHost));
    long n = cnt < GPCC_CHAIN_WTRACE_CAP ? cnt : GPCC_CHAIN_WTRACE_CAP;
    if (n > capacity_rows) n = capacity_rows;
    std::vector<unsigned long long> st(4 * (size_t)(n > 0 ? n : 1));
    const unsigned long long *src = h->d_chain_trace + (size_t)h->ws_streams * GPCC_CHAIN_MAX_EVALS * h->nt * GPCC_CHAIN_TRACE_WORDS;
    if (n > 0) HIPCHK(h, hipMemcpy(st.data(), src, sizeof(unsigned long long) * 4 * n, hipMemcpyDeviceToHost));
    int khz = 100000;
    (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, h->device);
    unsigned long long t0 = 0;   // the same origin as gpcc_chain_trace(evaluation 0): the first stamp of its chain
    HIPCHK(h, hipMemcpy(&t0, h->d_chain_trace, sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (!t0) {
        t0 = ~0ull;
        for (long i = 0; i < n; ++i) if (st[4 * i + 1] && st[4 * i + 1] < t0) t0 = st[4 * i + 1];
    }
    for (long i = 0; i < n; ++i) {
        out[6 * i] = (double)(st[4 * i] >> 56);
        out[6 * i + 1] = (double)((st[4 * i] >> 32) & 0xffffff);
        out[6 * i + 2] = (double)(st[4 * i] & 0xffffffffu);
        for (int e = 1; e < 4; ++e) out[6 * i + 2 + e] = st[4 * i + e] ? ((double)st[4 * i + e] - (double)t0) * 1e3 / khz : -1.0;
    }
    *rows_out = n;
    return 0;
}

extern "C" int gpcc_profile_enable(gpcc_handle_t h, int on)
{
    if (!h) return fail(h, GPCC_ERR_ARGUMENT, "NULL handle");
    h = primary(h);
    prof_collect(h);
    h->prof = on != 0;
    return 0;
}

extern "C" int gpcc_profile_reset(gpcc_handle_t h)
{
    if (!h) return fail(h, GPCC_ERR_ARGUMENT, "NULL handle");
    h = primary(h);
    prof_collect(h);
    for (int i = 0; i < GPCC_PROF_COUNT; ++i) { h->prof_n[i] = 0; h->prof_ms[i] = 0.0; }
    return 0;
}

extern "C" int gpcc_profile_get(gpcc_handle_t h, int which, long *launches, double *total_ms)
{
    if (!h || which < 0 || which >= GPCC_PROF_COUNT) return fail(h, GPCC_ERR_ARGUMENT, "bad argument");
    h = primary(h);
    prof_collect(h);
    if (launches) *launches = h->prof_n[which];
    if (total_ms) *total_ms = h->prof_ms[which];
    return 0;
}

// ------------------------------------------------------------------------------------------
extern "C" int gpcc_selftest(int device_id, double *tflops)
{
    GPCC_ON_DEVICE(nullptr, device_id);
    int rc = 0;
    double hA[64], hB[64], hD[256], ref[256];
    for (int i = 0; i < 16; ++i)
        for (int kk = 0; kk < 4; ++kk) hA[i * 4 + kk] = (double)(1 + i * 5 + kk * 3);  // asymmetric integers
    for (int kk = 0; kk < 4; ++kk)
        for (int j = 0; j < 16; ++j) hB[kk * 16 + j] = (double)(2 + kk * 7 - j * 2);
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double s = 0.0;
            for (int kk = 0; kk < 4; ++kk) s += hA[i * 4 + kk] * hB[kk * 16 + j];
            ref[i * 16 + j] = s;
        }
    double *d = nullptr;
    HIPCHK(nullptr, hipMalloc(&d, sizeof(double) * (64 + 64 + 256)));
    hipError_t e = hipMemcpy(d, hA, sizeof hA, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + 64, hB, sizeof hB, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        gpcc_selftest_map<<<1, 64>>>(d, d + 64, d + 128);
        e = hipMemcpy(hD, d + 128, sizeof hD, hipMemcpyDeviceToHost);
    }
    if (e != hipSuccess) { hipFree(d); return fail(nullptr, GPCC_ERR_HIP, "selftest: %s", hipGetErrorString(e)); }
    int bad = 0;
    for (int i = 0; i < 256; ++i)
        if (hD[i] != ref[i]) ++bad;
    if (bad) { hipFree(d); return fail(nullptr, GPCC_ERR_STATE, "f64 MFMA fragment map mismatch in %d of 256 elements", bad); }
    {   // the same check for v_mfma_f32_16x16x4_f32 (different C/D register map)
        float fA[64], fB[64], fD[256];
        for (int i = 0; i < 64; ++i) { fA[i] = (float)hA[i]; fB[i] = (float)hB[i]; }
        float *df = (float *)d;
        e = hipMemcpy(df, fA, sizeof fA, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(df + 64, fB, sizeof fB, hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            gpcc_selftest_map_f32<<<1, 64>>>(df, df + 64, df + 128);
            e = hipMemcpy(fD, df + 128, sizeof fD, hipMemcpyDeviceToHost);
        }
        if (e != hipSuccess) { hipFree(d); return fail(nullptr, GPCC_ERR_HIP, "selftest: %s", hipGetErrorString(e)); }
        for (int i = 0; i < 256; ++i)
            if ((double)fD[i] != ref[i]) ++bad;
        if (bad) { hipFree(d); return fail(nullptr, GPCC_ERR_STATE, "f32 MFMA fragment map mismatch in %d of 256 elements", bad); }
    }
    if (tflops) {
        const int iters = 20000, blocks = 2048;
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        gpcc_selftest_rate<<<blocks, 256>>>(d, 100);  // warm-up
        hipEventRecord(a, 0);
        gpcc_selftest_rate<<<blocks, 256>>>(d, iters);
        hipEventRecord(b, 0);
        e = hipEventSynchronize(b);
        float ms = 0.f;
        hipEventElapsedTime(&ms, a, b);
        hipEventDestroy(a); hipEventDestroy(b);
        const double flops = (double)blocks * 4 /*waves*/ * iters * 4 /*mfma*/ * 2048.0;
        *tflops = (e == hipSuccess && ms > 0) ? flops / (ms * 1e-3) / 1e12 : 0.0;
    }
    hipFree(d);
    return 0;
}
