// gpcc_kernels.hip.h -- device code of libgpcc_hip.so (gfx950 / CDNA4 only).
//
// Data layout in HBM (DESIGN.md "Layout"): every evaluation owns one SLOT.  A slot stores the
// lower triangle of its padded Np x Np matrix (Np = nt*128) as nt(nt+1)/2 TILES of 128x128
// elements (fp64 or fp32); tile (I,J), J<=I, sits at ((I(I+1)/2)+J)*16384 elements.  Inside a tile the
// 128 columns are cut into CHUNKS of 16 KiB: 128 rows x 16 doubles (resp. 32 floats), row-major = one
// contiguous block that LDS-DMA copies 1:1 into LDS and from which the MFMA fragments are read with two
// ds_read_b128 per fragment.  Inside a row (128 bytes) the eight 16-byte slots are XOR-swizzled (gpcc_sw)
// so that those reads are bank-conflict-free; the swizzle lives in the HBM layout, so the DMA stays a
// linear copy.  A row of tiles (I,0..I) is contiguous, so both operand streams of the left-looking update
// are purely sequential reads.
#pragma once
// (non-template kernels are `static`: this header is compiled into several objects, gpcc.jl_amd/build.py)
#include <hip/hip_runtime.h>
#include <math.h>

#define GPCC_TILE 128
#define GPCC_TILE_ELEMS (GPCC_TILE * GPCC_TILE) /* 16384 elements */
#define GPCC_CHUNK_BYTES 16384
#define GPCC_MAXL 8
#define GPCC_MAXRHS (GPCC_MAXL + 1)
#define GPCC_DIAG_THREADS 512
#define GPCC_DIAG_LD 130
#define GPCC_DINV_LD 17
#define GPCC_DIAG_LDS_BYTES \
    ((GPCC_TILE * GPCC_DIAG_LD + 8 * 16 * GPCC_DINV_LD + GPCC_MAXRHS * GPCC_TILE + 2 * GPCC_TILE + GPCC_MAXRHS * GPCC_MAXRHS + 3) * 8 + 16)
static_assert(GPCC_DIAG_LDS_BYTES <= 160 * 1024, "gpcc_diag_factor's LDS image must fit the 160 KiB of a gfx950 CU");
#define GPCC_GEMM_LDS_BYTES (2 * 2 * GPCC_CHUNK_BYTES)
#define GPCC_TRSM_ROWS_LDS_BYTES (4 * (GPCC_CHUNK_BYTES / 4 + GPCC_CHUNK_BYTES))   /* four stages of 4 KiB + 16 KiB */
#define GPCC_GEMM_THREADS 512
#define GPCC_RIGHT_LOOKING_MAX 12   /* larger groups: left-looking with a right-looking tail (gpcc_hip.hip: hybrid tail) */

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

// Precision traits.  Both element types share the byte geometry (16 KiB chunks, 128-byte rows, two
// 16-byte slots per fragment); they differ in the MFMA instruction and its C/D register map.
template <typename T> struct GpccPrec;
template <> struct GpccPrec<double> {
    typedef d4 acc_t;   // 16x16 accumulator fragment: 4 values per lane
    typedef d2 v16;     // one 16-byte LDS slot
    static constexpr int KC = 16, EP = 2, KSTEPS = 4, NCH = 8;  // cols/chunk, elems/16 B, MFMA k-steps/chunk, chunks/tile
    static __device__ __forceinline__ acc_t mfma(double a, double b, acc_t c)
    {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    // v_mfma_f64_16x16x4_f64 C/D: col = lane&15, row = (lane>>4) + 4*reg   (gpcc_selftest checks it)
    static __device__ __forceinline__ int crow(int q, int r) { return q + 4 * r; }
};
template <> struct GpccPrec<float> {
    typedef f4 acc_t;
    typedef f4 v16;
    static constexpr int KC = 32, EP = 4, KSTEPS = 8, NCH = 4;
    static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c)
    {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    // v_mfma_f32_16x16x4_f32 C/D: col = lane&15, row = 4*(lane>>4) + reg
    static __device__ __forceinline__ int crow(int q, int r) { return 4 * q + r; }
};

struct GpccCtx {
    void *tiles;     // slots x slot_stride elements (double or float)
    void *linv;      // slots x 16384 : inverse of the current diagonal block, tile layout
    double *z;       // slots x nrhs x Np : running right-hand sides  R - L[:, :k] W[:k]      (always fp64)
    double *w;       // slots x nrhs x Np : W = L^-1 R
    double *logdet;  // slots             : sum_i log L_ii
    double *gram;    // slots x nrhs^2    : W^T W  (nrhs = 1: |w|^2 = sqmahal)
    int *info;       // slots
    double *gpart;   // slots x nt(nt+1)/2 x nrhs^2 : per-tile partial sums of X' K0 X (fp32 refinement, gpcc_refine_partials)
    int linv_keep;   // 1: linv holds ALL nt inverses of a slot (slot*nt + k), kept for the backward solve of the fp32
                     //    refinement; 0: one tile per slot, overwritten every step
    double *kdiag;   // slots x Np : diag(K) as assembled, fp64 (fp32 mode only: numerator of the pivot ratios below)
    double *cond;    // slots x 2  : sum_i K_ii / d_i and max_i K_ii / d_i over the pivots d_i (fp32 mode only) -- the
                     //              a-posteriori conditioning measure behind the fp64 re-evaluation, DESIGN.md 4.7
    double tmid;     // midpoint of the observation times: centre of the separable-exponential form (gpcc_sep_point)
    double *sep;     // slots x 4 x Np : shifted times u, separable factors A, B, amplitudes a of every point (gpcc_sep_points; fold only)
    double *seps;    // slots x 4      : the kernel's scale of the distance s (gpcc_kernel_scale), its constants c1, c2 (gpcc_kernel_const)
    int *sepflag;    // slots x nt     : low byte: b + 1 if tile row I lies inside ONE band b (no padding), else 0;
                     //                  bit 8: all its points are in the range of the separable form (never set for rbf)
    int fold;        // 1: the off-diagonal tiles inside one band pair are NOT assembled -- gpcc_update_solve evaluates their elements
                     //    into its accumulators (fused left-looking groups; which tiles: gpcc_fold_mode);
                     // 2: the three-kernel path: likewise for tile columns J >= 1, in the first gpcc_panel_update job that touches the tile
    int fold_mixed;  // 1: ... also the tiles of a tile row that straddles two bands or holds padding (gpcc_fold_mode 3) and tiles that
                     //    need the direct evaluation (mode 4: rbf) -- the MIXED ("general") instantiations of the two kernels: the host
                     //    launches those when it sets this (handles with such tile rows, rbf handles)
    const double *t, *sig2, *resid;  // Np (padding: 0)
    const double *yv;                // Np: raw fluxes (only read by explicit 'Y' rows, see band codes)
    const int *band;                 // Np: >= 0 band of a real point; -1 identity padding;
                                     // <= -2 explicit row e = -2-code: e < L: indicator of band e (a column of Q),
                                     // e == L: the flux vector Y  (rows of the augmented systems)
    double sigma_b[GPCC_MAXL];
    long slot_stride;
    int L, N, Np, nt, kernel_id, marginalise_b;
    int nt_fact;   // tile columns that are factorised (== nt for the plain log-likelihood)
    int nrhs;      // 1: R = Y - bbar.  L+1 (woodbury): R = [Q | Y - bbar]
    int share_p;   // > 0: the evaluations of a group share their first share_p tile rows (same band-1 alpha, rho, delay):
                   //      only the group's first slot (the leader) assembles / factorises them, the others read its tiles,
                   //      inv(L_kk) and W_k for k < share_p -- bitwise the same values they would have computed
    int asm32;     // fp32 tiles: elements of tiles inside one band pair are EVALUATED in fp32 too (option "fp32_assemble")
    int store_l;   // gpcc_diag_factor also writes L_kk back (dense factor export); 0 on the log-likelihood path
    int woodbury;  // 1: the matrix is K0 = delayedCovariance + Sobs only; B = Q Sigma_b Q' enters through the
                   //    L x L capacitance matrix in fp64 (determinant lemma + Woodbury) -- the fp32 path
    unsigned *chain_words;          // != NULL: the group goes on into the persistent few-evaluation launch (gpcc_chain.hip.h), whose flag
    int chain_qbase, chain_ev_words;   // words gpcc_assemble_tiles zeroes on the way (header [0, qbase), then ev_words per evaluation)
};

struct GpccGroup {
    const double *delays, *alpha, *rho;  // whole-batch arrays (M x L, M x L, M)
    double *out_loglik;                  // whole-batch outputs
    int *out_info;
    double *out_cond;                    // 2 per evaluation (see GpccCtx::cond) or NULL
    int first;  // index of this group's first evaluation in the batch arrays
    int slot0;  // first slot of the stream that runs this group
    int cnt;    // evaluations in this group
    int spread; // fewer than 8 evaluations: job b -> (evaluation b % cnt, tile b / cnt), i.e. every evaluation's tiles
                // go round all 8 XCDs instead of staying on the one XCD that blockIdx % 8 selects (set by the host)
    // small-N kernels only -- requests of the optimiser in its own coordinates (gpcc_grid_loglik): xpar != NULL means
    // evaluation i has the unconstrained parameter vector xpar[i][0..L], unpacked ON THE DEVICE (`unpack`,
    // marginaliseb.jl:112-126: alpha = makepositive(x[1:L]) + 1e-8, rho = transformbetween(x[L+1], rhomin, rhomax),
    // gpcc_transforms.h) and the delay vector delays[xrow[i]][0..L-1] (a row of the candidate-delay table)
    const double *xpar = nullptr;
    const int *xrow = nullptr;
    double rhomin = 0.0, rhomax = 0.0;
};

__device__ __forceinline__ long gpcc_tile_off(int I, int J)
{
    return ((long)I * (I + 1) / 2 + J) * GPCC_TILE_ELEMS;
}
__device__ __forceinline__ long gpcc_linv_off(const GpccCtx &c, int slot, int k)
{
    return (c.linv_keep == 1 ? ((long)slot * c.nt + k) : (long)slot) * GPCC_TILE_ELEMS;
}
// Slot swizzle of row r (depends on row bits 1..3).  ds_read_b128 is served in 16-lane groups
// {q even, rows 0-3,12-15 | q odd, rows 4-11} (and the mirrored one); g maps the row pairs
// {0,1,6,7} -> {0,2,4,6} and {2,3,4,5} -> {1,3,5,7}, both sets closed under the xor with the slot
// bases 2q, so the 16 lanes of a group hit 16 different 16-byte bank groups of the 256-byte LDS row.
__device__ __forceinline__ int gpcc_sw(int r)
{
    const int p = (r >> 1) & 7;
    return (((p >> 2) & 1) << 2) | ((p & 1) << 1) | (((p >> 2) ^ (p >> 1)) & 1);
}
template <typename T>
__device__ __forceinline__ int gpcc_elem_off(int r, int col)
{
    typedef GpccPrec<T> P;
    const int cc = col % P::KC;
    return (col / P::KC) * (GPCC_TILE * P::KC) + r * P::KC + (((cc / P::EP) ^ gpcc_sw(r)) * P::EP) + (cc % P::EP);
}

// ------------------------------------------------------------------------------------------
// Stationary kernels, /root/reference/src/util.jl:15-52.  The assembly is HBM-write-bound only if
// an element costs ~30 instructions, so: the divisions by rho become multiplications by
// per-evaluation reciprocals (GpccKernelConst), and exp() of a non-positive argument is a
// straight-line Cody-Waite reduction + degree-13 Taylor/Horner + v_ldexp_f64 (|r| <= ln2/2:
// truncation 4e-18, total error ~1 ulp, like libm's).  Each element therefore agrees with the
// reference's expression to a few ulp (tests: 1e-13 relative), not bit for bit.
// ------------------------------------------------------------------------------------------
struct GpccKernelConst { double c1, c2; };

template <int KID>
__device__ __forceinline__ GpccKernelConst gpcc_kernel_const(double rho)
{
    GpccKernelConst kc;
    if (KID == 1) kc.c1 = 1.0 / (2.0 * rho);   // rbf: exp(-0.5 d^2 / (2 rho)) -- rho linear, as the reference
    else kc.c1 = 1.0 / rho;
    kc.c2 = (KID == 3) ? 1.0 / (3.0 * (rho * rho)) : 0.0;
    return kc;
}

__device__ __forceinline__ double gpcc_exp_nonpos(double x)  // x <= 0
{
    const double L2E = 1.4426950408889634074, LN2HI = 6.93147180369123816490e-01, LN2LO = 1.90821492927058770002e-10;
    const double n = rint(x * L2E);
    double r = fma(-n, LN2HI, x);
    r = fma(-n, LN2LO, r);
    double p = 1.6059043836821614599e-10;  // 1/13!
    p = fma(p, r, 2.0876756987868098979e-09);
    p = fma(p, r, 2.5052108385441718775e-08);
    p = fma(p, r, 2.7557319223985890653e-07);
    p = fma(p, r, 2.7557319223985890653e-06);
    p = fma(p, r, 2.4801587301587301587e-05);
    p = fma(p, r, 1.9841269841269841270e-04);
    p = fma(p, r, 1.3888888888888888889e-03);
    p = fma(p, r, 8.3333333333333333333e-03);
    p = fma(p, r, 4.1666666666666666667e-02);
    p = fma(p, r, 1.6666666666666666667e-01);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    const double nn = fmax(n, -1100.0);  // exp underflows to 0 well before; keeps the int conversion defined
    return ldexp(p, (int)nn);
}

// The same to ~1.4 ulp in 11 double-precision operations instead of 19: x = (64 n + j) ln2/64 + r, |r| <= ln2/128,
// exp(x) = 2^n 2^(j/64) p5(r) with the 64 values of 2^(j/64) in LDS (GPCC_EXP_TABLE_TO_LDS at kernel entry) and a degree-5
// polynomial (truncation 4e-17).  Worst case against a 60-digit reference over 2e5 arguments in [-700, 0]: 3.1e-16 relative.
// Measured on one box against the polynomial version (profiles/r03/ab_exp_table_vs_polynomial_same_box.log): the fp32 refinement
// pass 31.3 -> 28.8 ms per 1024 evaluations (its inner loop has LDS bandwidth to spare), but the tile assembly 16.3 -> 17.2 ms and
// the small-N kernels 0.40 -> 0.42 ms (N = 110) / 1.04 -> 1.15 ms (N = 150) -- the per-lane table gather (bank conflicts, the
// f64 -> i32 conversion) costs more there than eight fewer FMAs save.  So: the refinement pass and delayedCovariance use it, the
// assembly and the small-N kernels keep the polynomial.
static __device__ __constant__ double gpcc_exp_tab_c[64] = {   // (static: the header is compiled into several objects)
    1, 1.0108892860517005, 1.0218971486541166, 1.0330248790212284,
    1.0442737824274138, 1.0556451783605572, 1.0671404006768237, 1.0787607977571199,
    1.0905077326652577, 1.1023825833078409, 1.1143867425958924, 1.1265216186082418,
    1.1387886347566916, 1.1511892299529827, 1.1637248587775775, 1.1763969916502812,
    1.189207115002721, 1.2021567314527031, 1.215247359980469, 1.22848053610687,
    1.241857812073484, 1.2553807570246911, 1.2690509571917332, 1.2828700160787783,
    1.2968395546510096, 1.3109612115247644, 1.3252366431597413, 1.3396675240533029,
    1.3542555469368927, 1.3690024229745905, 1.383909881963832, 1.3989796725383112,
    1.4142135623730951, 1.42961333839197, 1.4451808069770467, 1.460917794180647,
    1.4768261459394993, 1.4929077282912648, 1.5091644275934228, 1.5255981507445384,
    1.5422108254079407, 1.5590044002378369, 1.5759808451078865, 1.593142151342267,
    1.6104903319492543, 1.6280274218573478, 1.6457554781539649, 1.6636765803267364,
    1.681792830507429, 1.7001063537185235, 1.7186192981224779, 1.7373338352737062,
    1.7562521603732995, 1.7753764925265212, 1.7947090750031072, 1.8142521755003989,
    1.8340080864093424, 1.8539791250833855, 1.8741676341103, 1.8945759815869656,
    1.9152065613971474, 1.9360617934922943, 1.9571441241754002, 1.9784560263879509};
#define GPCC_EXP_TABLE_TO_LDS(stab, tid)                  \
    do {                                                   \
        if ((tid) < 64) (stab)[(tid)] = gpcc_exp_tab_c[(tid)]; \
    } while (0)

__device__ __forceinline__ double gpcc_exp_nonpos_tab(double x, const double *stab)  // x <= 0; stab: the table in LDS
{
    const double INV = 92.332482616893656768, C1 = 0.010830424696223417413, C2 = 2.5728046223276691076e-14;   // 64/ln2; ln2/64 = C1 + C2
    x = fmax(x, -745.0);                                   // (exp underflows to 0 beyond; keeps the int conversion defined)
    const double kd = rint(x * INV);
    double r = fma(-kd, C1, x);                            // C1 has 36 significant bits: kd C1 is exact
    r = fma(-kd, C2, r);
    const int ki = (int)kd;
    double q = 8.3333333333333333333e-03;                  // 1/120
    q = fma(q, r, 4.1666666666666666667e-02);
    q = fma(q, r, 1.6666666666666666667e-01);
    q = fma(q, r, 0.5);
    q = fma(q, r, 1.0);
    q = fma(q, r, 1.0);
    return ldexp(stab[ki & 63] * q, ki >> 6);
}

template <int KID>
__device__ __forceinline__ double gpcc_kernel_eval(double xi, double xj, GpccKernelConst kc, const double *stab = nullptr)
{
    // stab != NULL (compile-time at every hot call site): the table-based exp; NULL: the polynomial one (utilities)
#ifdef GPCC_AB_POLY_EXP   /* A/B builds only (tools/ab_exp.sh): the round-1 polynomial exp everywhere */
    auto ex = [&](double a) { return gpcc_exp_nonpos(a); };
#else
    auto ex = [&](double a) { return stab ? gpcc_exp_nonpos_tab(a, stab) : gpcc_exp_nonpos(a); };
#endif
    if (KID == 0) {  // OU: exp(-|xi-xj|/rho)
        const double r = fabs(xi - xj);
        return ex(-(r * kc.c1));
    } else if (KID == 1) {  // rbf
        const double d = xi - xj;
        return ex(-((0.5 * (d * d)) * kc.c1));
    } else if (KID == 2) {  // matern32: (1 + sqrt3 r/rho) exp(-sqrt3 r/rho)
        const double r = fabs(xi - xj);
        const double t = (1.7320508075688772 * r) * kc.c1;
        return (1.0 + t) * ex(-t);
    } else {  // matern52: (1 + sqrt5 r/rho + 5 r^2/(3 rho^2)) exp(-sqrt5 r/rho)
        const double r = fabs(xi - xj);
        const double t = (2.23606797749979 * r) * kc.c1;
        return (1.0 + t + (5.0 * (r * r)) * kc.c2) * ex(-t);
    }
}

// fp32 evaluation of the same kernels for fp32 tiles (option "fp32_assemble"): the distance r = |x_i - x_j| is formed in fp64
// and rounded once, everything after it is fp32 (v_exp_f32 through __expf).  Relative error of an element ~ (3 + t) u32 with
// t the exponent's magnitude, against u32 / 2 for "fp64, rounded once" -- irrelevant for the quadratic forms, which the
// refinement pass recomputes from the exact fp64 elements, and a perturbation of the log-determinant of the same kind as (and
// smaller than) the fp32 factorisation's own; the fp64 element costs ~30 double-rate VALU operations, this one 2 + ~9 fp32.
template <int KID>
__device__ __forceinline__ float gpcc_kernel_eval_f32(float r, float c1, float c2)
{
    if (KID == 0) {
        return __expf(-(r * c1));
    } else if (KID == 1) {
        return __expf(-((0.5f * (r * r)) * c1));
    } else if (KID == 2) {
        const float t = (1.7320508f * r) * c1;
        return (1.0f + t) * __expf(-t);
    } else {
        const float t = (2.2360680f * r) * c1;
        return (1.0f + t + (5.0f * (r * r)) * c2) * __expf(-t);
    }
}

// The same kernels with their constants folded into ONE scale of the distance (round 3): t = |x_i - x_j| * kscale (rbf:
// (x_i - x_j)^2 * kscale), the difference first -- it is exact for nearby points -- then exp(-t) times the Matern polynomial
// in t.  Fewer roundings than gpcc_kernel_eval and 1-4 operations cheaper; used by the select-free assembly path, the
// refinement pass and the small-N kernels.
template <int KID>
__device__ __forceinline__ double gpcc_kernel_scale(GpccKernelConst kc)
{
    return (KID == 0) ? kc.c1 : (KID == 1) ? 0.5 * kc.c1 : (KID == 2) ? 1.7320508075688772 * kc.c1 : 2.23606797749979 * kc.c1;
}
template <int KID>
__device__ __forceinline__ double gpcc_kernel_eval_scaled(double xi, double xj, double kscale)
{
    const double d = xi - xj;
    const double t = (KID == 1) ? (d * d) * kscale : fabs(d) * kscale;
    double kv = gpcc_exp_nonpos(-t);
    if (KID == 2) kv = kv * (1.0 + t);
    else if (KID == 3) kv = kv * fma(t, fma(t, 1.0 / 3.0, 1.0), 1.0);
    return kv;
}

// ------------------------------------------------------------------------------------------
// The exponential kernels are SEPARABLE (round 4): for OU / Matern-3/2 / Matern-5/2 the element is a polynomial in t = s |u_i - u_j|
// times exp(-t), and
//     a_i a_j exp(-s |u_i - u_j|) = min(A_i B_j, A_j B_i),   A_p = a_p exp(-s (u_p - c)),   B_p = a_p exp(+s (u_p - c))
// (one of the two products is a a' e^-t, the other a a' e^+t; c any constant).  So the N^2 exponentials of an evaluation -- 19 of the
// ~28 double-precision operations of an element, what bounds the assembly (4.6 of 8 TB/s), the fp32 refinement pass and a quarter of
// a small-N evaluation -- become 2 N exponentials per evaluation (per tile: 512 instead of 16 384) and two multiplications and a
// minimum per element.  Accuracy: the argument s (u - c) is carried as a double-double (TwoSum of u - c, fma for the product), so
// A_p and B_p are good to ~2 ulp whatever the magnitude of the argument and an element to ~4 ulp -- plus |t| ulp(s) from the
// rounded scale s, which the direct form has too.  Range: |s (u - c)| <= GPCC_SEP_MAX keeps A, B finite; a product may overflow to
// +inf (the minimum then picks the other) or underflow to 0 where the true value is below 1e-520.  Outside the range -- and for rbf,
// which is not of this form -- the direct evaluation is used (decided per tile resp. per evaluation, uniformly).
// ------------------------------------------------------------------------------------------
#define GPCC_SEP_MAX 600.0
__device__ __forceinline__ double gpcc_exp_dd(double hi, double lo)   // exp(hi + lo), |lo| << 1, |hi| <~ 700
{
    const double L2E = 1.4426950408889634074, LN2HI = 6.93147180369123816490e-01, LN2LO = 1.90821492927058770002e-10;
    const double n = rint(hi * L2E);
    double r = fma(-n, LN2HI, hi);
    r = fma(-n, LN2LO, r);
    r += lo;
    double p = 1.6059043836821614599e-10;  // 1/13!
    p = fma(p, r, 2.0876756987868098979e-09);
    p = fma(p, r, 2.5052108385441718775e-08);
    p = fma(p, r, 2.7557319223985890653e-07);
    p = fma(p, r, 2.7557319223985890653e-06);
    p = fma(p, r, 2.4801587301587301587e-05);
    p = fma(p, r, 1.9841269841269841270e-04);
    p = fma(p, r, 1.3888888888888888889e-03);
    p = fma(p, r, 8.3333333333333333333e-03);
    p = fma(p, r, 4.1666666666666666667e-02);
    p = fma(p, r, 1.6666666666666666667e-01);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    const double nn = fmin(fmax(n, -1100.0), 1100.0);
    return ldexp(p, (int)nn);
}
// A = amp exp(-s (u - c)), B = amp exp(+s (u - c)); returns false (A, B undefined) if the argument is out of range
__device__ __forceinline__ bool gpcc_sep_point(double u, double c, double s, double amp, double &A, double &B)
{
#pragma clang fp contract(off)
    const double w = u - c;                         // TwoSum: u - c = w + wl exactly
    const double bb = w - u;
    const double wl = (u - (w - bb)) + (-c - bb);
    const double p = s * w;                         // s (u - c) = p + e to ~2^-104 relative
    const double e = __builtin_fma(s, w, -p) + s * wl;
    A = amp * gpcc_exp_dd(-p, -e);
    B = amp * gpcc_exp_dd(p, e);
    return fabs(p) <= GPCC_SEP_MAX;
}
// the element from the separable factors: aa' k(u_i, u_j) with t = s |u_i - u_j|   (KID 0 OU, 2 Matern-3/2, 3 Matern-5/2)
// (contract(off): the multiplications below carry no `contract` flag, so no caller can fuse its own addition -- the B term, say -- into
//  them: an element has the same bits wherever it is evaluated -- assembly, fold, refinement, small-N kernels --, whatever the compiler
//  version; the fmas the element is meant to have are written out)
template <int KID>
__device__ __forceinline__ double gpcc_sep_eval(double ui, double uj, double Ai, double Bi, double Aj, double Bj, double s)
{
#pragma clang fp contract(off)
    const double e = fmin(Ai * Bj, Aj * Bi);
    if (KID == 0) return e;
    const double t = fabs(ui - uj) * s;
    if (KID == 2) return __builtin_fma(e, t, e);
    return e * __builtin_fma(t, __builtin_fma(t, 1.0 / 3.0, 1.0), 1.0);
}

// ------------------------------------------------------------------------------------------
// gpcc_assemble_tiles: K = delayedCovariance + Sobs (+ B) for `cnt` evaluations, written once,
// lower-triangle tiles only, 16 B per lane fully coalesced (a workgroup store instruction
// covers 4 KiB contiguous).  HBM-write-bound: sizeof(T) * 128*128 * nt(nt+1)/2 bytes per evaluation.
// Replaces delayedCovariance.jl:23-31 + marginaliseb.jl:135 (+ Sobs + B) + :137 (symmetrise:
// a no-op, K is exactly symmetric, so only the lower triangle is materialised).  Elements are
// evaluated in fp64 and rounded once when T = float.  In woodbury mode the B term is left out here
// and the right-hand sides are [Q | Y - bbar].
// grid (nt*nt, cnt), block 256.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ bool diag_tile(int I, int J) { return I == J; }

// gpcc_sep_points (fold only): u, A, B, a of every point of `cnt` evaluations, the kernel's constants and the per-tile-row flags of
// GpccCtx -- the SAME values gpcc_assemble_tiles stages per tile (same functions, same arguments), computed once per evaluation.
// grid (nt, cnt), block 128.
template <int KID>
__global__ __launch_bounds__(GPCC_TILE) void gpcc_sep_points(GpccCtx c, GpccGroup g)
{
    const int I = blockIdx.x, m = blockIdx.y, slot = g.slot0 + m, r = threadIdx.x;
    const double *delays = g.delays + (long)(g.first + m) * c.L;
    const double *alpha = g.alpha + (long)(g.first + m) * c.L;
    const GpccKernelConst kc = gpcc_kernel_const<KID>(g.rho[g.first + m]);
    const double s = gpcc_kernel_scale<KID>(kc);
    const int gi = I * GPCC_TILE + r;
    const int b = c.band[gi], b0 = c.band[I * GPCC_TILE];
    const double u_ = (b >= 0) ? c.t[gi] - delays[b] : 0.0;
    const double a_ = (b >= 0) ? alpha[b] : 0.0;
    double A_ = 0.0, B_ = 0.0;
    bool ok = false;
    if (KID != 1) ok = gpcc_sep_point((b >= 0) ? u_ : c.tmid, c.tmid, s, a_, A_, B_);   // (padding: amplitude 0, at the centre -- as the assembly)
    double *sp = c.sep + (long)slot * 4 * c.Np;
    sp[gi] = u_;
    sp[c.Np + gi] = A_;
    sp[2 * (long)c.Np + gi] = B_;
    sp[3 * (long)c.Np + gi] = a_;
    const int oneband = __syncthreads_and((b >= 0 && b == b0) ? 1 : 0);
    const int inrange = __syncthreads_and(ok ? 1 : 0);
    if (r == 0) {
        c.sepflag[(long)slot * c.nt + I] = (oneband ? b0 + 1 : 0) | (inrange ? 0x100 : 0);
        if (I == 0) {
            c.seps[4 * (long)slot] = s;
            c.seps[4 * (long)slot + 1] = kc.c1;
            c.seps[4 * (long)slot + 2] = kc.c2;
        }
    }
}

// How tile (I,J), I != J, of a folded group gets its elements (fI, fJ: the flags of its two tile rows): 0 = assembled by
// gpcc_assemble_tiles and read back (points outside the separable range; rbf in fp64 or in a tile row that straddles bands; rbf in
// fp64), 1 = the separable form in fp64, rounded once for fp32 tiles, 2 = fp32 tiles evaluated in fp32 (GpccCtx::asm32; needs no B
// term), 3 = a tile row that straddles two bands or holds padding: the separable form with the B term and the padding decided per
// element (gpcc_assemble_tiles' general path), 4 = a tile inside one band pair that cannot take the separable form (rbf; points outside
// the range): the direct evaluation with its exponential, as the assembly's select-free path does it (28 operations per element: ~2 % of
// the job's matrix work, still cheaper than writing the tile and reading it back).  The SAME case distinction as in gpcc_assemble_tiles, so that folded and assembled
// tiles agree bitwise.  bt_out: the B term of a tile inside one band pair.
template <typename T>
__device__ __forceinline__ int gpcc_fold_mode(const GpccCtx &c, int fI, int fJ, double &bt_out)
{
    const int bI = fI & 0xff, bJ = fJ & 0xff;
    bt_out = 0.0;
    if (bI == 0 || bJ == 0) return (c.fold_mixed && c.kernel_id != 1 && (fI & fJ & 0x100) != 0) ? 3 : 0;
    double bt = 0.0;
    if (c.marginalise_b != 0 && !c.woodbury && bI == bJ) {
#pragma unroll
        for (int l = 0; l < GPCC_MAXL; ++l) bt = (bI - 1 == l) ? c.sigma_b[l] : bt;   // (no dynamic index into the kernel argument)
    }
    bt_out = bt;
    if (sizeof(T) == 4 && c.asm32 && bt == 0.0) return 2;
    if (c.kernel_id != 1 && (fI & fJ & 0x100) != 0) return 1;
    return c.fold_mixed ? 4 : 0;   // (the general instantiations only: the host launches those for every rbf handle)
}

// mode 4: amp * kernel(|u_i - u_j| * kscale) + bt, exactly gpcc_assemble_tiles' select-free direct path
template <int KID, typename T, typename ACC>
__device__ __forceinline__ void gpcc_fold_init_direct(ACC (&acc)[8], double ui, double ai, const double *cu, double acol, double kscale, double bt, int q)
{
    typedef GpccPrec<T> P;
    const double amp = ai * acol;
#pragma unroll
    for (int cf = 0; cf < 8; ++cf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = cf * 16 + P::crow(q, r);
            acc[cf][r] = -(T)(amp * gpcc_kernel_eval_scaled<KID>(ui, cu[j], kscale) + bt);
        }
}
template <int KID, typename T, typename ACC>
__device__ __forceinline__ void gpcc_fold_init_pu_direct(ACC (&acc)[2][4], const double *rp, const double *cp, long Np, double kscale, double bt, int q)
{
    typedef GpccPrec<T> P;
    double uj[4];
#pragma unroll
    for (int fn = 0; fn < 4; ++fn) uj[fn] = cp[fn * 16];
    const double acol = cp[3 * Np];
#pragma unroll
    for (int fm = 0; fm < 2; ++fm)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = fm * 16 + P::crow(q, r);
            const double ui = rp[i], amp = rp[3 * Np + i] * acol;
#pragma unroll
            for (int fn = 0; fn < 4; ++fn) acc[fm][fn][r] = -(T)(amp * gpcc_kernel_eval_scaled<KID>(ui, uj[fn], kscale) + bt);
        }
}

// the B term of a row's band (0 for padding): sigma_b[br] without a dynamic index into the kernel argument
__device__ __forceinline__ double gpcc_fold_bterm(const GpccCtx &c, int br)
{
    double bt = 0.0;
    if (c.marginalise_b != 0 && !c.woodbury) {
#pragma unroll
        for (int l = 0; l < GPCC_MAXL; ++l) bt = (br == l) ? c.sigma_b[l] : bt;
    }
    return bt;
}
// an element of a tile that is NOT inside one band pair (mode 3): gpcc_assemble_tiles' general path for an off-diagonal tile
template <int KID>
__device__ __forceinline__ double gpcc_fold_mixed(double ui, double uj, double Ai, double Bi, double Aj, double Bj, double s, int br, int bc,
                                                  double bterm)
{
#pragma clang fp contract(off)   /* the assembly adds the B term to the ROUNDED element (its diagonal select sits in between) */
    double val = gpcc_sep_eval<KID>(ui, uj, Ai, Bi, Aj, Bj, s);
    if (br == bc) val = val + bterm;
    if (br < 0 || bc < 0) val = 0.0;
    return val;
}
template <int KID, typename T, typename ACC>
__device__ __forceinline__ void gpcc_fold_init_mixed(ACC (&acc)[8], double ui, double Ai, double Bi, const double *cu, const double *cA,
                                                     const double *cB, double s, int br, double bterm, const int *cb, int q)
{
    typedef GpccPrec<T> P;
#pragma unroll
    for (int cf = 0; cf < 8; ++cf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = cf * 16 + P::crow(q, r);
            acc[cf][r] = -(T)gpcc_fold_mixed<KID>(ui, cu[j], Ai, Bi, cA[j], cB[j], s, br, cb[j], bterm);
        }
}
template <int KID, typename T, typename ACC>
__device__ __forceinline__ void gpcc_fold_init_pu_mixed(ACC (&acc)[2][4], const double *rp, const double *cp, long Np, double s, const double *ssb /* LDS: Sigma_b per band, 0 where B is not added */,
                                                        const int *rb, const int *cb, int q)
{
    typedef GpccPrec<T> P;
    double uj[4], Aj[4], Bj[4];
    int bj[4];
#pragma unroll
    for (int fn = 0; fn < 4; ++fn) {
        uj[fn] = cp[fn * 16];
        Aj[fn] = cp[Np + fn * 16];
        Bj[fn] = cp[2 * Np + fn * 16];
        bj[fn] = cb[fn * 16];
    }
#pragma unroll
    for (int fm = 0; fm < 2; ++fm)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = fm * 16 + P::crow(q, r);
            const double ui = rp[i], Ai = rp[Np + i], Bi = rp[2 * Np + i];
            const int br = rb[i];
            const double bterm = (br >= 0) ? ssb[br] : 0.0;   // (LDS: the selects over the kernel argument cost 16 SGPRs, and those spilled into VGPRs)
#pragma unroll
            for (int fn = 0; fn < 4; ++fn) acc[fm][fn][r] = -(T)gpcc_fold_mixed<KID>(ui, uj[fn], Ai, Bi, Aj[fn], Bj[fn], s, br, bj[fn], bterm);
        }
}

// the accumulators of tile (I,k) from the separable factors (gpcc_update_solve, fold): acc[cf][r'] = -K[row][16 cf + crow(q, r')],
// the element exactly as gpcc_assemble_tiles' select-free path forms it (same expression, same operand order, same rounding to T)
template <int KID, typename T, typename ACC>
__device__ __forceinline__ void gpcc_fold_init(ACC (&acc)[8], double ui, double Ai, double Bi, const double *cu, const double *cA,
                                               const double *cB, double s, double bt, int q)
{
    typedef GpccPrec<T> P;
#pragma unroll
    for (int cf = 0; cf < 8; ++cf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = cf * 16 + P::crow(q, r);
            acc[cf][r] = -(T)(gpcc_sep_eval<KID>(ui, cu[j], Ai, Bi, cA[j], cB[j], s) + bt);
        }
}
// ... and the fp32 evaluation of fp32 tiles (GpccCtx::asm32)
template <int KID, typename T, typename ACC>
__device__ __forceinline__ void gpcc_fold_init_f32(ACC (&acc)[8], double ui, float ai, const double *cu, float acol, float c1, float c2, int q)
{
    typedef GpccPrec<T> P;
    const float amp = ai * acol;
#pragma unroll
    for (int cf = 0; cf < 8; ++cf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = cf * 16 + P::crow(q, r);
            acc[cf][r] = -(T)(amp * gpcc_kernel_eval_f32<KID>((float)fabs(ui - cu[j]), c1, c2));
        }
}

// ... and in gpcc_panel_update's accumulator layout (the three-kernel path, fold = 2): acc[fm][fn][r'] = -K[ro + 16 fm + crow(q, r')]
// [co + 16 fn]; rp / cp point at the first row / this lane's first column of the per-point arrays (stride Np between u, A, B, a)
template <int KID, int MODE, typename T, typename ACC>
__device__ __forceinline__ void gpcc_fold_init_pu(ACC (&acc)[2][4], const double *rp, const double *cp, long Np, double s, double bt,
                                                  float c1, float c2, int q)
{
    typedef GpccPrec<T> P;
    double uj[4], Aj[4], Bj[4];
#pragma unroll
    for (int fn = 0; fn < 4; ++fn) {
        uj[fn] = cp[fn * 16];
        if (MODE == 1) {
            Aj[fn] = cp[Np + fn * 16];
            Bj[fn] = cp[2 * Np + fn * 16];
        }
    }
    const float acol = (MODE == 2) ? (float)cp[3 * Np] : 0.0f;
#pragma unroll
    for (int fm = 0; fm < 2; ++fm)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = fm * 16 + P::crow(q, r);
            const double ui = rp[i];
            if (MODE == 1) {
                const double Ai = rp[Np + i], Bi = rp[2 * Np + i];
#pragma unroll
                for (int fn = 0; fn < 4; ++fn) acc[fm][fn][r] = -(T)(gpcc_sep_eval<KID>(ui, uj[fn], Ai, Bi, Aj[fn], Bj[fn], s) + bt);
            } else {
                const float amp = (float)rp[3 * Np + i] * acol;
#pragma unroll
                for (int fn = 0; fn < 4; ++fn) acc[fm][fn][r] = -(T)(amp * gpcc_kernel_eval_f32<KID>((float)fabs(ui - uj[fn]), c1, c2));
            }
        }
}

template <int KID, bool EXT, typename T>
__global__ __launch_bounds__(256, sizeof(T) == 4 ? 3 : 4) void gpcc_assemble_tiles(GpccCtx c, GpccGroup g)
{
    typedef GpccPrec<T> P;
    const int I = blockIdx.x / c.nt, J = blockIdx.x % c.nt;
    if (J > I) return;
    const int m = blockIdx.y, slot = g.slot0 + m, tid = threadIdx.x;
    // gridDim.z = 4 (a few evaluations in front of the persistent launch): a workgroup writes one row quarter -- rows r0 + 32 z of
    // every thread -- of its tile: four times the workgroups for a launch that would otherwise be 36 .. 528 of them on 256 CUs
    const int qz = (gridDim.z > 1) ? (int)blockIdx.z : -1;
    const int first_row = (c.share_p && m > 0) ? c.share_p : 0;   // followers of a shared prefix skip the leader's rows
    if (I < first_row) return;
    // fold: this tile is never read from memory -- gpcc_update_solve evaluates it into its accumulators (flags: gpcc_sep_points)
    if (c.fold && I != J && (c.fold == 1 || J >= 1)) {   // (fold = 2, the three-kernel path: column 0 goes straight to the panel solve)
        double bt_;
        if (gpcc_fold_mode<T>(c, c.sepflag[(long)slot * c.nt + I], c.sepflag[(long)slot * c.nt + J], bt_) != 0) return;
    }
    const double *delays = g.delays + (long)(g.first + m) * c.L;
    const double *alpha = g.alpha + (long)(g.first + m) * c.L;
    const GpccKernelConst kc = gpcc_kernel_const<KID>(g.rho[g.first + m]);
    const double rho = g.rho[g.first + m];

    __shared__ __attribute__((aligned(16))) double su[2][GPCC_TILE], sa[2][GPCC_TILE], ssig[GPCC_TILE], ssb[GPCC_MAXL];
    __shared__ __attribute__((aligned(16))) double sA[2][GPCC_TILE], sB[2][GPCC_TILE];   // separable factors (gpcc_sep_point)
    __shared__ int sb[2][GPCC_TILE];
    __shared__ double syv[EXT ? GPCC_TILE : 1];   // fluxes of the tile's columns (explicit 'Y' rows only)
    if (tid < GPCC_MAXL) ssb[tid] = (tid < c.L) ? c.sigma_b[tid] : 0.0;
    bool sep_ok = true;   // this thread's point is inside the range of the separable form

    if (I == first_row && J == 0 && tid == 0) {  // per-slot state + the reference's argument checks
        int bad = 0;
        for (int l = 0; l < c.L; ++l)
            if (!(alpha[l] > 0.0)) bad = -1;  // delayedCovariance.jl:3
        if (bad == 0 && rho <= 0.0) bad = -2;  // delayedCovariance.jl:5-7
        c.info[slot] = bad;
        c.logdet[slot] = 0.0;
        for (int i = 0; i < c.nrhs * c.nrhs; ++i) c.gram[(long)slot * GPCC_MAXRHS * GPCC_MAXRHS + i] = 0.0;
        if (sizeof(T) == 4 && c.cond) c.cond[2 * (long)slot] = c.cond[2 * (long)slot + 1] = 0.0;
    }
    if (c.chain_words && I == first_row && J == 0) {   // the flag words of the launch that follows (a kernel boundary away): all zero
        unsigned *wz = c.chain_words + c.chain_qbase + (long)m * c.chain_ev_words;
        for (int i = tid; i < c.chain_ev_words; i += 256) wz[i] = 0u;
        if (m == 0)
            for (int i = tid; i < c.chain_qbase; i += 256) c.chain_words[i] = 0u;
    }
    {
        const int side = tid >> 7, r = tid & 127;
        const int gi = (side ? J : I) * GPCC_TILE + r;
        const int b = c.band[gi];
        sb[side][r] = b;
        const double u_ = (b >= 0) ? c.t[gi] - delays[b] : 0.0;  // x - delays[i], delayedCovariance.jl:27
        const double a_ = (b >= 0) ? alpha[b] : 0.0;
        su[side][r] = u_;
        sa[side][r] = a_;
        if (KID != 1) {   // (padding / explicit rows: amplitude 0, placed at the centre so that they never leave the range)
            double A_, B_;
            sep_ok = gpcc_sep_point((b >= 0) ? u_ : c.tmid, c.tmid, gpcc_kernel_scale<KID>(kc), a_, A_, B_);
            sA[side][r] = A_;
            sB[side][r] = B_;
        }
        if (side == 0) {
            ssig[r] = c.sig2[gi];
            if (I == J) {  // right-hand sides: z <- Y - bbar (last column), woodbury: the columns of Q before it
                double *zs = c.z + (long)slot * c.nrhs * c.Np;
                for (int j = 0; j < c.nrhs - 1; ++j) zs[(long)j * c.Np + gi] = (b == j) ? 1.0 : 0.0;
                zs[(long)(c.nrhs - 1) * c.Np + gi] = c.resid[gi];
            }
        } else if (EXT) {
            syv[r] = c.yv[gi];
        }
    }
    const bool sep = __syncthreads_and(sep_ok ? 1 : 0) != 0 && KID != 1;   // (also the barrier behind the staging)

    T *Tt = (T *)c.tiles + (long)slot * c.slot_stride + gpcc_tile_off(I, J);
    const bool diag = (I == J);
    const bool mb = c.marginalise_b != 0 && !c.woodbury;
    const int sp = tid & 7;  // this thread's 16-byte storage slot in every row it writes
    // Points are ordered by band (padding / explicit rows last), so a tile whose first and last row (column) carry
    // the same band id >= 0 lies inside one band pair: no diagonal, no padding, a uniform B term -- the plain
    // element (scale_i scale_j kernel + const) without any per-element select.  Most tiles are of this kind.
    const int rb0 = sb[0][0], cb0 = sb[1][0];
    if (!diag_tile(I, J) && rb0 >= 0 && cb0 >= 0 && rb0 == sb[0][GPCC_TILE - 1] && cb0 == sb[1][GPCC_TILE - 1]) {
        const double bt = (c.marginalise_b != 0 && !c.woodbury && rb0 == cb0) ? ssb[rb0] : 0.0;
        if (sizeof(T) == 4 && c.asm32 && bt == 0.0) {   // fp32 tiles, fp32 evaluation (see gpcc_kernel_eval_f32)
            const float c1 = (float)kc.c1, c2 = (float)kc.c2;
            const float acol = (float)sa[1][0];   // one band per side: one amplitude per side
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (qz >= 0 && j != qz) continue;
                const int r = (tid >> 3) + 32 * j;
                const double ur = su[0][r];
                const float amp = (float)sa[0][r] * acol;
                const int cs = (sp ^ gpcc_sw(r)) * P::EP;
#pragma unroll
                for (int ch = 0; ch < P::NCH; ++ch) {
                    const int col = ch * P::KC + cs;
                    typename P::v16 v;
#pragma unroll
                    for (int h = 0; h < P::EP; ++h)
                        v[h] = (T)(amp * gpcc_kernel_eval_f32<KID>((float)fabs(ur - su[1][col + h]), c1, c2));
                    *(typename P::v16 *)(Tt + ch * (GPCC_TILE * P::KC) + r * P::KC + sp * P::EP) = v;
                }
            }
            return;
        }
        // one band per side: one amplitude per side, and the kernel's constants fold into ONE scale of the distance,
        // t = |u_i - u_j| * kscale (the difference first: exact for nearby points) -- the same element as gpcc_kernel_eval's to an
        // ulp or two (fewer roundings, not more), 3-6 operations and one LDS read cheaper
        const double kscale = gpcc_kernel_scale<KID>(kc);
        if (KID != 1 && sep) {
            // the separable form: a a' e^-t = min(A_i B_j, A_j B_i) -- no exponential per element (512 per tile instead of 16 384):
            // 7 double-precision operations per element instead of ~28, so that the kernel is bound by its stores
            // (a thread's four rows r, r + 32, .. share their swizzle, hence their columns: chunk by chunk, the column factors
            //  of a chunk in registers for the four rows -- not all 16 columns x 3 arrays at once)
            const int r0_ = tid >> 3;
            const int cs = (sp ^ gpcc_sw(r0_)) * P::EP;
            double ur[4], Ar[4], Br[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ur[j] = su[0][r0_ + 32 * j];
                Ar[j] = sA[0][r0_ + 32 * j];
                Br[j] = sB[0][r0_ + 32 * j];
            }
#pragma unroll 2
            for (int ch = 0; ch < P::NCH; ++ch) {
                const int col = ch * P::KC + cs;
                double uc[P::EP], Ac[P::EP], Bc[P::EP];
#pragma unroll
                for (int h = 0; h < P::EP; ++h) {
                    uc[h] = su[1][col + h];
                    Ac[h] = sA[1][col + h];
                    Bc[h] = sB[1][col + h];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (qz >= 0 && j != qz) continue;
                    typename P::v16 v;
#pragma unroll
                    for (int h = 0; h < P::EP; ++h) v[h] = (T)(gpcc_sep_eval<KID>(ur[j], uc[h], Ar[j], Br[j], Ac[h], Bc[h], kscale) + bt);
                    // (non-temporal stores: measured, no difference -- 13.4-13.7 ms per 1024 evaluations either way)
                    *(typename P::v16 *)(Tt + ch * (GPCC_TILE * P::KC) + (r0_ + 32 * j) * P::KC + sp * P::EP) = v;
                }
            }
            return;
        }
        const double acol = sa[1][0];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (qz >= 0 && j != qz) continue;
                const int r = (tid >> 3) + 32 * j;
            const double ur = su[0][r], amp = sa[0][r] * acol;
            const int cs = (sp ^ gpcc_sw(r)) * P::EP;
#pragma unroll
            for (int ch = 0; ch < P::NCH; ++ch) {
                const int col = ch * P::KC + cs;
                typename P::v16 v;
#pragma unroll
                for (int h = 0; h < P::EP; ++h) {
                    v[h] = (T)(amp * gpcc_kernel_eval_scaled<KID>(ur, su[1][col + h], kscale) + bt);
                }
                *(typename P::v16 *)(Tt + ch * (GPCC_TILE * P::KC) + r * P::KC + sp * P::EP) = v;
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (qz >= 0 && j != qz) continue;
                const int r = (tid >> 3) + 32 * j;  // row data stays in registers across the chunks
        const int br = sb[0][r];
        const double ur = su[0][r], ar = sa[0][r], sg = ssig[r];
        const double Ar = (KID != 1) ? sA[0][r] : 0.0, Br = (KID != 1) ? sB[0][r] : 0.0;
        const double kscale_g = gpcc_kernel_scale<KID>(kc);
        const double bterm = (mb && br >= 0) ? ssb[br] : 0.0;
        const int cs = (sp ^ gpcc_sw(r)) * P::EP;  // logical column (inside a chunk) stored in slot sp
#pragma unroll
        for (int ch = 0; ch < P::NCH; ++ch) {
            const int col = ch * P::KC + cs;
            typename P::v16 v;
#pragma unroll
            for (int h = 0; h < P::EP; ++h) {
                const int cc = col + h;
                const int bc = sb[1][cc];
                // scale[i]*scale[j]*kernel(x - delays[i], y - delays[j]): the separable form here too (round 4: with three bands a
                // quarter of the tiles straddle a band boundary, and the 28-operation element made them cost as much as all the others)
                double val;
                if (KID != 1 && sep) val = gpcc_sep_eval<KID>(ur, su[1][cc], Ar, Br, sA[1][cc], sB[1][cc], kscale_g);
                else val = (ar * sa[1][cc]) * gpcc_kernel_eval<KID>(ur, su[1][cc], kc);
                if (diag && r == cc) val = val + sg;                         // + Sobs
                if (br == bc) val = val + bterm;                             // + B = Q Sigma_b Q'
                if (br < 0 || bc < 0) val = (diag && r == cc) ? 1.0 : 0.0;   // identity padding
                if (EXT) {  // explicit rows of an augmented system against real columns; 0 among themselves
                    if (br <= -2) {
                        const int e = -2 - br;
                        val = (bc >= 0) ? ((e < c.L) ? ((bc == e) ? 1.0 : 0.0) : syv[cc]) : 0.0;
                    } else if (bc <= -2) {
                        val = 0.0;  // never read: explicit points come last, so they are rows of the lower triangle
                    }
                }
                if (sizeof(T) == 4 && diag && r == cc && c.kdiag)   // unrounded diagonal of a real point (0: padding)
                    c.kdiag[(long)slot * c.Np + I * GPCC_TILE + r] = (br >= 0) ? val : 0.0;
                v[h] = (T)val;
            }
            *(typename P::v16 *)(Tt + ch * (GPCC_TILE * P::KC) + r * P::KC + sp * P::EP) = v;
        }
    }
}

// ------------------------------------------------------------------------------------------
// The MFMA kernels.  Measured on MI355X (tools/microbench.hip): v_mfma_f64_16x16x4_f64
// issues every 64 cycles per SIMD (77.6 TFLOP/s chip-wide) only when at least TWO waves of that
// SIMD have MFMAs to issue -- one wave alone gets one per ~139 cycles whatever its number of
// independent accumulators -- and VALU DFMA shares the same 78 TFLOP/s.  Hence: 8-wave workgroups,
// a 32x64 (panel update) or 16x128 (panel solve) sub-tile per wave = 8 accumulators,
// <= 128 VGPRs in all, two workgroups per CU = FOUR waves per SIMD, so barriers and LDS-DMA waits
// of one wave are covered by three others.  The fp32 instantiation (v_mfma_f32_16x16x4_f32, 32 cycles)
// has the same byte geometry: a 16 KiB chunk holds 32 k instead of 16 and costs 8 MFMA steps instead of 4.
// Operands go HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction, the
// 16 KiB chunk lands as a linear copy) into a 2-deep ring: 2 x (A 16 KiB + B 16 KiB) = 64 KiB.
// ------------------------------------------------------------------------------------------
// A follower of a shared prefix inherits the leader's failure only if it happened inside the prefix
// (argument error, or a non-positive pivot among the first share_p*128 columns).
__device__ __forceinline__ int gpcc_leader_failure(const GpccCtx &c, const GpccGroup &g)
{
    if (!c.share_p) return 0;
    const int li = c.info[g.slot0];
    return (li < 0 || (li > 0 && li <= c.share_p * GPCC_TILE)) ? li : 0;
}

// One LDS-DMA piece (1 KiB: 16 bytes per lane) in the SCALAR-ADDRESS form: global address = uniform 64-bit base (an SGPR pair) +
// one per-lane 32-bit offset + immediate, LDS address in M0 from a scalar.  The builtin, given a per-lane pointer, emits a
// 64-bit VALU add per piece and moves the LDS address VGPR -> v_readfirstlane -> M0; written out, a piece costs one s_mov and the
// instruction itself.  (The compiler does not count these loads in its vmcnt bookkeeping: every consumer here already waits with an
// explicit s_waitcnt vmcnt(N) in front of its barrier.)
// two consecutive pieces (2 KiB of one operand, the same M0): one M0 write for both
__device__ __forceinline__ void gpcc_dma_piece2(const void *gbase, unsigned voff, unsigned lds_addr)
{
#ifdef GPCC_AB_DMA_BUILTIN
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const char *)gbase + voff),
                                     (__attribute__((address_space(3))) void *)(size_t)lds_addr, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const char *)gbase + voff + 1024),
                                     (__attribute__((address_space(3))) void *)(size_t)(lds_addr + 1024), 16, 0, 0);
#else
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024"
                 :
                 : "s"(lds_addr), "v"(voff), "s"(gbase)
                 : "memory", "m0");
#endif
}
template <int IMM>
__device__ __forceinline__ void gpcc_dma_piece(const void *gbase, unsigned voff, unsigned lds_addr)
{
#ifdef GPCC_AB_DMA_BUILTIN   /* A/B builds only (tools/ab_dma.sh): the compiler's form of the same transfer (rounds 1-3) */
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const char *)gbase + voff + IMM),
                                     (__attribute__((address_space(3))) void *)(size_t)(lds_addr + IMM), 16, 0, 0);
#else
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%3"
                 :
                 : "s"(lds_addr), "v"(voff), "s"(gbase), "n"(IMM)
                 : "memory", "m0");
#endif
}
__device__ __forceinline__ unsigned gpcc_lds_addr(const void *p)
{
    return (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) const void *)p);
}
// a pointer that IS uniform over the wave, into scalar registers (a no-op where the compiler already keeps it there; where it was
// derived through vector instructions -- a tile index from sqrtf, say -- the "s" operand of the asm needs it said)
__device__ __forceinline__ const void *gpcc_uniform_ptr(const void *p)
{
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return (const void *)(((unsigned long long)hi << 32) | lo);
}

// the same with the stage's LDS byte address as a number (hot loops compute it once: a generic -> LDS pointer cast costs a
// null check and an aperture compare per use)
template <typename T>
__device__ __forceinline__ void gpcc_dma_chunk_at(const T *gA, const T *gB, unsigned stage_addr, int wave, int lane)
{
    constexpr int PIECE = 1024 / sizeof(T);
    const int uw = __builtin_amdgcn_readfirstlane(wave);
    const unsigned voff = (unsigned)lane * 16u;
    gpcc_dma_piece2(gpcc_uniform_ptr(gA + uw * 2 * PIECE), voff, stage_addr + uw * 2048);
    gpcc_dma_piece2(gpcc_uniform_ptr(gB + uw * 2 * PIECE), voff, stage_addr + GPCC_CHUNK_BYTES + uw * 2048);
}

template <typename T>
__device__ __forceinline__ void gpcc_dma_chunk(const T *gA, const T *gB, T *stage, int wave, int lane)
{
    constexpr int PIECE = 1024 / sizeof(T), CH = GPCC_CHUNK_BYTES / sizeof(T);
    // Round 4: a timing-only build without these instructions showed them to cost 9 % of the update kernel
    // (tools/timing_variants.sh) -- not the transfers, their ISSUE: per piece a 64-bit VALU address add and a VGPR -> readfirstlane ->
    // M0 detour, because `tid >> 6` does not tell the compiler that a wave's pieces are uniform.  Now: uniform wave index, scalar
    // bases (wave's two consecutive pieces = one base + immediate 1024), one per-lane offset.  (Issuing the row-I pieces behind the
    // first 16 MFMAs instead of all four at the top of the chunk: -0.5 % fp64, -3.5 % fp32, dropped.)
    const int uw = __builtin_amdgcn_readfirstlane(wave);
    const unsigned voff = (unsigned)lane * 16u;
    const void *pa = gpcc_uniform_ptr(gA + uw * 2 * PIECE), *pb = gpcc_uniform_ptr(gB + uw * 2 * PIECE);
    const unsigned la = gpcc_lds_addr(stage + uw * 2 * PIECE), lb = gpcc_lds_addr(stage + CH + uw * 2 * PIECE);
    gpcc_dma_piece2(pa, voff, la);          // (the instruction's immediate offset applies to the global AND the LDS address:
    gpcc_dma_piece2(pb, voff, lb);          //  the second piece of a wave is the same M0 with offset:1024)
}

// ------------------------------------------------------------------------------------------
// gpcc_panel_update (step k, tile I >= k):  T(I,k) -= sum_{j<ktiles} L(I,j) L(k,j)^T
// = LAPACK dpotrf's dsyrk/dgemm (reached from cholesky(K), marginaliseb.jl:139), left-looking:
// the K-loop runs over the chunks of the two contiguous tile rows I and k, the accumulator
// starts at -T(I,k) and is stored back negated (no read-modify-write).  The diagonal tile (I == k, dsyrk) is
// computed in full like the others (4.5 % of the update flops are redundant; letting the two waves strictly
// above the diagonal skip their MFMAs bought nothing in an in-process A/B, nor did staggering the two
// workgroups of a CU by half a chunk period).
// Blocks of one evaluation share blockIdx % 8, i.e. (as dispatched) an XCD and its L2: they all
// stream tile row k.
// ------------------------------------------------------------------------------------------
// RIGHT = true is the right-looking form used for small groups (<= GPCC_RIGHT_LOOKING_MAX evaluations, e.g. the
// single objective(alpha, rho) call of Nelder-Mead): after step k every trailing tile (I,J), I >= J > k, gets
// T(I,J) -= L(I,k) L(J,k)^T -- n(n+1)/2 short jobs (one tile of K) per evaluation instead of n long ones, so a
// handful of matrices still fills the chip; it re-reads and re-writes the trailing matrix every step, which the
// 256 MiB Infinity Cache absorbs for a few matrices but HBM would not for 256.
// kcol = first tile column of the K loop: 0 for the left-looking form, k for a right-looking step.  RIGHT with kcol = 0 and
// ktiles = k + 1 is the CATCH-UP of a group that switches from left- to right-looking at step k + 1: every trailing tile
// (I,J), I >= J > k, receives the whole sum over the finished columns 0..k at once.
template <typename T, bool RIGHT, bool MIXED = false>
__global__ __launch_bounds__(512, 4) void gpcc_panel_update(GpccCtx c, GpccGroup g, int k, int ktiles, int kcol)
{
    typedef GpccPrec<T> P;
    constexpr int CH = GPCC_CHUNK_BYTES / sizeof(T);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *smem = (T *)smem_raw;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform, and said so)
    const int wr = wave >> 1, wc = wave & 1, lr = lane & 15, q = lane >> 4;
    const int sw = gpcc_sw(lr);

    const int nrem = c.nt - k - 1;
    const bool shared = !RIGHT && c.share_p > k;   // step inside the shared prefix: B operand = the leader's row k
    const int per = RIGHT ? nrem * (nrem + 1) / 2 : (shared ? c.nt - c.share_p : c.nt - k);
    const int nmain = 8 * ((g.cnt + 7) / 8) * per;
    const int x = blockIdx.x & 7, qq = blockIdx.x >> 3;
    int m = g.spread ? (int)blockIdx.x % g.cnt : (qq / per) * 8 + x;
    const int jt = g.spread ? (int)blockIdx.x / g.cnt : qq % per;   // job of evaluation m
    int I, J;   // output tile (I,J); left-looking: J = k
    if (shared && (int)blockIdx.x >= nmain) {   // the leader's own rows k .. share_p-1
        m = 0;
        I = k + ((int)blockIdx.x - nmain);
        J = k;
    } else if (m >= g.cnt) {
        return;
    } else if (shared) {
        I = c.share_p + jt;
        J = k;
    } else if (RIGHT) {
        const int j = jt, first = k + 1;
        int a = (int)((sqrtf(8.0f * j + 1.0f) - 1.0f) * 0.5f);
        while (a * (a + 1) / 2 > j) --a;
        while ((a + 1) * (a + 2) / 2 <= j) ++a;
        I = first + a;
        J = first + (j - a * (a + 1) / 2);
    } else {
        I = k + jt;
        J = k;
    }
    const int slot = g.slot0 + m;
    if (c.info[slot] != 0 || gpcc_leader_failure(c, g) != 0) return;

    T *tiles = (T *)c.tiles + (long)slot * c.slot_stride;
    const T *btiles = shared ? (const T *)c.tiles + (long)g.slot0 * c.slot_stride : tiles;
    const T *gA = (const T *)gpcc_uniform_ptr(tiles + gpcc_tile_off(I, kcol));
    const T *gB = (const T *)gpcc_uniform_ptr(btiles + gpcc_tile_off(J, kcol));
    T *Tt = tiles + gpcc_tile_off(I, J);
    const int nch = P::NCH * ktiles;  // ktiles = k (left-looking), 1 (right-looking), or nt_fact (Schur complement)

    const unsigned smem_addr = gpcc_lds_addr(smem);
    gpcc_dma_chunk_at<T>(gA, gB, smem_addr, wave, lane);
    __shared__ double s_sb[MIXED ? GPCC_MAXL : 1];   // Sigma_b of the bands (what gpcc_fold_bterm selects from the kernel argument), for the mixed fold
    if (MIXED) {
        if (tid < GPCC_MAXL) s_sb[tid] = (c.marginalise_b != 0 && !c.woodbury) ? c.sigma_b[tid] : 0.0;
        __syncthreads();
    }

    typename P::acc_t acc[2][4];
    // fold = 2 (round 4): the FIRST job that touches an off-diagonal tile (I,J), J >= 1 -- the left-looking update of column J, or the
    // right-looking step / catch-up whose K loop starts at column 0 -- evaluates its elements instead of reading an assembled tile
    int fmode = 0;
    double bt = 0.0;
    if (c.fold == 2 && I != J && (!RIGHT || kcol == 0))
        fmode = gpcc_fold_mode<T>(c, __builtin_amdgcn_readfirstlane(c.sepflag[(long)slot * c.nt + I]),
                                  __builtin_amdgcn_readfirstlane(c.sepflag[(long)slot * c.nt + J]), bt);
    if (fmode != 0) {
        const double *sp = c.sep + (long)slot * 4 * c.Np;
        const double *rp = sp + I * GPCC_TILE + wr * 32, *cp = sp + J * GPCC_TILE + wc * 64 + lr;
        const double s = c.seps[4 * (long)slot];
        const float c1 = (float)c.seps[4 * (long)slot + 1], c2 = (float)c.seps[4 * (long)slot + 2];
        const long Np = c.Np;
        if (sizeof(T) == 4 && fmode == 2) {
            if (c.kernel_id == 0) gpcc_fold_init_pu<0, 2, T>(acc, rp, cp, Np, s, bt, c1, c2, q);
            else if (c.kernel_id == 1) gpcc_fold_init_pu<1, 2, T>(acc, rp, cp, Np, s, bt, c1, c2, q);
            else if (c.kernel_id == 2) gpcc_fold_init_pu<2, 2, T>(acc, rp, cp, Np, s, bt, c1, c2, q);
            else gpcc_fold_init_pu<3, 2, T>(acc, rp, cp, Np, s, bt, c1, c2, q);
        } else if (MIXED && fmode == 4) {
            if (c.kernel_id == 0) gpcc_fold_init_pu_direct<0, T>(acc, rp, cp, Np, s, bt, q);
            else if (c.kernel_id == 1) gpcc_fold_init_pu_direct<1, T>(acc, rp, cp, Np, s, bt, q);
            else if (c.kernel_id == 2) gpcc_fold_init_pu_direct<2, T>(acc, rp, cp, Np, s, bt, q);
            else gpcc_fold_init_pu_direct<3, T>(acc, rp, cp, Np, s, bt, q);
        } else if (MIXED && fmode == 3) {
            const int *rb = c.band + I * GPCC_TILE + wr * 32, *cb = c.band + J * GPCC_TILE + wc * 64 + lr;
            if (c.kernel_id == 0) gpcc_fold_init_pu_mixed<0, T>(acc, rp, cp, Np, s, s_sb, rb, cb, q);
            else if (c.kernel_id == 2) gpcc_fold_init_pu_mixed<2, T>(acc, rp, cp, Np, s, s_sb, rb, cb, q);
            else gpcc_fold_init_pu_mixed<3, T>(acc, rp, cp, Np, s, s_sb, rb, cb, q);
        } else {
            if (c.kernel_id == 0) gpcc_fold_init_pu<0, 1, T>(acc, rp, cp, Np, s, bt, c1, c2, q);
            else if (c.kernel_id == 2) gpcc_fold_init_pu<2, 1, T>(acc, rp, cp, Np, s, bt, c1, c2, q);
            else gpcc_fold_init_pu<3, 1, T>(acc, rp, cp, Np, s, bt, c1, c2, q);
        }
    } else {
#pragma unroll
        for (int fm = 0; fm < 2; ++fm)
#pragma unroll
            for (int fn = 0; fn < 4; ++fn)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc[fm][fn][r] = -Tt[gpcc_elem_off<T>(wr * 32 + fm * 16 + P::crow(q, r), wc * 64 + fn * 16 + lr)];
    }

    // per-lane LDS addresses: one base per 16-byte slot, fragments/stages are immediates from it.
    // lane (lr, q) holds k = KSTEPS*q .. KSTEPS*q + KSTEPS-1 of its row: the MFMA sums over q, the steps over s
    const T *pa0 = smem + (wr * 32 + lr) * P::KC + (((2 * q) ^ sw) * P::EP);
    const T *pa1 = smem + (wr * 32 + lr) * P::KC + (((2 * q + 1) ^ sw) * P::EP);
    const T *pb0 = smem + CH + (wc * 64 + lr) * P::KC + (((2 * q) ^ sw) * P::EP);
    const T *pb1 = smem + CH + (wc * 64 + lr) * P::KC + (((2 * q + 1) ^ sw) * P::EP);

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int ch2 = 0; ch2 < nch; ch2 += 2) {
#pragma unroll
        for (int st = 0; st < 2; ++st) {  // stage st holds chunk ch2+st (nch is even)
            const int ch = ch2 + st;
            if (ch + 1 < nch)
                gpcc_dma_chunk_at<T>(gA + (long)(ch + 1) * CH, gB + (long)(ch + 1) * CH, smem_addr + (st ^ 1) * 2 * GPCC_CHUNK_BYTES, wave, lane);
            const int so = st * 2 * CH;
            typename P::v16 a[2][2];
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                a[f][0] = *(const typename P::v16 *)(pa0 + so + f * 16 * P::KC);
                a[f][1] = *(const typename P::v16 *)(pa1 + so + f * 16 * P::KC);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {  // column fragments two at a time (register budget: 128)
                typename P::v16 b[2][2];
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    b[f][0] = *(const typename P::v16 *)(pb0 + so + (2 * h + f) * 16 * P::KC);
                    b[f][1] = *(const typename P::v16 *)(pb1 + so + (2 * h + f) * 16 * P::KC);
                }
#pragma unroll
                for (int s = 0; s < P::KSTEPS; ++s)
#pragma unroll
                    for (int fm = 0; fm < 2; ++fm)
#pragma unroll
                        for (int f = 0; f < 2; ++f)
                            acc[fm][2 * h + f] = P::mfma(a[fm][s / P::EP][s % P::EP], b[f][s / P::EP][s % P::EP], acc[fm][2 * h + f]);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
#pragma unroll
    for (int fm = 0; fm < 2; ++fm)
#pragma unroll
        for (int fn = 0; fn < 4; ++fn)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Tt[gpcc_elem_off<T>(wr * 32 + fm * 16 + P::crow(q, r), wc * 64 + fn * 16 + lr)] = -acc[fm][fn][r];
}

// ------------------------------------------------------------------------------------------
// gpcc_update_solve (step k, tile I > k; left-looking groups): the panel update AND the panel solve of tile (I,k) in one
// job --  T'(I,k) = T(I,k) - sum_{j<k} L(I,j) L(k,j)^T ;  L(I,k) = T'(I,k) inv(L_kk)^T ;  z_I -= L(I,k) w_k  -- so the
// updated tile never makes the round trip "written by the update, read and rewritten by the solve" (256 KiB per tile at
// ~4.3 TB/s: gpcc_panel_trsm is HBM-bound).  inv(L_kk) must exist when the job runs: the diagonal tile of column k is
// updated and factored first (gpcc_syrk_diag), then this kernel takes the rest of the column.
// The trick that keeps it inside the 128-register budget: the main loop accumulates the TRANSPOSED tile -- wave w owns
// T'^T[all 128 columns c][rows r = 16 w .. 16 w + 15] as eight 16x16 blocks (MFMA A operand = tile row k, B operand = tile
// row I) -- because a 16x16 accumulator block, as it sits in registers, IS a valid MFMA B operand whose k index is its ROW
// index (C/D layout: row = crow(q, reg), col = lane & 15).  So  L^T = X T'^T  (X = inv(L_kk), lower triangular) is
// out[i] = sum_{cf <= i} X[i][cf] acc[cf]  with the X blocks as A operands read from LDS and the accumulators as B operands
// straight from registers; going down from i = 7, block acc[i] is dead once out[i] is formed, so out[i] replaces it: one
// spare accumulator.  The lower blocks of X (72 KiB in fp64) are DMA'd into LDS after the K-loop (80 KiB of LDS per
// workgroup: still two workgroups per CU).  The result leaves through LDS in the tile's own byte layout, so the global
// stores are linear 16-byte copies.
// grid 8 * ceil(cnt / 8) * (nt - k - 1), block 512.
// ------------------------------------------------------------------------------------------
#define GPCC_UPSOLVE_LDS_BYTES (80 * 1024)
// one job: tile (I,k) of evaluation `slot` (the whole workgroup; smem = GPCC_UPSOLVE_LDS_BYTES of LDS)
template <typename T, bool MIXED = false>
__device__ __forceinline__ void gpcc_update_solve_job(const GpccCtx &c, const int k, const int I, const int slot, T *smem)
{
    typedef GpccPrec<T> P;
    constexpr int CH = GPCC_CHUNK_BYTES / sizeof(T);
    constexpr int PIECE = 1024 / sizeof(T), EPB = 16 / sizeof(T);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform, and said so)
    const int lr = lane & 15, q = lane >> 4;
    const int sw = gpcc_sw(lr);
    T *tiles = (T *)c.tiles + (long)slot * c.slot_stride;
    // (uniform over the workgroup, and SAID so: the per-chunk addresses of the LDS-DMA pieces are then scalar arithmetic)
    const T *gI = (const T *)gpcc_uniform_ptr(tiles + gpcc_tile_off(I, 0));   // tile row I: MFMA B operand (this wave's 16 rows)
    const T *gK = (const T *)gpcc_uniform_ptr(tiles + gpcc_tile_off(k, 0));   // tile row k: MFMA A operand (all 128 rows)
    T *Tt = tiles + gpcc_tile_off(I, k);
    const int nch = P::NCH * k;
    const unsigned smem_addr = gpcc_lds_addr(smem);
    // (Measured and dropped in round 4: the B operand -- this wave's own 16 rows of tile row I -- loaded straight into registers
    // instead of through LDS: +0.5 % with hand-written global loads, which the compiler cannot track -- it copies or spills the
    // destination registers before the data has landed (fp32 and gpcc_step returned garbage now and then) --, and with loads it can
    // track it inserts its own s_waitcnt that counts the untracked LDS-DMA pieces as older loads and stalls every chunk.)
    // (Measured and dropped in round 4: ONE barrier per TWO chunks -- the B operand is a wave's own rows and needs no barrier, the A
    // operand in a ring of four stages filled a pair ahead, 64 MFMAs per wave between barriers: correct, and not a percent faster
    // (fp32 -1 %).  What the timing-only "no barrier" build gains is waves running free of each other, not the barriers' own cost.)
    // The per-point vectors of tile column k that the fold evaluates from (u, A, B, a: 1 KiB each; the band ids) and w_k of the forward
    // substitution go to the 8 KiB of LDS above the inverse's 72 KiB by LDS-DMA, one piece per wave, BEFORE the first chunk's pieces:
    // read from global memory, a lane's 96 loads of 8 bytes went out in register-limited batches, each a full memory latency --
    // the timing-only build without the initialisation was 7.8 % faster (N = 2048: 16 %), profiles/r04/timing_only_job_fixed_cost.log
    constexpr int VEC_OFF = 72 * 1024;
    const bool staged = c.fold != 0;   // (uniform)
    if (staged) {
        const double *spk = c.sep + (long)slot * 4 * c.Np + k * GPCC_TILE;
        if (wave < 4) gpcc_dma_piece<0>(gpcc_uniform_ptr(spk + (long)wave * c.Np), (unsigned)lane * 16u, smem_addr + VEC_OFF + wave * 1024);
        else if (wave == 4) gpcc_dma_piece<0>(gpcc_uniform_ptr(c.band + k * GPCC_TILE), (unsigned)lane * 16u, smem_addr + VEC_OFF + 4096);
        else if (wave == 5) gpcc_dma_piece<0>(gpcc_uniform_ptr(c.w + (long)slot * c.nrhs * c.Np + k * GPCC_TILE), (unsigned)lane * 16u, smem_addr + VEC_OFF + 5120);
    }
    const double *vec = (const double *)((const char *)smem + VEC_OFF);   // u | A | B | a | band (int, 512 B; 512 B unused) | w_k of rhs 0
    if (nch > 0) gpcc_dma_chunk_at<T>(gI, gK, smem_addr, wave, lane);
    const T *pb0 = smem + (wave * 16 + lr) * P::KC + (((2 * q) ^ sw) * P::EP);        // stage: [row I chunk | row k chunk]
    const T *pb1 = smem + (wave * 16 + lr) * P::KC + (((2 * q + 1) ^ sw) * P::EP);

    typename P::acc_t acc[8];   // acc[cf][r'] = -T'^T[c = 16 cf + crow(q, r')][r = 16 wave + lr]
    // fold (round 4): a tile inside one band pair whose points are in the separable range was never assembled -- its elements are
    // evaluated here, from 2 x 128 points' factors (3 KiB, cached) instead of a 128 KiB tile written and read back; ~7 double-precision
    // operations per element = 0.5 % of the job's matrix work at the mean k, under the first chunk's LDS-DMA
    int fmode = 0;
    double bt = 0.0;
    if (c.fold)
        fmode = gpcc_fold_mode<T>(c, __builtin_amdgcn_readfirstlane(c.sepflag[(long)slot * c.nt + I]),
                                  __builtin_amdgcn_readfirstlane(c.sepflag[(long)slot * c.nt + k]), bt);
    if (fmode != 0) {
        const double *sp = c.sep + (long)slot * 4 * c.Np;
        const int ri = I * GPCC_TILE + wave * 16 + lr;
        // (this lane's row: four coalesced loads, issued before the wait)
        const double ui = sp[ri], Ai = sp[c.Np + ri], Bi = sp[2 * (long)c.Np + ri], ai = sp[3 * (long)c.Np + ri];
        const int br = MIXED ? c.band[ri] : 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the vectors (and, as it happens, the first chunk) have landed
        __syncthreads();
        const double *cu = vec, *cA = vec + GPCC_TILE, *cB = vec + 2 * GPCC_TILE;   // (LDS)
        const double acol = vec[3 * GPCC_TILE];
        const int *cb = (const int *)(vec + 4 * GPCC_TILE);
        if (sizeof(T) == 4 && fmode == 2) {
            const float c1 = (float)c.seps[4 * (long)slot + 1], c2 = (float)c.seps[4 * (long)slot + 2];
            if (c.kernel_id == 0) gpcc_fold_init_f32<0, T>(acc, ui, (float)ai, cu, (float)acol, c1, c2, q);
            else if (c.kernel_id == 1) gpcc_fold_init_f32<1, T>(acc, ui, (float)ai, cu, (float)acol, c1, c2, q);
            else if (c.kernel_id == 2) gpcc_fold_init_f32<2, T>(acc, ui, (float)ai, cu, (float)acol, c1, c2, q);
            else gpcc_fold_init_f32<3, T>(acc, ui, (float)ai, cu, (float)acol, c1, c2, q);
        } else if (MIXED && fmode == 4) {
            const double s = c.seps[4 * (long)slot];   // (gpcc_kernel_scale: also the scale of the direct form)
            if (c.kernel_id == 0) gpcc_fold_init_direct<0, T>(acc, ui, ai, cu, acol, s, bt, q);
            else if (c.kernel_id == 1) gpcc_fold_init_direct<1, T>(acc, ui, ai, cu, acol, s, bt, q);
            else if (c.kernel_id == 2) gpcc_fold_init_direct<2, T>(acc, ui, ai, cu, acol, s, bt, q);
            else gpcc_fold_init_direct<3, T>(acc, ui, ai, cu, acol, s, bt, q);
        } else if (MIXED && fmode == 3) {
            const double s = c.seps[4 * (long)slot];
            const double bterm = gpcc_fold_bterm(c, br);
            if (c.kernel_id == 0) gpcc_fold_init_mixed<0, T>(acc, ui, Ai, Bi, cu, cA, cB, s, br, bterm, cb, q);
            else if (c.kernel_id == 2) gpcc_fold_init_mixed<2, T>(acc, ui, Ai, Bi, cu, cA, cB, s, br, bterm, cb, q);
            else gpcc_fold_init_mixed<3, T>(acc, ui, Ai, Bi, cu, cA, cB, s, br, bterm, cb, q);
        } else {
            const double s = c.seps[4 * (long)slot];
            if (c.kernel_id == 0) gpcc_fold_init<0, T>(acc, ui, Ai, Bi, cu, cA, cB, s, bt, q);
            else if (c.kernel_id == 2) gpcc_fold_init<2, T>(acc, ui, Ai, Bi, cu, cA, cB, s, bt, q);
            else gpcc_fold_init<3, T>(acc, ui, Ai, Bi, cu, cA, cB, s, bt, q);
        }
    } else {
#pragma unroll
        for (int cf = 0; cf < 8; ++cf)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[cf][r] = -Tt[gpcc_elem_off<T>(wave * 16 + lr, cf * 16 + P::crow(q, r))];
    }

    const T *pa0 = smem + CH + lr * P::KC + (((2 * q) ^ sw) * P::EP);
    const T *pa1 = smem + CH + lr * P::KC + (((2 * q + 1) ^ sw) * P::EP);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int ch2 = 0; ch2 < nch; ch2 += 2) {
#pragma unroll
        for (int st = 0; st < 2; ++st) {  // stage st holds chunk ch2+st (nch is even)
            const int ch = ch2 + st;
            const int so = st * 2 * CH;
            if (ch + 1 < nch)
                gpcc_dma_chunk_at<T>(gI + (long)(ch + 1) * CH, gK + (long)(ch + 1) * CH, smem_addr + (st ^ 1) * 2 * GPCC_CHUNK_BYTES, wave, lane);
            typename P::v16 b[2];
            b[0] = *(const typename P::v16 *)(pb0 + so);
            b[1] = *(const typename P::v16 *)(pb1 + so);
#pragma unroll
            for (int h = 0; h < 2; ++h) {  // row-k fragments four at a time (register budget: 128)
                typename P::v16 a[4][2];
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    a[f][0] = *(const typename P::v16 *)(pa0 + so + (4 * h + f) * 16 * P::KC);
                    a[f][1] = *(const typename P::v16 *)(pa1 + so + (4 * h + f) * 16 * P::KC);
                }
#pragma unroll
                for (int s2 = 0; s2 < P::KSTEPS; ++s2)
#pragma unroll
                    for (int f = 0; f < 4; ++f)
                        acc[4 * h + f] = P::mfma(a[f][s2 / P::EP][s2 % P::EP], b[s2 / P::EP][s2 % P::EP], acc[4 * h + f]);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    {
        // ---- lower blocks of X = inv(L_kk) -> LDS, packed chunk by chunk: chunk ch2 of the tile holds columns KC ch2 .. of all
        // 128 rows; rows above the chunk's first column block are zero and skipped.  Block (i, cf) is then read at
        // xbase(cf) + rows 16 i ..: element (R, col) at xoff[chunk] + (R - R0[chunk]) KC + swizzled slot.
        constexpr int FPC = P::KC / 16;                       // 16-column blocks per chunk (fp64: 1, fp32: 2)
        const T *gX = (const T *)c.linv + gpcc_linv_off(c, slot, k);
        int xoff[P::NCH];                                     // (compile-time after unrolling)
        {
            int o = 0;
#pragma unroll
            for (int ch = 0; ch < P::NCH; ++ch) {
                xoff[ch] = o;
                o += (GPCC_TILE - 16 * FPC * ch) * P::KC;     // rows 16 FPC ch .. 127
            }
        }
#pragma unroll
        for (int ch = 0; ch < P::NCH; ++ch) {
            const int r0x = 16 * FPC * ch, npiece = (GPCC_TILE - r0x) * P::KC / PIECE;   // 1 KiB pieces of this chunk's lower rows
            for (int pc = __builtin_amdgcn_readfirstlane(wave); pc < npiece; pc += 8)
                gpcc_dma_piece<0>(gpcc_uniform_ptr(gX + (long)ch * CH + r0x * P::KC + pc * PIECE), (unsigned)lane * 16u, gpcc_lds_addr(smem + xoff[ch] + pc * PIECE));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        auto xload = [&](int i, int cf, T (&xa)[4]) {   // A operand of block (i, cf): X[16 i + lr][16 cf + crow(q, s2)], s2 = 0..3
            const int ch = cf / FPC, r0x = 16 * FPC * ch;
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                const int col = (cf % FPC) * 16 + P::crow(q, s2);      // column inside the chunk
                xa[s2] = smem[xoff[ch] + (16 * i + lr - r0x) * P::KC + (((col / P::EP) ^ sw) * P::EP) + (col % P::EP)];
            }
        };
        T xcur[4], xnxt[4];
        xload(7, 0, xcur);
#pragma unroll
        for (int i = 7; i >= 0; --i) {
            typename P::acc_t t;
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r] = 0;
#pragma unroll
            for (int cf = 0; cf <= i; ++cf) {
                // the next block's four values are requested before this block's MFMAs (the accumulator block's row index is
                // its k index: k-step s2 pairs X[..][16 cf + crow(q, s2)] with register s2 of acc[cf])
                if (cf < i) xload(i, cf + 1, xnxt);
                else if (i > 0) xload(i - 1, 0, xnxt);
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) t = P::mfma(xcur[s2], acc[cf][s2], t);
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) xcur[s2] = xnxt[s2];
                __builtin_amdgcn_sched_barrier(0);   // keep the 144 LDS reads of this epilogue from being hoisted into registers
            }
            acc[i] = t;   // = -L^T block i (acc was -T'^T); blocks > i are final, blocks < i still hold -T'^T
        }
        __syncthreads();   // every wave is done with X before the output staging overwrites it
    }
    // ---- out through LDS in the tile's own byte layout (rows 64 h .. 64 h + 63 of every chunk = 64 KiB in fp64), then linear
    // 16-byte copies to global; L = -acc (resp. T' = -acc)
    constexpr int HALVES = (int)(sizeof(T) * GPCC_TILE_ELEMS / 65536);   // fp64: 2, fp32: 1
    constexpr int HROWS = GPCC_TILE / HALVES;
#pragma unroll
    for (int hv = 0; hv < HALVES; ++hv) {
        if (wave * 16 / HROWS == hv) {
            const int rl = wave * 16 + lr - hv * HROWS;   // row inside the half
#pragma unroll
            for (int cf = 0; cf < 8; ++cf)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int col = cf * 16 + P::crow(q, r), ch = col / P::KC, cc = col % P::KC;
                    smem[ch * (HROWS * P::KC) + rl * P::KC + (((cc / P::EP) ^ sw) * P::EP) + (cc % P::EP)] = -acc[cf][r];
                }
        }
        __syncthreads();
#pragma unroll
        for (int ch = 0; ch < P::NCH; ++ch) {
            constexpr int PER = HROWS * P::KC / P::EP;   // 16-byte pieces of this half's rows in one chunk
            for (int pc = tid; pc < PER; pc += 512)
                *(typename P::v16 *)(Tt + (long)ch * CH + hv * HROWS * P::KC + pc * P::EP) =
                    *(const typename P::v16 *)(smem + ch * (HROWS * P::KC) + pc * P::EP);
        }
        if (hv + 1 < HALVES) __syncthreads();
    }
    {
        // forward substitution of logpdf's whitening: z_I[r] -= sum_c L(I,k)[r][c] w_k[c]; a wave owns its 16 rows completely
        for (int j = 0; j < c.nrhs; ++j) {
            double *zp = c.z + ((long)slot * c.nrhs + j) * c.Np + I * GPCC_TILE;
            const double *wp = c.w + ((long)slot * c.nrhs + j) * c.Np + k * GPCC_TILE;
            double pr = 0.0;
            if (staged && j == 0) {   // w_k of the first right-hand side sits in LDS since the job began (same values, same order)
                const double *wl = vec + 5 * GPCC_TILE;
#pragma unroll
                for (int cf = 0; cf < 8; ++cf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) pr = fma(-(double)acc[cf][r], wl[cf * 16 + P::crow(q, r)], pr);
            } else {
#pragma unroll
                for (int cf = 0; cf < 8; ++cf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) pr = fma(-(double)acc[cf][r], wp[cf * 16 + P::crow(q, r)], pr);
            }
            pr += __shfl_xor(pr, 16);
            pr += __shfl_xor(pr, 32);
            if (q == 0) zp[wave * 16 + lr] -= pr;
        }
    }
}

template <typename T, bool MIXED = false>
__global__ __launch_bounds__(512, 4) void gpcc_update_solve(GpccCtx c, GpccGroup g, int k)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int per = c.nt - k - 1;
    const int x = blockIdx.x & 7, qq = blockIdx.x >> 3;
    const int m = g.spread ? (int)blockIdx.x % g.cnt : (qq / per) * 8 + x;
    if (m >= g.cnt) return;
    const int I = k + 1 + (g.spread ? (int)blockIdx.x / g.cnt : qq % per);
    const int slot = g.slot0 + m;
    if (c.info[slot] != 0) return;
    gpcc_update_solve_job<T, MIXED>(c, k, I, slot, (T *)smem_raw);
}

// ------------------------------------------------------------------------------------------
// gpcc_panel_trsm (step k, tile I > k):  L(I,k) = T(I,k) inv(L_kk)^T  (dtrsm as an MFMA product
// with the explicit 128x128 inverse from gpcc_diag_factor; chunks of inv(L_kk) above the diagonal
// are zero and skipped), fused with the forward substitution of logpdf's whitening for every
// right-hand side: Z_I -= L(I,k) W_k (fp64).  Each wave owns 16 rows x all 128 columns, so the k-range of
// every column fragment is the same for all waves (balanced) and the row sums need no cross-wave reduction.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(512, 4) void gpcc_panel_trsm(GpccCtx c, GpccGroup g, int k)
{
    typedef GpccPrec<T> P;
    constexpr int CH = GPCC_CHUNK_BYTES / sizeof(T);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *smem = (T *)smem_raw;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform, and said so)
    const int lr = lane & 15, q = lane >> 4;
    const int sw = gpcc_sw(lr);

    const bool shared = c.share_p > k;   // inv(L_kk) and W_k of the shared prefix come from the leader
    const int per = shared ? c.nt - c.share_p : c.nt - k - 1;
    const int nmain = 8 * ((g.cnt + 7) / 8) * per;
    const int x = blockIdx.x & 7, qq = blockIdx.x >> 3;
    int m = g.spread ? (int)blockIdx.x % g.cnt : (qq / per) * 8 + x, I;
    if (shared && (int)blockIdx.x >= nmain) {   // the leader's own rows k+1 .. share_p-1
        m = 0;
        I = k + 1 + ((int)blockIdx.x - nmain);
    } else if (m >= g.cnt) {
        return;
    } else {
        I = (shared ? c.share_p : k + 1) + (g.spread ? (int)blockIdx.x / g.cnt : qq % per);
    }
    const int slot = g.slot0 + m;
    const int lslot = shared ? g.slot0 : slot;
    if (c.info[slot] != 0 || gpcc_leader_failure(c, g) != 0) return;

    T *tiles = (T *)c.tiles + (long)slot * c.slot_stride;
    T *Tt = tiles + gpcc_tile_off(I, k);
    const T *gA = (const T *)gpcc_uniform_ptr(Tt);                                                // the chunks of T(I,k), overwritten at the end
    const T *gB = (const T *)gpcc_uniform_ptr((const T *)c.linv + gpcc_linv_off(c, lslot, k));  // inv(L_kk): rows = output column, k = j

    gpcc_dma_chunk<T>(gA, gB, smem, wave, lane);
    typename P::acc_t acc[8];
#pragma unroll
    for (int fn = 0; fn < 8; ++fn)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[fn][r] = 0;
    const T *pa0 = smem + (wave * 16 + lr) * P::KC + (((2 * q) ^ sw) * P::EP);
    const T *pa1 = smem + (wave * 16 + lr) * P::KC + (((2 * q + 1) ^ sw) * P::EP);
    const T *pb0 = smem + CH + lr * P::KC + (((2 * q) ^ sw) * P::EP);
    const T *pb1 = smem + CH + lr * P::KC + (((2 * q + 1) ^ sw) * P::EP);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int ch = 0; ch < P::NCH; ++ch) {
        const int so = (ch & 1) * 2 * CH;
        if (ch + 1 < P::NCH)
            gpcc_dma_chunk<T>(gA + (ch + 1) * CH, gB + (ch + 1) * CH, smem + ((ch + 1) & 1) * 2 * CH, wave, lane);
        typename P::v16 a[2];
        a[0] = *(const typename P::v16 *)(pa0 + so);
        a[1] = *(const typename P::v16 *)(pa1 + so);
        constexpr int FPC = P::KC / 16;  // column fragments per chunk width
#pragma unroll
        for (int h = 0; h < 2; ++h) {  // column fragments in two groups of 4 (register budget)
            if (4 * h + 3 < ch * FPC) continue;   // fragments left of chunk ch see only zeros of inv(L_kk)
            typename P::v16 b[4][2];
#pragma unroll
            for (int f = 0; f < 4; ++f)
                if (4 * h + f >= ch * FPC) {
                    b[f][0] = *(const typename P::v16 *)(pb0 + so + (4 * h + f) * 16 * P::KC);
                    b[f][1] = *(const typename P::v16 *)(pb1 + so + (4 * h + f) * 16 * P::KC);
                }
#pragma unroll
            for (int s = 0; s < P::KSTEPS; ++s)
#pragma unroll
                for (int f = 0; f < 4; ++f)
                    if (4 * h + f >= ch * FPC)
                        acc[4 * h + f] = P::mfma(a[s / P::EP][s % P::EP], b[f][s / P::EP][s % P::EP], acc[4 * h + f]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int fn = 0; fn < 8; ++fn) Tt[gpcc_elem_off<T>(wave * 16 + P::crow(q, r), fn * 16 + lr)] = acc[fn][r];
    for (int j = 0; j < c.nrhs; ++j) {
        double *zp = c.z + ((long)slot * c.nrhs + j) * c.Np + I * GPCC_TILE;
        const double *wp = c.w + ((long)lslot * c.nrhs + j) * c.Np + k * GPCC_TILE;
        double wv[8], zold[4];
#pragma unroll
        for (int fn = 0; fn < 8; ++fn) wv[fn] = wp[fn * 16 + lr];
#pragma unroll
        for (int r = 0; r < 4; ++r) zold[r] = zp[wave * 16 + P::crow(q, r)];   // all loads first: one round trip, not four
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int R = wave * 16 + P::crow(q, r);
            double p = 0.0;
#pragma unroll
            for (int fn = 0; fn < 8; ++fn) p += (double)acc[fn][r] * wv[fn];
            p += __shfl_xor(p, 1);
            p += __shfl_xor(p, 2);
            p += __shfl_xor(p, 4);
            p += __shfl_xor(p, 8);
            if (lr == 0) zp[R] = zold[r] - p;
        }
    }
}

// ------------------------------------------------------------------------------------------
// gpcc_panel_trsm_rows: the panel solve for a FEW evaluations (gpcc_small_step's companion).  A 128x128 tile job is ~8 us
// of one CU's fp64 matrix pipe, and with one or two evaluations only nt-k-1 CUs would work; here a job is a QUARTER of a
// tile (32 rows), four times as many CUs per step, and the operand stream of a job runs three chunks ahead (LDS ring of
// four 20 KiB stages: 4 KiB of T(I,k) rows + a 16 KiB chunk of inv(L_kk)).  Wave (rh, cp): rows 16 rh .. +15 of the
// quarter, column fragments cp and 7-cp (the triangular inverse makes fragment f need chunks <= f: every wave the same work).
// Fused with the forward substitution z_I -= L(I,k) w_k like gpcc_panel_trsm.   grid cnt * (nt-k-1) * 4, block 512.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(512) void gpcc_panel_trsm_rows(GpccCtx c, GpccGroup g, int k)
{
    typedef GpccPrec<T> P;
    constexpr int CH = GPCC_CHUNK_BYTES / sizeof(T);          // elements of a full chunk (B operand)
    constexpr int AQ = CH / 4;                                // elements of 32 rows of a chunk (A operand)
    constexpr int STG = AQ + CH;                              // one ring stage
    constexpr int STAGES = 4;
    constexpr int PIECE = 1024 / sizeof(T), EPB = 16 / sizeof(T);
    constexpr int FPC = P::KC / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *smem = (T *)smem_raw;
    __shared__ double sred[8][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform, and said so)
    const int lr = lane & 15, q = lane >> 4, sw = gpcc_sw(lr);
    const int rh = wave & 1, cp = wave >> 1, f0 = cp, f1 = 7 - cp;
    const int m = (int)blockIdx.x % g.cnt, rest = (int)blockIdx.x / g.cnt;
    const int I = k + 1 + rest / 4, qr = rest % 4;
    const int slot = g.slot0 + m;
    if (c.info[slot] != 0) return;
    T *Tt = (T *)c.tiles + (long)slot * c.slot_stride + gpcc_tile_off(I, k);
    const T *gA = Tt + qr * AQ;                                        // rows 32 qr .. +31 of every chunk
    const T *gB = (const T *)c.linv + gpcc_linv_off(c, slot, k);
    auto dma = [&](int ch, int st) {   // three 1 KiB pieces per wave and chunk (the A pieces of waves 4-7 repeat those of 0-3)
        T *stage = smem + st * STG;
        const int uw = __builtin_amdgcn_readfirstlane(wave), pa = uw & 3;
        const unsigned voff = (unsigned)lane * 16u;
        gpcc_dma_piece<0>(gpcc_uniform_ptr(gA + (long)ch * CH + pa * PIECE), voff, gpcc_lds_addr(stage + pa * PIECE));
        const void *pbase = gpcc_uniform_ptr(gB + (long)ch * CH + uw * 2 * PIECE);
        const unsigned lb = gpcc_lds_addr(stage + AQ + uw * 2 * PIECE);
        gpcc_dma_piece2(pbase, voff, lb);
    };
#pragma unroll
    for (int pc = 0; pc < STAGES - 1 && pc < P::NCH; ++pc) dma(pc, pc);
    typename P::acc_t acc0, acc1;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc0[r] = acc1[r] = 0;
    const T *pa0 = smem + (rh * 16 + lr) * P::KC + (((2 * q) ^ sw) * P::EP);
    const T *pa1 = smem + (rh * 16 + lr) * P::KC + (((2 * q + 1) ^ sw) * P::EP);
    const T *pb0 = smem + AQ + lr * P::KC + (((2 * q) ^ sw) * P::EP);
    const T *pb1 = smem + AQ + lr * P::KC + (((2 * q + 1) ^ sw) * P::EP);
    if (P::NCH >= 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // chunk 0 has landed (3 DMAs per chunk and wave)
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int ch = 0; ch < P::NCH; ++ch) {
        if (ch + STAGES - 1 < P::NCH) dma(ch + STAGES - 1, (ch + STAGES - 1) % STAGES);
        const int so = (ch % STAGES) * STG;
        typename P::v16 a[2];
        a[0] = *(const typename P::v16 *)(pa0 + so);
        a[1] = *(const typename P::v16 *)(pa1 + so);
        if (f0 >= ch * FPC) {   // (wave-uniform)
            typename P::v16 b[2];
            b[0] = *(const typename P::v16 *)(pb0 + so + f0 * 16 * P::KC);
            b[1] = *(const typename P::v16 *)(pb1 + so + f0 * 16 * P::KC);
#pragma unroll
            for (int s = 0; s < P::KSTEPS; ++s) acc0 = P::mfma(a[s / P::EP][s % P::EP], b[s / P::EP][s % P::EP], acc0);
        }
        if (f1 >= ch * FPC) {
            typename P::v16 b[2];
            b[0] = *(const typename P::v16 *)(pb0 + so + f1 * 16 * P::KC);
            b[1] = *(const typename P::v16 *)(pb1 + so + f1 * 16 * P::KC);
#pragma unroll
            for (int s = 0; s < P::KSTEPS; ++s) acc1 = P::mfma(a[s / P::EP][s % P::EP], b[s / P::EP][s % P::EP], acc1);
        }
        if (ch + 1 < P::NCH) {
            const int later = ((ch + STAGES - 1 < P::NCH - 1) ? ch + STAGES - 1 : P::NCH - 1) - (ch + 1);   // chunks issued beyond ch+1
            if (later <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (later == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            __syncthreads();
        }
    }
    const int row0 = qr * 32 + rh * 16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        Tt[gpcc_elem_off<T>(row0 + P::crow(q, r), f0 * 16 + lr)] = acc0[r];
        Tt[gpcc_elem_off<T>(row0 + P::crow(q, r), f1 * 16 + lr)] = acc1[r];
    }
    // forward substitution: z_I[row] -= sum_c L(I,k)[row][c] w_k[c]; a wave holds 32 of the 128 columns of its 16 rows
    for (int j = 0; j < c.nrhs; ++j) {
        double *zp = c.z + ((long)slot * c.nrhs + j) * c.Np + I * GPCC_TILE;
        const double *wp = c.w + ((long)slot * c.nrhs + j) * c.Np + k * GPCC_TILE;
        const double w0 = wp[f0 * 16 + lr], w1 = wp[f1 * 16 + lr];
        __syncthreads();   // (sred of the previous right-hand side has been consumed)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double pr = (double)acc0[r] * w0 + (double)acc1[r] * w1;
            pr += __shfl_xor(pr, 1);
            pr += __shfl_xor(pr, 2);
            pr += __shfl_xor(pr, 4);
            pr += __shfl_xor(pr, 8);
            if (lr == 0) sred[wave][P::crow(q, r)] = pr;
        }
        __syncthreads();
        if (tid < 32) {   // row tid of the quarter: half rh2, the four column pairs in a fixed order
            const int rh2 = tid >> 4, rr = tid & 15;
            const double sum = ((sred[rh2][rr] + sred[2 + rh2][rr]) + sred[4 + rh2][rr]) + sred[6 + rh2][rr];
            zp[qr * 32 + tid] -= sum;
        }
    }
}

// ------------------------------------------------------------------------------------------
// gpcc_diag_factor: step k's 128x128 diagonal block, one workgroup (8 waves) per evaluation, all in
// LDS and ALWAYS in fp64 (an fp32 tile is widened on load, inv(L_kk) and L_kk are rounded on store):
// blocked dpotf2 (16-wide panels), its triangular inverse (the B operand of the MFMA panel
// solve), sum log L_ii (logdet of PDMat), W_k = inv(L_kk) Z_k and the Gram matrix W^T W (sqmahal);
// the last step writes loglik = -(N log 2pi + logdet K)/2 - (Y-bbar)' K^-1 (Y-bbar)/2
// (Distributions.logpdf, marginaliseb.jl:139), in woodbury mode through the L x L capacitance matrix.
//   per 16-block jb: (A) wave 0 factors the 16x16 diagonal block and inverts it in REGISTERS (lanes =
//   rows of L and columns of the inverse, broadcasts by v_readlane);  (B) panel rows below: P = A inv(D)^T by
//   MFMA;  (C) trailing update C -= P P^T by MFMA;  (X) row jb of inv(L) block-wise by MFMA: the accumulator
//   of S = sum_m L[i][m] X[m][j] is directly the B operand of X[i][j] = -inv(D_i) S.  Software-pipelined:
//   (A) of block jb+1 runs on wave 0 while six worker waves do (C), (X) and the forward substitution (W) of block jb.
//   X's off-diagonal blocks live transposed in the (dead) upper triangle of the LDS image.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double gpcc_bcast(double v, int srclane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
    return __hiloint2double(hi, lo);
}

// 1/sqrt(d) from ONE v_rsq_f64 (about 2^-23 relative) and two Newton steps y <- y + y (1/2 - (d/2) y^2): within
// about 1 ulp, a third of the instructions of sqrt() followed by an IEEE division.  The diagonal entry itself is then
// L_jj = d / sqrt(d) (one more rounding).  Pivots are O(sigma^2 .. alpha^2 N): no scaling for denormals; d <= 0 / NaN
// gives NaN (the caller has flagged the pivot by then).
__device__ __forceinline__ double gpcc_rsqrt(double d)
{
    const double hd = 0.5 * d;
    double y = __builtin_amdgcn_rsq(d);
    double r = __builtin_fma(-hd, y * y, 0.5);
    y = __builtin_fma(y, r, y);
    r = __builtin_fma(-hd, y * y, 0.5);
    return __builtin_fma(y, r, y);
}

// ------------------------------------------------------------------------------------------
// gpcc_potf2_core: the 16 x 16 Cholesky + inverse in the registers of ONE wave -- the serial heart of every diagonal step (tile
// kernels: gpcc_diag_body; the persistent few-evaluation launch: gpcc_chain_diag; the small-N family: gpcc_small_potf2).
// Lanes 0-15 own the rows of D (v[cc] = D[l][cc]), lanes 16-31 the columns of X = inv(L_D) (v[cc] = delta(cc, l) on entry), ONE
// right-looking instruction stream for both: once column j of L is final, v[j] <- v[j] / sqrt(d_j) is L[l][j] on an L lane and
// X[j][l] on an X lane, and the same v[cc] -= v[j] L[cc][j] updates the trailing row and the running sums of the inverse.
// Round 5: the pivot-to-pivot chain no longer goes through a lane broadcast.  The next pivot
//     d_{j+1} = A_j[j+1][j+1] - (A_j[j+1][j] / sqrt(d_j))^2
// needs two entries of row j + 1 in their state BEFORE pivot j -- both known while 1/sqrt(d_j) is still being computed -- so they are
// broadcast (v_readlane) early, off the chain, and d_{j+1} = fma(-lnx, lnx, e) with lnx = c y follows y in two operations on values
// every lane holds.  Likewise the entry of row j + 2 that column j + 2's update needs (g): columns j + 1 and j + 2 receive pivot j's
// update at once from uniform multipliers, the other columns one pivot later from the column broadcast through LDS (one store,
// uniform-address loads), whose round trip is then two pivots off the chain.  Chain per pivot: mul -> fma -> v_rsq_f64 -> two Newton
// steps = 9 dependent operations (it was those + 2 broadcasts + their hazard waits); the independent rank-1 updates are written
// BETWEEN the dependent operations and pinned there (sched_barrier): the compiler's own schedule put them behind the chain, so that
// wave 0 -- one in-order instruction stream -- paid chain latency PLUS their issue time, ~330 cycles per pivot.
// Every value is formed by the same operations on the same operands as before: the same bits (1/sqrt(d): v_rsq_f64 + two Newton
// steps, ~1 ulp; L_jj = d / sqrt(d)).  sum log L_jj of the block = -log prod 1/sqrt(d_j): the running product py (renormalised every
// four pivots: mantissa py, exponent sum pe -- no overflow whatever the scale of K).
// RHS (small-N family): with `last`, pivot 15 is the right-hand-side row -- its Schur complement is -w'w (quad), not a pivot.
// TRACK (fp32 tile path): sum and maximum of K_ii / d_i over the pivots (skd = diag(K) of these 16 rows).
// Returns j + 1 for the first non-positive (or NaN) pivot j, 0 if none.  sr: 80 doubles of LDS scratch.
// ------------------------------------------------------------------------------------------
template <bool RHS, bool TRACK>
__device__ __forceinline__ int gpcc_potf2_core(double (&v)[16], double *sr, const int lrv, const int qv, const int lane, const bool last, double &py, int &pe,
                                               double &quad, const double *skd, double &rs, double &rm)
{
    double cn[2][16];   // column j of D BEFORE its scaling, broadcast through LDS: set j & 1 (loaded during pivot j, consumed during pivot j + 1)
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) cn[0][cc] = cn[1][cc] = 0.0;
    double d = gpcc_bcast(v[0], 0), wp = 0.0;
    unsigned badm = 0u;   // bit j: pivot j was not positive (resolved once, behind the loop: as a running "first bad pivot" the compiler kept 16
                          // flags and walked them in ~100 scalar instructions at the end of every block -- on the critical path of every diagonal step)
    double t1 = 1.0, t2 = 1.0, t3 = 1.0, t4 = 1.0;   // prod 1/sqrt(d_j) as a pairwise tree: 4 multiplications behind the last pivot, not 16
    __builtin_amdgcn_sched_barrier(0);
    // pivot j - 1's update of column cc >= j + 2, one pivot late: v[cc] -= L[l][j-1] L[cc][j-1] = (v[j-1] y y) * (column entry before scaling)
#define GPCC_POTF2_FILL(k)                                                                                        \
    do {                                                                                                          \
        if (j >= 1 && j + 2 + (k) <= 15) v[j + 2 + (k)] = __builtin_fma(-wp, cn[(j - 1) & 1][j + 2 + (k)], v[j + 2 + (k)]); \
    } while (0)
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (RHS && j == 15 && last) {   // the right-hand-side row: its Schur complement is -w'w; not a pivot
            quad = -d;
            d = 1.0;
        }
        badm |= (__builtin_amdgcn_fcmp(d, 0.0, 2 /* ordered > */) == 0ull) ? (1u << j) : 0u;   // (d is the same in every lane: scalar; also catches NaN)
        const double hd = -0.5 * d;
        double y = __builtin_amdgcn_rsq(d);
        // off the chain, while 1/sqrt(d) is on its way: the entries of rows j + 1, j + 2 the NEXT pivot needs (columns j and j + 1 are
        // in their final pre-pivot state), and column j itself -- unscaled -- through LDS for the updates of the columns beyond j + 2
        double cj = 0.0, ej = 0.0, gj = 0.0;
        if (j < 15) {
            cj = gpcc_bcast(v[j], j + 1);
            ej = gpcc_bcast(v[j + 1], j + 1);
        }
        if (j < 14) gj = gpcc_bcast(v[j], j + 2);
        if (j < 13) {
            sr[qv == 0 ? lrv : 16 + lane] = v[j];   // (the other lanes store to a dead area: no branch)
#pragma unroll
            for (int cc = j + 3; cc < 16; ++cc) cn[j & 1][cc] = sr[cc];
        }
        GPCC_POTF2_FILL(0);
        __builtin_amdgcn_sched_barrier(0);
        double t = y * y;                       // Newton: y <- y + y (1/2 - (d/2) y^2), twice
        GPCC_POTF2_FILL(1);
        __builtin_amdgcn_sched_barrier(0);
        double r = __builtin_fma(hd, t, 0.5);
        GPCC_POTF2_FILL(2);
        __builtin_amdgcn_sched_barrier(0);
        y = __builtin_fma(y, r, y);
        GPCC_POTF2_FILL(3);
        __builtin_amdgcn_sched_barrier(0);
        t = y * y;
        GPCC_POTF2_FILL(4);
        __builtin_amdgcn_sched_barrier(0);
        r = __builtin_fma(hd, t, 0.5);
        GPCC_POTF2_FILL(5);
        __builtin_amdgcn_sched_barrier(0);
        y = __builtin_fma(y, r, y);
        GPCC_POTF2_FILL(6);
        __builtin_amdgcn_sched_barrier(0);
        double lnx = 0.0;
        if (j < 15) lnx = cj * y;               // L[j+1][j]
        GPCC_POTF2_FILL(7);
        __builtin_amdgcn_sched_barrier(0);
        if (j < 15) d = __builtin_fma(-lnx, lnx, ej);   // the next pivot: its v_rsq_f64 is the next instruction of the chain
        __builtin_amdgcn_sched_barrier(0);
        GPCC_POTF2_FILL(8);
        GPCC_POTF2_FILL(9);
        GPCC_POTF2_FILL(10);
        GPCC_POTF2_FILL(11);
        GPCC_POTF2_FILL(12);
        GPCC_POTF2_FILL(13);
        if ((j & 1) == 0) {
            t1 = y;
        } else {
            const double p1 = t1 * y;
            if ((j & 3) == 1) {
                t2 = p1;
            } else {
                double p2 = t2 * p1;             // four pivots: mantissa / exponent (no overflow whatever the scale of K)
                pe += __builtin_amdgcn_frexp_exp(p2);
                p2 = __builtin_amdgcn_frexp_mant(p2);
                if ((j & 7) == 3) {
                    t3 = p2;
                } else {
                    const double p3 = t3 * p2;
                    if (j == 7) t4 = p3;
                    else py *= t4 * p3;
                }
            }
        }
        if (TRACK) {
            const double ratio = skd[j] * (y * y);
            rs += ratio;
            rm = fmax(rm, ratio);
        }
        v[j] *= y;                              // L[l][j]; lane j: L[j][j] = d / sqrt(d)
        if (j < 15) v[j + 1] = __builtin_fma(-v[j], lnx, v[j + 1]);
        if (j < 14) {
            const double lg = gj * y;           // L[j+2][j]
            v[j + 2] = __builtin_fma(-v[j], lg, v[j + 2]);
        }
        if (j < 13) wp = v[j] * y;
        __builtin_amdgcn_sched_barrier(0);
    }
#undef GPCC_POTF2_FILL
    pe += __builtin_amdgcn_frexp_exp(py);
    py = __builtin_amdgcn_frexp_mant(py);
    return badm ? (int)__builtin_ctz(badm) + 1 : 0;
}

// loglik = -(N log 2pi + logdet K)/2 - (Y-bbar)' K^-1 (Y-bbar)/2  (Distributions.logpdf, marginaliseb.jl:139) from
// ld = sum_i log L_ii and the Gram matrix G = W'W of the whitened right-hand sides (nrhs x nrhs, modified in place).
// woodbury: the matrix that was factorised is K0 and R = [Q | r]:
//   K = K0 + Q Sigma_b Q':  logdet K = logdet K0 + sum log Sigma_b + logdet M,
//   r'K^-1 r = r'K0^-1 r - v' M^-1 v,  M = Sigma_b^-1 + Q'K0^-1 Q, v = Q'K0^-1 r   (all entries of G);
// a non-positive pivot of M sets bad = N + 1 + j.
__device__ __forceinline__ double gpcc_loglik_from_gram(const GpccCtx &c, double *sG, int nrhs, double ld, int &bad)
{
            const double log2pi = 1.8378770664093454835606594728112;
            double logdetK = 2.0 * ld, quad = sG[nrhs * nrhs - 1];
            if (c.woodbury && !bad) {
                // K = K0 + Q Sigma_b Q':  logdet K = logdet K0 + sum log Sigma_b + logdet M,
                // r'K^-1 r = r'K0^-1 r - v' M^-1 v,  M = Sigma_b^-1 + Q'K0^-1 Q, v = Q'K0^-1 r   (all from W'W)
                // (in place in the LDS copy of the Gram matrix: M(a,b) = sG[a*nrhs+b], v(a) = sG[a*nrhs+L])
                const int Lb = c.L;
                for (int a = 0; a < Lb; ++a) {
                    if (c.sigma_b[a] > 0.0) {
                        sG[a * nrhs + a] += 1.0 / c.sigma_b[a];
                        logdetK += log(c.sigma_b[a]);
                    } else {
                        // Sigma_b[a] = 0 (a band of constant fluxes): column a of Q contributes nothing to K; drop it
                        // from the capacitance system (unit pivot, zero couplings) -- K = K0 + Sobs is still PD
                        for (int p2 = 0; p2 <= Lb; ++p2) sG[a * nrhs + p2] = 0.0;
                        for (int i = 0; i < Lb; ++i) sG[i * nrhs + a] = 0.0;
                        sG[a * nrhs + a] = 1.0;
                    }
                }
                for (int j = 0; j < Lb; ++j) {  // Cholesky of M + forward solve
                    double d = sG[j * nrhs + j];
                    for (int p2 = 0; p2 < j; ++p2) d -= sG[j * nrhs + p2] * sG[j * nrhs + p2];
                    if (!(d > 0.0)) { bad = c.N + 1 + j; break; }
                    d = sqrt(d);
                    sG[j * nrhs + j] = d;
                    for (int i = j + 1; i < Lb; ++i) {
                        double s = sG[i * nrhs + j];
                        for (int p2 = 0; p2 < j; ++p2) s -= sG[i * nrhs + p2] * sG[j * nrhs + p2];
                        sG[i * nrhs + j] = s / d;
                    }
                    double s = sG[j * nrhs + Lb];
                    for (int p2 = 0; p2 < j; ++p2) s -= sG[j * nrhs + p2] * sG[p2 * nrhs + Lb];
                    s /= d;
                    sG[j * nrhs + Lb] = s;
                    logdetK += 2.0 * log(d);
                    quad -= s * s;
                }
            }
            return -((double)c.N * log2pi + logdetK) / 2.0 - quad / 2.0;
}

// PRELOADED: the caller (gpcc_small_step) has already put the updated tile into sT (fp64 image of the values of type T).
template <typename T, bool PRELOADED>
__device__ __forceinline__ void gpcc_diag_body(const GpccCtx &c, const GpccGroup &g, const int k, const int m, double *smem)
{
    typedef GpccPrec<T> P;
    typedef GpccPrec<double> PD;
    constexpr int LD = GPCC_DIAG_LD, DLD = GPCC_DINV_LD;
    double *sT = smem;                          // 128 x LD: lower = A -> L ; upper = inv(L)^T off-diagonal blocks
    double *sDinv = sT + GPCC_TILE * LD;        // 8 x 16 x DLD: inverses X_b of the 16x16 diagonal blocks, TRANSPOSED:
                                                // X_b[r][c] at sDinv[(16 b + c) DLD + r] (a column of X_b is contiguous)
    double *sz = sDinv + 8 * 16 * DLD;          // nrhs x 128: Z_k, later W_k
    double *sr = sz + GPCC_MAXRHS * GPCC_TILE;  // [0, 80): wave 0's column scratch in (A); [96, 112): prod 1/L_jj per 16-block (mantissa, exponent)
    double *skd = sr + GPCC_TILE;               // diag(K) of this block's rows as assembled (fp32 mode)
    double *sG = skd + GPCC_TILE;               // nrhs x nrhs Gram matrix
    double *sld = sG + GPCC_MAXRHS * GPCC_MAXRHS;   // sum log L_ii of this block
    int *sbad = (int *)(sG + GPCC_MAXRHS * GPCC_MAXRHS + 1);

    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform: wave-indexed loops and addresses stay scalar
    // Eight waves = two per SIMD: one wave alone issues an f64 MFMA only every ~139 cycles (profiles/r01/microbench_fp64.log),
    // two keep the pipe busy.  Wave 0 runs the register factorisations (A); wave 4 shares its SIMD and stays out of its
    // way; the other six are the MFMA workers of (C), (X), (W).
    constexpr int NWK = 6;
    const int wk = (wave == 0 || wave == 4) ? -1 : (wave < 4 ? wave - 1 : wave - 2);
    const int slot = g.slot0 + m, nrhs = c.nrhs;
    const bool last = (k == c.nt_fact - 1);
    int inf = c.info[slot];
    if (inf == 0 && m > 0) inf = gpcc_leader_failure(c, g);   // a failed shared prefix fails its followers too
    if (inf != 0) {
        if (last && tid == 0) {
            g.out_loglik[g.first + m] = __builtin_nan("");
            g.out_info[g.first + m] = inf;
        }
        return;
    }
    T *tiles = (T *)c.tiles + (long)slot * c.slot_stride;
    T *Tt = tiles + gpcc_tile_off(k, k);
    // 16-byte pieces, all of a thread's loads in flight (one load per iteration would serialise the HBM round trips)
    constexpr int NPIECE = GPCC_TILE_ELEMS / P::EP;   // pieces per tile
    constexpr int NT = GPCC_DIAG_THREADS;
    constexpr int UL = (NPIECE / NT < 16) ? NPIECE / NT : 16;   // loads in flight per thread (fp64: 16, fp32: 8)
    static_assert(NPIECE % (NT * UL) == 0, "tile pieces must divide evenly over the threads");
    if (!PRELOADED)
    for (int p0 = tid; p0 < NPIECE; p0 += NT * UL) {
        typename P::v16 v[UL];
#pragma unroll
        for (int u = 0; u < UL; ++u) v[u] = *(const typename P::v16 *)(Tt + (long)(p0 + NT * u) * P::EP);
#pragma unroll
        for (int u = 0; u < UL; ++u) {
            const int e = (p0 + NT * u) * P::EP;   // storage position of the piece's first element
            const int ch = e / (GPCC_TILE * P::KC), rem = e % (GPCC_TILE * P::KC), r = rem / P::KC, ks = rem % P::KC;
            const int kk = ((ks / P::EP) ^ gpcc_sw(r)) * P::EP;
#pragma unroll
            for (int h = 0; h < P::EP; ++h) sT[r * LD + ch * P::KC + kk + h] = (double)v[u][h];
        }
    }
    for (int e = tid; e < nrhs * GPCC_TILE; e += NT)
        sz[e] = c.z[((long)slot * nrhs + e / GPCC_TILE) * c.Np + k * GPCC_TILE + (e % GPCC_TILE)];
    if (tid == 0) *sbad = 0;
    const bool track = sizeof(T) == 4 && c.kdiag != nullptr;   // fp32 mode: pivot ratios K_ii / d_i
    if (track && tid < GPCC_TILE) skd[tid] = c.kdiag[(long)slot * c.Np + k * GPCC_TILE + tid];

    __syncthreads();   // tile and right-hand sides are in LDS
    // Software pipeline over the eight 16-column blocks.  Step jb: wave 0 folds column block jb-1 into the next
    // diagonal block and factors it (A), while the workers finish the trailing update of column block jb-1 (C) and
    // build row jb-1 of inv(L) (X) and of W; then all waves form the panel of column block jb (B).
    for (int jb = 0; jb <= 8; ++jb) {
        const int r0 = jb * 16, rp = r0 - 16;   // rp: first column of the previous block
        if (wave == 0) {
            if (jb > 0 && jb < 8) {   // C(jb-1) for the block the factorisation below needs: D_jb -= P_jb P_jb^T
                d4 x;
                double pa[4], pb[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) x[r] = sT[(r0 + q + 4 * r) * LD + r0 + lr];
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) {
                    pb[s2] = sT[(r0 + lr) * LD + rp + q + 4 * s2];
                    pa[s2] = -pb[s2];
                }
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) x = PD::mfma(pa[s2], pb[s2], x);
#pragma unroll
                for (int r = 0; r < 4; ++r) sT[(r0 + q + 4 * r) * LD + r0 + lr] = x[r];
            }
            if (jb < 8) {
                // ---- (A) 16x16 potf2 + inverse in registers.  Lanes 0-15: lane l owns row l of D (v[cc] = A[l][cc]);
                // lanes 16-31: lane 16 + l owns column l of X = inv(L_D) (v[cc] = delta(cc, l) - sum_j L[cc][j] X[j][l]);
                // lanes 32-63 shadow them.  Right-looking, ONE instruction stream for both: once column j of L is
                // final, v[j] <- v[j] / sqrt(d_j) is L[l][j] on an L lane and X[j][l] on an X lane, and the same
                // v[cc] -= v[j] L[cc][j] (L[cc][j] broadcast once from lane cc) updates the trailing row and the
                // running sums.  Entries right of the diagonal of D are never used (no masking on load, update or store).
                const bool xl = q != 0;
                double v[16];
                const double *row = sT + (r0 + lr) * LD + r0;   // 16 contiguous doubles, 16-byte aligned
#pragma unroll
                for (int cc = 0; cc < 16; ++cc) v[cc] = xl ? ((cc == lr) ? 1.0 : 0.0) : row[cc];
                double py = 1.0;   // prod of 1/sqrt(d_j) as mantissa ...
                int pe = 0;        // ... and exponent: no overflow whatever the scale of K
                double rs = 0.0, rm = 0.0, quad_ = 0.0;   // sum and max of K_ii / d_i over this block's pivots (fp32 mode only)
                const int bad = (sizeof(T) == 4 && track) ? gpcc_potf2_core<false, true>(v, sr, lr, q, lane, false, py, pe, quad_, skd + r0, rs, rm)
                                                          : gpcc_potf2_core<false, false>(v, sr, lr, q, lane, false, py, pe, quad_, skd + r0, rs, rm);
                if (lane < 32) {
                    // row l of L_D (its entries right of the diagonal are dead values in a dead area) resp. column l of
                    // X (exact zeros above the diagonal): 16 contiguous stores from every lane, no masks
                    double *dst = xl ? (sDinv + (jb * 16 + lr) * DLD) : (sT + (r0 + lr) * LD + r0);
#pragma unroll
                    for (int cc = 0; cc < 16; ++cc) dst[cc] = v[cc];
                    if (lane == 0 && bad && *sbad == 0) *sbad = r0 + bad;
                    if (lane == 0) {
                        sr[96 + jb] = py;
                        sr[104 + jb] = (double)pe;
                        sr[112 + jb] = rs;
                        sr[120 + jb] = rm;
                    }
                }
            }
        } else if (wk >= 0 && jb > 0) {
            const int jp = jb - 1;
            // ---- (C) rest of the trailing update of column block jp: C[rf][cf] -= P_rf P_cf^T, jp < cf <= rf, block
            // t = rr (rr+1)/2 + cc2 of the (7 - jp)-row triangle; t = 0 is wave 0's.  Two blocks at a time
            // (independent MFMA chains, all LDS reads of a pair in flight together).
            const int nb = 7 - jp, ntri = nb * (nb + 1) / 2;
            for (int t0 = 1 + wk; t0 < ntri; t0 += 2 * NWK) {
                int rfv[2], cfv[2];
                bool on[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int tt = t0 + NWK * u;
                    on[u] = tt < ntri;
                    const int te = on[u] ? tt : t0;
                    const int rr = (te >= 1) + (te >= 3) + (te >= 6) + (te >= 10) + (te >= 15) + (te >= 21);   // te < 28
                    rfv[u] = jp + 1 + rr;
                    cfv[u] = jp + 1 + te - rr * (rr + 1) / 2;
                }
                d4 x[2];
                double pa[2][4], pb[2][4];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) x[u][r] = sT[(rfv[u] * 16 + q + 4 * r) * LD + cfv[u] * 16 + lr];
#pragma unroll
                    for (int s2 = 0; s2 < 4; ++s2) {
                        pa[u][s2] = -sT[(rfv[u] * 16 + lr) * LD + rp + q + 4 * s2];
                        pb[u][s2] = sT[(cfv[u] * 16 + lr) * LD + rp + q + 4 * s2];
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
                        for (int r = 0; r < 4; ++r) sT[(rfv[u] * 16 + q + 4 * r) * LD + cfv[u] * 16 + lr] = x[u][r];
                    }
            }
        }
        if (jb > 0) {
            // ---- (X) row i = jb-1 of inv(L): X[i][j] = -inv(D_i) sum_{m = j}^{i-1} L[i][m] X[m][j], j < i (rows < i
            // are complete); the accumulator of the sum is directly the B operand of the product with inv(D_i).
            // Off-diagonal blocks live transposed in the (dead) upper triangle of the LDS image.
            // ---- (W) rows i of W_k = inv(L_kk) Z_k by forward substitution, W_i = inv(D_i) (Z_i - sum_{m<i} L[i][m] W_m):
            // the same two products with the right-hand sides (padded to 16 columns) in place of X[m][j]; in place in sz.
            // One task per worker (the two shortest share one); the last row (nothing left for wave 0 to factor) goes
            // over all eight waves.
            const int i = jb - 1;
            bool dow = false;
            int j0 = -1, j1 = -1;
            if (jb == 8) {          // all eight waves, one task each: W, X_0 .. X_6
                dow = wave == 0;
                j0 = wave - 1;
            } else if (wk >= 0) {   // tasks W, X_0, .., X_{i-1} (i <= 6) on the six workers; the two shortest share one
                dow = wk == 0;
                j0 = wk - 1;
                j1 = (wk == NWK - 1) ? NWK - 1 : -1;
            }
            if (dow) {
                const int zrow = (lr < nrhs) ? lr : 0;
                d4 S, S1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double zv = sz[zrow * GPCC_TILE + i * 16 + q + 4 * r];
                    S[r] = (lr < nrhs) ? -zv : 0.0;
                }
                for (int mm = 0; mm < i; ++mm) {
#pragma unroll
                    for (int s2 = 0; s2 < 4; ++s2) {
                        const double av = sT[(i * 16 + lr) * LD + mm * 16 + q + 4 * s2];   // L[i][mm]
                        const double wv = sz[zrow * GPCC_TILE + mm * 16 + q + 4 * s2];      // W_mm[k][rhs lr]
                        const double bv = (lr < nrhs) ? wv : 0.0;
                        if (s2 & 1) S1 = PD::mfma(av, bv, S1);
                        else S = PD::mfma(av, bv, S);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) S[r] += S1[r];
                d4 Y = {0.0, 0.0, 0.0, 0.0}, Y1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double dv = -sDinv[(i * 16 + q + 4 * r) * DLD + lr];
                    if (r & 1) Y1 = PD::mfma(dv, S[r], Y1);
                    else Y = PD::mfma(dv, S[r], Y);
                }
                if (lr < nrhs) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) sz[lr * GPCC_TILE + i * 16 + q + 4 * r] = Y[r] + Y1[r];
                }
            }
            for (int n = 0; n < 2; ++n) {
                const int j = n ? j1 : j0;
                if (j < 0 || j >= i) continue;
                d4 S = {0.0, 0.0, 0.0, 0.0}, S1 = {0.0, 0.0, 0.0, 0.0};   // two chains: half the dependent MFMA latency
                for (int mm = j; mm < i; ++mm) {
#pragma unroll
                    for (int s2 = 0; s2 < 4; ++s2) {
                        const double av = sT[(i * 16 + lr) * LD + mm * 16 + q + 4 * s2];               // L[i][mm]
                        const double bv = (mm == j) ? sDinv[(j * 16 + lr) * DLD + q + 4 * s2]          // X[j][j][k][c]
                                                    : sT[(j * 16 + lr) * LD + mm * 16 + q + 4 * s2];   // X[mm][j]^T
                        if (s2 & 1) S1 = PD::mfma(av, bv, S1);
                        else S = PD::mfma(av, bv, S);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) S[r] += S1[r];
                d4 Y = {0.0, 0.0, 0.0, 0.0}, Y1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int r = 0; r < 4; ++r) {  // the accumulator S (row q+4r, col lr) IS the B operand of k-step r
                    const double dv = -sDinv[(i * 16 + q + 4 * r) * DLD + lr];
                    if (r & 1) Y1 = PD::mfma(dv, S[r], Y1);
                    else Y = PD::mfma(dv, S[r], Y);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) sT[(j * 16 + lr) * LD + i * 16 + q + 4 * r] = Y[r] + Y1[r];   // X[i][j] transposed
            }
        }
        __syncthreads();
        if (jb == 8) break;
        // ---- (B) panel of column block jb: P = A[rf, jb] inv(D_jb)^T for the row fragments below, in place
        for (int rf = jb + 1 + wave; rf < 8; rf += NT / 64) {
            double av[4], bv[4];
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                av[s2] = sT[(rf * 16 + lr) * LD + r0 + q + 4 * s2];
                bv[s2] = sDinv[(jb * 16 + q + 4 * s2) * DLD + lr];   // B[k][c] = inv(D)[c][k]
            }
            d4 x = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) x = PD::mfma(av[s2], bv[s2], x);
#pragma unroll
            for (int r = 0; r < 4; ++r) sT[(rf * 16 + q + 4 * r) * LD + r0 + lr] = x[r];
        }
        __syncthreads();
    }
    // sz now holds W_k
    for (int e = tid; e < nrhs * GPCC_TILE; e += NT)
        c.w[((long)slot * nrhs + e / GPCC_TILE) * c.Np + k * GPCC_TILE + (e % GPCC_TILE)] = sz[e];
    // Gram matrix W^T W (accumulated over the steps) and sum log L_ii: one wave per entry, fixed reduction tree
    for (int e = wave; e < nrhs * nrhs; e += NT / 64) {
        const int ga = e / nrhs, gb = e % nrhs;
        double pr = sz[ga * GPCC_TILE + lane] * sz[gb * GPCC_TILE + lane] +
                    sz[ga * GPCC_TILE + 64 + lane] * sz[gb * GPCC_TILE + 64 + lane];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) pr += __shfl_xor(pr, o);
        if (lane == 0) {
            double *gp = c.gram + (long)slot * GPCC_MAXRHS * GPCC_MAXRHS + e;
            pr += *gp;
            *gp = pr;
            sG[e] = pr;
        }
    }
    if (wave == 3) {
        double pr = 0.0;   // sum log L_jj per 16-block
        if (lane < 8) pr = -(log(sr[96 + lane]) + sr[104 + lane] * 0.69314718055994530942);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) pr += __shfl_xor(pr, o);
        if (lane == 0) *sld = pr;
    }
    __syncthreads();
    if (c.share_p && k == c.share_p - 1) {
        // last step of the shared prefix (only the leader runs it): hand sum log L_ii and W'W over to the followers
        const double ldp = *sld + c.logdet[slot];   // same expression as the per-evaluation path below (bitwise identical results)
        double cs = 0.0, cm = 0.0;
        if (track) {
            cs = c.cond[2 * (long)slot];
            cm = c.cond[2 * (long)slot + 1];
            for (int b = 0; b < 8; ++b) {
                cs += sr[112 + b];
                cm = fmax(cm, sr[120 + b]);
            }
        }
        for (int f = 1 + tid; f < g.cnt; f += NT) {
            c.logdet[g.slot0 + f] = ldp;
            if (track) {
                c.cond[2 * (long)(g.slot0 + f)] = cs;
                c.cond[2 * (long)(g.slot0 + f) + 1] = cm;
            }
            for (int i = 0; i < nrhs * nrhs; ++i) c.gram[(long)(g.slot0 + f) * GPCC_MAXRHS * GPCC_MAXRHS + i] = sG[i];
        }
        __syncthreads();   // everybody has read the leader's running sum before thread 0 updates it below
    }
    if (tid == 0) {
        const double ld = *sld + c.logdet[slot];
        c.logdet[slot] = ld;
        if (track) {
            double cs = c.cond[2 * (long)slot], cm = c.cond[2 * (long)slot + 1];
            for (int b = 0; b < 8; ++b) {
                cs += sr[112 + b];
                cm = fmax(cm, sr[120 + b]);
            }
            c.cond[2 * (long)slot] = cs;
            c.cond[2 * (long)slot + 1] = cm;
            if (last && g.out_cond) {
                g.out_cond[2 * (long)(g.first + m)] = cs;
                g.out_cond[2 * (long)(g.first + m) + 1] = cm;
            }
        }
        int bad = *sbad;
        if (bad) c.info[slot] = k * GPCC_TILE + bad;
        if (last) {
            const double llv = gpcc_loglik_from_gram(c, sG, nrhs, ld, bad);
            g.out_loglik[g.first + m] = bad ? __builtin_nan("") : llv;
            g.out_info[g.first + m] = bad ? ((bad > c.N) ? bad : k * GPCC_TILE + bad) : 0;
        }
    }
    // ---- write inv(L_kk) (B operand of the panel solve) and L_kk, both in tile layout
    T *Linv = (T *)c.linv + gpcc_linv_off(c, slot, k);
    const bool store_l = c.store_l != 0;   // L_kk is read by nobody on the log-likelihood path (dense export only)
#pragma unroll 4
    for (int p0 = tid; p0 < NPIECE; p0 += NT) {   // 16-byte stores; unconditional LDS reads (address select), masked after
        const int e = p0 * P::EP;
        const int ch = e / (GPCC_TILE * P::KC), rem = e % (GPCC_TILE * P::KC), r = rem / P::KC, ks = rem % P::KC;
        const int col0 = ch * P::KC + ((ks / P::EP) ^ gpcc_sw(r)) * P::EP;
        typename P::v16 xo, lo;
#pragma unroll
        for (int h = 0; h < P::EP; ++h) {
            const int col = col0 + h;
            const double *src = ((col >> 4) == (r >> 4)) ? sDinv + ((r & ~15) + (col & 15)) * DLD + (r & 15) : sT + col * LD + r;
            const double xv = *src;
            xo[h] = (T)((col <= r) ? xv : 0.0);
        }
        *(typename P::v16 *)(Linv + e) = xo;
        if (store_l) {
#pragma unroll
            for (int h = 0; h < P::EP; ++h) lo[h] = (T)((col0 + h <= r) ? sT[r * LD + col0 + h] : 0.0);
            *(typename P::v16 *)(Tt + e) = lo;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(GPCC_DIAG_THREADS) void gpcc_diag_factor(GpccCtx c, GpccGroup g, int k)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    gpcc_diag_body<T, false>(c, g, k, blockIdx.x, smem);
}

// The lower triangle of a diagonal tile -- 36 of its 64 16x16 blocks, all the diagonal step reads -- dealt 5/5/5/5/4/4/4/4 to the
// eight waves (waves w and w+4 share a SIMD: one 5-block and one 4-block wave each): 56 % of the MFMA time of the full tile.
// Wave-specialised: NA blocks (RA, CA..CA+NA-1) and NB blocks (RB, CB..CB+NB-1).  ONE operand stream (both MFMA operands are
// fragments of the same tile row) through a 4-stage LDS ring with one barrier per chunk; the updated blocks (rounded to T, as the
// unfused path stores them) end up in the diagonal step's LDS image.  Used by gpcc_small_step (K = one tile) and gpcc_syrk_diag.
// BLK: the result goes into the packed 16 x 16 block image of the few-evaluation chain kernel (gpcc_chain.hip.h: gpcc_bi / gpcc_be) instead of
// the 128 x LD image.
__device__ __forceinline__ int gpcc_bi(int i, int j) { return (i * (i + 1) / 2 + j) * 256; }   // block (i, j), j <= i, in doubles
__device__ __forceinline__ int gpcc_be(int r, int c)                                              // element (r, c) of a block
{
    const int p = r ^ ((r >> 2) & 1);
    return p * 16 + (c ^ ((p >> 1) << 1));
}
template <typename T, int RA, int CA, int NA, int RB, int CB, int NB, bool BLK = false>
__device__ __forceinline__ void gpcc_syrk_lower_wave(const T *gRow, int nch, T *smem, const T *Tt, double *smem_d, int wave, int lane)
{
    typedef GpccPrec<T> P;
    constexpr int CH = GPCC_CHUNK_BYTES / sizeof(T), PIECE = 1024 / sizeof(T), EPB = 16 / sizeof(T), STAGES = 4;
    const int lr = lane & 15, q = lane >> 4, sw = gpcc_sw(lr);
    auto dma = [&](int ch) {   // two 1 KiB pieces per wave and chunk
        T *stage = smem + (ch % STAGES) * CH;
        const int uw = __builtin_amdgcn_readfirstlane(wave);
        const void *pbase = gpcc_uniform_ptr(gRow + (long)ch * CH + uw * 2 * PIECE);
        const unsigned la = gpcc_lds_addr(stage + uw * 2 * PIECE), voff = (unsigned)lane * 16u;
        gpcc_dma_piece2(pbase, voff, la);
    };
    typename P::acc_t acc[NA + NB];
#pragma unroll
    for (int i = 0; i < NA + NB; ++i) {
        const int R = (i < NA) ? RA : RB, C = (i < NA) ? CA + i : CB + (i - NA);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][r] = -Tt[gpcc_elem_off<T>(16 * R + P::crow(q, r), 16 * C + lr)];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the accumulator loads: the DMA counting below starts clean)
    for (int pc = 0; pc < STAGES - 1 && pc < nch; ++pc) dma(pc);
    const T *p0 = smem + lr * P::KC + (((2 * q) ^ sw) * P::EP);
    const T *p1 = smem + lr * P::KC + (((2 * q + 1) ^ sw) * P::EP);
    for (int ch = 0; ch < nch; ++ch) {
        // chunk ch must have landed; chunks ch+1, ch+2 (issued earlier) may still fly: 2 DMAs per chunk and wave.  ONE barrier
        // per chunk: behind it every wave has its pieces of chunk ch in LDS and has finished chunk ch-1, whose stage
        // ((ch-1) % 4 = (ch+3) % 4) is therefore free for the DMA of chunk ch+3 issued right after.
        const int later = ((ch + STAGES - 2 < nch - 1) ? ch + STAGES - 2 : nch - 1) - ch;
        if (later >= 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (later == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (ch + STAGES - 1 < nch) dma(ch + STAGES - 1);
        const int so = (ch % STAGES) * CH;
        typename P::v16 aA[2], aB[2];
        aA[0] = *(const typename P::v16 *)(p0 + so + RA * 16 * P::KC);
        aA[1] = *(const typename P::v16 *)(p1 + so + RA * 16 * P::KC);
        if (NB > 0) {
            aB[0] = *(const typename P::v16 *)(p0 + so + RB * 16 * P::KC);
            aB[1] = *(const typename P::v16 *)(p1 + so + RB * 16 * P::KC);
        }
#pragma unroll
        for (int i = 0; i < NA + NB; ++i) {
            const int C = (i < NA) ? CA + i : CB + (i - NA);
            typename P::v16 b[2];
            b[0] = *(const typename P::v16 *)(p0 + so + C * 16 * P::KC);
            b[1] = *(const typename P::v16 *)(p1 + so + C * 16 * P::KC);
#pragma unroll
            for (int s = 0; s < P::KSTEPS; ++s)
                acc[i] = P::mfma((i < NA) ? aA[s / P::EP][s % P::EP] : aB[s / P::EP][s % P::EP], b[s / P::EP][s % P::EP], acc[i]);
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NA + NB; ++i) {
        const int R = (i < NA) ? RA : RB, C = (i < NA) ? CA + i : CB + (i - NA);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (BLK) smem_d[gpcc_bi(R, C) + gpcc_be(P::crow(q, r), lr)] = (double)(T)(-acc[i][r]);
            else smem_d[(16 * R + P::crow(q, r)) * GPCC_DIAG_LD + 16 * C + lr] = (double)(T)(-acc[i][r]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// gpcc_small_step (step k of a group of a FEW evaluations, e.g. the single objective(alpha, rho) of Nelder-Mead): the
// right-looking trailing update T(I,J) -= L(I,k) L(J,k)^T for every tile k < J <= I, and -- in the workgroup that owns
// tile (k+1,k+1) -- straight on into gpcc_diag_body of step k+1 with the updated tile handed over in LDS (no global round
// trip, no launch in between).  One evaluation's critical path is diag -> solve -> update of ONE tile -> diag ...; with
// the diagonal step inside the update launch the rest of the trailing update runs beside it instead of before it:
// launches per evaluation 3 nt - 2 -> 2 nt - 1.   grid cnt * n(n+1)/2 (n = nt-k-1), block 512, LDS = the diagonal image.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(512) void gpcc_small_step(GpccCtx c, GpccGroup g, int k)
{
    typedef GpccPrec<T> P;
    constexpr int CH = GPCC_CHUNK_BYTES / sizeof(T);
    extern __shared__ __attribute__((aligned(16))) double smem_d[];
    T *smem = (T *)smem_d;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform, and said so)
    const int wr = wave >> 1, wc = wave & 1, lr = lane & 15, q = lane >> 4;
    const int sw = gpcc_sw(lr);
    const int m = (int)blockIdx.x % g.cnt, j = (int)blockIdx.x / g.cnt;   // job j of evaluation m; j = 0 is tile (k+1,k+1)
    const int slot = g.slot0 + m;
    if (c.info[slot] != 0) {
        if (j == 0) gpcc_diag_body<T, true>(c, g, k + 1, m, smem_d);   // (reports the failure on the last step, touches no LDS)
        return;
    }
    int a = (int)((sqrtf(8.0f * j + 1.0f) - 1.0f) * 0.5f);
    while (a * (a + 1) / 2 > j) --a;
    while ((a + 1) * (a + 2) / 2 <= j) ++a;
    const int I = k + 1 + a, J = k + 1 + (j - a * (a + 1) / 2);
    T *tiles = (T *)c.tiles + (long)slot * c.slot_stride;
    const T *gA = (const T *)gpcc_uniform_ptr(tiles + gpcc_tile_off(I, k)), *gB = (const T *)gpcc_uniform_ptr(tiles + gpcc_tile_off(J, k));
    T *Tt = tiles + gpcc_tile_off(I, J);
    if (j == 0) {   // tile (k+1,k+1): lower triangle only (both operands are tile (k+1,k)), then straight on into the diagonal step
        switch (__builtin_amdgcn_readfirstlane(wave)) {
        case 0: gpcc_syrk_lower_wave<T, 7, 0, 5, 0, 0, 0>(gA, P::NCH, smem, Tt, smem_d, wave, lane); break;
        case 1: gpcc_syrk_lower_wave<T, 6, 0, 5, 0, 0, 0>(gA, P::NCH, smem, Tt, smem_d, wave, lane); break;
        case 2: gpcc_syrk_lower_wave<T, 5, 0, 5, 0, 0, 0>(gA, P::NCH, smem, Tt, smem_d, wave, lane); break;
        case 3: gpcc_syrk_lower_wave<T, 4, 0, 5, 0, 0, 0>(gA, P::NCH, smem, Tt, smem_d, wave, lane); break;
        case 4: gpcc_syrk_lower_wave<T, 7, 5, 3, 0, 0, 1>(gA, P::NCH, smem, Tt, smem_d, wave, lane); break;
        case 5: gpcc_syrk_lower_wave<T, 6, 5, 2, 1, 0, 2>(gA, P::NCH, smem, Tt, smem_d, wave, lane); break;
        case 6: gpcc_syrk_lower_wave<T, 5, 5, 1, 2, 0, 3>(gA, P::NCH, smem, Tt, smem_d, wave, lane); break;
        default: gpcc_syrk_lower_wave<T, 3, 0, 4, 0, 0, 0>(gA, P::NCH, smem, Tt, smem_d, wave, lane); break;
        }
        gpcc_diag_body<T, true>(c, g, k + 1, m, smem_d);
        return;
    }
    gpcc_dma_chunk<T>(gA, gB, smem, wave, lane);
    typename P::acc_t acc[2][4];
#pragma unroll
    for (int fm = 0; fm < 2; ++fm)
#pragma unroll
        for (int fn = 0; fn < 4; ++fn)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                acc[fm][fn][r] = -Tt[gpcc_elem_off<T>(wr * 32 + fm * 16 + P::crow(q, r), wc * 64 + fn * 16 + lr)];
    const T *pa0 = smem + (wr * 32 + lr) * P::KC + (((2 * q) ^ sw) * P::EP);
    const T *pa1 = smem + (wr * 32 + lr) * P::KC + (((2 * q + 1) ^ sw) * P::EP);
    const T *pb0 = smem + CH + (wc * 64 + lr) * P::KC + (((2 * q) ^ sw) * P::EP);
    const T *pb1 = smem + CH + (wc * 64 + lr) * P::KC + (((2 * q + 1) ^ sw) * P::EP);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll 2
    for (int ch = 0; ch < P::NCH; ++ch) {   // one tile of K: the chunks of L(I,k) and L(J,k)
        const int st = ch & 1;
        if (ch + 1 < P::NCH)
            gpcc_dma_chunk<T>(gA + (long)(ch + 1) * CH, gB + (long)(ch + 1) * CH, smem + (st ^ 1) * 2 * CH, wave, lane);
        const int so = st * 2 * CH;
        typename P::v16 av[2][2];
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            av[f][0] = *(const typename P::v16 *)(pa0 + so + f * 16 * P::KC);
            av[f][1] = *(const typename P::v16 *)(pa1 + so + f * 16 * P::KC);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            typename P::v16 b[2][2];
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                b[f][0] = *(const typename P::v16 *)(pb0 + so + (2 * h + f) * 16 * P::KC);
                b[f][1] = *(const typename P::v16 *)(pb1 + so + (2 * h + f) * 16 * P::KC);
            }
#pragma unroll
            for (int s = 0; s < P::KSTEPS; ++s)
#pragma unroll
                for (int fm = 0; fm < 2; ++fm)
#pragma unroll
                    for (int f = 0; f < 2; ++f)
                        acc[fm][2 * h + f] = P::mfma(av[fm][s / P::EP][s % P::EP], b[f][s / P::EP][s % P::EP], acc[fm][2 * h + f]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#pragma unroll
    for (int fm = 0; fm < 2; ++fm)
#pragma unroll
        for (int fn = 0; fn < 4; ++fn)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Tt[gpcc_elem_off<T>(wr * 32 + fm * 16 + P::crow(q, r), wc * 64 + fn * 16 + lr)] = -acc[fm][fn][r];
}

// ------------------------------------------------------------------------------------------
// gpcc_syrk_diag (step k; companion of gpcc_update_solve): one workgroup per evaluation -- the lower triangle of the
// diagonal tile, T'(k,k) = T(k,k) - sum_{j<k} L(k,j) L(k,j)^T (36 of 64 blocks, dealt to the waves like
// gpcc_syrk_lower_wave; ONE operand stream: both MFMA operands are fragments of tile row k), through a 4-stage LDS ring
// (one workgroup per CU: nothing else hides the load latency), then straight on into the diagonal step with the tile
// handed over in LDS.  grid cnt, block 512, LDS = the diagonal image.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(512) void gpcc_syrk_diag(GpccCtx c, GpccGroup g, int k)
{
    extern __shared__ __attribute__((aligned(16))) double smem_d[];
    T *smem = (T *)smem_d;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform, and said so)
    const int m = blockIdx.x, slot = g.slot0 + m;
    if (c.info[slot] == 0) {
        const T *tiles = (const T *)c.tiles + (long)slot * c.slot_stride;
        const T *gRow = tiles + gpcc_tile_off(k, 0), *Tt = tiles + gpcc_tile_off(k, k);
        const int nch = GpccPrec<T>::NCH * k;
        switch (__builtin_amdgcn_readfirstlane(wave)) {
        case 0: gpcc_syrk_lower_wave<T, 7, 0, 5, 0, 0, 0>(gRow, nch, smem, Tt, smem_d, wave, lane); break;
        case 1: gpcc_syrk_lower_wave<T, 6, 0, 5, 0, 0, 0>(gRow, nch, smem, Tt, smem_d, wave, lane); break;
        case 2: gpcc_syrk_lower_wave<T, 5, 0, 5, 0, 0, 0>(gRow, nch, smem, Tt, smem_d, wave, lane); break;
        case 3: gpcc_syrk_lower_wave<T, 4, 0, 5, 0, 0, 0>(gRow, nch, smem, Tt, smem_d, wave, lane); break;
        case 4: gpcc_syrk_lower_wave<T, 7, 5, 3, 0, 0, 1>(gRow, nch, smem, Tt, smem_d, wave, lane); break;
        case 5: gpcc_syrk_lower_wave<T, 6, 5, 2, 1, 0, 2>(gRow, nch, smem, Tt, smem_d, wave, lane); break;
        case 6: gpcc_syrk_lower_wave<T, 5, 5, 1, 2, 0, 3>(gRow, nch, smem, Tt, smem_d, wave, lane); break;
        default: gpcc_syrk_lower_wave<T, 3, 0, 4, 0, 0, 0>(gRow, nch, smem, Tt, smem_d, wave, lane); break;
        }
    }
    gpcc_diag_body<T, true>(c, g, k, m, smem_d);   // (a failed evaluation is reported there)
}

// ------------------------------------------------------------------------------------------
// fp32 mode, refinement of the quadratic forms.  An fp32 factor L~ gives G~ = R' (L~ L~')^-1 R with a
// first-order error in E = L~ L~' - K0 that dominates the error of the log-likelihood (it is 10-100x the error of
// logdet: tr(K0^-1 E) averages out, w' E w does not).  For ANY approximate solution X ~ K0^-1 R
//     G = R' K0^-1 R = X'R + R'X - X' K0 X + O(|X - K0^-1 R|^2)
// so one backward solve X = L~^-T W (gpcc_back_solve) and one pass over the fp64 elements of K0, regenerated on the fly
// exactly as the assembly computes them before rounding (gpcc_refine_partials), make G second-order accurate;
// gpcc_refine_finish then repeats the log-likelihood arithmetic with it.  N^2 work against the N^3/3 of the factorisation.
// ------------------------------------------------------------------------------------------
// gpcc_back_solve: one workgroup per evaluation, X = L~^-T W in place of z (nrhs x Np, fp64), three right-hand sides per
// pass.  A transposed matrix-vector product per tile: thread (ls, rg) owns the 16-byte slot ls (4 columns) of rows rg,
// rg+16, ...; eight 16-byte loads per thread in flight (64 KiB per CU: the kernel is one latency-bound stream per
// evaluation), the 16 row groups combined by one shuffle and through LDS.  One pass over the strictly lower tiles and the nt
// inverses inv(L_kk) (linv_keep).  fp32 tiles only.
template <typename T>
__global__ __launch_bounds__(512) void gpcc_back_solve(GpccCtx c, GpccGroup g)
{
    static_assert(sizeof(T) == 4, "the refinement exists for the fp32 mode only");
    constexpr int RB = 3;   // right-hand sides per pass (woodbury with 2 bands: [Q | r] = 3)
    const int m = blockIdx.x, slot = g.slot0 + m, tid = threadIdx.x;
    if (c.info[slot] != 0 || gpcc_leader_failure(c, g) != 0) return;
    const int ls = tid & 31, rg = tid >> 5, wave = tid >> 6, nrhs = c.nrhs;
    const int col0 = (ls >> 3) * 32 + (ls & 7) * 4;   // the 4 logical columns of slot ls
    __shared__ double sx[RB][GPCC_TILE];
    __shared__ double sp[8][RB][GPCC_TILE];
    double *X = c.z + (long)slot * nrhs * c.Np;
    const int p = c.share_p;   // shared prefix: tiles, inverses and W of tile rows < p are the group leader's
    for (int a0 = 0; a0 < nrhs; a0 += RB) {
        const int na = (nrhs - a0 < RB) ? nrhs - a0 : RB;
        for (int k = c.nt - 1; k >= 0; --k) {
            const int kslot = (p && k < p) ? g.slot0 : slot;
            double acc[RB][4];
#pragma unroll
            for (int a = 0; a < RB; ++a)
#pragma unroll
                for (int h = 0; h < 4; ++h) acc[a][h] = 0.0;
            // the tile stream does not depend on x: the loads of the next tile are issued before this one's arithmetic
            auto tile_ptr = [&](int I) -> const T * {
                const int tslot = (p && I < p) ? g.slot0 : slot;   // (I < p implies k < p: the leader's tile)
                const T *tile = (I > k) ? (const T *)c.tiles + (long)tslot * c.slot_stride + gpcc_tile_off(I, k)
                                        : (const T *)c.linv + gpcc_linv_off(c, kslot, k);
                return tile + (ls >> 3) * (GPCC_TILE * 32);
            };
            f4 v[8], vn[8];
            {
                const T *base = tile_ptr(c.nt - 1);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int r = rg + 16 * u;
                    vn[u] = *(const f4 *)(base + r * 32 + (((ls & 7) ^ gpcc_sw(r)) * 4));
                }
            }
            for (int I = c.nt - 1; I >= k; --I) {
                // I > k: tile L(I,k) against x_I;  I == k: inv(L_kk) against b = w_k - (what the tiles below summed to)
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = vn[u];
                if (I > k) {
                    const T *base = tile_ptr(I - 1);
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int r = rg + 16 * u;
                        vn[u] = *(const f4 *)(base + r * 32 + (((ls & 7) ^ gpcc_sw(r)) * 4));
                    }
                }
                __syncthreads();
                if (I > k) {
                    for (int e = tid; e < na * GPCC_TILE; e += 512)
                        sx[e / GPCC_TILE][e % GPCC_TILE] = X[(long)(a0 + e / GPCC_TILE) * c.Np + I * GPCC_TILE + (e % GPCC_TILE)];
                } else {
#pragma unroll
                    for (int a = 0; a < RB; ++a)
#pragma unroll
                        for (int h = 0; h < 4; ++h) {
                            double vv = acc[a][h];
                            vv += __shfl_xor(vv, 32);   // the wave's two row groups
                            if ((tid & 32) == 0) sp[wave][a][col0 + h] = vv;
                            acc[a][h] = 0.0;
                        }
                    __syncthreads();
                    const double *Wk = c.w + (long)kslot * nrhs * c.Np;
                    for (int e = tid; e < na * GPCC_TILE; e += 512) {
                        const int a = e / GPCC_TILE, cc = e % GPCC_TILE;
                        double sum = 0.0;
#pragma unroll
                        for (int w8 = 0; w8 < 8; ++w8) sum += sp[w8][a][cc];
                        sx[a][cc] = Wk[(long)(a0 + a) * c.Np + k * GPCC_TILE + cc] - sum;
                    }
                }
                __syncthreads();
                if (na == RB) {   // the usual case (woodbury with two bands: 3 right-hand sides): straight-line code
#pragma unroll
                    for (int u = 0; u < 8; ++u)
#pragma unroll
                        for (int a = 0; a < RB; ++a) {
                            const double xv = sx[a][rg + 16 * u];
#pragma unroll
                            for (int h = 0; h < 4; ++h) acc[a][h] = fma((double)v[u][h], xv, acc[a][h]);
                        }
                } else {
#pragma unroll
                    for (int u = 0; u < 8; ++u)
#pragma unroll
                        for (int a = 0; a < RB; ++a)
                            if (a < na) {
                                const double xv = sx[a][rg + 16 * u];
#pragma unroll
                                for (int h = 0; h < 4; ++h) acc[a][h] = fma((double)v[u][h], xv, acc[a][h]);
                            }
                }
            }
            __syncthreads();
#pragma unroll
            for (int a = 0; a < RB; ++a)
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    double v = acc[a][h];
                    v += __shfl_xor(v, 32);
                    if ((tid & 32) == 0) sp[wave][a][col0 + h] = v;
                }
            __syncthreads();
            // (a follower of a shared prefix has its own x also on the prefix rows: they depend on its x below)
            for (int e = tid; e < na * GPCC_TILE; e += 512) {
                const int a = e / GPCC_TILE, cc = e % GPCC_TILE;
                double sum = 0.0;
#pragma unroll
                for (int w8 = 0; w8 < 8; ++w8) sum += sp[w8][a][cc];
                X[(long)(a0 + a) * c.Np + k * GPCC_TILE + cc] = sum;
            }
        }
        __syncthreads();
    }
}

// gpcc_refine_partials: per lower tile (I,J) the contribution to X' K0 X, K0 = delayedCovariance + Sobs in fp64 exactly
// as gpcc_assemble_tiles forms it before rounding.  Thread = one row of the tile and one half of its columns; only the
// row sums s_a[i] = sum_j K_ij x_a[j] are needed:  off-diagonal tiles stand for (I,J) and (J,I):
//     sum_ij K_ij (x_a[i] x_b[j] + x_a[j] x_b[i]) = sum_i (x_a[i] s_b[i] + x_b[i] s_a[i]).
// Deterministic: fixed reduction tree, one partial per tile (no atomics).   grid (nt*nt, cnt), block 256.
// NR: the number of right-hand sides at compile time (1: plain; 3, 4: woodbury with 2 / 3 bands), 0 = c.nrhs at run time --
// with it the element loop is straight-line code (nine uniform branches per element otherwise).
template <int KID, int NR>
__global__ __launch_bounds__(256) void gpcc_refine_partials(GpccCtx c, GpccGroup g)
{
    const int I = blockIdx.x / c.nt, J = blockIdx.x % c.nt;
    if (J > I) return;
    const int m = blockIdx.y, slot = g.slot0 + m, tid = threadIdx.x, nrhs = NR ? NR : c.nrhs;
    constexpr int NA = NR ? NR : GPCC_MAXRHS;
    if (c.info[slot] != 0 || gpcc_leader_failure(c, g) != 0) return;
    const double *delays = g.delays + (long)(g.first + m) * c.L;
    const double *alpha = g.alpha + (long)(g.first + m) * c.L;
    const GpccKernelConst kc = gpcc_kernel_const<KID>(g.rho[g.first + m]);
    __shared__ double su[2][GPCC_TILE], sa[2][GPCC_TILE], ssig[GPCC_TILE], sx[2][GPCC_MAXRHS][GPCC_TILE];
    __shared__ int sb[2][GPCC_TILE];
    __shared__ double sred[4][GPCC_MAXRHS * GPCC_MAXRHS];
    __shared__ double sexp[64];   // the table exp (diagonal / band-straddling tiles)
    __shared__ double sA[2][GPCC_TILE], sB[2][GPCC_TILE];   // separable factors of the tile's points (gpcc_sep_point)
    GPCC_EXP_TABLE_TO_LDS(sexp, tid);
    bool sep_ok = true;
    {
        const int side = tid >> 7, r = tid & 127;
        const int T0 = side ? J : I, gi = T0 * GPCC_TILE + r;
        const int b = c.band[gi];
        sb[side][r] = b;
        const double u_ = (b >= 0) ? c.t[gi] - delays[b] : 0.0, a_ = (b >= 0) ? alpha[b] : 0.0;
        su[side][r] = u_;
        sa[side][r] = a_;
        if (KID != 1) {   // (padding: amplitude 0, placed at the centre)
            double A_, B_;
            sep_ok = gpcc_sep_point((b >= 0) ? u_ : c.tmid, c.tmid, gpcc_kernel_scale<KID>(kc), a_, A_, B_);
            sA[side][r] = A_;
            sB[side][r] = B_;
        }
        if (side == 0) ssig[r] = c.sig2[gi];
        const double *X = c.z + (long)slot * nrhs * c.Np;
        for (int a = 0; a < nrhs; ++a) sx[side][a][r] = X[(long)a * c.Np + gi];
    }
    const bool sep = __syncthreads_and(sep_ok ? 1 : 0) != 0 && KID != 1;
    const int i = tid & 127, half = tid >> 7;
    const bool diag = (I == J);
    const int br = sb[0][i];
    const double ur = su[0][i], ar = sa[0][i], sg = ssig[i];
    double s[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) s[a] = 0.0;
    // tiles inside one band pair and off the diagonal (most of them: points are ordered by band): no selects
    const bool plain = !diag && sb[0][0] >= 0 && sb[1][0] >= 0 && sb[0][0] == sb[0][GPCC_TILE - 1] && sb[1][0] == sb[1][GPCC_TILE - 1];
    if (plain) {
        // one band per side: the amplitude product is one number per row (applied to the row sums) and the kernel's constants
        // fold into ONE scale of the distance, t = |u_i - u_j| * kscale (the difference first: it is exact for nearby points) --
        // 18 instead of 22 double-rate operations per element; the element agrees with the assembly's to an ulp or two
        const double kscale = (KID == 0) ? kc.c1 : (KID == 1) ? 0.5 * kc.c1 : (KID == 2) ? 1.7320508075688772 * kc.c1 : 2.23606797749979 * kc.c1;
        if (KID != 1 && sep) {   // the separable form (as the assembly): no exponential per element, the amplitudes inside the factors
            // Register-blocked (round 4): a thread owns FOUR rows (i0, i0 + 32, ...) and 16 columns, so the 3 + nrhs LDS reads of a column
            // serve four elements -- with one row per thread the loop was bound by the issue of its LDS reads (6 per element and wave
            // at nrhs = 3, 4 cycles each on the CU's one LDS, against 40 cycles of arithmetic on the wave's own SIMD)
            const int i0 = tid & 31, cg = tid >> 5;
            double ur4[4], Ar4[4], Br4[4], s4[4][NA];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ur4[r] = su[0][i0 + 32 * r];
                Ar4[r] = sA[0][i0 + 32 * r];
                Br4[r] = sB[0][i0 + 32 * r];
#pragma unroll
                for (int a = 0; a < NA; ++a) s4[r][a] = 0.0;
            }
#pragma unroll 2
            for (int jj = 0; jj < 16; ++jj) {
                const int j = cg * 16 + jj;
                const double uj = su[1][j], Aj = sA[1][j], Bj = sB[1][j];
                double xj[NA];
#pragma unroll
                for (int a = 0; a < NA; ++a) xj[a] = (NR || a < nrhs) ? sx[1][a][j] : 0.0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double kv = gpcc_sep_eval<KID>(ur4[r], uj, Ar4[r], Br4[r], Aj, Bj, kscale);
#pragma unroll
                    for (int a = 0; a < NA; ++a) s4[r][a] = fma(kv, xj[a], s4[r][a]);
                }
            }
            // this thread's contribution to every (a,b) over its four rows (off-diagonal tile: both (I,J) and (J,I)), then the fixed tree
            const int lane_ = tid & 63, wave_ = tid >> 6;
            for (int a = 0; a < nrhs; ++a)
                for (int b = a; b < nrhs; ++b) {
                    double v = 0.0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        double sa_ = 0.0, sb_ = 0.0;
#pragma unroll
                        for (int e = 0; e < NA; ++e) {   // (s4 stays in registers: no dynamic index)
                            sa_ = (e == a) ? s4[r][e] : sa_;
                            sb_ = (e == b) ? s4[r][e] : sb_;
                        }
                        v = fma(sx[0][a][i0 + 32 * r], sb_, v);
                        v = fma(sx[0][b][i0 + 32 * r], sa_, v);
                    }
#pragma unroll
                    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
                    if (lane_ == 0) sred[wave_][a * nrhs + b] = sred[wave_][b * nrhs + a] = v;
                }
            __syncthreads();
            if (tid < nrhs * nrhs) {
                const long tidx = (long)I * (I + 1) / 2 + J, ntri = (long)c.nt * (c.nt + 1) / 2;
                c.gpart[((long)slot * ntri + tidx) * (GPCC_MAXRHS * GPCC_MAXRHS) + tid] = ((sred[0][tid] + sred[1][tid]) + sred[2][tid]) + sred[3][tid];
            }
            return;
        } else {
#pragma unroll 4
        for (int jj = 0; jj < 64; ++jj) {
            const int j = half * 64 + jj;
            const double d = ur - su[1][j];
            const double t = (KID == 1) ? (d * d) * kscale : fabs(d) * kscale;
#ifdef GPCC_AB_POLY_EXP   /* A/B builds only (tools/ab_exp.sh) */
            double kv = gpcc_exp_nonpos(-t);
#else
            double kv = gpcc_exp_nonpos_tab(-t, sexp);
#endif
            if (KID == 2) kv *= 1.0 + t;
            else if (KID == 3) kv *= fma(t, fma(t, 1.0 / 3.0, 1.0), 1.0);
#pragma unroll
            for (int a = 0; a < NA; ++a)
                if (NR || a < nrhs) s[a] = fma(kv, sx[1][a][j], s[a]);
        }
        const double amp = ar * sa[1][0];
#pragma unroll
        for (int a = 0; a < NA; ++a) s[a] *= amp;
        }
    } else {
        const double Ar_g = (KID != 1) ? sA[0][i] : 0.0, Br_g = (KID != 1) ? sB[0][i] : 0.0;
        const double kscale_g = gpcc_kernel_scale<KID>(kc);
#pragma unroll 4
        for (int jj = 0; jj < 64; ++jj) {
            const int j = half * 64 + jj;
            const int bc = sb[1][j];
            double val;
            if (KID != 1 && sep) val = gpcc_sep_eval<KID>(ur, su[1][j], Ar_g, Br_g, sA[1][j], sB[1][j], kscale_g);   // (as the assembly)
            else val = (ar * sa[1][j]) * gpcc_kernel_eval<KID>(ur, su[1][j], kc, sexp);
            if (diag && i == j) val = val + sg;
            if (br < 0 || bc < 0) val = 0.0;   // padding / explicit rows: X is zero there
#pragma unroll
            for (int a = 0; a < NA; ++a)
                if (NR || a < nrhs) s[a] = fma(val, sx[1][a][j], s[a]);
        }
    }
    // per-thread contribution to every (a,b), reduced over the 256 threads in a fixed order
    const int lane = tid & 63, wave = tid >> 6;
    for (int a = 0; a < nrhs; ++a)
        for (int b = a; b < nrhs; ++b) {   // X' K0 X is symmetric: the upper triangle, mirrored
            double v = sx[0][a][i] * s[b];
            if (!diag) v += sx[0][b][i] * s[a];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
            if (lane == 0) sred[wave][a * nrhs + b] = sred[wave][b * nrhs + a] = v;
        }
    __syncthreads();
    if (tid < nrhs * nrhs) {
        const long tidx = (long)I * (I + 1) / 2 + J, ntri = (long)c.nt * (c.nt + 1) / 2;
        c.gpart[((long)slot * ntri + tidx) * (GPCC_MAXRHS * GPCC_MAXRHS) + tid] = ((sred[0][tid] + sred[1][tid]) + sred[2][tid]) + sred[3][tid];
    }
}

// gpcc_refine_finish: G = X'R + R'X - X' K0 X  (R = [Q | Y - bbar] in woodbury mode, else Y - bbar), then the same
// log-likelihood arithmetic as the last diagonal step, whose outputs it replaces.   grid cnt, block 256.
static __global__ __launch_bounds__(256) void gpcc_refine_finish(GpccCtx c, GpccGroup g)
{
    const int m = blockIdx.x, slot = g.slot0 + m, tid = threadIdx.x, nrhs = c.nrhs, n2 = nrhs * nrhs;
    if (c.info[slot] != 0 || gpcc_leader_failure(c, g) != 0) return;   // the diagonal kernel has reported the failure
    __shared__ double sacc[256];
    __shared__ double sG[GPCC_MAXRHS * GPCC_MAXRHS], sXR[GPCC_MAXRHS * GPCC_MAXRHS];
    const long ntri = (long)c.nt * (c.nt + 1) / 2;
    const double *Xown = c.z + (long)slot * nrhs * c.Np;
    for (int e = 0; e < n2; ++e) {
        const int a = e / nrhs, b = e % nrhs;
        double v = 0.0;   // X' K0 X: the per-tile partials, fixed order
        for (long t = tid; t < ntri; t += 256) v += c.gpart[((long)slot * ntri + t) * (GPCC_MAXRHS * GPCC_MAXRHS) + e];
        sacc[tid] = v;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if (tid < st) sacc[tid] += sacc[tid + st];
            __syncthreads();
        }
        if (tid == 0) sG[e] = sacc[0];
        __syncthreads();
        // X'R: column b of R is the indicator of band b (woodbury, b < nrhs - 1) or the residual Y - bbar
        v = 0.0;
        for (int i = tid; i < c.Np; i += 256) {
            const double xa = Xown[(long)a * c.Np + i];
            const double rb = (b < nrhs - 1) ? ((c.band[i] == b) ? 1.0 : 0.0) : c.resid[i];
            v = fma(xa, rb, v);
        }
        sacc[tid] = v;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if (tid < st) sacc[tid] += sacc[tid + st];
            __syncthreads();
        }
        if (tid == 0) sXR[e] = sacc[0];
        __syncthreads();
    }
    if (tid == 0) {
        double *G = sacc;   // (the reduction buffer is free by now; a local array indexed at run time lived in 656 B of private memory)
        for (int a = 0; a < nrhs; ++a)
            for (int b = 0; b < nrhs; ++b) G[a * nrhs + b] = (sXR[a * nrhs + b] + sXR[b * nrhs + a]) - 0.5 * (sG[a * nrhs + b] + sG[b * nrhs + a]);
        int bad = 0;
        const double llv = gpcc_loglik_from_gram(c, G, nrhs, c.logdet[slot], bad);
        g.out_loglik[g.first + m] = bad ? __builtin_nan("") : llv;
        g.out_info[g.first + m] = bad;
    }
}

// ------------------------------------------------------------------------------------------
// dense exports (tests, prediction): a square block of one slot's tiles -> column-major n x n
// ------------------------------------------------------------------------------------------
template <typename T>
static __global__ void gpcc_export_dense(GpccCtx c, int slot, double *out, int symmetric, int off, int n, double jitter)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)n * n) return;
    const int r0 = (int)(idx % n), c0 = (int)(idx / n);
    const int r = off + r0, col = off + c0;
    const T *tiles = (const T *)c.tiles + (long)slot * c.slot_stride;
    double v;
    if (r >= col)
        v = (double)tiles[gpcc_tile_off(r >> 7, col >> 7) + gpcc_elem_off<T>(r & 127, col & 127)];
    else
        v = symmetric ? (double)tiles[gpcc_tile_off(col >> 7, r >> 7) + gpcc_elem_off<T>(col & 127, r & 127)] : 0.0;
    if (r0 == c0) v += jitter;
    out[idx] = v;
}

// ------------------------------------------------------------------------------------------
// gpcc_load_dense: a caller-supplied symmetric matrix (column-major n x n) into the tile layout of a
// slot, z <- resid: logpdf(MvNormal(mu, Sigma), x) for an explicit Sigma (the test log-likelihood of
// predictTest, marginaliseb.jl:311-343) then runs on the same factorisation kernels.
// grid nt*nt, block 256.
// ------------------------------------------------------------------------------------------
template <typename T>
static __global__ void gpcc_load_dense(GpccCtx c, int slot, const double *dense, int n, const double *resid)
{
    const int I = blockIdx.x / c.nt, J = blockIdx.x % c.nt;
    if (J > I) return;
    const int tid = threadIdx.x;
    if (I == 0 && tid == 0) {
        c.info[slot] = 0;
        c.logdet[slot] = 0.0;
        c.gram[(long)slot * GPCC_MAXRHS * GPCC_MAXRHS] = 0.0;
    }
    T *Tt = (T *)c.tiles + (long)slot * c.slot_stride + gpcc_tile_off(I, J);
    for (int e = tid; e < GPCC_TILE_ELEMS; e += 256) {
        const int r = e & 127, col = e >> 7;   // consecutive threads walk down a column of the dense input
        const long gr = (long)I * GPCC_TILE + r, gc = (long)J * GPCC_TILE + col;
        Tt[gpcc_elem_off<T>(r, col)] = (T)((gr < n && gc < n) ? dense[gc * n + gr] : ((gr == gc) ? 1.0 : 0.0));
    }
    if (I == J && tid < GPCC_TILE) {
        const long gr = (long)I * GPCC_TILE + tid;
        c.z[(long)slot * c.Np + gr] = (gr < n) ? resid[gr] : 0.0;
    }
}

// ------------------------------------------------------------------------------------------
// delayedCovariance(kernel, scale, delays, rho, x, y), rectangular, column-major output
// (src/delayedCovariance.jl:1-35).  xu/yu arrive already shifted (x - delays[band]).
// ------------------------------------------------------------------------------------------
template <int KID>
static __global__ void gpcc_covariance_kernel(long nx, long ny, const double *xu, const double *xs, const double *yu,
                                       const double *ys, double rho, double *out)
{
    __shared__ double sexp[64];   // the table exp (2^(j/64) table + degree-5 polynomial; the refinement pass uses it too, the assembly kept
                                  // the degree-13 polynomial): this entry is how tests reach it
    GPCC_EXP_TABLE_TO_LDS(sexp, threadIdx.x);
    __syncthreads();
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nx * ny) return;
    const long r = idx % nx, col = idx / nx;
    out[idx] = (xs[r] * ys[col]) * gpcc_kernel_eval<KID>(xu[r], yu[col], gpcc_kernel_const<KID>(rho), sexp);
}

// ------------------------------------------------------------------------------------------
// getprobabilities (src/getprobabilities.jl:10-20): exp(joint - logsumexp(joint)), one block.
// ------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(1024) void gpcc_probabilities_kernel(int G, const double *ll, const double *lp,
                                                                  double *out)
{
    __shared__ double red[1024];
    const int tid = threadIdx.x;
    double mx = -INFINITY;
    for (int i = tid; i < G; i += 1024) {
        const double j = ll[i] + (lp ? lp[i] : 1.0);  // getprobabilities.jl:3: log-prior of ones
        mx = fmax(mx, j);
    }
    red[tid] = mx;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (tid < s) red[tid] = fmax(red[tid], red[tid + s]);
        __syncthreads();
    }
    mx = red[0];
    __syncthreads();
    double sum = 0.0;
    for (int i = tid; i < G; i += 1024) sum += exp(ll[i] + (lp ? lp[i] : 1.0) - mx);
    red[tid] = sum;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    const double lse = mx + log(red[0]);
    for (int i = tid; i < G; i += 1024) out[i] = exp(ll[i] + (lp ? lp[i] : 1.0) - lse);
}

// ------------------------------------------------------------------------------------------
// self-test: f64 MFMA fragment maps with asymmetric integer data, and a rate probe.
// ------------------------------------------------------------------------------------------
static __global__ void gpcc_selftest_map(const double *A /*16x4 row-major*/, const double *B /*4x16 row-major*/,
                                  double *D /*16x16 row-major*/)
{
    const int lane = threadIdx.x & 63, lr = lane & 15, q = lane >> 4;
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[lr * 4 + q], B[q * 16 + lr], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(q + 4 * r) * 16 + lr] = acc[r];
}

static __global__ __launch_bounds__(256) void gpcc_selftest_rate(double *sink, int iters)
{
    d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    const double x = 1.0 + threadIdx.x * 1e-9, y = 1.0 - threadIdx.x * 1e-9;
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
    }
    d4 s = a0 + a1 + a2 + a3;
    if (s[0] + s[1] + s[2] + s[3] == 12345.678) sink[0] = s[0];
}

static __global__ void gpcc_selftest_map_f32(const float *A /*16x4 row-major*/, const float *B /*4x16 row-major*/,
                                      float *D /*16x16 row-major*/)
{
    const int lane = threadIdx.x & 63, lr = lane & 15, q = lane >> 4;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = GpccPrec<float>::mfma(A[lr * 4 + q], B[q * 16 + lr], acc);
    for (int r = 0; r < 4; ++r) D[GpccPrec<float>::crow(q, r) * 16 + lr] = acc[r];
}
