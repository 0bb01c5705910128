"""Round 4: the diagonal step on half a CU's LDS (gpcc_diag_blocks: packed 16 x 16 block image, inverse in place) and the
one-launch step with look-ahead (gpcc_step: the workgroup that owns tile (k+1,k) goes on into the diagonal step of column k+1
while the rest of the launch updates column k).  Both keep the arithmetic and its order of the kernels they replace
(gpcc_diag_body; gpcc_syrk_diag + gpcc_update_solve), so every comparison here is BITWISE; the oracle comparisons of
tests/test_gpu_parity.py cover the kernels they are compared with.  (Measured: neither is faster than what it replaces -- the
fp64 matrix pipe, not LDS occupancy, bounds the step -- so both are OPTIONS, off by default: DESIGN.md 4.2e.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LL_RTOL = 1e-8
FP32_RTOL = 1e-3


@pytest.fixture(scope="module")
def gp():
    import torch
    torch.cuda.init()
    import gpcc_amd
    return gpcc_amd


def _rel(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b)) / np.abs(np.asarray(b)))


def _batch(Nl, y, M, seed, spread=12.0):
    from gpcc_amd import synthetic
    alpha, rho = synthetic.default_hyperparameters(y)
    L = len(Nl)
    rng = np.random.default_rng(seed)
    delays = np.concatenate([np.zeros((M, 1)), rng.random((M, L - 1)) * spread], 1)
    alphas = np.tile(alpha, (M, 1)) * (0.5 + rng.random((M, L)))
    rhos = rho * (0.5 + rng.random(M))
    return delays, alphas, rhos


@pytest.mark.parametrize("Nl,prec,mb", [([300, 280], "fp64", True), ([129], "fp64", True), ([520, 500, 490], "fp64", True),
                                        ([1100, 1000], "fp64", False), ([384, 300], "fp32", True), ([450, 400, 300], "fp32", True),
                                        ([700, 600], "fp32", False)])
def test_one_launch_step_returns_the_bits_of_the_two_launch_step(gp, oracle, Nl, prec, mb):
    """gpcc_step (option step_fused = 1) against gpcc_syrk_diag + gpcc_update_solve (step_fused = 0): groups of 3
    (spread job map), 41 and 130 evaluations (two groups of 64 + a remainder of 2), a ragged last tile, an argument error in
    the batch; fp32 handles with two and three bands (3 and 4 right-hand sides through the block image) with refinement and
    guard; then against the oracle."""
    from gpcc_amd import synthetic
    t, y, s, _ = synthetic.simulate_lightcurves(Nl, seed=21)
    for M in (3, 41, 130):
        delays, alphas, rhos = _batch(Nl, y, M, M)
        alphas[1, 0] = 0.0                       # argument error: info -1
        with gp.Objective(t, y, s, gp.matern32, marginalise_b=mb, precision=prec, slots_per_stream=64) as obj:
            for k, v in (("right_looking_max", 0), ("shared_prefix", 0), ("fused_solve_min", 1), ("split_min", 0), ("step_fused", 1)):
                obj.set_option(k, v)
            assert obj.get_option("step_fused") == 1
            a, ia = obj.loglik_batch(delays, alphas, rhos)
            a2, ia2 = obj.loglik_batch(delays, alphas, rhos)
            ca = obj.conditioning(M) if prec == "fp32" else None
            obj.set_option("step_fused", 0)
            b, ib = obj.loglik_batch(delays, alphas, rhos)
            cb = obj.conditioning(M) if prec == "fp32" else None
        assert np.array_equal(a, a2, equal_nan=True) and np.array_equal(ia, ia2)      # repeatable
        assert np.array_equal(ia, ib) and ia[1] == -1 and (np.delete(ia, 1) == 0).all()
        assert np.array_equal(a, b, equal_nan=True)                                    # BITWISE
        if ca is not None:
            assert np.array_equal(ca, cb, equal_nan=True)                              # the guard's pivot-ratio statistics too
        if M == 41 and sum(Nl) <= 1600:
            ok = ia == 0
            ref, rinfo = oracle.loglik_batch("matern32", t, y, s, delays, alphas, rhos, mb, nthreads=8)
            assert np.array_equal(rinfo == 0, ok)
            assert _rel(a[ok], ref[ok]) <= (LL_RTOL if prec == "fp64" else FP32_RTOL)


def test_a_failing_pivot_inside_the_look_ahead_job(gp):
    """A non-positive pivot met by the look-ahead job's diagonal step (noise-free rbf with a huge length scale: numerically singular)
    must stop that evaluation's later jobs and be reported exactly as by the two-launch path; the other evaluations of the
    group are untouched."""
    rng = np.random.default_rng(4)
    Nl = [420, 380]
    t = [np.sort(rng.random(n) * n / 3.0) for n in Nl]
    y = [rng.standard_normal(n) for n in Nl]
    s = [np.zeros(n) for n in Nl]
    M = 19
    delays = np.stack([np.zeros(M), rng.random(M) * 5], 1)
    alphas = 0.5 + rng.random((M, 2))
    rhos = 0.01 * (0.5 + rng.random(M))
    rhos[5] = 1e5
    res = []
    for sf in (1, 0):
        with gp.Objective(t, y, s, gp.rbf, marginalise_b=False, slots_per_stream=32) as obj:
            for k, v in (("right_looking_max", 0), ("shared_prefix", 0), ("fused_solve_min", 1), ("split_min", 0), ("step_fused", sf)):
                obj.set_option(k, v)
            res.append(obj.loglik_batch(delays, alphas, rhos))
    (a, ia), (b, ib) = res
    assert ia[5] > 0 and np.isnan(a[5]) and (np.delete(ia, 5) == 0).all()
    assert np.array_equal(ia, ib) and np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("Nl,prec,M", [([300, 280], "fp64", 20), ([129], "fp64", 5), ([520, 500, 490], "fp64", 41), ([384, 300], "fp32", 20),
                                       ([1024, 1030], "fp64", 3)])
def test_block_image_diagonal_step_returns_the_bits_of_the_square_image(gp, Nl, prec, M):
    """gpcc_diag_factor2 (option diag_blocks = 1: 80 KiB of LDS) against gpcc_diag_factor (158.7 KiB) on the
    three-kernel path -- left-looking, with the right-looking tail, and (M = 3) the right-looking small-group path whose
    gpcc_small_step carries the diagonal step inside the update launch."""
    from gpcc_amd import synthetic
    t, y, s, _ = synthetic.simulate_lightcurves(Nl, seed=22)
    delays, alphas, rhos = _batch(Nl, y, M, 100 + M)
    alphas[M - 1, 0] = -2.0
    out = []
    for db in (1, 0):
        with gp.Objective(t, y, s, gp.matern52, precision=prec, slots_per_stream=64) as obj:
            for k, v in (("shared_prefix", 0), ("fused_solve_min", 10 ** 6), ("split_min", 0), ("diag_blocks", db)):
                obj.set_option(k, v)
            out.append(obj.loglik_batch(delays, alphas, rhos))
            obj.set_option("hybrid_tail", 0)
            out.append(obj.loglik_batch(delays, alphas, rhos))
            obj.set_option("right_looking_max", 0)
            out.append(obj.loglik_batch(delays, alphas, rhos))
    for i in range(3):
        (a, ia), (b, ib) = out[i], out[3 + i]
        assert ia[M - 1] == -1 and (ia[:M - 1] == 0).all()
        assert np.array_equal(ia, ib) and np.array_equal(a, b, equal_nan=True), i


def test_dense_factor_and_model_matrix_through_the_block_image(gp):
    """gpcc_factor_dense (store_l: L_kk leaves the block image before inv(L_kk) overwrites it) and the augmented systems
    (predictTest / postb run the three-kernel path on a one-slot workspace) against the square-image kernels, bitwise, and the
    factor against LAPACK's potrf of the exported model matrix."""
    from gpcc_amd import synthetic
    Nl = [300, 270]
    t, y, s, _ = synthetic.simulate_lightcurves(Nl, seed=23)
    alpha, rho = synthetic.default_hyperparameters(y)
    d = [0.0, 1.3]
    got = []
    for db in (1, 0):
        with gp.Objective(t, y, s, gp.matern32) as obj:
            obj.set_option("diag_blocks", db)
            Lf, info = obj.factor(d, alpha, rho)
            assert info == 0
            mu, Sig = obj.predict(d, alpha, rho, [np.linspace(0, 90, 40), np.linspace(1, 80, 33)])
            pm, pS = obj.posterior_offsets(d, alpha, rho)
            K = obj.model_matrix(d, alpha, rho)
            got.append((Lf, mu, Sig, pm, pS))
    for a, b in zip(got[0], got[1]):
        assert np.array_equal(a, b)
    Lf = got[0][0]
    Lref = np.linalg.cholesky(K)
    assert np.max(np.abs(Lf - Lref)) / np.max(np.abs(Lref)) <= 1e-9
    assert np.array_equal(np.triu(Lf, 1), np.zeros_like(K))


@pytest.mark.parametrize("Nl,prec,mb", [([700, 600], "fp64", True), ([1100, 1000], "fp64", False), ([450, 400, 300], "fp32", True)])
def test_right_looking_look_ahead_returns_the_same_bits(gp, oracle, Nl, prec, mb):
    """Option look_ahead (three-kernel path, right-looking steps: column k+1 of the trailing update first, its diagonal step and panel
    solve on a helper stream beside the rest): groups that are right-looking throughout, left-looking with a right-looking tail, split
    into two halves on two streams (each with its own helper) -- bitwise the results of look_ahead = 0, an argument error included."""
    from gpcc_amd import synthetic
    t, y, s, _ = synthetic.simulate_lightcurves(Nl, seed=22)
    for M, opts in ((20, {"right_looking_max": 64, "fused_small_max": 0, "split_min": 0, "split_small": 0}),   # right-looking from step 0
                    (7, {"right_looking_max": 64, "fused_small_max": 0, "split_min": 0, "split_small": 0}),    # ... with the spread job map
                    (16, {"split_min": 0, "split_small": 0}),                                                 # left-looking + tail
                    (40, {}),                                                                                  # default dispatch (split halves)
                    (100, {"hybrid_occ": 4096, "hybrid_mall_mb": 4096})):                                      # a long tail
        delays, alphas, rhos = _batch(Nl, y, M, M)
        alphas[1, 0] = 0.0
        res = {}
        with gp.Objective(t, y, s, gp.matern32, marginalise_b=mb, precision=prec) as obj:
            obj.set_option("shared_prefix", 0)
            for k, v in opts.items():
                obj.set_option(k, v)
            assert obj.get_option("look_ahead") == 0     # an option, off by default (measured: DESIGN.md 4.2e)
            for la in (1, 0, 1):
                obj.set_option("look_ahead", la)
                ll, info = obj.loglik_batch(delays, alphas, rhos)
                res.setdefault(la, []).append((ll, info, obj.conditioning(M) if prec == "fp32" else None))
        (a, ia, ca), (a2, ia2, _) = res[1]
        (b, ib, cb), = res[0]
        assert np.array_equal(a, a2, equal_nan=True) and np.array_equal(ia, ia2)
        assert np.array_equal(ia, ib) and ia[1] == -1 and (np.delete(ia, 1) == 0).all()
        assert np.array_equal(a, b, equal_nan=True)                                    # BITWISE
        if ca is not None:
            assert np.array_equal(ca, cb, equal_nan=True)
        if M == 16:
            ok = ia == 0
            ref, rinfo = oracle.loglik_batch("matern32", t, y, s, delays, alphas, rhos, mb, nthreads=8)
            assert _rel(a[ok], ref[ok]) <= (LL_RTOL if prec == "fp64" else FP32_RTOL)


def test_a_failing_pivot_in_the_look_ahead_chain(gp):
    """A non-positive pivot in a diagonal step that runs on the helper stream: the evaluation reports it (info > 0), its later jobs
    return at once, the other evaluations of the group are unaffected; the same info with look_ahead = 0."""
    from gpcc_amd import synthetic
    Nl = [500, 400]
    t, y, s, _ = synthetic.simulate_lightcurves(Nl, seed=4)
    s0 = [np.full_like(x, 1e-9) for x in s]
    M = 16
    delays, alphas, rhos = _batch(Nl, y, M, 2)
    rhos[3] = 1e6                                  # rbf, no noise to speak of, enormous length scale: numerically singular
    out = []
    with gp.Objective(t, y, s0, gp.rbf, marginalise_b=False) as obj:
        for k, v in (("right_looking_max", 64), ("fused_small_max", 0), ("split_min", 0), ("split_small", 0), ("shared_prefix", 0)):
            obj.set_option(k, v)
        for la in (1, 0):
            obj.set_option("look_ahead", la)
            out.append(obj.loglik_batch(delays, alphas, rhos))
    (a, ia), (b, ib) = out
    assert ia[3] > 0 and np.array_equal(ia, ib)
    assert np.array_equal(a, b, equal_nan=True)
