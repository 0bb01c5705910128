"""Round 4: the assembly folded into the update (option fold_assembly, GpccCtx::fold).  On the fused left-looking path every
off-diagonal tile (I,k) is read exactly once, by the job that updates and solves it; with the fold that job evaluates the tile's
elements itself -- from the separable factors gpcc_sep_points computes once per evaluation -- instead of reading what
gpcc_assemble_tiles wrote.  The element is formed by the SAME expression in both places (gpcc_sep_eval + the B term), so the
log-likelihoods must agree BITWISE with fold_assembly = 0; tiles the flags exclude (a tile row that straddles two bands or holds
padding, a point outside the separable range) are still assembled and read.  Oracle comparisons ride along."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LL_RTOL = 1e-8


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


FUSED = (("right_looking_max", 0), ("shared_prefix", 0), ("fused_solve_min", 1), ("split_min", 0))
# the three-kernel path (fold = 2: the first gpcc_panel_update job that touches a tile evaluates it): left-looking with a right-looking
# tail, right-looking from step 0, the default dispatch of 40 (two halves on two streams)
THREE = {"tail": (16, (("shared_prefix", 0), ("split_min", 0), ("split_small", 0))),
         "left": (16, (("shared_prefix", 0), ("split_min", 0), ("split_small", 0), ("hybrid_tail", 0))),
         "right": (20, (("shared_prefix", 0), ("right_looking_max", 64), ("fused_small_max", 0), ("chain_max", 0), ("split_min", 0), ("split_small", 0))),
         "spread": (5, (("shared_prefix", 0), ("fused_small_max", 0), ("chain_max", 0), ("split_min", 0), ("split_small", 0))),
         "default40": (40, (("shared_prefix", 0),))}


@pytest.mark.parametrize("path", sorted(THREE))
@pytest.mark.parametrize("kname,prec,asm32", [("matern32", "fp64", 1), ("OU", "fp64", 1), ("matern52", "fp32", 1), ("rbf", "fp32", 1),
                                               ("matern32", "fp32", 0)])
def test_three_kernel_path_folded_tiles_return_the_bits_of_assembled_tiles(gp, oracle, path, kname, prec, asm32):
    from gpcc_amd import synthetic
    Nl = [700, 600]
    t, y, s, _ = synthetic.simulate_lightcurves(Nl, seed=35)
    M, opts = THREE[path]
    delays, alphas, rhos = _batch(Nl, y, M, 7)
    alphas[1, 0] = 0.0
    span = max(tt.max() for tt in t) - min(tt.min() for tt in t)
    rhos[2] = span / 5000.0
    out = {}
    with gp.Objective(t, y, s, getattr(gp, kname), precision=prec) as obj:
        for k, v in opts:
            obj.set_option(k, v)
        if prec == "fp32":
            obj.set_option("fp32_assemble", asm32)
        for fold in (1, 0):
            obj.set_option("fold_assembly", fold)
            ll, info = obj.loglik_batch(delays, alphas, rhos)
            out[fold] = (ll, info, obj.conditioning(M) if prec == "fp32" else None)
    a, ia, ca = out[1]
    b, ib, cb = out[0]
    assert ia[1] == -1 and np.array_equal(ia, ib)
    assert np.array_equal(a, b, equal_nan=True)                                        # BITWISE
    if ca is not None:
        assert np.array_equal(ca, cb, equal_nan=True)
    ok = ia == 0
    ref, rinfo = oracle.loglik_batch(kname, t, y, s, delays, alphas, rhos, True, nthreads=8)
    assert _rel(a[ok], ref[ok]) <= (LL_RTOL if prec == "fp64" else 1e-3)


@pytest.mark.parametrize("kname", ["OU", "matern32", "matern52", "rbf"])
@pytest.mark.parametrize("Nl,mb", [([700, 600], True), ([700, 600], False), ([450, 400, 300], True), ([1100, 1000], True),
                                   ([1024], True)])
def test_folded_tiles_return_the_bits_of_assembled_tiles(gp, oracle, kname, Nl, mb):
    """Same-band off-diagonal tiles with and without the B term, cross-band tiles, tile rows that straddle a band boundary, a ragged
    last tile (padding), one band whose size is a multiple of the tile (no padding at all); groups of 64 + a remainder; an argument
    error and an evaluation whose length scale puts its points OUTSIDE the separable range (flags 0: assembled and read)."""
    from gpcc_amd import synthetic
    t, y, s, _ = synthetic.simulate_lightcurves(Nl, seed=33)
    M = 70
    delays, alphas, rhos = _batch(Nl, y, M, 5)
    alphas[1, 0] = 0.0                       # argument error: info -1
    span = max(tt.max() for tt in t) - min(tt.min() for tt in t)
    rhos[2] = span / 5000.0                  # s (u - c) far beyond GPCC_SEP_MAX = 600 at the ends of the time axis
    out = {}
    with gp.Objective(t, y, s, getattr(gp, kname), marginalise_b=mb, slots_per_stream=64) as obj:
        for k, v in FUSED:
            obj.set_option(k, v)
        assert obj.get_option("fold_assembly") == 1          # the default
        for fold in (1, 0, 1):
            obj.set_option("fold_assembly", fold)
            ll, info = obj.loglik_batch(delays, alphas, rhos)
            out.setdefault(fold, []).append((ll, info))
    (a, ia), (a2, ia2) = out[1]
    (b, ib), = out[0]
    assert np.array_equal(a, a2, equal_nan=True) and np.array_equal(ia, ia2)          # repeatable
    assert ia[1] == -1
    assert np.array_equal(ia, ib)
    assert np.array_equal(a, b, equal_nan=True)                                        # BITWISE
    ok = ia == 0
    assert ok.sum() >= M - 2
    if sum(Nl) <= 1400:
        ref, rinfo = oracle.loglik_batch(kname, t, y, s, delays, alphas, rhos, mb, nthreads=8)
        assert np.array_equal(rinfo == 0, ok)
        assert _rel(a[ok], ref[ok]) <= LL_RTOL


def test_fold_is_taken_on_the_default_path_and_leaves_the_other_paths_alone(gp, oracle):
    """Default options, a group of 130 (fused path: folded) and of 20 (three-kernel / right-looking: never folded) against the oracle;
    the dense model matrix (gpcc_model_matrix: assembly only) is what it was."""
    from gpcc_amd import synthetic
    Nl = [400, 300]
    t, y, s, _ = synthetic.simulate_lightcurves(Nl, seed=8)
    for M in (130, 20):
        delays, alphas, rhos = _batch(Nl, y, M, 3)
        with gp.Objective(t, y, s, gp.matern32) as obj:
            obj.set_option("shared_prefix", 0)
            ll, info = obj.loglik_batch(delays, alphas, rhos)
            obj.set_option("fold_assembly", 0)
            ll0, info0 = obj.loglik_batch(delays, alphas, rhos)
            K = obj.model_matrix(delays[0], alphas[0], rhos[0])
        assert (info == 0).all() and (info0 == 0).all()
        assert np.array_equal(ll, ll0)
        ref, _ = oracle.loglik_batch("matern32", t, y, s, delays, alphas, rhos, True, nthreads=8)
        assert _rel(ll, ref) <= LL_RTOL
        Kref, _ = oracle.model_matrix("matern32", t, y, s, delays[0], alphas[0], rhos[0], True)
        assert np.max(np.abs(K - Kref)) <= 1e-12 * np.max(np.abs(Kref))


@pytest.mark.parametrize("kname", ["OU", "matern32", "matern52", "rbf"])
@pytest.mark.parametrize("Nl,mb,asm32", [([700, 600], True, 1), ([700, 600], False, 1), ([700, 600], True, 0), ([450, 400, 300], True, 1),
                                         ([700, 600], False, 0)])
def test_folded_fp32_tiles_return_the_bits_of_assembled_tiles(gp, oracle, kname, Nl, mb, asm32):
    """fp32 handles: tiles evaluated in fp32 (fp32_assemble = 1: every kernel, rbf included) or in fp64 and rounded once
    (fp32_assemble = 0: the separable form; rbf is then not folded at all) -- log-likelihoods, info and the guard's pivot-ratio
    statistics bitwise those of fold_assembly = 0; then against the oracle at the fp32 tolerance."""
    from gpcc_amd import synthetic
    t, y, s, _ = synthetic.simulate_lightcurves(Nl, seed=34)
    M = 70
    delays, alphas, rhos = _batch(Nl, y, M, 6)
    alphas[1, 0] = 0.0
    out = {}
    with gp.Objective(t, y, s, getattr(gp, kname), marginalise_b=mb, precision="fp32", slots_per_stream=64) as obj:
        for k, v in FUSED + (("fp32_assemble", asm32),):
            obj.set_option(k, v)
        for fold in (1, 0):
            obj.set_option("fold_assembly", fold)
            ll, info = obj.loglik_batch(delays, alphas, rhos)
            out[fold] = (ll, info, obj.conditioning(M), obj.get_option("fp32_guard_count"))
    a, ia, ca, na = out[1]
    b, ib, cb, nb = out[0]
    assert ia[1] == -1 and np.array_equal(ia, ib)
    assert np.array_equal(a, b, equal_nan=True)                                        # BITWISE
    assert np.array_equal(ca, cb, equal_nan=True)
    ok = ia == 0
    ref, rinfo = oracle.loglik_batch(kname, t, y, s, delays, alphas, rhos, mb, nthreads=8)
    assert np.array_equal(rinfo == 0, ok)
    assert _rel(a[ok], ref[ok]) <= 1e-3


def test_dense_export_at_headline_size_assembles_every_tile(gp, oracle):
    """gpcc_model_matrix at N = 4096 (2 x 2048: 496 off-diagonal tiles inside one band pair, the ones a folded factorisation would
    leave unwritten) after a folded batch on the same handle: every tile of the dense export is assembled (gpcc_assemble_tiles writes
    the whole lower triangle: the fold must never leak into the dense export) -- every element against the oracle's delayedCovariance
    + Sobs + B (marginaliseb.jl:135), exactly symmetric, no element left at zero."""
    from gpcc_amd import synthetic
    t, y, s, _ = synthetic.simulate_lightcurves([2048, 2048], seed=1)
    alpha, rho = synthetic.default_hyperparameters(y)
    with gp.Objective(t, y, s, "matern32", slots_per_stream=32) as obj:
        M = 32
        d = np.stack([np.zeros(M), np.linspace(0, 20, M)], 1)
        obj.loglik_batch(d, np.tile(alpha, (M, 1)), np.full(M, rho))     # a folded group first (left-looking, fused halves)
        K = obj.model_matrix([0.0, 2.5], alpha, rho)
    assert K.shape == (4096, 4096) and np.array_equal(K, K.T)
    Kref, _ = oracle.model_matrix("matern32", t, y, s, [0.0, 2.5], alpha, rho, True)
    np.testing.assert_allclose(K, Kref, rtol=1e-13, atol=1e-300)
    assert not np.any(K == 0.0)     # (an unwritten tile would read as the zeros of a fresh allocation, or as stale factor data)
