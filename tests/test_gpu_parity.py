"""GPU parity tests: the HIP path, called through the C ABI (gpcc_amd -> ctypes -> libgpcc_hip.so),
against the CPU oracle and the committed golden fixtures.

Tolerances: log-likelihood 1e-8 relative here (BASELINE north_star demands 1e-6 in fp64); matrix
elements 1e-13 relative (exp() of ocml vs libm differ in the last bits); bit-exactness is not
expected of floating-point work."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LL_RTOL = 1e-8


@pytest.fixture(scope="module")
def gp():
    # torch ships its own HIP runtime: when both live in one process, torch has to initialise the GPU
    # first (initialising it after libgpcc_hip.so has run many kernels reported "No HIP GPUs").
    import torch
    torch.cuda.init()
    import gpcc_amd
    return gpcc_amd


@pytest.fixture(autouse=True, params=["tile", "small"])
def kernel_family(request, monkeypatch):
    """Every test of this module runs twice: N <= 191 on the tile kernels (GPCC_SMALL_N=0: assembly + diagonal-block kernels,
    the only path of rounds 1-2) and on the small-N family (gpcc_small_eval: one wave per evaluation, the default).  Larger N
    take the tile kernels either way."""
    monkeypatch.setenv("GPCC_SMALL_N", "0" if request.param == "tile" else "1")
    return request.param


def _rel(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b)) / np.abs(np.asarray(b)))


def _capi_last_error(obj):
    from gpcc_amd import _capi
    lib = _capi.load()
    import ctypes
    lib.gpcc_last_error.restype = ctypes.c_char_p
    return lib.gpcc_last_error(obj._h) or b""


def test_selftest_mfma_map_and_rate(gp):
    tf = gp.selftest(0)
    print("fp64 MFMA rate: %.1f TFLOP/s" % tf)
    assert tf > 5.0


def test_covariance_golden(gp, golden):
    for c in golden["covariances"]:
        Kxy = gp.delayedCovariance(c["kernel"], c["scale"], c["delays"], c["rho"], c["x"], c["y"])
        Kxx = gp.delayedCovariance(gp.KERNELS[c["kernel"]], c["scale"], c["delays"], c["rho"], c["x"])
        np.testing.assert_allclose(Kxy, np.array(c["Kxy"]), rtol=1e-13, atol=1e-300)
        np.testing.assert_allclose(Kxx, np.array(c["Kxx"]), rtol=1e-13, atol=1e-300)


def test_covariance_errors(gp):
    x = [[0.0, 1.0], [0.5]]
    with pytest.raises(AssertionError):
        gp.delayedCovariance(gp.OU, [1.0, 0.0], [0.0, 0.0], 1.0, x)
    with pytest.raises(ValueError):
        gp.delayedCovariance(gp.OU, [1.0, 1.0], [0.0, 0.0], -1.0, x)
    with pytest.raises(TypeError):
        gp.delayedCovariance(lambda a, b: 0.0, [1.0, 1.0], [0.0, 0.0], 1.0, x)


def test_probabilities_golden(gp, golden):
    p = golden["probabilities"]
    np.testing.assert_allclose(gp.getprobabilities(p["loglik"]), p["p_flat"], rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose(gp.getprobabilities(p["loglik"], p["logprior"]), p["p_prior"], rtol=1e-12,
                               atol=1e-300)
    ll = np.random.default_rng(0).standard_normal((5, 7)) * 40 - 2000
    q = gp.getprobabilities(ll)
    assert q.shape == ll.shape and abs(q.sum() - 1) < 1e-12


@pytest.mark.parametrize("right_looking_max", [0, 16])   # left-looking (large groups) / right-looking (small groups)
def test_loglik_golden_cases(gp, golden, right_looking_max):
    worst = 0.0
    for c in golden["cases"]:
        with gp.Objective(c["t"], c["y"], c["sigma"], c["kernel"], marginalise_b=c["marginalise_b"],
                          slots_per_stream=4) as obj:
            obj.set_option("right_looking_max", right_looking_max)
            ll, info = obj.loglik_batch([c["delays"]], [c["alpha"]], [c["rho"]])
        assert info[0] == 0
        worst = max(worst, abs(ll[0] - c["loglik"]) / abs(c["loglik"]))
    print("worst relative error vs golden: %.3e" % worst)
    assert worst <= LL_RTOL


def test_device_exp_within_two_ulp_of_libm(gp, oracle):
    """The device's two exps against libm through the oracle: the 2^(j/64) table + degree-5 polynomial (gpcc_exp_nonpos_tab: fp32
    refinement pass, delayedCovariance) and the degree-13 polynomial (gpcc_exp_nonpos: tile assembly, small-N kernels).  OU with
    rho = 1, scale 1, delay 0 and one point at 0 makes the argument exact (-x), so the comparison is of the exponential alone:
    <= 2 ulp from 1e-6 to the underflow threshold.
    (Assembled elements at large r/rho deviate more -- |x| times a few 1e-16 -- because the device multiplies by 1/rho where
    the reference divides: a property of the argument, not of exp, and immaterial at e^-|x|.)"""
    rng = np.random.default_rng(2)
    x = np.concatenate([10.0 ** rng.uniform(-6, 2.845, 20000), rng.uniform(0, 40, 20000), np.arange(0, 64) * np.log(2) / 64])
    K = gp.delayedCovariance(gp.OU, [1.0], [0.0], 1.0, [x], [np.zeros(1)])[:, 0]
    ref = np.array([oracle.kernel("OU", v, 0.0, 1.0) for v in x])
    ok = ref > 1e-300
    ulp = np.abs(K[ok] - ref[ok]) / np.spacing(ref[ok])
    print("device exp (table) vs libm: worst %.2f ulp over %d arguments" % (ulp.max(), ok.sum()))
    assert ulp.max() <= 2.0
    # the assembly's exp: column 0 of the model matrix of one band whose first point sits at 0 (no B term, unit amplitude)
    for n, fam in ((150, 1), (150, 0), (600, 0)):
        xs = np.concatenate([[0.0], 10.0 ** rng.uniform(-6, 2.845, n - 1)])
        with gp.Objective([xs], [np.zeros(n)], [np.ones(n)], gp.OU, marginalise_b=False) as obj:
            obj.set_option("small_n", fam)
            col = obj.model_matrix([0.0], [1.0], 1.0)[1:, 0] if not fam else None
            if fam:   # the small-N kernels have no dense export: their elements are covered by the log-likelihood tests
                continue
        refc = np.array([oracle.kernel("OU", v, 0.0, 1.0) for v in xs[1:]])
        okc = refc > 1e-300
        ulpc = np.abs(col[okc] - refc[okc]) / np.spacing(refc[okc])
        print("device exp (polynomial, tile assembly, N = %d) vs libm: worst %.2f ulp" % (n, ulpc.max()))
        assert ulpc.max() <= 2.0


def test_model_matrix_and_factor_vs_oracle(gp, oracle):
    from gpcc_amd import synthetic
    t, y, s, _ = synthetic.simulate_lightcurves([171, 171, 171], seed=3)   # N = 513: ragged last tile
    delays, alpha, rho = [0.0, 1.3, 4.1], [0.8, 1.1, 1.7], 2.9
    for kname in ("OU", "rbf", "matern32", "matern52"):
        for mb in (True, False):
            with gp.Objective(t, y, s, kname, marginalise_b=mb, slots_per_stream=2) as obj:
                K = obj.model_matrix(delays, alpha, rho)
                Kref, rref = oracle.model_matrix(kname, t, y, s, delays, alpha, rho, mb)
                np.testing.assert_allclose(K, Kref, rtol=1e-13, atol=1e-300)
                assert np.array_equal(K, K.T)
                _, _, r = obj.constants()
                np.testing.assert_allclose(r, rref, rtol=1e-14, atol=1e-14)
                if kname == "matern32":
                    Lf, info = obj.factor(delays, alpha, rho)
                    Lref, iref = oracle.potrf_lower(Kref)
                    assert info == 0 and iref == 0
                    assert np.max(np.abs(Lf - Lref)) <= 1e-9 * np.max(np.abs(Lref))
                    assert np.array_equal(Lf, np.tril(Lf))


@pytest.mark.parametrize("prec", ["fp64", "fp32"])
def test_same_band_select_free_tiles_elementwise(gp, oracle, prec):
    """The select-free assembly path (a tile whose rows lie in ONE band and whose columns lie in ONE band, off the diagonal) with
    the `+ Sigma_b` term of a SAME-band tile (marginaliseb.jl:94-96, :135): bands of 400 and 150 points put tiles (1,0), (2,0), (2,1)
    wholly inside band 1 -- all four kernels, elements against the oracle at 1e-13 (fp64).  fp32 handles assemble K0 without B
    (it enters through the capacitance matrix) and evaluate these tiles in fp32: the exported matrix there is K0 rounded, 1e-6."""
    from gpcc_amd import synthetic
    Nl = [400, 150]
    t, y, s, _ = synthetic.simulate_lightcurves(Nl, seed=31)
    delays, alpha, rho = [0.0, 2.2], [1.3, 0.6], 3.1
    for kname in ("OU", "rbf", "matern32", "matern52"):
        with gp.Objective(t, y, s, kname, marginalise_b=True, precision=prec, slots_per_stream=2) as obj:
            K = obj.model_matrix(delays, alpha, rho)          # (the dense utilities run the literal fp64 model on every handle)
            Kref, _ = oracle.model_matrix(kname, t, y, s, delays, alpha, rho, True)
            big = np.abs(Kref) > 1e-30
            np.testing.assert_allclose(K[big], Kref[big], rtol=1e-13)
            # (elements e^-x deep in the tail carry the relative error of their ARGUMENT times x -- the device multiplies by a
            # per-evaluation 1/rho where the reference divides: 1e-16 x 600 at 1e-260, immaterial)
            np.testing.assert_allclose(K, Kref, rtol=1e-12, atol=1e-300)
            _, Sb, _ = obj.constants()
            blk = K[128:384, 0:128] - oracle.model_matrix(kname, t, y, s, delays, alpha, rho, False)[0][128:384, 0:128]
            np.testing.assert_allclose(blk, np.full_like(blk, Sb[0]), rtol=1e-10)   # the tile really carries + Sigma_b[band 1]
            ll, info = obj.loglik_batch([delays], [alpha], [rho])
            ref, rinfo = oracle.loglik_batch(kname, t, y, s, [delays], [alpha], [rho], True)
            assert info[0] == 0 and rinfo[0] == 0
            assert abs(ll[0] - ref[0]) <= (LL_RTOL if prec == "fp64" else 1e-3) * abs(ref[0])


@pytest.mark.parametrize("kname", ["OU", "rbf", "matern32", "matern52"])
def test_loglik_medium_vs_oracle_grouped(gp, oracle, kname):
    """N = 2 x 512, 22 evaluations with different (tau, alpha, rho) through groups of 4 on 2 streams."""
    from gpcc_amd import synthetic
    t, y, s, _ = synthetic.simulate_lightcurves([512, 512], seed=2, gap_band=1)
    rng = np.random.default_rng(7)
    M = 22
    delays = np.stack([np.zeros(M), rng.random(M) * 20], 1)
    alpha = 0.5 + rng.random((M, 2)) * 2
    rho = 1.0 + rng.random(M) * 5
    ref, rinfo = oracle.loglik_batch(kname, t, y, s, delays, alpha, rho, True, nthreads=8)
    for rl in (0, 16):
        with gp.Objective(t, y, s, kname, slots_per_stream=4, streams=2) as obj:
            obj.set_option("right_looking_max", rl)
            ll, info = obj.loglik_batch(delays, alpha, rho)
            ok = rinfo == 0
            assert np.array_equal(info == 0, ok)
            assert _rel(ll[ok], ref[ok]) <= LL_RTOL
            # a second call reuses the slots
            ll2, _ = obj.loglik_batch(delays[:5], alpha[:5], rho[:5])
            assert np.array_equal(ll2[ok[:5]], ll[:5][ok[:5]])
            # the closure form
            if ok[0]:
                assert obj(alpha[0], rho[0], delays[0]) == ll[0]


def test_status_codes(gp, golden):
    c = golden["nonpd"]
    with gp.Objective(c["t"], c["y"], c["sigma"], c["kernel"], marginalise_b=c["marginalise_b"]) as obj:
        ll, info = obj.loglik_batch([c["delays"], c["delays"], c["delays"], [0.0, 1.0]],
                                    [c["alpha"], [1.0, -1.0], c["alpha"], c["alpha"]],
                                    [c["rho"], 1.0, 0.0, 2.0])
        assert info[0] > 0 and np.isnan(ll[0])          # PosDefException
        assert info[1] == -1 and np.isnan(ll[1])        # alpha <= 0
        assert info[2] == -2 and np.isnan(ll[2])        # rho <= 0
        with pytest.raises(gp.PosDefException):
            obj(c["alpha"], c["rho"], c["delays"])
        with pytest.raises(AssertionError):
            obj([1.0, -1.0], 1.0, c["delays"])
        with pytest.raises(ValueError):
            obj(c["alpha"], 0.0, c["delays"])
        ll, info = obj.loglik_batch(np.zeros((0, 2)), np.zeros((0, 2)), np.zeros(0))   # empty batch
        assert len(ll) == 0


def test_fixed_b_variant_and_single_band(gp, oracle):
    from gpcc_amd import synthetic
    t, y, s, _ = synthetic.simulate_lightcurves([300], seed=5)
    with gp.Objective(t, y, s, gp.matern52, marginalise_b=False) as obj:
        ll, info = obj.loglik_batch([[0.0], [3.3], [-8.0]], [[1.2]] * 3, [2.2] * 3)
        ref, _ = oracle.loglik_batch("matern52", t, y, s, [[0.0]], [[1.2]], [2.2], False)
    assert (info == 0).all() and _rel(ll, np.repeat(ref, 3)) <= LL_RTOL
    assert np.ptp(ll) <= 1e-10 * abs(ll[0])     # single band: independent of the delay


def test_full_size_properties_and_oracle(gp, oracle):
    """BASELINE size (2 x 2048, Matern-3/2, fp64): size-independent properties on the GPU, plus two
    evaluations against the oracle."""
    from gpcc_amd import synthetic
    t, y, s, _ = synthetic.simulate_lightcurves([2048, 2048], seed=1)
    alpha, rho = synthetic.default_hyperparameters(y)
    grid = np.linspace(0.0, 20.0, 12)
    delays = np.stack([np.zeros_like(grid), grid], 1)
    M = len(grid)
    with gp.Objective(t, y, s, gp.matern32) as obj:
        ll, info = obj.loglik_batch(delays, np.tile(alpha, (M, 1)), np.full(M, rho))      # 12 <= 16: right-looking
        assert (info == 0).all()
        obj.set_option("right_looking_max", 0)
        ll_left, _ = obj.loglik_batch(delays, np.tile(alpha, (M, 1)), np.full(M, rho))    # left-looking
        assert _rel(ll_left, ll) <= 1e-11
        # adding a constant to every delay leaves K unchanged
        ll_s, _ = obj.loglik_batch(delays + 3.5, np.tile(alpha, (M, 1)), np.full(M, rho))
        assert _rel(ll_s, ll) <= 1e-9
        ref, rinfo = oracle.loglik_batch("matern32", t, y, s, delays[[1, 7]], np.tile(alpha, (2, 1)),
                                         np.full(2, rho), True, nthreads=2)
        assert (rinfo == 0).all() and _rel(ll[[1, 7]], ref) <= LL_RTOL
    # permuting observations inside a band leaves the log-likelihood unchanged
    rng = np.random.default_rng(9)
    p0, p1 = rng.permutation(2048), rng.permutation(2048)
    tp, yp, sp = [t[0][p0], t[1][p1]], [y[0][p0], y[1][p1]], [s[0][p0], s[1][p1]]
    with gp.Objective(tp, yp, sp, gp.matern32) as obj:
        ll_p, _ = obj.loglik_batch(delays[:3], np.tile(alpha, (3, 1)), np.full(3, rho))
    assert _rel(ll_p, ll[:3]) <= 1e-9
    p = gp.getprobabilities(ll)
    assert abs(p.sum() - 1) < 1e-12


# ---- BASELINE.json configurations (parity-test cases; bench.py measures configs "N=4096 Matern-3/2") ----
def test_cfg2_two_band_1024_grid256(gp, oracle):
    """cfg2: 2 x 1024, Matern-3/2 fp64, 256-point delay grid: every 32nd point against the oracle."""
    from gpcc_amd import synthetic
    t, y, s, _ = synthetic.simulate_lightcurves([1024, 1024], seed=1)
    alpha, rho = synthetic.default_hyperparameters(y)
    grid = np.linspace(0.0, 20.0, 256)
    delays = np.stack([np.zeros_like(grid), grid], 1)
    with gp.Objective(t, y, s, gp.matern32) as obj:
        ll, info = obj.loglik_batch(delays, np.tile(alpha, (256, 1)), np.full(256, rho))
    assert (info == 0).all()
    idx = np.arange(0, 256, 32)
    ref, _ = oracle.loglik_batch("matern32", t, y, s, delays[idx], np.tile(alpha, (len(idx), 1)),
                                 np.full(len(idx), rho), True, nthreads=8)
    assert _rel(ll[idx], ref) <= LL_RTOL
    p = gp.getprobabilities(ll)
    assert abs(p.sum() - 1) < 1e-12


@pytest.mark.parametrize("kname", ["OU", "rbf", "matern52"])
def test_cfg3_other_kernels_full_size(gp, oracle, kname):
    """cfg3: 2 x 2048 with the other three kernels (Matern-3/2 is covered above): one oracle sample
    each plus the delay-shift invariance over a small grid."""
    from gpcc_amd import synthetic
    t, y, s, _ = synthetic.simulate_lightcurves([2048, 2048], seed=2)
    alpha, rho = synthetic.default_hyperparameters(y)
    grid = np.linspace(0.0, 20.0, 6)
    delays = np.stack([np.zeros_like(grid), grid], 1)
    with gp.Objective(t, y, s, kname) as obj:
        ll, info = obj.loglik_batch(delays, np.tile(alpha, (6, 1)), np.full(6, rho))
        ll_s, info_s = obj.loglik_batch(delays - 1.25, np.tile(alpha, (6, 1)), np.full(6, rho))
    ok = info == 0
    assert ok.any() and np.array_equal(ok, info_s == 0)
    assert _rel(ll_s[ok], ll[ok]) <= 1e-8
    j = int(np.flatnonzero(ok)[0])
    ref, rinfo = oracle.loglik_batch(kname, t, y, s, delays[[j]], [alpha], [rho], True)
    assert rinfo[0] == 0 and _rel(ll[[j]], ref) <= LL_RTOL


def test_cfg4_three_band_2d_grid(gp, oracle):
    """cfg4: 3 x 1365 (N = 4095: ragged last tile), Matern-3/2, a 6 x 6 corner of the 2-D delay grid,
    flattened like README.md:233-235; two points against the oracle."""
    from gpcc_amd import synthetic
    t, y, s, _ = synthetic.simulate_lightcurves([1365, 1365, 1365], seed=3)
    alpha, rho = synthetic.default_hyperparameters(y)
    g1 = np.linspace(0.5, 6.0, 6)
    d2, d3 = np.meshgrid(g1, g1, indexing="ij")
    delays = np.stack([np.zeros(36), d2.ravel(), d3.ravel()], 1)
    with gp.Objective(t, y, s, gp.matern32) as obj:
        ll, info = obj.loglik_batch(delays, np.tile(alpha, (36, 1)), np.full(36, rho))
    assert (info == 0).all()
    ref, rinfo = oracle.loglik_batch("matern32", t, y, s, delays[[4, 29]], np.tile(alpha, (2, 1)),
                                     np.full(2, rho), True, nthreads=2)
    assert (rinfo == 0).all() and _rel(ll[[4, 29]], ref) <= LL_RTOL
    post = gp.getprobabilities(ll.reshape(6, 6))
    assert post.shape == (6, 6) and abs(post.sum() - 1) < 1e-12


# ---- section 8(f) rows: predictTest and the posterior of the offsets ----------------------------------------
def _reference_predict(oracle, kname, t, y, s, delays, alpha, rho, ttest):
    """marginaliseb.jl:259-289 restated with the oracle's matrices and numpy solves (test-side)."""
    L = len(t)
    K, resid = oracle.model_matrix(kname, t, y, s, delays, alpha, rho, True)            # KSobsB, Y - bbar
    mub = np.array([np.mean(a) for a in y])
    Sigb = 100 * np.array([np.var(a, ddof=1) for a in y])
    bt = np.concatenate([np.full(len(a), l) for l, a in enumerate(t)])
    bs = np.concatenate([np.full(len(a), l) for l, a in enumerate(ttest)])
    kB = oracle.delayed_covariance(kname, alpha, delays, rho, t, ttest) + (bt[:, None] == bs[None, :]) * Sigb[bt][:, None]
    cB = oracle.delayed_covariance(kname, alpha, delays, rho, ttest) + (bs[:, None] == bs[None, :]) * Sigb[bs][:, None]
    Sig = cB - kB.T @ np.linalg.solve(K, kB)
    Sig = (Sig + Sig.T) / 2 + 1e-8 * np.eye(len(bs))
    mu = kB.T @ np.linalg.solve(K, resid) + mub[bs]
    return mu, Sig


@pytest.mark.parametrize("shape", [([60, 50], [40, 40]), ([150, 140, 130], [30, 0, 45]), ([700, 650], [200, 150])])
def test_predict_vs_reference_formulas(gp, oracle, shape):
    from gpcc_amd import synthetic
    Nl, Nt = shape
    t, y, s, _ = synthetic.simulate_lightcurves(Nl, seed=4)
    L = len(Nl)
    delays = [0.0, 2.0, 4.0][:L]
    alpha = [1.0, 1.4, 0.8][:L]
    rho = 3.1
    rng = np.random.default_rng(1)
    span = max(float(np.max(a)) for a in t)
    ttest = [np.sort(rng.random(n) * (span + 10) - 5) for n in Nt]
    with gp.Objective(t, y, s, gp.matern32) as obj:
        mu, Sig = obj.predict(delays, alpha, rho, ttest)
        mu2, Sig2 = obj.predict(delays, alpha, rho, ttest)          # repeatable
    assert np.array_equal(mu, mu2) and np.array_equal(Sig, Sig2)
    mu_ref, Sig_ref = _reference_predict(oracle, "matern32", t, y, s, delays, alpha, rho, ttest)
    scale = np.max(np.abs(Sig_ref))
    assert np.max(np.abs(mu - mu_ref)) <= 1e-8 * np.max(np.abs(mu_ref))
    assert np.max(np.abs(Sig - Sig_ref)) <= 1e-8 * scale
    assert np.array_equal(Sig, Sig.T)


def test_predict_fixed_b_variant(gp, oracle):
    """predictTest of the fixed-b variant (gpccfixdelay.jl:244-266): no B* term in kB* / cB, KSobsB = K + Sobs,
    mean kB*' (KSobsB \\ (Y - Qb)) + Q* b with b = (Q'Q) \\ Q'Y (the band means, :94-96)."""
    from gpcc_amd import synthetic
    t, y, s, _ = synthetic.simulate_lightcurves([210, 190, 120], seed=12)
    delays, alpha, rho = [0.0, 2.0, 4.0], [1.2, 0.9, 2.0], 2.7
    rng = np.random.default_rng(5)
    ttest = [np.sort(rng.random(n) * 90 - 5) for n in (33, 0, 41)]
    with gp.Objective(t, y, s, gp.matern52, marginalise_b=False) as obj:
        mu, Sig = obj.predict(delays, alpha, rho, ttest)
        with pytest.raises(gp.GpccError):          # postb exists only where b is marginalised (marginaliseb.jl:248-252)
            obj.posterior_offsets(delays, alpha, rho)
    K, resid = oracle.model_matrix("matern52", t, y, s, delays, alpha, rho, False)        # K + Sobs, Y - Qb
    b = np.array([np.mean(a) for a in y])
    bs = np.concatenate([np.full(len(a), l) for l, a in enumerate(ttest)]).astype(int)
    kB = oracle.delayed_covariance("matern52", alpha, delays, rho, t, ttest)
    cB = oracle.delayed_covariance("matern52", alpha, delays, rho, ttest)
    Sref = cB - kB.T @ np.linalg.solve(K, kB)
    Sref = (Sref + Sref.T) / 2 + 1e-8 * np.eye(len(bs))
    mref = kB.T @ np.linalg.solve(K, resid) + b[bs]
    assert np.max(np.abs(mu - mref)) <= 1e-8 * np.max(np.abs(mref))
    assert np.max(np.abs(Sig - Sref)) <= 1e-8 * np.max(np.abs(Sref))
    assert np.array_equal(Sig, Sig.T)


def test_posterior_offsets_vs_reference_formulas(gp, oracle):
    from gpcc_amd import synthetic
    t, y, s, _ = synthetic.simulate_lightcurves([300, 260, 220], seed=6)
    delays, alpha, rho = [0.0, 2.0, 4.0], [1.0, 2.1, 3.7], 3.5
    with gp.Objective(t, y, s, gp.OU) as obj:
        mu, Sig = obj.posterior_offsets(delays, alpha, rho)
    K0, _ = oracle.model_matrix("OU", t, y, s, delays, alpha, rho, False)          # Sobs + K (no B)
    Nl = [len(a) for a in t]
    Q = np.zeros((sum(Nl), 3))
    o = 0
    for l, n in enumerate(Nl):
        Q[o:o + n, l] = 1
        o += n
    Y = np.concatenate(y)
    Sigb = np.diag(100 * np.array([np.var(a, ddof=1) for a in y]))
    mub = np.array([np.mean(a) for a in y])
    Sref = np.linalg.inv(np.linalg.inv(Sigb) + Q.T @ np.linalg.solve(K0, Q))       # marginaliseb.jl:248
    mref = Sref @ (Q.T @ np.linalg.solve(K0, Y) + np.linalg.solve(Sigb, mub))       # :250
    np.testing.assert_allclose(Sig, Sref, rtol=1e-7, atol=1e-12)
    np.testing.assert_allclose(mu, mref, rtol=1e-7)


def test_mvnormal_logpdf_dense(gp, oracle):
    rng = np.random.default_rng(2)
    for n in (1, 37, 128, 300):
        A = rng.standard_normal((n, n))
        Sig = A @ A.T + n * np.eye(n)
        mu, x = rng.standard_normal(n), rng.standard_normal(n)
        Lc = np.linalg.cholesky(Sig)
        zz = np.linalg.solve(Lc, x - mu)
        ref = -0.5 * (n * np.log(2 * np.pi) + 2 * np.sum(np.log(np.diag(Lc)))) - 0.5 * zz @ zz
        assert abs(gp.mvnormal_logpdf(mu, Sig, x) - ref) <= 1e-10 * abs(ref)
    with pytest.raises(gp.PosDefException):
        gp.mvnormal_logpdf(np.zeros(3), np.diag([1.0, -1.0, 1.0]), np.zeros(3))


def test_gpcc_end_to_end_small(gp, oracle):
    """gpcc() on a simulatedata-sized problem (cfg1: 60 + 50 observations, iterations = 50, true delays):
    the fit improves on the random start, pred() has the reference's three call forms, postb is sane."""
    from gpcc_amd import synthetic
    t, y, s, td = synthetic.simulate_lightcurves([60, 50], seed=1, gap_band=1, span=20.0)
    loglikel, pred, (alpha, postb, rho) = gp.gpcc(t, y, s, kernel=gp.matern32, delays=td, iterations=50, rhomax=20.0)
    ref, rinfo = oracle.loglik_batch("matern32", t, y, s, [td], [alpha], [rho], True)
    assert rinfo[0] == 0 and abs(loglikel - ref[0]) <= 1e-8 * abs(ref[0])     # returned value == objective at the optimum
    assert alpha.shape == (2,) and np.all(alpha > 0) and 0.1 < rho < 20.0
    trange = np.arange(-10.0, 25.0, 0.5)
    mu_b, sd_b = pred(trange)                                                  # per band (README.md:119-120)
    assert len(mu_b) == 2 and mu_b[0].shape == trange.shape and np.all(sd_b[1] >= 1e-3)
    mu_j, Sig_j = pred([trange, trange])                                       # joint
    assert np.allclose(np.concatenate(mu_b), mu_j) and Sig_j.shape == (2 * len(trange),) * 2
    mu_postb, Sig_postb = postb
    assert abs(mu_postb[0] - np.mean(y[0])) < 3.0 and np.all(np.linalg.eigvalsh(Sig_postb) > 0)
    # test log-likelihood (README.md:150-153): equals the dense formula with the joint prediction
    tt = [np.array([9.0, 10.0, 11.0]), np.array([9.5, 10.5, 11.5])]
    yt = [np.array([6.34, 5.49, 5.38]), np.array([13.08, 12.37, 15.69])]
    st = [np.array([0.34, 0.42, 0.2]), np.array([0.87, 0.8, 0.66])]
    got = pred(tt, yt, st)
    mu, Sig = pred(tt)
    Sig = Sig + np.diag(np.concatenate(st) ** 2)
    r = np.concatenate(yt) - mu
    refll = -0.5 * (6 * np.log(2 * np.pi) + np.linalg.slogdet(Sig)[1] + r @ np.linalg.solve(Sig, r))
    assert abs(got - refll) <= 1e-9 * abs(refll)


def test_predict_loglik_nearestposdef_fallback(gp):
    """marginaliseb.jl:327-341: a PosDefException in the test log-likelihood is retried once on
    nearestposdef(Sigma_pred; minimumeigenvalue = 1e-6).  The joint prediction is stubbed with an indefinite
    covariance so that the first device Cholesky fails."""
    from gpcc_amd import fit
    rng = np.random.default_rng(11)
    n = 40
    A = rng.standard_normal((n, n))
    Sig = A @ A.T
    w, V = np.linalg.eigh(Sig)
    w[:3] = [-0.5, -1e-3, 0.0]
    Sig = (V * w) @ V.T
    Sig = 0.5 * (Sig + Sig.T)
    mu = rng.standard_normal(n)

    class Stub:
        L, device = 2, 0

        def predict(self, delays, alpha, rho, ttest):
            return mu.copy(), Sig.copy()

    pred = fit.Predictor(Stub(), [0.0, 1.0], [1.0, 1.0], 2.0)
    tt = [np.arange(n // 2, dtype=float)] * 2
    yt = [rng.standard_normal(n // 2)] * 2
    st = [np.zeros(n // 2)] * 2
    with pytest.raises(gp.PosDefException):
        gp.mvnormal_logpdf(mu, Sig, np.concatenate(yt))
    got = pred(tt, yt, st)
    P = fit.nearestposdef(Sig, minimumeigenvalue=1e-6)
    r = np.concatenate(yt) - mu
    Lc = np.linalg.cholesky(P)
    z = np.linalg.solve(Lc, r)
    ref = -0.5 * (n * np.log(2 * np.pi) + 2 * np.sum(np.log(np.diag(Lc)))) - 0.5 * z @ z
    assert abs(got - ref) <= 1e-6 * abs(ref)       # the lifted 1e-6 eigenvalues amplify rounding (cond ~ 1e8)


def test_gpcc_grid_device_matches_oracle_injected(gp, oracle):
    """The lock-step fit over a small delay grid with the device objective vs the same host logic over the
    oracle: optimised log-likelihoods agree (trajectories may differ in the last bits of the objective)."""
    from gpcc_amd import fit, synthetic
    t, y, s, _ = synthetic.simulate_lightcurves([60, 50], seed=1, gap_band=1, span=20.0)
    grid = np.arange(0.0, 6.01, 1.0)
    cand = np.stack([np.zeros_like(grid), grid], 1)
    dev = gp.gpcc_grid(t, y, s, kernel=gp.OU, candidatedelays=cand, iterations=80, rhomax=20.0)

    class O:
        def loglik_batch(self, d, a, r):
            return oracle.loglik_batch("OU", t, y, s, d, a, r, True, nthreads=8)

    ref = fit.gpcc_grid(t, y, s, kernel="OU", candidatedelays=cand, iterations=80, rhomax=20.0, objective=O())
    assert np.max(np.abs(dev.loglikel - ref.loglikel) / np.abs(ref.loglikel)) <= 1e-6
    p = gp.getprobabilities(dev.loglikel)
    assert abs(grid[np.argmax(p)] - 2.0) <= 1.0


def test_native_grid_fit_equals_python_host_logic(gp, oracle):
    """gpcc_grid_loglik (C++ lock-step Nelder-Mead inside the library) vs the numpy host logic driving the same
    device objective with the library's own `unpack`: identical requests, hence identical bits; the returned value
    is the objective at the returned hyper-parameters (checked against the oracle); restarts; 3 bands; the
    NULL-init_params recipe equals passing gpcc_initial_params' output."""
    from gpcc_amd import api, fit, synthetic
    for Nl, kern, R, iters in (([60, 50], "matern32", 1, 40), ([45, 40, 35], "OU", 3, 25)):
        L = len(Nl)
        t, y, s, _ = synthetic.simulate_lightcurves(Nl, seed=2, span=20.0)
        rng = np.random.default_rng(5)
        cand = np.concatenate([np.zeros((9, 1)), rng.uniform(0.0, 8.0, (9, L - 1))], axis=1)
        with gp.Objective(t, y, s, kern) as obj:
            obj.set_option("fit_speculate", 0)   # (the numpy mirror makes the plain requests; speculation: test_gpu_small_n.py)
            nat = fit.gpcc_grid(t, y, s, kernel=kern, candidatedelays=cand, iterations=iters, numberofrestarts=R,
                                rhomax=30.0, objective=obj, engine="native")
            py = fit.gpcc_grid(t, y, s, kernel=kern, candidatedelays=cand, iterations=iters, numberofrestarts=R,
                               rhomax=30.0, objective=obj, engine="python", unpack=api.unpack_params)
            assert np.array_equal(nat.loglikel, py.loglikel)
            assert np.array_equal(nat.alpha, py.alpha) and np.array_equal(nat.rho, py.rho)
            assert np.array_equal(nat.iterations_done, py.iterations_done)
            assert (nat.f_calls, nat.rounds) == (py.f_calls, py.rounds)
            ref, rinfo = oracle.loglik_batch(kern, t, y, s, cand, nat.alpha, nat.rho, True, nthreads=8)
            assert (rinfo == 0).all() and np.max(np.abs(nat.loglikel - ref) / np.abs(ref)) <= 1e-8
            # the library's own random candidates: NULL == explicit
            init = obj.initial_params(numberofrestarts=R, initialrandom=4, rhomin=0.1, rhomax=30.0, seed=7)
            assert init.shape == (R, 4, L + 1) and np.all(np.isfinite(init))
            a = obj.grid_loglik(cand, iters, numberofrestarts=R, initialrandom=4, rhomin=0.1, rhomax=30.0, seed=7)
            b = obj.grid_loglik(cand, iters, numberofrestarts=R, initialrandom=4, rhomin=0.1, rhomax=30.0, seed=123,
                                init_params=init)
            assert all(np.array_equal(u, v) for u, v in zip(a[:5], b[:5])) and a[5] == b[5]
            assert (a[3] == 0).all() and np.all(a[0] > ref - 50.0)
            c = obj.grid_loglik(cand, iters, numberofrestarts=R, initialrandom=4, rhomin=0.1, rhomax=30.0, seed=8)
            assert not np.array_equal(a[0], c[0])                       # the seed matters
            with pytest.raises(gp.GpccError):
                obj.grid_loglik(cand, iters, rhomin=5.0, rhomax=5.0)


# ---- fp32 path: K0 = delayedCovariance + Sobs factorised in fp32 (fp64 diagonal blocks and right-hand sides),
# ---- the offset prior B = Q Sigma_b Q' through the L x L capacitance matrix in fp64.  Bar: 1e-3 relative.
FP32_RTOL = 1e-3


def test_fp32_golden_cases(gp, golden):
    worst = 0.0
    for c in golden["cases"]:
        with gp.Objective(c["t"], c["y"], c["sigma"], c["kernel"], marginalise_b=c["marginalise_b"], precision="fp32",
                          slots_per_stream=4) as obj:
            ll, info = obj.loglik_batch([c["delays"]], [c["alpha"]], [c["rho"]])
        assert info[0] == 0, (c["kernel"], c["marginalise_b"], info)
        worst = max(worst, abs(ll[0] - c["loglik"]) / abs(c["loglik"]))
    print("fp32 worst relative error vs golden: %.3e" % worst)
    assert worst <= FP32_RTOL


@pytest.mark.parametrize("kname,mb", [("matern52", True), ("matern32", True), ("OU", False), ("rbf", True)])
def test_fp32_medium_vs_oracle(gp, oracle, kname, mb):
    from gpcc_amd import synthetic
    t, y, s, _ = synthetic.simulate_lightcurves([1024, 1024], seed=2, gap_band=1)
    alpha, rho = synthetic.default_hyperparameters(y)
    rng = np.random.default_rng(3)
    M = 12
    delays = np.stack([np.zeros(M), rng.random(M) * 20], 1)
    alphas = np.tile(alpha, (M, 1)) * (0.7 + 0.6 * rng.random((M, 2)))
    rhos = 2.0 + 3 * rng.random(M)
    ref, rinfo = oracle.loglik_batch(kname, t, y, s, delays, alphas, rhos, mb, nthreads=12)
    ok = rinfo == 0
    assert ok.sum() >= M // 2
    res = {}
    with gp.Objective(t, y, s, kname, marginalise_b=mb, precision="fp32") as obj:
        obj.set_option("fp32_chain", 0)   # the fp32 tile kernels are what this test is about (a group of 12 would go to the fp64 twin)
        for mode in (0, 1):   # elements of the fp32 tiles evaluated in fp64 and rounded once / evaluated in fp32 (option "fp32_assemble")
            obj.set_option("fp32_assemble", mode)
            ll, info = obj.loglik_batch(delays, alphas, rhos)
            assert (info[ok] == 0).all()
            res[mode] = ll
            err = _rel(ll[ok], ref[ok])
            print("fp32 %s mb=%s fp32_assemble=%d: max rel err %.3e" % (kname, mb, mode, err))
            assert err <= FP32_RTOL
    assert not np.array_equal(res[0][ok], res[1][ok])      # the option does select another assembly ...
    assert _rel(res[1][ok], res[0][ok]) <= 1e-5            # ... whose results sit far inside the bar of the fp32 path


def test_fp32_status_codes(gp, golden):
    c = golden["nonpd"]
    with gp.Objective(c["t"], c["y"], c["sigma"], c["kernel"], marginalise_b=False, precision="fp32") as obj:
        ll, info = obj.loglik_batch([c["delays"], c["delays"], c["delays"]], [c["alpha"], [1.0, -1.0], c["alpha"]],
                                    [c["rho"], 1.0, 0.0])
        assert info[0] > 0 and np.isnan(ll[0]) and info[1] == -1 and info[2] == -2


def _large_cases(tags):
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "tests", "golden", "gpcc_golden_large.json")) as f:
        cases = json.load(f)["cases"]
    groups = {}
    for c in cases:
        if c["tag"] in tags:
            groups.setdefault((tuple(c["Nl"]), c["seed"], c["sigma"], c["kernel"], c["marginalise_b"]), []).append(c)
    return groups


def _regenerate(case):
    from gpcc_amd import synthetic
    t, y, s, _ = synthetic.simulate_lightcurves(case["Nl"], seed=case["seed"], sigma=case["sigma"])
    chk = [float(np.sum(np.concatenate(t))), float(np.sum(np.concatenate(y))), float(np.sum(np.concatenate(s) ** 2))]
    np.testing.assert_allclose(chk, case["data_checksum"], rtol=1e-12, err_msg="synthetic generator drifted")
    return t, y, s


@pytest.mark.parametrize("tags", [("cfg2", "cfg3", "cfg4", "illcond", "adversarial"), ("cfg5",)])
def test_baseline_size_goldens_fp64_and_fp32(gp, tags):
    """BASELINE.json configs 2-5 (incl. cfg5: 2 x 8192 = N 16384, Matern-5/2), ill-conditioned N = 2048 problems and the survivors
    of the adversarial search against the fp32 guard (tools/adversarial_fp32.py; the first four broke the round-2 guard)
    against the committed scipy/LAPACK values of tests/golden/gpcc_golden_large.json -- an independent check at the
    sizes the oracle cannot reach in test time.  fp64 device <= 1e-8, fp32 device <= 1e-3 (BASELINE north_star)."""
    worst = {"fp64": 0.0, "fp32": 0.0}
    for key, cs in _large_cases(tags).items():
        t, y, s = _regenerate(cs[0])
        ref = np.array([c["loglik"] for c in cs])
        args = ([c["delays"] for c in cs], [c["alpha"] for c in cs], [c["rho"] for c in cs])
        for prec, tol in (("fp64", LL_RTOL), ("fp32", FP32_RTOL)):
            with gp.Objective(t, y, s, cs[0]["kernel"], marginalise_b=cs[0]["marginalise_b"], precision=prec,
                              slots_per_stream=8) as obj:
                # a handful of evaluations take the small-group kernels (gpcc_small_step); right_looking_max = 0 sends the
                # same batch down the left-looking path of the big sweeps (gpcc_syrk_diag + gpcc_update_solve, nt up to 128)
                for rlm in (None, 0):
                    if rlm is not None:
                        obj.set_option("right_looking_max", rlm)
                        obj.set_option("shared_prefix", 0)   # (identical band-1 parameters would otherwise select that mode's kernels)
                        obj.set_option("fused_solve_min", 1) # (and small groups the three-kernel path)
                    ll, info = obj.loglik_batch(*args)
                    assert (info == 0).all(), (key, prec, rlm, info)
                    err = _rel(ll, ref)
                    worst[prec] = max(worst[prec], err)
                    assert err <= tol, (cs[0]["tag"], cs[0]["kernel"], prec, rlm, err)
    print("%s vs LAPACK goldens: worst rel fp64 %.2e, fp32 %.2e" % ("/".join(tags), worst["fp64"], worst["fp32"]))


def test_device_pointer_api_matches_host_api(gp):
    """gpcc_loglik_batch_device (torch CUDA tensors, asynchronous on torch's stream) vs the host-pointer
    form, and gpcc_probabilities_device vs getprobabilities."""
    import ctypes

    import torch

    from gpcc_amd import _capi, synthetic
    t, y, s, _ = synthetic.simulate_lightcurves([300, 280], seed=8)
    alpha, rho = synthetic.default_hyperparameters(y)
    M = 37
    grid = np.linspace(0.0, 10.0, M)
    delays = np.stack([np.zeros(M), grid], 1)
    dev = torch.device("cuda", 0)
    with gp.Objective(t, y, s, gp.matern32, slots_per_stream=8, streams=2) as obj:
        ref, rinfo = obj.loglik_batch(delays, np.tile(alpha, (M, 1)), np.full(M, rho))
        side = torch.cuda.Stream(dev)
        with torch.cuda.stream(side):   # a non-default caller stream: fork/join must order the work behind it
            d_d = torch.as_tensor(delays, device=dev)
            d_a = torch.as_tensor(np.tile(alpha, (M, 1)), device=dev)
            d_r = torch.full((M,), float(rho), dtype=torch.float64, device=dev)
            out, info = obj.loglik_batch_device(d_d, d_a, d_r)
            prob = torch.empty(M, dtype=torch.float64, device=dev)
            _capi.check(_capi.load().gpcc_probabilities_device(M, out.data_ptr(), None, prob.data_ptr(),
                                                               ctypes.c_void_p(side.cuda_stream)))
        side.synchronize()
        assert np.array_equal(out.cpu().numpy(), ref) and np.array_equal(info.cpu().numpy(), rinfo)
        np.testing.assert_allclose(prob.cpu().numpy(), gp.getprobabilities(ref), rtol=1e-12)


def test_abi_misuse_is_reported_not_crashed(gp):
    """Every entry point returns an error code + message for bad arguments (never throws across the boundary)."""
    import ctypes

    from gpcc_amd import _capi
    lib = _capi.load()
    h = ctypes.c_void_p()
    Nl = (ctypes.c_int * 1)(3)
    v = (ctypes.c_double * 3)(0.0, 1.0, 2.0)
    assert lib.gpcc_create(ctypes.byref(h), 0, Nl, v, v, v, 0, 1, 0, 0) == -1 and "L=0" in _capi.last_error()
    assert lib.gpcc_create(ctypes.byref(h), 1, Nl, v, v, v, 9, 1, 0, 0) == -1 and "kernel_id" in _capi.last_error()
    assert lib.gpcc_create(ctypes.byref(h), 1, Nl, v, v, v, 0, 1, 7, 0) == -1 and "precision" in _capi.last_error()
    assert lib.gpcc_create(ctypes.byref(h), 1, Nl, v, v, v, 0, 1, 0, 99) == -1 and "device_id" in _capi.last_error()
    one = (ctypes.c_int * 1)(1)
    assert lib.gpcc_create(ctypes.byref(h), 1, one, v, v, v, 0, 1, 0, 0) == -1      # var(y) undefined with n = 1
    assert lib.gpcc_create(ctypes.byref(h), 1, Nl, v, v, v, 0, 1, 0, 0) == 0 and h.value
    assert lib.gpcc_set_option(h, b"no_such_option", 1) == -1 and "unknown option" in _capi.last_error(h)
    assert lib.gpcc_set_option(h, b"streams", 99) == -1
    out = (ctypes.c_double * 1)()
    info = (ctypes.c_int * 1)()
    assert lib.gpcc_loglik_batch(h, -1, v, v, v, out, info) == -1
    assert lib.gpcc_loglik_batch(h, 1, None, v, v, out, info) == -1 and "NULL" in _capi.last_error(h)
    assert lib.gpcc_loglik_batch(None, 1, v, v, v, out, info) == -1
    assert lib.gpcc_profile_get(h, 99, None, None) == -1
    assert lib.gpcc_probabilities(0, v, None, v, 0) == -1
    assert lib.gpcc_destroy(h) == 0 and lib.gpcc_destroy(None) == 0


@pytest.mark.parametrize("sigma", [0.75, 0.1])
def test_hyperparameter_envelope(gp, oracle, sigma):
    """The ranges a Nelder-Mead run visits (README.md:172 uses rhomax = 300): alpha 1e-2..1e2, rho 0.1..300, on the
    benchmark's noise level and on sigma = 0.1 (cond(K0) ~ alpha^2 N_eff / sigma^2 up to ~1e8).  fp64 keeps <= 1e-9
    where the oracle itself is that well determined; fp32 keeps the 1e-3 bar EVERYWHERE: evaluations whose mean or largest
    pivot ratio exceeds the guard's limits are repeated in fp64, the rest are plain fp32.  With the guard
    off the same batch breaks the bar, i.e. the guard is what holds it."""
    from gpcc_amd import synthetic
    rng = np.random.default_rng(11)
    t, y, s, _ = synthetic.simulate_lightcurves([330, 300], seed=3, gap_band=1, sigma=sigma)
    M = 96
    delays = np.stack([np.zeros(M), rng.random(M) * 20], 1)
    alpha = 10.0 ** rng.uniform(-2, 2, (M, 2))
    rho = 10.0 ** rng.uniform(-1, np.log10(300), M)
    ref, rinfo = oracle.loglik_batch("matern32", t, y, s, delays, alpha, rho, True, nthreads=16)
    with gp.Objective(t, y, s, gp.matern32) as obj:
        ll, info = obj.loglik_batch(delays, alpha, rho)
    assert np.array_equal(info == 0, rinfo == 0)
    ok = rinfo == 0
    assert _rel(ll[ok], ref[ok]) <= (1e-9 if sigma > 0.5 else 1e-7)
    with gp.Objective(t, y, s, gp.matern32, precision="fp32") as obj:
        ll32, info32 = obj.loglik_batch(delays, alpha, rho)
        repeated = obj.get_option("fp32_guard_count")
        ratios = obj.conditioning(M)[:, 0] / 630.0
        largest = obj.conditioning(M)[:, 1]
        obj.set_option("fp32_guard", 0)
        raw, rawinfo = obj.loglik_batch(delays, alpha, rho)
    assert (info32[ok] == 0).all()
    err, rawerr = _rel(ll32[ok], ref[ok]), _rel(raw[ok & (rawinfo == 0)], ref[ok & (rawinfo == 0)])
    print("sigma %.2f: fp32 guarded %.2e (%d of %d evaluations repeated in fp64, mean pivot ratio %.1f .. %.1e), raw fp32 %.2e"
          % (sigma, err, repeated, M, np.nanmin(ratios), np.nanmax(ratios), rawerr))
    assert err <= FP32_RTOL
    assert 0 < repeated < M                       # both regimes occur in this batch
    if sigma < 0.5:
        assert rawerr > FP32_RTOL                 # without the guard fp32 (even refined) does not hold the bar here
    unguarded = ok & (rawinfo == 0) & (ratios <= 250.0) & (largest <= 4.0e3)   # evaluations that stayed in fp32 (factorised in fp32, mean AND
                                                              # largest pivot ratio below the guard's limits) are bitwise the raw results
    assert np.isinf(ratios[rawinfo > 0]).all()                # a broken-down fp32 factorisation reports no pivot ratios (and is repeated)
    assert np.array_equal(ll32[unguarded], raw[unguarded])


# ---- section 8(f).4: evaluations of a group that share band 1's (alpha, rho, delay) reuse the leader's leading tile rows
@pytest.mark.parametrize("Nl,prec", [([300, 280], "fp64"), ([2048, 2048], "fp64"), ([260, 200, 190], "fp64"), ([384, 300], "fp32")])
def test_shared_prefix_is_bitwise_identical(gp, Nl, prec):
    from gpcc_amd import synthetic
    t, y, s, _ = synthetic.simulate_lightcurves(Nl, seed=7)
    alpha, rho = synthetic.default_hyperparameters(y)
    L = len(Nl)
    M = 70                                         # two groups of 40: 40 and 30 evaluations (> right_looking_max)
    rng = np.random.default_rng(5)
    delays = np.concatenate([np.zeros((M, 1)), rng.random((M, L - 1)) * 15], 1)
    alphas = np.tile(alpha, (M, 1))
    alphas[:, 1:] *= 0.6 + 0.8 * rng.random((M, L - 1))      # the other bands' amplitudes differ per evaluation
    rhos = np.full(M, rho)
    with gp.Objective(t, y, s, gp.matern32, precision=prec, slots_per_stream=40) as obj:
        assert obj.get_option("share_tiles") == Nl[0] // 128
        obj.set_option("chain_max", 12)            # (the remainder group of 30 would take the persistent launch at the smallest size: this test compares the tile paths)
        obj.set_option("shared_prefix", 0)
        obj.set_option("hybrid_tail", 0)           # (the right-looking tail sums in another order; the bitwise comparison is between plain left-looking runs)
        dflt, dinfo = obj.loglik_batch(delays, alphas, rhos)     # default plain path: panel solve fused into the update
        obj.set_option("fused_solve", 0)           # the three-kernel path, whose kernels the shared-prefix mode runs
        ref, rinfo = obj.loglik_batch(delays, alphas, rhos)
        obj.set_option("shared_prefix", 1)         # auto: the host-pointer API sees identical band-1 parameters
        ll, info = obj.loglik_batch(delays, alphas, rhos)
        assert (rinfo == 0).all() and np.array_equal(info, rinfo) and np.array_equal(ll, ref)
        assert (dinfo == 0).all() and _rel(dflt, ref) <= (1e-11 if prec == "fp64" else 1e-5)   # same arithmetic, other summation order
        # not shareable (band-1 delay differs): auto mode must fall back to the plain path
        d2 = delays.copy()
        d2[3, 0] = 0.25
        a, ia = obj.loglik_batch(d2, alphas, rhos)
        obj.set_option("shared_prefix", 0)
        b, ib = obj.loglik_batch(d2, alphas, rhos)
        assert np.array_equal(a, b) and np.array_equal(ia, ib)


def test_shared_prefix_failure_inside_prefix(gp):
    """A non-positive pivot inside the shared rows (duplicate times in band 1, sigma = 0, fixed-b variant) is
    reported for every evaluation exactly as without sharing."""
    rng = np.random.default_rng(2)
    t0 = rng.random(300) * 50
    t0[17] = t0[5]                                 # duplicate time -> singular K0 with sigma = 0
    t = [t0, rng.random(200) * 50]
    y = [rng.standard_normal(300), rng.standard_normal(200)]
    s = [np.zeros(300), np.full(200, 0.3)]
    M = 30
    delays = np.stack([np.zeros(M), np.linspace(0, 5, M)], 1)
    alphas, rhos = np.tile([1.0, 1.3], (M, 1)), np.full(M, 2.0)
    with gp.Objective(t, y, s, gp.OU, marginalise_b=False, slots_per_stream=32) as obj:
        obj.set_option("shared_prefix", 0)
        ref, rinfo = obj.loglik_batch(delays, alphas, rhos)
        obj.set_option("shared_prefix", 1)
        ll, info = obj.loglik_batch(delays, alphas, rhos)
    assert (rinfo > 0).all() and np.array_equal(info, rinfo) and np.isnan(ll).all()


def test_randomised_differential_vs_oracle(gp, oracle):
    """60 random problems (1-4 bands, ragged sizes down to 1-2 points per band, all kernels, both b-modes, random
    batch sizes that cross the right-looking / left-looking / shared-prefix switches) against the oracle."""
    rng = np.random.default_rng(20240918)
    knames = ["OU", "rbf", "matern32", "matern52"]
    worst = 0.0
    for trial in range(60):
        L = int(rng.integers(1, 5))
        mb = bool(rng.integers(0, 2))
        lo = 2 if mb else 1
        Nl = [int(rng.integers(lo, 40)) if rng.random() < 0.3 else int(rng.integers(lo, 330)) for _ in range(L)]
        t = [rng.random(n) * rng.uniform(5, 60) for n in Nl]
        y = [rng.uniform(-5, 30) + rng.uniform(0.2, 3) * np.sin(0.2 * t[l] + l) + rng.standard_normal(Nl[l]) * 0.4
             for l in range(L)]
        s = [rng.uniform(0.05, 1.0, n) for n in Nl]
        kname = knames[int(rng.integers(0, 4))]
        M = int(rng.choice([1, 3, 25, 40]))
        delays = rng.uniform(-5, 20, (M, L))
        alpha = 10.0 ** rng.uniform(-1, 1, (M, L))
        rho = 10.0 ** rng.uniform(-0.7, 1.7, M)
        if M >= 25 and rng.random() < 0.5:          # make the batch shareable: same band-1 parameters
            delays[:, 0], alpha[:, 0], rho[:] = delays[0, 0], alpha[0, 0], rho[0]
        ref, rinfo = oracle.loglik_batch(kname, t, y, s, delays, alpha, rho, mb, nthreads=8)
        with gp.Objective(t, y, s, kname, marginalise_b=mb, slots_per_stream=32) as obj:
            ll, info = obj.loglik_batch(delays, alpha, rho)
        ok = rinfo == 0
        assert np.array_equal(info == 0, ok), (trial, Nl, kname, mb, info, rinfo)
        if ok.any():
            worst = max(worst, _rel(ll[ok], ref[ok]))
    print("randomised differential test: worst relative error %.3e" % worst)
    assert worst <= LL_RTOL


EXTREME_SHAPES = [([2, 3, 2, 5, 2, 4, 3, 2], "matern32", True), ([40, 3, 17, 64, 2, 90, 31, 128], "OU", True),
                  ([40, 3, 17, 64, 2, 90, 31, 128], "matern52", False), ([1], "rbf", False), ([2], "rbf", True),
                  ([127], "OU", True), ([128], "OU", True), ([129], "OU", True), ([1, 1], "OU", False),
                  ([255, 257], "matern32", True)]


@pytest.mark.parametrize("prec,tol", [("fp64", 1e-9), ("fp32", FP32_RTOL)])
def test_extreme_shapes(gp, oracle, prec, tol):
    """The maximum band count (8), one- and two-observation bands, totals on either side of a tile edge (127 / 128 /
    129), N = 1: device == oracle; a ninth band and a one-observation band with marginalised offsets (var undefined
    in the reference, marginaliseb.jl:91) are argument errors."""
    rng = np.random.default_rng(0)
    for Nl, kern, mb in EXTREME_SHAPES:
        L, M = len(Nl), 5
        t = [rng.uniform(0, 30, n) for n in Nl]
        y = [rng.standard_normal(n) * 2 + 5 * l for l, n in enumerate(Nl)]
        s = [rng.uniform(0.3, 1.0, n) for n in Nl]
        delays = np.concatenate([np.zeros((M, 1)), rng.uniform(0, 5, (M, L - 1))], 1)
        alpha, rho = rng.uniform(0.5, 2.0, (M, L)), rng.uniform(1.0, 6.0, M)
        with gp.Objective(t, y, s, kern, marginalise_b=mb, precision=prec) as obj:
            ll, info = obj.loglik_batch(delays, alpha, rho)
        ref, rinfo = oracle.loglik_batch(kern, t, y, s, delays, alpha, rho, mb)
        assert (rinfo == 0).all() and np.array_equal(info, rinfo), (Nl, kern, info, rinfo)
        assert np.max(np.abs(ll - ref) / np.abs(ref)) <= tol, (Nl, kern, mb, prec)
    with pytest.raises(gp.GpccError):
        gp.Objective([np.arange(3.0)] * 9, [np.arange(3.0)] * 9, [np.ones(3)] * 9, "OU")
    with pytest.raises(gp.GpccError):
        gp.Objective([np.arange(1.0)], [np.arange(1.0)], [np.ones(1)], "OU", marginalise_b=True)


def test_extreme_scales_of_the_covariance(gp, oracle):
    """Amplitudes far outside any sensible fit (K scaled by 1e-60 .. 1e+60: L_jj from 1e-30 to 1e+30): the diagonal kernel's
    log-determinant (product of pivot reciprocals per 16-block, mantissa/exponent split) must neither overflow nor underflow."""
    from gpcc_amd import synthetic
    t, y, s, _ = synthetic.simulate_lightcurves([150, 130], seed=3)
    for scale in (1e-30, 1e-12, 1.0, 1e12, 1e30):
        ys = [a * scale for a in y]
        ss = [a * scale for a in s]
        alpha = np.array([[1.3 * scale, 0.9 * scale]])
        with gp.Objective(t, ys, ss, gp.matern32) as obj:
            ll, info = obj.loglik_batch([[0.0, 2.0]], alpha, [3.5])
        ref, rinfo = oracle.loglik_batch("matern32", t, ys, ss, [[0.0, 2.0]], alpha, [3.5], True)
        assert info[0] == 0 and rinfo[0] == 0 and np.isfinite(ll[0])
        assert abs(ll[0] - ref[0]) <= 1e-9 * abs(ref[0]), (scale, ll[0], ref[0])


@pytest.mark.parametrize("prec,mb,tol", [("fp32", True, FP32_RTOL), ("fp64", False, 1e-8), ("fp32", False, FP32_RTOL)])
def test_native_grid_fit_other_handles(gp, oracle, prec, mb, tol):
    """gpcc_grid_loglik over an fp32 handle and over the fixed-offset objective (gpccfixdelay.jl:131-139): the returned
    value is the objective at the returned hyper-parameters, and the fit improved on its best random start."""
    from gpcc_amd import fit, synthetic
    t, y, s, _ = synthetic.simulate_lightcurves([80, 70], seed=4, span=25.0)
    cand = np.stack([np.zeros(5), np.array([0.0, 1.0, 2.0, 3.0, 7.5])], 1)
    with gp.Objective(t, y, s, gp.matern52, marginalise_b=mb, precision=prec) as obj:
        res = fit.gpcc_grid(t, y, s, kernel=gp.matern52, candidatedelays=cand, iterations=60, rhomax=40.0, objective=obj)
        start = fit.gpcc_grid(t, y, s, kernel=gp.matern52, candidatedelays=cand, iterations=0, rhomax=40.0, objective=obj)
    ref, rinfo = oracle.loglik_batch("matern52", t, y, s, cand, res.alpha, res.rho, mb, nthreads=8)
    assert (rinfo == 0).all() and np.max(np.abs(res.loglikel - ref) / np.abs(ref)) <= tol
    assert np.all(res.loglikel >= start.loglikel - tol * np.abs(start.loglikel)) and np.any(res.loglikel > start.loglikel)
    assert np.all(res.alpha > 0) and np.all((res.rho > 0.1) & (res.rho < 40.0))


def test_c_abi_from_plain_c(gp, oracle, tmp_path):
    """The boundary without Python or torch in the process: tests/abi/abi_smoke.c (plain C, gcc) links libgpcc_hip.so,
    evaluates a small batch, normalises it, runs the native per-delay fit and provokes two errors; the printed numbers are
    checked here against the oracle on the same closed-form light curves."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from gpcc_amd import build as gbuild
    libdir = os.path.dirname(gbuild.LIB_PATH)
    exe = str(tmp_path / "abi_smoke")
    cc = subprocess.run(["gcc", "-O1", "-std=c99", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "abi", "abi_smoke.c"),
                         "-o", exe, "-L", libdir, "-lgpcc_hip", "-lm", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"],
                        capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    lines = [ln.split() for ln in run.stdout.splitlines()]
    assert lines[-1] == ["done"]
    # the same light curves
    N0, N1 = 150, 131
    i = np.arange(N0 + N1)
    band = (i >= N0).astype(int)
    j = np.where(band == 1, i - N0, i).astype(float)
    tt = np.fmod(7.3 * j + 0.37 * j * j, 50.0) + 0.01 * j
    yy = np.where(band == 1, 15.0, 6.0) + np.where(band == 1, 1.5, 1.0) * np.sin(0.35 * (tt - np.where(band == 1, 2.0, 0.0))) \
        + 0.3 * np.cos(12.9898 * i)
    sg = 0.4 + 0.1 * np.abs(np.sin(1.7 * i))
    t, y, s = [tt[:N0], tt[N0:]], [yy[:N0], yy[N0:]], [sg[:N0], sg[N0:]]
    m = np.arange(5)
    delays = np.stack([np.zeros(5), 1.0 * m], 1)
    alpha = np.stack([1.0 + 0.1 * m, 1.5 - 0.1 * m], 1)
    rho = 3.0 + 0.5 * m
    ref, rinfo = oracle.loglik_batch("matern32", t, y, s, delays, alpha, rho, True)
    got = np.array([float(ln[2]) for ln in lines if ln[0] == "loglik"])
    ginfo = np.array([int(ln[3]) for ln in lines if ln[0] == "loglik"])
    gprob = np.array([float(ln[4]) for ln in lines if ln[0] == "loglik"])
    assert (rinfo == 0).all() and (ginfo == 0).all() and np.max(np.abs(got - ref) / np.abs(ref)) <= 1e-9
    np.testing.assert_allclose(gprob, oracle.probabilities(ref), rtol=1e-6, atol=1e-300)
    fits = [ln for ln in lines if ln[0] == "fit"]
    assert len(fits) == 3
    fl = np.array([float(f[2]) for f in fits])
    fa = np.array([[float(f[3]), float(f[4])] for f in fits])
    fr = np.array([float(f[5]) for f in fits])
    assert all(int(f[6]) == 0 and 0 <= int(f[7]) <= 20 for f in fits)
    fref, finfo = oracle.loglik_batch("matern32", t, y, s, np.array([[0.0, 0.0], [0.0, 2.0], [0.0, 4.0]]), fa, fr, True)
    assert (finfo == 0).all() and np.max(np.abs(fl - fref) / np.abs(fref)) <= 1e-8     # value == objective at the optimum
    stats = [ln for ln in lines if ln[0] == "stats"][0]
    assert int(stats[1]) > 3 * 3 and int(stats[2]) < int(stats[1])
    assert [ln for ln in lines if ln[0] == "badrho"][0][1:] == ["-2", "1"]
    assert [ln for ln in lines if ln[0] == "nullcall"][0][1] == "-1"


def test_handle_lifetimes_leave_no_device_memory_behind(gp):
    """create / use (every entry that allocates) / destroy, 24 times over, fp64 and fp32, including the native fit:
    free device memory returns to where it was (hipMemGetInfo through torch)."""
    import torch

    from gpcc_amd import synthetic
    t, y, s, _ = synthetic.simulate_lightcurves([300, 280], seed=1)
    alpha, rho = synthetic.default_hyperparameters(y)
    M = 12
    d = np.stack([np.zeros(M), np.linspace(0, 10, M)], 1)
    free0 = None
    for it in range(24):
        with gp.Objective(t, y, s, "matern32", precision="fp32" if it % 2 else "fp64", slots_per_stream=16) as obj:
            if it % 4 == 1:
                obj.set_option("fp32_chain", 0)   # fp32 tiles for the few-evaluation calls too (it % 4 == 3: they go to the fp64 twin and its persistent launch)
            obj.loglik_batch(d, np.tile(alpha, (M, 1)), np.full(M, rho))
            obj.loglik_batch(d[:3], np.tile(alpha, (3, 1)), np.full(3, rho))      # small group: spread map
            if it % 4 == 0:
                obj.predict([0.0, 2.0], alpha, rho, [np.linspace(0, 50, 40)] * 2)
                obj.posterior_offsets([0.0, 2.0], alpha, rho)
                obj.model_matrix([0.0, 2.0], alpha, rho)
                obj.grid_loglik(d[:4], 3, rhomax=30.0)
            if it % 2:   # fp32: an ill-conditioned evaluation makes the accuracy guard create (and the handle later free) its fp64 workspace
                obj.loglik_batch(d[:2], np.tile(alpha * 60.0, (2, 1)), np.full(2, rho))
                if it % 4 == 1:
                    assert obj.get_option("fp32_guard_count") > 0
                else:
                    assert obj.get_option("fp32_chain_count") >= 2 and obj.get_option("fp32_guard_count") == 0
        if it % 6 == 0:
            with gp.Objective(t, y, s, "matern32", devices=[0, 0]) as multi:          # a multi-device handle: sub-handles, gather buffers
                multi.loglik_batch(d, np.tile(alpha, (M, 1)), np.full(M, rho))
        gp.getprobabilities(np.zeros(10))
        gp.delayedCovariance("OU", [1.0, 1.0], [0.0, 1.0], 2.0, [t[0][:10], t[1][:10]])
        torch.cuda.synchronize()
        free, _ = torch.cuda.mem_get_info()
        if it == 3:
            free0 = free
    assert abs(free - free0) < 64 * 2**20, (free0, free)


@pytest.mark.parametrize("Nl,prec,tol", [([300, 280], "fp64", 1e-11), ([129], "fp64", 1e-11), ([520, 500, 490], "fp64", 1e-11),
                                         ([384, 300], "fp32", 1e-5)])
def test_fused_solve_path_agrees_with_three_kernel_path(gp, oracle, Nl, prec, tol):
    """The default left-looking path (gpcc_syrk_diag + gpcc_update_solve: panel solve inside the update) against
    the round-1 kernels (update / diagonal / solve as three launches, `fused_solve` = 0) and against the oracle: same
    arithmetic in another summation order.  Group sizes above and below 8 (the job map differs), a ragged last tile, an
    invalid and a non-positive-definite evaluation in the batch."""
    from gpcc_amd import synthetic
    t, y, s, _ = synthetic.simulate_lightcurves(Nl, seed=13)
    alpha, rho = synthetic.default_hyperparameters(y)
    L = len(Nl)
    for M in (3, 41):
        rng = np.random.default_rng(M)
        delays = np.concatenate([np.zeros((M, 1)), rng.random((M, L - 1)) * 12], 1)
        alphas = np.tile(alpha, (M, 1)) * (0.5 + rng.random((M, L)))
        rhos = rho * (0.5 + rng.random(M))
        alphas[1, 0] = 0.0                       # argument error: info -1
        with gp.Objective(t, y, s, gp.matern52, precision=prec, slots_per_stream=16) as obj:
            obj.set_option("right_looking_max", 0)    # left-looking also for the group of 3
            obj.set_option("shared_prefix", 0)
            obj.set_option("fused_solve_min", 1)      # (by default only groups of >= 112 evaluations take the fused path)
            assert obj.get_option("fused_solve") == 1
            a, ia = obj.loglik_batch(delays, alphas, rhos)
            a2, ia2 = obj.loglik_batch(delays, alphas, rhos)
            obj.set_option("fused_solve", 0)
            b, ib = obj.loglik_batch(delays, alphas, rhos)
        assert np.array_equal(a, a2, equal_nan=True) and np.array_equal(ia, ia2)      # repeatable
        assert np.array_equal(ia, ib) and ia[1] == -1 and (np.delete(ia, 1) == 0).all()
        ok = ia == 0
        assert _rel(a[ok], b[ok]) <= tol
        ref, rinfo = oracle.loglik_batch("matern52", t, y, s, delays, alphas, rhos, True, nthreads=8)
        assert np.array_equal(rinfo == 0, ok)
        assert _rel(a[ok], ref[ok]) <= (LL_RTOL if prec == "fp64" else FP32_RTOL)


def test_workspace_shrinks_when_the_memory_has_gone(gp):
    """The group size is chosen from the memory that is free when the handle is created and the workspace is allocated on the first
    batch; if another tenant of the GPU has taken the memory in between, the handle runs smaller groups (one stream, then half the
    slots, ...) instead of failing -- same results."""
    import torch
    from gpcc_amd import synthetic
    t, y, s, _ = synthetic.simulate_lightcurves([1024, 1024], seed=5)
    alpha, rho = synthetic.default_hyperparameters(y)
    M = 40
    delays = np.stack([np.zeros(M), np.linspace(0, 15, M)], 1)
    alphas, rhos = np.tile(alpha, (M, 1)), np.full(M, rho)
    with gp.Objective(t, y, s, gp.matern32) as obj:
        for k, v in (("hybrid_tail", 0), ("split_min", 0)):      # (the bits of an evaluation then do not depend on its group)
            obj.set_option(k, v)
        ref, rinfo = obj.loglik_batch(delays, alphas, rhos)
        want = obj.get_option("streams") * obj.get_option("slots_per_stream") * obj.get_option("bytes_per_slot")
        assert obj.get_option("workspace_streams") == obj.get_option("streams")          # nothing shrank here
        assert obj.get_option("workspace_slots") == obj.get_option("slots_per_stream")
    assert (rinfo == 0).all() and want > 3 << 30
    with gp.Objective(t, y, s, gp.matern32) as obj:              # created while the memory is free ...
        for k, v in (("hybrid_tail", 0), ("split_min", 0)):
            obj.set_option(k, v)
        torch.cuda.init()
        free_b, _ = torch.cuda.mem_get_info(0)
        try:
            hog = torch.empty(free_b - (2 << 30), dtype=torch.uint8, device="cuda:0")   # ... then all but 2 GiB of it goes away
        except RuntimeError as e:                                                       # (not this test's subject)
            pytest.skip("could not occupy the GPU's memory: %s" % str(e)[:80])
        try:
            opts = (obj.get_option("streams"), obj.get_option("slots_per_stream"))
            ll, info = obj.loglik_batch(delays, alphas, rhos)
            got = obj.get_option("workspace_streams") * obj.get_option("workspace_slots") * obj.get_option("bytes_per_slot")
            assert (obj.get_option("streams"), obj.get_option("slots_per_stream")) == opts   # the caller's options are NOT rewritten ...
            assert b"workspace runs" in _capi_last_error(obj)                                 # ... and the shrink is reported
        finally:
            del hog
            torch.cuda.empty_cache()
    assert got <= want // 2                                      # (it did shrink: one stream and/or fewer slots)
    assert np.array_equal(info, rinfo) and np.array_equal(ll, ref)


@pytest.mark.parametrize("Nl,prec,M,cs", [([700, 650], "fp64", 37, 256), ([1024, 1030], "fp64", 75, 32), ([400, 380], "fp64", 130, 256),
                                           ([900, 800], "fp32", 30, 256)])
def test_split_groups_return_the_same_bits(gp, Nl, prec, M, cs):
    """Option "split_min": a group runs as two halves on two streams (same slots, each half its own launches).  With the
    size-dependent choices pinned (plain left-looking three-kernel path), every evaluation's arithmetic is the same, so the
    results are bitwise those of the unsplit group -- odd sizes, a remainder group, an argument error and an fp32 handle
    (refinement + guard) included; with the defaults back on, they agree to rounding."""
    from gpcc_amd import synthetic
    t, y, s, _ = synthetic.simulate_lightcurves(Nl, seed=11)
    alpha, rho = synthetic.default_hyperparameters(y)
    L = len(Nl)
    rng = np.random.default_rng(2)
    delays = np.concatenate([np.zeros((M, 1)), rng.random((M, L - 1)) * 12], 1)
    alphas = np.tile(alpha, (M, 1)) * (0.7 + 0.6 * rng.random((M, L)))
    rhos = np.full(M, rho) * (0.5 + rng.random(M))
    alphas[M - 2, 0] = -1.0
    with gp.Objective(t, y, s, gp.matern32, precision=prec, streams=2, slots_per_stream=cs) as obj:
        for k, v in (("shared_prefix", 0), ("hybrid_tail", 0), ("right_looking_max", 0), ("fused_solve_min", 10 ** 6), ("split_min", 0)):
            obj.set_option(k, v)
        ref, rinfo = obj.loglik_batch(delays, alphas, rhos)
        for k, v in (("split_min", 2), ("split_max", 10 ** 6), ("split_nt_min", 1)):
            obj.set_option(k, v)
        ll, info = obj.loglik_batch(delays, alphas, rhos)
        assert np.array_equal(info, rinfo) and rinfo[M - 2] == -1 and (np.delete(rinfo, M - 2) == 0).all()
        assert np.array_equal(ll, ref, equal_nan=True)
        for k, v in (("hybrid_tail", 1), ("right_looking_max", 12), ("fused_solve_min", 112)):   # halves pick their paths by their own size
            obj.set_option(k, v)
        l2, i2 = obj.loglik_batch(delays, alphas, rhos)
        ok = rinfo == 0
        assert np.array_equal(i2, rinfo) and _rel(l2[ok], ref[ok]) <= (1e-11 if prec == "fp64" else 1e-5)


@pytest.mark.parametrize("Nl,prec,M,tol", [([700, 650], "fp64", 20, 1e-11), ([1024, 1024], "fp64", 40, 1e-11), ([520, 500, 490], "fp64", 13, 1e-11),
                                            ([900, 800], "fp32", 30, 1e-5)])
def test_right_looking_tail_agrees_with_plain_left_looking(gp, oracle, Nl, prec, M, tol):
    """Groups of 13-111 evaluations run left-looking and finish right-looking (one catch-up launch, then right-looking steps
    on trailing matrices that fit the Infinity Cache): same factorisation, another summation order -- against the plain
    left-looking path, against the oracle, with a non-positive-definite evaluation and an argument error in the group."""
    from gpcc_amd import synthetic
    t, y, s, _ = synthetic.simulate_lightcurves(Nl, seed=9)
    alpha, rho = synthetic.default_hyperparameters(y)
    L = len(Nl)
    rng = np.random.default_rng(1)
    delays = np.concatenate([np.zeros((M, 1)), rng.random((M, L - 1)) * 12], 1)
    alphas = np.tile(alpha, (M, 1)) * (0.7 + 0.6 * rng.random((M, L)))
    rhos = np.full(M, rho) * (0.5 + rng.random(M))
    alphas[3, 0] = -1.0
    with gp.Objective(t, y, s, gp.matern32, precision=prec) as obj:
        obj.set_option("shared_prefix", 0)
        obj.set_option("hybrid_tail", 0)
        ref, rinfo = obj.loglik_batch(delays, alphas, rhos)
        obj.set_option("hybrid_tail", 1)
        ll, info = obj.loglik_batch(delays, alphas, rhos)
        for budget in (40, 4000):            # an early and a late switch to the right-looking form
            obj.set_option("hybrid_mall_mb", budget)
            l2, i2 = obj.loglik_batch(delays, alphas, rhos)
            assert np.array_equal(i2, rinfo) and _rel(l2[rinfo == 0], ref[rinfo == 0]) <= tol, budget
    assert rinfo[3] == -1 and (np.delete(rinfo, 3) == 0).all() and np.array_equal(info, rinfo)
    ok = rinfo == 0
    assert _rel(ll[ok], ref[ok]) <= tol
    if sum(Nl) <= 2100 and prec == "fp64":
        o, oi = oracle.loglik_batch("matern32", t, y, s, delays[ok][:6], alphas[ok][:6], rhos[ok][:6], True, nthreads=8)
        assert (oi == 0).all() and _rel(ll[ok][:6], o) <= LL_RTOL


def test_constant_band_zero_prior_variance(gp, oracle):
    """A band of constant fluxes has Sigma_b = 100 var(y_l) = 0: K = delayedCovariance + Sobs (+ B of the other bands) is
    still positive definite and the reference returns a finite log-likelihood.  The fp32 path's capacitance system
    (Woodbury) must drop that column instead of dividing by zero; postb has no meaning there (inv(Sigma_b))."""
    from gpcc_amd import synthetic
    t, y, s, _ = synthetic.simulate_lightcurves([150, 140, 90], seed=9)
    y[1] = np.full_like(y[1], 4.25)
    delays, alpha, rho = [[0.0, 1.0, 3.0], [0.0, 4.0, 2.0]], [[1.1, 0.7, 1.9], [0.8, 1.0, 1.2]], [2.5, 4.0]
    ref, rinfo = oracle.loglik_batch("matern32", t, y, s, delays, alpha, rho, True)
    assert (rinfo == 0).all() and np.all(np.isfinite(ref))
    for prec, tol in (("fp64", LL_RTOL), ("fp32", FP32_RTOL)):
        with gp.Objective(t, y, s, gp.matern32, precision=prec) as obj:
            ll, info = obj.loglik_batch(delays, alpha, rho)
            assert (info == 0).all() and _rel(ll, ref) <= tol, (prec, ll, ref)
            assert obj.constants()[1][1] == 0.0
            with pytest.raises(gp.GpccError):
                obj.posterior_offsets(delays[0], alpha[0], rho[0])


# ---- multi-device handles behind the C ABI (gpcc_create_multi) ---------------------------------
def test_multi_device_handle_matches_single_device(gp, oracle):
    """device_ids = [0]: bitwise equal to the single-device handle.  device_ids = [0, 0] (the one-GPU rehearsal of the
    sharded path: two sub-handles, two worker threads, contiguous blocks): RCCL refuses a communicator with a
    repeated device, so this rehearsal gathers through host memory (gather_mode = GPCC_GATHER_HOST = 2) -- the
    ncclAllGather branch needs >= 2 GPUs and is not exercised on this box."""
    from gpcc_amd import synthetic
    t, y, s, _ = synthetic.simulate_lightcurves([300, 280], seed=8)
    alpha, rho = synthetic.default_hyperparameters(y)
    M = 37
    delays = np.stack([np.zeros(M), np.linspace(0.0, 10.0, M)], 1)
    alphas, rhos = np.tile(alpha, (M, 1)), np.full(M, rho)
    alphas[5, 1] = -1.0                      # an invalid point inside the first block
    rhos[30] = 0.0                           # and one inside the second
    # left- and right-looking updates round differently and the choice follows the group size, which sharding
    # changes: pin the left-looking form on both sides for the bitwise comparison; likewise the shared-prefix mode
    # (a block of a sharded batch may be shareable where the whole batch is not; its kernels sum in another order)
    with gp.Objective(t, y, s, gp.matern32) as single:
        single.set_option("right_looking_max", 0)
        single.set_option("hybrid_tail", 0)     # (the step where a group turns right-looking follows its size, which sharding changes)
        single.set_option("shared_prefix", 0)
        ref, rinfo = single.loglik_batch(delays, alphas, rhos)
        fit_ref = single.grid_loglik(delays[:6], 8, seed=3)
    assert rinfo[5] == -1 and rinfo[30] == -2 and (np.delete(rinfo, [5, 30]) == 0).all()
    for devs, mode in (([0], 2), ([0, 0], 2), ([0, 0, 0], 2)):
        with gp.Objective(t, y, s, gp.matern32, devices=devs) as multi:
            assert multi.get_option("n_devices") == len(devs) and multi.get_option("gather_mode") == mode
            multi.set_option("right_looking_max", 0)       # applies to every device
            multi.set_option("hybrid_tail", 0)
            multi.set_option("shared_prefix", 0)
            ll, info = multi.loglik_batch(delays, alphas, rhos)
            assert np.array_equal(ll, ref, equal_nan=True) and np.array_equal(info, rinfo), devs
            blk = -(-M // len(devs))
            for which in range(len(devs)):    # every device holds the whole gathered vector
                gl, gi = multi.gathered(which)
                assert gl.shape == (len(devs), blk)
                assert np.array_equal(gl.ravel()[:M], ref, equal_nan=True) and np.array_equal(gi.ravel()[:M], rinfo)
                assert np.isnan(gl.ravel()[M:]).all()
            comp_ms, gather_ms, total_ms = multi.multi_stats()    # of that batch: per-device share, gather phase, whole call
            assert comp_ms.shape == (len(devs),) and np.all(comp_ms > 0) and gather_ms >= 0 and total_ms >= comp_ms.max() * 0.5
            few, finfo = multi.loglik_batch(delays[:2], alphas[:2], rhos[:2])       # fewer evaluations than devices
            assert np.array_equal(few, ref[:2]) and (finfo == 0).all()
            one = multi([alpha[0], alpha[1]], rho, delays[3])                        # objective(alpha, rho): M = 1
            assert one == ref[3]
            fit = multi.grid_loglik(delays[:6], 8, seed=3)                           # the sharded per-delay fit
            for a, b in zip(fit[:5], fit_ref[:5]):
                assert np.array_equal(a, b)
            mu, Sig = multi.predict(delays[3], alpha, rho, [np.array([1.0, 2.0]), np.array([3.0])])   # on device_ids[0]
            assert np.all(np.isfinite(mu)) and Sig.shape == (3, 3)
    with pytest.raises(gp.GpccError):
        gp.Objective(t, y, s, gp.matern32, devices=[0, 99])


def test_two_processes_share_one_gpu(gp, oracle, tmp_path, kernel_family):
    """the reference parallelises with pmap WORKER PROCESSES, so the library must be safe for several processes
    sharing one GPU (no process-global state beyond the handle).  Two fresh child processes, each with its own handle on
    device 0, evaluate at the same time (file barrier) on the small-N path and on the tile path; both are checked against
    the oracle."""
    import os
    import subprocess
    import sys
    from gpcc_amd import synthetic
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, GPCC_SMALL_N="0" if kernel_family == "tile" else "1")
    procs = [subprocess.Popen([sys.executable, os.path.join(here, "_shared_gpu_worker.py"), str(r), str(tmp_path)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in (0, 1)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    for rank in (0, 1):
        res = np.load(os.path.join(tmp_path, "result_%d.npz" % rank))
        for tag, Nl in (("small", [60, 50]), ("tiles", [330, 310])):
            t, y, s, _ = synthetic.simulate_lightcurves(Nl, seed=11 + rank, span=20.0)
            alpha, rho = synthetic.default_hyperparameters(y)
            d = res[tag + "_delays"]
            ref, rinfo = oracle.loglik_batch("matern32", t, y, s, d, np.tile(alpha, (len(d), 1)), np.full(len(d), rho), True, nthreads=8)
            assert (rinfo == 0).all() and _rel(res[tag], ref) <= LL_RTOL, (rank, tag)


def _gpu_count():
    import torch
    return torch.cuda.device_count()


def test_multi_gpu_rccl_gather_and_foreign_current_device(gp, oracle):
    """Needs >= 2 GPUs (skipped on the one-GPU box): distinct devices take the RCCL route (ncclCommInitAll + grouped
    ncclAllGather); every device must hold the whole rank-major vector; results equal the single-device handle's; and the
    single-matrix utilities work from a thread whose current device is NOT the handle's (the DeviceGuard of every entry)."""
    import threading
    import torch
    ndev = _gpu_count()
    if ndev < 2:
        pytest.skip("needs at least 2 GPUs")
    from gpcc_amd import synthetic
    t, y, s, _ = synthetic.simulate_lightcurves([300, 280], seed=8)
    alpha, rho = synthetic.default_hyperparameters(y)
    M = 101
    delays = np.stack([np.zeros(M), np.linspace(0.0, 10.0, M)], 1)
    alphas, rhos = np.tile(alpha, (M, 1)), np.full(M, rho)
    with gp.Objective(t, y, s, gp.matern32) as single:
        single.set_option("right_looking_max", 0)
        single.set_option("hybrid_tail", 0)     # (the step where a group turns right-looking follows its size, which sharding changes)
        single.set_option("shared_prefix", 0)
        ref, rinfo = single.loglik_batch(delays, alphas, rhos)
        K_ref = single.model_matrix(delays[3], alpha, rho)
    devs = list(range(min(ndev, 4)))
    with gp.Objective(t, y, s, gp.matern32, devices=devs) as multi:
        assert multi.get_option("gather_mode") == 1          # GPCC_GATHER_RCCL
        multi.set_option("right_looking_max", 0)
        multi.set_option("hybrid_tail", 0)
        multi.set_option("shared_prefix", 0)
        ll, info = multi.loglik_batch(delays, alphas, rhos)
        assert np.array_equal(ll, ref) and np.array_equal(info, rinfo)
        for which in range(len(devs)):
            gl, gi = multi.gathered(which)
            assert np.array_equal(gl.ravel()[:M], ref) and np.array_equal(gi.ravel()[:M], rinfo)
        comp_ms, gather_ms, total_ms = multi.multi_stats()
        assert np.all(comp_ms > 0)
    # a handle on the LAST device, driven from a thread whose current device is 0
    out = {}
    with gp.Objective(t, y, s, gp.matern32, device=ndev - 1) as far:
        def work():
            torch.cuda.set_device(0)
            out["K"] = far.model_matrix(delays[3], alpha, rho)
            out["mu"], out["Sig"] = far.predict(delays[3], alpha, rho, [np.array([1.0, 2.0]), np.array([3.0])])
            out["postb"] = far.posterior_offsets(delays[3], alpha, rho)
            out["dev"] = torch.cuda.current_device()
        th = threading.Thread(target=work)
        th.start()
        th.join()
    assert out["dev"] == 0                                   # the caller's current device is restored
    np.testing.assert_allclose(out["K"], K_ref, rtol=1e-13)
    assert np.all(np.isfinite(out["mu"])) and out["Sig"].shape == (3, 3) and np.all(np.isfinite(out["postb"][0]))


def test_multi_device_rccl_route_with_one_rank(gp, monkeypatch):
    """The RCCL branch of the multi-device handle (ncclCommInitAll, ncclAllGather inside a group, ncclCommDestroy) needs
    distinct devices; on a one-GPU box it can only run as a ONE-rank communicator (GPCC_MULTI_FORCE_RCCL=1).  That proves the
    library's RCCL calls load, link and complete on the hardware -- not that a multi-rank gather is right."""
    from gpcc_amd import synthetic
    t, y, s, _ = synthetic.simulate_lightcurves([200, 180], seed=4)
    alpha, rho = synthetic.default_hyperparameters(y)
    M = 9
    delays = np.stack([np.zeros(M), np.linspace(0.0, 8.0, M)], 1)
    with gp.Objective(t, y, s, gp.OU) as single:
        ref, rinfo = single.loglik_batch(delays, np.tile(alpha, (M, 1)), np.full(M, rho))
    monkeypatch.setenv("GPCC_MULTI_FORCE_RCCL", "1")
    with gp.Objective(t, y, s, gp.OU, devices=[0]) as multi:
        assert multi.get_option("gather_mode") == 1          # GPCC_GATHER_RCCL
        for _ in range(2):
            ll, info = multi.loglik_batch(delays, np.tile(alpha, (M, 1)), np.full(M, rho))
            assert np.array_equal(ll, ref) and np.array_equal(info, rinfo)
        gl, gi = multi.gathered(0)
        assert np.array_equal(gl.ravel()[:M], ref)


def test_torch_imported_after_the_library_shares_one_hip_runtime():
    """Round 1 saw "No HIP GPUs are available" when torch initialised the GPU AFTER libgpcc_hip.so had: two HIP/HSA
    runtimes in one process (gpcc_amd/_capi.py::_share_torch_rocm_runtime explains the loader rule).  A fresh process
    uses the library first, imports torch afterwards, runs a torch GPU op, uses the library again, and must have
    exactly one libamdhip64 / libhsa-runtime64 mapped."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import sys
sys.path.insert(0, %r)
assert "torch" not in sys.modules
import numpy as np
import gpcc_amd
from gpcc_amd import _capi, synthetic
t, y, s, _ = synthetic.simulate_lightcurves([70, 60], seed=2)
a, r = synthetic.default_hyperparameters(y)
with gpcc_amd.Objective(t, y, s, "OU") as obj:
    first = obj.loglik_batch([[0.0, 1.0]], [a], [r])[0][0]
import torch
x = torch.arange(8, dtype=torch.float64, device="cuda:0")
assert float((x * x).sum().item()) == 140.0
with gpcc_amd.Objective(t, y, s, "OU") as obj:
    d = torch.tensor([[0.0, 1.0]], dtype=torch.float64, device="cuda:0")
    out, info = obj.loglik_batch_device(d, torch.tensor([list(a)], dtype=torch.float64, device="cuda:0"),
                                        torch.tensor([r], dtype=torch.float64, device="cuda:0"))
    torch.cuda.synchronize()
    assert float(out[0].item()) == first and int(info[0].item()) == 0
libs = _capi.loaded_rocm_libraries()
for stem in ("libamdhip64", "libhsa-runtime64"):
    n = [p for p in libs if stem in p]
    assert len(n) == 1, libs
print("ok", libs)
""" % root
    run = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0 and "ok" in run.stdout, run.stdout[-2000:] + run.stderr[-4000:]
