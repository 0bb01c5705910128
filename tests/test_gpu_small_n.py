"""GPU parity tests of the small-N family (gpcc_small.hip.h: one launch per batch, one wave per evaluation, the matrix in
registers) -- the sizes of the reference's own documentation, N = 110 and N = 150 (README.md:156-287, simulatedata.jl:119).
All through the C ABI, against the CPU oracle, the golden fixtures, and the tile kernels of rounds 1-2."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LL_RTOL = 1e-8          # asserted; north_star asks for 1e-6 in fp64
KNAMES = ["OU", "rbf", "matern32", "matern52"]


@pytest.fixture(scope="module")
def gp():
    import torch
    torch.cuda.init()
    import gpcc_amd
    return gpcc_amd


def _problem(rng, Nl, sigma_lo=0.3):
    t = [rng.uniform(0, 25, n) for n in Nl]
    y = [rng.standard_normal(n) * 2 + 5 * l + np.sin(0.3 * t[l]) for l, n in enumerate(Nl)]
    s = [rng.uniform(sigma_lo, 1.0, n) for n in Nl]
    return t, y, s


def _params(rng, M, L):
    delays = np.concatenate([np.zeros((M, 1)), rng.uniform(0, 6, (M, L - 1))], 1)
    return delays, rng.uniform(0.4, 3.0, (M, L)), rng.uniform(0.5, 8.0, M)


def test_small_path_is_the_default_up_to_383_points(gp):
    rng = np.random.default_rng(1)
    for n, active in ((1, 1), (110, 1), (150, 1), (191, 1), (192, 1), (300, 1), (383, 1), (384, 0), (500, 0)):
        t, y, s = _problem(rng, [n])
        with gp.Objective(t, y, s, "OU", marginalise_b=False) as obj:
            assert obj.get_option("small_n_max") == 383
            assert obj.get_option("small_n_active") == active, n
            obj.loglik_batch(np.zeros((3, 1)), np.ones((3, 1)), np.full(3, 2.0))
            assert obj.get_option("small_n_count") == (3 if active else 0)
            obj.set_option("small_n", 0)
            assert obj.get_option("small_n_active") == 0


def test_every_size_1_to_191_vs_oracle_and_tile_path(gp, oracle):
    """All totals N = 1 .. 191 (every 16-block count, every position of the last real row inside its block), 1-3 bands,
    all kernels, both b-modes: small-N kernel == oracle (1e-9) and == the tile kernels (1e-11)."""
    rng = np.random.default_rng(7)
    worst_o = worst_t = 0.0
    for N in range(1, 192):
        L = 1 + (N % 3) if N >= 6 else 1
        mb = bool(N % 2) and N >= 2 * L
        cuts = np.sort(rng.choice(np.arange(2, N - 1), L - 1, replace=False)) if L > 1 else np.array([], dtype=int)
        Nl = [int(x) for x in np.diff(np.concatenate([[0], cuts, [N]]))]
        if mb and min(Nl) < 2:
            mb = False
        kname = KNAMES[N % 4]
        t, y, s = _problem(rng, Nl)
        delays, alpha, rho = _params(rng, 4, L)
        ref, rinfo = oracle.loglik_batch(kname, t, y, s, delays, alpha, rho, mb)
        with gp.Objective(t, y, s, kname, marginalise_b=mb) as obj:
            ll, info = obj.loglik_batch(delays, alpha, rho)
            assert obj.get_option("small_n_count") == 4
            obj.set_option("small_n", 0)
            ll_t, info_t = obj.loglik_batch(delays, alpha, rho)
        assert (rinfo == 0).all() and (info == 0).all() and (info_t == 0).all(), (N, Nl, kname, info, rinfo)
        worst_o = max(worst_o, np.max(np.abs(ll - ref) / np.abs(ref)))
        worst_t = max(worst_t, np.max(np.abs(ll - ll_t) / np.abs(ll_t)))
        assert worst_o <= 1e-9 and worst_t <= 1e-11, (N, Nl, kname, mb, worst_o, worst_t)
    print("N = 1..191: worst vs oracle %.2e, worst vs tile kernels %.2e" % (worst_o, worst_t))


def test_four_waves_per_evaluation_sizes_192_to_383_and_small_batches(gp, oracle):
    """gpcc_smallw_eval: (a) N = 192 .. 383 (every instantiated block count, sizes on and off a 16-block edge) against the oracle
    and the tile kernels; (b) N <= 191 in batches small enough to take the four-wave kernel against the one-wave kernel (same
    factorisation; sum log L_ii is combined across waves in another order: <= 1e-13)."""
    rng = np.random.default_rng(17)
    worst_o = worst_t = 0.0
    for N in (192, 199, 207, 208, 223, 224, 239, 255, 256, 271, 287, 300, 303, 319, 320, 335, 351, 352, 367, 382, 383):
        L = 1 + N % 3
        cuts = np.sort(rng.choice(np.arange(2, N - 1), L - 1, replace=False)) if L > 1 else np.array([], dtype=int)
        Nl = [int(x) for x in np.diff(np.concatenate([[0], cuts, [N]]))]
        mb = bool(N % 2) and min(Nl) >= 2
        kname = KNAMES[N % 4]
        t, y, s = _problem(rng, Nl)
        delays, alpha, rho = _params(rng, 5, L)
        alpha[4, 0] = -1.0
        ref, rinfo = oracle.loglik_batch(kname, t, y, s, delays, alpha, rho, mb, nthreads=5)
        with gp.Objective(t, y, s, kname, marginalise_b=mb) as obj:
            assert obj.get_option("small_n_active") == 1
            ll, info = obj.loglik_batch(delays, alpha, rho)
            obj.set_option("small_n", 0)
            ll_t, info_t = obj.loglik_batch(delays, alpha, rho)
        assert np.array_equal(info, rinfo) and np.array_equal(info_t, rinfo) and rinfo[4] == -1 and (rinfo[:4] == 0).all(), (N, info, rinfo)
        worst_o = max(worst_o, np.max(np.abs(ll[:4] - ref[:4]) / np.abs(ref[:4])))
        worst_t = max(worst_t, np.max(np.abs(ll[:4] - ll_t[:4]) / np.abs(ll_t[:4])))
        assert worst_o <= 1e-9 and worst_t <= 1e-11, (N, Nl, kname, mb, worst_o, worst_t)
    print("N = 192..383 (four waves per evaluation): worst vs oracle %.2e, vs tile kernels %.2e" % (worst_o, worst_t))
    worst = 0.0
    for N, kname in [(n, KNAMES[n % 4]) for n in (64, 79, 110, 111, 112, 150, 159, 176, 191)] + \
                    [(16 * nb - 1 - (nb % 3), k) for nb in range(5, 13) for k in KNAMES]:   # every instantiation of both families
        t, y, s = _problem(rng, [N - N // 2, N // 2])
        delays, alpha, rho = _params(rng, 40, 2)
        with gp.Objective(t, y, s, kname) as obj:
            obj.set_option("small_wide_max", 0)          # one wave per evaluation
            a, ia = obj.loglik_batch(delays, alpha, rho)
            obj.set_option("small_wide_max", 256)        # four waves per evaluation for this batch of 40
            b, ib = obj.loglik_batch(delays, alpha, rho)
        assert (ia == 0).all() and (ib == 0).all()
        assert np.array_equal(a, b), N
    # a not-positive-definite evaluation stops all four waves (duplicated time, no noise, no B term)
    n = 300
    tt = [np.sort(rng.uniform(0, 20, n))]
    tt[0][200] = tt[0][199]
    with gp.Objective(tt, [rng.standard_normal(n)], [np.zeros(n)], "rbf", marginalise_b=False) as obj:
        ll, info = obj.loglik_batch([[0.0]], [[1.0]], [30.0])
    assert info[0] > 0 and np.isnan(ll[0])


def test_golden_cases_on_the_small_path(gp, golden):
    worst, n = 0.0, 0
    for c in golden["cases"]:
        with gp.Objective(c["t"], c["y"], c["sigma"], c["kernel"], marginalise_b=c["marginalise_b"]) as obj:
            if not obj.get_option("small_n_active"):
                continue
            ll, info = obj.loglik_batch([c["delays"]], [c["alpha"]], [c["rho"]])
        assert info[0] == 0
        worst = max(worst, abs(ll[0] - c["loglik"]) / abs(c["loglik"]))
        n += 1
    print("%d golden cases on the small-N path, worst relative error %.3e" % (n, worst))
    assert n >= 30 and worst <= LL_RTOL


def test_status_codes_and_not_positive_definite(gp, oracle):
    """alpha <= 0 -> -1, rho <= 0 -> -2 (delayedCovariance.jl:3, :5-7); a duplicated time with sigma = 0 and no B term is
    singular: info = the order of the first non-positive pivot, as LAPACK reports it (oracle), loglik = NaN; the other
    evaluations of the batch are unaffected."""
    rng = np.random.default_rng(3)
    for dup in (5, 16, 37, 100):
        n = 101
        t = [np.sort(rng.uniform(0, 20, n))]
        t[0][dup] = t[0][dup - 1]
        y = [rng.standard_normal(n)]
        s = [np.zeros(n)]
        delays, alpha, rho = np.zeros((4, 1)), np.array([[1.0], [2.0], [-1.0], [1.5]]), np.array([2.0, 3.0, 1.0, -0.5])
        with gp.Objective(t, y, s, "OU", marginalise_b=False) as obj:
            assert obj.get_option("small_n_active") == 1
            ll, info = obj.loglik_batch(delays, alpha, rho)
        ref, rinfo = oracle.loglik_batch("OU", t, y, s, delays, alpha, rho, False)
        assert info[2] == -1 and info[3] == -2 and np.isnan(ll[2]) and np.isnan(ll[3])
        # exactly singular in exact arithmetic; in floating point the pivot at the duplicate is a rounding residue of either
        # sign (the oracle's too), so: either flagged AT the duplicate (LAPACK's info = its order) with NaN, or passed
        for i in (0, 1):
            assert info[i] in (0, dup + 1) and rinfo[i] in (0, dup + 1), (dup, info, rinfo)
            if info[i]:
                assert np.isnan(ll[i])
    # a clearly indefinite matrix: negative noise variance cannot happen, so use a huge B-free alpha with rbf (rank-deficient)
    n = 120
    t = [np.linspace(0, 1, n)]
    y = [rng.standard_normal(n)]
    s = [np.zeros(n)]
    with gp.Objective(t, y, s, "rbf", marginalise_b=False) as obj:
        ll, info = obj.loglik_batch([[0.0]], [[1.0]], [50.0])
    ref, rinfo = oracle.loglik_batch("rbf", t, y, s, [[0.0]], [[1.0]], [50.0], False)
    assert rinfo[0] > 0 and info[0] > 0 and np.isnan(ll[0])
    assert abs(int(info[0]) - int(rinfo[0])) <= 3, (info, rinfo)   # numerically rank ~10: the first failing pivot may differ by rounding


def test_large_batches_and_device_pointers(gp, oracle):
    """12 321 evaluations in one launch (README.md:227: the 111 x 111 grid of the three-band example) -- spot-checked
    against the oracle, all against the tile kernels; and the device-pointer entry on a torch stream."""
    import torch
    from gpcc_amd import synthetic
    t, y, s, _ = synthetic.simulate_lightcurves([60, 50, 40], seed=1, gap_band=1, span=20.0)
    alpha0, rho0 = synthetic.default_hyperparameters(y)
    g = np.arange(0.5, 6.0 + 1e-9, 0.05)
    d2, d3 = np.meshgrid(g, g, indexing="ij")
    delays = np.stack([np.zeros(d2.size), d2.ravel(), d3.ravel()], 1)
    M = len(delays)
    assert M == 12321
    rng = np.random.default_rng(5)
    alpha = alpha0 * rng.uniform(0.7, 1.4, (M, 3))
    rho = rho0 * rng.uniform(0.5, 2.0, M)
    with gp.Objective(t, y, s, gp.matern32) as obj:
        ll, info = obj.loglik_batch(delays, alpha, rho)
        dd, da, dr = (torch.tensor(a, device="cuda") for a in (delays, alpha, rho))
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            out, oinfo = obj.loglik_batch_device(dd, da, dr)
        st.synchronize()
        obj.set_option("small_n", 0)
        ll_t, info_t = obj.loglik_batch(delays, alpha, rho)
    assert (info == 0).all() and (info_t == 0).all() and (oinfo.cpu().numpy() == 0).all()
    assert np.array_equal(out.cpu().numpy(), ll)
    assert np.max(np.abs(ll - ll_t) / np.abs(ll_t)) <= 1e-11
    pick = rng.choice(M, 64, replace=False)
    ref, rinfo = oracle.loglik_batch("matern32", t, y, s, delays[pick], alpha[pick], rho[pick], True, nthreads=8)
    assert (rinfo == 0).all() and np.max(np.abs(ll[pick] - ref) / np.abs(ref)) <= 1e-9


def test_fp32_handles_take_the_fp64_small_path(gp, oracle):
    rng = np.random.default_rng(11)
    t, y, s = _problem(rng, [60, 50], sigma_lo=0.05)
    delays, alpha, rho = _params(rng, 16, 2)
    alpha *= 30.0    # ill-conditioned for fp32; the small-N kernel computes in fp64 whatever the handle's precision
    ref, rinfo = oracle.loglik_batch("matern52", t, y, s, delays, alpha, rho, True)
    with gp.Objective(t, y, s, "matern52", precision="fp32") as obj:
        ll, info = obj.loglik_batch(delays, alpha, rho)
        assert obj.get_option("small_n_count") == 16 and obj.get_option("fp32_guard_count") == 0
        assert np.all(obj.conditioning(16) == 0.0)
    assert (info == 0).all() and np.max(np.abs(ll - ref) / np.abs(ref)) <= 1e-9


def test_extreme_scales_on_the_small_path(gp, oracle):
    from gpcc_amd import synthetic
    t, y, s, _ = synthetic.simulate_lightcurves([80, 70], seed=3)
    for scale in (1e-30, 1e-12, 1.0, 1e12, 1e30):
        ys = [a * scale for a in y]
        ss = [a * scale for a in s]
        alpha = np.array([[1.3 * scale, 0.9 * scale]])
        with gp.Objective(t, ys, ss, gp.matern32) as obj:
            assert obj.get_option("small_n_active") == 1
            ll, info = obj.loglik_batch([[0.0, 2.0]], alpha, [3.5])
        ref, rinfo = oracle.loglik_batch("matern32", t, ys, ss, [[0.0, 2.0]], alpha, [3.5], True)
        assert info[0] == 0 and rinfo[0] == 0 and np.isfinite(ll[0])
        assert abs(ll[0] - ref[0]) <= 1e-9 * abs(ref[0]), (scale, ll[0], ref[0])


def test_readme_sweep_fit_matches_the_tile_path(gp):
    """The README's sweep (N = 110, delays 0:0.2:20, per-delay fit): the lock-step optimiser over the small-N kernel and over
    the tile kernels reaches the same optimised log-likelihoods (both are the same objective to ~1e-13; trajectories may
    part at a comparison, so the bar is on the optimum, 1e-6)."""
    from gpcc_amd import synthetic
    t, y, s, _ = synthetic.simulate_lightcurves([60, 50], seed=1, gap_band=1, span=20.0)
    grid = np.arange(0.0, 20.0 + 1e-9, 0.2)
    cand = np.stack([np.zeros_like(grid), grid], 1)
    with gp.Objective(t, y, s, gp.matern32) as obj:
        a = obj.grid_loglik(cand, 200, rhomax=300.0, seed=1)
        obj.set_option("small_n", 0)
        b = obj.grid_loglik(cand, 200, rhomax=300.0, seed=1)
    assert (a[3] == 0).all() and (b[3] == 0).all()
    assert np.max(np.abs(a[0] - b[0]) / np.abs(b[0])) <= 1e-6
    p = gp.getprobabilities(a[0])
    assert abs(grid[int(np.argmax(p))] - 2.0) <= 0.4 + 1e-9


def test_fit_device_unpack_and_speculative_rounds_change_no_bit(gp):
    """gpcc_grid_loglik on the small-N path: (a) the kernel unpacking the optimiser's vectors itself (gpcc_transforms.h on
    the device) gives the same bits as the host unpacking them (same header on the host); (b) speculative rounds (all four
    candidate points of an iteration evaluated at once) walk exactly the trajectory of the plain rounds: every output
    identical, fewer rounds, more evaluations."""
    from gpcc_amd import synthetic
    for Nl, kern, R in (([60, 50], "matern32", 1), ([60, 50, 40], "OU", 2)):
        L = len(Nl)
        t, y, s, _ = synthetic.simulate_lightcurves(Nl, seed=1, gap_band=1, span=20.0)
        rng = np.random.default_rng(9)
        cand = np.concatenate([np.zeros((23, 1)), rng.uniform(0.0, 8.0, (23, L - 1))], axis=1)
        res = {}
        with gp.Objective(t, y, s, kern) as obj:
            for spec, dev in ((0, 0), (0, 1), (1, 1), (1, 0)):
                obj.set_option("fit_speculate", spec)
                obj.set_option("fit_device_unpack", dev)
                res[spec, dev] = obj.grid_loglik(cand, 150, numberofrestarts=R, rhomax=100.0, seed=3)
        base = res[0, 0]
        assert (base[3] == 0).all()
        for key, r in res.items():
            assert all(np.array_equal(u, v) for u, v in zip(r[:5], base[:5])), key
        assert res[0, 1][5] == base[5]                       # same evaluations, same rounds
        assert res[1, 1][5] == res[1, 0][5]
        (fc0, rd0), (fc1, rd1) = base[5], res[1, 1][5]
        print("%s: plain %d evaluations in %d rounds; speculative %d in %d" % (Nl, fc0, rd0, fc1, rd1))
        assert rd1 < 0.8 * rd0 and fc1 > fc0


def test_unpack_on_the_device_is_bitwise_the_hosts(gp):
    """One evaluation per parameter vector through the optimiser-request form of the kernel (device unpack) against
    gpcc_unpack_params (host) + the plain form: identical bits, including extreme arguments of the transforms."""
    from gpcc_amd import api, synthetic
    t, y, s, _ = synthetic.simulate_lightcurves([30, 25], seed=2, span=20.0)
    rng = np.random.default_rng(4)
    X = np.concatenate([rng.uniform(-6, 6, (300, 3)), rng.uniform(-40, 45, (200, 3)),
                        np.array([[31.0, -700.0, 0.0], [-30.0, 30.0, 40.0], [0.0, 0.0, -38.0]])])
    alpha, rho = api.unpack_params(X, 2, 0.1, 300.0)
    assert np.all(alpha > 0) and np.all(rho >= 0.1) and np.all(rho <= 300.0)
    cand = np.array([[0.0, 1.5]])
    with gp.Objective(t, y, s, "matern32") as obj:
        ll, info = obj.loglik_batch(np.repeat(cand, len(X), 0), alpha, rho)
        # iterations = 0: the fit evaluates the candidates (through the device unpack), then the affine simplex + its centroid
        for i in range(0, len(X), 50):
            a = obj.grid_loglik(cand, 0, initialrandom=50, rhomin=0.1, rhomax=300.0, init_params=X[i:i + 50][None, :, :]) if i + 50 <= len(X) else None
            if a is None:
                continue
            obj.set_option("fit_device_unpack", 0)
            b = obj.grid_loglik(cand, 0, initialrandom=50, rhomin=0.1, rhomax=300.0, init_params=X[i:i + 50][None, :, :])
            obj.set_option("fit_device_unpack", 1)
            assert all(np.array_equal(u, v) for u, v in zip(a[:5], b[:5])), i
    assert np.isfinite(ll[info == 0]).all()


def test_fit_sliced_over_host_threads_changes_no_bit(gp):
    """A large grid is cut into slices that run on their own host threads / streams / pinned buffers (lanes): same
    results as one slice, for every thread count, with restarts straddling the cuts."""
    from gpcc_amd import synthetic
    t, y, s, _ = synthetic.simulate_lightcurves([40, 35, 30], seed=6, gap_band=1, span=20.0)
    rng = np.random.default_rng(2)
    G = 257
    cand = np.concatenate([np.zeros((G, 1)), rng.uniform(0.0, 8.0, (G, 2))], axis=1)
    with gp.Objective(t, y, s, "matern52") as obj:
        obj.set_option("fit_threads", 1)
        base = obj.grid_loglik(cand, 60, numberofrestarts=3, rhomax=100.0, seed=3)
        for T in (2, 3, 4):
            obj.set_option("fit_threads", T)
            r = obj.grid_loglik(cand, 60, numberofrestarts=3, rhomax=100.0, seed=3)
            assert all(np.array_equal(u, v) for u, v in zip(r[:5], base[:5])), T
        with pytest.raises(gp.GpccError):
            obj.set_option("fit_threads", 5)
    assert (base[3] == 0).all()
