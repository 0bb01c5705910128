"""The lock-step batched Nelder-Mead against a plain scalar restatement of Optim.jl's loop (the
structure of Optim.jl v1 `update_state!(::NelderMeadState)` / `after_while!`): every problem of a
batch must follow exactly the trajectory it follows alone."""
import numpy as np

from gpcc_amd.neldermead import BatchedNelderMead


def scalar_nm(f, x0, iterations, g_tol=1e-6):
    n = len(x0)
    al, be, ga, de = 1.0, 1 + 2 / n, 0.75 - 1 / (2 * n), 1 - 1 / n
    S = [np.array(x0, float)]
    for i in range(n):
        v = np.array(x0, float)
        v[i] = 1.5 * v[i] + 0.025
        S.append(v)
    fs = np.array([f(v) for v in S])
    calls = n + 1
    order = list(np.argsort(fs, kind="stable"))

    def nmobj():
        return np.sqrt(np.var(fs, ddof=1) * n / (n + 1))

    it = 0
    converged = nmobj() <= g_tol
    while not converged and it < iterations:
        it += 1
        c = np.mean([S[i] for i in order[:n]], axis=0)
        xh = S[order[n]]
        xr = c + al * (c - xh)
        fr = f(xr); calls += 1
        shrink = False
        if fr < fs[order[0]]:
            xe = c + be * (xr - c)
            fe = f(xe); calls += 1
            if fe < fr:
                S[order[n]], fs[order[n]] = xe, fe
            else:
                S[order[n]], fs[order[n]] = xr, fr
            order = [order[n]] + order[:n]
        elif fr < fs[order[n - 1]]:
            S[order[n]], fs[order[n]] = xr, fr
            order = list(np.argsort(fs, kind="stable"))
        else:
            if fr < fs[order[n]]:
                xc = c + ga * (xr - c)
                fc = f(xc); calls += 1
                if fc < fr:
                    S[order[n]], fs[order[n]] = xc, fc
                    order = list(np.argsort(fs, kind="stable"))
                else:
                    shrink = True
            else:
                xc = c - ga * (xr - c)
                fc = f(xc); calls += 1
                if fc < fs[order[n]]:
                    S[order[n]], fs[order[n]] = xc, fc
                    order = list(np.argsort(fs, kind="stable"))
                else:
                    shrink = True
        if shrink:
            xl = S[order[0]].copy()
            for j in range(1, n + 1):
                o = order[j]
                S[o] = xl + de * (S[o] - xl)
                fs[o] = f(S[o]); calls += 1
            order = list(np.argsort(fs, kind="stable"))
        converged = nmobj() <= g_tol
    c = np.mean([S[i] for i in order[:n]], axis=0)
    fc = f(c); calls += 1
    if fc < fs[order[0]]:
        return c, fc, it, calls
    return S[order[0]], fs[order[0]], it, calls


def rosen(x):
    return 100.0 * (x[1] - x[0] ** 2) ** 2 + (1 - x[0]) ** 2


def himmel(x):
    return (x[0] ** 2 + x[1] - 11) ** 2 + (x[0] + x[1] ** 2 - 7) ** 2


def bowl3(x):
    return (x[0] - 1) ** 2 + 3 * (x[1] + 2) ** 2 + 0.5 * (x[2] - 0.3) ** 2 + 0.1 * np.sin(5 * x[0]) + 2.0


def test_batched_matches_scalar_trajectories():
    rng = np.random.default_rng(0)
    for fun, n in ((rosen, 2), (himmel, 2), (bowl3, 3)):
        x0 = rng.standard_normal((17, n)) * 2
        for iters in (7, 60, 2000):
            nm = BatchedNelderMead(x0, lambda pid, X: np.array([fun(x) for x in X]), iterations=iters, g_tol=1e-6)
            xb, fb = nm.run()
            calls = 0
            for p in range(len(x0)):
                xs, fsv, it, c = scalar_nm(fun, x0[p], iters, 1e-6)
                calls += c
                assert nm.iterations_done[p] == it
                np.testing.assert_allclose(xb[p], xs, rtol=1e-12, atol=1e-12)
                assert abs(fb[p] - fsv) <= 1e-12 * max(1.0, abs(fsv))
            assert nm.f_calls == calls
    assert nm.rounds < nm.f_calls           # evaluations really were batched


def test_rejected_points_behave_like_safewrapper():
    # a wall of non-finite values: the optimiser must stay on the finite side and still converge
    def f(x):
        return np.inf if x[0] < -0.5 else (x[0] - 0.2) ** 2 + (x[1] + 0.1) ** 2

    x0 = np.array([[1.0, 1.0], [0.0, 2.0], [3.0, -3.0]])
    nm = BatchedNelderMead(x0, lambda pid, X: np.array([f(x) if np.isfinite(f(x)) else np.nan for x in X]), 500)
    xb, fb = nm.run()
    assert np.all(fb < 1e-5) and np.allclose(xb, [0.2, -0.1], atol=5e-3)


def native_nm(fun, x0, iterations, g_tol=1e-6):
    """gpcc_neldermead_batch (the C++ optimiser inside gpcc_grid_loglik) over a Python objective."""
    import ctypes

    from gpcc_amd import _capi
    lib = _capi.load()
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    P, n = x0.shape
    seen = []

    def cb(ctx, K, pidx, X, out):
        Xa = np.ctypeslib.as_array(X, shape=(K, n))
        pa = np.ctypeslib.as_array(pidx, shape=(K,))
        seen.append(pa.copy())
        vals = fun(pa, Xa)
        for i in range(K):
            out[i] = vals[i]
        return 0

    cfun = _capi.BATCH_OBJECTIVE(cb)
    xmin, fmin = np.empty((P, n)), np.empty(P)
    its = np.empty(P, dtype=np.int32)
    stats = (ctypes.c_longlong * 2)()
    dp = ctypes.POINTER(ctypes.c_double)
    rc = lib.gpcc_neldermead_batch(P, n, iterations, g_tol, x0.ctypes.data_as(dp), cfun, None, xmin.ctypes.data_as(dp),
                                   fmin.ctypes.data_as(dp), its.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), stats)
    assert rc == 0, _capi.last_error()
    return xmin, fmin, its, (stats[0], stats[1]), seen


def test_native_optimiser_equals_numpy_optimiser_bitwise():
    """Same requests in the same order, same decisions, same bits: the C++ lock-step Nelder-Mead of
    gpcc_grid_loglik vs neldermead.BatchedNelderMead (and through it the scalar Optim restatement above)."""
    rng = np.random.default_rng(4)
    for fun, n in ((rosen, 2), (himmel, 2), (bowl3, 3)):
        x0 = rng.standard_normal((23, n)) * 2
        fb = lambda pid, X: np.array([fun(x) for x in X])          # noqa: E731
        for iters in (0, 1, 9, 75, 3000):
            seen_py = []

            def fpy(pid, X):
                seen_py.append(np.array(pid))
                return fb(pid, X)

            nm = BatchedNelderMead(x0, fpy, iterations=iters, g_tol=1e-6)
            xp, fp = nm.run()
            xn, fn, its, (calls, rounds), seen = native_nm(fb, x0, iters)
            assert np.array_equal(xn, xp) and np.array_equal(fn, fp)
            assert np.array_equal(its, nm.iterations_done) and calls == nm.f_calls and rounds == nm.rounds
            assert len(seen) == len(seen_py) and all(np.array_equal(a, b) for a, b in zip(seen, seen_py))


def test_native_optimiser_rejected_points_and_errors():
    def f(pid, X):
        return np.array([np.nan if x[0] < -0.5 else (x[0] - 0.2) ** 2 + (x[1] + 0.1) ** 2 for x in X])

    x0 = np.array([[1.0, 1.0], [0.0, 2.0], [3.0, -3.0]])
    nm = BatchedNelderMead(x0, f, 500)
    xp, fp = nm.run()
    xn, fn, its, _, _ = native_nm(f, x0, 500)
    assert np.array_equal(xn, xp) and np.array_equal(fn, fp) and np.all(fn < 1e-5)
    # all points rejected: finishes, reports +Inf
    xn, fn, its, _, _ = native_nm(lambda pid, X: np.full(len(X), np.inf), x0, 50)
    assert np.all(np.isposinf(fn))
    from gpcc_amd import _capi
    assert _capi.load().gpcc_neldermead_batch(3, 0, 5, 1e-6, None, _capi.BATCH_OBJECTIVE(lambda *a: 0), None, None, None,
                                              None, None) == -1


def test_unpack_params_matches_numpy_transforms():
    from gpcc_amd import api, fit
    rng = np.random.default_rng(9)
    X = np.concatenate([rng.standard_normal((50, 4)) * 8, [[40.0, -40.0, 0.0, 700.0], [31.0, 29.9, -745.0, -800.0]]])
    alpha, rho = api.unpack_params(X, 3, 0.1, 300.0)
    np.testing.assert_allclose(alpha, fit.makepositive(X[:, :3]) + 1e-8, rtol=1e-14)
    with np.errstate(over="ignore"):
        np.testing.assert_allclose(rho, fit.transformbetween(X[:, 3], 0.1, 300.0), rtol=1e-14)
    assert np.all(alpha > 0) and np.all((rho >= 0.1) & (rho <= 300.0))


def test_shared_transforms_within_four_ulp_of_libm():
    """gpcc_transforms.h (the exp / log the host AND the device unpack with): softplus and the logistic map against numpy
    (libm) over the whole range an optimiser can reach, incl. the branches x > 30, exp(x) below an ulp of 1, overflow."""
    from gpcc_amd import api
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-745, 709, 20000), rng.uniform(-40, 40, 20000), rng.uniform(-2, 2, 20000),
                        np.array([-800.0, -745.0, -37.0, -36.0, 0.0, 29.999, 30.0, 30.001, 709.0, 800.0])])
    X = np.stack([x, x], 1)
    alpha, rho = api.unpack_params(X, 1, 0.1, 300.0)
    sp = np.where(x > 30.0, x, np.log1p(np.exp(np.minimum(x, 30.0)))) + 1e-8
    with np.errstate(over="ignore"):
        lg = 0.1 + (300.0 - 0.1) / (1.0 + np.exp(-x))
    assert np.all(np.abs(alpha[:, 0] - sp) <= 4 * np.spacing(sp))
    assert np.all(np.abs(rho - lg) <= 4 * np.spacing(lg))
    assert np.all(alpha > 0) and rho.min() >= 0.1 and rho.max() <= 300.0
