"""Third-party witnesses for the CALLER's arithmetic, as tests/test_oracle_thirdparty.py has them for
a1-a9: code that knows nothing of this repository must reproduce what the restatements compute.

* a12 -- `optimize(f, x0, NelderMead(), Options(iterations, g_tol))`, src/gpccfixdelay_marginaliseb.jl:205-211.  Witness: SciPy's
  Nelder-Mead with `adaptive=True` (the same Gao-Han parameters Optim's AdaptiveParameters uses: alpha 1, beta 1 + 2/n,
  gamma 0.75 - 1/(2n), delta 1 - 1/n) started from the simplex Optim's AffineSimplexer(a = 0.025, b = 0.5) builds.  Both are the
  textbook iteration (reflect; expand / accept / outside or inside contraction / shrink), so the SEQUENCE OF EVALUATED POINTS of
  gpcc_neldermead_batch and of SciPy must coincide until rounding or a tie-break separates them (SciPy forms x_r as
  (1 + rho) xbar - rho x_h, the restatement as c + alpha (c - x_h); SciPy accepts an outside contraction on <=, Optim on <).
  The test reports the first divergent evaluation and demands a long common prefix.
* a11 -- MiscUtil.makepositive / transformbetween (marginaliseb.jl:112-114; MiscUtil.jl is not under /root/reference, the
  definitions softplus and a + (b - a) logistic(x) are ASSUMED): scipy.special's log1p / expit versions of those definitions
  against gpcc_unpack_params.

None of this lifts "parity unpinned": the witnesses know the textbook algorithms, not Optim.jl's or MiscUtil.jl's source
(tools/pin_reference.jl is the route to that).  Nothing outside tests/ imports scipy.optimize."""
import numpy as np
import pytest

scipy_optimize = pytest.importorskip("scipy.optimize")


def rosen3(x):
    return 100.0 * (x[1] - x[0] ** 2) ** 2 + (1 - x[0]) ** 2 + 100.0 * (x[2] - x[1] ** 2) ** 2 + (1 - x[1]) ** 2


def bowl4(x):
    a = np.array([1.0, 3.0, 0.5, 7.0])
    c = np.array([0.3, -1.2, 2.0, 0.7])
    return float(np.sum(a * (x - c) ** 2) + 0.3 * np.sin(x[0] * x[1]) + 0.1 * x[2] * x[3])


def loglike3(x):
    # shaped like the negative objective of a 2-band fit in the optimiser's coordinates: softplus amplitudes, a logistic length scale
    a1, a2 = np.log1p(np.exp(x[0])), np.log1p(np.exp(x[1]))
    rho = 0.1 + 299.9 / (1.0 + np.exp(-x[2]))
    return float((a1 - 2.0) ** 2 + 3.0 * (a2 - 0.7) ** 2 + 0.002 * (rho - 12.0) ** 2 + 0.5 * np.log(a1 * a2 + 1.0))


def native_points(fun, x0, iterations):
    """every point gpcc_neldermead_batch evaluates for ONE problem, in order (g_tol = 0: it runs all its iterations)"""
    import ctypes

    from gpcc_amd import _capi
    lib = _capi.load()
    n = len(x0)
    pts = []

    def cb(ctx, K, pidx, X, out):
        Xa = np.ctypeslib.as_array(X, shape=(K, n))
        for i in range(K):
            pts.append(Xa[i].copy())
            out[i] = fun(Xa[i])
        return 0

    cfun = _capi.BATCH_OBJECTIVE(cb)
    x0a = np.ascontiguousarray(x0, dtype=np.float64).reshape(1, n)
    xmin, fmin = np.empty((1, n)), np.empty(1)
    dp = ctypes.POINTER(ctypes.c_double)
    rc = lib.gpcc_neldermead_batch(1, n, iterations, 0.0, x0a.ctypes.data_as(dp), cfun, None, xmin.ctypes.data_as(dp),
                                   fmin.ctypes.data_as(dp), None, None)
    assert rc == 0
    return np.array(pts), xmin[0], fmin[0]


def scipy_points(fun, x0, iterations):
    n = len(x0)
    sim = np.repeat(np.asarray(x0, float)[None, :], n + 1, axis=0)        # Optim's AffineSimplexer(a = 0.025, b = 0.5)
    for i in range(n):
        sim[i + 1, i] = (1.0 + 0.5) * x0[i] + 0.025
    pts = []

    def f(x):
        pts.append(np.array(x, float))
        return fun(x)

    res = scipy_optimize.minimize(f, x0, method="Nelder-Mead",
                                  options=dict(adaptive=True, initial_simplex=sim, maxiter=iterations, maxfev=10 ** 9,
                                               xatol=0.0, fatol=0.0))
    return np.array(pts), res.x, res.fun


@pytest.mark.parametrize("fun,x0", [(rosen3, [-1.2, 1.0, 0.7]), (bowl4, [2.0, 1.0, -1.0, 3.0]), (loglike3, [0.5, -0.3, 0.1]),
                                    (rosen3, [0.3, 0.4, 0.5])])
def test_scipy_nelder_mead_walks_the_same_points(fun, x0):
    iters = 120
    a, xa, fa = native_points(fun, np.array(x0), iters)
    b, xb, fb = scipy_points(fun, np.array(x0), iters)
    n = len(x0)
    assert np.array_equal(a[:n + 1], b[:n + 1])                      # the AffineSimplexer simplex, vertex by vertex
    m = min(len(a), len(b))
    scale = 1.0 + np.abs(b[:m])
    close = np.all(np.abs(a[:m] - b[:m]) <= 1e-9 * scale, axis=1)
    first = int(np.argmin(close)) if not close.all() else m
    print("%s from %s: %d native / %d scipy evaluations, first divergent evaluation %s"
          % (fun.__name__, x0, len(a), len(b), first if first < m else "none (common prefix %d)" % m))
    # every iteration costs 1-2 evaluations (n when it shrinks): >= 150 common evaluations are >= 75 identical iterations --
    # reflection, expansion, both contractions, acceptance rules and the ordering of the simplex all agree that long
    assert first >= 150, first
    if first >= m:                                                    # never separated: the minimisers agree too
        assert abs(fa - fb) <= 1e-9 * (1.0 + abs(fb)) or fa <= fb    # (the restatement also tries the final centroid, Optim's after_while!)


def test_parameter_transforms_against_scipy_special():
    """unpack (marginaliseb.jl:112-126) under the ASSUMED MiscUtil definitions, written with scipy.special primitives:
    makepositive(x) = log1p(exp(x)) (softplus; x for large x), transformbetween(x, a, b) = a + (b - a) expit(x)."""
    from scipy import special

    from gpcc_amd import api
    rng = np.random.default_rng(3)
    X = np.concatenate([rng.standard_normal((200, 4)) * 6, [[35.0, -35.0, 0.0, 50.0], [700.0, -700.0, 1e-9, -50.0]]])
    alpha, rho = api.unpack_params(X, 3, 0.1, 300.0)
    with np.errstate(over="ignore"):
        sp = np.where(X[:, :3] > 30.0, X[:, :3], special.log1p(np.exp(np.minimum(X[:, :3], 30.0))))
    np.testing.assert_allclose(alpha, sp + 1e-8, rtol=4e-15, atol=0)
    np.testing.assert_allclose(rho, 0.1 + (300.0 - 0.1) * special.expit(X[:, 3]), rtol=4e-15, atol=0)
