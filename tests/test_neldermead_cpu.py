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
