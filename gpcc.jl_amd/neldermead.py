"""Lock-step batched Nelder-Mead (host logic, numpy): P independent minimisations advance together so
that every round submits ONE batch of objective evaluations -- what turns the reference's
strictly sequential per-delay optimiser (src/gpccfixdelay_marginaliseb.jl:203-215,
`optimize(safenegativeobj, x0, NelderMead(), opt)`) into GPU-sized batches while keeping each
problem's own trajectory.

The per-problem algorithm restates Optim.jl v1's NelderMead as the reference uses it (defaults):
AdaptiveParameters (alpha = 1, beta = 1 + 2/n, gamma = 0.75 - 1/(2n), delta = 1 - 1/n),
AffineSimplexer (a = 0.025, b = 0.5), one reflection per iteration followed by expansion / outside
or inside contraction / shrink, convergence when sqrt(var(f_simplex) * n/(n+1)) <= g_tol, and after
the loop the centroid of the n best vertices is evaluated and returned if it beats the best vertex.
Optim.jl is not vendored under /root/reference and cannot be run here, so trajectory parity with it is
unpinned ("Nelder-Mead trajectory parity")."""
import numpy as np

REFLECT, EXPAND, OUTSIDE, INSIDE, SHRINK, FINAL, DONE = range(7)


def _nmobjective(fs):
    n1 = fs.shape[1]
    with np.errstate(invalid="ignore"):
        return np.sqrt(np.var(fs, axis=1, ddof=1) * ((n1 - 1) / n1))


def _sortperm(fs):
    return np.argsort(fs, axis=1, kind="stable")   # Julia's sortperm is stable


class BatchedNelderMead:
    """Minimise f for P problems of dimension n.  `fbatch(pidx, X)` evaluates the rows of X (K, n),
    row i belonging to problem pidx[i], and returns K values (NaN / +inf = rejected point, like the
    reference's safewrapper around the negative objective, marginaliseb.jl:153)."""

    def __init__(self, x0, fbatch, iterations, g_tol=1e-6):
        self.x0 = np.array(x0, dtype=np.float64)
        self.P, self.n = self.x0.shape
        self.fbatch = fbatch
        self.iterations = int(iterations)
        self.g_tol = g_tol
        n = self.n
        self.alpha, self.beta = 1.0, 1.0 + 2.0 / n
        self.gamma, self.delta = 0.75 - 1.0 / (2 * n), 1.0 - 1.0 / n
        self.f_calls = 0
        self.rounds = 0

    def _eval(self, pidx, X):
        f = np.asarray(self.fbatch(pidx, X), dtype=np.float64)
        self.f_calls += len(pidx)
        self.rounds += 1
        return np.where(np.isnan(f), np.inf, f)

    def run(self):
        P, n = self.P, self.n
        ar = np.arange(P)
        S = np.repeat(self.x0[:, None, :], n + 1, axis=1)          # AffineSimplexer(a = 0.025, b = 0.5)
        for i in range(n):
            S[:, i + 1, i] = (1.0 + 0.5) * self.x0[:, i] + 0.025
        fs = self._eval(np.repeat(ar, n + 1), S.reshape(-1, n)).reshape(P, n + 1)
        order = _sortperm(fs)
        it = np.zeros(P, dtype=np.int64)
        phase = np.full(P, REFLECT)
        phase[(_nmobjective(fs) <= self.g_tol) | (self.iterations <= 0)] = FINAL
        xr, fr, cen = np.zeros((P, n)), np.zeros(P), np.zeros((P, n))
        xmin, fmin = np.zeros((P, n)), np.zeros(P)

        def centroid(idx):
            return np.take_along_axis(S[idx], order[idx][:, :n, None], axis=1).mean(axis=1)

        while (phase != DONE).any():
            req = phase.copy()                     # phases as of request time; `phase` receives the transitions
            groups = {ph: ar[req == ph] for ph in (REFLECT, EXPAND, OUTSIDE, INSIDE, SHRINK, FINAL)}
            pid_l, X_l = [], []
            g = groups[REFLECT]
            if len(g):
                cen[g] = centroid(g)
                xr[g] = cen[g] + self.alpha * (cen[g] - S[g, order[g, n]])
                pid_l.append(g); X_l.append(xr[g])
            g = groups[EXPAND]
            if len(g):
                pid_l.append(g); X_l.append(cen[g] + self.beta * (xr[g] - cen[g]))
            g = groups[OUTSIDE]
            if len(g):
                pid_l.append(g); X_l.append(cen[g] + self.gamma * (xr[g] - cen[g]))
            g = groups[INSIDE]
            if len(g):
                pid_l.append(g); X_l.append(cen[g] - self.gamma * (xr[g] - cen[g]))
            g = groups[SHRINK]
            if len(g):
                xl = S[g, order[g, 0]].copy()
                for j in range(1, n + 1):
                    o = order[g, j]
                    S[g, o] = xl + self.delta * (S[g, o] - xl)
                    pid_l.append(g); X_l.append(S[g, o])
            g = groups[FINAL]
            if len(g):
                cen[g] = centroid(g)
                pid_l.append(g); X_l.append(cen[g])
            X = np.concatenate(X_l)
            fv = self._eval(np.concatenate(pid_l), X)

            pos = 0

            def take(k):
                nonlocal pos
                v, x = fv[pos:pos + k], X[pos:pos + k]
                pos += k
                return v, x

            finished = []                           # problems that completed an iteration this round
            g = groups[REFLECT]
            if len(g):
                v, _ = take(len(g))
                fr[g] = v
                fl, fsh, fh = fs[g, order[g, 0]], fs[g, order[g, n - 1]], fs[g, order[g, n]]
                to_exp = v < fl
                acc = ~to_exp & (v < fsh)
                to_out = ~to_exp & ~acc & (v < fh)
                to_in = ~to_exp & ~acc & ~to_out
                a = g[acc]
                if len(a):
                    S[a, order[a, n]] = xr[a]
                    fs[a, order[a, n]] = fr[a]
                    order[a] = _sortperm(fs[a])
                    finished.append(a)
                phase[g[to_exp]], phase[g[to_out]], phase[g[to_in]] = EXPAND, OUTSIDE, INSIDE
            g = groups[EXPAND]
            if len(g):
                v, x = take(len(g))
                better = v < fr[g]
                hi = order[g, n].copy()
                S[g, hi] = np.where(better[:, None], x, xr[g])
                fs[g, hi] = np.where(better, v, fr[g])
                order[g] = np.concatenate([hi[:, None], order[g, :n]], axis=1)   # the new point is the lowest
                phase[g] = REFLECT
                finished.append(g)
            for ph in (OUTSIDE, INSIDE):
                g = groups[ph]
                if len(g):
                    v, x = take(len(g))
                    ok = (v < fr[g]) if ph == OUTSIDE else (v < fs[g, order[g, n]])
                    a = g[ok]
                    if len(a):
                        S[a, order[a, n]] = x[ok]
                        fs[a, order[a, n]] = v[ok]
                        order[a] = _sortperm(fs[a])
                        phase[a] = REFLECT
                        finished.append(a)
                    phase[g[~ok]] = SHRINK
            g = groups[SHRINK]
            if len(g):
                for j in range(1, n + 1):
                    v, _ = take(len(g))
                    fs[g, order[g, j]] = v
                order[g] = _sortperm(fs[g])
                phase[g] = REFLECT
                finished.append(g)
            g = groups[FINAL]
            if len(g):
                v, x = take(len(g))
                fbest, xbest = fs[g, order[g, 0]], S[g, order[g, 0]]
                use_c = v < fbest
                xmin[g] = np.where(use_c[:, None], x, xbest)
                fmin[g] = np.where(use_c, v, fbest)
                phase[g] = DONE
            if finished:
                d = np.concatenate(finished)
                it[d] += 1
                stop = (_nmobjective(fs[d]) <= self.g_tol) | (it[d] >= self.iterations)
                phase[d[stop]] = FINAL
        self.iterations_done = it
        return xmin, fmin
