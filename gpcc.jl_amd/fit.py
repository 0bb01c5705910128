"""gpcc(): the per-delay model fit of the reference (src/gpccfixdelay_marginaliseb.jl:46-53, :56-352),
host logic in Python over the device objective.  A whole grid of candidate delays is fitted in
lock-step (neldermead.BatchedNelderMead): every optimiser round is ONE gpcc_loglik_batch call.

Restated from the reference: parameter packing `unpack` (:112-126), initial rho values (:160-176),
`sampleα` (:188), `sampleunconstrainedsolution` (:195-196), `getsolution` (:203-215: the best of
`initialrandom` random candidates starts Nelder-Mead), restarts (:222-226), returned value
`-result.minimum` (:351).

Not reproducible here (no Julia): MersenneTwister's stream (numpy's PCG64 is used; like the reference,
every delay of a grid sees the SAME random draws because each gpcc call seeds its own generator with
`seed`), Optim's exact trajectory, and MiscUtil's transforms, whose source is not under
/root/reference: `makepositive` is taken to be softplus and `transformbetween(x, a, b)` to be
a + (b - a) * logistic(x), the package's documented purpose; `safewrapper` is taken to turn exceptions
(PosDefException) into +Inf of the negative objective."""
import numpy as np

from .api import Objective, PosDefException, mvnormal_logpdf
from .neldermead import BatchedNelderMead


def makepositive(x):
    x = np.asarray(x, dtype=np.float64)
    return np.where(x > 30.0, x, np.log1p(np.exp(np.minimum(x, 30.0))))


def invmakepositive(y):
    y = np.asarray(y, dtype=np.float64)
    return np.where(y > 30.0, y, np.log(np.expm1(np.minimum(y, 30.0))))


def transformbetween(x, a, b):
    x = np.asarray(x, dtype=np.float64)
    return a + (b - a) / (1.0 + np.exp(-x))


def invtransformbetween(y, a, b):
    u = (np.asarray(y, dtype=np.float64) - a) / (b - a)
    return np.log(u) - np.log1p(-u)


def nearestposdef(A, minimumeigenvalue=1e-6):
    """MiscUtil.nearestposdef(A; minimumeigenvalue) as used at marginaliseb.jl:331 -- MiscUtil's source is not
    available (unregistered dependency), so this is the assumed definition: symmetrise, eigendecompose, lift every
    eigenvalue below `minimumeigenvalue` up to it, recompose, symmetrise.  Host-side like the reference's (it only
    runs after a PosDefException on an Ntest x Ntest predictive covariance, never on the delay-grid path)."""
    A = np.asarray(A, dtype=np.float64)
    S = 0.5 * (A + A.T)
    w, V = np.linalg.eigh(S)
    R = (V * np.maximum(w, minimumeigenvalue)) @ V.T
    return 0.5 * (R + R.T)


class UniformDelayPrior:
    """Uniform(0, upper) over the delay, with the log-density vector getprobabilities takes as its second argument."""

    def __init__(self, upper):
        self.lower, self.upper = 0.0, float(upper)

    def logpdf(self, delays):
        d = np.asarray(delays, dtype=np.float64)
        return np.where((d >= self.lower) & (d <= self.upper), -np.log(self.upper - self.lower), -np.inf)


def uniformpriordelay(*, L, z):
    """src/uniformpriordelay.jl:10-16: Uniform(0, 10^1.559 (L 1e-44)^0.549 (1+z)); L luminosity, z redshift."""
    return UniformDelayPrior(10.0 ** 1.559 * (L * 1e-44) ** 0.549 * (1.0 + z))


def logrange(a, b, n):
    return np.exp(np.linspace(np.log(a), np.log(b), n))


class GridFit:
    """Result of gpcc_grid: loglikel[G] (= -minimum, what README.md:172-174 feeds to
    getprobabilities), alpha[G, L], rho[G], f_calls, rounds."""

    def __init__(self, loglikel, alpha, rho, f_calls, rounds, iterations_done):
        self.loglikel, self.alpha, self.rho = loglikel, alpha, rho
        self.f_calls, self.rounds, self.iterations_done = f_calls, rounds, iterations_done


def gpcc_grid(tarray, yarray, stdarray, *, kernel, candidatedelays, iterations, seed=1, numberofrestarts=1,
              initialrandom=5, rhomin=0.1, rhomax=20.0, objective=None, device=0, marginalise_b=True, engine=None,
              unpack=None):
    """Fits the GPCC model for each row of candidatedelays (G, L): the README's
    `map(delay -> gpcc(...; delays = [0; delay])[1], candidatedelays)` as one lock-step batch.

    engine "native" (default with a device Objective): the whole fit is one gpcc_grid_loglik call, the random
    candidates drawn here (numpy) and handed over as init_params.  engine "python" (default when another
    objective is injected): the same algorithm in numpy (neldermead.py) over objective.loglik_batch; `unpack`
    may replace the numpy parameter transforms (api.unpack_params = the library's own)."""
    cand = np.ascontiguousarray(np.atleast_2d(candidatedelays), dtype=np.float64)
    G, L = cand.shape
    assert L == len(tarray) == len(yarray) == len(stdarray)          # marginaliseb.jl:78
    own = objective is None
    obj = objective if objective is not None else Objective(tarray, yarray, stdarray, kernel,
                                                            marginalise_b=marginalise_b, device=device)
    try:
        R = int(numberofrestarts)
        rg = np.random.default_rng(seed)
        if R in (1, 2):                                                  # :160-176
            rho0 = rg.uniform(rhomin + 1e-3, rhomax - 1e-3, R)
        else:
            rho0 = logrange(rhomin + 1e-3, rhomax - 1e-3, R)
        vary = np.array([np.var(np.asarray(y, dtype=np.float64), ddof=1) for y in yarray])
        # the same candidates for every delay (each reference call re-seeds): (R, initialrandom, L+1)
        cands = np.empty((R, initialrandom, L + 1))
        for i in range(R):
            for c in range(initialrandom):
                cands[i, c, :L] = invmakepositive(vary * (rg.random(L) * (1.2 - 0.8) + 0.8))   # sampleα, :188
                cands[i, c, L] = invtransformbetween(rho0[i], rhomin, rhomax)
        P = G * R                                                        # problem p = (delay p // R, restart p % R)
        if engine is None:
            engine = "native" if isinstance(obj, Objective) else "python"
        if engine == "native":
            ll, alpha, rho, info, its, (f_calls, rounds) = obj.grid_loglik(
                cand, iterations, numberofrestarts=R, initialrandom=initialrandom, rhomin=rhomin, rhomax=rhomax,
                seed=seed, init_params=cands)
            return GridFit(ll, alpha, rho, f_calls, rounds, its.astype(np.int64))

        def negobj(pidx, X):
            if unpack is not None:
                alpha, rho = unpack(X, L, rhomin, rhomax)
            else:
                alpha = makepositive(X[:, :L]) + 1e-8                    # makeα, :112
                rho = transformbetween(X[:, L], rhomin, rhomax)          # makeρ, :114
            ll, info = obj.loglik_batch(cand[pidx // R], alpha, rho)
            return np.where(info == 0, -ll, np.inf)                      # safewrapper(negativeobjective), :149-153

        # argmin over the random candidates (:209)
        pid = np.repeat(np.arange(P), initialrandom)
        Xc = cands[np.tile(np.repeat(np.arange(R), initialrandom), G), np.tile(np.arange(initialrandom), P)]
        f0 = negobj(pid, Xc).reshape(P, initialrandom)
        best = np.argmin(f0, axis=1)
        x0 = Xc.reshape(P, initialrandom, L + 1)[np.arange(P), best]
        nm = BatchedNelderMead(x0, negobj, iterations=iterations, g_tol=1e-6)   # Optim.Options(g_tol = 1e-6), :205
        xmin, fmin = nm.run()
        fmin = fmin.reshape(G, R)
        pick = np.argmin(fmin, axis=1)                                   # best restart, :224
        xsel = xmin.reshape(G, R, L + 1)[np.arange(G), pick]
        a_sel, r_sel = (unpack(xsel, L, rhomin, rhomax) if unpack is not None else
                        (makepositive(xsel[:, :L]) + 1e-8, transformbetween(xsel[:, L], rhomin, rhomax)))
        return GridFit(-fmin[np.arange(G), pick], a_sel, r_sel, nm.f_calls + P * initialrandom, nm.rounds + 1,
                       nm.iterations_done.reshape(G, R)[np.arange(G), pick])
    finally:
        if own:
            obj.close()


class Predictor:
    """The `predictTest` closure returned by gpcc() (marginaliseb.jl:259-343), three call forms:
      pred(ttest)                       ttest = list of L arrays  -> (mu_pred, Sigma_pred), joint    (:259-289)
      pred(ttest)                       ttest = one array / range -> (mu per band, sigma per band)   (:293-307)
      pred(ttest, ytest, sigmatest)     lists of L arrays         -> test log-likelihood             (:311-343)
    All linear algebra runs on the device (Objective.predict / mvnormal_logpdf)."""

    def __init__(self, objective, delays, alpha, rho):
        self.obj, self.delays, self.alpha, self.rho = objective, np.array(delays, float), np.array(alpha, float), float(rho)

    def __call__(self, ttest, ytest=None, sigmatest=None):
        L = self.obj.L
        joint = isinstance(ttest, (list, tuple)) and len(ttest) == L and all(np.ndim(a) == 1 for a in ttest)
        if ytest is not None:
            mu, Sig = self.obj.predict(self.delays, self.alpha, self.rho, ttest)
            s2 = np.concatenate([np.asarray(a, dtype=np.float64) for a in sigmatest]) ** 2
            Sig = Sig + np.diag(s2)                                       # + Sobs*, :317-319
            yv = np.concatenate([np.asarray(a, dtype=np.float64) for a in ytest])
            try:
                return mvnormal_logpdf(mu, Sig, yv, device=self.obj.device)
            except PosDefException:
                # :327-341 -- retry once on nearestposdef(Sigma; minimumeigenvalue = 1e-6); a second
                # PosDefException propagates, as in the reference
                return mvnormal_logpdf(mu, nearestposdef(Sig, minimumeigenvalue=1e-6), yv, device=self.obj.device)
        if joint:
            return self.obj.predict(self.delays, self.alpha, self.rho, [np.asarray(a, dtype=np.float64) for a in ttest])
        tt = np.asarray(ttest, dtype=np.float64).ravel()
        n = len(tt)
        mu, Sig = self.obj.predict(self.delays, self.alpha, self.rho, [tt] * L)
        d = np.diag(Sig)
        return ([mu[l * n:(l + 1) * n] for l in range(L)],
                [np.sqrt(np.maximum(d[l * n:(l + 1) * n], 1e-6)) for l in range(L)])      # :301-303


def gpcc(tarray, yarray, stdarray, *, kernel, delays, iterations, seed=1, numberofrestarts=1, initialrandom=5,
         rhomin=0.1, rhomax, device=0):
    """loglikel, pred, (alpha, postb, rho) = gpcc(tarray, yarray, stdarray; kernel, delays, iterations, ...)
    -- src/gpccfixdelay_marginaliseb.jl:46-53.  postb is returned as (mu_postb, Sigma_postb), the
    parameters of the reference's MvNormal (:252)."""
    delays = np.asarray(delays, dtype=np.float64)
    assert len(delays) == len(tarray) == len(yarray) == len(stdarray)              # :78
    obj = Objective(tarray, yarray, stdarray, kernel, marginalise_b=True, device=device)
    res = gpcc_grid(tarray, yarray, stdarray, kernel=kernel, candidatedelays=delays[None, :], iterations=iterations,
                    seed=seed, numberofrestarts=numberofrestarts, initialrandom=initialrandom, rhomin=rhomin,
                    rhomax=rhomax, objective=obj)
    alpha, rho = res.alpha[0], float(res.rho[0])
    postb = obj.posterior_offsets(delays, alpha, rho)
    return float(res.loglikel[0]), Predictor(obj, delays, alpha, rho), (alpha, postb, rho)


def singlegp(tobs, yobs, sigmaobs, *, kernel, iterations, seed=1, numberofrestarts=1, initialrandom=5, rhomin=0.1,
             rhomax, device=0):
    """src/util.jl:95-99: one band, delay [0.0] -- gpccfixdelay([tobs], [yobs], [sigmaobs]; tau = [0.0], ...)."""
    return gpcc([tobs], [yobs], [sigmaobs], kernel=kernel, delays=[0.0], iterations=iterations, seed=seed,
                numberofrestarts=numberofrestarts, initialrandom=initialrandom, rhomin=rhomin, rhomax=rhomax, device=device)
