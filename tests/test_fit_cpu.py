"""Host logic of gpcc_grid (lock-step fit over a delay grid) with the CPU oracle injected as the
objective (tests may use the oracle; the product path uses gpcc_amd.Objective = the HIP library)."""
import numpy as np

from gpcc_amd import fit, synthetic


class OracleObjective:
    """Stand-in with Objective's loglik_batch signature."""

    def __init__(self, oracle, kernel, t, y, s):
        self.o, self.k, self.t, self.y, self.s = oracle, kernel, t, y, s
        self.calls = 0

    def loglik_batch(self, delays, alpha, rho):
        self.calls += 1
        return self.o.loglik_batch(self.k, self.t, self.y, self.s, delays, alpha, rho, True, nthreads=8)


def test_transforms_roundtrip():
    x = np.linspace(-20, 40, 13)
    assert np.allclose(fit.invmakepositive(fit.makepositive(x)), x, atol=1e-9)
    r = fit.transformbetween(x, 0.1, 20.0)
    assert np.all((r >= 0.1) & (r <= 20.0)) and np.allclose(fit.invtransformbetween(r[3:9], 0.1, 20.0), x[3:9])
    assert np.allclose(fit.logrange(0.101, 19.999, 4)[[0, -1]], [0.101, 19.999])


def test_uniformpriordelay():
    pr = fit.uniformpriordelay(L=1e44, z=0.0)
    assert abs(pr.upper - 10.0 ** 1.559) < 1e-12                     # L = 1e44, z = 0: the bare constant
    pr = fit.uniformpriordelay(L=6e43, z=0.0258)
    lp = pr.logpdf([-1.0, 0.0, 5.0, pr.upper, pr.upper + 1e-9])
    assert np.isneginf(lp[[0, 4]]).all() and np.allclose(lp[1:4], -np.log(pr.upper))


def test_nearestposdef_properties():
    rng = np.random.default_rng(3)
    A = rng.standard_normal((12, 12))
    S = A + A.T                                         # indefinite
    P = fit.nearestposdef(S, minimumeigenvalue=1e-6)
    assert np.array_equal(P, P.T) and np.linalg.eigvalsh(P).min() >= 1e-6 * (1 - 1e-6)
    w, V = np.linalg.eigh(S)
    assert np.allclose(V.T @ P @ V, np.diag(np.maximum(w, 1e-6)), atol=1e-10)    # same eigenvectors, lifted spectrum
    G = A @ A.T + np.eye(12)                            # already positive definite: unchanged
    assert np.allclose(fit.nearestposdef(G), G, rtol=0, atol=1e-11)


def test_gpcc_grid_lockstep_equals_single_delay_runs(oracle):
    t, y, s, _ = synthetic.simulate_lightcurves([40, 30], seed=5, span=20.0)
    cand = np.stack([np.zeros(4), np.array([0.5, 2.0, 3.5, 9.0])], 1)
    obj = OracleObjective(oracle, "matern32", t, y, s)
    res = fit.gpcc_grid(t, y, s, kernel="matern32", candidatedelays=cand, iterations=40, rhomax=20.0, objective=obj)
    assert res.loglikel.shape == (4,) and res.alpha.shape == (4, 2) and np.all(res.alpha > 0)
    assert np.all((res.rho > 0.1) & (res.rho < 20.0)) and obj.calls == res.rounds
    for g in range(4):   # every delay follows the trajectory it follows alone (same seed => same start)
        one = fit.gpcc_grid(t, y, s, kernel="matern32", candidatedelays=cand[[g]], iterations=40, rhomax=20.0,
                            objective=OracleObjective(oracle, "matern32", t, y, s))
        assert one.loglikel[0] == res.loglikel[g] and np.array_equal(one.alpha[0], res.alpha[g])
    # the optimiser improves on the best random start and the returned value is objective(alpha, rho)
    ll, info = oracle.loglik_batch("matern32", t, y, s, cand, res.alpha, res.rho, True)
    assert np.allclose(ll, res.loglikel, rtol=1e-12)


def test_gpcc_grid_restarts_and_posterior_mode(oracle):
    t, y, s, _ = synthetic.simulate_lightcurves([60, 50], seed=1, gap_band=1, span=20.0)
    grid = np.arange(0.0, 8.01, 1.0)
    cand = np.stack([np.zeros_like(grid), grid], 1)
    obj = OracleObjective(oracle, "OU", t, y, s)
    r1 = fit.gpcc_grid(t, y, s, kernel="OU", candidatedelays=cand, iterations=120, rhomax=20.0, objective=obj)
    r3 = fit.gpcc_grid(t, y, s, kernel="OU", candidatedelays=cand, iterations=120, rhomax=20.0, objective=obj,
                       numberofrestarts=3)
    assert np.all(r3.loglikel >= r1.loglikel - 2.0)    # more restarts do not make the fit much worse
    p = oracle.probabilities(r3.loglikel)
    assert abs(grid[np.argmax(p)] - 2.0) <= 1.0          # true delay 2.0 (README.md:156-179, qualitative)
