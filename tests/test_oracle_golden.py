"""The C oracle (oracle/*.c) against the independent numpy/scipy restatement's fixtures
(tests/golden/make_golden.py).  Two restatements of /root/reference/src sharing no code must agree
to <= 1e-12 relative -- the only pin available while the reference itself is not executable."""
import numpy as np
import pytest


def test_loglik_cases(golden, oracle):
    assert len(golden["cases"]) >= 40
    for c in golden["cases"]:
        ll, info = oracle.loglik_batch(c["kernel"], c["t"], c["y"], c["sigma"], [c["delays"]], [c["alpha"]],
                                       [c["rho"]], c["marginalise_b"])
        assert info[0] == c["info"] == 0
        assert abs(ll[0] - c["loglik"]) <= 1e-12 * abs(c["loglik"]), (c["kernel"], c["marginalise_b"])


def test_loglik_batched_threads_match_serial(golden, oracle):
    c = golden["cases"][10]
    M = 6
    rng = np.random.default_rng(0)
    L = len(c["t"])
    delays = rng.random((M, L)) * 5
    alpha = 0.5 + rng.random((M, L))
    rho = 0.5 + 3 * rng.random(M)
    a, ia = oracle.loglik_batch(c["kernel"], c["t"], c["y"], c["sigma"], delays, alpha, rho, True, nthreads=1)
    b, ib = oracle.loglik_batch(c["kernel"], c["t"], c["y"], c["sigma"], delays, alpha, rho, True, nthreads=4)
    assert np.array_equal(a, b) and np.array_equal(ia, ib)


def test_covariances(golden, oracle):
    for c in golden["covariances"]:
        Kxy = oracle.delayed_covariance(c["kernel"], c["scale"], c["delays"], c["rho"], c["x"], c["y"])
        Kxx = oracle.delayed_covariance(c["kernel"], c["scale"], c["delays"], c["rho"], c["x"])
        np.testing.assert_allclose(Kxy, np.array(c["Kxy"]), rtol=1e-14, atol=1e-16)
        np.testing.assert_allclose(Kxx, np.array(c["Kxx"]), rtol=1e-14, atol=1e-16)
        assert Kxy.shape == (7, 7) and np.array_equal(Kxx, Kxx.T)


def test_nonpd(golden, oracle):
    c = golden["nonpd"]
    ll, info = oracle.loglik_batch(c["kernel"], c["t"], c["y"], c["sigma"], [c["delays"]], [c["alpha"]],
                                   [c["rho"]], c["marginalise_b"])
    assert info[0] > 0 and np.isnan(ll[0])


def test_probabilities(golden, oracle):
    p = golden["probabilities"]
    np.testing.assert_allclose(oracle.probabilities(p["loglik"]), p["p_flat"], rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose(oracle.probabilities(p["loglik"], p["logprior"]), p["p_prior"], rtol=1e-12,
                               atol=1e-300)


def test_argument_errors(oracle):
    x = [[0.0, 1.0], [0.5]]
    with pytest.raises(AssertionError):   # delayedCovariance.jl:3
        oracle.delayed_covariance("OU", [1.0, 0.0], [0.0, 0.0], 1.0, x)
    with pytest.raises(ValueError):       # delayedCovariance.jl:5-7
        oracle.delayed_covariance("OU", [1.0, 1.0], [0.0, 0.0], 0.0, x)
