#!/usr/bin/env python3
"""Generates tests/golden/gpcc_golden.json.

The reference (/root/reference, pure Julia) cannot be executed in this image and holds no
golden vectors (test/runtests.jl:4-6 is empty), so these fixtures are minted by an INDEPENDENT
numpy/scipy restatement of the path -- vectorised broadcasting + LAPACK (scipy.linalg) +
scipy.special.logsumexp, sharing no code with oracle/*.c -- and the C oracle is then required to
agree with them to <= 1e-12 relative (tests/test_oracle_golden.py).  PARITY UNPINNED against the
real reference; see DESIGN.md "Oracle".

Restated from: src/util.jl:15-52 (kernels), src/delayedCovariance.jl:1-38,
src/gpccfixdelay_marginaliseb.jl:85-98,133-141, src/gpccfixdelay.jl:85-96,131-139,
src/getprobabilities.jl:1-20.

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os

import numpy as np
import scipy.linalg as sla
from scipy.special import logsumexp

HERE = os.path.dirname(os.path.abspath(__file__))


def k_OU(xi, xj, rho):
    return np.exp(-np.abs(xi - xj) / rho)


def k_rbf(xi, xj, rho):
    return np.exp(-0.5 * (xi - xj) ** 2 / (2 * rho))


def k_matern32(xi, xj, rho):
    r = np.abs(xi - xj)
    return (1 + np.sqrt(3.0) * r / rho) * np.exp(-np.sqrt(3.0) * r / rho)


def k_matern52(xi, xj, rho):
    r = np.abs(xi - xj)
    return (1 + np.sqrt(5.0) * r / rho + (5 * r ** 2) / (3 * rho ** 2)) * np.exp(-np.sqrt(5.0) * r / rho)


KERNELS = {"OU": k_OU, "rbf": k_rbf, "matern32": k_matern32, "matern52": k_matern52}


def delayed_covariance(kname, scale, delays, rho, x, y=None):
    if y is None:
        y = x
    assert all(s > 0 for s in scale)
    if rho <= 0:
        raise ValueError("rho <= 0")
    k = KERNELS[kname]
    rows = []
    for i in range(len(scale)):
        rows.append([scale[i] * scale[j] * k((x[i] - delays[i])[:, None], (y[j] - delays[j])[None, :], rho)
                     for j in range(len(scale))])
    return np.block(rows)


def objective(kname, t, y, s, delays, alpha, rho, marginalise_b):
    L = len(t)
    Nl = [len(a) for a in t]
    Y = np.concatenate(y)
    Q = np.zeros((sum(Nl), L))
    o = 0
    for l, n in enumerate(Nl):
        Q[o:o + n, l] = 1.0
        o += n
    Sobs = np.diag(np.concatenate(s) ** 2)
    mub = np.array([np.mean(a) for a in y])
    K = delayed_covariance(kname, alpha, delays, rho, t) + Sobs
    if marginalise_b:
        Sigb = 100 * np.diag([np.var(a, ddof=1) for a in y])
        K = K + Q @ Sigb @ Q.T
        bbar = Q @ mub
    else:
        b = np.linalg.solve(Q.T @ Q, Q.T @ Y)
        bbar = Q @ b
    K = (K + K.T) / 2
    try:
        c = sla.cholesky(K, lower=True)
    except np.linalg.LinAlgError:
        return float("nan"), 1
    z = sla.solve_triangular(c, Y - bbar, lower=True)
    N = len(Y)
    ll = -0.5 * (N * np.log(2 * np.pi) + 2 * np.sum(np.log(np.diag(c)))) - 0.5 * (z @ z)
    return float(ll), 0


def getprobabilities(ll, logprior=None):
    ll = np.asarray(ll, dtype=float)
    joint = ll + (np.ones_like(ll) if logprior is None else np.asarray(logprior, dtype=float))
    return np.exp(joint - logsumexp(joint))


def make_data(rng, Nl):
    """simulatedata-flavoured draw (not the reference's stream): unsorted times, offsets, noise."""
    t = [rng.random(n) * 20.0 for n in Nl]
    if len(Nl) > 1 and Nl[1] >= 4:
        h = Nl[1] // 2
        t[1] = np.concatenate([rng.random(h) * 8.0, 12.0 + rng.random(Nl[1] - h) * 8.0])
    base = [6.0, 15.0, 25.0]
    y = [base[l] + (1.0 + 0.5 * l) * np.sin(0.3 * (t[l] - 2.0 * l)) + 0.5 * rng.standard_normal(n)
         for l, n in enumerate(Nl)]
    s = [0.3 + 0.6 * rng.random(n) for n in Nl]
    return t, y, s


def main():
    rng = np.random.default_rng(20240917)
    cases = []
    shapes = [[37], [60, 50], [60, 50, 40], [5, 3], [128, 72], [1, 1]]
    for Nl in shapes:
        t, y, s = make_data(rng, Nl)
        L = len(Nl)
        for kname in KERNELS:
            for mb in (True, False):
                if min(Nl) < 2 and mb:
                    continue  # var(y_l) undefined (n-1 = 0)
                delays = np.concatenate([[0.0], rng.random(L - 1) * 6.0])
                alpha = 0.5 + 2.0 * rng.random(L)
                rho = 0.5 + 6.0 * rng.random()
                ll, info = objective(kname, t, y, s, delays, alpha, rho, mb)
                cases.append(dict(kernel=kname, marginalise_b=mb, t=[a.tolist() for a in t],
                                  y=[a.tolist() for a in y], sigma=[a.tolist() for a in s],
                                  delays=delays.tolist(), alpha=alpha.tolist(), rho=float(rho),
                                  loglik=ll, info=info))
    # covariance fixtures (rectangular x != y too), small enough to store whole
    covs = []
    for kname in KERNELS:
        x = [rng.random(4) * 10, rng.random(3) * 10]
        yy = [rng.random(2) * 10, rng.random(5) * 10]
        scale = 0.5 + rng.random(2)
        delays = np.array([0.0, 1.7])
        rho = 2.3
        covs.append(dict(kernel=kname, x=[a.tolist() for a in x], y=[a.tolist() for a in yy],
                         scale=scale.tolist(), delays=delays.tolist(), rho=rho,
                         Kxy=delayed_covariance(kname, scale, delays, rho, x, yy).tolist(),
                         Kxx=delayed_covariance(kname, scale, delays, rho, x).tolist()))
    # a deliberately non-positive-definite input: duplicate time, sigma = 0, b-term off
    t = [np.array([1.0, 1.0, 2.5]), np.array([0.3, 4.0])]
    y = [np.array([1.0, 2.0, 0.5]), np.array([3.0, 3.5])]
    s = [np.zeros(3), np.zeros(2)]
    ll, info = objective("OU", t, y, s, [0.0, 0.0], [1.0, 1.0], 2.0, False)
    assert info != 0
    nonpd = dict(kernel="OU", marginalise_b=False, t=[a.tolist() for a in t], y=[a.tolist() for a in y],
                 sigma=[a.tolist() for a in s], delays=[0.0, 0.0], alpha=[1.0, 1.0], rho=2.0, info_positive=True)
    # getprobabilities
    ll = rng.standard_normal(17) * 30 - 400
    lp = rng.standard_normal(17)
    probs = dict(loglik=ll.tolist(), logprior=lp.tolist(), p_flat=getprobabilities(ll).tolist(),
                 p_prior=getprobabilities(ll, lp).tolist())
    out = dict(note="independent numpy/scipy restatement; reference not executable (no Julia); parity unpinned",
               cases=cases, covariances=covs, nonpd=nonpd, probabilities=probs)
    with open(os.path.join(HERE, "gpcc_golden.json"), "w") as f:
        json.dump(out, f)
    print("wrote %d loglik cases, %d covariance cases" % (len(cases), len(covs)))


if __name__ == "__main__":
    main()
