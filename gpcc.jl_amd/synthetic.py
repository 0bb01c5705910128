"""Synthetic irregularly-sampled light curves (numpy only, host side).

Restates the statistical recipe of the reference's generator
(/root/reference/src/simulatedata.jl:96-162) at arbitrary size, as SURVEY.md section 8(d)
prescribes: per band, times ~ U(0, T) i.i.d. and UNSORTED with T = N_l/3 days (the reference
has 60 observations on [0, 20], simulatedata.jl:119-121); a latent OU process with rho = 3.5
drawn exactly by its AR(1) recursion on the merged, delay-shifted, sorted time axis (instead of
the reference's O(N^3) MvNormal draw, :128-145); y_l = alpha_l^2 f(t_l - tau_l) + b_l + sigma eps
(alpha enters twice in the reference: inside C at :128 and again at :153);
alpha = (1, 1.5, 2), b = (6, 15, 25), true delays (0, 2, 4), sigma = 0.75 (:105-111, :153-159).
The reference's MersenneTwister stream cannot be reproduced without Julia; numpy's PCG64 is used.
"""
import numpy as np

TRUE_DELAYS = (0.0, 2.0, 4.0)
ALPHAS = (1.0, 1.5, 2.0)
OFFSETS = (6.0, 15.0, 25.0)
RHO_TRUE = 3.5


def simulate_lightcurves(Nl, seed=1, sigma=0.75, rho=RHO_TRUE, gap_band=None, span=None):
    """Returns (tarray, yarray, stdarray, truedelays) as lists of float64 arrays, band order kept.

    gap_band: optional band index that gets the reference's mid-gap pattern
    (half the points in [0, 0.4T], half in [0.6T, T]; simulatedata.jl:121).
    """
    Nl = [int(n) for n in Nl]
    L = len(Nl)
    if L < 1 or L > len(TRUE_DELAYS):
        raise ValueError("1 <= number of bands <= %d" % len(TRUE_DELAYS))
    rng = np.random.default_rng(seed)
    tarray = []
    for l, n in enumerate(Nl):
        T = span if span is not None else n / 3.0
        if gap_band is not None and l == gap_band:
            h = n // 2
            t = np.concatenate([rng.random(h) * 0.4 * T, 0.6 * T + rng.random(n - h) * 0.4 * T])
        else:
            t = rng.random(n) * T
        tarray.append(t)
    # latent OU on the merged shifted axis, exact AR(1): f(s_k) = phi f(s_{k-1}) + sqrt(1-phi^2) e
    u = np.concatenate([tarray[l] - TRUE_DELAYS[l] for l in range(L)])
    order = np.argsort(u, kind="stable")
    us = u[order]
    e = rng.standard_normal(len(us))
    f_sorted = np.empty_like(us)
    f_sorted[0] = e[0]
    phi = np.exp(-np.diff(us) / rho)
    sd = np.sqrt(np.maximum(0.0, 1.0 - phi * phi))
    for k in range(1, len(us)):
        f_sorted[k] = phi[k - 1] * f_sorted[k - 1] + sd[k - 1] * e[k]
    f = np.empty_like(us)
    f[order] = f_sorted
    yarray, stdarray = [], []
    off = 0
    for l, n in enumerate(Nl):
        fl = f[off:off + n]
        off += n
        y = ALPHAS[l] ** 2 * fl + OFFSETS[l] + sigma * rng.standard_normal(n)
        yarray.append(y)
        stdarray.append(np.full(n, sigma))
    return tarray, yarray, stdarray, list(TRUE_DELAYS[:L])


def default_hyperparameters(yarray):
    """Fixed hyper-parameters for fixed-hyper sweeps: alpha_l = var(y_l)
    (centre of sampleα, gpccfixdelay_marginaliseb.jl:188), rho = 3.5."""
    return np.array([np.var(y, ddof=1) for y in yarray]), RHO_TRUE
