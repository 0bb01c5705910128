"""ctypes binding of the CPU oracle (oracle/libgpcc_oracle.so).

TEST INFRASTRUCTURE ONLY -- see oracle/gpcc_oracle.h.  Imported by tests/, by
__graft_entry__.smoke() and by bench.py's cpu_baseline leg; never by the product
package (gpcc.jl_amd/).  PARITY UNPINNED (no executable reference, no reference golden vectors).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgpcc_oracle.so")

KERNEL_IDS = {"OU": 0, "rbf": 1, "matern32": 2, "matern52": 3}


def build(force=False):
    """Compile the C restatement (gcc) if the .so is missing or stale."""
    srcs = [os.path.join(_HERE, f) for f in ("oracle_model.c", "oracle_chol.c", "gpcc_oracle.h", "Makefile")]
    if not force and os.path.exists(_LIB_PATH):
        so_m = os.path.getmtime(_LIB_PATH)
        if all(os.path.getmtime(s) <= so_m for s in srcs):
            return _LIB_PATH
    subprocess.run(["make", "-C", _HERE, "clean"], check=True, capture_output=True)
    subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        dp = ctypes.POINTER(ctypes.c_double)
        ip = ctypes.POINTER(ctypes.c_int)
        L.gpcc_oracle_kernel.restype = ctypes.c_double
        L.gpcc_oracle_kernel.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double]
        L.gpcc_oracle_covariance.restype = ctypes.c_int
        L.gpcc_oracle_covariance.argtypes = [ctypes.c_int, ctypes.c_int, dp, dp, ctypes.c_double, ip, dp, ip, dp, dp]
        L.gpcc_oracle_precompute.restype = ctypes.c_int
        L.gpcc_oracle_precompute.argtypes = [ctypes.c_int, ip, dp, ctypes.c_int, dp, dp, dp]
        L.gpcc_oracle_model_matrix.restype = ctypes.c_int
        L.gpcc_oracle_model_matrix.argtypes = [ctypes.c_int, ctypes.c_int, ip, dp, dp, dp, ctypes.c_int, dp, dp,
                                               ctypes.c_double, dp, dp]
        L.gpcc_oracle_potrf_lower.restype = ctypes.c_int
        L.gpcc_oracle_potrf_lower.argtypes = [ctypes.c_int, dp, ctypes.c_int]
        L.gpcc_oracle_loglik_batch.restype = ctypes.c_int
        L.gpcc_oracle_loglik_batch.argtypes = [ctypes.c_int, ctypes.c_int, ip, dp, dp, dp, ctypes.c_int, ctypes.c_int,
                                               dp, dp, dp, dp, ip, ctypes.c_int]
        L.gpcc_oracle_probabilities.restype = ctypes.c_int
        L.gpcc_oracle_probabilities.argtypes = [ctypes.c_int, dp, dp, dp]
        L.gpcc_oracle_max_threads.restype = ctypes.c_int
        _lib = L
    return _lib


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _ip(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))


def _flatten(arrs):
    Nl = np.array([len(a) for a in arrs], dtype=np.int32)
    flat = _d(np.concatenate([np.asarray(a, dtype=np.float64) for a in arrs])) if len(arrs) else _d([])
    return Nl, flat


def kernel(name, xi, xj, rho):
    return lib().gpcc_oracle_kernel(KERNEL_IDS[name], float(xi), float(xj), float(rho))


def delayed_covariance(kernel_name, scale, delays, rho, x, y=None):
    """delayedCovariance(kernel, scale, delays, rho, x[, y]) -> (sum Nx, sum Ny) ndarray."""
    if y is None:
        y = x
    Nx, fx = _flatten(x)
    Ny, fy = _flatten(y)
    scale = _d(scale)
    delays = _d(delays)
    L = len(scale)
    assert L == len(x) == len(y) == len(delays)
    out = np.empty((int(Ny.sum()), int(Nx.sum())), dtype=np.float64)  # col-major (Nx, Ny) == C-order transpose
    rc = lib().gpcc_oracle_covariance(KERNEL_IDS[kernel_name], L, _dp(scale), _dp(delays), float(rho),
                                      _ip(Nx), _dp(fx), _ip(Ny), _dp(fy), _dp(out))
    if rc == -1:
        raise AssertionError("all(scale .> 0)")
    if rc == -2:
        raise ValueError("rho=%.8f is <= 0" % rho)
    if rc:
        raise RuntimeError("oracle error %d" % rc)
    return out.T


def model_matrix(kernel_name, tarray, yarray, stdarray, delays, alpha, rho, marginalise_b=True):
    Nl, t = _flatten(tarray)
    _, y = _flatten(yarray)
    _, s = _flatten(stdarray)
    N = int(Nl.sum())
    K = np.empty((N, N), dtype=np.float64)
    resid = np.empty(N, dtype=np.float64)
    delays = _d(delays)
    alpha = _d(alpha)
    rc = lib().gpcc_oracle_model_matrix(KERNEL_IDS[kernel_name], len(Nl), _ip(Nl), _dp(t), _dp(y), _dp(s),
                                        int(bool(marginalise_b)), _dp(delays), _dp(alpha), float(rho), _dp(K),
                                        _dp(resid))
    if rc:
        raise RuntimeError("oracle error %d" % rc)
    return K.T, resid  # K symmetric; .T restores column-major semantics


def loglik_batch(kernel_name, tarray, yarray, stdarray, delays, alpha, rho, marginalise_b=True, nthreads=1):
    """objective(alpha, rho) for M (tau, alpha, rho) triples -> (loglik[M], info[M])."""
    Nl, t = _flatten(tarray)
    _, y = _flatten(yarray)
    _, s = _flatten(stdarray)
    L = len(Nl)
    delays = _d(np.atleast_2d(delays))
    alpha = _d(np.atleast_2d(alpha))
    rho = _d(np.atleast_1d(rho))
    M = len(rho)
    assert delays.shape == (M, L) and alpha.shape == (M, L)
    ll = np.empty(M, dtype=np.float64)
    info = np.zeros(M, dtype=np.int32)
    rc = lib().gpcc_oracle_loglik_batch(KERNEL_IDS[kernel_name], L, _ip(Nl), _dp(t), _dp(y), _dp(s),
                                        int(bool(marginalise_b)), M, _dp(delays), _dp(alpha), _dp(rho), _dp(ll),
                                        _ip(info), int(nthreads))
    if rc:
        raise RuntimeError("oracle error %d" % rc)
    return ll, info


def potrf_lower(A):
    """returns (L, info); A symmetric ndarray."""
    A = np.array(A, dtype=np.float64, order="F")
    n = A.shape[0]
    info = lib().gpcc_oracle_potrf_lower(n, A.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), n)
    return np.tril(A), info


def probabilities(loglik, logprior=None):
    ll = _d(loglik)
    shape = ll.shape
    ll = ll.reshape(-1)
    out = np.empty_like(ll)
    lp = None
    if logprior is not None:
        lpa = _d(logprior).reshape(-1)
        assert lpa.shape == ll.shape
        lp = _dp(lpa)
    rc = lib().gpcc_oracle_probabilities(len(ll), _dp(ll), lp, _dp(out))
    if rc:
        raise RuntimeError("oracle error %d" % rc)
    return out.reshape(shape)


def max_threads():
    return lib().gpcc_oracle_max_threads()
