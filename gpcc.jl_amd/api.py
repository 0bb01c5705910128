"""Host-side mirror of the reference's interface for the hot path (Python, because the image has
no Julia; INTEGRATION.md carries the Julia `ccall` shim over the same C ABI).

Names and argument meaning follow /root/reference/src:
  delayedCovariance(kernel, scale, delays, rho, x[, y])   src/delayedCovariance.jl:1-38
  getprobabilities(loglikel[, logpriorpdfvalues])         src/getprobabilities.jl:1-20
  Objective(tarray, yarray, stdarray; kernel)             the closure objective(alpha, rho) of
                                                          src/gpccfixdelay_marginaliseb.jl:133-141
                                                          (marginalise_b=False: src/gpccfixdelay.jl:131-139)
Every numeric result comes from libgpcc_hip.so (HIP kernels); nothing here computes on the CPU.
"""
import ctypes

import numpy as np

from . import _capi
from ._capi import GpccError, c_double_p, c_int_p


class Kernel:
    """Identity token for one of the reference's kernel functions (GPCC.OU, GPCC.rbf,
    GPCC.matern32, GPCC.matern52; src/util.jl:15-52).  The reference accepts any Julia callable;
    only these four have device implementations."""

    def __init__(self, name):
        self.name = name
        self.id = _capi.KERNEL_IDS[name]

    def __repr__(self):
        return "GPCC.%s" % self.name


OU = Kernel("OU")
rbf = Kernel("rbf")
matern32 = Kernel("matern32")
matern52 = Kernel("matern52")
KERNELS = {k.name: k for k in (OU, rbf, matern32, matern52)}


class PosDefException(np.linalg.LinAlgError):
    """Julia's LinearAlgebra.PosDefException(info): raised by Objective.__call__ like cholesky(K)
    does inside MvNormal(b, K) (marginaliseb.jl:139)."""

    def __init__(self, info):
        super().__init__("matrix is not positive definite; Cholesky factorization failed (info=%d)" % info)
        self.info = info


def _kernel(kernel):
    if isinstance(kernel, Kernel):
        return kernel
    if isinstance(kernel, str) and kernel in KERNELS:
        return KERNELS[kernel]
    raise TypeError("kernel must be one of gpcc_amd.OU / rbf / matern32 / matern52 (got %r); other callables "
                    "have no device implementation" % (kernel,))


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _dp(a):
    return a.ctypes.data_as(c_double_p)


def _ip(a):
    return a.ctypes.data_as(c_int_p)


def _flatten(arrays):
    Nl = np.array([len(a) for a in arrays], dtype=np.int32)
    flat = _d(np.concatenate([np.asarray(a, dtype=np.float64).ravel() for a in arrays])) if len(arrays) else _d([])
    return Nl, flat


def _raise_reference_error(err):
    """Maps the C ABI's argument errors back to the exceptions the Julia code raises."""
    if "AssertionError" in err.message:
        raise AssertionError("all(scale .> 0)") from None    # @assert, delayedCovariance.jl:3
    if "is <= 0" in err.message:
        raise ValueError(err.message) from None               # error(...), delayedCovariance.jl:5-7
    raise err


def delayedCovariance(kernel, scale, delays, rho, x, y=None, device=0):
    """delayedCovariance(kernel, scale, delays, rho, x[, y]) -> (sum Nx, sum Ny) ndarray."""
    kernel = _kernel(kernel)
    if y is None:
        y = x
    scale, delays = _d(scale), _d(delays)
    L = len(scale)
    assert L == len(x) == len(y), "L == length(x) == length(y)"   # delayedCovariance.jl:11
    assert len(delays) == L
    Nx, fx = _flatten(x)
    Ny, fy = _flatten(y)
    out = np.empty((int(Ny.sum()), int(Nx.sum())), dtype=np.float64)  # column-major (Nx, Ny)
    try:
        _capi.check(_capi.load().gpcc_covariance(kernel.id, L, _dp(scale), _dp(delays), float(rho), _ip(Nx), _dp(fx),
                                                 _ip(Ny), _dp(fy), _dp(out), int(device)))
    except GpccError as e:
        _raise_reference_error(e)
    return out.T


def getprobabilities(loglikel, logpriorpdfvalues=None, device=0):
    """exp.(joint .- logsumexp(joint)), joint = loglikel .+ logprior; same shape as the input.
    The 1-argument form uses a log-prior of ones like the reference (getprobabilities.jl:3)."""
    ll = _d(loglikel)
    shape = ll.shape
    ll = ll.reshape(-1)
    lp = None
    if logpriorpdfvalues is not None:
        lpa = _d(logpriorpdfvalues)
        if lpa.shape != shape:
            raise ValueError("loglikel and logpriorpdfvalues differ in shape")
        lpa = lpa.reshape(-1)
        lp = _dp(lpa)
    out = np.empty_like(ll)
    _capi.check(_capi.load().gpcc_probabilities(len(ll), _dp(ll), lp, _dp(out), int(device)))
    return out.reshape(shape)


def mvnormal_logpdf(mu, Sigma, x, device=0):
    """logpdf(MvNormal(mu, Sigma), x) with an explicit covariance, on the device (marginaliseb.jl:325);
    raises PosDefException when the Cholesky factorisation fails."""
    Sigma = np.ascontiguousarray(np.asarray(Sigma, dtype=np.float64).T)   # column-major for the ABI
    n = Sigma.shape[0]
    mu, x = _d(mu), _d(x)
    assert Sigma.shape == (n, n) and mu.shape == (n,) and x.shape == (n,)
    ll, info = ctypes.c_double(0.0), ctypes.c_int(0)
    _capi.check(_capi.load().gpcc_mvnormal_logpdf(n, _dp(Sigma), _dp(mu), _dp(x), ctypes.byref(ll), ctypes.byref(info),
                                                  int(device)))
    if info.value > 0:
        raise PosDefException(info.value)
    return ll.value


class Objective:
    """The marginal log-likelihood objective(alpha, rho) of gpccfixdelay, bound to one data set and
    living on one GPU -- or, with devices=[...], replicated on several GPUs of this process (gpcc_create_multi:
    batches are sharded in contiguous blocks, one all-gather collects them).  The delay vector is an argument
    (the reference captures tau in the closure; a grid sweep varies it), so one handle serves a whole delay grid."""

    def __init__(self, tarray, yarray, stdarray, kernel, marginalise_b=True, precision="fp64", device=0,
                 streams=None, slots_per_stream=None, devices=None):
        self._h = None
        self.kernel = _kernel(kernel)
        L = len(tarray)
        assert L == len(yarray) == len(stdarray), "L == length(yarray) == length(tarray) == length(stdarray)"
        Nl, t = _flatten(tarray)
        Ny, y = _flatten(yarray)
        Ns, s = _flatten(stdarray)
        assert np.array_equal(Nl, Ny) and np.array_equal(Nl, Ns), "band lengths differ between t, y, sigma"
        self.L, self.Nl, self.N = L, Nl.copy(), int(Nl.sum())
        self.marginalise_b = bool(marginalise_b)
        self.device = int(device)
        lib = _capi.load()
        h = ctypes.c_void_p()
        if devices is None:
            _capi.check(lib.gpcc_create(ctypes.byref(h), L, _ip(Nl), _dp(t), _dp(y), _dp(s), self.kernel.id,
                                        int(self.marginalise_b), _capi.PRECISION_IDS[precision], self.device))
        else:
            devs = np.ascontiguousarray(devices, dtype=np.int32)
            self.device = int(devs[0]) if len(devs) else 0
            _capi.check(lib.gpcc_create_multi(ctypes.byref(h), L, _ip(Nl), _dp(t), _dp(y), _dp(s), self.kernel.id,
                                              int(self.marginalise_b), _capi.PRECISION_IDS[precision], _ip(devs), len(devs)))
        self._h = h
        for key, val in (("streams", streams), ("slots_per_stream", slots_per_stream)):
            if val is not None:
                self.set_option(key, int(val))

    # -- lifetime -------------------------------------------------------------------------------
    def close(self):
        if self._h is not None:
            _capi.load().gpcc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _chk(self, rc):
        if rc != 0:
            raise GpccError(rc, _capi.last_error(self._h))

    def set_option(self, key, value):
        self._chk(_capi.load().gpcc_set_option(self._h, key.encode(), int(value)))

    def get_option(self, key):
        return _capi.load().gpcc_get_option(self._h, key.encode())

    def constants(self):
        """(mu_b, Sigma_b diagonal, Y - bbar) as precomputed at marginaliseb.jl:85-98."""
        mu, sb, r = np.empty(self.L), np.empty(self.L), np.empty(self.N)
        self._chk(_capi.load().gpcc_get_constants(self._h, _dp(mu), _dp(sb), _dp(r)))
        return mu, sb, r

    def conditioning(self, M):
        """fp32 handles: (M, 2) array [sum_i K_ii/d_i, max_i K_ii/d_i] of the last batch (gpcc_get_conditioning)."""
        out = np.empty((int(M), 2), dtype=np.float64)
        self._chk(_capi.load().gpcc_get_conditioning(self._h, int(M), _dp(out)))
        return out

    def gathered(self, which=0):
        """Multi-device handles: what the ONE all-gather of the last call left on device `which`.
        After loglik_batch: (loglik[n_devices, blk], info[n_devices, blk]) (padding: NaN / 0).
        After grid_loglik (the fit, sharded by delay): rows[n_devices, blk, L + 4] = [loglik, info, iterations, rho, alpha(L)] per
        fitted delay; device i's row j is delay j * n_devices + i of the grid (round-robin deal)."""
        blk = ctypes.c_long(0)
        self._chk(_capi.load().gpcc_multi_gathered(self._h, int(which), ctypes.byref(blk), None, 0))
        n = self.get_option("n_devices")
        w = self.get_option("gather_width")
        buf = np.empty(w * blk.value * n, dtype=np.float64)
        self._chk(_capi.load().gpcc_multi_gathered(self._h, int(which), ctypes.byref(blk), _dp(buf), buf.size))
        if w != 2:
            return buf.reshape(n, blk.value, w)
        buf = buf.reshape(n, 2, blk.value)
        return buf[:, 0, :], buf[:, 1, :].astype(np.int32)

    def multi_stats(self):
        """Multi-device handles: timing of the last loglik_batch -> (compute_ms per device, gather_ms, total_ms)."""
        n = self.get_option("n_devices")
        comp = np.empty(n, dtype=np.float64)
        g, tot = ctypes.c_double(0.0), ctypes.c_double(0.0)
        self._chk(_capi.load().gpcc_multi_stats(self._h, _dp(comp), ctypes.byref(g), ctypes.byref(tot)))
        return comp, g.value, tot.value

    def chain_trace(self, evaluation=0):
        """Stamps of the last persistent few-evaluation launch (option "chain_trace" = 1): (nt, 80) microseconds (gpcc_chain_trace)."""
        nt = self.get_option("Np") // 128
        out = np.empty(80 * nt, dtype=np.float64)
        self._chk(_capi.load().gpcc_chain_trace(self._h, int(evaluation), _dp(out), out.size))
        return out.reshape(nt, 80)

    def chain_jobs_trace(self, capacity=32768):
        """The workers' jobs of the last persistent launch: (rows, 6) [kind, step, index, fetched, ready, done] (gpcc_chain_jobs_trace)."""
        out = np.empty(6 * capacity, dtype=np.float64)
        n = ctypes.c_long(0)
        self._chk(_capi.load().gpcc_chain_jobs_trace(self._h, _dp(out), capacity, ctypes.byref(n)))
        return out[:6 * n.value].reshape(n.value, 6)

    # -- the hot path ---------------------------------------------------------------------------
    def _params(self, delays, alpha, rho):
        rho = _d(np.atleast_1d(rho))
        M = len(rho)
        delays = _d(np.atleast_2d(delays))
        alpha = _d(np.atleast_2d(alpha))
        if delays.shape != (M, self.L) or alpha.shape != (M, self.L):
            raise ValueError("delays and alpha must be (M, L) = (%d, %d)" % (M, self.L))
        return M, delays, alpha, rho

    def loglik_batch(self, delays, alpha, rho):
        """objective for M (tau, alpha, rho) triples -> (loglik[M], info[M]).  info follows LAPACK
        potrf (>0: not positive definite, loglik NaN); -1: alpha <= 0; -2: rho <= 0."""
        M, delays, alpha, rho = self._params(delays, alpha, rho)
        ll = np.empty(M, dtype=np.float64)
        info = np.zeros(M, dtype=np.int32)
        self._chk(_capi.load().gpcc_loglik_batch(self._h, M, _dp(delays), _dp(alpha), _dp(rho), _dp(ll), _ip(info)))
        return ll, info

    def __call__(self, alpha, rho, delays):
        """objective(alpha, rho) for one delay vector, raising what the reference raises."""
        ll, info = self.loglik_batch([delays], [alpha], [rho])
        if info[0] == -1:
            raise AssertionError("all(scale .> 0)")
        if info[0] == -2:
            raise ValueError("ρ=%.8f is <= 0" % rho)
        if info[0] > 0:
            raise PosDefException(int(info[0]))
        return float(ll[0])

    def loglik_batch_device(self, delays, alpha, rho, out=None, info=None):
        """Same on torch CUDA tensors (float64, contiguous), asynchronous on torch's current stream."""
        import torch
        M = rho.numel()
        for tns in (delays, alpha, rho):
            assert tns.is_cuda and tns.dtype == torch.float64 and tns.is_contiguous()
        assert delays.numel() == M * self.L and alpha.numel() == M * self.L
        if out is None:
            out = torch.empty(M, dtype=torch.float64, device=rho.device)
        if info is None:
            info = torch.empty(M, dtype=torch.int32, device=rho.device)
        stream = torch.cuda.current_stream(rho.device).cuda_stream
        self._chk(_capi.load().gpcc_loglik_batch_device(self._h, M, delays.data_ptr(), alpha.data_ptr(),
                                                        rho.data_ptr(), out.data_ptr(), info.data_ptr(), stream))
        return out, info

    # -- dense views (prediction, tests) ------------------------------------------------------------
    def model_matrix(self, delays, alpha, rho):
        """K = delayedCovariance + Sobs + B (marginaliseb.jl:135), (N, N) ndarray."""
        delays, alpha = _d(delays), _d(alpha)
        K = np.empty((self.N, self.N), dtype=np.float64)
        try:
            self._chk(_capi.load().gpcc_model_matrix(self._h, _dp(delays), _dp(alpha), float(rho), _dp(K)))
        except GpccError as e:
            _raise_reference_error(e)
        return K.T

    def factor(self, delays, alpha, rho):
        """Lower Cholesky factor of K -> (L, info)."""
        delays, alpha = _d(delays), _d(alpha)
        Lf = np.empty((self.N, self.N), dtype=np.float64)
        info = ctypes.c_int(0)
        try:
            self._chk(_capi.load().gpcc_factor_dense(self._h, _dp(delays), _dp(alpha), float(rho), _dp(Lf),
                                                     ctypes.byref(info)))
        except GpccError as e:
            _raise_reference_error(e)
        return Lf.T, info.value

    # -- prediction / posterior of the offsets (reference: marginaliseb.jl:248-252, :259-343) ----------------
    def predict(self, delays, alpha, rho, ttest):
        """predictTest(ttest::Vector{Vector}) at (tau, alpha, rho): joint (mu_pred, Sigma_pred) for the test
        times ttest[l] of every band, Sigma_pred including JITTER*I (marginaliseb.jl:259-289)."""
        assert len(ttest) == self.L
        delays, alpha = _d(delays), _d(alpha)
        Nt, tt = _flatten(ttest)
        n = int(Nt.sum())
        mu = np.empty(n, dtype=np.float64)
        Sig = np.empty((n, n), dtype=np.float64)
        ll, info = ctypes.c_double(0.0), ctypes.c_int(0)
        try:
            self._chk(_capi.load().gpcc_predict(self._h, _dp(delays), _dp(alpha), float(rho), _ip(Nt), _dp(tt), _dp(mu),
                                                _dp(Sig), ctypes.byref(ll), ctypes.byref(info)))
        except GpccError as e:
            _raise_reference_error(e)
        if info.value > 0:
            raise PosDefException(info.value)
        return mu, Sig.T

    def posterior_offsets(self, delays, alpha, rho):
        """(mu_postb, Sigma_postb) of marginaliseb.jl:248-252 (the reference wraps them in MvNormal)."""
        delays, alpha = _d(delays), _d(alpha)
        mu = np.empty(self.L, dtype=np.float64)
        Sig = np.empty((self.L, self.L), dtype=np.float64)
        info = ctypes.c_int(0)
        try:
            self._chk(_capi.load().gpcc_posterior_offsets(self._h, _dp(delays), _dp(alpha), float(rho), _dp(mu), _dp(Sig),
                                                          ctypes.byref(info)))
        except GpccError as e:
            _raise_reference_error(e)
        if info.value > 0:
            raise PosDefException(info.value)
        return mu, Sig.T

    # -- the per-delay fit over a grid (gpcc_grid_loglik) ---------------------------------------------
    def grid_loglik(self, candidatedelays, iterations, numberofrestarts=1, initialrandom=5, rhomin=0.1, rhomax=20.0,
                    seed=1, init_params=None):
        """Optimised log-likelihood per row of candidatedelays (G, L): README.md:172-174 as one native call.
        Returns (loglikel[G], alpha[G, L], rho[G], info[G], iterations[G], (f_calls, rounds))."""
        cand = np.ascontiguousarray(np.atleast_2d(candidatedelays), dtype=np.float64)
        G = cand.shape[0]
        if cand.shape[1] != self.L:
            raise ValueError("candidatedelays must be (G, %d)" % self.L)
        if init_params is not None:
            init_params = _d(init_params)
            if init_params.size != numberofrestarts * initialrandom * (self.L + 1):
                raise ValueError("init_params must be (numberofrestarts, initialrandom, L + 1)")
        ll = np.empty(G, dtype=np.float64)
        alpha = np.empty((G, self.L), dtype=np.float64)
        rho = np.empty(G, dtype=np.float64)
        info = np.empty(G, dtype=np.int32)
        its = np.empty(G, dtype=np.int32)
        stats = (ctypes.c_longlong * 2)()
        self._chk(_capi.load().gpcc_grid_loglik(self._h, G, _dp(cand), int(iterations), int(numberofrestarts),
                                                int(initialrandom), float(rhomin), float(rhomax), int(seed),
                                                _dp(init_params) if init_params is not None else None, _dp(ll),
                                                _dp(alpha), _dp(rho), _ip(info), _ip(its), stats))
        return ll, alpha, rho, info, its, (int(stats[0]), int(stats[1]))

    def initial_params(self, numberofrestarts=1, initialrandom=5, rhomin=0.1, rhomax=20.0, seed=1):
        out = np.empty((numberofrestarts, initialrandom, self.L + 1), dtype=np.float64)
        self._chk(_capi.load().gpcc_initial_params(self._h, int(numberofrestarts), int(initialrandom), float(rhomin),
                                                   float(rhomax), int(seed), _dp(out)))
        return out

    # -- profiling (bench.py) ------------------------------------------------------------------------
    def profile(self, on):
        self._chk(_capi.load().gpcc_profile_enable(self._h, int(bool(on))))

    def profile_reset(self):
        self._chk(_capi.load().gpcc_profile_reset(self._h))

    def profile_get(self):
        """{kernel: (launches, total_ms)} measured with HIP events on the launching stream."""
        out = {}
        for i, name in enumerate(_capi.PROF_NAMES):
            n, ms = ctypes.c_long(0), ctypes.c_double(0.0)
            self._chk(_capi.load().gpcc_profile_get(self._h, i, ctypes.byref(n), ctypes.byref(ms)))
            out[name] = (n.value, ms.value)
        return out


def unpack_params(X, L, rhomin, rhomax):
    """`unpack` of marginaliseb.jl:112-126 through the library (host arithmetic, no device needed)."""
    X = np.ascontiguousarray(np.atleast_2d(X), dtype=np.float64)
    M = X.shape[0]
    alpha = np.empty((M, L), dtype=np.float64)
    rho = np.empty(M, dtype=np.float64)
    _capi.check(_capi.load().gpcc_unpack_params(M, int(L), _dp(X), float(rhomin), float(rhomax), _dp(alpha), _dp(rho)))
    return alpha, rho


def selftest(device=0, rate=True):
    """f64 MFMA fragment-map check (+ measured fp64 MFMA TFLOP/s)."""
    tf = ctypes.c_double(0.0)
    _capi.check(_capi.load().gpcc_selftest(int(device), ctypes.byref(tf) if rate else None))
    return tf.value


def build_info():
    """What the loaded library was built from: "src=<hash of its sources> defines=[...]" (gpcc_build_info)."""
    return _capi.load().gpcc_build_info().decode()
