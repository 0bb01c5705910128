"""ctypes binding of the C ABI declared in include/gpcc_hip.h (csrc/libgpcc_hip.so).

This is the same boundary the Julia shim of INTEGRATION.md binds with `ccall`.  There is no CPU
fallback: if the shared library is missing, or no HIP device is present, calls raise."""
import ctypes
import os
import sys

from .build import LIB_PATH

c_double_p = ctypes.POINTER(ctypes.c_double)
c_int_p = ctypes.POINTER(ctypes.c_int)
c_long_p = ctypes.POINTER(ctypes.c_long)

KERNEL_IDS = {"OU": 0, "rbf": 1, "matern32": 2, "matern52": 3}
PRECISION_IDS = {"fp64": 0, "fp32": 1}
PROF_NAMES = ("assemble", "panel_update", "diag_factor", "panel_trsm", "refine", "small_step", "small_eval")

# every symbol include/gpcc_hip.h declares: name -> (restype, argtypes)
BATCH_OBJECTIVE = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_long, ctypes.POINTER(ctypes.c_long),
                                   ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double))

SIGNATURES = {
    "gpcc_version": (ctypes.c_int, []),
    "gpcc_build_info": (ctypes.c_char_p, []),
    "gpcc_last_error": (ctypes.c_char_p, [ctypes.c_void_p]),
    "gpcc_create": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, c_int_p, c_double_p, c_double_p,
                                   c_double_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "gpcc_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "gpcc_create_multi": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, c_int_p, c_double_p, c_double_p,
                                         c_double_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_int_p, ctypes.c_int]),
    "gpcc_multi_gathered": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_long_p, c_double_p, ctypes.c_long]),
    "gpcc_multi_stats": (ctypes.c_int, [ctypes.c_void_p, c_double_p, c_double_p, c_double_p]),
    "gpcc_set_option": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_long]),
    "gpcc_get_option": (ctypes.c_long, [ctypes.c_void_p, ctypes.c_char_p]),
    "gpcc_get_constants": (ctypes.c_int, [ctypes.c_void_p, c_double_p, c_double_p, c_double_p]),
    "gpcc_get_conditioning": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_double_p]),
    "gpcc_loglik_batch": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_double_p, c_double_p, c_double_p,
                                         c_double_p, c_int_p]),
    "gpcc_loglik_batch_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                                ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "gpcc_model_matrix": (ctypes.c_int, [ctypes.c_void_p, c_double_p, c_double_p, ctypes.c_double, c_double_p]),
    "gpcc_factor_dense": (ctypes.c_int, [ctypes.c_void_p, c_double_p, c_double_p, ctypes.c_double, c_double_p,
                                         c_int_p]),
    "gpcc_predict": (ctypes.c_int, [ctypes.c_void_p, c_double_p, c_double_p, ctypes.c_double, c_int_p, c_double_p,
                                    c_double_p, c_double_p, c_double_p, c_int_p]),
    "gpcc_posterior_offsets": (ctypes.c_int, [ctypes.c_void_p, c_double_p, c_double_p, ctypes.c_double, c_double_p,
                                              c_double_p, c_int_p]),
    "gpcc_mvnormal_logpdf": (ctypes.c_int, [ctypes.c_int, c_double_p, c_double_p, c_double_p, c_double_p, c_int_p,
                                            ctypes.c_int]),
    "gpcc_covariance": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, c_double_p, c_double_p, ctypes.c_double, c_int_p,
                                       c_double_p, c_int_p, c_double_p, c_double_p, ctypes.c_int]),
    "gpcc_grid_loglik": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_double_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_double, ctypes.c_double, ctypes.c_ulonglong, c_double_p, c_double_p,
                                        c_double_p, c_double_p, c_int_p, c_int_p, ctypes.POINTER(ctypes.c_longlong)]),
    "gpcc_initial_params": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double,
                                           ctypes.c_ulonglong, c_double_p]),
    "gpcc_neldermead_batch": (ctypes.c_int, [ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_double, c_double_p,
                                             BATCH_OBJECTIVE, ctypes.c_void_p, c_double_p, c_double_p, c_int_p,
                                             ctypes.POINTER(ctypes.c_longlong)]),
    "gpcc_unpack_params": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, c_double_p, ctypes.c_double, ctypes.c_double,
                                          c_double_p, c_double_p]),
    "gpcc_probabilities": (ctypes.c_int, [ctypes.c_int, c_double_p, c_double_p, c_double_p, ctypes.c_int]),
    "gpcc_probabilities_device": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                 ctypes.c_void_p]),
    "gpcc_profile_enable": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "gpcc_profile_reset": (ctypes.c_int, [ctypes.c_void_p]),
    "gpcc_profile_get": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_long_p, c_double_p]),
    "gpcc_chain_trace": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_double_p, ctypes.c_int]),
    "gpcc_chain_jobs_trace": (ctypes.c_int, [ctypes.c_void_p, c_double_p, ctypes.c_long, c_long_p]),
    "gpcc_selftest": (ctypes.c_int, [ctypes.c_int, c_double_p]),
}


class GpccError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("libgpcc_hip error %d: %s" % (code, message))
        self.code = code
        self.message = message


_lib = None
_preloaded = []


def _share_torch_rocm_runtime():
    """One HIP/HSA runtime per process.  libgpcc_hip.so asks the loader for `libamdhip64.so.7` / `librccl.so.1`
    (DT_NEEDED = the SONAMEs, resolved to /opt/rocm/lib).  A torch ROCm wheel bundles its own copies under
    torch/lib/ and asks for them by FILE name (`libamdhip64.so`, `librccl.so`, RPATH $ORIGIN).  glibc reuses an
    already-loaded object when the requested name equals its SONAME or the path it was loaded from, so:
      torch first, then libgpcc_hip   -> our `libamdhip64.so.7` matches the SONAME of torch's copy: ONE runtime;
      libgpcc_hip first, then torch   -> `libamdhip64.so` matches nothing loaded: torch maps its own copy beside
                                         /opt/rocm's, two HSA runtimes open the GPU and the second finds none
                                         ("No HIP GPUs are available", round-1 gpurun_out/test12.log).
    Whenever a torch installation with bundled ROCm libraries exists, map THOSE first (by path): our
    SONAME requests bind to them now, and torch's later file-name requests resolve to the same paths.  Either import
    order then shares one runtime.  Without torch (a Julia or C host) the system libraries are used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    libdir = os.path.join(os.path.dirname(spec.origin), "lib")
    for name in ("libamdhip64.so", "librccl.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            _preloaded.append(ctypes.CDLL(path))   # RTLD_LOCAL: a global-scope librccl ahead of torch ends in a double free at exit


def loaded_rocm_libraries():
    """Paths of every libamdhip64 / libhsa-runtime64 / librccl mapped into this process (from /proc/self/maps)."""
    seen = []
    with open("/proc/self/maps") as f:
        for line in f:
            path = line.rsplit(" ", 1)[-1].strip()
            base = os.path.basename(path)
            if base.startswith(("libamdhip64", "libhsa-runtime64", "librccl")) and path not in seen:
                seen.append(path)
    return seen


def _assert_one_rocm_runtime():
    """Two HIP (or HSA) runtimes in one process is the failure _share_torch_rocm_runtime exists to prevent -- e.g. a torch
    wheel whose bundled libamdhip64 carries another SONAME than the `libamdhip64.so.7` libgpcc_hip.so was linked against.
    The second runtime finds no GPU ("No HIP GPUs are available") far from its cause, so say it here."""
    libs = loaded_rocm_libraries()
    for stem in ("libamdhip64", "libhsa-runtime64"):
        paths = sorted({os.path.realpath(p) for p in libs if os.path.basename(p).startswith(stem)})
        if len(paths) > 1:
            msg = ("two copies of %s are mapped into this process (%s): libgpcc_hip.so and another component (torch?) may "
                   "each open the GPU through their own runtime.  Build libgpcc_hip.so against the ROCm libraries the other "
                   "component ships, or import it in a process of its own." % (stem, ", ".join(paths)))
            if stem == "libamdhip64":
                raise ImportError(msg)
            # a second HSA runtime that is only mapped (rocprofv3 preloads the system copy for its tool library while the
            # HIP runtime in use is torch's) does no harm: say so once, do not refuse
            import warnings
            warnings.warn(msg, RuntimeWarning, stacklevel=3)


def load():
    """Loads csrc/libgpcc_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950).  gpcc_amd has no CPU fallback." % LIB_PATH)
        _share_torch_rocm_runtime()
        lib = ctypes.CDLL(LIB_PATH)
        _assert_one_rocm_runtime()
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the library does not export the symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def last_error(handle=None):
    msg = load().gpcc_last_error(handle)
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc, handle=None):
    if rc != 0:
        raise GpccError(rc, last_error(handle))
    return rc
