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
PROF_NAMES = ("assemble", "panel_update", "diag_factor", "panel_trsm")

# every symbol include/gpcc_hip.h declares: name -> (restype, argtypes)
BATCH_OBJECTIVE = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_long, ctypes.POINTER(ctypes.c_long),
                                   ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double))

SIGNATURES = {
    "gpcc_version": (ctypes.c_int, []),
    "gpcc_last_error": (ctypes.c_char_p, [ctypes.c_void_p]),
    "gpcc_create": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, c_int_p, c_double_p, c_double_p,
                                   c_double_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "gpcc_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "gpcc_set_option": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_long]),
    "gpcc_get_option": (ctypes.c_long, [ctypes.c_void_p, ctypes.c_char_p]),
    "gpcc_get_constants": (ctypes.c_int, [ctypes.c_void_p, c_double_p, c_double_p, c_double_p]),
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
    "gpcc_selftest": (ctypes.c_int, [ctypes.c_int, c_double_p]),
}


class GpccError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("libgpcc_hip error %d: %s" % (code, message))
        self.code = code
        self.message = message


_lib = None


def load():
    """Loads csrc/libgpcc_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950).  gpcc_amd has no CPU fallback." % LIB_PATH)
        if "torch" in sys.modules:
            # torch wheels bundle their own HIP runtime; when both live in one process torch must
            # initialise the GPU first (the other order has been seen to end in "No HIP GPUs are available")
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        lib = ctypes.CDLL(LIB_PATH)
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
