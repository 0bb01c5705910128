"""Builds csrc/libgpcc_hip.so (hipcc, gfx950 only).  Cross-compiles without a GPU."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("GPCC_HIP_LIB") or os.path.join(CSRC, "libgpcc_hip.so")
_SOURCES = ["gpcc_hip.hip", "gpcc_kernels.hip.h"]
_HEADER = os.path.join(os.path.dirname(_HERE), "include", "gpcc_hip.h")


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    so_m = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in _SOURCES] + [_HEADER]
    return any(os.path.exists(d) and os.path.getmtime(d) > so_m for d in deps)


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -shared -fPIC ... -> csrc/libgpcc_hip.so"""
    if not force and not _stale():
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value",
           "-o", LIB_PATH, os.path.join(CSRC, "gpcc_hip.hip")]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode:
        print(" ".join(cmd))
        print(res.stdout, res.stderr)
    if res.returncode:
        raise RuntimeError("hipcc failed:\n" + res.stderr)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
