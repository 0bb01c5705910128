"""Builds csrc/libgpcc_hip.so (hipcc, gfx950 only).  Cross-compiles without a GPU.

What a library was built FROM travels with it: the SHA-256 of the sources and of the extra compiler defines is compiled into the
library (`gpcc_build_info()`) and written beside it (`<library>.buildinfo`); the library is stale when that record does not match the
tree.  Extra defines (GPCC_BUILD_DEFINES: the A/B switches of tools/) are refused for the default output path: the product library is
always the plain build."""
import hashlib
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
DEFAULT_LIB_PATH = os.path.join(CSRC, "libgpcc_hip.so")
LIB_PATH = os.environ.get("GPCC_HIP_LIB") or DEFAULT_LIB_PATH
_SOURCES = ["gpcc_hip.hip", "gpcc_small_inst.hip", "gpcc_chain_inst.hip", "gpcc_buildinfo.hip", "gpcc_kernels.hip.h", "gpcc_chain.hip.h",
            "gpcc_chain_args.h", "gpcc_chain_queue.h", "gpcc_small.hip.h", "gpcc_fit.h", "gpcc_transforms.h"]
# what each object is compiled from (an object is reused from csrc/_obj while the hash of these files and of its flags stands)
_DEPS = {
    "gpcc_hip.hip": ["gpcc_hip.hip", "gpcc_kernels.hip.h", "gpcc_small.hip.h", "gpcc_chain_args.h", "gpcc_chain_queue.h", "gpcc_fit.h", "gpcc_transforms.h"],
    "gpcc_chain_inst.hip": ["gpcc_chain_inst.hip", "gpcc_chain.hip.h", "gpcc_chain_args.h", "gpcc_chain_queue.h", "gpcc_kernels.hip.h"],
    "gpcc_small_inst.hip": ["gpcc_small_inst.hip", "gpcc_small.hip.h", "gpcc_kernels.hip.h", "gpcc_transforms.h"],
    "gpcc_buildinfo.hip": ["gpcc_buildinfo.hip"],
}
_HEADER = os.path.join(os.path.dirname(_HERE), "include", "gpcc_hip.h")


def _defines():
    d = os.environ.get("GPCC_BUILD_DEFINES", "").split()
    if d and os.path.realpath(LIB_PATH) == os.path.realpath(DEFAULT_LIB_PATH):
        raise RuntimeError("GPCC_BUILD_DEFINES=%r with the default library path %s: A/B builds go to their own file (set GPCC_HIP_LIB); "
                           "the product library is always the plain build" % (" ".join(d), DEFAULT_LIB_PATH))
    return d


def source_hash():
    """SHA-256 (first 16 hex digits) of every source of the library and of the extra defines."""
    h = hashlib.sha256()
    for path in [os.path.join(CSRC, s) for s in _SOURCES] + [_HEADER]:
        if os.path.exists(path):
            h.update(os.path.basename(path).encode() + b"\0")
            with open(path, "rb") as f:
                h.update(f.read())
    h.update(" ".join(os.environ.get("GPCC_BUILD_DEFINES", "").split()).encode())
    return h.hexdigest()[:16]


def build_info_string():
    d = " ".join(os.environ.get("GPCC_BUILD_DEFINES", "").split())
    return "src=%s defines=[%s]" % (source_hash(), d)


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    try:
        with open(LIB_PATH + ".buildinfo") as f:
            return f.read().strip() != build_info_string()
    except OSError:
        return True


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -shared -fPIC ... -> csrc/libgpcc_hip.so"""
    if not force and not _stale():
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    tmp = "%s.tmp%d" % (LIB_PATH, os.getpid())      # link elsewhere, then rename: no reader ever sees a partial file
    # librccl from the ROCm tree this hipcc belongs to (ROCM_PATH, else <hipcc>/../lib), not a hard-coded /opt/rocm
    rocm = os.environ.get("ROCM_PATH") or os.path.dirname(os.path.dirname(os.path.realpath(shutil.which(hipcc) or hipcc)))
    libdir = os.path.join(rocm, "lib")
    if not os.path.isdir(libdir):
        libdir = "/opt/rocm/lib"
    # Eleven objects compiled side by side: the host + tile kernels, the persistent few-evaluation kernel, the small-N families once
    # per (family, kernel id), the build record (as ONE translation unit the library took 7.5 minutes to build; the objects are
    # independent, no device linking).  Objects are kept in csrc/_obj and reused while their own sources and flags are unchanged.
    from concurrent.futures import ThreadPoolExecutor
    flags = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-pass-failed", "-Wno-inline-asm", "-c"]
    flags += _defines()     # A/B builds only (tools/ab_*.sh: e.g. -DGPCC_AB_POLY_EXP into GPCC_HIP_LIB; refused for the default path)
    info = build_info_string()
    objdir = os.path.join(CSRC, "_obj") if os.path.realpath(LIB_PATH) == os.path.realpath(DEFAULT_LIB_PATH) else LIB_PATH + ".obj"
    os.makedirs(objdir, exist_ok=True)
    jobs = [(os.path.join(objdir, "gpcc_hip.o"), [os.path.join(CSRC, "gpcc_hip.hip")], "gpcc_hip.hip"),
            (os.path.join(objdir, "gpcc_chain.o"), [os.path.join(CSRC, "gpcc_chain_inst.hip")], "gpcc_chain_inst.hip"),
            (os.path.join(objdir, "gpcc_buildinfo.o"), ['-DGPCC_BUILD_INFO_STR="%s"' % info, os.path.join(CSRC, "gpcc_buildinfo.hip")], "gpcc_buildinfo.hip")]
    for wide in (1, 0):
        for kid in range(4):
            jobs.append((os.path.join(objdir, "small_%d_%d.o" % (wide, kid)),
                         ["-DGPCC_INST_WIDE=%d" % wide, "-DGPCC_INST_KID=%d" % kid, os.path.join(CSRC, "gpcc_small_inst.hip")], "gpcc_small_inst.hip"))

    def object_key(job):
        h = hashlib.sha256()
        h.update(" ".join(flags + job[1]).encode())
        for dep in _DEPS[job[2]]:
            h.update(dep.encode() + b"\0")
            with open(os.path.join(CSRC, dep), "rb") as f:
                h.update(f.read())
        return h.hexdigest()

    class _Reused:
        returncode, stdout, stderr = 0, "", ""

    def compile_one(job):
        obj, args, _ = job
        cmd = flags + args + ["-o", obj]
        key = object_key(job)
        try:
            if not force and os.path.exists(obj) and open(obj + ".key").read() == key:
                return cmd, _Reused()
        except OSError:
            pass
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode == 0:
            with open(obj + ".key", "w") as f:
                f.write(key)
        return cmd, res

    workers = max(1, min(len(jobs), int(os.environ.get("GPCC_BUILD_JOBS", "0")) or (os.cpu_count() or 4)))
    with ThreadPoolExecutor(workers) as pool:
        results = list(pool.map(compile_one, jobs))
    failed = [(c, r) for c, r in results if r.returncode]
    if not failed:
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + [j[0] for j in jobs] + ["-L" + libdir, "-lrccl", "-pthread"]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode:
            failed = [(cmd, res)]
        elif verbose:
            print(" ".join(cmd))
            print(res.stdout, res.stderr)
    if failed:      # every failing command with ITS diagnostics (not the link line with one compile's stderr)
        if os.path.exists(tmp):
            os.remove(tmp)
        report = "\n".join("$ %s\n%s%s" % (" ".join(c), r.stdout, r.stderr) for c, r in failed)
        print(report)
        raise RuntimeError("hipcc failed (%d command%s):\n%s" % (len(failed), "" if len(failed) == 1 else "s", report))
    os.replace(tmp, LIB_PATH)
    with open(LIB_PATH + ".buildinfo", "w") as f:
        f.write(info + "\n")
    return LIB_PATH


def ensure_present(local_rank=0, timeout_s=900.0):
    """For launchers that start one process per GPU: local rank 0 (re)builds the library when it is missing OR older
    than its sources (the same content-hash check as build()), the other ranks wait until an up-to-date file is there --
    so a benchmark never times a stale library after an edit of csrc/."""
    import time
    if local_rank == 0:
        return build(force=False)
    t0 = time.time()
    while _stale():
        if time.time() - t0 > timeout_s:
            raise RuntimeError("%s did not become current within %.0f s" % (LIB_PATH, timeout_s))
        time.sleep(1.0)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
