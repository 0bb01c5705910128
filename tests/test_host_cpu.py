"""CPU-side checks of the product package: the C-ABI library loads and exports every symbol of
include/gpcc_hip.h, fails loudly without a GPU (no fallback), host logic of the sharding."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from gpcc_amd import _capi
    from gpcc_amd import build
    build.build()
    return _capi.load()


def test_library_exports_every_declared_symbol(lib):
    from gpcc_amd import _capi
    header = open(os.path.join(ROOT, "include", "gpcc_hip.h")).read()
    declared = set(re.findall(r"\b(gpcc_[a-z_0-9]+)\s*\(", header))
    assert declared, "no prototypes found"
    assert declared == set(_capi.SIGNATURES), declared ^ set(_capi.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.gpcc_version() >= 100


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import gpcc_amd
    with pytest.raises(gpcc_amd.GpccError) as ei:
        gpcc_amd.getprobabilities([0.0, 1.0])
    assert "no HIP device" in str(ei.value)
    with pytest.raises(gpcc_amd.GpccError):
        gpcc_amd.Objective([[0.0, 1.0]], [[1.0, 2.0]], [[0.1, 0.1]], gpcc_amd.OU)
    with pytest.raises(gpcc_amd.GpccError):
        gpcc_amd.delayedCovariance(gpcc_amd.OU, [1.0], [0.0], 1.0, [[0.0, 1.0]])


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "gpcc.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.lower(), os.path.join(dirpath, f)


def test_shard_bounds_partition():
    from gpcc_amd import shard_bounds
    for G in (0, 1, 7, 256, 1024, 65536):
        for world in (1, 2, 3, 8):
            covered = []
            for r in range(world):
                lo, hi = shard_bounds(G, world, r)
                covered.extend(range(lo, hi))
                assert hi - lo in (G // world, G // world + 1)
            assert covered == list(range(G))


def test_synthetic_recipe():
    from gpcc_amd import synthetic
    t, y, s, td = synthetic.simulate_lightcurves([300, 200, 100], seed=4, gap_band=1)
    assert [len(a) for a in t] == [300, 200, 100] and td == [0.0, 2.0, 4.0]
    assert not np.all(np.diff(t[0]) > 0)            # unsorted, like the reference
    assert abs(np.mean(y[0]) - 6) < 1.5 and abs(np.mean(y[2]) - 25) < 4
    gap = t[1]
    assert ((gap <= 0.4 * 200 / 3) | (gap >= 0.6 * 200 / 3)).all()
    t2, y2, _, _ = synthetic.simulate_lightcurves([300, 200, 100], seed=4, gap_band=1)
    assert all(np.array_equal(a, b) for a, b in zip(t, t2)) and all(np.array_equal(a, b) for a, b in zip(y, y2))


def test_persistent_launch_job_order_covers_every_job_and_never_waits_forward(tmp_path):
    """The queue order of the persistent few-evaluation launch (csrc/gpcc_chain_queue.h, the same header the kernel compiles): for
    nt = 2 .. 64 (N <= 8192: all the default policy ever gives to the launch), with and without helpers, whole-tile and quarter-tile
    updates, bulk jobs of at most 1, 2, 4, 8 columns -- every solve and every update exactly once, and every input of a job produced
    earlier in the order or by the chain (whose own needs are earlier too): the oldest unfinished job can always run, which is the
    launch's whole argument against deadlock.  Once under AddressSanitizer + UBSan (nt <= 32), once plain for the whole range."""
    import subprocess
    src = os.path.join(ROOT, "tests", "abi", "chain_queue_check.cpp")
    for tag, flags in (("san", ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-DNT_MAX=32"]), ("full", ["-O2"])):
        exe = str(tmp_path / ("chain_queue_check_" + tag))
        cc = subprocess.run(["g++", "-std=c++17"] + flags + [src, "-o", exe], capture_output=True, text=True)
        assert cc.returncode == 0, cc.stderr
        run = subprocess.run([exe], capture_output=True, text=True, timeout=900)
        assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
        assert "0 failures" in run.stdout and ("nt = 2 .. %d" % (32 if tag == "san" else 64)) in run.stdout


def test_native_optimiser_under_sanitizers(tmp_path):
    """gpcc_fit.h (the C++ host logic of gpcc_grid_loglik) compiled host-only with AddressSanitizer + UBSan."""
    import subprocess
    exe = str(tmp_path / "fit_sanitize")
    src = os.path.join(ROOT, "tests", "abi", "fit_host_sanitize.cpp")
    cc = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", src, "-o", exe],
                        capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr[-3000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr[-3000:]
    assert "rosenbrock" in run.stdout and "bowl5: ok" in run.stdout and "initial_params: ok" in run.stdout
    assert "speculative rounds:" in run.stdout      # bitwise the plain trajectories, fewer rounds (checked inside)
    assert "round-robin deal:" in run.stdout        # the grid's deal over n devices and its reassembly: G % n != 0, G < n, G = 0


def test_julia_shim_file_matches_integration_md_and_the_header():
    """julia/gpcchip.jl is the shim of INTEGRATION.md section 1 verbatim (never executed: no Julia here), and every C symbol
    it ccalls is declared by include/gpcc_hip.h."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shim = open(os.path.join(root, "julia", "gpcchip.jl")).read()
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    body = shim[shim.index("# Thin ccall layer"):]
    assert body in doc
    header = open(os.path.join(root, "include", "gpcc_hip.h")).read()
    syms = set(re.findall(r"ccall\(\(:(\w+), LIB\)", shim))
    assert {"gpcc_create", "gpcc_destroy", "gpcc_loglik_batch", "gpcc_grid_loglik", "gpcc_probabilities", "gpcc_last_error"} <= syms
    for sym in syms:
        assert re.search(r"\b%s\(" % sym, header), sym


def test_every_option_key_is_documented_in_the_header():
    """Each key gpcc_set_option / gpcc_get_option accept is named in include/gpcc_hip.h."""
    src = open(os.path.join(ROOT, "gpcc.jl_amd", "csrc", "gpcc_hip.hip")).read()
    header = open(os.path.join(ROOT, "include", "gpcc_hip.h")).read()
    keys = set(re.findall(r'!strcmp\(key, "([A-Za-z_0-9]+)"\)', src))
    assert len(keys) >= 30
    # (a row of the option table -- " *   key   default  meaning" -- or the key in quotes in an entry point's comment)
    missing = sorted(k for k in keys if '"%s"' % k not in header and not re.search(r"^ \*   %s\s" % k, header, re.M))
    assert not missing, missing
