"""The persistent few-evaluation launch (csrc/gpcc_chain.hip.h: groups of at most `chain_max` evaluations on an fp64 handle at
N > 128 -- call site 1 of the boundary, one objective(alpha, rho), marginaliseb.jl:133-141) against the golden fixtures, the CPU
oracle, and the launch-per-step path it replaces (chain_max = 0), through the C ABI.

Tolerances: 1e-8 relative against golden / oracle (the north-star bar is 1e-6), 1e-11 against the launch-per-step path (the same
tile algorithm in another summation order)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LL_RTOL = 1e-8


@pytest.fixture(scope="module")
def gp():
    import torch
    torch.cuda.init()
    import gpcc_amd
    return gpcc_amd


def _rel(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b)) / np.abs(np.asarray(b)))


def test_chain_golden_cases(gp, golden, monkeypatch):
    """The 44 golden log-likelihoods on the tile kernels (GPCC_SMALL_N=0), one evaluation per call: those with N > 128 take the
    persistent launch (two tiles), the others the one-tile path -- both must give the golden value."""
    monkeypatch.setenv("GPCC_SMALL_N", "0")
    worst, took = 0.0, 0
    for c in golden["cases"]:
        with gp.Objective(c["t"], c["y"], c["sigma"], c["kernel"], marginalise_b=c["marginalise_b"], slots_per_stream=4) as obj:
            ll, info = obj.loglik_batch([c["delays"]], [c["alpha"]], [c["rho"]])
            took += obj.get_option("chain_count")
        assert info[0] == 0
        worst = max(worst, abs(ll[0] - c["loglik"]) / abs(c["loglik"]))
    print("worst relative error vs golden: %.3e; %d of %d cases took the persistent launch" % (worst, took, len(golden["cases"])))
    assert worst <= LL_RTOL
    assert took > 0


@pytest.mark.parametrize("Nl,kname,mb", [([257, 256], "matern32", True), ([512, 512], "OU", True), ([512, 512], "rbf", False),
                                         ([2048, 2048], "matern32", True), ([1365, 1365, 1365], "matern52", True)])
def test_chain_vs_oracle_and_launch_per_step_path(gp, oracle, Nl, kname, mb):
    """N = 513 (ragged last tile), 1024, 4096, 4095 (three bands): M = 1, 2, 5, 12 evaluations per call with different (tau, alpha,
    rho) -- against the oracle, against the same handle with chain_max = 0, bitwise repeatable, chain_count says the path ran."""
    from gpcc_amd import synthetic
    t, y, s, _ = synthetic.simulate_lightcurves(Nl, seed=3, gap_band=1)
    L = len(Nl)
    rng = np.random.default_rng(11)
    M = 12
    delays = np.concatenate([np.zeros((M, 1)), rng.random((M, L - 1)) * 10], 1)
    alpha = 0.5 + rng.random((M, L)) * 2
    rho = 1.0 + rng.random(M) * 5
    nref = 12 if sum(Nl) <= 1100 else 3
    ref, rinfo = oracle.loglik_batch(kname, t, y, s, delays[:nref], alpha[:nref], rho[:nref], mb, nthreads=8)
    assert (rinfo == 0).all()
    with gp.Objective(t, y, s, kname, marginalise_b=mb, slots_per_stream=16) as obj:
        assert obj.get_option("chain_max") == 32 and obj.get_option("chain_work_max") == 4096 and obj.get_option("chain_wide_work_max") == 1024
        obj.set_option("chain_work_max", 1 << 30)   # (the kernel itself is under test: every group size takes it)
        out = {}
        for m in (1, 2, 5, 12):
            before = obj.get_option("chain_count")
            ll, info = obj.loglik_batch(delays[:m], alpha[:m], rho[:m])
            assert obj.get_option("chain_count") == before + m, "the group did not take the persistent launch"
            assert (info == 0).all()
            out[m] = ll
            ll2, _ = obj.loglik_batch(delays[:m], alpha[:m], rho[:m])
            assert np.array_equal(ll, ll2), "not repeatable"
        # an evaluation does not depend on the group it travels in
        for m in (1, 2, 5):
            assert np.array_equal(out[m], out[12][:m])
        assert _rel(out[12][:nref], ref) <= LL_RTOL
        # the closure form: one objective(alpha, rho)
        assert obj(alpha[0], rho[0], delays[0]) == out[12][0]
        obj.set_option("chain_max", 0)
        before = obj.get_option("chain_count")
        ll_old, info_old = obj.loglik_batch(delays, alpha, rho)
        assert obj.get_option("chain_count") == before and (info_old == 0).all()
        print("N = %d %s: chain vs oracle %.2e, vs launch-per-step path %.2e" % (sum(Nl), kname, _rel(out[12][:nref], ref), _rel(out[12], ll_old)))
        assert _rel(out[12], ll_old) <= 1e-11


def test_chain_status_codes(gp):
    """A non-positive pivot in the FIRST, a middle and the last tile, argument errors, and valid evaluations in the same group: info as
    the launch-per-step path reports it (LAPACK-style: the order of the first non-positive pivot), NaN log-likelihood.  The singular
    pair: two observations at the same time, far from all others (the OU kernel underflows to exactly 0 towards them), unit amplitude,
    sigma = 0, no B term -- their 2 x 2 block is [[1, 1], [1, 1]] whatever was eliminated before, the second pivot exactly 0."""
    from gpcc_amd import synthetic
    t, y, s, _ = synthetic.simulate_lightcurves([300, 340], seed=4)
    rho = 3.5
    for dup in (5, 300 + 20, 300 + 339):     # the second copy sits in tile 0, tile 2, the last tile (N = 640: 5 tiles)
        t2 = [t[0].copy(), t[1].copy()]
        s2 = [np.full(300, 0.5), np.full(340, 0.5)]
        b, i = (0, dup) if dup < 300 else (1, dup - 300)
        t2[b][i] = t2[b][i - 1] = 1.0e6
        s2[b][i] = s2[b][i - 1] = 0.0
        with gp.Objective(t2, y, s2, "OU", marginalise_b=False, slots_per_stream=8) as obj:
            d = [[0.0, 0.0]] * 4
            a = [[1.0, 1.0], [1.0, -1.0], [1.0, 1.0], [1.0, 1.0]]
            r = [rho, 1.0, 0.0, rho]
            ll, info = obj.loglik_batch(d, a, r)
            assert obj.get_option("chain_count") == 4
            obj.set_option("chain_max", 0)
            ll0, info0 = obj.loglik_batch(d, a, r)
        print("second copy at %d: info %s (launch-per-step path: %s)" % (dup, info, info0))
        assert info[1] == -1 and info[2] == -2 and np.isnan(ll[1]) and np.isnan(ll[2])
        assert info[0] == info[3] == dup + 1 and np.isnan(ll[0]) and np.isnan(ll[3])
        assert np.array_equal(info, info0)
    alpha, rho = synthetic.default_hyperparameters(y)
    # valid and invalid evaluations side by side
    with gp.Objective(t, y, s, "matern32", slots_per_stream=8) as obj:
        ll, info = obj.loglik_batch([[0.0, 1.0], [0.0, 2.0], [0.0, 3.0]], [list(alpha), [0.0, 1.0], list(alpha)], [rho, rho, rho])
        assert info[0] == 0 and info[1] == -1 and info[2] == 0 and np.isnan(ll[1])
        ok, _ = obj.loglik_batch([[0.0, 1.0], [0.0, 3.0]], [list(alpha)] * 2, [rho, rho])
        assert np.array_equal(ok, ll[[0, 2]])
        with pytest.raises(AssertionError):
            obj([0.0, 1.0], rho, [0.0, 1.0])
        with pytest.raises(ValueError):
            obj(alpha, -1.0, [0.0, 1.0])


def test_chain_is_fp64_only_and_other_paths_unchanged(gp, oracle):
    """The persistent launch works on fp64 tiles.  An fp32 handle hands a call that an fp64 handle's policy would give to it to its
    fp64 twin (option fp32_chain, on): the fp64 handle's bits, nothing for the guard to do; with fp32_chain = 0 it keeps the fp32
    launch-per-step path.  Groups above chain_max and the dense utilities keep their paths (chain_count unchanged) and their results."""
    import torch
    from gpcc_amd import synthetic
    t, y, s, _ = synthetic.simulate_lightcurves([400, 400], seed=6)
    alpha, rho = synthetic.default_hyperparameters(y)
    d = np.stack([np.zeros(3), np.array([0.5, 1.5, 2.5])], 1)
    A3, R3 = np.tile(alpha, (3, 1)), np.full(3, rho)
    ref, _ = oracle.loglik_batch("matern32", t, y, s, d, A3, R3, True, nthreads=4)
    with gp.Objective(t, y, s, "matern32") as obj:
        ll64, _ = obj.loglik_batch(d, A3, R3)
        assert obj.get_option("chain_count") == 3
    with gp.Objective(t, y, s, "matern32", precision="fp32") as obj:
        ll, info = obj.loglik_batch(d, A3, R3)
        assert obj.get_option("chain_count") == 0 and obj.get_option("fp32_chain_count") == 3 and (info == 0).all()
        assert np.array_equal(ll, ll64) and _rel(ll, ref) <= 1e-8
        assert np.array_equal(obj.conditioning(3), np.zeros((3, 2))) and obj.get_option("fp32_guard_count") == 0
        dev = torch.device("cuda", 0)   # the device-pointer entry goes the same way
        out = torch.empty(3, dtype=torch.float64, device=dev)
        oinfo = torch.empty(3, dtype=torch.int32, device=dev)
        obj.loglik_batch_device(torch.as_tensor(d, device=dev), torch.as_tensor(A3, device=dev), torch.as_tensor(R3, device=dev), out=out, info=oinfo)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), ll64) and obj.get_option("fp32_chain_count") == 6
        bad = A3.copy()
        bad[1, 0] = -1.0   # an argument error beside valid evaluations comes back as the fp64 handle reports it
        llb, infob = obj.loglik_batch(d, bad, R3)
        assert infob[1] == -1 and np.isnan(llb[1]) and infob[0] == 0 and infob[2] == 0 and llb[0] == ll64[0]
        M = 21              # beyond the policy (21 x 7^2 > 1024): fp32 tiles as before
        dd = np.stack([np.zeros(M), np.linspace(0, 6, M)], 1)
        before = obj.get_option("fp32_chain_count")
        ll13, info13 = obj.loglik_batch(dd, np.tile(alpha, (M, 1)), np.full(M, rho))
        assert obj.get_option("fp32_chain_count") == before and (info13 == 0).all()
        obj.set_option("fp32_chain", 0)
        ll32, info32 = obj.loglik_batch(d, A3, R3)
        assert obj.get_option("fp32_chain_count") == before and (info32 == 0).all() and _rel(ll32, ref) <= 1e-3
        assert not np.array_equal(ll32, ll64)
    with gp.Objective(t, y, s, "matern32", slots_per_stream=32) as obj:
        M = 21
        dd = np.stack([np.zeros(M), np.linspace(0, 6, M)], 1)
        ll13, info = obj.loglik_batch(dd, np.tile(alpha, (M, 1)), np.full(M, rho))
        assert obj.get_option("chain_count") == 0 and (info == 0).all()
        ll12, _ = obj.loglik_batch(dd[:12], np.tile(alpha, (12, 1)), np.full(12, rho))
        assert obj.get_option("chain_count") == 12
        assert _rel(ll12, ll13[:12]) <= 1e-11
        ll16, info16 = obj.loglik_batch(dd[:16], np.tile(alpha, (16, 1)), np.full(16, rho))   # 13 .. 20 evaluations at N = 800: the launch too
        assert obj.get_option("chain_count") == 28 and (info16 == 0).all() and np.array_equal(ll16[:12], ll12)
        obj.set_option("chain_max", 12)
        ll16b, _ = obj.loglik_batch(dd[:16], np.tile(alpha, (16, 1)), np.full(16, rho))       # ... or, as before, two halves on two streams
        assert obj.get_option("chain_count") == 28 and _rel(ll16b, ll16) <= 1e-11
        obj.set_option("chain_max", 32)
        K = obj.model_matrix(d[0], alpha, rho)
        assert np.array_equal(K, K.T)
        Lf, finfo = obj.factor(d[0], alpha, rho)
        assert finfo == 0 and obj.get_option("chain_count") == 28


def test_chain_default_policy(gp):
    """Which groups take the persistent launch by default: up to 12 evaluations while evaluations x (N/128)^2 <= chain_work_max = 4096 -- 4
    evaluations at N = 4096, 12 at N = 2048 (profiles/r05/latency_small_batches.log: above, the launch-per-step path is faster) -- and
    13 .. chain_max = 32 while <= chain_wide_work_max = 1024 (32 at N <= 512, 16 at N = 1024; chain_13_to_32_evaluations_ab.log)."""
    from gpcc_amd import synthetic
    for Nb, takes, not_any_more in ((2048, 4, 5), (1024, 12, 13), (768, 12, 13), (512, 16, 17), (400, 20, 21), (220, 32, 33)):
        t, y, s, _ = synthetic.simulate_lightcurves([Nb, Nb], seed=5)
        alpha, rho = synthetic.default_hyperparameters(y)
        with gp.Objective(t, y, s, "matern32", slots_per_stream=40) as obj:
            for M, expect in ((takes, takes), (not_any_more, 0)):
                dd = np.stack([np.zeros(M), np.linspace(0, 3, M)], 1)
                before = obj.get_option("chain_count")
                ll, info = obj.loglik_batch(dd, np.tile(alpha, (M, 1)), np.full(M, rho))
                assert (info == 0).all() and obj.get_option("chain_count") - before == expect
            if Nb == 2048:   # 6 evaluations at N = 4096 run as two halves of 3 on two streams: NOT two persistent launches side by side
                dd = np.stack([np.zeros(6), np.linspace(0, 3, 6)], 1)
                before = obj.get_option("chain_count")
                ll6, info = obj.loglik_batch(dd, np.tile(alpha, (6, 1)), np.full(6, rho))
                assert (info == 0).all() and obj.get_option("chain_count") == before
                ll3, _ = obj.loglik_batch(dd[:3], np.tile(alpha, (3, 1)), np.full(3, rho))
                assert obj.get_option("chain_count") == before + 3 and _rel(ll3, ll6[:3]) <= 1e-11


def test_chain_many_calls_and_two_streams(gp):
    """A batch whose LAST group is small (the others are full groups on alternating streams) and 200 single calls in a row: the flag
    words are re-zeroed per launch, nothing accumulates."""
    from gpcc_amd import synthetic
    t, y, s, _ = synthetic.simulate_lightcurves([320, 320], seed=8)
    alpha, rho = synthetic.default_hyperparameters(y)
    with gp.Objective(t, y, s, "matern32", slots_per_stream=16, streams=2) as obj:
        M = 16 * 3 + 5     # groups of 16, 16, 16 and 5 on alternating workspace streams (each with its own flag words): all four are persistent launches
        dd = np.stack([np.zeros(M), np.linspace(0, 9, M)], 1)
        rr = rho * np.linspace(0.9, 1.1, M)   # (distinct hyper-parameters: a fixed-hyper sweep would share its prefix tiles instead)
        ll, info = obj.loglik_batch(dd, np.tile(alpha, (M, 1)), rr)
        assert (info == 0).all() and obj.get_option("chain_count") == M
        obj.set_option("chain_max", 12)   # only the last group
        ll12, info = obj.loglik_batch(dd, np.tile(alpha, (M, 1)), rr)
        assert (info == 0).all() and obj.get_option("chain_count") == M + 5 and _rel(ll12, ll) <= 1e-11
        obj.set_option("chain_max", 0)
        ll0, _ = obj.loglik_batch(dd, np.tile(alpha, (M, 1)), rr)
        assert _rel(ll, ll0) <= 1e-11
        obj.set_option("chain_max", 12)
        first = None
        for _ in range(200):
            v = obj(alpha, rho, [0.0, 2.0])
            first = v if first is None else first
            assert v == first


def test_chain_device_pointer_entry_and_concurrent_handles(gp):
    """The device-pointer entry (gpcc_loglik_batch_device: what one process per GPU calls) with a few evaluations takes the persistent
    launch on the caller's stream and returns the host-pointer entry's bits; four threads with a handle each -- four persistent launches
    competing for the CUs, the shape of tools/concurrent_callers.py -- return what a single caller gets, call after call."""
    import threading

    import torch
    from gpcc_amd import synthetic
    t, y, s, _ = synthetic.simulate_lightcurves([330, 310], seed=9)
    alpha, rho = synthetic.default_hyperparameters(y)
    M = 3
    dd = np.stack([np.zeros(M), np.linspace(0.5, 4.0, M)], 1)
    aa, rr = np.tile(alpha, (M, 1)), np.full(M, rho)
    with gp.Objective(t, y, s, "matern32", slots_per_stream=16) as obj:
        ll, info = obj.loglik_batch(dd, aa, rr)
        before = obj.get_option("chain_count")
        dev = torch.device("cuda:0")
        out, oinfo = obj.loglik_batch_device(torch.tensor(dd, device=dev), torch.tensor(aa, device=dev), torch.tensor(rr, device=dev))
        torch.cuda.synchronize()
        assert obj.get_option("chain_count") == before + M
        assert np.array_equal(out.cpu().numpy(), ll) and (oinfo.cpu().numpy() == 0).all()
    ref = ll[0]
    results = [None] * 4

    def caller(i):
        with gp.Objective(t, y, s, "matern32", slots_per_stream=16, streams=1) as o:
            results[i] = [o(alpha, rho, dd[0]) for _ in range(60)]

    th = [threading.Thread(target=caller, args=(i,)) for i in range(4)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    for r in results:
        assert r is not None and all(v == ref for v in r)


def test_chain_worker_count_changes_no_bit(gp):
    """The size of the persistent launch is a scheduling matter: chain_workers_max (what caller processes sharing a GPU set) changes
    the grid, never a result -- a tile's sums do not depend on which workgroup forms them, and any number of workers >= 1 drains the
    queue."""
    from gpcc_amd import synthetic
    t, y, s, _ = synthetic.simulate_lightcurves([700, 600], seed=10)   # 11 tile steps
    alpha, rho = synthetic.default_hyperparameters(y)
    M = 2
    dd = np.stack([np.zeros(M), np.array([0.7, 2.9])], 1)
    aa, rr = np.tile(alpha, (M, 1)), np.full(M, rho)
    with gp.Objective(t, y, s, "matern52", slots_per_stream=16, streams=1) as obj:
        assert obj.get_option("chain_workers_max") == 0
        ll, info = obj.loglik_batch(dd, aa, rr)
        full = obj.get_option("chain_last_grid")
        assert (info == 0).all() and full > 100
        for cap in (1, 7, 40):
            obj.set_option("chain_workers_max", cap)
            ll_c, info_c = obj.loglik_batch(dd, aa, rr)
            grid = obj.get_option("chain_last_grid")
            assert np.array_equal(ll_c, ll) and (info_c == 0).all() and grid < full and grid <= max(12 + cap, 48), (cap, grid)
        obj.set_option("chain_workers_max", 0)
        ll_c, _ = obj.loglik_batch(dd, aa, rr)
        assert np.array_equal(ll_c, ll) and obj.get_option("chain_last_grid") == full
        with pytest.raises(Exception):
            obj.set_option("chain_workers_max", -1)


def test_call_of_more_evaluations_than_workspace_slots(gp):
    """A handle whose workspace has fewer slots than a call the policy would give to the persistent launch (8 slots, 12 and 30
    evaluations): the call runs group by group on the general path -- never one launch over more slots than exist."""
    from gpcc_amd import synthetic
    t, y, s, _ = synthetic.simulate_lightcurves([260, 250], seed=15)
    alpha, rho = synthetic.default_hyperparameters(y)
    M = 30
    dd = np.stack([np.zeros(M), np.linspace(0.2, 6.0, M)], 1)
    aa, rr = np.tile(alpha, (M, 1)), rho * np.linspace(0.8, 1.2, M)
    with gp.Objective(t, y, s, "matern32", slots_per_stream=40) as obj:
        ref, rinfo = obj.loglik_batch(dd, aa, rr)
        assert obj.get_option("chain_count") == M and (rinfo == 0).all()
    with gp.Objective(t, y, s, "matern32", slots_per_stream=8, streams=1) as obj:
        for m in (12, 30):
            ll, info = obj.loglik_batch(dd[:m], aa[:m], rr[:m])
            assert (info == 0).all() and np.array_equal(ll, ref[:m])   # (groups of 8 take the persistent launch one after the other: the same bits)
        assert obj.get_option("workspace_slots") == 8


def test_fp32_multi_device_handle_hands_few_evaluations_to_the_fp64_twins(gp):
    """A multi-device fp32 handle (two sub-handles on device 0) splits a call of four evaluations into two shares of two: each
    sub-handle's fp64 twin evaluates its share with the persistent launch -- the fp64 handle's bits."""
    from gpcc_amd import synthetic
    t, y, s, _ = synthetic.simulate_lightcurves([330, 300], seed=14)
    alpha, rho = synthetic.default_hyperparameters(y)
    M = 4
    dd = np.stack([np.zeros(M), np.linspace(0.4, 3.1, M)], 1)
    aa, rr = np.tile(alpha, (M, 1)), np.full(M, rho)
    with gp.Objective(t, y, s, "OU") as obj:
        ll64, info64 = obj.loglik_batch(dd, aa, rr)
    with gp.Objective(t, y, s, "OU", precision="fp32", devices=[0, 0]) as multi:
        ll, info = multi.loglik_batch(dd, aa, rr)
    assert (info64 == 0).all() and (info == 0).all() and np.array_equal(ll, ll64)


def test_chain_column_blocks_change_no_bit(gp, oracle):
    """Bulk update jobs of 1, 2, 4 or 8 columns (chain_batch; forced on at this size by chain_batch_min = 0): the same sums in the same order,
    so the same bits -- single evaluations and a group, with and without the helper workgroups and quarter-tile jobs -- and the oracle's
    value.  (The job order itself is checked for every size on the host: tests/abi/chain_queue_check.cpp.)"""
    from gpcc_amd import synthetic
    t, y, s, _ = synthetic.simulate_lightcurves([1500, 1400], seed=12)   # 23 tile steps: blocks of 8 exist
    alpha, rho = synthetic.default_hyperparameters(y)
    M = 7
    dd = np.stack([np.zeros(M), np.linspace(0.3, 5.0, M)], 1)
    aa, rr = np.tile(alpha, (M, 1)) * np.linspace(0.8, 1.2, M)[:, None], np.full(M, rho)
    ref, rinfo = oracle.loglik_batch("matern32", t, y, s, dd[:2], aa[:2], rr[:2], True, nthreads=2)
    with gp.Objective(t, y, s, "matern32", slots_per_stream=16, streams=1) as obj:
        assert obj.get_option("chain_batch") == 8
        obj.set_option("chain_work_max", 1 << 30)
        obj.set_option("chain_batch_min", 0)
        out = {}
        for w in (1, 2, 4, 8):
            obj.set_option("chain_batch", w)
            before = obj.get_option("chain_count")
            one = np.array([obj(aa[i], rr[i], dd[i]) for i in range(2)])
            grp, info = obj.loglik_batch(dd, aa, rr)
            assert obj.get_option("chain_count") == before + 2 + M and (info == 0).all()
            out[w] = (one, grp)
        for w in (2, 4, 8):
            assert np.array_equal(out[w][0], out[1][0]) and np.array_equal(out[w][1], out[1][1]), w
        assert np.array_equal(out[8][0], out[8][1][:2])   # an evaluation does not depend on the group it travels in
        assert (rinfo == 0).all() and _rel(out[8][0], ref) <= 1e-8
        with pytest.raises(Exception):
            obj.set_option("chain_batch", 3)
