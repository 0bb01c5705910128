"""Round 4: the per-delay FIT on a multi-device handle is sharded BY DELAY (device i fits the delays i, i + n, ... with its own
lock-step optimiser and every fast path of a single-device fit), ONE gather at the end (gpcc_hip.hip: multi_grid_loglik).
One-GPU rehearsals with repeated device ids (host gather); the RCCL branch of the same function runs as a one-rank communicator
(GPCC_MULTI_FORCE_RCCL=1) and, where >= 2 GPUs exist, for real."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gp():
    import torch
    torch.cuda.init()
    import gpcc_amd
    return gpcc_amd


def _gpu_count():
    import torch
    return torch.cuda.device_count()


def _check_fit(gp, Nl, devs, G, iterations, monkeypatch=None, restarts=1):
    from gpcc_amd import synthetic
    t, y, s, _ = synthetic.simulate_lightcurves(Nl, seed=5)
    L = len(Nl)
    rng = np.random.default_rng(7)
    cand = np.concatenate([np.zeros((G, 1)), rng.random((G, L - 1)) * 6.0], 1)
    with gp.Objective(t, y, s, gp.matern32) as single:
        ref = single.grid_loglik(cand, iterations, numberofrestarts=restarts, rhomax=300.0, seed=4)
    with gp.Objective(t, y, s, gp.matern32, devices=devs) as multi:
        fit = multi.grid_loglik(cand, iterations, numberofrestarts=restarts, rhomax=300.0, seed=4)
        n = len(devs)
        blk = -(-G // n)
        rows = [multi.gathered(w) for w in range(n)]
        comp_ms, gather_ms, total_ms = multi.multi_stats()
        assert multi.get_option("gather_width") == L + 4
    names = ("loglik", "alpha", "rho", "info", "iterations")
    for name, a, b in zip(names, fit[:5], ref[:5]):
        assert np.array_equal(a, b), (name, devs)                         # BITWISE the single-device fit, delay by delay
    for w in range(n):                                                    # every device holds every fitted delay
        r = rows[w]
        assert r.shape == (n, blk, L + 4)
        for i in range(n):
            for j in range(blk):
                g = j * n + i
                if g < G:
                    assert r[i, j, 0] == ref[0][g] and r[i, j, 3] == ref[2][g] and np.array_equal(r[i, j, 4:], ref[1][g])
                    assert int(r[i, j, 1]) == ref[3][g] and int(r[i, j, 2]) == ref[4][g]
                else:
                    assert np.isnan(r[i, j, 0])
    assert comp_ms.shape == (n,) and np.all(comp_ms[:min(n, G)] > 0) and total_ms >= comp_ms.max() * 0.5
    return ref


@pytest.mark.parametrize("Nl,G,iters", [([60, 50], 101, 60), ([60, 50, 40], 37, 40)])
def test_fit_sharded_by_delay_matches_single_device(gp, Nl, G, iters):
    """README sizes (N = 110, 150; the small-N fast path with device unpack, speculation and slices inside every sub-handle):
    {0,0,0,0}, {0,0,0} (an uneven deal) and {0} against the single-device fit, bit for bit."""
    for devs in ([0, 0, 0, 0], [0, 0, 0], [0]):
        _check_fit(gp, Nl, devs, G, iters)


def test_fit_sharded_by_delay_fewer_delays_than_devices_and_restarts(gp):
    _check_fit(gp, [60, 50], [0, 0, 0, 0], 3, 30)
    _check_fit(gp, [60, 50], [0, 0], 9, 25, restarts=3)


def test_fit_sharded_by_delay_tile_kernels(gp):
    """N = 580 (tile kernels): the sub-handles' groups are smaller than the single handle's, so pin the size-dependent choices
    like test_multi_device_handle_matches_single_device does -- here the defaults, compared to rounding."""
    from gpcc_amd import synthetic
    t, y, s, _ = synthetic.simulate_lightcurves([300, 280], seed=8)
    cand = np.stack([np.zeros(7), np.linspace(0.5, 9.0, 7)], 1)
    with gp.Objective(t, y, s, gp.matern32) as single:
        ref = single.grid_loglik(cand, 6, seed=3)
    with gp.Objective(t, y, s, gp.matern32, devices=[0, 0, 0]) as multi:
        fit = multi.grid_loglik(cand, 6, seed=3)
    np.testing.assert_allclose(fit[0], ref[0], rtol=1e-10)
    assert np.array_equal(fit[3], ref[3])


def test_fit_gather_through_rccl_with_one_rank(gp, monkeypatch):
    """GPCC_MULTI_FORCE_RCCL=1: device_ids = [0] as a ONE-rank RCCL communicator -- the ncclAllGather branch of the fit's gather
    executes (the multi-rank collective needs >= 2 GPUs: below)."""
    monkeypatch.setenv("GPCC_MULTI_FORCE_RCCL", "1")
    from gpcc_amd import synthetic
    t, y, s, _ = synthetic.simulate_lightcurves([60, 50], seed=5)
    cand = np.stack([np.zeros(11), np.linspace(0.0, 5.0, 11)], 1)
    with gp.Objective(t, y, s, gp.matern32) as single:
        ref = single.grid_loglik(cand, 30, seed=4)
    with gp.Objective(t, y, s, gp.matern32, devices=[0]) as multi:
        assert multi.get_option("gather_mode") == 1
        fit = multi.grid_loglik(cand, 30, seed=4)
        rows = multi.gathered(0)
    for a, b in zip(fit[:5], ref[:5]):
        assert np.array_equal(a, b)
    assert np.array_equal(rows[0, :, 0], ref[0])


@pytest.mark.skipif(_gpu_count() < 2, reason="needs >= 2 GPUs (the driver's multi-GPU box)")
def test_fit_sharded_over_real_devices_with_rccl_gather(gp):
    n = min(_gpu_count(), 8)
    ref = _check_fit(gp, [60, 50, 40], list(range(n)), 53, 40)
    assert np.isfinite(ref[0]).all()
