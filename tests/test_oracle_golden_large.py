"""The C oracle against the scipy/LAPACK restatement at BASELINE.json sizes
(tests/golden/make_golden_large.py -> gpcc_golden_large.json).  The light curves are regenerated from
gpcc_amd.synthetic seeds; a checksum stored with every case tells generator drift from a wrong result.
cfg5 (N = 16384: ~25 min of scalar C Cholesky per evaluation) is left to the GPU suite."""
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def large():
    with open(os.path.join(ROOT, "tests", "golden", "gpcc_golden_large.json")) as f:
        return json.load(f)


def regenerate(case):
    from gpcc_amd import synthetic
    t, y, s, _ = synthetic.simulate_lightcurves(case["Nl"], seed=case["seed"], sigma=case["sigma"])
    chk = [float(np.sum(np.concatenate(t))), float(np.sum(np.concatenate(y))), float(np.sum(np.concatenate(s) ** 2))]
    np.testing.assert_allclose(chk, case["data_checksum"], rtol=1e-12, err_msg="synthetic generator drifted")
    return t, y, s


def test_fixture_covers_baseline_configs(large):
    tags = {c["tag"] for c in large["cases"]}
    assert {"cfg2", "cfg3", "cfg4", "cfg5", "illcond"} <= tags
    assert {c["kernel"] for c in large["cases"] if c["tag"] == "cfg3"} == {"OU", "rbf", "matern32", "matern52"}
    assert any(c["Nl"] == [8192, 8192] and c["kernel"] == "matern52" for c in large["cases"])


def test_oracle_matches_lapack_restatement_at_baseline_sizes(large, oracle):
    """N = 2048 (cfg2, ill-conditioned cases), N = 4096 (cfg3: one delay per kernel + the fixed-b case) and
    N = 4095 (cfg4); evaluations of one light-curve set go through the oracle as one threaded batch."""
    groups = {}
    for c in large["cases"]:
        if c["tag"] == "cfg5":
            continue
        if c["tag"] == "cfg3" and c["marginalise_b"] and c["delays"] != [0.0, 2.0]:
            continue
        key = (tuple(c["Nl"]), c["seed"], c["sigma"], c["kernel"], c["marginalise_b"])
        groups.setdefault(key, []).append(c)
    worst = 0.0
    for key, cs in groups.items():
        t, y, s = regenerate(cs[0])
        ll, info = oracle.loglik_batch(cs[0]["kernel"], t, y, s, [c["delays"] for c in cs], [c["alpha"] for c in cs],
                                       [c["rho"] for c in cs], cs[0]["marginalise_b"], nthreads=len(cs))
        assert (info == 0).all()
        for c, v in zip(cs, ll):
            rel = abs(v - c["loglik"]) / abs(c["loglik"])
            worst = max(worst, rel)
            # (the adversarial cases have cond(K) ~ 1e10: two fp64 Cholesky factorisations differ by ~2e-10 there)
            assert rel <= (1e-8 if c["tag"] == "adversarial" else 1e-10), (c["tag"], c["kernel"], rel)
    print("oracle vs LAPACK restatement at N = 2048..4096: worst rel %.2e" % worst)
