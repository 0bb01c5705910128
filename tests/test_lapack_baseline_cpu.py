"""oracle/lapack_baseline.py (bench.py's cpu_baseline leg: C-restatement assembly + OpenBLAS dpotrf/dtrtrs, own process)
against the oracle's own Cholesky on a small problem, both parallel shapes."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lapack_baseline_matches_oracle(oracle):
    from gpcc_amd import synthetic
    delays = [[0.0, 0.0], [0.0, 2.0], [0.0, 7.5]]
    run = subprocess.run([sys.executable, "-m", "oracle.lapack_baseline", "--n-per-band", "96", "--delays", json.dumps(delays),
                          "--workers", "2", "--evals-per-worker", "2", "--kernel", "matern52", "--seed", "5"],
                         capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert run.returncode == 0, run.stderr[-2000:]
    rec = json.loads(run.stdout.strip().splitlines()[-1])
    t, y, s, _ = synthetic.simulate_lightcurves([96, 96], seed=5)
    alpha, rho = synthetic.default_hyperparameters(y)
    ref, info = oracle.loglik_batch("matern52", t, y, s, delays, np.tile(alpha, (3, 1)), np.full(3, rho), True)
    assert (info == 0).all() and rec["info"] == [0, 0, 0]
    np.testing.assert_allclose(rec["loglik"], ref, rtol=1e-11)
    np.testing.assert_allclose(rec["blas_loglik"], ref, rtol=1e-11)
    assert rec["pmap"]["workers"] == 2 and rec["pmap"]["evals"] == 4 and rec["blas"]["evals_per_s"] > 0
