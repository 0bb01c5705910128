"""gpcc.jl_amd -- MI355X-native marginal-log-likelihood hot path of GPCC.jl (import as ``gpcc_amd``).

Public surface mirrors the reference for this path: the kernel tokens OU / rbf / matern32 /
matern52, delayedCovariance, getprobabilities, and Objective (the objective(alpha, rho) closure
of gpccfixdelay) -- all backed by csrc/libgpcc_hip.so through the C ABI of include/gpcc_hip.h."""
from . import synthetic  # noqa: F401
from ._capi import GpccError  # noqa: F401
from .api import (KERNELS, OU, Kernel, Objective, PosDefException, build_info, delayedCovariance,  # noqa: F401
                  getprobabilities, matern32, matern52, mvnormal_logpdf, rbf, selftest)
from .distributed import shard_bounds, sharded_grid_fit, sharded_loglik  # noqa: F401
from .fit import GridFit, Predictor, gpcc, gpcc_grid, singlegp, uniformpriordelay  # noqa: F401,E402
