"""gpcc.jl_amd -- MI355X-native marginal-log-likelihood hot path of GPCC.jl (import as ``gpcc_amd``)."""
