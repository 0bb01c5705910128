#!/usr/bin/env julia
# tools/pin_reference.jl -- pins this repository's oracle to the REAL reference (HITS-AIN/GPCC.jl).
#
# NEVER EXECUTED in the build image: it has no Julia toolchain (SURVEY.md 8(c)).  It is the "first session on any machine
# that has Julia + the packages" step: until its output exists, every parity claim of this repository reads "against the
# build's CPU restatement of the reference, reference itself not executable" (DESIGN.md 2).
#
# What it does: feeds the committed fixtures (tests/golden/gpcc_golden.json: 44 log-likelihood cases, rectangular
# covariances, a non-PD case, getprobabilities vectors) through the reference's OWN functions and writes the deviations to
# tests/golden/reference_deltas.json:
#   1. GPCC.delayedCovariance                       src/delayedCovariance.jl:1-38      vs "Kxy" / "Kxx"
#   2. the body of the closure objective(α, ρ)       src/gpccfixdelay_marginaliseb.jl:85-98, :133-141
#      (fixed-b: src/gpccfixdelay.jl:85-96, :131-139; that file is not include()d by src/GPCC.jl, so its four lines are
#      spelled out below with the package's own delayedCovariance / makematrixsymmetric! / MvNormal)
#   3. GPCC.getprobabilities (both methods)          src/getprobabilities.jl:1-20
#   4. MiscUtil.makepositive / transformbetween (+ inverses) and nearestposdef on a grid -- the build restated these from
#      their names (gpcc.jl_amd/csrc/gpcc_fit.h:24-31: softplus, logistic, eigenvalue lift); this prints what they really are
#   5. one short gpcc(...; iterations = 20, seed = 1) run on the first 2-band case: minimum, α, ρ, postb -- the optimiser
#      trajectory (Optim's Nelder-Mead, MersenneTwister draws) is unpinned in the build, so only the record is kept.
#   6. Optim's NelderMead() on a deterministic objective from a fixed start, with the reference's options: per-iteration
#      value / g_tol measure / step type / centroid ("optim_neldermead_trace") -- what pins gpcc_neldermead_batch.
#
# Usage (from an environment where `using GPCC` works, e.g. the package's own project with MiscUtil dev'ed in):
#     julia --project=/path/to/GPCC.jl tools/pin_reference.jl [/path/to/this/repo]
# Then, in the repo:  python -m pytest tests/test_reference_deltas.py   (asserts every delta <= 1e-10 relative).

using Pkg
try
    @eval using JSON
catch
    Pkg.add("JSON")
    @eval using JSON
end
using Optim            # a dependency of GPCC.jl (Project.toml:12)
using GPCC, MiscUtil, Distributions, LinearAlgebra, Statistics, Random, Printf

root = length(ARGS) >= 1 ? ARGS[1] : normpath(joinpath(@__DIR__, ".."))
golden = JSON.parsefile(joinpath(root, "tests", "golden", "gpcc_golden.json"))
kern = Dict("OU" => GPCC.OU, "rbf" => GPCC.rbf, "matern32" => GPCC.matern32, "matern52" => GPCC.matern52)
vv(a) = [Float64.(collect(v)) for v in a]
mat(rows) = permutedims(reduce(hcat, [Float64.(collect(r)) for r in rows]))   # JSON list of rows -> Matrix
relerr(a, b) = maximum(abs.(a .- b) ./ max.(abs.(b), 1e-300))

# ---- 2. the closure body, verbatim in structure -------------------------------------------------------------------
function reference_objective(kernel, tarray, yarray, stdarray, τ, α, ρ, marginalise_b::Bool)
    Y = reduce(vcat, yarray)                                  # marginaliseb.jl:85
    Q = GPCC.Qmatrix(length.(tarray))                         # :87, util.jl:56-70
    Sobs = Diagonal(reduce(vcat, stdarray) .^ 2)              # :89
    if marginalise_b
        μb = map(mean, yarray)                                # :92
        Σb = 100 * diagm(map(var, yarray))                    # :94
        B = Q * Σb * Q'                                       # :96
        b̄ = Q * μb                                            # :98
        K = GPCC.delayedCovariance(kernel, α, τ, ρ, tarray) + Sobs + B   # :135
        MiscUtil.makematrixsymmetric!(K)                      # :137
        return logpdf(MvNormal(b̄, K), Y)                      # :139
    else
        b = (Q'Q) \ Q' * Y                                    # gpccfixdelay.jl:94
        K = GPCC.delayedCovariance(kernel, α, τ, ρ, tarray) + Sobs        # gpccfixdelay.jl:133
        MiscUtil.makematrixsymmetric!(K)
        return logpdf(MvNormal(Q * b, K), Y)                  # gpccfixdelay.jl:137
    end
end

out = Dict{String,Any}("julia" => string(VERSION), "note" => "deltas of the build's fixtures against the real reference")

# ---- 1. delayedCovariance -------------------------------------------------------------------------------------------
covs = []
for c in golden["covariances"]
    k = kern[c["kernel"]]
    x, y = vv(c["x"]), vv(c["y"])
    scale, delays, ρ = Float64.(c["scale"]), Float64.(c["delays"]), Float64(c["rho"])
    Kxy = GPCC.delayedCovariance(k, scale, delays, ρ, x, y)
    Kxx = GPCC.delayedCovariance(k, scale, delays, ρ, x)
    push!(covs, Dict("kernel" => c["kernel"], "rel_Kxy" => relerr(Kxy, mat(c["Kxy"])), "rel_Kxx" => relerr(Kxx, mat(c["Kxx"]))))
end
out["covariances"] = covs

# ---- 2. objective ---------------------------------------------------------------------------------------------------
cases = []
for (i, c) in enumerate(golden["cases"])
    ll = reference_objective(kern[c["kernel"]], vv(c["t"]), vv(c["y"]), vv(c["sigma"]), Float64.(c["delays"]),
                             Float64.(c["alpha"]), Float64(c["rho"]), Bool(c["marginalise_b"]))
    push!(cases, Dict("index" => i - 1, "kernel" => c["kernel"], "marginalise_b" => c["marginalise_b"], "reference_loglik" => ll,
                      "fixture_loglik" => c["loglik"], "rel" => abs(ll - c["loglik"]) / abs(c["loglik"])))
end
out["cases"] = cases
c = golden["nonpd"]
out["nonpd_throws_PosDefException"] = try
    reference_objective(kern[c["kernel"]], vv(c["t"]), vv(c["y"]), vv(c["sigma"]), Float64.(c["delays"]), Float64.(c["alpha"]),
                        Float64(c["rho"]), Bool(c["marginalise_b"]))
    false
catch e
    isa(e, PosDefException)
end

# ---- 3. getprobabilities -------------------------------------------------------------------------------------------
p = golden["probabilities"]
ll, lp = Float64.(p["loglik"]), Float64.(p["logprior"])
out["probabilities"] = Dict("rel_flat" => relerr(getprobabilities(ll), Float64.(p["p_flat"])),
                            "rel_prior" => relerr(getprobabilities(ll, lp), Float64.(p["p_prior"])))

# ---- 4. MiscUtil's transforms: what they really are ------------------------------------------------------------------
xs = collect(-30.0:2.5:30.0)
softplus(x) = x > 0 ? x + log1p(exp(-x)) : log1p(exp(x))          # the build's reading of makepositive
logistic_between(x, a, b) = a + (b - a) / (1 + exp(-x))            # the build's reading of transformbetween
out["miscutil"] = Dict(
    "x" => xs,
    "makepositive" => MiscUtil.makepositive.(xs),
    "makepositive_minus_softplus" => maximum(abs.(MiscUtil.makepositive.(xs) .- softplus.(xs))),
    "transformbetween_0.1_20" => [MiscUtil.transformbetween(x, 0.1, 20.0) for x in xs],
    "transformbetween_minus_logistic" => maximum(abs.([MiscUtil.transformbetween(x, 0.1, 20.0) for x in xs] .- logistic_between.(xs, 0.1, 20.0))),
    "inverse_roundtrip" => maximum(abs.([MiscUtil.invmakepositive(MiscUtil.makepositive(x)) for x in xs[5:end-4]] .- xs[5:end-4])))
let A = [2.0 1.0 0.0; 1.0 -0.5 0.3; 0.0 0.3 1e-9]
    N = MiscUtil.nearestposdef(A; minimumeigenvalue = 1e-6)        # marginaliseb.jl:331
    E = eigen(Symmetric(A))
    lifted = E.vectors * Diagonal(max.(E.values, 1e-6)) * E.vectors'   # the build's reading (gpcc.jl_amd/fit.py: nearestposdef)
    out["nearestposdef"] = Dict("result" => [N[i, :] for i in 1:3], "minus_eigenvalue_lift" => maximum(abs.(N .- lifted)))
end

# ---- 5. one short end-to-end run (record only) -----------------------------------------------------------------------
let c = first(filter(c -> length(c["t"]) == 2 && c["marginalise_b"] && c["kernel"] == "OU", golden["cases"]))
    res = gpcc(vv(c["t"]), vv(c["y"]), vv(c["sigma"]); kernel = GPCC.OU, delays = Float64.(c["delays"]), iterations = 20, seed = 1,
               numberofrestarts = 1, initialrandom = 5, rhomin = 0.1, rhomax = 20.0)
    loglikel, pred, rest = res[1], res[2], res[3:end]             # (loglikel, predictTest, (α, postb, ρ)) or flattened
    α, postb, ρ = length(rest) == 1 ? rest[1] : rest
    out["gpcc_iterations20_seed1"] = Dict("loglikel" => loglikel, "alpha" => α, "rho" => ρ, "postb_mean" => mean(postb),
                                          "postb_cov" => [cov(postb)[i, :] for i in 1:length(α)], "delays" => c["delays"])
end

# ---- 6. Optim's Nelder-Mead, iteration by iteration (pins gpcc_neldermead_batch / gpcc_fit.h / neldermead.py) ----------------
# A deterministic objective with no randomness and no MiscUtil in it: the reference's own closure at FIXED delays on one
# fixture, minimised from a FIXED start in (log alpha_1, log alpha_2, log rho) coordinates with exactly the options the
# reference passes (marginaliseb.jl:205-211).  Recorded per iteration: f at the best vertex, the g_tol measure Optim tests,
# the step type and the centroid (extended trace) -- enough to compare decision by decision with
# tests/test_neldermead_cpu.py::reference_neldermead on the same objective.
let c = first(filter(c -> length(c["t"]) == 2 && c["marginalise_b"] && c["kernel"] == "matern32", golden["cases"]))
    t, y, sg, τ = vv(c["t"]), vv(c["y"]), vv(c["sigma"]), Float64.(c["delays"])
    negobj(x) = try
        -reference_objective(GPCC.matern32, t, y, sg, τ, exp.(x[1:2]), exp(x[3]), true)
    catch e
        isa(e, PosDefException) ? Inf : rethrow(e)
    end
    x0 = [log.(Float64.(c["alpha"])); log(Float64(c["rho"]))] .+ [0.3, -0.2, 0.5]
    opt = Optim.Options(iterations = 60, show_trace = false, store_trace = true, extended_trace = true, g_tol = 1e-6)
    res = Optim.optimize(negobj, x0, Optim.NelderMead(), opt)
    tr = Optim.trace(res)
    out["optim_neldermead_trace"] = Dict(
        "x0" => x0, "objective" => "-objective(exp.(x[1:2]), exp(x[3])) of golden case (2 bands, marginalise_b, matern32), delays fixed",
        "iterations" => Optim.iterations(res), "f_calls" => Optim.f_calls(res), "minimum" => Optim.minimum(res),
        "minimizer" => Optim.minimizer(res), "converged" => Optim.converged(res),
        "value" => [s.value for s in tr], "g_norm" => [s.g_norm for s in tr],
        "step_type" => [get(s.metadata, "step_type", "") for s in tr],
        "centroid" => [get(s.metadata, "centroid", Float64[]) for s in tr])
end

worst = maximum(vcat([d["rel"] for d in cases], [d["rel_Kxy"] for d in covs], [d["rel_Kxx"] for d in covs],
                     [out["probabilities"]["rel_flat"], out["probabilities"]["rel_prior"]]))
out["worst_rel"] = worst
open(joinpath(root, "tests", "golden", "reference_deltas.json"), "w") do f
    JSON.print(f, out, 1)
end
@printf("worst relative deviation of the fixtures from the reference: %.3e (bar 1e-10)\n", worst)
@printf("makepositive - softplus: %.3e   transformbetween - logistic: %.3e   nearestposdef - eigenvalue lift: %.3e\n",
        out["miscutil"]["makepositive_minus_softplus"], out["miscutil"]["transformbetween_minus_logistic"],
        out["nearestposdef"]["minus_eigenvalue_lift"])
