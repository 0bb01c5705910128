# gpcchip.jl -- the reference-side binding of libgpcc_hip.so (include/gpcc_hip.h) a GPCC.jl maintainer would add as
# src/gpcchip.jl and `include` from src/GPCC.jl.  INTEGRATION.md section 1 explains it and shows the two edits inside GPCC.jl.
# NEVER EXECUTED: the build image has no Julia toolchain (SURVEY.md section 8(c)); syntax-reviewed only.  The same C ABI is
# exercised end to end by the Python ctypes host (gpcc.jl_amd/) and from plain C (tests/abi/abi_smoke.c).
# Thin ccall layer over include/gpcc_hip.h.  Host logic (parameter transforms, Nelder–Mead,
# restarts, printing) stays in gpccfixdelay_marginaliseb.jl untouched.
module GPCCHip

using Random, Statistics, LinearAlgebra, Distributions, MiscUtil    # all already dependencies of GPCC.jl

const LIB = get(ENV, "GPCC_HIP_LIB", "libgpcc_hip.so")

# Several Julia tasks / threads of ONE process, each with its own handle, each calling objective(α, ρ) (the pmap shape inside a process):
# the HIP runtime spreads a process's streams over 4 hardware queues by default, so at most 4 launches run side by side.  8 queues:
# 14 700 - 18 200 instead of 10 300 - 13 700 evaluations/s in total for 8 callers at N = 1024 (profiles/r05/concurrent_callers_round5b.log).
# Must be in the environment before the first HIP call of the process, i.e. before the library is used.
haskey(ENV, "GPU_MAX_HW_QUEUES") || (ENV["GPU_MAX_HW_QUEUES"] = "8")

# kernel function identity -> id (src/util.jl:15-52); any other callable stays on the Julia path
# (`include`d from src/GPCC.jl, this module's parent IS GPCC -- `Main.GPCC` would only exist after `using GPCC` in Main)
const _G = parentmodule(@__MODULE__)
kernelid(k) = k === _G.OU ? 0 : k === _G.rbf ? 1 :
              k === _G.matern32 ? 2 : k === _G.matern52 ? 3 : -1

lasterror(h) = unsafe_string(ccall((:gpcc_last_error, LIB), Cstring, (Ptr{Cvoid},), h))

mutable struct Handle
    ptr::Ptr{Cvoid}
    L::Int
end

"Replaces the precompute at gpccfixdelay_marginaliseb.jl:85-98 (marginalise_b=false: gpccfixdelay.jl:85-96)."
function create(tarray, yarray, stdarray; kernel, marginalise_b = true, device = 0)
    id = kernelid(kernel)
    id < 0 && return nothing                      # caller keeps the pure-Julia objective
    L  = length(tarray)
    Nl = Cint.(length.(tarray))
    t, y, s = reduce(vcat, tarray), reduce(vcat, yarray), reduce(vcat, stdarray)   # band order, user order
    ref = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:gpcc_create, LIB), Cint,
               (Ref{Ptr{Cvoid}}, Cint, Ptr{Cint}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint, Cint, Cint),
               ref, L, Nl, Float64.(t), Float64.(y), Float64.(s), id, marginalise_b ? 1 : 0, 0, device)
    rc == 0 || error("gpcc_create: " * lasterror(C_NULL))
    h = Handle(ref[], L)
    finalizer(x -> ccall((:gpcc_destroy, LIB), Cint, (Ptr{Cvoid},), x.ptr), h)
    return h
end

"objective for M triples.  delays, alpha: L×M Matrix{Float64} (column-major L×M == row-major M×L of the ABI)."
function loglik_batch(h::Handle, delays::Matrix{Float64}, alpha::Matrix{Float64}, rho::Vector{Float64})
    M = length(rho)
    @assert size(delays) == (h.L, M) && size(alpha) == (h.L, M)
    ll, info = Vector{Float64}(undef, M), Vector{Cint}(undef, M)
    rc = ccall((:gpcc_loglik_batch, LIB), Cint,
               (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cint}),
               h.ptr, M, delays, alpha, rho, ll, info)
    rc == 0 || error("gpcc_loglik_batch: " * lasterror(h.ptr))
    return ll, info
end

"Drop-in body of objective(α, ρ) (gpccfixdelay_marginaliseb.jl:133-141): throws what the Julia code throws."
function objective(h::Handle, τ, α, ρ)
    ll, info = loglik_batch(h, reshape(Float64.(τ), :, 1), reshape(Float64.(α), :, 1), [Float64(ρ)])
    info[1] == -1 && throw(AssertionError("all(scale .> 0)"))            # delayedCovariance.jl:3
    info[1] == -2 && error("ρ=$(ρ) is <= 0")                             # delayedCovariance.jl:5-7
    info[1] >  0 && throw(LinearAlgebra.PosDefException(info[1]))        # cholesky inside MvNormal, :139
    return ll[1]
end

"""
The whole README sweep `map(d -> gpcc(...; delays = [0; d])[1], candidatedelays)` (README.md:172-174) as one call.
`delays` is L×G.  The random candidates are drawn HERE exactly as gpccfixdelay_marginaliseb.jl:160-196 draws them
(same MersenneTwister(seed), same order), so every delay starts where the reference starts it.
"""
function grid_loglik(h::Handle, delays::Matrix{Float64}, yarray; iterations, seed = 1, numberofrestarts = 1,
                     initialrandom = 5, rhomin = 0.1, rhomax)
    L, G = size(delays)
    rg = MersenneTwister(seed)                                                   # :62
    initialρ = numberofrestarts <= 2 ? rand(rg, Uniform(rhomin + 1e-3, rhomax - 1e-3), numberofrestarts) :
                                        collect(MiscUtil.logrange(rhomin + 1e-3, rhomax - 1e-3, numberofrestarts))
    init = Array{Float64}(undef, L + 1, initialrandom, numberofrestarts)        # column-major == R×C×(L+1) row-major
    for i in 1:numberofrestarts, c in 1:initialrandom
        α = map(var, yarray) .* (rand(rg, L) * (1.2 - 0.8) .+ 0.8)             # sampleα, :188
        init[:, c, i] = [invmakepositive.(α); invtransformbetween(initialρ[i], rhomin, rhomax)]   # :195-196
    end
    ll, ρ, info = Vector{Float64}(undef, G), Vector{Float64}(undef, G), Vector{Cint}(undef, G)
    α = Matrix{Float64}(undef, L, G)
    rc = ccall((:gpcc_grid_loglik, LIB), Cint,
               (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Cint, Cint, Cint, Cdouble, Cdouble, Culonglong, Ptr{Cdouble},
                Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cint}, Ptr{Cint}, Ptr{Clonglong}),
               h.ptr, G, delays, iterations, numberofrestarts, initialrandom, rhomin, rhomax, seed, init,
               ll, α, ρ, info, C_NULL, C_NULL)
    rc == 0 || error("gpcc_grid_loglik: " * lasterror(h.ptr))
    return ll, α, ρ, info
end

"getprobabilities(loglikel[, logprior]) (getprobabilities.jl:1-20); same shape as the input."
function probabilities(loglikel::Array{Float64}, logprior = nothing; device = 0)
    out = similar(loglikel)
    lp  = logprior === nothing ? C_NULL : Float64.(vec(logprior))        # ccall roots the array for the call
    rc = ccall((:gpcc_probabilities, LIB), Cint, (Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint),
               length(loglikel), loglikel, lp, out, device)
    rc == 0 || error("gpcc_probabilities: " * lasterror(C_NULL))
    return out
end

end # module
