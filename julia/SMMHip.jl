# SMMHip.jl — Julia host layer over libsmmhip.so (the C ABI of include/smmhip.h).
#
# Stdlib only (Libdl), so that it loads without a package registry.  It gives SMM.jl's BGP path
# a `backend = hip` without touching the rest of the package: `MAlgoBGPHip <: MAlgo` plugs into
# `run!` through the one method SMM.jl dispatches on, `computeNextIteration!(algo)`
# (src/mopt/AlgoAbstract.jl:45), and materialises `BGPChain`-shaped views for the readers
# (`history`, `summary`, `params`, plotting; src/mopt/AlgoBGP.jl:117-206).
#
# NOT EXERCISED IN THIS REPOSITORY'S CI: the build image has no `julia` binary.  The same ABI is
# exercised through the ctypes binding (smm.jl_amd/_abi.py, tests/); struct layouts below mirror
# include/smmhip.h field for field (tests/test_abi.py checks the C side of that contract).
module SMMHip

using Libdl

export MAlgoBGPHip, hip_run!, hip_step!, hip_history, hip_state, hip_eval_batch, hip_register_objective

const LIB = Ref{Ptr{Cvoid}}(C_NULL)

function __init__()
    path = get(ENV, "SMMHIP_LIB", joinpath(@__DIR__, "..", "smm.jl_amd", "csrc", "libsmmhip.so"))
    LIB[] = Libdl.dlopen(path)          # throws if the library is missing: there is no CPU fallback
    v = ccall(Libdl.dlsym(LIB[], :smm_abi_version), Cint, ())
    v == 1 || error("libsmmhip ABI version $v, expected 1")
end

sym(s::Symbol) = Libdl.dlsym(LIB[], s)

# ---- mirror of include/smmhip.h -------------------------------------------------------------
struct SmmProblem
    np::Cint; nm::Cint; ns::Cint; objective_id::Cint
    init::Ptr{Cdouble}; lb::Ptr{Cdouble}; ub::Ptr{Cdouble}
    mom::Ptr{Cdouble}; w::Ptr{Cdouble}; obj_params::Ptr{Cdouble}
    n_obj_params::Cint; reserved::Cint
end

struct SmmBgpOpts
    N::Cint; maxiter::Cint
    sigma::Ptr{Cdouble}; acc_tuner::Ptr{Cdouble}; min_improve::Ptr{Cdouble}
    sigma_update_steps::Cint; smpl_iters::Cint
    sigma_adjust_by::Cdouble
    batch_size::Cint; exchange_from_iter::Cint
    seed::UInt64
    chain_offset::Cint; N_global::Cint; device::Cint; reserved::Cint
end

struct SmmHistory
    value::Ptr{Cdouble}; prob::Ptr{Cdouble}; curr_val::Ptr{Cdouble}; best_val::Ptr{Cdouble}
    params::Ptr{Cdouble}; sim_moments::Ptr{Cdouble}
    best_id::Ptr{Int32}; exchanged::Ptr{Int32}; accepted::Ptr{UInt8}; status::Ptr{Int8}
end

struct SmmState
    iter::Cint; reserved::Cint
    sigma::Ptr{Cdouble}; accept_rate::Ptr{Cdouble}
    la_value::Ptr{Cdouble}; la_prob::Ptr{Cdouble}; la_params::Ptr{Cdouble}; la_sim_moments::Ptr{Cdouble}
    la_status::Ptr{Int8}; n_noex::Ptr{Int32}; n_acc_noex::Ptr{Int32}
    best_val::Ptr{Cdouble}; best_id::Ptr{Int32}
end

const OBJ_NORM, OBJ_BANANA, OBJ_NORM_FAILBOX, OBJ_DENSE = Cint(0), Cint(1), Cint(2), Cint(3)

last_error(ctx) = unsafe_string(ccall(sym(:smm_last_error), Cstring, (Ptr{Cvoid},), ctx))
check(ctx, rc) = rc == 0 ? nothing : error("smmhip error $rc: $(last_error(ctx))")

# ---- the algorithm object ---------------------------------------------------------------------
"""
    MAlgoBGPHip(names, init, lb, ub, moment_names, mom, w, opts; objective = :norm)

GPU-resident counterpart of `MAlgoBGP(m, opts)` (AlgoBGP.jl:505-537).  `opts` is SMM.jl's Dict
(keys N, maxiter, maxtemp, sigma, sigma_update_steps, sigma_adjust_by, smpl_iters, min_improve,
acc_tuners, batch_size; `seed`, `device` are new).  With SMM.jl loaded, build the arguments from an
`MProb`: names = keys(m.params_to_sample), init = m.initial_value, bounds from params_to_sample,
moments/weights from m.moments.
"""
mutable struct MAlgoBGPHip
    ctx::Ptr{Cvoid}
    opts::Dict
    i::Int
    N::Int; np::Int; nm::Int
    pnames::Vector{Symbol}; mnames::Vector{Symbol}
end

function MAlgoBGPHip(pnames, init, lb, ub, mnames, mom, w, opts::Dict; objective = :norm, ns::Integer = 10000,
                     obj_params::Vector{Float64} = Float64[])
    N = Int(opts["N"])
    temps = N > 1 ? collect(range(1.0, stop = Float64(opts["maxtemp"]), length = N)) : [1.0]   # AlgoBGP.jl:508
    sigma = get(opts, "sigma", 0.05) .* temps                                                    # :518
    mi = Float64.(get(opts, "min_improve", fill(0.5, N)))                                        # :522
    acc = Float64.(get(opts, "acc_tuners", fill(2.0, N)))                                        # :523
    init = Float64.(init); lb = Float64.(lb); ub = Float64.(ub); mom = Float64.(mom); w = Float64.(w)
    oid = objective isa Integer ? Cint(objective) :      # handle of hip_register_objective
          objective == :norm ? OBJ_NORM : objective == :banana ? OBJ_BANANA : objective == :dense ? OBJ_DENSE :
          error("unknown device objective $objective")
    ctx = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve init lb ub mom w sigma mi acc obj_params begin
        p = SmmProblem(length(init), length(mom), ns, oid, pointer(init), pointer(lb), pointer(ub),
                       pointer(mom), pointer(w), isempty(obj_params) ? C_NULL : pointer(obj_params), length(obj_params), 0)
        o = SmmBgpOpts(N, Int(opts["maxiter"]), pointer(sigma), pointer(acc), pointer(mi),
                       get(opts, "sigma_update_steps", 10), get(opts, "smpl_iters", 1000),
                       Float64(get(opts, "sigma_adjust_by", 0.01)),
                       get(opts, "batch_size", length(init)), 2, UInt64(get(opts, "seed", 12)),
                       0, N, get(opts, "device", 0), 0)
        rc = ccall(sym(:smm_ctx_create), Cint, (Ref{SmmProblem}, Ref{SmmBgpOpts}, Ptr{Cvoid}, Ref{Ptr{Cvoid}}),
                   p, o, C_NULL, ctx)
        rc == 0 || error("smm_ctx_create failed ($rc): $(last_error(C_NULL))")
    end
    a = MAlgoBGPHip(ctx[], opts, 0, N, length(init), length(mom), Symbol.(collect(pnames)), Symbol.(collect(mnames)))
    finalizer(x -> ccall(sym(:smm_ctx_destroy), Cvoid, (Ptr{Cvoid},), x.ctx), a)
    return a
end

"one `computeNextIteration!` (AlgoBGP.jl:589-640) — or `n` of them in one enqueue"
function hip_step!(a::MAlgoBGPHip, n::Integer = 1)
    check(a.ctx, ccall(sym(:smm_bgp_step), Cint, (Ptr{Cvoid}, Cint), a.ctx, n))
    a.i += n
    return a
end

"`run!(algo)` (AlgoAbstract.jl:27-76) without per-iteration hooks: all remaining iterations at once"
hip_run!(a::MAlgoBGPHip) = hip_step!(a, Int(a.opts["maxiter"]) - a.i)

"""
    hip_history(a) -> NamedTuple of arrays, iteration-major like the ABI ([t][chain], params [t][k][chain]);
in Julia's column-major terms: value[chain, t], params[chain, k, t].  `exchanged`, `best_id` and chain
ids are 1-based exactly as in `BGPChain` (AlgoBGP.jl:42-110).
"""
function hip_history(a::MAlgoBGPHip)
    N, T, np, nm = a.N, a.i, a.np, a.nm
    value = Matrix{Float64}(undef, N, T); prob = similar(value); curr = similar(value); best = similar(value)
    pars = Array{Float64}(undef, N, np, T); simm = Array{Float64}(undef, N, nm, T)
    bid = Matrix{Int32}(undef, N, T); exch = similar(bid); acc = Matrix{UInt8}(undef, N, T); st = Matrix{Int8}(undef, N, T)
    GC.@preserve value prob curr best pars simm bid exch acc st begin
        h = SmmHistory(pointer(value), pointer(prob), pointer(curr), pointer(best), pointer(pars), pointer(simm),
                       pointer(bid), pointer(exch), pointer(acc), pointer(st))
        check(a.ctx, ccall(sym(:smm_get_history), Cint, (Ptr{Cvoid}, Cint, Cint, Ref{SmmHistory}), a.ctx, 0, T, h))
    end
    return (value = value, prob = prob, curr_val = curr, best_val = best, params = pars, sim_moments = simm,
            best_id = bid, exchanged = exch, accepted = acc .!= 0, status = st)
end

"per-chain state: what `save`/`readMalgo`/`restart!` need besides the history (AlgoAbstract.jl:83-102)"
function hip_state(a::MAlgoBGPHip)
    N, np, nm = a.N, a.np, a.nm
    sigma = Vector{Float64}(undef, N); rate = similar(sigma); lav = similar(sigma); lap = similar(sigma)
    lapar = Matrix{Float64}(undef, N, np); lasm = Matrix{Float64}(undef, N, nm)
    last = Vector{Int8}(undef, N); nno = Vector{Int32}(undef, N); nac = similar(nno)
    bv = similar(sigma); bi = Vector{Int32}(undef, N)
    it = Ref{Cint}(0)
    GC.@preserve sigma rate lav lap lapar lasm last nno nac bv bi begin
        s = Ref(SmmState(0, 0, pointer(sigma), pointer(rate), pointer(lav), pointer(lap), pointer(lapar), pointer(lasm),
                         pointer(last), pointer(nno), pointer(nac), pointer(bv), pointer(bi)))
        check(a.ctx, ccall(sym(:smm_get_state), Cint, (Ptr{Cvoid}, Ref{SmmState}), a.ctx, s))
        it[] = s[].iter
    end
    return (iter = Int(it[]), sigma = sigma, accept_rate = rate, la_value = lav, la_prob = lap, la_params = lapar,
            la_sim_moments = lasm, la_status = last, n_noex = nno, n_acc_noex = nac, best_val = bv, best_id = bi)
end

"""
    hip_eval_batch(a, params) -> (value, sim_moments, status)

Batched `evaluateObjective(m, p)` (mprob.jl:175-188) for the columns... `params` is M x np (one row per
point; the ABI wants [np][M], which is this matrix in column-major order).  Serves `doSlices`
(slices.jl:263) and `FD_gradient` (econometrics.jl:42).
"""
function hip_eval_batch(a::MAlgoBGPHip, params::Matrix{Float64})
    M = size(params, 1)
    size(params, 2) == a.np || error("params must be M x np")
    value = Vector{Float64}(undef, M); simm = Matrix{Float64}(undef, M, a.nm); st = Vector{Int8}(undef, M)
    GC.@preserve params value simm st begin
        check(a.ctx, ccall(sym(:smm_eval_batch), Cint,
                           (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Int8}),
                           a.ctx, pointer(params), M, pointer(value), pointer(simm), pointer(st)))
    end
    return value, simm, st
end

"""
    hip_register_objective(src) -> objective id

Compile a user objective (HIP/C++ text defining `SMM_USER_OBJECTIVE(theta, np, mom, w, nm, udata, n_udata,
sim_moments, value, status)`, see include/smmhip.h) for the device: the counterpart of `addEvalFunc!(m, f)`
(mprob.jl:159).  Pass the returned id as `objective = id` to `MAlgoBGPHip`.
"""
function hip_register_objective(src::AbstractString)
    id = Ref{Int32}(0)
    rc = ccall(sym(:smm_register_user_objective), Cint, (Cstring, Ref{Int32}), src, id)
    rc == 0 || error("smm_register_user_objective failed ($rc): $(last_error(C_NULL))")
    return Int(id[])
end

# ---- glue for SMM.jl (only evaluated when SMM is loaded next to this module) ----------------
#   import SMM: computeNextIteration!, MAlgo
#   struct HipAlgo <: MAlgo; inner::MAlgoBGPHip; m::MProb; opts::Dict; i::Int; end
#   computeNextIteration!(algo::HipAlgo) = (hip_step!(algo.inner, 1); nothing)       # AlgoAbstract.jl:45
# and `serialNormal(2, 200)` becomes
#   a = MAlgoBGPHip([:p1, :p2], [0.2, -0.2], [-3, -20], [3, 20], [:mu1, :mu2], [-1.0, 10.0], [1.0, 1.0],
#                   Dict("N" => 3, "maxiter" => 200, "maxtemp" => 5, "smpl_iters" => 1000,
#                        "min_improve" => zeros(3), "acc_tuners" => [20.0, 2.0, 1.0]))
#   hip_run!(a); h = hip_history(a)

end # module
