# SMMHip.jl — raw Julia binding of libsmmhip.so (the C ABI of include/smmhip.h, ABI version 2).
#
# Stdlib only (Libdl): loads without a package registry and without SMM.jl.  The drop-in layer for SMM.jl itself —
# `MAlgoBGPHip <: SMM.MAlgo`, `computeNextIteration!`, real `SMM.BGPChain` objects filled from the device, `save`,
# `restart!` — is julia/SMMHipBackend.jl, built on the functions of this file.
#
# NOT EXECUTED IN THIS REPOSITORY'S CI: the build image has no `julia` binary.  What CAN be checked without one is:
# tests/test_julia_layer.py parses the `struct` blocks below and compares field order, types, offsets and sizes
# with the C header (through a compiled probe), checks that every `ccall` names an exported symbol with the header's
# argument count, and that the glue defines the methods the reference dispatches on.  The same ABI is exercised
# end to end through the ctypes binding (smm.jl_amd/_abi.py, tests/).
module SMMHip

using Libdl

export HipBGP, hip_create, hip_destroy!, hip_step!, hip_iter, hip_history, hip_state, hip_set_state!, hip_eval_batch,
       hip_register_objective, hip_record_doubles
export hip_eval_batch_noseed, hip_stream, hip_sync, hip_local_step!, hip_export_records!, hip_exchange!, hip_sharded_step!, hip_sharded_finish!,
       hip_a2a_capacity, hip_export_values!, hip_a2a_pack!, hip_a2a_apply!, hip_record_doubles
export hip_step_async!, hip_p2p_init, hip_p2p_attach!, hip_p2p_step!, hip_p2p_finish!, hip_set_persistent!, hip_persistent_info, P2P_HANDLE_BYTES

const ABI_VERSION = 3
const LIB = Ref{Ptr{Cvoid}}(C_NULL)

"path of the library: ENV[\"SMMHIP_LIBRARY\"] or the in-tree build"
libpath() = get(ENV, "SMMHIP_LIBRARY", joinpath(@__DIR__, "..", "smm.jl_amd", "csrc", "libsmmhip.so"))

function __init__()
    LIB[] = Libdl.dlopen(libpath())          # throws if the library is missing: there is no CPU fallback
    v = ccall(Libdl.dlsym(LIB[], :smm_abi_version), Cint, ())
    v == ABI_VERSION || error("libsmmhip ABI version $v, this binding is written for $ABI_VERSION")
end

sym(s::Symbol) = Libdl.dlsym(LIB[], s)

# ---- mirror of include/smmhip.h (field for field; tests/test_julia_layer.py checks offsets and sizes) -----------
struct SmmProblem
    np::Cint
    nm::Cint
    ns::Cint
    objective_id::Cint
    init::Ptr{Cdouble}
    lb::Ptr{Cdouble}
    ub::Ptr{Cdouble}
    mom::Ptr{Cdouble}
    w::Ptr{Cdouble}
    obj_params::Ptr{Cdouble}
    n_obj_params::Cint
    reserved::Cint
end

struct SmmBgpOpts
    N::Cint
    maxiter::Cint
    sigma::Ptr{Cdouble}
    acc_tuner::Ptr{Cdouble}
    min_improve::Ptr{Cdouble}
    sigma_update_steps::Cint
    smpl_iters::Cint
    sigma_adjust_by::Cdouble
    batch_size::Cint
    exchange_from_iter::Cint
    seed::UInt64
    chain_offset::Cint
    N_global::Cint
    device::Cint
    chol_per_chain::Cint
    chol_L::Ptr{Cdouble}
    dist_fun::Cint
    reserved::Cint
end

struct SmmTables
    probs_acc::Ptr{Cdouble}
    prop_normals::Ptr{Cdouble}
    prop_tries::Cint
    n_pairs::Cint
    pairs::Ptr{Int32}
    Z::Ptr{Cdouble}
end

struct SmmHistory
    value::Ptr{Cdouble}
    prob::Ptr{Cdouble}
    curr_val::Ptr{Cdouble}
    best_val::Ptr{Cdouble}
    params::Ptr{Cdouble}
    sim_moments::Ptr{Cdouble}
    best_id::Ptr{Int32}
    exchanged::Ptr{Int32}
    accepted::Ptr{UInt8}
    status::Ptr{Int8}
end

struct SmmState
    iter::Cint
    reserved::Cint
    sigma::Ptr{Cdouble}
    accept_rate::Ptr{Cdouble}
    la_value::Ptr{Cdouble}
    la_prob::Ptr{Cdouble}
    la_params::Ptr{Cdouble}
    la_sim_moments::Ptr{Cdouble}
    la_status::Ptr{Int8}
    n_noex::Ptr{Int32}
    n_acc_noex::Ptr{Int32}
    best_val::Ptr{Cdouble}
    best_id::Ptr{Int32}
end

struct SmmTiming
    step_ms::Cdouble
    iter_kernel_ms::Cdouble
    exch_kernel_ms::Cdouble
    chain_evals::Int64
    iters::Cint
    reserved::Cint
    null_bracket_ms::Cdouble
end

# smm_objective_t
const OBJ_NORM = Cint(0)
const OBJ_BANANA = Cint(1)
const OBJ_NORM_FAILBOX = Cint(2)
const OBJ_DENSE = Cint(3)
const OBJ_USER_BASE = Cint(1000)
# smm_dist_fun_t (opts["dist_fun"], AlgoBGP.jl:537)
const DIST_MINUS = Cint(0)
const DIST_ABSDIFF = Cint(1)
const DIST_RELDIFF = Cint(2)

struct SMMHipError <: Exception
    code::Int
    msg::String
end
Base.showerror(io::IO, e::SMMHipError) = print(io, "smmhip error ", e.code, ": ", e.msg)

last_error(ctx::Ptr{Cvoid}) = unsafe_string(ccall(sym(:smm_last_error), Cstring, (Ptr{Cvoid},), ctx))
check(ctx::Ptr{Cvoid}, rc::Integer) = rc == 0 ? nothing : throw(SMMHipError(Int(rc), last_error(ctx)))

# ---- one device context = the chains of one MAlgoBGP (shard) ----------------------------------------------------
"""
    HipBGP

Handle of one device context (`smm_ctx_create`): the chains of one `MAlgoBGP` on one GPU.  `N`, `np`, `nm`, `maxiter`
are kept for buffer sizes.  Destroyed by `hip_destroy!` or the finalizer (whichever comes first; never twice).
"""
mutable struct HipBGP
    ctx::Ptr{Cvoid}
    N::Int
    np::Int
    nm::Int
    maxiter::Int
end

function hip_destroy!(h::HipBGP)
    if h.ctx != C_NULL
        ccall(sym(:smm_ctx_destroy), Cvoid, (Ptr{Cvoid},), h.ctx)
        h.ctx = C_NULL                      # the finalizer (or a second call) finds nothing to free
    end
    return nothing
end

"""
    hip_create(init, lb, ub, mom, w, sigma, acc_tuner, min_improve; maxiter, ns = 10000, objective_id = OBJ_NORM, ...)

`MAlgoBGP(m, opts)` + the `BGPChain` constructors (AlgoBGP.jl:505-537, :78-109) as one device context.  The per-chain
vectors `sigma`, `acc_tuner`, `min_improve` are GLOBAL (length `N_global`, default `N = length(sigma)`): the reference's
default 3-entry lists must be expanded by the caller (the glue does); shorter vectors are an error here, not a read
past the end of a Julia array inside the library.
"""
function hip_create(init::Vector{Float64}, lb::Vector{Float64}, ub::Vector{Float64}, mom::Vector{Float64}, w::Vector{Float64},
                    sigma::Vector{Float64}, acc_tuner::Vector{Float64}, min_improve::Vector{Float64};
                    maxiter::Integer, ns::Integer = 10000, objective_id::Integer = OBJ_NORM,
                    obj_params::Vector{Float64} = Float64[], N::Integer = length(sigma), N_global::Integer = length(sigma),
                    chain_offset::Integer = 0, sigma_update_steps::Integer = 10, sigma_adjust_by::Real = 0.01,
                    smpl_iters::Integer = 1000, batch_size::Integer = length(init), seed::Integer = 12, device::Integer = 0,
                    chol_L::Union{Nothing,Array{Float64}} = nothing, dist_fun::Integer = DIST_MINUS)
    np, nm = length(init), length(mom)
    length(lb) == np && length(ub) == np || throw(ArgumentError("lb / ub need one entry per parameter"))
    length(w) == nm || throw(ArgumentError("w needs one entry per moment"))
    (length(sigma) >= N_global && length(acc_tuner) >= N_global && length(min_improve) >= N_global) ||
        throw(ArgumentError("sigma / acc_tuner / min_improve need N_global = $N_global entries (AlgoBGP.jl:518-523)"))
    per_chain = 0
    Lrow = Float64[]
    if chol_L !== nothing
        # row-major [np][np] (shared) or [N_global][np][np]: Julia arrays are column-major, so L[k, j] of a Matrix is
        # transposed into the row-major order the header asks for
        if ndims(chol_L) == 2
            size(chol_L) == (np, np) || throw(ArgumentError("chol_L must be np x np"))
            Lrow = vec(permutedims(chol_L, (2, 1)))
        else
            size(chol_L) == (np, np, N_global) || throw(ArgumentError("per-chain chol_L must be np x np x N_global (L[:, :, c])"))
            Lrow = vec(permutedims(chol_L, (2, 1, 3)))
            per_chain = 1
        end
    end
    ctx = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve init lb ub mom w sigma acc_tuner min_improve obj_params Lrow begin
        p = SmmProblem(np, nm, ns, objective_id, pointer(init), pointer(lb), pointer(ub), pointer(mom), pointer(w),
                       isempty(obj_params) ? Ptr{Cdouble}(C_NULL) : pointer(obj_params), length(obj_params), 0)
        o = SmmBgpOpts(N, maxiter, pointer(sigma), pointer(acc_tuner), pointer(min_improve), sigma_update_steps, smpl_iters,
                       Float64(sigma_adjust_by), batch_size, 2, UInt64(seed), chain_offset, N_global, device, per_chain,
                       isempty(Lrow) ? Ptr{Cdouble}(C_NULL) : pointer(Lrow), dist_fun, 0)
        rc = ccall(sym(:smm_ctx_create), Cint, (Ref{SmmProblem}, Ref{SmmBgpOpts}, Ptr{SmmTables}, Ref{Ptr{Cvoid}}),
                   p, o, C_NULL, ctx)
        rc == 0 || throw(SMMHipError(Int(rc), last_error(Ptr{Cvoid}(C_NULL))))
    end
    h = HipBGP(ctx[], Int(N), np, nm, Int(maxiter))
    finalizer(hip_destroy!, h)
    return h
end

"`n` x `computeNextIteration!` (AlgoBGP.jl:589-640, incl. `exchangeMoves!` :647-716) in one enqueue; blocks"
function hip_step!(h::HipBGP, n::Integer = 1)
    check(h.ctx, ccall(sym(:smm_bgp_step), Cint, (Ptr{Cvoid}, Cint), h.ctx, n))
    return h
end

"""
    hip_step_async!(h, n)

The same, enqueued only: returns at once, `hip_sync(h)` waits (and reports a hard error of the reference, AlgoBGP.jl:341,409).
Steps of n >= 2 iterations take the persistent form where the context qualifies (include/smmhip.h, smm_set_persistent).
"""
function hip_step_async!(h::HipBGP, n::Integer = 1)
    check(h.ctx, ccall(sym(:smm_bgp_step_async), Cint, (Ptr{Cvoid}, Cint), h.ctx, n))
    return h
end

"the persistent form of `hip_step!` (include/smmhip.h): on by default where the context qualifies"
hip_set_persistent!(h::HipBGP, on::Bool) = (check(h.ctx, ccall(sym(:smm_set_persistent), Cint, (Ptr{Cvoid}, Cint), h.ctx, on ? 1 : 0)); h)
"(would the next step take the persistent form, launches of it so far, repairs so far)"
function hip_persistent_info(h::HipBGP)
    a = Ref{Int32}(0); l = Ref{Int32}(0); r = Ref{Int32}(0)
    check(h.ctx, ccall(sym(:smm_get_persistent), Cint, (Ptr{Cvoid}, Ref{Int32}, Ref{Int32}, Ref{Int32}), h.ctx, a, l, r))
    return (a[] != 0, Int(l[]), Int(r[]))
end

"completed iterations (after a hard error: the failing iteration, see include/smmhip.h)"
hip_iter(h::HipBGP) = hip_state(h).iter

hip_record_doubles(h::HipBGP) = Int(ccall(sym(:smm_bgp_record_doubles), Cint, (Ptr{Cvoid},), h.ctx))

# The sharded forms (one HipBGP per GPU and process; device pointers of the caller's communication library, e.g. the
# buffers of an MPI.jl / RCCL wrapper; everything is enqueued on hip_stream(h)).  include/smmhip.h describes the protocols.
hip_stream(h::HipBGP) = ccall(sym(:smm_stream), Ptr{Cvoid}, (Ptr{Cvoid},), h.ctx)
hip_sync(h::HipBGP) = (check(h.ctx, ccall(sym(:smm_sync), Cint, (Ptr{Cvoid},), h.ctx)); h)
hip_local_step!(h::HipBGP) = (check(h.ctx, ccall(sym(:smm_bgp_local_step), Cint, (Ptr{Cvoid},), h.ctx)); h)
hip_export_records!(h::HipBGP, rec::Ptr{Cvoid}) = (check(h.ctx, ccall(sym(:smm_bgp_export_records_dev), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), h.ctx, rec)); h)
hip_exchange!(h::HipBGP, gathered::Ptr{Cvoid}) = (check(h.ctx, ccall(sym(:smm_bgp_exchange_dev), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), h.ctx, gathered)); h)
hip_sharded_step!(h::HipBGP, prev::Ptr{Cvoid}, next::Ptr{Cvoid}) =
    (check(h.ctx, ccall(sym(:smm_bgp_sharded_step), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), h.ctx, prev, next)); h)
hip_sharded_finish!(h::HipBGP, gathered::Ptr{Cvoid}) =
    (check(h.ctx, ccall(sym(:smm_bgp_sharded_finish), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), h.ctx, gathered)); h)
# the values form for long records: all-gather of N doubles per rank, then one all-to-all of hip_a2a_capacity(h) * RW doubles per pair
hip_a2a_capacity(h::HipBGP) = Int(ccall(sym(:smm_bgp_a2a_capacity), Cint, (Ptr{Cvoid},), h.ctx))
hip_export_values!(h::HipBGP, vals::Ptr{Cvoid}) = (check(h.ctx, ccall(sym(:smm_bgp_export_values_dev), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), h.ctx, vals)); h)
hip_a2a_pack!(h::HipBGP, vals_all::Ptr{Cvoid}, send::Ptr{Cvoid}) =
    (check(h.ctx, ccall(sym(:smm_bgp_a2a_pack_dev), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), h.ctx, vals_all, send)); h)
hip_a2a_apply!(h::HipBGP, recv::Ptr{Cvoid}) = (check(h.ctx, ccall(sym(:smm_bgp_a2a_apply_dev), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), h.ctx, recv)); h)

# The p2p form (no collective at all: every rank owns a window that the others map through HIP IPC; include/smmhip.h).  The handles
# are 64 plain bytes: any transport hands them round once — julia/SMMHipSharded.jl does it with Distributed alone.
const P2P_HANDLE_BYTES = 64
"this rank's window: returns its IPC handle (for the other PROCESSES) as a Vector{UInt8}"
function hip_p2p_init(h::HipBGP)
    handle = zeros(UInt8, P2P_HANDLE_BYTES)
    win = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve handle check(h.ctx, ccall(sym(:smm_bgp_p2p_init), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Ref{Ptr{Cvoid}}), h.ctx, pointer(handle), win))
    return handle
end
"rank `rank`'s window by its IPC handle (ranks are 0-based: chain_offset / N)"
function hip_p2p_attach!(h::HipBGP, rank::Integer, handle::Vector{UInt8})
    length(handle) == P2P_HANDLE_BYTES || throw(ArgumentError("an IPC handle is $P2P_HANDLE_BYTES bytes"))
    GC.@preserve handle check(h.ctx, ccall(sym(:smm_bgp_p2p_attach), Cint, (Ptr{Cvoid}, Cint, Ptr{UInt8}, Ptr{Cvoid}), h.ctx, rank, pointer(handle), C_NULL))
    return h
end
"n iterations of this shard, enqueued (every rank calls it with the same n); `hip_sync` waits"
hip_p2p_step!(h::HipBGP, n::Integer) = (check(h.ctx, ccall(sym(:smm_bgp_p2p_step), Cint, (Ptr{Cvoid}, Cint), h.ctx, n)); h)
"settle the last iteration into the context (before history / state are read); every rank calls it"
hip_p2p_finish!(h::HipBGP) = (check(h.ctx, ccall(sym(:smm_bgp_p2p_finish), Cint, (Ptr{Cvoid},), h.ctx)); h)

"""
    hip_history(h, t0, t1) -> NamedTuple

Iterations `t0+1 .. t1` (0-based half-open `[t0, t1)` as in the ABI).  The ABI's buffers are iteration-major
(`[t][chain]`, params `[t][k][chain]`); in Julia's column-major terms `value[chain, t]`, `params[chain, k, t]`.
`exchanged`, `best_id` are 1-based exactly as in `BGPChain` (AlgoBGP.jl:42-110): 0 = no exchange, -1 = unset.
"""
function hip_history(h::HipBGP, t0::Integer, t1::Integer)
    N, T, np, nm = h.N, t1 - t0, h.np, h.nm
    value = Matrix{Float64}(undef, N, T); prob = similar(value); curr = similar(value); best = similar(value)
    pars = Array{Float64}(undef, N, np, T); simm = Array{Float64}(undef, N, nm, T)
    bid = Matrix{Int32}(undef, N, T); exch = similar(bid); acc = Matrix{UInt8}(undef, N, T); st = Matrix{Int8}(undef, N, T)
    GC.@preserve value prob curr best pars simm bid exch acc st begin
        hs = SmmHistory(pointer(value), pointer(prob), pointer(curr), pointer(best), pointer(pars), pointer(simm),
                        pointer(bid), pointer(exch), pointer(acc), pointer(st))
        check(h.ctx, ccall(sym(:smm_get_history), Cint, (Ptr{Cvoid}, Cint, Cint, Ref{SmmHistory}), h.ctx, t0, t1, hs))
    end
    return (value = value, prob = prob, curr_val = curr, best_val = best, params = pars, sim_moments = simm,
            best_id = bid, exchanged = exch, accepted = acc, status = st)
end

"per-chain state: what `save` / `readMalgo` / `restart!` need besides the history (AlgoAbstract.jl:83-102)"
function hip_state(h::HipBGP)
    N, np, nm = h.N, h.np, h.nm
    sigma = Vector{Float64}(undef, N); rate = similar(sigma); lav = similar(sigma); lap = similar(sigma)
    lapar = Matrix{Float64}(undef, N, np); lasm = Matrix{Float64}(undef, N, nm)
    last = Vector{Int8}(undef, N); nno = Vector{Int32}(undef, N); nac = similar(nno)
    bv = similar(sigma); bi = Vector{Int32}(undef, N)
    it = 0
    GC.@preserve sigma rate lav lap lapar lasm last nno nac bv bi begin
        s = Ref(SmmState(0, 0, pointer(sigma), pointer(rate), pointer(lav), pointer(lap), pointer(lapar), pointer(lasm),
                         pointer(last), pointer(nno), pointer(nac), pointer(bv), pointer(bi)))
        check(h.ctx, ccall(sym(:smm_get_state), Cint, (Ptr{Cvoid}, Ref{SmmState}), h.ctx, s))
        it = Int(s[].iter)
    end
    return (iter = it, sigma = sigma, accept_rate = rate, la_value = lav, la_prob = lap, la_params = lapar,
            la_sim_moments = lasm, la_status = last, n_noex = nno, n_acc_noex = nac, best_val = bv, best_id = bi)
end

"""
    hip_set_state!(h, state, history)

`restart!` (AlgoBGP.jl:804-884) with clean resume semantics: upload a state (as returned by `hip_state`, `iter` completed
iterations) and the history of iterations `1 .. iter` (as returned by `hip_history(h, 0, iter)`); stepping continues at
`iter + 1`.  Also clears a sticky hard error.
"""
function hip_set_state!(h::HipBGP, s::NamedTuple, hist::NamedTuple)
    GC.@preserve s hist begin
        st = SmmState(s.iter, 0, pointer(s.sigma), pointer(s.accept_rate), pointer(s.la_value), pointer(s.la_prob),
                      pointer(s.la_params), pointer(s.la_sim_moments), pointer(s.la_status), pointer(s.n_noex),
                      pointer(s.n_acc_noex), pointer(s.best_val), pointer(s.best_id))
        hs = SmmHistory(pointer(hist.value), pointer(hist.prob), pointer(hist.curr_val), pointer(hist.best_val),
                        pointer(hist.params), pointer(hist.sim_moments), pointer(hist.best_id), pointer(hist.exchanged),
                        pointer(hist.accepted), pointer(hist.status))
        check(h.ctx, ccall(sym(:smm_set_state), Cint, (Ptr{Cvoid}, Ref{SmmState}, Ref{SmmHistory}), h.ctx, st, hs))
    end
    return h
end

"""
    hip_eval_batch(h, params) -> (value, sim_moments, status)

Batched `evaluateObjective(m, p)` (mprob.jl:175-188): `params` is M x np, one row per point (the ABI wants `[np][M]`,
which is this matrix in column-major order).  Serves `doSlices` (slices.jl:153) and `FD_gradient` (econometrics.jl:42).
"""
function hip_eval_batch(h::HipBGP, params::Matrix{Float64})
    M = size(params, 1)
    size(params, 2) == h.np || throw(ArgumentError("params must be M x np"))
    value = Vector{Float64}(undef, M); simm = Matrix{Float64}(undef, M, h.nm); st = Vector{Int8}(undef, M)
    GC.@preserve params value simm st begin
        check(h.ctx, ccall(sym(:smm_eval_batch), Cint,
                           (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Int8}),
                           h.ctx, pointer(params), M, pointer(value), pointer(simm), pointer(st)))
    end
    return value, simm, st
end

"""
    hip_eval_batch_noseed(h, params, base_seed) -> (value, sim_moments, status)

The same with `options[:noseed] = true` (ObjExamples.jl:71-75): evaluation i draws its own shocks, keyed by `base_seed + i` —
the repetitions of `getSigma` (econometrics.jl:125-145).
"""
function hip_eval_batch_noseed(h::HipBGP, params::Matrix{Float64}, base_seed::Integer)
    M = size(params, 1)
    size(params, 2) == h.np || throw(ArgumentError("params must be M x np"))
    value = Vector{Float64}(undef, M); simm = Matrix{Float64}(undef, M, h.nm); st = Vector{Int8}(undef, M)
    GC.@preserve params value simm st begin
        check(h.ctx, ccall(sym(:smm_eval_batch_noseed), Cint,
                           (Ptr{Cvoid}, Ptr{Cdouble}, Cint, UInt64, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Int8}),
                           h.ctx, pointer(params), M, UInt64(base_seed), pointer(value), pointer(simm), pointer(st)))
    end
    return value, simm, st
end

"""
    hip_register_objective(src) -> objective id

Compile a user objective (HIP/C++ text defining `SMM_USER_OBJECTIVE(theta, np, mom, w, nm, udata, n_udata, sim_moments,
value, status)`, see include/smmhip.h) for the device: the counterpart of `addEvalFunc!(m, f)` (mprob.jl:159).
"""
function hip_register_objective(src::AbstractString)
    id = Ref{Int32}(0)
    rc = ccall(sym(:smm_register_user_objective), Cint, (Cstring, Ref{Int32}), src, id)
    rc == 0 || throw(SMMHipError(Int(rc), last_error(Ptr{Cvoid}(C_NULL))))
    return Int(id[])
end

end # module
