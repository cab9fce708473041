# SMMHipBackend.jl — the drop-in layer for SMM.jl: the BGP path of floswald/SMM.jl on libsmmhip.so.
#
#     using SMM
#     include("julia/SMMHip.jl"); include("julia/SMMHipBackend.jl"); using .SMMHipBackend
#     MA = MAlgoBGPHip(mprob, opts)        # instead of MAlgoBGP(mprob, opts)          (AlgoBGP.jl:497-539)
#     run!(MA)                              # SMM.run! unchanged: it calls computeNextIteration!(MA) (AlgoAbstract.jl:27-76)
#     summary(MA); history(MA.chains[1]); SMM.params(MA.chains[1]); CI(MA.chains[1])   # the reference's own readers
#     save(MA, "run.jld2"); readMalgo("run.jld2")  # a plain SMM.MAlgoBGP on disk: readable without this layer
#     restart!(MA, 100)                     # more iterations, continuing the same chains
#
# How it stays a drop-in: `MAlgoBGPHip <: SMM.MAlgo` has the fields the reference's generic code touches (`m`, `opts`, `i`,
# `chains`, `anim`, `dist_fun`), and `chains` holds REAL `SMM.BGPChain` objects (built by the reference's own constructor),
# filled from the device history the first time somebody looks at them after a step (`getproperty(algo, :chains)` syncs).
# So `history(c)`, `summary(c)`, `params`, `allAccepted`, `best`, `mean`/`median`/`CI`, the plotting recipes — everything
# the reference defines on `BGPChain` (AlgoBGP.jl:117-206) — runs unmodified on what the GPU computed.
#
# Only the per-iteration work changes: `computeNextIteration!(algo::MAlgoBGPHip)` is one call into the library (proposal,
# objective, accept/reject, set_eval! for all chains and exchangeMoves!, AlgoBGP.jl:589-640) instead of pmap/map over chains.
#
# The objective must be a DEVICE objective: `SMM.objfunc_norm` maps to the built-in one; anything else is given as
# `opts["hip_objective"]` = an id of SMMHip.hip_register_objective(source) (or :banana / :dense), with `opts["hip_ns"]`,
# `opts["hip_obj_params"]` as needed.  A Julia closure cannot run inside the GPU iteration.
#
# NOT EXECUTED IN THIS REPOSITORY'S CI (no julia binary in the image); see the header of SMMHip.jl for what is checked.
module SMMHipBackend

using SMM
using OrderedCollections: OrderedDict          # (what SMM.jl itself depends on: Project.toml, src/SMM.jl:10)
import SMM: MAlgo, MAlgoBGP, MProb, Eval, BGPChain, Slice, computeNextIteration!, run!, summary, history, save, readMalgo, restart!,
            extendBGPChain!
import Base: getproperty, show
using ..SMMHip

export MAlgoBGPHip, sync_chains!, flush_steps!, hip_context, serialNormalHip, evaluateObjectivesHip, doSlicesHip, FD_gradient_hip, getSigmaHip

"""
    MAlgoBGPHip(m::MProb, opts::Dict)

GPU-resident counterpart of `MAlgoBGP(m, opts)` (AlgoBGP.jl:505-537): same `opts` keys (`N`, `maxiter`, `maxtemp`, `sigma`,
`sigma_update_steps`, `sigma_adjust_by`, `smpl_iters`, `min_improve`, `acc_tuners`, `batch_size`, `save_frequency`,
`filename`), plus `seed`, `device`, `chol_L` (general Gaussian proposals) and the `hip_*` keys described above.
"""
mutable struct MAlgoBGPHip <: MAlgo
    m::MProb                      # AlgoBGP.jl:498
    opts::Dict                    # :499
    i::Int                        # :500  iteration (set by run!, AlgoAbstract.jl:42)
    chains::Vector{BGPChain}      # :501  real SMM.BGPChain objects, refreshed from the device on access
    anim::Any                     # :502  (plots of the proposal cloud are not produced by the device path)
    dist_fun::Function            # :503  `-` (:537) or the function of the menu entry chosen (device_dist_fun)
    hip::SMMHip.HipBGP            # the device context
    synced::Int                   # iterations already materialised in `chains`
    stepped::Int                  # iterations enqueued on the device so far (host-side count: asking it needs no device synchronisation)
    deferred::Int                 # single steps of computeNextIteration! counted but not enqueued yet (flush_steps!)
    pnames::Vector{Symbol}        # parameter order on the device = keys(m.params_to_sample)
    mnames::Vector{Symbol}        # moment order on the device = keys(m.moments)
end

# per-chain option vectors as the reference reads them (AlgoBGP.jl:522-523), expanded to exactly N entries
function chain_vector(opts::Dict, key::String, default::Float64, N::Int)
    v = Float64.(get(opts, key, fill(default, N)))
    length(v) >= N || throw(ArgumentError("opts[\"$key\"] needs one entry per chain (N = $N), AlgoBGP.jl:522-523"))
    return v[1:N]
end

function device_objective(m::MProb, opts::Dict)
    if haskey(opts, "hip_objective")
        o = opts["hip_objective"]
        o isa Integer && return Cint(o)
        o == :norm && return SMMHip.OBJ_NORM
        o == :banana && return SMMHip.OBJ_BANANA
        o == :dense && return SMMHip.OBJ_DENSE
        throw(ArgumentError("unknown opts[\"hip_objective\"] = $o"))
    end
    m.objfunc === SMM.objfunc_norm && return SMMHip.OBJ_NORM
    throw(ArgumentError("the objective of this MProb is a Julia function; the GPU iteration needs a device objective: " *
                        "SMM.objfunc_norm, or opts[\"hip_objective\"] = SMMHip.hip_register_objective(source)"))
end

function reference_chains(m::MProb, opts::Dict, N::Int, temps::Vector{Float64}, mi::Vector{Float64}, acc::Vector{Float64})
    # exactly the comprehension of the reference's constructor (AlgoBGP.jl:512-536): every field of BGPChain exists and has
    # the reference's initial value (best_val = Inf, best_id = -1, accepted = false, exchanged = 0, ...)
    return BGPChain[BGPChain(i, opts["maxiter"];
                             m = m,
                             sig = get(opts, "sigma", 0.05) * temps[i],
                             upd = get(opts, "sigma_update_steps", 10),
                             upd_by = get(opts, "sigma_adjust_by", 0.01),
                             smpl_iters = get(opts, "smpl_iters", 1000),
                             min_improve = mi[i],
                             acc_tuner = acc[i],
                             batch_size = get(opts, "batch_size", length(m.params_to_sample))) for i in 1:N]
end

"""
    hip_context(m, opts; N_local, chain_offset, device) -> (hip, N, temps, mi, acc, dist_fun, dist_id, pnames, mnames)

The device context of `MAlgoBGP(m, opts)` (AlgoBGP.jl:505-537): ladder, per-chain vectors, flat problem.  `N_local` / `chain_offset`
make it ONE SHARD of the population `opts["N"]` (julia/SMMHipSharded.jl: one process per GPU); the per-chain vectors stay global.
"""
function hip_context(m::MProb, opts::Dict; N_local::Int = Int(opts["N"]), chain_offset::Int = 0, device::Int = Int(get(opts, "device", 0)))
    N = Int(opts["N"])
    temps = N > 1 ? collect(range(1.0, stop = Float64(opts["maxtemp"]), length = N)) : [1.0]      # AlgoBGP.jl:508
    mi = chain_vector(opts, "min_improve", 0.5, N)                                                  # :522
    acc = chain_vector(opts, "acc_tuners", 2.0, N)                                                  # :523
    dist_fun = get(opts, "hip_dist_fun", get(opts, "dist_fun", -))                                  # :537 (hip_dist_fun: the menu entry of a saved run, as_reference)
    dist_id = device_dist_fun(dist_fun)
    pnames = Symbol[Symbol(k) for k in keys(m.params_to_sample)]
    mnames = Symbol[Symbol(k) for k in keys(m.moments)]
    init = Float64[m.initial_value[k] for k in keys(m.params_to_sample)]
    lb = Float64[v[:lb] for (k, v) in m.params_to_sample]
    ub = Float64[v[:ub] for (k, v) in m.params_to_sample]
    mom = Float64[v[:value] for (k, v) in m.moments]
    w = Float64[v[:weight] for (k, v) in m.moments]
    sigma = get(opts, "sigma", 0.05) .* temps                                                        # :518
    hip = SMMHip.hip_create(init, lb, ub, mom, w, sigma, acc, mi;
                            maxiter = Int(opts["maxiter"]), ns = Int(get(opts, "hip_ns", 10000)),
                            objective_id = device_objective(m, opts),
                            obj_params = Float64.(get(opts, "hip_obj_params", Float64[])),
                            sigma_update_steps = Int(get(opts, "sigma_update_steps", 10)),
                            sigma_adjust_by = Float64(get(opts, "sigma_adjust_by", 0.01)),
                            smpl_iters = Int(get(opts, "smpl_iters", 1000)),
                            batch_size = Int(get(opts, "batch_size", length(init))),
                            seed = Int(get(opts, "seed", 12)), device = device,
                            N = N_local, N_global = N, chain_offset = chain_offset,
                            chol_L = get(opts, "chol_L", nothing), dist_fun = dist_id)
    return hip, N, temps, mi, acc, dist_fun, dist_id, pnames, mnames
end

function MAlgoBGPHip(m::MProb, opts::Dict)
    # opts["animate"] (AlgoBGP.jl:621-624) makes the reference push a plot frame of its chains into algo.anim every iteration:
    # that is a host-side per-iteration hook on objects this backend fills lazily; refused rather than silently ignored
    get(opts, "animate", false) == true &&
        throw(ArgumentError("opts[\"animate\"] is not supported by the GPU backend: run without it and plot the synced chains afterwards"))
    hip, N, temps, mi, acc, dist_fun, dist_id, pnames, mnames = hip_context(m, opts)
    return MAlgoBGPHip(m, opts, 0, reference_chains(m, opts, N, temps, mi, acc), nothing,
                       dist_fun isa Function ? dist_fun : host_dist_fun(dist_id), hip, 0, 0, 0, pnames, mnames)
end

# opts["dist_fun"] (AlgoBGP.jl:494,537) -> smm_dist_fun_t.  The reference takes any function of two objective values; inside the
# exchange kernels only the header's menu runs: `-` (the default), :absdiff (|a - b|), :reldiff ((a - b) / |a|).
function device_dist_fun(f)
    f === (-) && return SMMHip.DIST_MINUS
    f in (:minus, "-", "minus") && return SMMHip.DIST_MINUS
    f in (:absdiff, "absdiff") && return SMMHip.DIST_ABSDIFF
    f in (:reldiff, "reldiff") && return SMMHip.DIST_RELDIFF
    throw(ArgumentError("dist_fun: the device runs `-` (AlgoBGP.jl:537), :absdiff or :reldiff (smm_dist_fun_t), not an arbitrary function"))
end
host_dist_fun(id) = id == SMMHip.DIST_ABSDIFF ? ((a, b) -> abs(a - b)) : id == SMMHip.DIST_RELDIFF ? ((a, b) -> (a - b) / abs(a)) : (-)

# ---- the one method SMM.jl dispatches on (AlgoAbstract.jl:45; README.md:105-107) -------------------------------
"""
    computeNextIteration!(algo::MAlgoBGPHip)

One BGP iteration of all chains on the device: next_eval for every chain (AlgoBGP.jl:272-294) and exchangeMoves!
(:647-716).  `run!` has set `algo.i` (AlgoAbstract.jl:42); a hard error of the reference (negative objective :341, no draw
in support :409) is thrown as `SMMHip.SMMHipError`.
"""
function computeNextIteration!(algo::MAlgoBGPHip)
    # Nothing is enqueued per call, and nobody waits for the device: the reference's unchanged loop (`for i in 1:maxiter ...
    # computeNextIteration!(algo)`, AlgoAbstract.jl:38-45) would otherwise pay a library call and a device synchronisation per
    # iteration and — one iteration at a time — never take the persistent form (include/smmhip.h: steps of n >= 2).  The calls are
    # COUNTED and handed to the library as one asynchronous step when somebody looks (`algo.chains`, `save`, `summary`, ...) or a
    # look-ahead window's worth has come together.  A hard error of the reference (AlgoBGP.jl:341,409) therefore surfaces at that
    # point, naming its iteration, not inside the call of that iteration.
    d = getfield(algo, :deferred) + 1
    setfield!(algo, :deferred, d)
    d >= DEFER_MAX && flush_steps!(algo)
    return nothing
end

const DEFER_MAX = 256            # a look-ahead window of the library

"""
    flush_steps!(algo)

Enqueue the iterations `computeNextIteration!` has counted since the last flush as ONE asynchronous step (no device synchronisation).
"""
function flush_steps!(algo::MAlgoBGPHip)
    d = getfield(algo, :deferred)
    d == 0 && return nothing
    setfield!(algo, :deferred, 0)
    SMMHip.hip_step_async!(getfield(algo, :hip), d)       # (throws before anything is enqueued: a step past maxiter, a failed context)
    setfield!(algo, :stepped, getfield(algo, :stepped) + d)
    return nothing
end

"""
    run!(algo::MAlgoBGPHip)

`SMM.run!` works unchanged (one library call per iteration, with the progress meter).  This method does the same job
without per-iteration host work: all iterations between two save points are enqueued in ONE call (AlgoAbstract.jl:27-76).
"""
function run!(algo::MAlgoBGPHip)
    t0 = time()
    maxiter = Int(algo.opts["maxiter"])
    sf = get(algo.opts, "save_frequency", 0)
    fn = get(algo.opts, "filename", "")
    # (iterations somebody already asked for through computeNextIteration! — enqueued or still counted — are part of the run: `algo.i` alone would
    # step past maxiter when the caller had not advanced it itself, ADVICE r5)
    algo.i = max(algo.i, getfield(algo, :stepped) + getfield(algo, :deferred))
    while algo.i < maxiter
        n = maxiter - algo.i
        if sf > 0 && fn != ""
            n = min(n, sf - (algo.i % sf))
        end
        flush_steps!(algo)
        try
            SMMHip.hip_step!(getfield(algo, :hip), n)
            setfield!(algo, :stepped, getfield(algo, :stepped) + n)
        catch
            # a hard error: the device stands at the failing iteration (include/smmhip.h) — what was completed before it must still
            # reach `algo.chains` (ADVICE r4: the count was only raised behind a successful call)
            setfield!(algo, :stepped, SMMHip.hip_iter(getfield(algo, :hip)))
            rethrow()
        end
        algo.i += n
        if sf > 0 && fn != "" && algo.i % sf == 0
            save(algo, fn)
        end
    end
    algo.opts["time"] = round((time() - t0) / 60, digits = 1)
    if fn != ""
        save(algo, fn)
    end
    return nothing
end

# ---- BGPChain-shaped views --------------------------------------------------------------------------------------
"""
    sync_chains!(algo)

Fill `algo.chains` (real `SMM.BGPChain` objects) with what the device has computed since the last sync: `evals[t]`
(`value`, `params`, `simMoments`, `prob`, `accepted`, `status`), `accepted`, `exchanged`, `best_val`, `best_id`, `curr_val`,
`iter`, `sigma`, `accept_rate` (AlgoBGP.jl:42-110, set_eval! :220-245, set_exchanged! :246-249, set_acceptRate! :253-257).
Called by `getproperty(algo, :chains)`; cheap when nothing new happened.
"""
function sync_chains!(algo::MAlgoBGPHip)
    hip = getfield(algo, :hip)
    chains = getfield(algo, :chains)
    # nothing was stepped since the last sync: the chains are current (no device synchronisation, no download) — `algo.chains` is read
    # by every reader the reference defines, often many times in a row
    flush_steps!(algo)
    getfield(algo, :stepped) == getfield(algo, :synced) && return chains
    try
        SMMHip.hip_sync(hip)          # (the steps were only enqueued; a hard error of the reference is reported here)
    catch
        setfield!(algo, :stepped, SMMHip.hip_iter(hip))    # the failing iteration: everything up to it is materialised by the next read
        rethrow()
    end
    st = SMMHip.hip_state(hip)
    done = st.iter
    first = getfield(algo, :synced)
    # the exchange of iteration t rewrites the records of iteration t (swap_ev_ij!, :734-749) after they were first written:
    # the last synced iteration is downloaded again
    t0 = max(first - 1, 0)
    if done > t0
        h = SMMHip.hip_history(hip, t0, done)
        m = getfield(algo, :m)
        pn, mn = getfield(algo, :pnames), getfield(algo, :mnames)
        for (ci, c) in enumerate(chains)
            for (col, t) in enumerate(t0+1:done)
                ev = Eval(m)                                   # data moments and weights from the MProb (Eval.jl)
                ev.value = h.value[ci, col]
                ev.prob = h.prob[ci, col]
                ev.accepted = h.accepted[ci, col] != 0
                ev.status = Int(h.status[ci, col])
                ev.params = OrderedDict{Symbol,Float64}(pn[k] => h.params[ci, k, col] for k in 1:length(pn))
                ev.simMoments = OrderedDict{Symbol,Float64}(mn[k] => h.sim_moments[ci, k, col] for k in 1:length(mn))
                c.evals[t] = ev
                c.accepted[t] = ev.accepted
                c.exchanged[t] = Int(h.exchanged[ci, col])
                c.best_val[t] = h.best_val[ci, col]
                c.best_id[t] = Int(h.best_id[ci, col])
                c.curr_val[t] = h.curr_val[ci, col]
            end
        end
    end
    for (ci, c) in enumerate(chains)
        c.iter = done
        c.sigma = st.sigma[ci]
        c.accept_rate = st.accept_rate[ci]
    end
    setfield!(algo, :synced, done)
    return chains
end

function getproperty(algo::MAlgoBGPHip, s::Symbol)
    s === :chains && return sync_chains!(algo)
    return getfield(algo, s)
end

"`summary(m::MAlgoBGP)` (AlgoBGP.jl:541-550) on the synced chains"
summary(algo::MAlgoBGPHip) = vcat([summary(c) for c in algo.chains]...)

function show(io::IO, algo::MAlgoBGPHip)
    print(io, "\nBGP Algorithm with $(length(getfield(algo, :chains))) chains on libsmmhip (MI355X)\n")
    print(io, "============================\n\n")
    print(io, "Algorithm\n---------\n")
    print(io, "Current iteration: $(algo.i)\n")
    print(io, "Number of params to estimate: $(length(algo.m.params_to_sample))\n")
    print(io, "Number of moments to match: $(length(algo.m.moments))\n\n")
end

# ---- save / readMalgo / restart! --------------------------------------------------------------------------------
"""
    as_reference(algo::MAlgoBGPHip) -> SMM.MAlgoBGP

The run as a plain `SMM.MAlgoBGP` (same MProb, opts, iteration and the synced chains): what is written to disk, so that a
file saved from the GPU path is read by the reference's own `readMalgo` (AlgoAbstract.jl:95-102) on any machine.
"""
function as_reference(algo::MAlgoBGPHip)
    # the reference stores opts["dist_fun"] in a field typed ::Function (AlgoBGP.jl:503,537): a menu entry (:absdiff, "reldiff", ...)
    # would throw there, so the plain algo gets the host function and the menu entry moves to "hip_dist_fun" (read back on resume)
    opts = copy(algo.opts)
    if haskey(opts, "dist_fun") && !(opts["dist_fun"] isa Function)
        opts["hip_dist_fun"] = opts["dist_fun"]
        opts["dist_fun"] = getfield(algo, :dist_fun)
    end
    ref = MAlgoBGP(algo.m, opts)
    ref.chains = algo.chains
    ref.i = algo.i
    return ref
end

"`save(algo, filename)` (AlgoAbstract.jl:83-88): JLD2 file holding an `SMM.MAlgoBGP` named `algo`"
save(algo::MAlgoBGPHip, filename::AbstractString) = save(as_reference(algo), filename)

# number of iterations a chain counts towards its accept rate / how many of them accepted (set_acceptRate!, :253-257)
noex_counts(c::BGPChain) = (count(c.exchanged[1:c.iter] .== 0), count(c.accepted[1:c.iter] .& (c.exchanged[1:c.iter] .== 0)))

"""
    MAlgoBGPHip(ref::MAlgoBGP; extra_iter = 0)

Continue a run of the reference (for example one read back by `readMalgo`) on the GPU: a device context with
`maxiter + extra_iter` iterations of history, loaded with the chains' state after iteration `ref.i`.
"""
function MAlgoBGPHip(ref::MAlgoBGP; extra_iter::Int = 0)
    opts = copy(ref.opts)
    done = ref.chains[1].iter
    opts["maxiter"] = Int(ref.opts["maxiter"]) + extra_iter
    algo = MAlgoBGPHip(ref.m, opts)
    hip = getfield(algo, :hip)
    N, np, nm = hip.N, hip.np, hip.nm
    pn, mn = getfield(algo, :pnames), getfield(algo, :mnames)
    if done > 0
        value = Matrix{Float64}(undef, N, done); prob = similar(value); curr = similar(value); best = similar(value)
        pars = Array{Float64}(undef, N, np, done); simm = fill(NaN, N, nm, done)
        bid = Matrix{Int32}(undef, N, done); exch = similar(bid); accd = Matrix{UInt8}(undef, N, done); stt = Matrix{Int8}(undef, N, done)
        sigma = Vector{Float64}(undef, N); rate = similar(sigma); lav = similar(sigma); lap = similar(sigma)
        lapar = Matrix{Float64}(undef, N, np); lasm = fill(NaN, N, nm); last = Vector{Int8}(undef, N)
        nno = Vector{Int32}(undef, N); nac = similar(nno); bv = similar(sigma); bi = Vector{Int32}(undef, N)
        for (ci, c) in enumerate(ref.chains)
            for t in 1:done
                ev = c.evals[t]
                value[ci, t] = ev.value; prob[ci, t] = ev.prob; curr[ci, t] = c.curr_val[t]; best[ci, t] = c.best_val[t]
                bid[ci, t] = c.best_id[t]; exch[ci, t] = c.exchanged[t]; accd[ci, t] = c.accepted[t] ? 1 : 0; stt[ci, t] = ev.status
                for k in 1:np
                    pars[ci, k, t] = ev.params[pn[k]]
                end
                for k in 1:nm
                    simm[ci, k, t] = get(ev.simMoments, mn[k], NaN)
                end
            end
            la = SMM.getLastAccepted(c)                       # AlgoBGP.jl:209-217
            sigma[ci] = c.sigma; rate[ci] = c.accept_rate
            lav[ci] = la.value; lap[ci] = la.prob; last[ci] = la.status
            for k in 1:np
                lapar[ci, k] = la.params[pn[k]]
            end
            for k in 1:nm
                lasm[ci, k] = get(la.simMoments, mn[k], NaN)
            end
            nno[ci], nac[ci] = noex_counts(c)
            bv[ci] = c.best_val[done]; bi[ci] = c.best_id[done]
        end
        s = (iter = done, sigma = sigma, accept_rate = rate, la_value = lav, la_prob = lap, la_params = lapar,
             la_sim_moments = lasm, la_status = last, n_noex = nno, n_acc_noex = nac, best_val = bv, best_id = bi)
        h = (value = value, prob = prob, curr_val = curr, best_val = best, params = pars, sim_moments = simm,
             best_id = bid, exchanged = exch, accepted = accd, status = stt)
        SMMHip.hip_set_state!(hip, s, h)
    end
    setfield!(algo, :stepped, done)
    algo.i = done
    return algo
end

"""
    restart!(algo::MAlgoBGPHip, extraIter)

`restart!(algo, extraIter)` (AlgoBGP.jl:804-884): `extraIter` more iterations of the same chains.  The reference re-runs
iteration `algo.i` after extending (its loop starts at the old `maxiter`); here stepping continues at `algo.i + 1`.
"""
function restart!(algo::MAlgoBGPHip, extraIter::Int)
    ext = MAlgoBGPHip(as_reference(algo); extra_iter = extraIter)     # a new device context: the history capacity is fixed at creation
    SMMHip.hip_destroy!(getfield(algo, :hip))
    setfield!(algo, :hip, getfield(ext, :hip))                        # `algo` stays the object the caller holds, as in the reference
    setfield!(algo, :chains, getfield(ext, :chains))
    setfield!(algo, :opts, getfield(ext, :opts))
    setfield!(algo, :synced, 0)
    setfield!(algo, :stepped, getfield(ext, :stepped))
    setfield!(algo, :deferred, 0)
    run!(algo)
    return nothing
end

# ---- Examples.jl:118-153 ------------------------------------------------------------------------------------------
"`SMM.serialNormal(npars, niter)` on the device (2 parameters / 2 moments, 3 chains)"
function serialNormalHip(niter::Int = 200; nchains::Int = 3, acc_tuners = [20.0, 2.0, 1.0])
    pb = OrderedDict("p1" => [0.2, -3, 3], "p2" => [-0.2, -20, 20])
    moms = SMM.DataFrame(name = ["mu1", "mu2"], value = [-1.0, 10.0], weight = ones(2))
    mprob = MProb()
    SMM.addSampledParam!(mprob, pb)
    SMM.addMoment!(mprob, moms)
    SMM.addEvalFunc!(mprob, SMM.objfunc_norm)
    opts = Dict("N" => nchains, "maxiter" => niter, "maxtemp" => 5, "coverage" => 0.02, "smpl_iters" => 1000,
                "parallel" => false, "min_improve" => zeros(nchains), "acc_tuners" => Float64.(acc_tuners), "animate" => false)
    MA = MAlgoBGPHip(mprob, opts)
    run!(MA)
    return MA
end

# ------------------------------------------------------------------------------------------
# The other callers of evaluateObjective (slices.jl, econometrics.jl), as batches on the device: where the reference maps
# evaluateObjective over a grid (pmap / map), the whole grid is ONE call of smm_eval_batch.  (smm.jl_amd/callers.py is the
# tested mirror of the same drivers, tests/test_callers.py.)
# ------------------------------------------------------------------------------------------
"a one-chain context that only evaluates: `opts` as for MAlgoBGPHip (hip_objective, hip_ns, hip_obj_params, device)"
function eval_context(m::MProb, opts::Dict = Dict())
    init = Float64[m.initial_value[k] for k in keys(m.params_to_sample)]
    lb = Float64[v[:lb] for (k, v) in m.params_to_sample]
    ub = Float64[v[:ub] for (k, v) in m.params_to_sample]
    mom = Float64[v[:value] for (k, v) in m.moments]
    w = Float64[v[:weight] for (k, v) in m.moments]
    return SMMHip.hip_create(init, lb, ub, mom, w, [0.05], [1.0], [0.0]; maxiter = 1, ns = Int(get(opts, "hip_ns", 10000)),
                             objective_id = device_objective(m, opts), obj_params = Float64.(get(opts, "hip_obj_params", Float64[])),
                             device = Int(get(opts, "device", 0)))
end

"""
    evaluateObjectivesHip(m, ps; opts = Dict(), noseed_base = nothing) -> Vector{Eval}

`evaluateObjective(m, p)` (mprob.jl:175-205) for every parameter dict of `ps`, as one batch on the device.
"""
function evaluateObjectivesHip(m::MProb, ps::Vector; opts::Dict = Dict(), noseed_base = nothing)
    h = eval_context(m, opts)
    pk = collect(keys(m.params_to_sample))
    mk = collect(keys(m.moments))
    P = Float64[p[k] for p in ps, k in pk]                      # M x np
    value, simm, st = noseed_base === nothing ? SMMHip.hip_eval_batch(h, P) : SMMHip.hip_eval_batch_noseed(h, P, noseed_base)
    SMMHip.hip_destroy!(h)
    evs = Eval[]
    for (i, p) in enumerate(ps)
        ev = Eval(m, p)
        ev.value = value[i]
        ev.status = Int(st[i])
        for (j, k) in enumerate(mk)
            ev.simMoments[k] = simm[i, j]
        end
        push!(evs, ev)
    end
    return evs
end

"`doSlices(m, npoints)` (slices.jl:250-290): all np * npoints evaluations in one batch"
function doSlicesHip(m::MProb, npoints::Int; opts::Dict = Dict())
    res = Slice(m.initial_value, m.moments)
    ps = Any[]
    tags = Symbol[]
    for (pp, bb) in m.params_to_sample
        for pval in range(bb[:lb], stop = bb[:ub], length = npoints)
            p = deepcopy(m.initial_value)
            p[pp] = pval
            push!(ps, p)
            push!(tags, pp)
        end
    end
    for (pp, ev) in zip(tags, evaluateObjectivesHip(m, ps; opts = opts))
        SMM.add!(res, pp, ev)
    end
    return res
end

"`FD_gradient(m, p)` (econometrics.jl:29-85): the 1 + k (forward) or 2k (central) evaluations in one batch; rows in the order of `p`"
function FD_gradient_hip(m::MProb, p::Union{Dict,OrderedDict}; step_perc = 0.01, diff_method = :forward, use_range = true, opts::Dict = Dict())
    diff_method in (:forward, :central) || error("only :central and :foward implemented")
    rs = SMM.range_length(m)
    ks = collect(keys(p))
    hs = Float64[(use_range ? rs[k] : p[k]) * step_perc for k in ks]
    ps = Any[deepcopy(p)]
    for (k, h) in zip(ks, hs)
        if diff_method == :forward
            q = deepcopy(p); q[k] = p[k] + h; push!(ps, q)
        else
            q = deepcopy(p); q[k] = p[k] + 0.5 * h; push!(ps, q)
            q = deepcopy(p); q[k] = p[k] - 0.5 * h; push!(ps, q)
        end
    end
    evs = evaluateObjectivesHip(m, ps; opts = opts)
    mk = collect(keys(m.moments))
    g(ev) = Float64[ev.simMoments[k] for k in mk]
    gp = g(evs[1])
    D = zeros(length(ks), length(mk))
    for (i, h) in enumerate(hs)
        D[i, :] = diff_method == :forward ? (g(evs[1 + i]) .- gp) ./ h : (g(evs[2 * i]) .- g(evs[2 * i + 1])) ./ h
    end
    return D
end

"`getSigma(m, p, reps)` (econometrics.jl:125-145): `reps` evaluations with their own shock sequences (seed + i) in one batch"
function getSigmaHip(m::MProb, p::Union{Dict,OrderedDict}, reps::Int; seed::Integer = 0, opts::Dict = Dict())
    evs = evaluateObjectivesHip(m, Any[deepcopy(p) for i in 1:reps]; opts = opts, noseed_base = seed)
    mk = collect(keys(m.moments))
    X = Float64[ev.simMoments[k] for ev in evs, k in mk]
    mu = sum(X, dims = 1) ./ reps
    Xc = X .- mu
    return (Xc' * Xc) ./ (reps - 1)
end

end # module
