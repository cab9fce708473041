# SMMHipSharded.jl — the BGP path of SMM.jl over several GPUs of one node: one Julia worker process per GPU, `Distributed` only.
#
#     using Distributed; addprocs(8)                       # one worker per GPU (the reference's own way: test/test_algoBGP.jl:43-54)
#     @everywhere begin
#         using SMM
#         include("julia/SMMHip.jl"); include("julia/SMMHipBackend.jl"); include("julia/SMMHipSharded.jl")
#         using .SMMHipSharded
#     end
#     sh = ShardedBGPHip(mprob, opts)                       # opts["N"] chains in all, N / nworkers() per GPU
#     run!(sh)                                              # opts["maxiter"] iterations
#     h = history_sharded(sh)                               # value[chain, t], exchanged[chain, t], ... of the whole population
#
# What replaces what.  The reference's parallel branch (AlgoBGP.jl:596-605) ships every proposal to a worker with `pmap` and every
# `Eval` back, once per chain and iteration; exchangeMoves! (:647-691) then runs on the master.  Here the chains LIVE on the workers'
# GPUs — contiguous blocks of opts["N"] / G chains, all per-chain state and history on the owning GPU — and the iteration needs no
# Julia-side communication at all: the shards exchange through windows of device memory that every rank maps once (HIP IPC, the p2p
# form of include/smmhip.h), and where the context qualifies a whole step is ONE persistent kernel launch per rank whose tiles talk
# through a ring of tagged words in those windows (smm.jl_amd/csrc/smm_chain_persist_loc.hpp).  Distributed carries, once, 64 bytes
# per rank — the IPC handles — and afterwards only the calls "step n" / "finish" / "give me your history".
#
# The barrier the library asks for between a finish and the next step's first publication (include/smmhip.h) is the `@sync` around
# the workers' calls: the master returns from `sync_sharded!` only when every worker has.
#
# Results are those of the single-GPU run with the same opts (RNG counters use global chain ids), to the bit.
#
# NOT EXECUTED IN THIS REPOSITORY'S CI (no julia binary in the image); tests/test_julia_layer.py checks what can be checked without
# one: every `ccall`-level function it uses exists in SMMHip.jl, the packages it loads are the standard library's or SMM.jl's own.
module SMMHipSharded

using Distributed
using SMM
import SMM: MProb, run!
using ..SMMHip
using ..SMMHipBackend

export ShardedBGPHip, step_sharded!, sync_sharded!, history_sharded, state_sharded, destroy_sharded!

# this worker's shard (one per process: a worker drives one GPU)
const SHARD = Ref{Union{Nothing,SMMHip.HipBGP}}(nothing)

"on a worker: create this rank's shard, return its window's IPC handle"
function shard_create(m::MProb, opts::Dict, rank::Int, G::Int)
    N = Int(opts["N"])
    N % G == 0 || throw(ArgumentError("opts[\"N\"] = $N chains do not split into $G equal shards"))
    n = N ÷ G
    # the HIP device of this rank: opts["devices"] (one ordinal per rank) where given, else the rank itself
    devs = get(opts, "devices", nothing)
    dev = devs === nothing ? rank : Int(devs[rank + 1])
    hip, = SMMHipBackend.hip_context(m, opts; N_local = n, chain_offset = rank * n, device = dev)
    SHARD[] = hip
    return SMMHip.hip_p2p_init(hip)
end

"on a worker: map the other ranks' windows"
function shard_attach(rank::Int, handles::Vector{Vector{UInt8}})
    hip = SHARD[]
    for (r, h) in enumerate(handles)
        r - 1 == rank || SMMHip.hip_p2p_attach!(hip, r - 1, h)
    end
    return nothing
end

shard_step(n::Int) = (SMMHip.hip_p2p_step!(SHARD[], n); nothing)                       # enqueued only
shard_finish() = (SMMHip.hip_p2p_finish!(SHARD[]); SMMHip.hip_sync(SHARD[]); nothing)   # the ranks' rendezvous (errors are agreed upon here)
shard_history(t0::Int, t1::Int) = SMMHip.hip_history(SHARD[], t0, t1)
shard_state() = SMMHip.hip_state(SHARD[])
shard_destroy() = (SHARD[] === nothing || SMMHip.hip_destroy!(SHARD[]); SHARD[] = nothing; nothing)

"""
    ShardedBGPHip(m::MProb, opts::Dict; pool = workers())

`MAlgoBGP(m, opts)` (AlgoBGP.jl:505-537) as `length(pool)` shards, worker `pool[r + 1]` driving GPU `r`.
"""
mutable struct ShardedBGPHip
    m::MProb
    opts::Dict
    i::Int                      # iterations enqueued so far
    pool::Vector{Int}
end

function ShardedBGPHip(m::MProb, opts::Dict; pool::Vector{Int} = workers())
    G = length(pool)
    G >= 1 && G <= 8 || throw(ArgumentError("one to eight workers (the GPUs of one node)"))
    handles = Vector{Vector{UInt8}}(undef, G)
    @sync for (r, w) in enumerate(pool)
        @async handles[r] = remotecall_fetch(shard_create, w, m, opts, r - 1, G)
    end
    # every window is mapped everywhere before anybody stores into one: the second @sync is that barrier
    @sync for (r, w) in enumerate(pool)
        @async remotecall_fetch(shard_attach, w, r - 1, handles)
    end
    return ShardedBGPHip(m, opts, 0, pool)
end

"`n` iterations on every shard (enqueued: the workers return at once; `sync_sharded!` waits)"
function step_sharded!(sh::ShardedBGPHip, n::Int)
    @sync for w in sh.pool
        @async remotecall_fetch(shard_step, w, n)
    end
    sh.i += n
    return sh
end

"settle the last iteration on every shard and wait for the devices; returns when EVERY rank has (the barrier of include/smmhip.h)"
function sync_sharded!(sh::ShardedBGPHip)
    @sync for w in sh.pool
        @async remotecall_fetch(shard_finish, w)
    end
    return sh
end

"""
    run!(sh::ShardedBGPHip)

`run!` (AlgoAbstract.jl:27-76) over the shards: all remaining iterations in one step per save interval.
"""
function run!(sh::ShardedBGPHip)
    maxiter = Int(sh.opts["maxiter"])
    sf = Int(get(sh.opts, "save_frequency", 0))
    while sh.i < maxiter
        n = maxiter - sh.i
        if sf > 0
            n = min(n, sf - (sh.i % sf))
        end
        step_sharded!(sh, n)
        sync_sharded!(sh)
    end
    return nothing
end

"the whole population's history of iterations `t0 + 1 .. t1`: the shards' NamedTuples (SMMHip.hip_history) joined along the chains"
function history_sharded(sh::ShardedBGPHip, t0::Int = 0, t1::Int = sh.i)
    parts = Vector{Any}(undef, length(sh.pool))
    @sync for (r, w) in enumerate(sh.pool)
        @async parts[r] = remotecall_fetch(shard_history, w, t0, t1)
    end
    joined = map(keys(parts[1])) do k
        k => cat((getfield(p, k) for p in parts)...; dims = 1)       # the chain is the first index of every field
    end
    return (; joined...)
end

"the shards' states (SMMHip.hip_state), in rank order"
function state_sharded(sh::ShardedBGPHip)
    parts = Vector{Any}(undef, length(sh.pool))
    @sync for (r, w) in enumerate(sh.pool)
        @async parts[r] = remotecall_fetch(shard_state, w)
    end
    return parts
end

"drop the shards' contexts — every rank together: nobody unmaps a window a peer may still store into"
function destroy_sharded!(sh::ShardedBGPHip)
    sync_sharded!(sh)
    @sync for w in sh.pool
        @async remotecall_fetch(shard_destroy, w)
    end
    return nothing
end

end # module
