# julia/reference_golden.jl — THE PIN THIS BUILD CANNOT MAKE ITSELF.
#
# Drives the REFERENCE'S OWN deterministic pieces of the BGP path (floswald/SMM.jl) with every random number injected, and writes what
# they produce as golden vectors: tests/golden/ref_bgp.json (+ ref_Z.bin).  tests/test_golden.py::test_oracle_matches_reference_vectors
# replays the same inputs through the build's CPU oracle (oracle/smm_oracle.c) and — on a GPU box — through libsmmhip, and compares:
# bookkeeping exactly, floating point within 1e-12 relative.  With these files committed, the oracle is pinned to the reference itself
# ("parity unpinned" in DESIGN.md / oracle/smm_oracle.c's header goes away).
#
# No julia binary exists in the build image: THIS FILE HAS NEVER BEEN EXECUTED.  It needs Julia >= 1.6 with SMM.jl (which brings
# Random, Distributions, StatsBase, DataFrames, OrderedCollections, JSON):
#
#     julia --project=/path/to/SMM.jl julia/reference_golden.jl [tests/golden]
#
# What runs is the reference's code, not a restatement (file:line of /root/reference = SMM.jl at the surveyed commit):
#     SMM.MProb / addSampledParam! / addMoment! / addEvalFunc!          src/mopt/mprob.jl:29-155         (the serialNormal problem, Examples.jl:373-416)
#     SMM.MAlgoBGP(m, opts)  ->  BGPChain constructors                   src/mopt/AlgoBGP.jl:497-539, :78-109
#     SMM.evaluateObjective(m, p)  ->  SMM.objfunc_norm                  src/mopt/mprob.jl:175-188, src/mopt/ObjExamples.jl:59-116
#     SMM.doAcceptReject!(c, ev), SMM.set_eval!(c, ev)                   src/mopt/AlgoBGP.jl:324-392, :220-245     (= next_eval, :272-294, minus proposal)
#     SMM.exchangeMoves!(algo)  ->  swap_ev_ij!, set_exchanged!          src/mopt/AlgoBGP.jl:647-716, :734-749, :246-249
#     SMM.mapto_01, SMM.mapto_ab                                         src/mopt/mprob.jl:246-272
# What is injected, and how the reference is made to take it:
#     probs_acc   BGPChain.probs_acc = rand(n) (:85) is a plain field: overwritten with a table before the first iteration.
#     proposals   `proposal` (:424-471) draws through SMM.RAND = RandomDevice() (src/SMM.jl:60, AlgoBGP.jl:404), which cannot be seeded, so
#                 its TWO arithmetic lines are inlined below around the reference's own mapto_01 / mapto_ab: x = mu01 .+ sqrt(abs2(sigma)) .* z
#                 (what rand(MvNormal(mu01, sigma)) computes in Distributions: ScalMat(abs2(sigma)), unwhiten! multiplies by its sqrt), with z
#                 from a table.  The table is small enough (|z| <= 1) that the first try always lies in [0,1]^k — asserted — so mysample's
#                 rejection loop (:400-410) never needs a second draw.
#     Z           objfunc_norm re-seeds the GLOBAL generator with 1234 on every call and draws rand(MvNormal(mu, I), 10000).  The same
#                 seed + a zero mean yields the shocks themselves (mu .+ 1.0 .* z with mu = 0): exported as ref_Z.bin so that the oracle
#                 simulates from THE SAME 2 x 10000 normals whatever Julia's randn is in the maintainer's version.
#     pairs       exchangeMoves! samples its pairs with StatsBase.sample on the global generator: it is seeded per iteration, the pairs
#                 are drawn once by the same call to know them, the generator is seeded again and exchangeMoves! itself runs.
using SMM, Random, JSON, DataFrames
using OrderedCollections: OrderedDict
import StatsBase

outdir = length(ARGS) >= 1 ? ARGS[1] : joinpath(@__DIR__, "..", "tests", "golden")
N, T = 6, 40                          # chains, iterations (small on purpose: the fixture is a few hundred KB with Z)

# ---- the serialNormal problem (Examples.jl:373-416), two parameters / two moments ----
pb = OrderedDict()
pb["p1"] = [0.2, -3, 3]
pb["p2"] = [-0.2, -20, 20]
moms = DataFrame(name = ["mu1", "mu2"], value = [-1.0, 10.0], weight = ones(2))
mprob = MProb()
addSampledParam!(mprob, pb)
SMM.addMoment!(mprob, moms)
SMM.addEvalFunc!(mprob, SMM.objfunc_norm)

opts = Dict("N" => N, "maxiter" => T, "maxtemp" => 2, "sigma" => 0.05, "sigma_update_steps" => 10, "sigma_adjust_by" => 0.01,
            "smpl_iters" => 1000, "parallel" => false, "min_improve" => [0.0, 0.0, 0.05, 0.0, 0.5, 0.0],
            "acc_tuners" => [20.0, 10.0, 5.0, 2.0, 1.5, 1.0], "animate" => false)
algo = MAlgoBGP(mprob, opts)

# (maxtemp 2: sigma 0.05 .. 0.10 in [0,1]-space, so that with |z| < 1 a chain near the optimum — mu01 = (0.33, 0.75) — cannot step out of the box)
# ---- injected tables (a 64-bit LCG: nothing of Julia's generators in the inputs) ----
mutable struct LCG; s::UInt64; end
function u01!(g::LCG)
    g.s = g.s * 0x5851f42d4c957f2d + 0x14057b7ef767814f
    return Float64(g.s >> 11) / 9007199254740992.0          # 53 bits in [0,1)
end
g = LCG(0x0123456789abcdef)
probs_acc = [u01!(g) for t in 1:T, c in 1:N]                 # [T][N]
zprop = [2.0 * u01!(g) - 1.0 for t in 1:T, c in 1:N, k in 1:2]   # [T][N][np], |z| < 1
for c in 1:N
    algo.chains[c].probs_acc = probs_acc[:, c]
end

# ---- the shock matrix objfunc_norm will see (ObjExamples.jl:71-79) ----
Random.seed!(1234)
Z = rand(SMM.MvNormal(zeros(2), SMM.PDiagMat(ones(2))), 10000)      # [nm][ns]: exactly the z of mu .+ 1.0 .* z
# layout: moment 1's 10000 shocks, then moment 2's; little-endian float64
open(joinpath(outdir, "ref_Z.bin"), "w") do io
    for k in 1:2, s in 1:10000
        write(io, Float64(Z[k, s]))
    end
end

lb = [v[:lb] for (k, v) in mprob.params_to_sample]
ub = [v[:ub] for (k, v) in mprob.params_to_sample]
pairs_all = Vector{Vector{Vector{Int}}}()
tries_in_support = true

for t in 1:T
    algo.i = t
    for c in algo.chains
        # ---- next_eval (AlgoBGP.jl:272-294) with the proposal's draw injected ----
        c.iter += 1
        if c.iter == 1
            pp = c.m.initial_value                                       # proposal, :426-427
        else
            ev_old = SMM.getLastAccepted(c)
            mu = SMM.paramd(ev_old)
            mu01 = SMM.mapto_01(mu, lb, ub)                              # :437
            x = mu01 .+ sqrt(abs2(c.sigma)) .* zprop[t, c.id, :]         # rand(MvNormal(mu01, c.sigma)), :442 — see the header
            all(0.0 .<= x .<= 1.0) || (global tries_in_support = false)  # mysample's first try must do, :405
            pp = OrderedDict(zip(collect(keys(mu)), SMM.mapto_ab(x, lb, ub)))   # :457
        end
        ev = SMM.evaluateObjective(c.m, pp)                              # :283 (objfunc_norm re-seeds 1234 itself)
        SMM.doAcceptReject!(c, ev)                                       # :287
        SMM.set_eval!(c, ev)                                             # :290
    end
    if algo.i >= 2 && N > 1                                              # computeNextIteration!, :637-639
        props = [(i, j) for i in 1:N, j in 1:N if (i < j)]               # exchangeMoves!, :653
        Random.seed!(100000 + t)
        drawn = StatsBase.sample(props, N < 3 ? N - 1 : N, replace = false)   # :656 — the same call, to know the pairs
        push!(pairs_all, [[p[1], p[2]] for p in drawn])
        Random.seed!(100000 + t)
        SMM.exchangeMoves!(algo)
    else
        push!(pairs_all, Vector{Vector{Int}}())
    end
end
tries_in_support || error("an injected proposal left [0,1]^k: shrink zprop or sigma (the fixture assumes mysample's first try)")

# ---- what the reference holds afterwards ----
chains = []
for c in algo.chains
    push!(chains, Dict(
        "value" => [c.evals[t].value for t in 1:T], "prob" => [c.evals[t].prob for t in 1:T],
        "status" => [c.evals[t].status for t in 1:T], "accepted" => [Int(c.accepted[t]) for t in 1:T],
        "exchanged" => c.exchanged[1:T], "curr_val" => c.curr_val[1:T], "best_val" => c.best_val[1:T], "best_id" => c.best_id[1:T],
        "params" => [[c.evals[t].params[k] for k in keys(c.evals[t].params)] for t in 1:T],
        "sim_moments" => [[get(c.evals[t].simMoments, Symbol(m), NaN) for m in ("mu1", "mu2")] for t in 1:T],
        "sigma" => c.sigma, "accept_rate" => c.accept_rate))
end
out = Dict("N" => N, "T" => T, "ns" => 10000, "init" => [0.2, -0.2], "lb" => lb, "ub" => ub, "mom" => [-1.0, 10.0], "w" => [1.0, 1.0],
           "sigma0" => [c_sigma for c_sigma in (0.05 .* collect(range(1.0, stop = 2, length = N)))],
           "acc_tuners" => opts["acc_tuners"], "min_improve" => opts["min_improve"], "sigma_update_steps" => 10, "sigma_adjust_by" => 0.01,
           "probs_acc" => [[probs_acc[t, c] for c in 1:N] for t in 1:T],
           "prop_normals" => [[[zprop[t, c, k] for k in 1:2] for c in 1:N] for t in 1:T],
           "pairs" => pairs_all,                                         # 1-based (i, j), in the order exchangeMoves! walked them
           "chains" => chains,
           "julia_version" => string(VERSION), "note" => "generated by julia/reference_golden.jl from the reference's own functions")
open(joinpath(outdir, "ref_bgp.json"), "w") do io
    JSON.print(io, out)
end
println("wrote ", joinpath(outdir, "ref_bgp.json"), " and ref_Z.bin: ", N, " chains x ", T, " iterations; exchanged ",
        sum(sum(c["exchanged"] .!= 0) for c in chains), " chain-iterations")
