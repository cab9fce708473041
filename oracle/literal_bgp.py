"""literal_bgp.py — a SECOND, literal restatement of the reference's BGP path, to cross-check oracle/smm_oracle.c.

TEST INFRASTRUCTURE ONLY (imported by tests/test_literal_oracle.py; never by smm.jl_amd/, bench.py's timed region or the
library).  PARITY UNPINNED like the C oracle: the reference is Julia, nothing of it runs in the build image.
This file does NOT make parity green.  What it removes is one blind spot: smm_oracle.c and the HIP kernels share a
structure-of-arrays formulation with shortcuts that the reference does not have —
    * "the current value of a chain" carried as la_value instead of curr_val[iter-1] / getLastAccepted(c).value,
    * the accept rate from running counters n_noex / n_acc_noex instead of mean(accepted[1:iter][exchanged[1:iter] .== 0]),
    * records as flat blocks that are copied, instead of Eval objects that are deep-copied,
so a misreading hidden in such a shortcut would be invisible to every HIP-vs-oracle test.  Here the path is written the way the
reference writes it: per-chain objects holding evals[] / accepted[] / exchanged[] / best_val[] / curr_val[] arrays of length
maxiter, lastAccepted as a findlast over accepted[1:iter], set_acceptRate! recomputing the mean over the whole history on every
call, set_eval! deep-copying the Eval, exchangeMoves! walking the pairs over those objects.  Plain Python, no numpy, slow on
purpose (O(iter) per call like the reference); tests feed it the injected randomness tables of the golden fixtures and of
random small cases and demand BIT-equality with the C oracle.

Each function cites the reference lines it follows (/root/reference/src/mopt/...).  Randomness is injected (the reference's
RandomDevice / Random.seed!(1234) / sample streams cannot be reproduced): probs_acc[t][c], prop_normals[t][try][k][c],
pairs[t][q] = (i, j) 0-based, Z[k][s].  The numerical contract of the objective's sample mean (include/smmhip.h,
SMM_REDUCE_LANES) is the one thing taken from the build rather than from the reference: without a fixed summation order two
implementations cannot be compared bit by bit.
"""
import copy
import math


def _fma(a, b, c):
    """a * b + c with ONE rounding (exact rational arithmetic, then the correctly rounded conversion): Python 3.10 has no math.fma"""
    from fractions import Fraction
    return float(Fraction(a) * Fraction(b) + Fraction(c))


def contract_exp(x):
    """the exponential of the numerical contract (include/smmhip.h: range reduction, the Taylor coefficients 1/2! .. 1/13! by Estrin's
    scheme in fma, ldexp), in plain Python floats"""
    if x != x:
        return x
    if x > 709.782712893383973096:
        return math.inf
    if x < -745.13321910194110842:
        return 0.0
    k = float(round(x * 1.44269504088896338700e+00))     # round half to even, as rint
    r = _fma(-k, 6.93147180369123816490e-01, x)
    r = _fma(-k, 1.90821492927058770002e-10, r)
    r2 = r * r
    r4 = r2 * r2
    r8 = r4 * r4
    a0, a1, a2 = _fma(1.0 / 6.0, r, 0.5), _fma(1.0 / 120.0, r, 1.0 / 24.0), _fma(1.0 / 5040.0, r, 1.0 / 720.0)
    a3, a4, a5 = _fma(1.0 / 362880.0, r, 1.0 / 40320.0), _fma(1.0 / 39916800.0, r, 1.0 / 3628800.0), _fma(1.0 / 6227020800.0, r, 1.0 / 479001600.0)
    b0, b1, b2 = _fma(a1, r2, a0), _fma(a3, r2, a2), _fma(a5, r2, a4)
    q = _fma(b2, r8, _fma(b1, r4, b0))
    p = _fma(r2, q, r)
    return math.ldexp(1.0 + p, int(k))

REDUCE_LANES = 512   # include/smmhip.h: the ns draws of a moment are summed as 512 lane-strided partial sums, ...

OBJ_NORM, OBJ_NORM_FAILBOX = 0, 2   # smm_objective_t (include/smmhip.h)


class Eval:
    """Eval.jl:33-106, the fields the path touches: value (default -1.0, :84), params, simMoments, status (:90), prob, accepted"""

    def __init__(self, params):
        self.value = -1.0
        self.params = list(params)
        self.simMoments = None       # (an empty Dict in the reference until setMoments!)
        self.status = -1
        self.prob = 0.0
        self.accepted = False


class BGPChain:
    """BGPChain, AlgoBGP.jl:42-110 (constructor :78-109)"""

    def __init__(self, cid, n, sigma, acc_tuner, min_improve, sigma_update_steps, sigma_adjust_by, smpl_iters, batches, probs_acc):
        self.id = cid
        self.iter = 0
        self.evals = [None] * n
        self.best_id = [-1] * n                  # :83
        self.best_val = [math.inf] * n           # :82 (ones(n) * Inf)
        self.curr_val = [math.inf] * n           # :81
        self.probs_acc = list(probs_acc)         # :85 rand(n): injected
        self.accepted = [False] * n              # :88
        self.exchanged = [0] * n                 # :90
        self.accept_rate = 0.0                   # :89
        self.acc_tuner = acc_tuner
        self.sigma = sigma
        self.sigma_update_steps = sigma_update_steps
        self.sigma_adjust_by = sigma_adjust_by
        self.smpl_iters = smpl_iters
        self.min_improve = min_improve
        self.batches = batches                   # :95-103, for batch sizes that divide np


def lastAccepted(c):
    """AlgoBGP.jl:209-215: findlast(c.accepted[1:c.iter]) (1-based iteration)"""
    if c.iter == 1:
        return 1
    for t in range(c.iter, 0, -1):
        if c.accepted[t - 1]:
            return t
    raise RuntimeError("no accepted iteration")   # findlast -> nothing: cannot happen, iteration 1 accepts everything


def getLastAccepted(c):
    """:217"""
    return c.evals[lastAccepted(c) - 1]


def set_eval(c, ev):
    """set_eval!, AlgoBGP.jl:220-245"""
    it = c.iter
    c.evals[it - 1] = copy.deepcopy(ev)          # :221
    c.accepted[it - 1] = ev.accepted             # :222
    if it == 1:                                  # :224-227
        c.best_val[0] = ev.value
        c.curr_val[0] = ev.value
        c.best_id[0] = it
    else:
        if ev.accepted:                          # :230-234
            c.curr_val[it - 1] = ev.value
        else:
            c.curr_val[it - 1] = c.curr_val[it - 2]
        if ev.value < c.best_val[it - 2]:        # :235-237 (the proposal's value, accepted or not)
            c.best_val[it - 1] = ev.value
            c.best_id[it - 1] = it
        else:                                    # :238-242
            c.best_val[it - 1] = c.best_val[it - 2]
            c.best_id[it - 1] = c.best_id[it - 2]


def set_exchanged(c, i):
    """:246-249"""
    c.exchanged[c.iter - 1] = i


def set_acceptRate(c):
    """set_acceptRate!, AlgoBGP.jl:253-257: mean(acc[noex]) over the WHOLE history so far, recomputed on every call"""
    acc = [c.accepted[t] for t in range(c.iter) if c.exchanged[t] == 0]
    c.accept_rate = (sum(1 for a in acc if a) / len(acc)) if acc else math.nan   # (mean of an empty Bool vector is NaN)


class NegativeObjective(Exception):
    pass


class NoDrawInSupport(Exception):
    pass


def doAcceptReject(c, eval_new):
    """doAcceptReject!, AlgoBGP.jl:324-392"""
    if c.iter == 1:                               # :326-332
        eval_new.prob = 1.0
        eval_new.accepted = True
        eval_new.status = 1
        c.accepted[c.iter - 1] = eval_new.accepted
        set_acceptRate(c)
        return
    eval_old = getLastAccepted(c)                 # :334
    if eval_new.status < 0:                       # :336-338
        eval_new.prob = 0.0
        eval_new.accepted = False
    else:
        if not (eval_new.value >= 0):             # :341
            raise NegativeObjective("chain %d iteration %d" % (c.id, c.iter))
        x = c.acc_tuner * (eval_old.value - eval_new.value)
        e = contract_exp(x)                       # (Base.exp in the reference; the contract's exponential here: include/smmhip.h)
        eval_new.prob = e if e != e else min(1.0, e)   # minimum([1.0, e]) propagates NaN, :344
        if not math.isfinite(eval_new.prob):      # :350-353
            eval_new.prob = 0.0
            eval_new.accepted = False
            eval_new.status = -1
        elif not math.isfinite(eval_old.value):   # :355-359
            eval_new.prob = 1.0
            eval_new.accepted = True
        else:                                     # :360-367
            eval_new.status = 1
            eval_new.accepted = eval_new.prob > c.probs_acc[c.iter - 1]
    c.accepted[c.iter - 1] = eval_new.accepted    # :373
    set_acceptRate(c)                             # :374
    if c.iter % c.sigma_update_steps == 0:        # :381-390
        if c.accept_rate > 0.234:
            c.sigma = c.sigma * (1.0 + c.sigma_adjust_by)
        else:
            c.sigma = c.sigma * (1.0 - c.sigma_adjust_by)


def mysample(mu01, sigma, normals_of_try, smpl_iters, max_injected):
    """mysample, AlgoBGP.jl:400-410, on MvNormal(mu01, sigma::Float64) (isotropic, sigma the std-dev): x = mu + sigma * z, the
    whole vector redrawn until every component is in [0, 1] (inclusive, :405).  normals_of_try(r) -> the try's standard normals."""
    for r in range(min(smpl_iters, max_injected)):
        z = normals_of_try(r)
        x = [m + sigma * zz for m, zz in zip(mu01, z)]
        if all(v >= 0.0 for v in x) and all(v <= 1.0 for v in x):
            return x
    raise NoDrawInSupport()                        # :409


def proposal(c, m, normals):
    """proposal, AlgoBGP.jl:424-471; mapto_01 / mapto_ab mprob.jl:246-249, :270-272.  normals(r, k): standard normal of try r,
    parameter k, for this chain and iteration."""
    if c.iter == 1:
        return list(m["init"])                     # :426-427
    ev_old = getLastAccepted(c)                    # :429
    mu, lb, ub = ev_old.params, m["lb"], m["ub"]
    mu01 = [(x - a) / (b - a) for x, a, b in zip(mu, lb, ub)]          # mprob.jl:248
    pp = [0.0] * len(mu01)                         # :445
    for batch in c.batches:                        # one batch: :441-442; several: :444-453 (errors are raised, not swallowed)
        x = mysample([mu01[k] for k in batch], c.sigma, lambda r: [normals(r, k) for k in batch], c.smpl_iters, m["tries"])
        for k, v in zip(batch, x):
            pp[k] = v
    return [z * (b - a) + a for z, a, b in zip(pp, lb, ub)]            # mprob.jl:271


def reduce_partials(p):
    """the numerical contract's tree (include/smmhip.h): inside each group of 64 partial sums the halving tree (offsets 32 .. 1),
    the 8 group totals added left to right"""
    tot = 0.0
    for g in range(REDUCE_LANES // 64):
        q = p[64 * g:64 * g + 64]
        off = 32
        while off >= 1:
            for i in range(off):
                q[i] = q[i] + q[i + off]
            off //= 2
        tot = q[0] if g == 0 else tot + q[0]
    return tot


def objfunc_norm(ev, m):
    """objfunc_norm, ObjExamples.jl:59-116: X[k, s] = mu_k + z[k, s] (:76-78), simM = mean(X, dims = 2) (:79), per moment
    ((simM_k - mom_k) / w_k)^2 — without the division when the moment has no weight (:90-100) —, value = their mean (:101)"""
    mu, Z, ns = ev.params, m["Z"], m["ns"]
    nm = len(m["mom"])
    simM, v = [], []
    for k in range(nm):
        part = [0.0] * REDUCE_LANES
        for s in range(ns):                        # lane s % 512 takes the draws s, s + 512, ... in that order
            x = Z[k][s] + mu[k]
            part[s % REDUCE_LANES] = part[s % REDUCE_LANES] + x
        sm = reduce_partials(part) / ns
        simM.append(sm)
        d = sm - m["mom"][k]
        if not math.isnan(m["w"][k]):
            d = d / m["w"][k]
        v.append(d * d)
    total = v[0]
    for x in v[1:]:
        total = total + x
    ev.value = total / nm                          # :101
    ev.simMoments = simM                           # :106
    ev.status = 1                                  # :110
    return ev


def evaluateObjective(m, p):
    """mprob.jl:175-188: ev = Eval(m, p); ev = m.objfunc(ev) inside try; an exception leaves the fresh Eval with status -2"""
    ev = Eval(p)
    try:
        if m["objective_id"] == OBJ_NORM_FAILBOX and m["objp"][0] <= p[0] <= m["objp"][1]:
            raise RuntimeError("objective failed")   # the role of Testobj_fails, ObjExamples.jl:27-32
        ev = objfunc_norm(ev, m)
    except RuntimeError:
        ev.status = -2                             # mprob.jl:183-186
    return ev


def next_eval(c, m, normals):
    """next_eval, AlgoBGP.jl:272-294"""
    c.iter += 1
    pp = proposal(c, m, normals)
    ev = evaluateObjective(m, pp)
    doAcceptReject(c, ev)
    set_eval(c, ev)
    return c


def swap_ev_ij(chains, i, j):
    """swap_ev_ij!, AlgoBGP.jl:734-749 (i, j 1-based chain ids)"""
    ci, cj = chains[i - 1], chains[j - 1]
    ei = getLastAccepted(ci)
    ej = getLastAccepted(cj)
    set_eval(ci, ej)
    set_eval(cj, ei)
    set_exchanged(ci, j)
    set_exchanged(cj, i)


def dist_fun_of(kind):
    """opts["dist_fun"], AlgoBGP.jl:537: `-` by default; the build's menu (smm_dist_fun_t) for the others"""
    if kind == 0:
        return lambda a, b: a - b
    if kind == 1:
        return lambda a, b: abs(a - b)
    return lambda a, b: (a - b) / abs(a)


def exchangeMoves(chains, pairs, dist_fun):
    """exchangeMoves!, AlgoBGP.jl:647-691: for every sampled pair, in order: swap when dist_fun(v_i, v_j) > min_improve_i"""
    for (i, j) in pairs:                           # (1-based ids, i < j)
        evi = getLastAccepted(chains[i - 1])
        evj = getLastAccepted(chains[j - 1])
        if dist_fun(evi.value, evj.value) > chains[i - 1].min_improve:      # :688
            swap_ev_ij(chains, i, j)


class MAlgoBGP:
    """MAlgoBGP, AlgoBGP.jl:497-539, on flat inputs: problem dict m (init, lb, ub, mom, w, ns, Z, objective_id, objp, tries),
    per-chain sigma / acc_tuner / min_improve vectors, and the injected tables"""

    def __init__(self, m, N, maxiter, sigma, acc_tuner, min_improve, sigma_update_steps, sigma_adjust_by, smpl_iters, batch_size,
                 probs_acc, prop_normals, pairs, dist_fun=0, exchange_from_iter=2):
        npar = len(m["init"])
        batches = [list(range(b0, b0 + batch_size)) for b0 in range(0, npar, batch_size)]
        self.m = m
        self.i = 0
        self.N = N
        self.prop_normals = prop_normals           # [t][try][k][c]
        self.pairs = pairs                         # [t][q] = (i, j), 0-based
        self.dist_fun = dist_fun_of(dist_fun)
        self.exchange_from_iter = exchange_from_iter
        self.chains = [BGPChain(c + 1, maxiter, sigma[c], acc_tuner[c], min_improve[c], sigma_update_steps, sigma_adjust_by,
                                smpl_iters, batches, [probs_acc[t][c] for t in range(maxiter)]) for c in range(N)]

    def computeNextIteration(self):
        """computeNextIteration!, AlgoBGP.jl:589-640 (serial branch :614; exchange :637-639)"""
        t = self.i
        for c, ch in enumerate(self.chains):
            next_eval(ch, self.m, lambda r, k, c=c: self.prop_normals[t - 1][r][k][c])
        for ch in self.chains:
            assert ch.iter == self.i               # :630-632
        if self.i >= self.exchange_from_iter and self.N > 1:
            exchangeMoves(self.chains, [(int(i) + 1, int(j) + 1) for (i, j) in self.pairs[t - 1]], self.dist_fun)

    def run(self, n):
        """run!, AlgoAbstract.jl:38-45"""
        for _ in range(n):
            self.i += 1
            self.computeNextIteration()

    # ---- what the callers read back (history, AlgoBGP.jl:138-160; the scalar state of save / restart!) ----
    def history(self):
        T, N = self.i, self.N
        h = {k: [[None] * N for _ in range(T)] for k in ("value", "prob", "curr_val", "best_val", "best_id", "exchanged", "accepted", "status",
                                                           "params", "sim_moments")}
        for c, ch in enumerate(self.chains):
            for t in range(T):
                ev = ch.evals[t]
                h["value"][t][c] = ev.value; h["prob"][t][c] = ev.prob; h["status"][t][c] = ev.status
                h["curr_val"][t][c] = ch.curr_val[t]; h["best_val"][t][c] = ch.best_val[t]; h["best_id"][t][c] = ch.best_id[t]
                h["exchanged"][t][c] = ch.exchanged[t]; h["accepted"][t][c] = 1 if ch.accepted[t] else 0
                h["params"][t][c] = list(ev.params)
                h["sim_moments"][t][c] = list(ev.simMoments) if ev.simMoments is not None else [math.nan] * len(self.m["mom"])
        return h

    def state(self):
        s = {"sigma": [], "accept_rate": [], "la_value": [], "la_prob": [], "la_status": [], "la_params": [], "n_noex": [], "n_acc_noex": [],
             "best_val": [], "best_id": []}
        for ch in self.chains:
            la = getLastAccepted(ch)
            noex = [t for t in range(ch.iter) if ch.exchanged[t] == 0]
            # the accept rate as the reference would report it NOW (after the last exchange): set_acceptRate! is only called inside
            # doAcceptReject!, so the stored field is the one of the last accept step
            s["sigma"].append(ch.sigma); s["accept_rate"].append(ch.accept_rate)
            s["la_value"].append(la.value); s["la_prob"].append(la.prob); s["la_status"].append(la.status); s["la_params"].append(list(la.params))
            s["n_noex"].append(len(noex)); s["n_acc_noex"].append(sum(1 for t in noex if ch.accepted[t]))
            s["best_val"].append(ch.best_val[ch.iter - 1]); s["best_id"].append(ch.best_id[ch.iter - 1])
        return s
