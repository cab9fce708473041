"""The other callers of evaluateObjective, driven through the batched device evaluation:
slices of the objective (slices.jl: Slice, doSlices, optSlices) and the sandwich standard errors (econometrics.jl:
FD_gradient, getSigma, get_stdErrors).  Where the reference maps evaluateObjective over a grid of parameter vectors (pmap or
map), here the whole grid is ONE call of smm_eval_batch (include/smmhip.h); the arithmetic on the results is host-side numpy.

`evaluator(m, P [np][M], noseed_base=None) -> (value [M], simM [nm][M], status [M])` can be injected (the tests run the
same drivers over a CPU evaluator); the default is the device.
"""
import time as _time
from collections import OrderedDict

import numpy as np

from .host import Eval, _eval_context, _flat_problem, ms_names, ps2s_names


def device_evaluator(m, P, noseed_base=None):
    ctx = _eval_context(m, _flat_problem(m))
    return ctx.eval_batch(P) if noseed_base is None else ctx.eval_batch_noseed(P, noseed_base)


def evaluateObjectives(m, plist, evaluator=None, noseed_base=None):
    """evaluateObjective(m, p) (mprob.jl:175-205) for a list of parameter dicts, as one batch; returns the Evals"""
    evaluator = device_evaluator if evaluator is None else evaluator
    names = ps2s_names(m)
    evs = [p if isinstance(p, Eval) else Eval(m, p) for p in plist]
    if not evs:
        return evs
    P = np.array([[ev.params[k] for ev in evs] for k in names], float)
    t0 = _time.time()
    v, sm, st = evaluator(m, P, noseed_base)
    dt = (_time.time() - t0) / len(evs)
    for i, ev in enumerate(evs):
        ev.value, ev.status, ev.time = float(v[i]), int(st[i]), dt
        for k, x in zip(ms_names(m), sm[:, i]):
            ev.simMoments[k] = float(x)
    return evs


def range_length(m):
    """mprob.jl:212-218"""
    return OrderedDict((k, v["ub"] - v["lb"]) for k, v in m.params_to_sample.items())


# ------------------------------------------------------------------------------------------
# slices.jl
# ------------------------------------------------------------------------------------------
class Slice:
    """slices.jl:26-39: res[p][value of p] = {"moments": simMoments, "value": value}; p0, m0 = initial parameters, data moments"""

    def __init__(self, p, m):
        self.res = OrderedDict((k, OrderedDict()) for k in p)
        self.p0 = OrderedDict(p)
        self.m0 = OrderedDict(m)

    def add(self, p, ev):   # add!, slices.jl:41-43
        self.res[p][ev.params[p]] = {"moments": OrderedDict(ev.simMoments), "value": ev.value}

    def get(self, p, m):    # get, slices.jl:45-60: sorted by the parameter's value
        x = np.array(list(self.res[p].keys()), float)
        y = np.array([v["value"] if m == "value" else v["moments"][m] for v in self.res[p].values()], float)
        ix = np.argsort(x, kind="stable")
        return {"x": x[ix], "y": y[ix]}


def doSlices(m, npoints, parallel=False, evaluator=None):
    """doSlices(m, npoints), slices.jl:250-290: for every sampled parameter a grid of npoints values over its bounds, the
    others at their initial values.  All np * npoints evaluations are one batch.  (`parallel` is accepted and ignored.)"""
    res = Slice(m.initial_value, m.moments)
    plist, tags = [], []
    for pp, bb in m.params_to_sample.items():
        for pval in np.linspace(bb["lb"], bb["ub"], npoints):
            p = OrderedDict(m.initial_value)
            p[pp] = float(pval)
            plist.append(p); tags.append(pp)
    for pp, ev in zip(tags, evaluateObjectives(m, plist, evaluator)):
        res.add(pp, ev)   # (a failed evaluation is stored with its status's value, as the reference stores what comes back)
    return res


def optSlices(m, npoints, parallel=False, tol=1e-5, update=None, filename=None, evaluator=None, maxiter=1000):
    """optSlices, slices.jl:114-242: cyclic coordinate search on grids.  Per cycle and parameter: a grid of npoints values over
    the parameter's current range, the others at the best point so far; the grid point with the smallest finite value becomes
    the best point (strict <, first wins); after a cycle the ranges shrink around the best point by the factor `update` (if
    given) and the cycle's moves dvec decide convergence (norm <= tol).  Returns {"best": {"p", "value"}, "history": rows,
    "iterations": n, "converged": bool}.  (The reference's trace file is JLD2; `filename`, if given, gets the same dict as .npz-friendly JSON.)
    `maxiter` bounds the cycles (the reference has no bound)."""
    ranges = OrderedDict((k, dict(v)) for k, v in m.params_to_sample.items())
    bestp = OrderedDict(m.initial_value)
    dvec = OrderedDict((k, np.inf) for k in ranges)   # (the sampled parameters only: a fixed one never moves and would keep the norm at inf)
    dout = {"history": []}
    delta, it = np.inf, 0
    while delta > tol and it < maxiter:
        it += 1
        for pp, bb in ranges.items():
            cur_param = OrderedDict(bestp)
            plist = []
            for pval in np.linspace(bb["lb"], bb["ub"], npoints):
                p = OrderedDict(cur_param)
                p[pp] = float(pval)
                plist.append(p)
            vv = evaluateObjectives(m, plist, evaluator)
            minv = np.inf
            bestp = OrderedDict(cur_param)
            for iv, ev in enumerate(vv):
                dout["history"].append({"iter": it, "param": pp, "val_idx": iv + 1, "p": OrderedDict(ev.params), "value": ev.value})
                if np.isfinite(ev.value) and ev.value < minv:
                    minv = ev.value
                    bestp = OrderedDict(ev.params)
                    dout["best"] = {"p": OrderedDict(ev.params), "value": ev.value}
            dvec[pp] = cur_param[pp] - bestp[pp]
        if update is not None:   # shrink the search ranges around the best point, inside the current ones (:222-229)
            for k, v in bestp.items():
                if k in ranges:
                    r = (ranges[k]["ub"] - ranges[k]["lb"]) / 2
                    ranges[k]["lb"] = max(v - update * r, ranges[k]["lb"])
                    ranges[k]["ub"] = min(v + update * r, ranges[k]["ub"])
        delta = float(np.linalg.norm(list(dvec.values())))
    dout["iterations"] = it
    dout["converged"] = bool(delta <= tol)   # (False: stopped by maxiter)
    if filename:
        import json
        with open(filename, "w") as f:
            json.dump(dout, f)
    return dout


# ------------------------------------------------------------------------------------------
# econometrics.jl
# ------------------------------------------------------------------------------------------
def FD_gradient(m, p, step_perc=0.01, diff_method="forward", use_range=True, evaluator=None):
    """FD_gradient, econometrics.jl:29-85: finite-difference Jacobian of the simulated moments, (k, n) = (parameters of p,
    moments).  Step h_k = step_perc * (range of parameter k) or step_perc * p_k.  forward: (g(p + h e_k) - g(p)) / h; central:
    (g(p + h/2 e_k) - g(p - h/2 e_k)) / h.  All 1 + k (or 2k) evaluations are one batch.  Rows are in the order of `p` (the
    reference collects them through an unordered Dict)."""
    if diff_method not in ("forward", "central"):
        raise ValueError("only central and forward implemented")   # :71
    rs = range_length(m)
    keys = list(p.keys())
    hs = [(rs[k] if use_range else p[k]) * step_perc for k in keys]
    plist = [OrderedDict(p)]
    for k, h in zip(keys, hs):
        if diff_method == "forward":
            q = OrderedDict(p); q[k] = p[k] + h
            plist.append(q)
        else:
            q = OrderedDict(p); q[k] = p[k] + 0.5 * h
            plist.append(q)
            q = OrderedDict(p); q[k] = p[k] - 0.5 * h
            plist.append(q)
    evs = evaluateObjectives(m, plist, evaluator)
    mn = ms_names(m)
    g = lambda ev: np.array([ev.simMoments[k] for k in mn])
    gp = g(evs[0])
    D = np.zeros((len(keys), len(mn)))
    for i, h in enumerate(hs):
        D[i] = (g(evs[1 + i]) - gp) / h if diff_method == "forward" else (g(evs[1 + 2 * i]) - g(evs[2 + 2 * i])) / h
    return D


def getSigma(m, p, reps, seed=0, evaluator=None):
    """getSigma, econometrics.jl:125-145: covariance matrix of the simulated moments over `reps` evaluations at p, each with its
    own shock sequence (options[:noseed] = true: smm_eval_batch_noseed, evaluation i keyed by seed + i; the reference leaves
    the generator unseeded).  Sample covariance (divisor reps - 1), like Statistics.cov."""
    evs = evaluateObjectives(m, [OrderedDict(p) for _ in range(reps)], evaluator, noseed_base=seed)
    mn = ms_names(m)
    X = np.array([[ev.simMoments[k] for k in mn] for ev in evs])
    return np.atleast_2d(np.cov(X, rowvar=False, ddof=1))


def get_stdErrors(m, p, reps=300, seed=0, evaluator=None):
    """get_stdErrors, econometrics.jl:101-118: sqrt(diag(S)), S = (J W J')^+ (J W Sigma W J') (J W J')^+, J = FD_gradient,
    Sigma = getSigma, W = diag(moment weights)."""
    Sigma = getSigma(m, p, reps, seed=seed, evaluator=evaluator)
    J = FD_gradient(m, p, evaluator=evaluator)
    W = np.diag([m.moments[k]["weight"] for k in ms_names(m)])
    A = np.linalg.pinv(J @ W @ J.T)
    SE = A @ (J @ W @ Sigma @ W @ J.T) @ A
    return OrderedDict(zip(p.keys(), np.sqrt(np.diag(SE))))
