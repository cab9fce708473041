"""Host-side mirror of the reference's API surface for the BGP path (src/SMM.jl:31-57).

The reference is Julia; no julia binary exists in the build image, so the host layer above the C
ABI is written in Python with the reference's names, argument meaning and error behaviour
(`addSampledParam!` -> `addSampledParam`, `run!` -> `run`, ...).  A Julia maintainer binds the
same C ABI with `ccall`.  Everything numerical happens in libsmmhip.so.

  MProb, addParam, addSampledParam, addMoment, addEvalFunc          mprob.jl:29-159
  Eval and its accessors                                            Eval.jl:20-238
  evaluateObjective                                                 mprob.jl:175-205
  objfunc_norm / banana: device objectives                          ObjExamples.jl:59-116, 251-265
  BGPChain (a view on the downloaded history), MAlgoBGP             AlgoBGP.jl:42-110, 497-539
  computeNextIteration, run, restart, history, summary, ...         AlgoBGP.jl:589-640, AlgoAbstract.jl:27-76
"""
import time as _time
from collections import OrderedDict

import numpy as np

from . import _abi as A
from .backend import BGPOpts, Problem, Tables, hip_context


# ------------------------------------------------------------------------------------------
# objectives: the reference stores a Julia function in MProb.objfunc (mprob.jl:159) and calls
# it with an Eval (mprob.jl:182).  On the GPU the objective is a device function selected by id.
# ------------------------------------------------------------------------------------------
class DeviceObjective:
    def __init__(self, name, objective_id, ns=10000, needs_square=False):
        self.name, self.objective_id, self.ns, self.needs_square = name, objective_id, ns, needs_square

    def __call__(self, ev, **opts):  # f(ev::Eval; kwargs...)::Eval, evaluated on the device
        return evaluateObjective(ev._mprob, ev) if getattr(ev, "_mprob", None) is not None else _no_mprob()

    def __repr__(self):
        return "<device objective %s>" % self.name


def _no_mprob():
    raise ValueError("a device objective needs an Eval built from an MProb (Eval(mprob, p))")


def user_objective(source, name="user_objective", n_sums=None, lanes=256):
    """A user-written device objective (MProb.objfunc of the reference, mprob.jl:159): HIP/C++ text that defines
    SMM_USER_OBJECTIVE(...) — or, with n_sums=k, the map-reduce pair SMM_USER_PARTIAL / SMM_USER_FINISH evaluated by
    `lanes` threads per chain — see include/smmhip.h.  addEvalFunc(m, user_objective(src));
    m.objfunc_opts["obj_params"] = [...] passes udata."""
    from .backend import register_user_objective
    return DeviceObjective(name, register_user_objective(source, n_sums, lanes), ns=1)


objfunc_norm = DeviceObjective("objfunc_norm", A.SMM_OBJ_NORM, needs_square=True)   # ObjExamples.jl:59-116
banana = DeviceObjective("banana", A.SMM_OBJ_BANANA, ns=1)                           # ObjExamples.jl:251-265
dense_sim = DeviceObjective("dense_sim", A.SMM_OBJ_DENSE, ns=1)                     # synthetic dense simulation (include/smmhip.h)
dense_sim2 = DeviceObjective("dense_sim2", A.SMM_OBJ_DENSE2, ns=1)                  # ... with the 256 x 256 stage: BASELINE config 5 as worded


class MProb:
    """mprob.jl:29-53"""

    def __init__(self):
        self.initial_value = OrderedDict()
        self.params_to_sample = OrderedDict()
        self.objfunc = None
        self.objfunc_opts = {}
        self.moments = OrderedDict()

    def __repr__(self):
        return "MProb: %d parameters to sample, %d moments, objective %r" % (
            len(self.params_to_sample), len(self.moments), self.objfunc)


def addParam(m, name_or_dict, init=None):
    """addParam!, mprob.jl:60-75"""
    if isinstance(name_or_dict, dict):
        for k, v in name_or_dict.items():
            m.initial_value[str(k)] = v
    else:
        m.initial_value[str(name_or_dict)] = init
    return m


def addSampledParam(m, name_or_dict, init=None, lb=None, ub=None):
    """addSampledParam!, mprob.jl:81-98: a name with (init, lb, ub), or a dict name -> [init, lb, ub]"""
    if isinstance(name_or_dict, dict):
        for k, v in name_or_dict.items():
            addSampledParam(m, k, v[0], v[1], v[2])
        return m
    if not ub > lb:
        raise AssertionError("ub>lb")  # @assert ub>lb, mprob.jl:82
    m.initial_value[str(name_or_dict)] = init
    m.params_to_sample[str(name_or_dict)] = {"lb": lb, "ub": ub}
    return m


def addMoment(m, name, value=None, weight=1.0):
    """addMoment!, mprob.jl:123-155: (name, value[, weight]), a dict name -> {value, weight}, or a
    table with columns name/value/weight (pandas DataFrame or dict of columns)."""
    if hasattr(name, "columns") or (isinstance(name, dict) and "name" in name and "value" in name):
        names, values = list(name["name"]), list(name["value"])
        weights = list(name["weight"]) if "weight" in name else [1.0] * len(names)
        for n, v, w in zip(names, values, weights):
            addMoment(m, n, v, w)
        return m
    if isinstance(name, dict):
        for k, d in name.items():
            addMoment(m, k, d["value"], d["weight"])
        return m
    m.moments[str(name)] = {"value": value, "weight": weight}
    return m


def addEvalFunc(m, f):
    """addEvalFunc!, mprob.jl:159"""
    m.objfunc = f
    return m


def ps_names(m):
    return list(m.initial_value.keys())


def ps2s_names(m):
    return list(m.params_to_sample.keys())


def ms_names(m):
    return list(m.moments.keys())


def _flat_problem(m):
    if not isinstance(m.objfunc, DeviceObjective):
        raise TypeError("MProb.objfunc must be a device objective (objfunc_norm, banana, dense_sim, or user_objective(src) "
                        "for your own): arbitrary host closures cannot run inside the GPU iteration")
    names = ps2s_names(m)
    init = [m.initial_value[k] for k in names]
    lb = [m.params_to_sample[k]["lb"] for k in names]
    ub = [m.params_to_sample[k]["ub"] for k in names]
    mom = [m.moments[k]["value"] for k in ms_names(m)]
    w = [np.nan if m.moments[k]["weight"] is None else m.moments[k]["weight"] for k in ms_names(m)]
    return Problem(init, lb, ub, mom, w, ns=m.objfunc_opts.get("ns", m.objfunc.ns), objective_id=m.objfunc.objective_id,
                   obj_params=m.objfunc_opts.get("obj_params"))


class Eval:
    """Eval.jl:20-155.  Constructors: Eval(), Eval(mprob), Eval(mprob, p), Eval(p, moments_table)."""

    def __init__(self, a=None, b=None):
        self.value = -1.0
        self.time = _time.time()
        self.status = -1
        self.params = OrderedDict()
        self.simMoments = OrderedDict()
        self.dataMoments = OrderedDict()
        self.dataMomentsW = OrderedDict()
        self.prob = 0.0
        self.accepted = False
        self.options = {}
        self._mprob = None
        if isinstance(a, MProb):
            self._mprob = a
            for k, d in a.moments.items():
                self.dataMoments[k] = d["value"]
                self.dataMomentsW[k] = d["weight"]
            p = a.initial_value if b is None else b  # Eval(mprob) uses the initial value, Eval.jl:108-130
            for k, v in p.items():
                self.params[str(k)] = v
        elif a is not None:  # Eval(p::Dict, mom::DataFrame), Eval.jl:48-80
            for col in ("name", "value", "weight"):
                if col not in b:
                    raise ValueError("moment dataframe needs column named `%s`" % col)
            for n, v, w in zip(b["name"], b["value"], b["weight"]):
                self.dataMoments[str(n)] = v
                self.dataMomentsW[str(n)] = w
            for k, v in a.items():
                self.params[str(k)] = v[0] if np.ndim(v) else v

    def __eq__(self, o):  # Eval.jl:157-166
        return (self.value == o.value and self.status == o.status and self.params == o.params and
                self.simMoments == o.simMoments and self.dataMoments == o.dataMoments and self.accepted == o.accepted)


def param(ev, which=None):
    if which is None:
        return np.array(list(ev.params.values()), float)
    if isinstance(which, (list, tuple)):
        return np.array([ev.params[k] for k in which], float)
    return ev.params[which]


def paramd(ev):
    return ev.params


def dataMoment(ev, which=None):
    if which is None:
        return np.array(list(ev.dataMoments.values()), float)
    if isinstance(which, (list, tuple)):
        return np.array([ev.dataMoments[k] for k in which], float)
    return ev.dataMoments[which]


def dataMomentd(ev):
    return ev.dataMoments


def dataMomentW(ev, which=None):
    if which is None:
        return np.array(list(ev.dataMomentsW.values()), float)
    if isinstance(which, (list, tuple)):
        return np.array([ev.dataMomentsW[k] for k in which], float)
    return ev.dataMomentsW[which]


def dataMomentWd(ev):
    return ev.dataMomentsW


def setValue(ev, value):
    ev.value = float(value)


def setMoments(ev, k, value=None):
    if isinstance(k, dict):
        for kk, v in k.items():
            ev.simMoments[str(kk)] = v
    else:
        ev.simMoments[str(k)] = value


def fill(p, ev):
    """fill(p, ev): copy the parameters onto the fields of a user object, Eval.jl:207-211"""
    for k, v in ev.params.items():
        setattr(p, k, v)


def _eval_context(m, prob):
    """the device context behind evaluateObjective, kept on the MProb and rebuilt when anything it holds a device copy
    of changes (dimensions, objective, data moments, weights, objective parameters; byte-wise, so NaN weights compare equal)"""
    op = b"" if prob.obj_params is None else prob.obj_params.tobytes()
    key = (prob.np, prob.nm, prob.ns, prob.objective_id, prob.mom.tobytes(), prob.w.tobytes(), op,
           prob.init.tobytes(), prob.lb.tobytes(), prob.ub.tobytes())
    cached = getattr(m, "_eval_ctx", None)
    if cached is None or cached[0] != key:
        if cached is not None:
            cached[1].close()
        opts = BGPOpts(N=1, maxiter=1, sigma=[0.05], acc_tuner=[1.0], min_improve=[0.0])
        cached = (key, hip_context(prob, opts))
        m._eval_ctx = cached
    return cached[1]


def evaluateObjective(m, p_or_ev):
    """evaluateObjective(m, p) / (m, ev), mprob.jl:175-205, as a batch of one on the device."""
    ev = p_or_ev if isinstance(p_or_ev, Eval) else Eval(m, p_or_ev)
    prob = _flat_problem(m)
    ctx = (None, _eval_context(m, prob))
    names = ps2s_names(m)
    theta = np.array([[ev.params[k]] for k in names], float)
    t0 = _time.time()
    v, sm, st = ctx[1].eval_batch(theta)
    ev.value, ev.status = float(v[0]), int(st[0])
    for k, x in zip(ms_names(m), sm[:, 0]):
        ev.simMoments[k] = float(x)
    ev.time = _time.time() - t0
    return ev


# ------------------------------------------------------------------------------------------
# BGPChain: AlgoBGP.jl:42-110.  The chain's per-iteration arrays are views on the downloaded
# structure-of-arrays history; evals[t] materialises an Eval on demand.
# ------------------------------------------------------------------------------------------
class _Evals:
    def __init__(self, chain):
        self._c = chain

    def __len__(self):
        return self._c._algo.opts["maxiter"]

    def __getitem__(self, t):
        c = self._c
        if isinstance(t, slice):
            return [self[i] for i in range(*t.indices(len(self)))]
        if isinstance(t, (np.ndarray, list)):
            idx = np.flatnonzero(t) if np.asarray(t).dtype == bool else t
            return [self[int(i)] for i in idx]
        h = c._h()
        if t < 0:
            t += h.value.shape[0]
        ev = Eval(c.m, OrderedDict((k, float(h.params[t, i, c._j])) for i, k in enumerate(ps2s_names(c.m))))
        ev.value = float(h.value[t, c._j]); ev.prob = float(h.prob[t, c._j])
        ev.accepted = bool(h.accepted[t, c._j]); ev.status = int(h.status[t, c._j])
        for i, k in enumerate(ms_names(c.m)):
            ev.simMoments[k] = float(h.sim_moments[t, i, c._j])
        return ev


class BGPChain:
    def __init__(self, algo, j):
        self._algo, self._j = algo, j
        self.id = j + 1
        self.m = algo.m
        self.evals = _Evals(self)

    def _h(self):
        return self._algo._history()

    def _col(self, f, fill, dtype):
        h, n = self._h(), self._algo.opts["maxiter"]
        out = np.full(n, fill, dtype)
        out[: h.value.shape[0]] = getattr(h, f)[:, self._j]
        return out

    iter = property(lambda s: s._algo.i)
    accepted = property(lambda s: s._col("accepted", 0, np.uint8).astype(bool))
    exchanged = property(lambda s: s._col("exchanged", 0, np.int64))
    best_val = property(lambda s: s._col("best_val", np.inf, float))
    curr_val = property(lambda s: s._col("curr_val", np.inf, float))
    best_id = property(lambda s: s._col("best_id", -1, np.int64))
    sigma = property(lambda s: float(s._algo._state().sigma[s._j]))
    accept_rate = property(lambda s: float(s._algo._state().accept_rate[s._j]))
    acc_tuner = property(lambda s: float(s._algo._acc_tuner[s._j]))
    min_improve = property(lambda s: float(s._algo._min_improve[s._j]))
    sigma_update_steps = property(lambda s: s._algo._flat["sigma_update_steps"])
    sigma_adjust_by = property(lambda s: s._algo._flat["sigma_adjust_by"])
    smpl_iters = property(lambda s: s._algo._flat["smpl_iters"])


def allAccepted(c):
    """AlgoBGP.jl:117"""
    return c.evals[c.accepted[: c.iter]]


def params(c, accepted_only=True):
    """AlgoBGP.jl:120-131: dict name -> vector of parameter values"""
    h = c._h()
    sel = h.accepted[:, c._j].astype(bool) if accepted_only else np.ones(h.value.shape[0], bool)
    return {k: h.params[sel, i, c._j].copy() for i, k in enumerate(ps2s_names(c.m))}


def history(c):
    """history(c::BGPChain), AlgoBGP.jl:138-160: columns iter, value, accepted, curr_val, best_val, prob,
    exchanged, <params...> (a pandas DataFrame when pandas is importable, else a dict of columns)."""
    h = c._h()
    n = h.value.shape[0]
    cols = OrderedDict()
    cols["iter"] = np.arange(1, n + 1)
    cols["value"] = h.value[:, c._j].copy()
    cols["accepted"] = h.accepted[:, c._j].astype(bool)
    cols["curr_val"] = h.curr_val[:, c._j].copy()
    cols["best_val"] = h.best_val[:, c._j].copy()
    cols["prob"] = h.prob[:, c._j].copy()
    cols["exchanged"] = h.exchanged[:, c._j].astype(np.int64)
    for i, k in enumerate(ps2s_names(c.m)):
        cols[k] = h.params[:, i, c._j].copy()
    try:
        import pandas as pd
        return pd.DataFrame(cols)
    except Exception:  # pragma: no cover
        return cols


def best(c):
    """best(c) -> (val, idx), AlgoBGP.jl:167 (1-based index like findmin)"""
    v = c._h().value[:, c._j]
    i = int(np.argmin(v))
    return float(v[i]), i + 1


def mean(c):
    return {k: float(np.mean(v)) for k, v in params(c).items()}


def median(c):
    return {k: float(np.median(v)) for k, v in params(c).items()}


def CI(c, level=0.95):
    return {k: np.quantile(v, [(1 - level) / 2, 1 - (1 - level) / 2]) for k, v in params(c).items()}


def summary(x):
    """summary(c::BGPChain) AlgoBGP.jl:197-206 / summary(m::MAlgoBGP) :541-550"""
    if isinstance(x, MAlgoBGP):
        rows = [summary(c) for c in x.chains]
        try:
            import pandas as pd
            return pd.DataFrame(rows)
        except Exception:  # pragma: no cover
            return rows
    ex = x.exchanged
    ex_with = ex[ex != 0]
    most = int(np.bincount(ex_with).argmax()) if len(ex_with) else 0  # mode(ex_with)
    return OrderedDict(id=x.id, acc_rate=x.accept_rate, perc_exchanged=100.0 * np.sum(ex != 0) / len(ex),
                       exchanged_most_with=most, best_val=float(x.best_val[-1]))


# ------------------------------------------------------------------------------------------
# MAlgoBGP: AlgoBGP.jl:497-539
# ------------------------------------------------------------------------------------------
_DEFAULT_OPTS = {"N": 3, "maxiter": 100, "maxtemp": 2, "sigma": 0.05, "sigma_update_steps": 10, "sigma_adjust_by": 0.01,
                 "smpl_iters": 1000, "parallel": False, "min_improve": [0.0] * 3, "acc_tuners": [2.0] * 3}
_IGNORED_OPTS = ("coverage", "mixprob", "acc_tuner", "maxdists")  # read by nothing in the reference either


def _dist_fun_id(f):
    """opts["dist_fun"] (AlgoBGP.jl:494,537: any Julia function of two objective values, default `-`) -> smm_dist_fun_t.
    The device offers a menu: "-" / operator.sub (default), "absdiff" (|a - b|), "reldiff" ((a - b) / |a|), or the ids
    themselves; an arbitrary host callable cannot run inside the exchange kernels."""
    import operator
    if f is None or f is operator.sub or f in ("-", "minus", "sub", A.SMM_DIST_MINUS):
        return A.SMM_DIST_MINUS
    if f in ("absdiff", "abs", A.SMM_DIST_ABSDIFF):
        return A.SMM_DIST_ABSDIFF
    if f in ("reldiff", "relative", A.SMM_DIST_RELDIFF):
        return A.SMM_DIST_RELDIFF
    raise NotImplementedError("dist_fun: the device runs '-' (AlgoBGP.jl:537), 'absdiff' or 'reldiff' (smm_dist_fun_t, include/smmhip.h), "
                              "not an arbitrary host function")


class MAlgoBGP:
    def __init__(self, m, opts=None, tables=None):
        opts = dict(_DEFAULT_OPTS) if opts is None else opts
        self.m, self.opts, self.i = m, opts, 0
        N = int(opts["N"])
        if N > 1:
            temps = np.linspace(1.0, float(opts["maxtemp"]), N)  # range(1.0, stop=maxtemp, length=N), :508
        else:
            temps = np.ones(1)
        sigma = opts.get("sigma", 0.05) * temps                                   # :518
        self._min_improve = np.asarray(opts.get("min_improve", [0.5] * N), float)[:N]   # :522
        self._acc_tuner = np.asarray(opts.get("acc_tuners", [2.0] * N), float)[:N]      # :523
        if len(self._min_improve) < N or len(self._acc_tuner) < N:
            raise IndexError("min_improve / acc_tuners need one entry per chain (AlgoBGP.jl:522-523)")
        self._dist_fun = _dist_fun_id(opts.get("dist_fun", None))                 # :537
        prob = _flat_problem(m)
        self._flat = dict(sigma_update_steps=int(opts.get("sigma_update_steps", 10)),
                          sigma_adjust_by=float(opts.get("sigma_adjust_by", 0.01)),
                          smpl_iters=int(opts.get("smpl_iters", 1000)))
        bo = BGPOpts(N=N, maxiter=int(opts["maxiter"]), sigma=sigma, acc_tuner=self._acc_tuner,
                     min_improve=self._min_improve, batch_size=opts.get("batch_size", None),
                     seed=int(opts.get("seed", 12)), device=int(opts.get("device", 0)),
                     dist_fun=self._dist_fun,
                     chol_L=opts.get("chol_L", None),   # general Gaussian proposals (not in the reference: include/smmhip.h)
                     **self._flat)
        self._prob, self._bopts, self._tables = prob, bo, tables
        self._ctx = hip_context(prob, bo, tables)
        self._hist = None
        self._st = None
        self.chains = [BGPChain(self, j) for j in range(N)]
        self.dist_fun = lambda a, b: a - b

    def __getitem__(self, key):  # algo["N"], AlgoAbstract.jl:13-19
        return self.opts[key]

    def _invalidate(self):
        self._hist = None
        self._st = None

    def _history(self):
        if self._hist is None:
            self._hist = self._ctx.history(0, self.i)
        return self._hist

    def _state(self):
        if self._st is None:
            self._st = self._ctx.state()
        return self._st


def computeNextIteration(algo):
    """computeNextIteration!(algo::MAlgoBGP), AlgoBGP.jl:589-640: one iteration of all chains + exchangeMoves!.
    The reference's run! sets algo.i = i before the call (AlgoAbstract.jl:38-45); here algo.i is set to the iteration the
    device has completed, so both `algo.i = i; computeNextIteration(algo)` and a bare call keep chains/history in step."""
    algo._ctx.step(1)
    algo._invalidate()
    algo.i = algo._ctx.state().iter


def run(algo):
    """run!(algo), AlgoAbstract.jl:27-76.  Without per-iteration hooks all remaining iterations are
    enqueued in one call; with opts["save_frequency"] the loop is cut at the save points."""
    t0 = _time.time()
    maxiter = int(algo["maxiter"])
    sf, fn = algo.opts.get("save_frequency"), algo.opts.get("filename")
    while algo.i < maxiter:
        n = maxiter - algo.i
        if sf and fn:
            n = min(n, sf - (algo.i % sf))
        algo._ctx.step(n)
        algo.i += n
        algo._invalidate()
        if sf and fn and algo.i % sf == 0:
            save(algo, fn)
    algo.opts["time"] = round((_time.time() - t0) / 60.0, 1)
    if fn:
        save(algo, fn)
    return algo


def save(algo, filename):
    """save(algo, filename), AlgoAbstract.jl:83-88 (JLD2 there; a self-describing .npz here)"""
    h, s = algo._ctx.history(0, algo.i), algo._ctx.state()
    d = {"i": algo.i}
    d.update({"h_" + f: getattr(h, f) for f in A.HistoryBuffers.FIELDS})
    d.update({"s_" + f: getattr(s, f) for f in A.StateBuffers.FIELDS})
    np.savez(filename if filename.endswith(".npz") else filename + ".npz", **d)


def readMalgo(algo, filename):
    """readMalgo, AlgoAbstract.jl:95-102: restore a saved run into an MAlgoBGP built from the same MProb/opts"""
    z = np.load(filename if filename.endswith(".npz") else filename + ".npz")
    i = int(z["i"])
    hb = A.HistoryBuffers(i, algo._ctx.N, algo._ctx.np, algo._ctx.nm)
    sb = A.StateBuffers(algo._ctx.N, algo._ctx.np, algo._ctx.nm)
    for f in A.HistoryBuffers.FIELDS:
        getattr(hb, f)[...] = z["h_" + f]
    for f in A.StateBuffers.FIELDS:
        getattr(sb, f)[...] = z["s_" + f]
    sb.iter = i
    algo._ctx.set_state(sb, hb)
    algo.i = i
    algo._invalidate()
    return algo


def restart(algo, extra_iter):
    """restart!(algo, extraIter), AlgoBGP.jl:804-884, with clean resume semantics (continue at i+1): the
    history capacity is extended by building a new device context and uploading the saved state."""
    new_maxiter = int(algo.opts["maxiter"]) + int(extra_iter)
    tb = algo._tables
    if tb is not None and tb.covers_iterations() is not None and tb.covers_iterations() < new_maxiter:
        # injected per-iteration randomness (parity runs) was made for the old maxiter: the extended run would read past it
        raise ValueError("restart: the injected randomness tables cover %d iterations, the extended run needs %d — "
                         "build the MAlgoBGP with tables for the whole run, or without tables" % (tb.covers_iterations(), new_maxiter))
    h, s = algo._ctx.history(0, algo.i), algo._ctx.state()
    algo.opts["maxiter"] = new_maxiter
    bo = algo._bopts
    bo.maxiter = new_maxiter
    algo._ctx.close()
    algo._ctx = hip_context(algo._prob, bo, tb)
    s.iter = algo.i
    algo._ctx.set_state(s, h)
    algo._invalidate()
    return run(algo)


# ------------------------------------------------------------------------------------------
# Example drivers: Examples.jl:118-153 (serialNormal), :373-446 (snorm_impl)
# ------------------------------------------------------------------------------------------
def snorm_impl(opts, niter=200, npar=2):
    """snorm_impl(opts, niter; npar), Examples.jl:373-416.  For npar > 2 the reference draws the extra parameters' bounds,
    start values and target moments from Julia's global generator after Random.seed!(12) (:390-405); that stream cannot be
    reproduced here, so the same construction draws from numpy's default_rng(12) (same ranges, different numbers)."""
    pb = OrderedDict()
    pb["p1"] = [0.2, -3, 3]
    pb["p2"] = [-0.2, -20, 20]
    moms = {"name": ["mu1", "mu2"], "value": [-1.0, 10.0], "weight": [1.0, 1.0]}
    if npar > 2:
        rng = np.random.default_rng(12)

        def map_range(a1, a2, b1, b2, x):
            return b1 + (x - a1) * (b2 - b1) / (a2 - a1)
        spaces = np.concatenate([rng.random(2), [4.0 ** -4], (np.linspace(0.25, 0.45, npar - 1) ** -4.0)[::-1]])   # :391
        for j in range(3, npar + 1):
            sp = float(spaces[j - 1])
            pb["p%d" % j] = [map_range(0, 1, -sp, sp, rng.random()), -sp, sp]
            y = map_range(0, 1, -sp, sp, rng.random())
            moms["name"].append("mu%d" % j); moms["value"].append(y); moms["weight"].append(y)
    mprob = MProb()
    addSampledParam(mprob, pb)
    addMoment(mprob, moms)
    addEvalFunc(mprob, objfunc_norm)
    MA = MAlgoBGP(mprob, opts)
    run(MA)
    return MA


def serialNormal(npars=2, niter=200):
    """SMM.serialNormal(npars, niter), Examples.jl:118-153"""
    nchains = 3
    opts = {"N": nchains, "maxiter": niter, "maxtemp": 5, "coverage": 0.02, "smpl_iters": 1000, "parallel": False,
            "min_improve": [0.0] * nchains, "acc_tuners": [20.0, 2.0, 1.0], "animate": False}
    return snorm_impl(opts, niter, npar=npars)
