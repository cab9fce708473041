"""oracle/literal_bgp.py (objects, arrays per chain, findlast / mean over the history / deepcopy: the reference's own shape) against
oracle/smm_oracle.c (structure of arrays with running counters: the shape the kernels share) — bit for bit, on the injected tables of
the golden fixtures and on random small cases.  Both are restatements written for this build (parity with the reference stays unpinned,
DESIGN.md 1c); what the comparison removes is the risk that a shortcut the C oracle and the kernels SHARE hides a misreading."""
import os
import sys

import numpy as np
import pytest

import common as cm
import smm_jl_amd as S
from smm_jl_amd import _abi as A
from test_golden import CASES, load, tables_of

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import literal_bgp as L  # noqa: E402


def literal_of(prob, opts, tab):
    assert opts.N == opts.N_global and opts.chol_L is None
    m = {"init": prob.init.tolist(), "lb": prob.lb.tolist(), "ub": prob.ub.tolist(), "mom": prob.mom.tolist(), "w": prob.w.tolist(),
         "ns": prob.ns, "Z": tab.Z.tolist(), "objective_id": prob.objective_id,
         "objp": None if prob.obj_params is None else prob.obj_params.tolist(), "tries": tab.prop_normals.shape[1]}
    bs = prob.np if opts.batch_size is None else opts.batch_size
    return L.MAlgoBGP(m, opts.N, opts.maxiter, opts.sigma.tolist(), opts.acc_tuner.tolist(), opts.min_improve.tolist(),
                      opts.sigma_update_steps, opts.sigma_adjust_by, opts.smpl_iters, bs, tab.probs_acc.tolist(),
                      tab.prop_normals.tolist(), tab.pairs.tolist() if tab.pairs is not None else [[] for _ in range(opts.maxiter)],
                      dist_fun=opts.dist_fun, exchange_from_iter=opts.exchange_from_iter)


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and bool(np.all((a == b) | ((a != a) & (b != b))))


def assert_equal_to_c_oracle(lit, h, s, T):
    """h, s: history and state in the ABI's layout ([t][N], params [t][np][N]) from the C oracle or a golden file"""
    lh, ls = lit.history(), lit.state()
    for f in ("value", "prob", "curr_val", "best_val", "best_id", "exchanged", "accepted", "status"):
        assert same(np.array(lh[f]), h[f][:T]), "history." + f
    assert same(np.transpose(np.array(lh["params"]), (0, 2, 1)), h["params"][:T]), "history.params"
    assert same(np.transpose(np.array(lh["sim_moments"]), (0, 2, 1)), h["sim_moments"][:T]), "history.sim_moments"
    for f in ("sigma", "accept_rate", "la_value", "la_status", "n_noex", "n_acc_noex", "best_val", "best_id"):
        assert same(np.array(ls[f]), s[f]), "state." + f
    assert same(np.array(ls["la_params"]).T, s["la_params"]), "state.la_params"


@pytest.mark.parametrize("name", ["g2_c1_trajectory", "g3_exchange_order", "g4a_failbox", "g4b_sigma_batches"])
def test_literal_restatement_reproduces_the_golden_runs(name):
    z = load(name)
    prob, opts, T = CASES[name]()
    lit = literal_of(prob, opts, tables_of(z))
    lit.run(T)
    h = {f: z["h_" + f] for f in A.HistoryBuffers.FIELDS}
    s = {f: z["s_" + f] for f in A.StateBuffers.FIELDS}
    assert_equal_to_c_oracle(lit, h, s, T)
    assert (z["h_exchanged"] != 0).any()


def fuzz_case(i):
    rng = np.random.default_rng(1000 + i)
    npar = int(rng.choice([1, 2, 2, 3, 4, 6]))
    N = int(rng.integers(1, 13))
    T = int(rng.integers(5, 41))
    ns = int(rng.choice([1, 7, 64, 200, 513, 700]))
    divs = [d for d in range(1, npar + 1) if npar % d == 0]
    bs = int(rng.choice(divs))
    prob, opts = cm.general_normal(npar, N, T, ns=ns, seed=int(rng.integers(1, 1 << 30)), batch_size=bs,
                                   sigma_update_steps=int(rng.choice([1, 2, 3, 5, 10])), sigma_adjust_by=float(rng.choice([0.01, 0.05, 0.2])),
                                   dist_fun=int(rng.choice([0, 0, 0, 1, 2])))
    kind = rng.integers(0, 5)
    if kind == 0:
        opts.min_improve[:] = float(rng.choice([0.05, 0.5, -0.1]))
    elif kind == 1:
        opts.min_improve[:] = rng.uniform(-0.2, 0.6, N)
    if rng.random() < 0.25:
        prob.w[rng.integers(0, npar)] = np.nan           # a moment without a weight (ObjExamples.jl:96-97)
    if rng.random() < 0.25 and npar >= 1:
        prob.objective_id = A.SMM_OBJ_NORM_FAILBOX
        c0 = float(prob.init[0])
        prob.obj_params = A.f64([c0 + 0.02, c0 + 0.4])   # a box next to the start value: some proposals "throw"
    if rng.random() < 0.15:
        opts.sigma[:] = opts.sigma * 3.0                  # wide steps: redraws, now and then no draw in support
    tab = cm.random_tables(prob, opts, tries=int(rng.choice([6, 24, 24, 24])), seed=int(rng.integers(1, 1 << 30)))
    return prob, opts, tab, T


@pytest.mark.parametrize("i", range(50))
def test_literal_restatement_equals_the_c_oracle_on_random_cases(O, i):
    prob, opts, tab, T = fuzz_case(i)
    o = O.OracleContext(prob, opts, tab)
    lit = literal_of(prob, opts, tab)
    rc, done = 0, 0
    try:
        o.step(T)
        done = T
    except A.SMMHipError as e:
        rc, msg = e.code, str(e)
    if rc == 0:
        lit.run(T)
    else:   # the reference aborts run! with the same error (AlgoBGP.jl:341,409): the literal form must raise it in the same iteration
        want = {A.SMM_ERR_NO_DRAW_IN_SUPPORT: L.NoDrawInSupport, A.SMM_ERR_NEGATIVE_OBJECTIVE: L.NegativeObjective}[rc]
        with pytest.raises(want):
            lit.run(T)
        import re
        it = int(re.search(r"iter (\d+)", msg).group(1))
        assert lit.i == it, (lit.i, msg)     # ... in the same iteration
        return
    h, s = o.history(), o.state()
    assert_equal_to_c_oracle(lit, {f: getattr(h, f) for f in A.HistoryBuffers.FIELDS}, {f: getattr(s, f) for f in A.StateBuffers.FIELDS}, done)


def test_the_random_cases_cover_what_they_should(O):
    """the fuzz mix is only worth something if it reaches the branches: exchanges, failed objectives, per-chain and negative
    thresholds, the other distance functions, weight-less moments, and runs that stop with `no draw in support`"""
    st = dict(full=0, exch=0, fail2=0, err=0, mi=0, dist=0, nanw=0, batches=0)
    for i in range(50):
        prob, opts, tab, T = fuzz_case(i)
        o = O.OracleContext(prob, opts, tab)
        try:
            o.step(T)
            h = o.history()
            st["full"] += 1; st["exch"] += int((h.exchanged != 0).any()); st["fail2"] += int((h.status == -2).any())
        except A.SMMHipError:
            st["err"] += 1
        st["mi"] += int((opts.min_improve != 0).any()); st["dist"] += int(opts.dist_fun != 0); st["nanw"] += int(np.isnan(prob.w).any())
        st["batches"] += int(opts.batch_size not in (None, prob.np))
    assert st["full"] >= 35 and st["exch"] >= 22 and st["fail2"] >= 4 and st["err"] >= 3 and st["mi"] >= 10 and st["dist"] >= 10 \
        and st["nanw"] >= 5 and st["batches"] >= 8, st
