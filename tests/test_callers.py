"""The other callers of evaluateObjective (smm.jl_amd/callers.py: slices.jl, econometrics.jl of the reference) — their host
logic on CPU with an injected evaluator (an analytic objective), and (-m gpu) the same drivers on the device against the oracle."""
from collections import OrderedDict

import numpy as np
import pytest

import common as cm  # noqa: F401  (path set-up)
import smm_jl_amd as S


def toy_problem():
    m = S.MProb()
    S.addSampledParam(m, OrderedDict([("a", [0.5, -2.0, 2.0]), ("b", [1.0, 0.0, 4.0])]))
    S.addMoment(m, {"name": ["m1", "m2", "m3"], "value": [1.0, 3.0, 0.0], "weight": [1.0, 2.0, 0.5]})
    S.addEvalFunc(m, S.objfunc_norm)   # (never called: the evaluator below stands in for the device)
    return m


A = np.array([[2.0, 0.0], [0.5, 1.0], [-1.0, 3.0]])   # moments = A theta + noise


def linear_evaluator(m, P, noseed_base=None):
    M = P.shape[1]
    sm = A @ P
    if noseed_base is not None:   # evaluation i draws its own shocks, keyed by base + i
        for i in range(M):
            sm[:, i] += np.random.default_rng(noseed_base + i).standard_normal(3) * np.array([1.0, 0.1, 2.0])
    mom = np.array([1.0, 3.0, 0.0])[:, None]
    v = ((sm - mom) ** 2).mean(axis=0)
    return v, sm, np.ones(M, np.int8)


def test_doSlices_grid_and_get():
    m = toy_problem()
    s = S.doSlices(m, 5, evaluator=linear_evaluator)
    assert list(s.res.keys()) == ["a", "b"] and s.p0 == m.initial_value
    d = s.get("a", "value")
    assert np.array_equal(d["x"], np.linspace(-2.0, 2.0, 5))            # slices.jl:259: range(lb, stop = ub, length = npoints)
    for x, y in zip(d["x"], d["y"]):                                     # the other parameter at its initial value
        sm = A @ np.array([x, 1.0])
        assert y == ((sm - np.array([1.0, 3.0, 0.0])) ** 2).mean()
    d2 = s.get("b", "m3")
    assert np.array_equal(d2["x"], np.linspace(0.0, 4.0, 5)) and np.allclose(d2["y"], -0.5 + 3.0 * d2["x"])


def test_optSlices_converges_and_shrinks_ranges():
    m = toy_problem()
    out = S.optSlices(m, 41, tol=1e-3, update=0.5, evaluator=linear_evaluator)
    theta = np.linalg.lstsq(A, np.array([1.0, 3.0, 0.0]), rcond=None)[0]   # the minimiser of the toy objective
    best = np.array([out["best"]["p"]["a"], out["best"]["p"]["b"]])
    assert np.allclose(best, theta, atol=0.02) and out["iterations"] >= 2
    assert m.params_to_sample["a"]["lb"] == -2.0                             # the problem's own bounds are not touched
    rows = out["history"]
    assert rows[0]["iter"] == 1 and rows[0]["param"] == "a" and rows[0]["val_idx"] == 1 and len(rows) == 41 * 2 * out["iterations"]
    # a cycle's grid for b runs at the best a of that cycle (cyclic coordinate descent, slices.jl:140-148)
    a_best_1 = min((r for r in rows if r["iter"] == 1 and r["param"] == "a"), key=lambda r: r["value"])["p"]["a"]
    assert all(r["p"]["a"] == a_best_1 for r in rows if r["iter"] == 1 and r["param"] == "b")


@pytest.mark.parametrize("method", ["forward", "central"])
def test_FD_gradient_of_a_linear_moment_function(method):
    m = toy_problem()
    p = OrderedDict([("a", 0.3), ("b", 2.0)])
    D = S.FD_gradient(m, p, diff_method=method, evaluator=linear_evaluator)
    assert D.shape == (2, 3) and np.allclose(D, A.T, rtol=1e-9)              # (k, n): row k = d moments / d p_k
    D2 = S.FD_gradient(m, p, step_perc=0.05, use_range=False, diff_method=method, evaluator=linear_evaluator)
    assert np.allclose(D2, A.T, rtol=1e-9)
    with pytest.raises(ValueError):
        S.FD_gradient(m, p, diff_method="backward", evaluator=linear_evaluator)


def test_getSigma_and_sandwich_standard_errors():
    m = toy_problem()
    p = OrderedDict([("a", 0.3), ("b", 2.0)])
    Sig = S.getSigma(m, p, 200, seed=11, evaluator=linear_evaluator)
    X = np.array([A @ np.array([0.3, 2.0]) + np.random.default_rng(11 + i).standard_normal(3) * np.array([1.0, 0.1, 2.0]) for i in range(200)])
    assert np.allclose(Sig, np.cov(X, rowvar=False, ddof=1), rtol=1e-12)
    se = S.get_stdErrors(m, p, reps=200, seed=11, evaluator=linear_evaluator)
    J, W = A.T, np.diag([1.0, 2.0, 0.5])
    B = np.linalg.pinv(J @ W @ J.T)
    ref = np.sqrt(np.diag(B @ (J @ W @ Sig @ W @ J.T) @ B))
    assert list(se.keys()) == ["a", "b"] and np.allclose(list(se.values()), ref, rtol=1e-6)


@pytest.mark.gpu
def test_callers_on_the_device_match_the_oracle():
    # the same drivers, once over smm_eval_batch / smm_eval_batch_noseed and once over the oracle's twins
    from oracle import oracle as O
    from smm_jl_amd.callers import _flat_problem
    m = S.MProb()
    S.addSampledParam(m, OrderedDict([("p1", [0.2, -3, 3]), ("p2", [-0.2, -20, 20])]))
    S.addMoment(m, {"name": ["mu1", "mu2"], "value": [-1.0, 10.0], "weight": [1.0, 1.0]})
    S.addEvalFunc(m, S.objfunc_norm)
    prob = _flat_problem(m)
    octx = O.OracleContext(prob, S.BGPOpts(N=1, maxiter=1, sigma=[0.05], acc_tuner=[1.0], min_improve=[0.0]))

    def oracle_evaluator(mm, P, noseed_base=None):
        return octx.eval_batch(P) if noseed_base is None else octx.eval_batch_noseed(P, noseed_base)

    sd, so = S.doSlices(m, 17), S.doSlices(m, 17, evaluator=oracle_evaluator)
    for pp in ("p1", "p2"):
        for what in ("value", "mu1", "mu2"):
            a, b = sd.get(pp, what), so.get(pp, what)
            assert np.array_equal(a["x"], b["x"]) and np.allclose(a["y"], b["y"], rtol=1e-12, atol=1e-13)
    p = OrderedDict([("p1", -0.7), ("p2", 8.0)])
    assert np.allclose(S.FD_gradient(m, p), S.FD_gradient(m, p, evaluator=oracle_evaluator), rtol=1e-9, atol=1e-12)
    assert np.allclose(S.FD_gradient(m, p), np.eye(2), atol=1e-9)     # the moments are the parameters plus fixed shocks
    assert np.allclose(S.getSigma(m, p, 64, seed=5), S.getSigma(m, p, 64, seed=5, evaluator=oracle_evaluator), rtol=1e-9)
    out = S.optSlices(m, 25, tol=1e-2, update=0.4)
    assert abs(out["best"]["p"]["p1"] + 1.0) < 0.1 and abs(out["best"]["p"]["p2"] - 10.0) < 0.5
    se = S.get_stdErrors(m, p, reps=100, seed=3)
    assert all(0.0 < v < 1.0 for v in se.values())                    # sd of a mean of 10000 unit normals: ~0.01
