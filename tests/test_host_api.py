"""Host-side mirror of the reference API (no device needed): test/test_MProb.jl, test/test_Eval.jl."""
from collections import OrderedDict

import numpy as np
import pytest

import smm_jl_amd as S


def test_mprob_constructor_and_methods():
    # test/test_MProb.jl:17-47
    mprob = S.MProb()
    assert isinstance(mprob, S.MProb)
    S.addParam(mprob, OrderedDict([("a", 0.1), ("b", 0.2), ("c", 0.5)]))
    assert S.ps_names(mprob) == ["a", "b", "c"] and len(mprob.params_to_sample) == 0
    mprob = S.MProb()
    S.addSampledParam(mprob, OrderedDict([("a", [0.1, 0, 1]), ("b", [0.2, 0, 1])]))
    assert S.ps_names(mprob) == ["a", "b"] and S.ps2s_names(mprob) == ["a", "b"]
    mprob = S.MProb()
    S.addSampledParam(mprob, "a", 0.1, 0, 1)
    S.addSampledParam(mprob, "b", 0.1, 0, 1)
    moms = {"name": ["alpha", "beta", "gamma"], "value": [0.8, 0.7, 0.5], "weight": list(np.random.rand(3))}
    S.addMoment(mprob, moms)
    S.addEvalFunc(mprob, S.objfunc_norm)
    assert callable(mprob.objfunc)
    assert S.ps_names(mprob) == ["a", "b"] and S.ms_names(mprob) == ["alpha", "beta", "gamma"]
    with pytest.raises(AssertionError):
        S.addSampledParam(mprob, "z", 0.1, 1, 0)  # @assert ub>lb, mprob.jl:82


def test_eval_accessors():
    # test/test_Eval.jl:30-64
    p = OrderedDict([("a", 3.1), ("b", 4.9)])
    moms = {"name": ["alpha", "beta", "gamma"], "value": [0.8, 0.7, 0.5], "weight": [0.1, 0.2, 0.3]}
    ev = S.Eval(p, moms)
    assert S.param(ev, "a") == 3.1 and np.array_equal(S.param(ev), [3.1, 4.9]) and S.paramd(ev) == p
    assert S.dataMoment(ev, "alpha") == 0.8 and np.array_equal(S.dataMomentW(ev), [0.1, 0.2, 0.3])
    assert ev.status == -1 and ev.value == -1.0 and ev.prob == 0.0 and not ev.accepted  # Eval.jl:32-46
    S.setMoments(ev, {"alpha": 0.78, "beta": 0.81})
    assert ev.simMoments["alpha"] == 0.78
    S.setValue(ev, 4.2)
    assert ev.value == 4.2

    class MyP:
        a = 0.0
        b = 0.0
    x = MyP()
    S.fill(x, ev)
    assert x.a == 3.1 and x.b == 4.9
    with pytest.raises(ValueError):
        S.Eval(p, {"name": ["alpha"], "value": [1.0]})  # needs a weight column, Eval.jl:63


def test_eval_from_mprob():
    m = S.MProb()
    S.addSampledParam(m, OrderedDict([("p1", [0.2, -3, 3]), ("p2", [-0.2, -20, 20])]))
    S.addMoment(m, "mu1", -1.0, 1.0)
    S.addMoment(m, "mu2", 10.0)
    ev = S.Eval(m)
    assert list(ev.params.items()) == [("p1", 0.2), ("p2", -0.2)]
    assert ev.dataMoments == {"mu1": -1.0, "mu2": 10.0} and ev.dataMomentsW == {"mu1": 1.0, "mu2": 1.0}
    ev2 = S.Eval(m, {"p1": 1.0, "p2": 2.0})
    assert S.param(ev2, "p2") == 2.0
    assert ev == S.Eval(m) and not (ev == ev2)


def test_host_closures_are_rejected():
    from smm_jl_amd.host import _flat_problem
    m = S.MProb()
    S.addSampledParam(m, "a", 0.1, 0, 1)
    S.addMoment(m, "alpha", 0.5)
    S.addEvalFunc(m, lambda ev: ev)
    with pytest.raises(TypeError):
        _flat_problem(m)
