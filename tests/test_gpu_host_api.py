"""The reference-shaped host API on the GPU: MAlgoBGP / run / history / summary / save / restart
(test/test_algoBGP.jl, test/test_AlgoAbstract.jl, test/test_BGPchain.jl, test/test_objfunc.jl)."""
import os
from collections import OrderedDict

import numpy as np
import pytest

import common as cm
import smm_jl_amd as S

pytestmark = pytest.mark.gpu


def make_mprob(p2=(-0.2, -20, 20), mom=(-1.0, 10.0)):
    m = S.MProb()
    S.addSampledParam(m, OrderedDict([("p1", [0.2, -3, 3]), ("p2", list(p2))]))
    S.addMoment(m, {"name": ["mu1", "mu2"], "value": list(mom), "weight": [1.0, 1.0]})
    S.addEvalFunc(m, S.objfunc_norm)
    return m


def test_constructor_defaults():
    # test_algoBGP.jl:14-28
    MA = S.MAlgoBGP(make_mprob())
    assert len(MA.chains) == 3 and MA.i == 0
    c = MA.chains[0]
    assert c.id == 1 and c.iter == 0 and not c.accepted.any() and (c.exchanged == 0).all()
    assert c.sigma == pytest.approx(0.05) and MA.chains[2].sigma == pytest.approx(0.1)  # 0.05 * range(1, 2, length=3)


def test_serial_normal_history_shape_and_oracle(O):
    # test_algoBGP.jl:30-38: serialNormal(2,20); history is 20 x 9
    MA = S.serialNormal(2, 20)
    h = S.history(MA.chains[0])
    assert h.shape == (20, 9)
    assert list(h.columns) == ["iter", "value", "accepted", "curr_val", "best_val", "prob", "exchanged", "p1", "p2"]
    assert MA.i == 20 and "time" in MA.opts
    prob, opts = cm.serial_normal(N=3, T=20)
    o = O.OracleContext(prob, opts, S.Tables(Z=MA._ctx.Z())); o.step(20)
    cm.assert_history_equal(MA._history(), o.history())
    sm = S.summary(MA)
    assert list(sm.columns) == ["id", "acc_rate", "perc_exchanged", "exchanged_most_with", "best_val"] and len(sm) == 3


def test_objfunc_norm_eval():
    # test_objfunc.jl:22-29: at the truth the simulated moments are within 0.1 of the data moments
    m = S.MProb()
    S.addSampledParam(m, OrderedDict([("a", [0.0, -1, 1]), ("b", [0.0, -1, 1])]))
    S.addMoment(m, "mu1", 0.0, None)
    S.addMoment(m, "mu2", 0.0, None)
    S.addEvalFunc(m, S.objfunc_norm)
    ev = S.evaluateObjective(m, {"a": 0.0, "b": 0.0})
    assert ev.status == 1 and all(abs(ev.simMoments[k] - ev.dataMoments[k]) < 0.1 for k in ev.dataMoments)
    assert ev.value == pytest.approx(np.mean([v ** 2 for v in ev.simMoments.values()]))


def test_chain_views_and_accept_rule():
    # test_BGPchain.jl:95-144 on a finished run
    MA = S.MAlgoBGP(make_mprob(), {"N": 3, "maxiter": 60, "maxtemp": 5, "smpl_iters": 1000, "min_improve": [0.0] * 3,
                                   "acc_tuners": [20.0, 2.0, 1.0]})
    S.run(MA)
    c = MA.chains[0]
    assert c.iter == 60 and c.accepted[0] and c.evals[0].prob == 1 and c.evals[0].accepted
    assert S.param(c.evals[0]).tolist() == [0.2, -0.2]
    evs = S.allAccepted(c)
    assert len(evs) == c.accepted.sum() and all(e.accepted for e in evs)
    pr = S.params(c)
    assert set(pr) == {"p1", "p2"} and len(pr["p1"]) == c.accepted.sum()
    v, i = S.best(c)
    assert v == S.history(c)["value"].min() and 1 <= i <= 60
    assert set(S.mean(c)) == set(S.median(c)) == set(S.CI(c)) == {"p1", "p2"}


def test_statistical_recovery():
    # test_algoBGP.jl:57-121 (N=2, 200 iterations, atol 1.0 on the median of chain 1's accepted draws)
    MA = S.MAlgoBGP(make_mprob(p2=(-0.2, -2, 2), mom=(-1.0, 1.0)),
                    {"N": 2, "maxiter": 200, "maxtemp": 5, "sigma": 0.05, "sigma_update_steps": 201, "smpl_iters": 1000,
                     "min_improve": [0.0, 0.0], "acc_tuners": [5.0, 1.0]})
    S.run(MA)
    med = S.median(MA.chains[0])
    assert abs(med["p1"] + 1.0) < 1.0 and abs(med["p2"] - 1.0) < 1.0


def test_save_load_and_restart(tmp_path, O):
    # test_AlgoAbstract.jl:47-61 (save every 5, read back field by field) and restart!
    opts = {"N": 2, "maxiter": 10, "maxtemp": 3, "smpl_iters": 1000, "min_improve": [0.0, 0.0], "acc_tuners": [2.0, 2.0],
            "save_frequency": 5, "filename": str(tmp_path / "run")}
    MA = S.MAlgoBGP(make_mprob(), dict(opts))
    S.run(MA)
    assert os.path.exists(str(tmp_path / "run.npz"))
    MB = S.readMalgo(S.MAlgoBGP(make_mprob(), dict(opts)), str(tmp_path / "run"))
    for f in S._abi.HistoryBuffers.FIELDS:
        assert np.array_equal(getattr(MA._history(), f), getattr(MB._history(), f), equal_nan=True), f
    for a, b in zip(MA.chains, MB.chains):
        assert a.accept_rate == b.accept_rate and a.sigma == b.sigma and np.array_equal(a.best_val, b.best_val)
    # restart!(algo, 15) continues the same chains: equals an uninterrupted 25-iteration run
    MB.opts.pop("filename"); MB.opts.pop("save_frequency")
    S.restart(MB, 15)
    full = dict(opts); full["maxiter"] = 25; full.pop("filename"); full.pop("save_frequency")
    MC = S.MAlgoBGP(make_mprob(), full)
    S.run(MC)
    assert MB.i == 25
    for f in S._abi.HistoryBuffers.FIELDS:
        assert np.array_equal(getattr(MB._history(), f), getattr(MC._history(), f), equal_nan=True), f


def test_compute_next_iteration_direct_calls():
    # computeNextIteration!(algo) called directly (the seam README.md:105-107 documents) keeps algo.i, chains
    # and history in step with the device; equals run!
    o = {"N": 3, "maxiter": 12, "maxtemp": 5, "smpl_iters": 1000, "min_improve": [0.0] * 3, "acc_tuners": [20.0, 2.0, 1.0]}
    MA = S.MAlgoBGP(make_mprob(), dict(o))
    for i in range(1, 13):
        MA.i = i                      # as run! does, AlgoAbstract.jl:38-45
        S.computeNextIteration(MA)
        assert MA.i == i and MA.chains[0].iter == i and len(S.history(MA.chains[1])) == i
    MB = S.MAlgoBGP(make_mprob(), dict(o))
    for i in range(12):
        S.computeNextIteration(MB)    # bare calls
    MC = S.MAlgoBGP(make_mprob(), dict(o))
    S.run(MC)
    for f in S._abi.HistoryBuffers.FIELDS:
        assert np.array_equal(getattr(MA._history(), f), getattr(MC._history(), f), equal_nan=True), f
        assert np.array_equal(getattr(MB._history(), f), getattr(MC._history(), f), equal_nan=True), f


def test_evaluate_objective_cache_follows_the_problem():
    # the cached evaluation context must notice changed objective parameters / moments, and NaN weights must not
    # rebuild it on every call
    m = make_mprob()
    m.objfunc = S.objfunc_norm
    e1 = S.evaluateObjective(m, {"p1": 0.2, "p2": -0.2})
    c1 = m._eval_ctx[1]
    e2 = S.evaluateObjective(m, {"p1": 0.2, "p2": -0.2})
    assert m._eval_ctx[1] is c1 and e1.value == e2.value
    S.addMoment(m, "mu1", 5.0, 1.0)                      # a data moment changes: new device copy
    e3 = S.evaluateObjective(m, {"p1": 0.2, "p2": -0.2})
    assert m._eval_ctx[1] is not c1 and e3.value != e1.value
    mn = S.MProb()
    S.addSampledParam(mn, OrderedDict([("a", [0.0, -1, 1]), ("b", [0.0, -1, 1])]))
    S.addMoment(mn, "mu1", 0.0, None); S.addMoment(mn, "mu2", 0.0, None)   # no weights: NaN
    S.addEvalFunc(mn, S.objfunc_norm)
    S.evaluateObjective(mn, {"a": 0.0, "b": 0.0})
    cn = mn._eval_ctx[1]
    S.evaluateObjective(mn, {"a": 0.1, "b": 0.0})
    assert mn._eval_ctx[1] is cn


def test_snorm_impl_more_parameters():
    # snorm_impl(opts, niter; npar = 4), Examples.jl:390-405 (bounds from our own seeded generator): runs on the general kernel
    opts = {"N": 4, "maxiter": 30, "maxtemp": 3, "smpl_iters": 1000, "min_improve": [0.0] * 4, "acc_tuners": [10.0, 5.0, 2.0, 1.0]}
    MA = S.snorm_impl(opts, 30, npar=4)
    h = S.history(MA.chains[0])
    assert h.shape == (30, 11) and list(h.columns)[-4:] == ["p1", "p2", "p3", "p4"] and MA.i == 30
    assert np.isfinite(h["value"]).all() and (np.diff(h["best_val"]) <= 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("protocol", ["records", "values"])
def test_sharded_protocols_on_one_rank(S, protocol):
    # ShardedBGP over the library's stream with the device engine, world size 1: both host protocols (the record all-gather in
    # its fused form; the values form: export_values / a2a_pack / a2a_apply) against the single-shard loop
    import torch
    from smm_jl_amd.dist import HipShardEngine, ShardedBGP
    prob, opts = cm.serial_normal(N=200, T=40, ns=200)
    a = S.hip_context(prob, opts)
    a.step(40)
    b = S.hip_context(prob, opts)
    sh = ShardedBGP(HipShardEngine(b, torch.device("cuda", 0)), protocol=protocol)
    sh.step(25); sh.step(15)
    sh.sync()
    cm.assert_history_equal(a.history(), b.history(), exact_floats=True)
    cm.assert_state_equal(a.state(), b.state(), rtol=0)
    assert (a.history().exchanged != 0).any()
