"""User-defined device objectives (include/smmhip.h SMM_USER_OBJECTIVE)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import common as cm  # noqa: E402
from user_objective_src import AR1_SOURCE, PANEL_SOURCE, ar1_numpy  # noqa: E402


def ar1_problem(S, oid, N, T, fail_above=None, seed=5):
    udata = [400.0] + ([fail_above] if fail_above is not None else [])
    prob = S.Problem(init=[0.3, 1.0], lb=[-0.95, 0.1], ub=[0.95, 3.0], mom=[0.0, 0.12, 0.06], w=[0.05, 0.05, 0.05], ns=1,
                     objective_id=oid, obj_params=udata)
    opts = S.BGPOpts(N=N, maxiter=T, sigma=0.05 * cm.temps(N, 4.0), acc_tuner=np.geomspace(3.0, 0.5, N) if N > 1 else [2.0],
                     min_improve=np.zeros(N), seed=seed, N_global=N)
    return prob, opts


def test_oracle_user_objective_hook(O, S):
    # CPU only: the oracle calls the gcc build of the objective source; against an independent numpy restatement
    oid = 1000 + 63
    O.register_user_objective(AR1_SOURCE, oid)
    prob, opts = ar1_problem(S, oid, N=1, T=1, fail_above=0.5)
    o = O.OracleContext(prob, opts)
    rng = np.random.default_rng(0)
    th = np.stack([rng.uniform(-0.9, 0.9, 12), rng.uniform(0.2, 2.5, 12)])
    v, sm, st = o.eval_batch(th)
    for i in range(12):
        smr, vr, sr = ar1_numpy(th[:, i], prob.mom, prob.w, prob.obj_params)
        assert st[i] == sr and np.allclose(sm[:, i], smr, rtol=1e-13, atol=1e-15)
        assert (v[i] == -1.0) if sr < 0 else np.isclose(v[i], vr, rtol=1e-12)
    assert (st < 0).any() and (st > 0).any()


@pytest.mark.gpu
def test_user_objective_eval_batch_and_run(S, O):
    oid = S.register_user_objective(AR1_SOURCE)
    assert oid >= 1000
    O.register_user_objective(AR1_SOURCE, oid)
    prob, opts = ar1_problem(S, oid, N=24, T=40, fail_above=0.8)
    h = S.hip_context(prob, opts)
    o = O.OracleContext(prob, opts)
    rng = np.random.default_rng(1)
    th = np.stack([rng.uniform(-0.9, 0.9, 300), rng.uniform(0.2, 2.5, 300)])
    vh, smh, sth = h.eval_batch(th)
    vo, smo, sto = o.eval_batch(th)
    assert np.array_equal(sth, sto) and np.array_equal(smh, smo) and np.array_equal(vh, vo)   # pure arithmetic: bit-exact
    assert (sth < 0).any()
    h.step(40); o.step(40)
    hh = h.history()
    cm.assert_history_equal(hh, o.history())
    cm.assert_state_equal(h.state(), o.state())
    assert (hh.exchanged != 0).any() and (hh.status == -2).any() and hh.accepted[1:].any()


@pytest.mark.gpu
def test_user_objective_sharded_equals_single(S, O):
    from test_gpu_parity import sharded_run_fused
    from smm_jl_amd import _abi as A
    oid = S.register_user_objective(AR1_SOURCE)
    prob, opts = ar1_problem(S, oid, N=32, T=20)
    single = S.hip_context(prob, opts)
    single.step(20)
    ctxs = sharded_run_fused(S, prob, opts, 2, 20)
    hs = single.history()
    for r, c in enumerate(ctxs):
        hr = c.history()
        for f in A.HistoryBuffers.FIELDS:
            assert np.array_equal(getattr(hr, f), getattr(hs, f)[..., r * 16:(r + 1) * 16], equal_nan=True), (f, r)


@pytest.mark.gpu
def test_user_objective_through_the_host_api(S):
    m = S.MProb()
    S.addSampledParam(m, {"rho": [0.3, -0.95, 0.95], "sigma": [1.0, 0.1, 3.0]})
    S.addMoment(m, {"name": ["m1", "m2", "m3"], "value": [0.0, 0.12, 0.06], "weight": [0.05, 0.05, 0.05]})
    S.addEvalFunc(m, S.user_objective(AR1_SOURCE, name="ar1"))
    m.objfunc_opts["obj_params"] = [400.0]
    MA = S.MAlgoBGP(m, {"N": 6, "maxiter": 60, "maxtemp": 3, "smpl_iters": 1000, "min_improve": [0.0] * 6,
                        "acc_tuners": [3.0, 2.0, 1.5, 1.0, 0.7, 0.5]})
    S.run(MA)
    h = S.history(MA.chains[0])
    assert len(h["value"]) == 60 and np.isfinite(h["value"]).all() and min(h["best_val"]) < h["value"][0]


@pytest.mark.gpu
def test_user_objective_compile_error_is_reported(S):
    with pytest.raises(RuntimeError) as e:
        S.register_user_objective("SMM_USER_OBJECTIVE(const double* theta) { this is not C }")
    assert "compile" in str(e.value)


def panel_problem(S, oid, N, T, seed=9, fail_above=0.85):
    prob = S.Problem(init=[0.3, 1.0], lb=[-0.95, 0.1], ub=[0.95, 3.0], mom=[0.0, 0.12, 0.06], w=[0.05, 0.05, 0.05], ns=1,
                     objective_id=oid, obj_params=[40.0, 1000.0] + ([fail_above] if fail_above is not None else []))   # 40 periods x 1000 agents; fails above rho = 0.85
    opts = S.BGPOpts(N=N, maxiter=T, sigma=0.05 * cm.temps(N, 4.0), acc_tuner=np.geomspace(3.0, 0.5, N) if N > 1 else [2.0],
                     min_improve=np.zeros(N), seed=seed, N_global=N)
    return prob, opts


def test_oracle_map_reduce_objective(O, S):
    # CPU only: lanes form in the oracle; 1000 agents over 256 lanes == the same sums in a different order (close, not equal)
    oid = 1000 + 62
    O.register_user_objective(PANEL_SOURCE, oid, n_sums=3, lanes=256)
    prob, opts = panel_problem(S, oid, N=1, T=1)
    o = O.OracleContext(prob, opts)
    th = np.array([[0.2, 0.7, 0.9], [1.0, 0.5, 2.0]])
    v, sm, st = o.eval_batch(th)
    assert list(st) == [1, 1, -2] and v[2] == -1.0
    # independent check of moment 2 for theta 0: plain python over all agents
    tot = 0.0
    for a in range(1000):
        s_ = 12345 + 7919 * a
        y = 0.0
        for _ in range(40):
            s_ = (s_ * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
            y = 0.2 * y + 1.0 * (float(s_ >> 11) / 9007199254740992.0 - 0.5)
            tot += y * y
    assert np.isclose(sm[1, 0], tot / 40000.0, rtol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("lanes", [64, 256, 1024])
def test_map_reduce_user_objective(S, O, lanes):
    oid = S.register_user_objective(PANEL_SOURCE, n_sums=3, lanes=lanes)
    O.register_user_objective(PANEL_SOURCE, oid, n_sums=3, lanes=lanes)
    prob, opts = panel_problem(S, oid, N=16, T=25)
    h = S.hip_context(prob, opts)
    o = O.OracleContext(prob, opts)
    rng = np.random.default_rng(2)
    th = np.stack([rng.uniform(-0.9, 0.9, 100), rng.uniform(0.2, 2.5, 100)])
    vh, smh, sth = h.eval_batch(th)
    vo, smo, sto = o.eval_batch(th)
    assert np.array_equal(sth, sto) and np.array_equal(smh, smo) and np.array_equal(vh, vo)   # same reduction order: bit-exact
    h.step(25); o.step(25)
    cm.assert_history_equal(h.history(), o.history())
    cm.assert_state_equal(h.state(), o.state())


@pytest.mark.gpu
def test_map_reduce_registration_checks(S):
    with pytest.raises(RuntimeError):
        S.register_user_objective(PANEL_SOURCE, n_sums=3, lanes=100)     # not a multiple of 64
    with pytest.raises(RuntimeError):
        S.register_user_objective(PANEL_SOURCE, n_sums=0, lanes=64)


@pytest.mark.gpu
@pytest.mark.parametrize("N,steps,fail_above", [(64, [1, 30, 9], 0.8), (256, [40], None), (32, [2, 3, 15], 0.8), (2048, [25], 0.9), (8192, [12], None)])
def test_user_objective_in_the_persistent_loop(S, O, N, steps, fail_above):
    # VERDICT r4 "Next #2 (c)": "bring your own objective" (mprob.jl:159,182) on the persistent form — the library compiles
    # k_chain_persist_gen once more, with the user's source inside (hiprtc, on demand), for populations in whole groups of 32 up to 8192
    # chains, min_improve == 0, np, nm <= 16.  Against the oracle (gcc build of the same text) and against the three launches per
    # iteration (proposal / the user's kernel / accept): identical to the bit; failing evaluations (status -2) included
    T = sum(steps)
    oid = S.register_user_objective(AR1_SOURCE)
    O.register_user_objective(AR1_SOURCE, oid)
    prob, opts = ar1_problem(S, oid, N=N, T=T, fail_above=fail_above)
    h = S.hip_context(prob, opts)
    assert h.persistent_info()[0] is True
    c = S.hip_context(prob, opts)
    c.set_persistent(False)
    o = O.OracleContext(prob, opts, threads=O.max_threads()) if N <= 2048 else None
    for n in steps:
        h.step(n); c.step(n)
        if o is not None:
            o.step(n)
    avail, launches, repairs = h.persistent_info()
    assert launches >= 1 and repairs == 0, (launches, repairs)
    assert c.persistent_info()[1] == 0
    hh = h.history()
    cm.assert_history_equal(hh, c.history(), exact_floats=True)
    cm.assert_state_equal(h.state(), c.state(), rtol=0)
    if o is not None:
        cm.assert_history_equal(hh, o.history())
        cm.assert_state_equal(h.state(), o.state())
    assert (hh.exchanged != 0).any() and hh.accepted[1:].any()
    if fail_above is not None and N >= 64:
        assert (hh.status == -2).any()


@pytest.mark.gpu
def test_user_objective_persistent_form_hard_error_and_where_it_does_not_apply(S, O):
    from smm_jl_amd import _abi as A
    oid = S.register_user_objective(AR1_SOURCE)
    prob, opts = ar1_problem(S, oid, N=100, T=10)            # not whole groups of 32
    assert S.hip_context(prob, opts).persistent_info()[0] is False
    oid2 = S.register_user_objective(PANEL_SOURCE, n_sums=3, lanes=1024)   # the map-reduce form with more lanes than a tile has: its own launches
    p2, o2 = panel_problem(S, oid2, N=64, T=10)
    assert S.hip_context(p2, o2).persistent_info()[0] is False
    # a hard error inside a launch (AlgoBGP.jl:409): replayed on the per-iteration launches to the failing iteration
    prob, opts = ar1_problem(S, oid, N=64, T=30)
    opts.sigma[:] = 40.0
    opts.smpl_iters = 2
    h = S.hip_context(prob, opts)
    c = S.hip_context(prob, opts)
    c.set_persistent(False)
    errs = []
    for ctx in (h, c):
        with pytest.raises(A.SMMHipError) as ei:
            ctx.step(30)
        errs.append(str(ei.value))
    assert errs[0] == errs[1] and "no draw in support" in errs[0], errs
    assert h.persistent_info()[2] >= 1
    cm.assert_history_equal(h.history(), c.history(), exact_floats=True)


@pytest.mark.gpu
@pytest.mark.parametrize("lanes,N,steps,fail_above,mi", [(256, 64, [1, 20, 9], None, 0.0), (64, 48, [30], 0.5, 0.0), (512, 40, [2, 3, 15], 0.5, 0.05), (128, 256, [25], None, 0.5),
                                                    (256, 1000, [12], 0.6, 0.0)])
def test_map_reduce_user_objective_in_the_persistent_loop(S, O, lanes, N, steps, fail_above, mi):
    # VERDICT r5 "Next #4": the form a real SIMULATION objective takes (MProb.objfunc, mprob.jl:159,182: a sum over many independent units —
    # SMM_USER_PARTIAL / SMM_USER_FINISH, `lanes` lanes per evaluation) inside the persistent loop: the library compiles k_chain_persist_tile
    # once more, with the user's source inside (hiprtc, on demand); a tile's 512 lanes evaluate 512 / lanes chains at a time with the reduction
    # order of the stand-alone kernel.  Against the three launches per iteration (proposal / the user's kernel / accept) to the bit, and against
    # the oracle (gcc build of the same text); failing evaluations (status -2), thresholds (the reference's default 0.5 included), partial tiles
    T = sum(steps)
    oid = S.register_user_objective(PANEL_SOURCE, n_sums=3, lanes=lanes)
    O.register_user_objective(PANEL_SOURCE, oid, n_sums=3, lanes=lanes)
    prob, opts = panel_problem(S, oid, N=N, T=T, fail_above=fail_above)
    opts.min_improve[:] = mi
    h = S.hip_context(prob, opts)
    assert h.describe()["persistent"] == "tile_user", h.describe()
    c = S.hip_context(prob, opts)
    c.set_persistent(False)
    o = O.OracleContext(prob, opts, threads=O.max_threads())
    for n in steps:
        h.step(n); c.step(n); o.step(n)
    avail, launches, repairs = h.persistent_info()
    assert launches >= 1 and repairs == 0, (launches, repairs)
    assert c.persistent_info()[1] == 0
    hh = h.history()
    cm.assert_history_equal(hh, c.history(), exact_floats=True)
    cm.assert_state_equal(h.state(), c.state(), rtol=0)
    cm.assert_history_equal(hh, o.history(), exact_floats=True)
    cm.assert_state_equal(h.state(), o.state(), rtol=0)
    assert hh.accepted[1:].any()
    if mi < 0.4:
        assert (hh.exchanged != 0).any()
    if fail_above is not None:
        assert (hh.status == -2).any()
