"""SMM_OBJ_DENSE2 — BASELINE config 5 AS WORDED: the synthetic dense simulation with a 256 x 256 matvec per evaluation
(include/smmhip.h: x = B theta, h1 = tanh x, g = A2 h1, h2 = tanh g, y = A h2; the plugin seam is MProb.objfunc, mprob.jl:159,182;
the value is objfunc_norm's form, ObjExamples.jl:90-101).  CPU part: the oracle's restatement against plain numpy (tolerance: numpy sums
in another order) and against an exact rational evaluation of the contract's fma chains on a tiny case.  GPU part (-m gpu): the FP64 MFMA
path — per-iteration kernel, batched evaluation, the persistent tile kernel — BIT-identical to the oracle (the summation orders and the
tanh are numerical contract)."""
import numpy as np
import pytest

import common as cm
from smm_jl_amd import _abi as A

D = A.SMM_DENSE_D


def dense2_problem(npar, nm, N, T, seed=3, explicit=True, **kw):
    from smm_jl_amd import BGPOpts, Problem
    rng = np.random.default_rng(seed)
    objp = None
    if explicit:
        objp = np.concatenate([rng.standard_normal(D * npar) / np.sqrt(npar), rng.standard_normal(D * D) / np.sqrt(D),
                               rng.standard_normal(nm * D) / np.sqrt(D)])
    prob = Problem(init=rng.uniform(-0.3, 0.3, npar), lb=-np.ones(npar), ub=np.ones(npar), mom=rng.uniform(-0.5, 0.5, nm),
                   w=rng.uniform(0.5, 2.0, nm), ns=1, objective_id=A.SMM_OBJ_DENSE2, obj_params=objp)
    opts = BGPOpts(N=kw.pop("N_local", N), maxiter=T, sigma=0.02 * cm.temps(N, 4), acc_tuner=np.geomspace(20, 1, N) if N > 1 else [2.0],
                   min_improve=np.zeros(N), N_global=N, seed=seed, **kw)
    return prob, opts


def _numpy_dense2(prob, objp, p):
    npar, nm = prob.np, prob.nm
    B = objp[:D * npar].reshape(D, npar); A2 = objp[D * npar:D * npar + D * D].reshape(D, D); Am = objp[D * npar + D * D:].reshape(nm, D)
    y = Am @ np.tanh(A2 @ np.tanh(B @ p))
    w = np.asarray(prob.w)[:, None]
    dd = (y - np.asarray(prob.mom)[:, None]) / w
    return (dd * dd).mean(axis=0), y


@pytest.mark.parametrize("npar,nm", [(1, 1), (6, 5), (50, 50)])
def test_oracle_dense2_against_numpy(S, O, npar, nm):
    prob, opts = dense2_problem(npar, nm, N=4, T=2)
    o = O.OracleContext(prob, opts, S.Tables())
    p = np.random.default_rng(1).uniform(-1, 1, (npar, 37))
    v, m, st = o.eval_batch(p)
    vn, mn = _numpy_dense2(prob, np.asarray(prob.obj_params), p)
    np.testing.assert_allclose(m, mn, rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(v, vn, rtol=1e-11)
    assert (st == 1).all()


def test_oracle_dense2_generated_matrices(S, O):
    # empty obj_params: [B, A2, A] from the counter generator, N(0,1)/sqrt(fan-in); the generated blob evaluates like an explicit one
    prob, opts = dense2_problem(7, 9, N=4, T=2, explicit=False)
    g = O.gen_dense2(opts.seed, 7, 9)
    assert g.shape == (D * 7 + D * D + 9 * D,)
    assert abs(g[:D * 7].std() * np.sqrt(7) - 1) < 0.05 and abs(g[D * 7:].std() * np.sqrt(D) - 1) < 0.02
    # (spec v1's generator draws the same stream: its B is the same, its A is what v2 uses as the first rows of A2)
    assert np.array_equal(O.gen_dense(opts.seed, 7, 9)[:D * 7], g[:D * 7])
    p = np.random.default_rng(2).uniform(-1, 1, (7, 11))
    o = O.OracleContext(prob, opts, S.Tables())
    from smm_jl_amd import Problem
    prob2 = Problem(init=prob.init, lb=prob.lb, ub=prob.ub, mom=prob.mom, w=prob.w, ns=1, objective_id=A.SMM_OBJ_DENSE2, obj_params=g)
    o2 = O.OracleContext(prob2, opts, S.Tables())
    for a, b in zip(o.eval_batch(p), o2.eval_batch(p)):
        assert np.array_equal(a, b)


def test_oracle_dense2_contract_exact_rational(S, O):
    # the contract on a small case, every fma evaluated exactly (fractions) and rounded once: the oracle's chains to the bit
    from fractions import Fraction as F
    npar, nm = 2, 1
    prob, opts = dense2_problem(npar, nm, N=1, T=2, seed=11)
    objp = np.asarray(prob.obj_params)
    B = objp[:D * npar].reshape(D, npar); A2 = objp[D * npar:D * npar + D * D].reshape(D, D); Am = objp[D * npar + D * D:].reshape(nm, D)
    theta = np.array([0.3, -0.7])

    def fma(a, b, c):
        return float(F(a) * F(b) + F(c))   # float(Fraction) rounds to nearest even: a correctly rounded fma

    tanh = lambda x: float(O.dense_tanh(np.array([x]))[0])
    h1 = []
    for d in range(D):
        acc = 0.0
        for p in range(npar):
            acc = fma(B[d, p], theta[p], acc)
        h1.append(tanh(acc))
    h2 = []
    for j in range(D):
        acc = 0.0
        for d in range(D):
            acc = fma(A2[j, d], h1[d], acc)
        h2.append(tanh(acc))
    tot = None
    for wv in range(8):
        acc = 0.0
        for d in range(32 * wv, 32 * wv + 32):
            acc = fma(Am[0, d], h2[d], acc)
        tot = acc if tot is None else tot + acc
    o = O.OracleContext(prob, opts, S.Tables())
    v, m, st = o.eval_batch(theta[:, None])
    assert m[0, 0] == tot
    dd = (tot - prob.mom[0]) / prob.w[0]
    assert v[0] == dd * dd


def test_dense2_rejects_a_blob_of_the_wrong_size(S):
    from smm_jl_amd import Problem
    prob, opts = dense2_problem(3, 2, N=4, T=2)
    bad = Problem(init=prob.init, lb=prob.lb, ub=prob.ub, mom=prob.mom, w=prob.w, ns=1, objective_id=A.SMM_OBJ_DENSE2,
                  obj_params=np.zeros(D * 3 + 2 * D))   # spec v1's blob
    if A.load().smm_device_count() < 1:
        pytest.skip("no device: smm_ctx_create fails earlier")
    with pytest.raises(S.SMMHipError):
        S.hip_context(bad, opts)


# ------------------------------------------------------------------------------------------ the device
def _pair(S, O, prob, opts):
    h = S.hip_context(prob, opts)
    return h, O.OracleContext(prob, opts, S.Tables(Z=h.Z()), threads=O.max_threads())


@pytest.mark.gpu
@pytest.mark.parametrize("npar,nm", [(1, 1), (3, 2), (6, 5), (17, 33), (50, 50), (64, 64)])
def test_dense2_eval_batch_bit_identical(S, O, npar, nm):
    prob, opts = dense2_problem(npar, nm, N=4, T=2)
    h, o = _pair(S, O, prob, opts)
    rng = np.random.default_rng(1)
    for M in (1, 15, 16, 17, 200):
        p = rng.uniform(-1, 1, (npar, M))
        for a, b in zip(h.eval_batch(p), o.eval_batch(p)):
            assert np.array_equal(a, b)


@pytest.mark.gpu
def test_dense2_generated_matrices_bit_identical(S, O):
    prob, opts = dense2_problem(7, 9, N=4, T=2, explicit=False)
    h, o = _pair(S, O, prob, opts)
    p = np.random.default_rng(2).uniform(-1, 1, (7, 40))
    for a, b in zip(h.eval_batch(p), o.eval_batch(p)):
        assert np.array_equal(a, b)


def _exact(h, o):
    cm.assert_history_equal(h.history(), o.history(), exact_floats=True)
    cm.assert_state_equal(h.state(), o.state(), rtol=0)


@pytest.mark.gpu
@pytest.mark.parametrize("N", [1, 5, 16, 100])
def test_dense2_bgp_per_iteration_kernels(S, O, N):
    prob, opts = dense2_problem(6, 5, N=N, T=30)
    h, o = _pair(S, O, prob, opts)
    h.set_persistent(False)
    assert "dense2" in h.describe()["chain"], h.describe()
    h.step(30); o.step(30)
    _exact(h, o)


@pytest.mark.gpu
@pytest.mark.parametrize("npar,nm,N,mi,bs,steps", [(6, 5, 48, 0.0, None, [30]), (50, 50, 64, 0.0, None, [1, 5, 2, 12]), (50, 50, 112, 0.05, None, [20]),
                                                  (17, 33, 32, 0.0, None, [25]), (56, 60, 16, 0.0, None, [20]), (50, 50, 48, 0.0, 25, [20]),
                                                  (3, 2, 256, 0.5, None, [30]), (2, 1, 512, 0.0, None, [20])])
def test_dense2_persistent_tile_form(S, O, npar, nm, N, mi, bs, steps):
    prob, opts = dense2_problem(npar, nm, N=N, T=sum(steps), **({"batch_size": bs} if bs else {}))
    opts.min_improve[:] = mi
    h, o = _pair(S, O, prob, opts)
    c = S.hip_context(prob, opts)
    c.set_persistent(False)
    assert h.describe()["persistent"] == "tile_dense2", h.describe()
    for n in steps:
        h.step(n); c.step(n); o.step(n)
    assert h.persistent_info()[1] >= 1 and h.persistent_info()[2] == 0, h.persistent_info()
    cm.assert_history_equal(h.history(), c.history(), exact_floats=True)
    cm.assert_state_equal(h.state(), c.state(), rtol=0)
    _exact(h, o)


@pytest.mark.gpu
def test_c5_as_worded_full_size_bit_identical(S, O):
    import bench
    prob, opts = bench.build_problem("c5", 4096, 4096, 0, 40, 0)     # BASELINE configs[4]: 50 parameters, 256 x 256 matvec per evaluation, 4096 chains
    assert prob.objective_id == A.SMM_OBJ_DENSE2
    h, o = _pair(S, O, prob, opts)
    assert h.describe()["persistent"] == "tile_dense2", h.describe()
    for n in (1, 25, 14):
        h.step(n); o.step(n)
    _exact(h, o)
    hh = h.history()
    assert (hh.exchanged != 0).any() and 0.02 < hh.accepted[1:].mean() < 0.98
