"""Committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py with the oracle):
the oracle must reproduce them here; the HIP path must reproduce them on the GPU (-m gpu)."""
import os

import numpy as np
import pytest

import common as cm
import smm_jl_amd as S
from smm_jl_amd import _abi as A

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name + ".npz"))


def tables_of(z):
    return S.Tables(probs_acc=z["t_probs_acc"], prop_normals=z["t_prop_normals"], pairs=z["t_pairs"], Z=z["t_Z"])


def check_run(ctx, z, T, rtol):
    ctx.step(T)
    h, s = ctx.history(), ctx.state()
    for f in cm.INT_FIELDS:
        assert np.array_equal(getattr(h, f), z["h_" + f]), f
    for f in cm.F64_FIELDS:
        np.testing.assert_allclose(getattr(h, f), z["h_" + f], rtol=rtol, atol=0, equal_nan=True, err_msg=f)
    for f in ("la_status", "n_noex", "n_acc_noex", "best_id"):
        assert np.array_equal(getattr(s, f), z["s_" + f]), f
    for f in ("sigma", "accept_rate", "la_value", "la_params", "best_val"):
        np.testing.assert_allclose(getattr(s, f), z["s_" + f], rtol=rtol, atol=0, equal_nan=True, err_msg=f)


CASES = {
    "g2_c1_trajectory": lambda: cm.serial_normal(N=3, T=200, ns=500) + (200,),
    "g3_exchange_order": lambda: cm.serial_normal(N=6, T=8, ns=200, acc_tuners=[1.0] * 6, min_improve=0.0) + (8,),
    "g4a_failbox": lambda: cm.serial_normal(N=8, T=60, ns=200, objective_id=A.SMM_OBJ_NORM_FAILBOX,
                                            obj_params=[-0.2, 0.1]) + (60,),
}


def g4b():
    prob, opts = cm.general_normal(4, N=10, T=50, ns=128, batch_size=2, sigma_update_steps=5, sigma_adjust_by=0.1)
    opts.min_improve[:] = 0.01
    return prob, opts, 50


def g5():
    npar, N, T = 10, 16, 30
    prob = S.Problem(init=np.zeros(npar), lb=-2 * np.ones(npar), ub=2 * np.ones(npar), mom=np.zeros(npar),
                     w=np.ones(npar), ns=1, objective_id=A.SMM_OBJ_BANANA)
    opts = S.BGPOpts(N=N, maxiter=T, sigma=0.02 * cm.temps(N, 5), acc_tuner=np.geomspace(20, 1, N), min_improve=np.zeros(N))
    return prob, opts, T


CASES["g4b_sigma_batches"] = g4b
CASES["g5_banana10"] = g5


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_reproduces_golden_run(O, name):
    z = load(name)
    prob, opts, T = CASES[name]()
    check_run(O.OracleContext(prob, opts, tables_of(z)), z, T, rtol=0)


def test_oracle_reproduces_golden_objective(O):
    z = load("g1_objfunc_norm")
    prob, opts = cm.serial_normal(N=3, T=1, ns=z["Z"].shape[1])
    v, sm, st = O.OracleContext(prob, opts, S.Tables(Z=z["Z"])).eval_batch(z["params"])
    assert np.array_equal(v, z["value"]) and np.array_equal(sm, z["sim_moments"]) and np.array_equal(st, z["status"])
    probu, _ = cm.serial_normal(N=3, T=1, ns=z["Z"].shape[1], w=(np.nan, np.nan))
    vu, _, _ = O.OracleContext(probu, opts, S.Tables(Z=z["Z"])).eval_batch(z["params"])
    assert np.array_equal(vu, z["value_unweighted"])
    # hand check of one grid point against the definition (ObjExamples.jl:79-101)
    i = 17
    m = z["params"][:, i] + np.array([z["Z"][k].mean() for k in range(2)])
    assert z["value"][i] == pytest.approx(np.mean((m - np.array([-1.0, 10.0])) ** 2), rel=1e-12)


def test_default_shock_matrix_fingerprint(O):
    z = load("g1b_default_Z")
    Zd = O.gen_Z(12, 2, 10000)
    assert np.array_equal(Zd[:, :32], z["head"]) and np.array_equal(Zd[:, ::997], z["strided"])   # (the generator's functions are the contract's: no libm in it)
    np.testing.assert_allclose(Zd.sum(1), z["col_sums"], rtol=1e-12)


def test_golden_exchange_case_has_multi_pair_chains():
    z = load("g3_exchange_order")
    assert (np.bincount(z["t_pairs"][0].ravel()) >= 3).sum() >= 2  # chains 0 and 2 occur in >= 3 pairs
    assert (z["h_exchanged"] != 0).any()


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_hip_reproduces_golden_run(name):
    z = load(name)
    prob, opts, T = CASES[name]()
    check_run(S.hip_context(prob, opts, tables_of(z)), z, T, rtol=0)   # bit for bit: the exponential is part of the numerical contract (include/smmhip.h)


@pytest.mark.gpu
def test_hip_reproduces_golden_objective():
    z = load("g1_objfunc_norm")
    prob, opts = cm.serial_normal(N=3, T=1, ns=z["Z"].shape[1])
    v, sm, st = S.hip_context(prob, opts, S.Tables(Z=z["Z"])).eval_batch(z["params"])
    assert np.array_equal(v, z["value"]) and np.array_equal(sm, z["sim_moments"]) and np.array_equal(st, z["status"])


@pytest.mark.gpu
def test_hip_default_shock_matrix_fingerprint():
    z = load("g1b_default_Z")
    prob, opts = cm.serial_normal(N=3, T=1)
    Zd = S.hip_context(prob, opts).Z()
    np.testing.assert_allclose(Zd[:, :32], z["head"], rtol=1e-14)
    np.testing.assert_allclose(Zd.sum(1), z["col_sums"], rtol=1e-12)
