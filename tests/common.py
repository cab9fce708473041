"""Shared builders for the tests: the reference's example problems as flat Problem/BGPOpts."""
import numpy as np

import smm_jl_amd as S
from smm_jl_amd import _abi as A


def temps(N, maxtemp):
    # range(1.0, stop=maxtemp, length=N), AlgoBGP.jl:508
    return np.linspace(1.0, maxtemp, N) if N > 1 else np.ones(1)


def serial_normal(N=3, T=200, ns=10000, acc_tuners=None, min_improve=0.0, maxtemp=5.0, sigma0=0.05, seed=12,
                  p2_bounds=(-20.0, 20.0), mom=(-1.0, 10.0), w=(1.0, 1.0), objective_id=A.SMM_OBJ_NORM,
                  obj_params=None, **kw):
    """serialNormal(2, T): Examples.jl:118-153 + snorm_impl :373-416 (N=3, acc_tuners=[20,2,1])."""
    if acc_tuners is None:
        acc_tuners = [20.0, 2.0, 1.0] if N == 3 else np.geomspace(20.0, 1.0, N)
    prob = S.Problem(init=[0.2, -0.2], lb=[-3.0, p2_bounds[0]], ub=[3.0, p2_bounds[1]], mom=list(mom), w=list(w),
                     ns=ns, objective_id=objective_id, obj_params=obj_params)
    opts = S.BGPOpts(N=kw.pop("N_local", N), maxiter=T, sigma=sigma0 * temps(N, maxtemp),
                     acc_tuner=np.broadcast_to(np.asarray(acc_tuners, float), (N,)).copy(),
                     min_improve=np.broadcast_to(np.asarray(min_improve, float), (N,)).copy(), seed=seed,
                     N_global=N, **kw)
    return prob, opts


def general_normal(npar, N, T, ns=1000, seed=7, batch_size=None, **kw):
    """an np-dimensional objfunc_norm problem in the spirit of snorm_impl(npar>2), Examples.jl:392-405"""
    rng = np.random.default_rng(seed)
    half = rng.uniform(1.0, 5.0, npar)
    init = rng.uniform(-0.5, 0.5, npar) * half
    mom = rng.uniform(-0.5, 0.5, npar) * half
    w = rng.uniform(0.5, 2.0, npar)
    prob = S.Problem(init=init, lb=-half, ub=half, mom=mom, w=w, ns=ns)
    opts = S.BGPOpts(N=kw.pop("N_local", N), maxiter=T, sigma=0.05 * temps(N, 3.0),
                     acc_tuner=np.geomspace(10.0, 1.0, N) if N > 1 else np.array([2.0]),
                     min_improve=np.zeros(N), seed=seed, batch_size=batch_size, N_global=N, **kw)
    return prob, opts


def random_tables(prob, opts, tries=24, seed=99, pairs=True, Z=True):
    rng = np.random.default_rng(seed)
    T, N, Ng = opts.maxiter, opts.N, opts.N_global
    K = Ng - 1 if Ng < 3 else Ng
    ptab = None
    if pairs and K > 0:
        ptab = np.empty((T, K, 2), np.int32)
        M = Ng * (Ng - 1) // 2
        for t in range(T):
            m = rng.choice(M, size=K, replace=False)
            j = np.floor((1 + np.sqrt(1 + 8 * m.astype(float))) / 2).astype(np.int64)
            j = np.where(j * (j - 1) // 2 > m, j - 1, j)
            j = np.where((j + 1) * j // 2 <= m, j + 1, j)
            ptab[t, :, 1] = j
            ptab[t, :, 0] = m - j * (j - 1) // 2
    return S.Tables(probs_acc=rng.random((T, N)),
                    prop_normals=rng.standard_normal((T, tries, prob.np, N)),
                    pairs=ptab,
                    Z=rng.standard_normal((prob.nm, prob.ns)) if Z else None)


INT_FIELDS = ("best_id", "exchanged", "accepted", "status")
F64_FIELDS = ("value", "prob", "curr_val", "best_val", "params", "sim_moments")


def assert_history_equal(ha, hb, rtol=1e-9, exact_floats=False, atol=0.0):
    """bit-exact on bookkeeping (accepted / exchanged / best_id / status); floats within rtol
    (BASELINE.json north_star: 1e-6 relative on the objective; we hold 1e-9)."""
    for f in INT_FIELDS:
        a, b = getattr(ha, f), getattr(hb, f)
        assert a.shape == b.shape, f
        bad = np.argwhere(a != b)
        assert bad.size == 0, "%s differs at %s (first of %d)" % (f, bad[0], len(bad))
    for f in F64_FIELDS:
        a, b = getattr(ha, f), getattr(hb, f)
        if exact_floats:
            assert np.array_equal(a, b, equal_nan=True), f
        else:
            np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, equal_nan=True, err_msg=f)


def assert_state_equal(sa, sb, rtol=1e-9, atol=0.0):
    assert sa.iter == sb.iter
    for f in ("la_status", "n_noex", "n_acc_noex", "best_id"):
        assert np.array_equal(getattr(sa, f), getattr(sb, f)), f
    for f in ("sigma", "accept_rate", "la_value", "la_prob", "la_params", "la_sim_moments", "best_val"):
        np.testing.assert_allclose(getattr(sa, f), getattr(sb, f), rtol=rtol, atol=atol, equal_nan=True, err_msg=f)
