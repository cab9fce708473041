"""Shared builders for the tests: the reference's example problems as flat Problem/BGPOpts."""
import numpy as np

import smm_jl_amd as S
from smm_jl_amd import _abi as A


from smm_jl_amd.workloads import general_normal, serial_normal, temps  # noqa: E402,F401  (the builders live in the package: bench.py does not depend on tests/)


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
