"""The HIP path reproduces the oracle BIT FOR BIT on generated randomness — every floating-point field of the history and of the state, not
only the bookkeeping: the elementary functions of the path (the logarithm, sine and cosine of the generator's Box-Muller transform, the
exponential of the acceptance probability, AlgoBGP.jl:344; the dense objective's tanh) are part of the numerical contract
(include/smmhip.h), fixed sequences of correctly rounded operations in smm_rng.hpp / smm_chain.hpp and in the oracle.  north_star asks for
bit-exact accept / swap bookkeeping and 1e-6 relative on the objective; this holds array_equal on everything, at every BASELINE
configuration's shape and in every form of the chain kernels (persistent: loc, loc_wide, gen, tile_sim, tile_dense, tile_dense2; one launch per
iteration; large shards).  Replaces run!'s loop over computeNextIteration! (AlgoAbstract.jl:38-45, AlgoBGP.jl:589-640)."""
import numpy as np
import pytest

import common as cm
from test_gpu_parity import dense_problem

pytestmark = pytest.mark.gpu

FIELDS = cm.INT_FIELDS + cm.F64_FIELDS
STATE = ("sigma", "accept_rate", "la_value", "la_params", "best_val", "la_status", "n_noex", "n_acc_noex", "best_id")


def _exact(S, O, prob, opts, steps, persistent=True, form=None):
    h = S.hip_context(prob, opts)
    if not persistent:
        h.set_persistent(False)
    if form is not None:
        assert h.describe()["persistent"] == form, h.describe()
    o = O.OracleContext(prob, opts, S.Tables(Z=h.Z()), threads=O.max_threads())
    for n in steps:
        h.step(n); o.step(n)
    hh, oh, hs, os_ = h.history(), o.history(), h.state(), o.state()
    for f in FIELDS:
        assert np.array_equal(getattr(hh, f), getattr(oh, f), equal_nan=True), f
    for f in STATE:
        assert np.array_equal(getattr(hs, f), getattr(os_, f), equal_nan=True), f
    return h


def test_c1_bit_identical(S, O):
    prob, opts = cm.serial_normal(N=3, T=200, ns=500)     # BASELINE configs[0]: SMM.serialNormal(2, 200), 3 chains
    _exact(S, O, prob, opts, [200])


@pytest.mark.parametrize("persistent", [True, False])
def test_c2_full_size_bit_identical(S, O, persistent):
    T = 200 if persistent else 40
    prob, opts = cm.serial_normal(N=4096, T=T, ns=10000)   # BASELINE configs[1]: the headline workload, whole history
    h = _exact(S, O, prob, opts, [1, 120, 79] if persistent else [T], persistent, form="loc")
    assert (h.persistent_info()[1] >= 1) == persistent
    assert 0.05 < (h.history().exchanged != 0).mean() < 0.6 and 0.1 < h.history().accepted[1:].mean() < 0.9


def test_reference_default_threshold_bit_identical(S, O):
    prob, opts = cm.serial_normal(N=4096, T=80, ns=2000, min_improve=0.5)   # AlgoBGP.jl:522: the reference's default min_improve
    _exact(S, O, prob, opts, [80], form="loc_wide")


def test_c3_shape_bit_identical(S, O):
    import bench
    prob, opts = bench.build_problem("c3", 32768, 32768, 0, 6, 0)         # BASELINE configs[2] on one device: 32768 chains, ns = 10000
    _exact(S, O, prob, opts, [6])


def test_c4_bit_identical(S, O):
    import bench
    prob, opts = bench.build_problem("c4", 8192, 8192, 0, 100, 0)           # BASELINE configs[3]: banana, 10 parameters, 8192 chains
    _exact(S, O, prob, opts, [100], form="gen")


def test_c5_bit_identical(S, O):
    import bench
    prob, opts = bench.build_problem("c5", 4096, 4096, 0, 60, 0)            # BASELINE configs[4] AS WORDED: dense, 50 parameters, a 256 x 256 matvec per evaluation, FP64 MFMA
    _exact(S, O, prob, opts, [60], form="tile_dense2")


def test_c5_without_the_256x256_stage_bit_identical(S, O):
    import bench
    prob, opts = bench.build_problem("c5v1", 4096, 4096, 0, 60, 0)          # the instance of rounds 2-5 (SMM_OBJ_DENSE)
    _exact(S, O, prob, opts, [60], form="tile_dense")


@pytest.mark.parametrize("npar,N,ns", [(6, 4096, 1000), (18, 333, 1000), (3, 100, 10000)])
def test_reference_example_sizes_bit_identical(S, O, npar, N, ns):
    prob, opts = cm.general_normal(npar, N=N, T=50, ns=ns)                  # Examples.jl:210-230, 232-319: 6 and 18 parameters
    _exact(S, O, prob, opts, [50])


def test_failing_objective_and_batches_bit_identical(S, O):
    from smm_jl_amd import _abi as A
    prob, opts = cm.serial_normal(N=512, T=60, ns=300, objective_id=A.SMM_OBJ_NORM_FAILBOX, obj_params=[-0.2, 0.1])   # status -2, mprob.jl:183-186
    _exact(S, O, prob, opts, [60])
    prob, opts = cm.general_normal(4, N=64, T=50, ns=128, batch_size=2, sigma_update_steps=5, sigma_adjust_by=0.1)
    _exact(S, O, prob, opts, [50])
    prob, opts = dense_problem(S, O, 17, 33, N=48, T=30, explicit=False)      # the dense objective's generated matrices too
    _exact(S, O, prob, opts, [30])
