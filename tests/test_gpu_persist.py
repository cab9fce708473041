"""The persistent form of smm_bgp_step (smm.jl_amd/csrc/smm_chain_persist.hpp: one launch per look-ahead window, tiles coupled by a
ring of tagged slots instead of a kernel boundary) against the oracle and against the one-launch-per-iteration kernels.
Replaces the loop of run! over computeNextIteration! (AlgoAbstract.jl:38-45, AlgoBGP.jl:589-640); what is compared is the whole
history and state — bit-exact bookkeeping, floats within 1e-9 (in practice identical)."""
import numpy as np
import pytest

import common as cm
from smm_jl_amd import _abi as A

pytestmark = pytest.mark.gpu


def _pair(S, O, prob, opts, tab=None):
    h = S.hip_context(prob, opts, tab)
    t = tab if tab is not None else S.Tables()
    o = O.OracleContext(prob, opts, S.Tables(probs_acc=t.probs_acc, prop_normals=t.prop_normals, pairs=t.pairs, Z=h.Z()))
    return h, o


def _same(ha, hb, sa, sb):
    cm.assert_history_equal(ha, hb, exact_floats=True)
    cm.assert_state_equal(sa, sb, rtol=0)


@pytest.mark.parametrize("N,ns,steps", [(17, 200, [40]), (64, 1000, [1, 5, 2, 20, 12]), (333, 10000, [25, 25]), (2, 100, [30]), (16, 100, [30]),
                                       (33, 513, [30]), (100, 64, [300]), (1000, 300, [30]), (48, 10240, [3, 37])])
def test_persistent_form_against_oracle_and_per_iteration_kernels(S, O, N, ns, steps):
    T = sum(steps)
    prob, opts = cm.serial_normal(N=N, T=T, ns=ns, seed=5)
    h, o = _pair(S, O, prob, opts)
    c = S.hip_context(prob, opts)
    c.set_persistent(False)
    for n in steps:
        h.step(n); o.step(n); c.step(n)
    avail, launches, repairs = h.persistent_info()
    assert launches >= 1 and repairs == 0, (launches, repairs)          # the form under test really ran, and nothing was replayed
    assert c.persistent_info()[1] == 0
    cm.assert_history_equal(h.history(), o.history())
    cm.assert_state_equal(h.state(), o.state())
    _same(h.history(), c.history(), h.state(), c.state())                # the two forms: identical to the bit


def test_persistent_form_c2_full_size_across_plan_windows(S, O):
    # BASELINE configs[1] (4096 chains, ns = 10000) over 300 iterations: crosses the 256-iteration look-ahead window
    prob, opts = cm.serial_normal(N=4096, T=300)
    h, o = _pair(S, O, prob, opts)
    h.step(300); o.step(300)
    assert h.persistent_info()[1] >= 2 and h.persistent_info()[2] == 0
    cm.assert_history_equal(h.history(), o.history())
    cm.assert_state_equal(h.state(), o.state())
    assert 0.15 < (h.history().exchanged != 0).mean() < 0.4               # the exchange is really at work


@pytest.mark.parametrize("npar", [1, 2])
def test_persistent_form_injected_tables_and_one_parameter(S, O, npar):
    # every source of randomness injected (probs_acc, proposal normals incl. tries past the first four, pair lists); np = 1 and 2
    if npar == 2:
        prob, opts = cm.serial_normal(N=80, T=50, ns=700, seed=3, sigma0=0.02)
    else:
        prob, opts = cm.general_normal(1, N=80, T=50, ns=700, seed=3)
    tab = cm.random_tables(prob, opts, tries=24, seed=11)
    h, o = _pair(S, O, prob, opts, tab)
    for n in (2, 30, 18):
        h.step(n); o.step(n)
    assert h.persistent_info()[1] >= 2 and h.persistent_info()[2] == 0
    cm.assert_history_equal(h.history(), o.history())   # (bit-identical since round 5: tests/test_gpu_bitexact.py)
    cm.assert_state_equal(h.state(), o.state())


def test_persistent_form_failing_objective_and_restart(S, O):
    # objective "exceptions" (status -2, mprob.jl:183-186) inside persistent launches; then a state round trip and more steps
    prob, opts = cm.serial_normal(N=96, T=90, ns=400, seed=8, objective_id=A.SMM_OBJ_NORM_FAILBOX, obj_params=[0.05, 0.4])
    h, o = _pair(S, O, prob, opts)
    h.step(40); o.step(40)
    assert (h.history(0, 40).status == -2).any()
    st, hi = h.state(), h.history(0, 40)
    h2 = S.hip_context(prob, opts)
    h2.set_state(st, hi)
    h2.step(50); h.step(50); o.step(50)
    assert h.persistent_info()[2] == 0 and h2.persistent_info()[1] >= 1
    cm.assert_history_equal(h.history(), o.history())
    _same(h.history(), h2.history(), h.state(), h2.state())


def test_persistent_form_hard_error_is_replayed_on_the_per_iteration_path(S, O):
    # AlgoBGP.jl:409 inside a persistent launch: its tiles run on, the library rolls back to the state it saved and repeats the
    # iterations one launch each — the documented state at the failing iteration (tests/test_gpu_parity.py::test_hard_error_...)
    N, T, tfail = 40, 30, 17
    prob, opts = cm.serial_normal(N=N, T=T, ns=200, sigma0=0.01)
    tab = cm.random_tables(prob, opts, tries=8)
    tab.prop_normals[tfail - 1] = 1e9
    h, o = _pair(S, O, prob, opts, tab)
    h.step(5); o.step(5)
    with pytest.raises(A.SMMHipError) as eh:
        h.step(20)
    with pytest.raises(A.SMMHipError):
        o.step(20)
    assert eh.value.code == A.SMM_ERR_NO_DRAW_IN_SUPPORT and "iteration %d" % tfail in str(eh.value)
    assert h.persistent_info()[1] >= 1 and h.persistent_info()[2] == 1
    assert h.state().iter == tfail
    hh, ho = h.history(0, T), o.history(0, T)
    for f in cm.INT_FIELDS:
        np.testing.assert_array_equal(getattr(hh, f)[:tfail - 1], getattr(ho, f)[:tfail - 1], err_msg=f)
    assert np.isnan(hh.value[tfail:]).all() and (hh.status[tfail:] == 0).all() and (hh.best_id[tfail:] == -1).all()
    with pytest.raises(A.SMMHipError):
        h.step(1)   # sticky


def test_persistent_form_mixed_with_read_backs_and_single_iterations(S, O):
    # persistent steps between read-backs (which settle the pending exchange: the next launch must start from a closed state) and
    # single iterations: every hand-over of the plain state blocks between the two forms
    prob, opts = cm.serial_normal(N=64, T=60, ns=300, seed=2)
    h, o = _pair(S, O, prob, opts)
    h.step(7); o.step(7)
    cm.assert_state_equal(h.state(), o.state())          # flush in between
    h.step(1); o.step(1)
    h.step(20); o.step(20)
    cm.assert_history_equal(h.history(0, 28), o.history(0, 28))
    h.step(3); o.step(3)
    h.step(29); o.step(29)
    assert h.persistent_info()[1] >= 3 and h.persistent_info()[2] == 0
    cm.assert_history_equal(h.history(), o.history())
    cm.assert_state_equal(h.state(), o.state())


@pytest.mark.parametrize("ring,slow_us", [(2, 0), (2, 40), (4, 15)])
def test_persistent_form_under_skew_and_a_short_ring(S, O, monkeypatch, hooks, ring, slow_us):
    # the ring's overrun guard and every waiting path: a ring of 2 / 4 iterations instead of 8, and one tile's control wave idling
    # before each of its publications — the others wait for its slots, look again, and must not overwrite what it has not read
    monkeypatch.setenv("SMMHIP_PR_RING", str(ring))
    if slow_us:
        monkeypatch.setenv("SMMHIP_PR_SLOW_TILE", "3")
        monkeypatch.setenv("SMMHIP_PR_SLOW_US", str(slow_us))
    prob, opts = cm.serial_normal(N=400, T=120, ns=600, seed=21)
    h, o = _pair(S, O, prob, opts)
    h.step(120); o.step(120)
    assert h.persistent_info()[1] >= 1 and h.persistent_info()[2] == 0
    cm.assert_history_equal(h.history(), o.history())
    cm.assert_state_equal(h.state(), o.state())


def test_persistent_form_can_be_switched_off_and_reports_itself(S):
    prob, opts = cm.serial_normal(N=64, T=20, ns=300)
    h = S.hip_context(prob, opts)
    assert h.persistent_info()[0] is True
    h.set_persistent(False)
    assert h.persistent_info()[0] is False
    h.step(20)
    assert h.persistent_info()[1] == 0
    # one threshold > 0 for all chains (the reference's default, AlgoBGP.jl:522): the form on locally numbered cones (round 5,
    # tests/test_gpu_persist_loc.py); more than two moments: the tile form (tests/test_gpu_persist_tile.py); thresholds by chain (round 6): the same
    # forms, a threshold per slot position — unless one of them is negative
    p2, o2 = cm.serial_normal(N=64, T=20, ns=300, min_improve=0.05)
    assert S.hip_context(p2, o2).persistent_info()[0] is True
    p4, o4 = cm.serial_normal(N=64, T=20, ns=300, min_improve=np.linspace(0.0, 0.5, 64))
    assert S.hip_context(p4, o4).persistent_info()[0] is True
    p4, o4 = cm.serial_normal(N=64, T=20, ns=300, min_improve=np.linspace(-0.1, 0.5, 64))
    assert S.hip_context(p4, o4).persistent_info()[0] is False
    p3, o3 = cm.general_normal(4, N=32, T=10, ns=200)
    assert S.hip_context(p3, o3).persistent_info()[0] is True
    o3.min_improve[:] = np.linspace(0.0, 0.5, 32)
    assert S.hip_context(p3, o3).persistent_info()[0] is True
    o3.min_improve[:] = np.linspace(-0.1, 0.5, 32)
    assert S.hip_context(p3, o3).persistent_info()[0] is False


def test_persistent_form_hard_error_in_a_step_across_a_plan_window(S, O):
    # found by tools/soak.py (C2, iteration 110224 of bench.py's seed: a legitimate AlgoBGP.jl:409): the failing step starts with an
    # exchange still to be applied and its launches cross the end of a look-ahead window (256 iterations), so that the plan window has
    # moved on when the library rolls back to the snapshot — the replay must rebuild the window of the snapshot's iteration, not resolve
    # the pending exchange out of the wrong one (it read another window's plan: out-of-range at best, a memory fault at worst)
    N, T, tfail = 48, 400, 300
    prob, opts = cm.serial_normal(N=N, T=T, ns=100, sigma0=0.01)
    tab = cm.random_tables(prob, opts, tries=8)
    tab.prop_normals[tfail - 1] = 1e9
    h, o = _pair(S, O, prob, opts, tab)
    h.step_async(250); h.sync()          # (no read-back: the exchange of iteration 250 stays pending)
    o.step(250)
    with pytest.raises(A.SMMHipError) as eh:
        h.step_async(100); h.sync()      # 251 .. 350: window boundary at 256, failure at 300
    with pytest.raises(A.SMMHipError):
        o.step(100)
    assert eh.value.code == A.SMM_ERR_NO_DRAW_IN_SUPPORT and "iteration %d" % tfail in str(eh.value)
    assert h.persistent_info()[2] == 1 and h.state().iter == tfail
    hh, ho = h.history(0, T), o.history(0, T)
    for f in cm.INT_FIELDS:
        np.testing.assert_array_equal(getattr(hh, f)[:tfail - 1], getattr(ho, f)[:tfail - 1], err_msg=f)
    np.testing.assert_allclose(hh.value[:tfail - 1], ho.value[:tfail - 1], rtol=1e-9)


@pytest.mark.parametrize("ahead", [0, 3])
def test_hard_error_of_a_single_iteration_ahead_of_a_persistent_launch_on_the_stream(S, O, ahead):
    # found by tools/fuzz_errors.py (2 of 600 cases, round 5): AlgoBGP.jl:409 raised by a ONE-LAUNCH-PER-ITERATION kernel (the first iteration
    # behind a read-back) with a persistent launch enqueued right behind it, before anybody has looked at the error word.  That launch sees
    # the word at its entry and stores nothing — there is nothing to replay: the library used to roll back to the snapshot (taken behind the
    # failing iteration), clear the word and run on, and the error was LOST.  `ahead`: iterations between the failing one and the snapshot.
    N, T, tfail = 40, 60, 21
    prob, opts = cm.serial_normal(N=N, T=T, ns=200, sigma0=0.01)
    tab = cm.random_tables(prob, opts, tries=8)
    tab.prop_normals[tfail - 1] = 1e9
    h, o = _pair(S, O, prob, opts, tab)
    h.step(tfail - 1); o.step(tfail - 1)
    cm.assert_state_equal(h.state(), o.state())     # (the read-back settles the exchange: iteration tfail is a launch of its own)
    with pytest.raises(A.SMMHipError) as eh:
        h.step_async(1)                              # iteration tfail: fails on the device, nobody looks
        for _ in range(ahead):
            h.step_async(1)                          # ... further single launches: they see the word and store nothing
        h.step_async(8)                              # a persistent launch behind them (a snapshot is taken in front of it)
        h.step_async(5)
        h.sync()
    with pytest.raises(A.SMMHipError):
        o.step(20)
    assert eh.value.code == A.SMM_ERR_NO_DRAW_IN_SUPPORT and "iteration %d" % tfail in str(eh.value)
    assert h.state().iter == tfail
    hh, ho = h.history(0, T), o.history(0, T)
    for f in cm.INT_FIELDS:
        np.testing.assert_array_equal(getattr(hh, f)[:tfail - 1], getattr(ho, f)[:tfail - 1], err_msg=f)
    assert np.array_equal(hh.value[:tfail - 1], ho.value[:tfail - 1])
    assert np.isnan(hh.value[tfail:]).all() and (hh.status[tfail:] == 0).all()
    with pytest.raises(A.SMMHipError):
        h.step(1)   # sticky
