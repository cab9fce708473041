"""Parity of the HIP path (through the C ABI of libsmmhip.so) against the CPU oracle on the
same seeded inputs.  Bookkeeping bit-exact, floating point within 1e-9 relative (the
north star asks for 1e-6)."""
import os

import numpy as np
import pytest

import common as cm
from smm_jl_amd import _abi as A

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_pair(S, O, prob, opts, tables=None, **okw):
    """a HIP context and an oracle context fed with identical inputs.  The shock matrix Z (the
    seed-1234 draws of ObjExamples.jl:74-79) is injected randomness: when the caller gives none the
    library's default Z is read back and handed to the oracle (identical since round 5: the generator's functions are the contract's)."""
    h = S.hip_context(prob, opts, tables)
    t = tables if tables is not None else S.Tables()
    to = S.Tables(probs_acc=t.probs_acc, prop_normals=t.prop_normals, pairs=t.pairs, Z=h.Z())
    o = O.OracleContext(prob, opts, to, **okw)
    return h, o


def run_both(S, O, prob, opts, tables, T=None):
    T = opts.maxiter if T is None else T
    h, o = make_pair(S, O, prob, opts, tables)
    h.step(T); o.step(T)
    return h, o


def test_eval_batch_matches_oracle(S, O):
    prob, opts = cm.serial_normal(N=3, T=2)
    h, o = make_pair(S, O, prob, opts)
    assert np.array_equal(h.Z(), O.gen_Z(opts.seed, 2, prob.ns))   # same generator, the contract's own log / sine / cosine (include/smmhip.h)
    rng = np.random.default_rng(0)
    for M in (1, 7, 8, 9, 100):
        p = np.stack([rng.uniform(-3, 3, M), rng.uniform(-20, 20, M)])
        vh, mh, sh = h.eval_batch(p); vo, mo, so = o.eval_batch(p)
        assert np.array_equal(vh, vo) and np.array_equal(mh, mo) and np.array_equal(sh, so)


def test_eval_batch_analytic_anchor(S):
    # Z == 0  =>  value = mean(((mu-mom)/w)^2): serialNormal start -> 52.74 (ObjExamples.jl:90-101)
    prob, opts = cm.serial_normal(N=3, T=2)
    h = S.hip_context(prob, opts, S.Tables(Z=np.zeros((2, prob.ns))))
    v, m, s = h.eval_batch(np.array([[0.2], [-0.2]]))
    assert abs(v[0] - 52.74) < 1e-12 and np.allclose(m[:, 0], [0.2, -0.2], atol=1e-15) and s[0] == 1


@pytest.mark.parametrize("ns", [1, 63, 255, 256, 257, 1000, 10000])
def test_eval_batch_ragged_ns(S, O, ns):
    prob, opts = cm.serial_normal(N=3, T=2, ns=ns)
    h, o = make_pair(S, O, prob, opts)
    p = np.array([[0.2, -1.0, 2.5], [-0.2, 10.0, -19.0]])
    vh, mh, _ = h.eval_batch(p); vo, mo, _ = o.eval_batch(p)
    assert np.array_equal(vh, vo) and np.array_equal(mh, mo)


def test_c1_serial_normal_builtin_rng(S, O):
    # BASELINE config C1: serialNormal(2,200), 3 chains
    prob, opts = cm.serial_normal(N=3, T=200)
    h, o = run_both(S, O, prob, opts, None)
    cm.assert_history_equal(h.history(), o.history())
    cm.assert_state_equal(h.state(), o.state())


@pytest.mark.parametrize("N", [1, 2, 3, 5, 8, 9, 64, 100])
def test_injected_tables_exact(S, O, N):
    prob, opts = cm.serial_normal(N=N, T=40, ns=500)
    tab = cm.random_tables(prob, opts)
    h, o = run_both(S, O, prob, opts, tab)
    hh, ho = h.history(), o.history()
    cm.assert_history_equal(hh, ho, rtol=1e-12)
    # (tolerances from before round 5, when exp() was libm's / ocml's: the runs are bit-identical now, tests/test_gpu_bitexact.py)
    for f in ("value", "params", "sim_moments", "curr_val", "best_val"):
        assert np.array_equal(getattr(hh, f), getattr(ho, f), equal_nan=True), f
    cm.assert_state_equal(h.state(), o.state(), rtol=1e-12)


def test_chunked_steps_equal_one_call(S, O):
    prob, opts = cm.serial_normal(N=16, T=30, ns=300)
    h1, o = run_both(S, O, prob, opts, None)
    h2 = S.hip_context(prob, opts)
    for n in (1, 1, 5, 0, 13, 10):
        h2.step(n)
    cm.assert_history_equal(h1.history(), h2.history(), exact_floats=True)
    cm.assert_history_equal(h1.history(), o.history())


@pytest.mark.parametrize("npar,bs", [(4, None), (4, 1), (4, 2), (6, 3), (18, 1), (18, None), (5, 5), (64, 8)])
def test_general_dims_and_batches(S, O, npar, bs):
    prob, opts = cm.general_normal(npar, N=12, T=25, ns=300, batch_size=bs)
    h, o = run_both(S, O, prob, opts, None)
    cm.assert_history_equal(h.history(), o.history())
    cm.assert_state_equal(h.state(), o.state())


def test_failbox_status_minus2(S, O):
    # objective "exception" -> status -2, prob 0, rejected, value -1 recorded (mprob.jl:183-186, AlgoBGP.jl:336-338)
    prob, opts = cm.serial_normal(N=8, T=60, ns=200, objective_id=A.SMM_OBJ_NORM_FAILBOX, obj_params=[-0.2, 0.1])
    h, o = run_both(S, O, prob, opts, None)
    hh = h.history()
    assert (hh.status == -2).sum() > 0
    assert not hh.accepted[hh.status == -2].any() or (hh.exchanged[hh.status == -2] != 0).all()
    cm.assert_history_equal(hh, o.history())


def test_banana(S, O):
    from smm_jl_amd import Problem, BGPOpts
    npar, N, T = 10, 200, 40
    prob = Problem(init=np.zeros(npar), lb=-2 * np.ones(npar), ub=2 * np.ones(npar), mom=np.zeros(npar),
                   w=np.ones(npar), ns=1, objective_id=A.SMM_OBJ_BANANA)
    opts = BGPOpts(N=N, maxiter=T, sigma=0.02 * cm.temps(N, 5), acc_tuner=np.geomspace(20, 1, N), min_improve=np.zeros(N))
    h, o = run_both(S, O, prob, opts, None)
    cm.assert_history_equal(h.history(), o.history())


def test_error_negative_objective(S, O):
    # NaN data moment -> NaN value -> "AlgoBGP assumes ... non-negative" (AlgoBGP.jl:341)
    prob, opts = cm.serial_normal(N=4, T=5, ns=100, mom=(np.nan, 10.0))
    h, o = make_pair(S, O, prob, opts)
    with pytest.raises(A.SMMHipError) as eh:
        h.step(3)
    with pytest.raises(A.SMMHipError) as eo:
        o.step(3)
    assert eh.value.code == eo.value.code == A.SMM_ERR_NEGATIVE_OBJECTIVE


def test_error_no_draw_in_support(S, O):
    # mysample exhausts smpl_iters (AlgoBGP.jl:409)
    prob, opts = cm.serial_normal(N=4, T=5, ns=100, sigma0=1e6, smpl_iters=2)
    h, o = make_pair(S, O, prob, opts)
    with pytest.raises(A.SMMHipError) as eh:
        h.step(3)
    with pytest.raises(A.SMMHipError) as eo:
        o.step(3)
    assert eh.value.code == eo.value.code == A.SMM_ERR_NO_DRAW_IN_SUPPORT


def test_hard_error_stops_the_run_at_the_failing_iteration(S, O):
    # AlgoBGP.jl:409 aborts run! inside the failing iteration.  Here: the failing iteration completes, every later launch of
    # the same step sees the sticky error word and stores nothing, iter reports the failing iteration, the context refuses
    # to go on.  Injected normals throw every try of iteration 5 out of the box.
    N, T, tfail = 20, 12, 5
    prob, opts = cm.serial_normal(N=N, T=T, ns=200)
    tab = cm.random_tables(prob, opts, tries=3)
    tab.prop_normals[tfail - 1] = 1e9
    h, o = make_pair(S, O, prob, opts, tab)
    with pytest.raises(A.SMMHipError) as eh:
        h.step(9)
    with pytest.raises(A.SMMHipError) as eo:
        o.step(9)
    assert eh.value.code == eo.value.code == A.SMM_ERR_NO_DRAW_IN_SUPPORT
    assert "iteration %d" % tfail in str(eh.value) and "chain 1," in str(eh.value)   # the first failing chain, deterministically
    st = h.state()
    assert st.iter == tfail and o.state().iter == tfail - 1   # (the oracle, like the reference, never finishes the failing iteration)
    hh = h.history(0, T)
    ho = o.history(0, T)
    for f in cm.INT_FIELDS:   # iterations before the failing one: complete and equal (incl. their exchanges)
        np.testing.assert_array_equal(getattr(hh, f)[:tfail - 1], getattr(ho, f)[:tfail - 1], err_msg=f)
    for f in cm.F64_FIELDS:
        np.testing.assert_allclose(getattr(hh, f)[:tfail - 1], getattr(ho, f)[:tfail - 1], rtol=1e-12, err_msg=f)
    # iterations after the failing one: untouched (the constructor's fill)
    assert np.isnan(hh.value[tfail:]).all() and (hh.status[tfail:] == 0).all() and (hh.accepted[tfail:] == 0).all()
    assert (hh.best_id[tfail:] == -1).all() and (hh.exchanged[tfail:] == 0).all() and np.isinf(hh.curr_val[tfail:]).all()
    assert (hh.exchanged[tfail - 1] == 0).all()           # exchangeMoves! of the failing iteration never ran
    for call in (lambda: h.step(1), h.local_step):           # sticky
        with pytest.raises(A.SMMHipError) as e2:
            call()
        assert e2.value.code == A.SMM_ERR_NO_DRAW_IN_SUPPORT
    # negative / NaN objective (AlgoBGP.jl:341) through the general kernel as well (np = 4)
    prob4, opts4 = cm.general_normal(4, N=6, T=8, ns=100)
    prob4.mom[1] = np.nan
    g = S.hip_context(prob4, opts4)
    with pytest.raises(A.SMMHipError) as e4:
        g.step(6)
    assert e4.value.code == A.SMM_ERR_NEGATIVE_OBJECTIVE and g.state().iter == 2
    assert np.isnan(g.history(0, 8).value[2:]).all()


def test_three_phase_calls_after_step(S, O):
    # smm_bgp_step leaves the exchange of its last iteration to the next chain kernel; local_step / export
    # must settle it first.  step(5) then three iterations through the three-phase calls == step(8).
    import torch
    prob, opts = cm.serial_normal(N=48, T=8, ns=300)
    a, o = run_both(S, O, prob, opts, None)
    b = S.hip_context(prob, opts)
    b.step(5)
    buf = torch.empty((48, b.record_doubles()), dtype=torch.float64, device="cuda")
    for _ in range(3):
        b.local_step()
        b.export_records_dev(buf.data_ptr())
        b.sync()
        b.exchange_dev(buf.data_ptr())
        b.sync()
    cm.assert_history_equal(a.history(), b.history(), exact_floats=True)
    cm.assert_history_equal(b.history(), o.history())
    # and the export alone after a step sees the records AFTER that step's last exchange
    c = S.hip_context(prob, opts)
    c.step(5)
    c.export_records_dev(buf.data_ptr())
    c.sync()
    o5 = O.OracleContext(prob, opts, S.Tables(Z=c.Z()))
    o5.step(5)
    np.testing.assert_allclose(buf.cpu().numpy()[:, 0], o5.state().la_value, rtol=1e-9)


@pytest.mark.parametrize("npar,N,T", [(2, 5, 30), (2, 16, 30), (2, 17, 25), (2, 100, 40), (2, 4096, 12), (1, 37, 30), (2, 5000, 8),
                                      (3, 50, 30), (4, 100, 30), (3, 4096, 10), (4, 6000, 8)])
def test_norm_kernel_equals_general_kernel(S, O, npar, N, T, monkeypatch, hooks):
    # k_chain_iter_norm (16-chain tiles, np == nm <= 2) against the general k_chain_iter on the same problem: bit-identical
    if npar == 2:
        prob, opts = cm.serial_normal(N=N, T=T, ns=1000 if N > 1000 else 10000, objective_id=A.SMM_OBJ_NORM_FAILBOX,
                                      obj_params=[0.5, 0.9], sigma0=0.2)
    else:
        prob, opts = cm.general_normal(npar, N=N, T=T, ns=777 if N < 1000 else 2000)
    a = S.hip_context(prob, opts)
    a.step(T)
    monkeypatch.setenv("SMMHIP_NORM_FAST", "0")
    b = S.hip_context(prob, opts)
    b.step(T)
    monkeypatch.delenv("SMMHIP_NORM_FAST")
    cm.assert_history_equal(a.history(), b.history(), exact_floats=True)
    cm.assert_state_equal(a.state(), b.state(), rtol=0)
    hh = a.history()
    assert (hh.exchanged != 0).any() or N == 1
    if npar == 2:
        assert (hh.status == -2).any()
    if N <= 100:
        o = O.OracleContext(prob, opts, S.Tables(Z=a.Z()))
        o.step(T)
        cm.assert_history_equal(hh, o.history())
        cm.assert_state_equal(a.state(), o.state())


@pytest.mark.parametrize("tries", [1, 3, 5, 24])
def test_norm_kernel_injected_tries(S, O, tries):
    # the try groups of the 4-lane proposal (tries 0-3, 4-7, later ones from memory; exhausted tables are a hard error)
    prob, opts = cm.serial_normal(N=40, T=30, ns=300, sigma0=0.6, smpl_iters=50)
    tab = cm.random_tables(prob, opts, tries=tries)
    h, o = make_pair(S, O, prob, opts, tab)
    eh = eo = None
    try:
        h.step(30)
    except A.SMMHipError as e:
        eh = e
    try:
        o.step(30)
    except A.SMMHipError as e:
        eo = e
    assert (eh is None) == (eo is None)
    if eh is not None:
        assert eh.code == eo.code == A.SMM_ERR_NO_DRAW_IN_SUPPORT
        assert h.state().iter == o.state().iter + 1
    else:
        cm.assert_history_equal(h.history(), o.history(), rtol=1e-12)
    assert tries > 3 or eh is not None       # sigma0 = 0.6: three tries do not last 30 iterations x 40 chains


def test_norm_kernel_rng_tries_beyond_the_pregenerated(S, O):
    # sigma so large that chains regularly need more than the 8 pre-generated tries: the in-kernel generator path
    prob, opts = cm.serial_normal(N=33, T=25, ns=100, sigma0=1.5, smpl_iters=100000)
    h, o = run_both(S, O, prob, opts, None)
    cm.assert_history_equal(h.history(), o.history())
    cm.assert_state_equal(h.state(), o.state())


def test_maxiter_guard(S):
    prob, opts = cm.serial_normal(N=3, T=4, ns=100)
    h = S.hip_context(prob, opts)
    h.step(4)
    with pytest.raises(A.SMMHipError) as e:
        h.step(1)
    assert e.value.code == A.SMM_ERR_MAXITER


def test_bad_batch_rejected(S):
    prob, opts = cm.general_normal(4, N=3, T=4, batch_size=3)
    with pytest.raises(A.SMMHipError) as e:
        S.hip_context(prob, opts)
    assert e.value.code == A.SMM_ERR_BAD_BATCH


def test_state_roundtrip_restart(S, O):
    # save / readMalgo / restart! (AlgoAbstract.jl:83-102, AlgoBGP.jl:804-884): stop after 12, resume in a NEW ctx
    prob, opts = cm.serial_normal(N=10, T=30, ns=300)
    full, o = run_both(S, O, prob, opts, None)
    a = S.hip_context(prob, opts); a.step(12)
    st, hist = a.state(), a.history()
    b = S.hip_context(prob, opts)
    b.set_state(st, hist)
    b.step(18)
    cm.assert_history_equal(full.history(), b.history(), exact_floats=True)
    cm.assert_state_equal(full.state(), b.state(), rtol=0)


def sharded_run(S, prob, opts_full, G, T):
    """G contexts on one GPU emulate G ranks; the all-gather is a host concatenation here
    (the RCCL form is exercised by bench.py --gpus N and tests/test_dist_gloo.py)."""
    import torch
    from smm_jl_amd import BGPOpts
    N = opts_full.N_global // G
    ctxs = []
    for r in range(G):
        o = BGPOpts(N=N, maxiter=opts_full.maxiter, sigma=opts_full.sigma, acc_tuner=opts_full.acc_tuner,
                    min_improve=opts_full.min_improve, sigma_update_steps=opts_full.sigma_update_steps,
                    sigma_adjust_by=opts_full.sigma_adjust_by, smpl_iters=opts_full.smpl_iters,
                    batch_size=opts_full.batch_size, seed=opts_full.seed, chain_offset=r * N,
                    N_global=opts_full.N_global, dist_fun=opts_full.dist_fun, chol_L=opts_full.chol_L)
        ctxs.append(S.hip_context(prob, o))
    R = ctxs[0].record_doubles()
    gathered = torch.empty((G, N, R), dtype=torch.float64, device="cuda")
    for _ in range(T):
        for r, c in enumerate(ctxs):
            c.local_step()
            c.export_records_dev(gathered[r].data_ptr())
        for c in ctxs:
            c.sync()
        for c in ctxs:
            c.exchange_dev(gathered.data_ptr())
        for c in ctxs:
            c.sync()
    return ctxs


def sharded_run_fused(S, prob, opts_full, G, T, finish_every=None):
    """the two-enqueue form (smm_bgp_sharded_step): all shards write their slices of the same gather buffer, which
    stands in for the in-place all-gather; two buffers alternate"""
    import torch
    from smm_jl_amd import BGPOpts
    N = opts_full.N_global // G
    ctxs = []
    for r in range(G):
        o = BGPOpts(N=N, maxiter=opts_full.maxiter, sigma=opts_full.sigma, acc_tuner=opts_full.acc_tuner,
                    min_improve=opts_full.min_improve, sigma_update_steps=opts_full.sigma_update_steps,
                    sigma_adjust_by=opts_full.sigma_adjust_by, smpl_iters=opts_full.smpl_iters,
                    batch_size=opts_full.batch_size, seed=opts_full.seed, chain_offset=r * N,
                    N_global=opts_full.N_global, dist_fun=opts_full.dist_fun, chol_L=opts_full.chol_L)
        ctxs.append(S.hip_context(prob, o))
    R = ctxs[0].record_doubles()
    bufs = [torch.zeros((G, N, R), dtype=torch.float64, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()   # (the fills run on torch's stream, the contexts on their own)
    cur = None
    for it in range(T):
        nxt = 0 if cur is None else cur ^ 1
        for c in ctxs:
            c.sharded_step(bufs[cur].data_ptr() if cur is not None else 0, bufs[nxt].data_ptr())
        for c in ctxs:
            c.sync()
        cur = nxt
        if finish_every and (it + 1) % finish_every == 0 and it + 1 < T:
            for c in ctxs:
                c.sharded_finish(bufs[cur].data_ptr())
                c.sync()
            cur = None
    for c in ctxs:
        c.sharded_finish(bufs[cur].data_ptr() if cur is not None else 0)
        c.sync()
    return ctxs


@pytest.mark.parametrize("G,N,T,fe", [(2, 32, 25, None), (4, 32, 25, 7), (2, 9000, 6, None), (4, 600, 300, 97)])
def test_fused_sharded_equals_single(S, O, G, N, T, fe):
    # 9000: above the LDS kernels (global-memory level walk); 600 x 300: crosses a look-ahead window with an open exchange
    N -= N % G
    prob, opts = cm.serial_normal(N=N, T=T, ns=64 if N > 100 else 300)
    single = S.hip_context(prob, opts)
    single.step(T)
    ctxs = sharded_run_fused(S, prob, opts, G, T, finish_every=fe)
    hs = single.history()
    n = N // G
    for r, c in enumerate(ctxs):
        hr = c.history()
        for f in A.HistoryBuffers.FIELDS:
            assert np.array_equal(getattr(hr, f), getattr(hs, f)[..., r * n:(r + 1) * n], equal_nan=True), (f, r)
        st, ss = c.state(), single.state()
        assert np.array_equal(st.sigma, ss.sigma[r * n:(r + 1) * n]) and np.array_equal(st.accept_rate, ss.accept_rate[r * n:(r + 1) * n])
    assert (hs.exchanged != 0).any()
    if N <= 100:
        o = O.OracleContext(prob, opts, S.Tables(Z=single.Z()))
        o.step(T)
        cm.assert_history_equal(hs, o.history())


@pytest.mark.parametrize("G", [2, 4])
def test_sharded_equals_single(S, O, G):
    prob, opts = cm.serial_normal(N=32, T=25, ns=300)
    single, o = run_both(S, O, prob, opts, None)
    ctxs = sharded_run(S, prob, opts, G, 25)
    hs = single.history()
    n = 32 // G
    for r, c in enumerate(ctxs):
        hr = c.history()
        for f in A.HistoryBuffers.FIELDS:
            assert np.array_equal(getattr(hr, f), getattr(hs, f)[..., r * n:(r + 1) * n], equal_nan=True), (f, r)
    cm.assert_history_equal(hs, o.history())


def test_c2_full_size_against_oracle(S, O):
    # BASELINE config C2 (headline): 4096 chains x 200 iterations, ns = 10000
    prob, opts = cm.serial_normal(N=4096, T=200)
    h, o = make_pair(S, O, prob, opts, threads=min(O.max_threads(), len(__import__("os").sched_getaffinity(0))))
    h.step(200)
    o.step(200)
    hh, ho = h.history(), o.history()
    cm.assert_history_equal(hh, ho)
    cm.assert_state_equal(h.state(), o.state())
    # size independent properties on the full run
    assert hh.accepted[0].all() and (hh.prob[0] == 1).all() and (hh.best_id[0] == 1).all()
    assert (np.diff(hh.best_val, axis=0) <= 0).all()
    ex = hh.exchanged
    t, c = np.nonzero(ex)
    assert np.array_equal(ex[t, ex[t, c] - 1] != 0, np.ones(len(t), bool))  # partners are marked too
    assert (ex[0] == 0).all()  # no exchange in iteration 1 (AlgoBGP.jl:637)


@pytest.mark.parametrize("N", [2, 3, 50, 1000])
def test_any_size_exchange_kernel_matches(S, O, N, monkeypatch, hooks):
    # the barrier-round resolution kernel (used above N_global = 8192) against the LDS data-flow one
    prob, opts = cm.serial_normal(N=N, T=30, ns=200)
    a, o = run_both(S, O, prob, opts, None)
    monkeypatch.setenv("SMMHIP_ANY_EXCHANGE", "1")
    b = S.hip_context(prob, opts)
    b.step(30)
    cm.assert_history_equal(a.history(), b.history(), exact_floats=True)
    cm.assert_history_equal(a.history(), o.history())


@pytest.mark.parametrize("N", [2, 3, 50, 1000, 4096])
def test_inline_walk_equals_resolve_kernel(S, O, N, monkeypatch, hooks):
    # default single-shard path: the exchange walk runs in the prologue of the next chain kernel (every tile,
    # redundantly); against the stand-alone k_exch_resolve_lvl kernel and the oracle.  Non-uniform thresholds
    # (the plan's per-pair column instead of one scalar) and stepping in uneven pieces (an unresolved exchange
    # crosses smm_bgp_step calls; history reads in between force the stand-alone resolution).
    rng = np.random.default_rng(N)
    for uniform in (True, False):
        mi = 0.0 if uniform else rng.uniform(-0.2, 0.4, N)
        prob, opts = cm.serial_normal(N=N, T=24, ns=64, min_improve=mi)
        a, o = make_pair(S, O, prob, opts, None)
        for n in (1, 1, 2, 3, 5, 12):
            a.step(n)
            if n == 3:
                a.history()       # flush: resolves the open exchange with the stand-alone kernel
        o.step(24)
        monkeypatch.setenv("SMMHIP_INLINE_WALK", "0")
        b = S.hip_context(prob, opts)
        monkeypatch.delenv("SMMHIP_INLINE_WALK")
        b.step(24)
        ha = a.history()
        cm.assert_history_equal(ha, b.history(), exact_floats=True)
        cm.assert_history_equal(ha, o.history())
        cm.assert_state_equal(a.state(), o.state())
        if N > 3:
            assert (ha.exchanged != 0).any()


@pytest.mark.parametrize("N", [9, 24, 50, 1000])
@pytest.mark.parametrize("banana", [False, True])
def test_two_tiles_per_workgroup_forced(S, O, N, banana, monkeypatch, hooks):
    # two tiles per workgroup (one inline exchange walk per CU) is chosen above 2048 chains; force it at small and odd tile counts
    monkeypatch.setenv("SMMHIP_TPW", "2")
    if banana:
        prob = S.Problem(init=np.zeros(4), lb=-2 * np.ones(4), ub=2 * np.ones(4), mom=np.zeros(4), w=np.ones(4), ns=1,
                         objective_id=A.SMM_OBJ_BANANA)
        opts = S.BGPOpts(N=N, maxiter=30, sigma=0.05 * cm.temps(N, 4), acc_tuner=np.geomspace(2.0, 0.1, N), min_improve=np.zeros(N),
                         N_global=N, seed=3)
    else:
        prob, opts = cm.serial_normal(N=N, T=30, ns=200)
    h, o = run_both(S, O, prob, opts, None)
    cm.assert_history_equal(h.history(), o.history())
    cm.assert_state_equal(h.state(), o.state())
    assert banana or (h.history().exchanged != 0).any()


@pytest.mark.parametrize("N", [2, 3, 50, 1000, 4096, 6000])
def test_dataflow_exchange_kernel_matches(S, O, N, monkeypatch, hooks):
    # the ticket (data-flow) resolution kernel against the level-synchronous ones (16-byte chain slots up to
    # N_global = 4096, split slots up to 8192: N = 6000)
    prob, opts = cm.serial_normal(N=N, T=12, ns=64)
    a, o = run_both(S, O, prob, opts, None)
    monkeypatch.setenv("SMMHIP_DATAFLOW_EXCHANGE", "1")
    b = S.hip_context(prob, opts)
    b.step(12)
    cm.assert_history_equal(a.history(), b.history(), exact_floats=True)
    cm.assert_history_equal(a.history(), o.history())


@pytest.mark.parametrize("N", [2, 3, 50, 1000, 4096])
def test_big_exchange_kernels_match(S, O, N, monkeypatch, hooks):
    # the global-memory level plan + walk (8192 < N_global <= 65535) forced at small sizes
    prob, opts = cm.serial_normal(N=N, T=12, ns=64)
    a, o = run_both(S, O, prob, opts, None)
    monkeypatch.setenv("SMMHIP_BIG_EXCHANGE", "1")
    b = S.hip_context(prob, opts)
    b.step(12)
    cm.assert_history_equal(a.history(), b.history(), exact_floats=True)
    cm.assert_history_equal(a.history(), o.history())


def test_n_global_above_8192(S, O):
    # 8-GPU-sized population resolved on one GPU: 20000 chains
    prob, opts = cm.serial_normal(N=20000, T=5, ns=32)
    h, o = make_pair(S, O, prob, opts, threads=8)
    h.step(5); o.step(5)
    cm.assert_history_equal(h.history(), o.history())
    assert (h.history().exchanged != 0).sum() > 0


def test_c3_population_8_temperature_levels(S, O):
    # BASELINE config 3 shape on one GPU: 32768 chains = 8 temperature levels x 4096 replicas
    # (sigma = 0.05 * linspace(1,5,8)[level], acc_tuner log-spaced 20 -> 1 per level)
    from smm_jl_amd import BGPOpts
    L, R_, T = 8, 4096, 4
    prob, _ = cm.serial_normal(N=3, T=T, ns=32)
    opts = BGPOpts(N=L * R_, maxiter=T, sigma=np.repeat(0.05 * np.linspace(1, 5, L), R_),
                   acc_tuner=np.repeat(np.geomspace(20, 1, L), R_), min_improve=np.zeros(L * R_))
    h, o = make_pair(S, O, prob, opts, threads=8)
    h.step(T); o.step(T)
    hh = h.history()
    cm.assert_history_equal(hh, o.history())
    lev = (np.nonzero(hh.exchanged)[1] // R_, (hh.exchanged[hh.exchanged != 0] - 1) // R_)
    assert (lev[0] != lev[1]).any()  # exchanges between temperature levels happened


def _all_cores(O):
    import os
    return max(1, min(O.max_threads(), len(os.sched_getaffinity(0))))


def c3_opts(L=8, R_=4096, T=50):
    """BASELINE config 3: 8 temperature levels x 4096 replicas"""
    from smm_jl_amd import BGPOpts
    return BGPOpts(N=L * R_, maxiter=T, sigma=np.repeat(0.05 * np.linspace(1, 5, L), R_),
                   acc_tuner=np.repeat(np.geomspace(20, 1, L), R_), min_improve=np.zeros(L * R_))


def test_c3_real_workload_32768_chains_ns10000(S, O):
    # BASELINE config 3 on its own workload: 32768 chains, ns = 10000, 50 iterations, whole history
    # against the oracle -- the big-population plan / walk kernels with the real objective behind them
    T = 50
    prob, _ = cm.serial_normal(N=3, T=T, ns=10000)
    opts = c3_opts(T=T)
    h, o = make_pair(S, O, prob, opts, threads=_all_cores(O))
    h.step(T); o.step(T)
    hh = h.history()
    # (atol: a simulated moment is a mean of O(1) draws and can come out at 1e-7; a one-ulp difference of the proposal --
    # the generator's sincos/log before round 5 -- was 1e-16 absolute there; bit-identical now: tests/test_gpu_bitexact.py)
    cm.assert_history_equal(hh, o.history(), atol=1e-13)
    cm.assert_state_equal(h.state(), o.state(), atol=1e-13)
    ex = hh.exchanged
    assert (ex[0] == 0).all() and (ex[1:] != 0).mean() > 0.05
    t, c = np.nonzero(ex)
    assert (ex[t, ex[t, c] - 1] != 0).all()                      # partners are marked too
    assert (np.diff(hh.best_val, axis=0) <= 0).all()
    lev = (c // 4096, (ex[t, c] - 1) // 4096)
    assert (lev[0] != lev[1]).any()                              # exchanges between temperature levels happened


def test_n20000_real_workload_ns10000(S, O):
    # 8192 < N_global <= 65535 with the real objective: 20000 chains, ns = 10000, 25 iterations
    prob, opts = cm.serial_normal(N=20000, T=25, ns=10000)
    h, o = make_pair(S, O, prob, opts, threads=_all_cores(O))
    h.step(25); o.step(25)
    cm.assert_history_equal(h.history(), o.history(), atol=1e-13)
    cm.assert_state_equal(h.state(), o.state(), atol=1e-13)
    assert (h.history().exchanged != 0).sum() > 0


def test_c3_fused_sharded_8x4096_real_workload(S, O):
    # the 8-GPU form of C3 emulated on one GPU: 8 shards x 4096 chains, ns = 10000, through smm_bgp_sharded_step; 70
    # iterations cross a look-ahead window (60 iterations at this size) with an exchange open.  Every shard against
    # the oracle's single-population run.
    T, G, n = 70, 8, 4096
    prob, _ = cm.serial_normal(N=3, T=T, ns=10000)
    opts = c3_opts(T=T)
    ctxs = sharded_run_fused(S, prob, opts, G, T)
    o = O.OracleContext(prob, opts, S.Tables(Z=ctxs[0].Z()), threads=_all_cores(O))
    o.step(T)
    ho, so = o.history(), o.state()
    for r, c in enumerate(ctxs):
        hr, sr = c.history(), c.state()
        sl = slice(r * n, (r + 1) * n)
        for f in cm.INT_FIELDS:
            assert np.array_equal(getattr(hr, f), getattr(ho, f)[..., sl]), (f, r)
        for f in cm.F64_FIELDS:
            np.testing.assert_allclose(getattr(hr, f), getattr(ho, f)[..., sl], rtol=1e-9, atol=1e-13, equal_nan=True, err_msg="%s shard %d" % (f, r))
        assert np.array_equal(sr.n_noex, so.n_noex[sl]) and np.array_equal(sr.n_acc_noex, so.n_acc_noex[sl])
        np.testing.assert_allclose(sr.sigma, so.sigma[sl], rtol=1e-12)
        np.testing.assert_allclose(sr.la_value, so.la_value[sl], rtol=1e-9, atol=1e-13)
    assert (ho.exchanged != 0).mean() > 0.05


def test_exchange_worst_case_star_pairs(S, O, hooks):
    # injected pair list in which every pair touches chain 0: dependency depth == number of pairs
    N, T = 40, 6
    prob, opts = cm.serial_normal(N=N, T=T, ns=64, acc_tuners=np.ones(N), min_improve=0.0)
    tab = cm.random_tables(prob, opts)
    tab.pairs[:, :, 0] = 0
    tab.pairs[:, :, 1] = 1 + (np.arange(N)[None, :] * 7 + np.arange(T)[:, None]) % (N - 1)
    tab.probs_acc[:] *= 0.1
    h, o = run_both(S, O, prob, opts, tab)
    assert (h.history().exchanged != 0).sum() > 0
    cm.assert_history_equal(h.history(), o.history(), rtol=1e-12)
    for env in ("SMMHIP_BIG_EXCHANGE", "SMMHIP_DATAFLOW_EXCHANGE", "SMMHIP_ANY_EXCHANGE"):
        with pytest.MonkeyPatch.context() as mp:
            mp.setenv(env, "1")
            b = S.hip_context(prob, opts, tab)
            b.step(T)
            cm.assert_history_equal(h.history(), b.history(), exact_floats=True)


def test_big_plan_with_more_levels_than_its_lds_histogram(S, monkeypatch, hooks):
    # k_exch_plan_big counts the pairs of the first 1024 levels in LDS and the rest in global memory: an injected list in which every
    # pair touches chain 0 (1500 pairs, 1500 levels) through the global-memory level plan and walk, against the default kernels
    N, T = 1500, 5
    prob, opts = cm.serial_normal(N=N, T=T, ns=64, acc_tuners=np.ones(N), min_improve=0.0)
    tab = cm.random_tables(prob, opts)
    tab.pairs[:, :, 0] = 0
    tab.pairs[:, :, 1] = 1 + (np.arange(N)[None, :] * 7 + np.arange(T)[:, None]) % (N - 1)
    a = S.hip_context(prob, opts, tab)
    a.step(T)
    monkeypatch.setenv("SMMHIP_BIG_EXCHANGE", "1")
    b = S.hip_context(prob, opts, tab)
    b.step(T)
    assert (a.history().exchanged != 0).any()
    cm.assert_history_equal(a.history(), b.history(), exact_floats=True)


def test_rows_fallback_with_slots_from_the_accept_step(S, O):
    # a single shard of more than 8192 chains takes the initial slots of k_exch_resolve_rows from its accept step (no k_exch_keys
    # pre-pass); where an iteration's plan does not fit the rows form — here: an injected pair list 40 levels deep — the kernel falls
    # back to the key walk, whose 16-bit slots it then makes itself
    N, T = 9000, 5
    prob, opts = cm.serial_normal(N=N, T=T, ns=32, min_improve=0.0)
    tab = cm.random_tables(prob, opts, tries=24)
    tab.pairs[1::2, :40, 0] = 0                       # odd iterations: forty pairs through chain 0, one after the other
    tab.pairs[1::2, :40, 1] = 1 + np.arange(40)[None, :] * 3
    h, o = run_both(S, O, prob, opts, tab)
    assert (h.history().exchanged != 0).sum() > 0
    cm.assert_history_equal(h.history(), o.history(), rtol=1e-12)
    cm.assert_state_equal(h.state(), o.state(), rtol=1e-12)


def test_n_global_between_4096_and_8192(S, O):
    prob, opts = cm.serial_normal(N=5000, T=6, ns=32)
    h, o = make_pair(S, O, prob, opts, threads=8)
    h.step(6); o.step(6)
    cm.assert_history_equal(h.history(), o.history())


def test_c4_banana_8192_chains(S, O):
    # BASELINE config 4: banana objective generalised to 10 params / 10 moments, 8192 chains, 1 GPU
    from smm_jl_amd import Problem, BGPOpts
    npar, N, T = 10, 8192, 60
    prob = Problem(init=np.zeros(npar), lb=-2 * np.ones(npar), ub=2 * np.ones(npar), mom=np.zeros(npar),
                   w=np.ones(npar), ns=1, objective_id=A.SMM_OBJ_BANANA)
    # min_improve < 0: exchange also when chain i gets (slightly) worse, AlgoBGP.jl:682-688
    opts = BGPOpts(N=N, maxiter=T, sigma=0.02 * cm.temps(N, 5), acc_tuner=np.geomspace(20, 1, N), min_improve=-0.05 * np.ones(N))
    h, o = run_both(S, O, prob, opts, None)
    hh = h.history()
    cm.assert_history_equal(hh, o.history())
    cm.assert_state_equal(h.state(), o.state())
    assert (hh.exchanged != 0).mean() > 0.01 and (np.diff(hh.best_val, axis=0) <= 0).all()


@pytest.mark.parametrize("npar,N", [(2, 37), (2, 1000), (1, 50), (3, 200), (4, 64)])
def test_narrow_norm_kernel_forced(S, O, monkeypatch, hooks, npar, N):
    # k_chain_iter_norm_narrow (workgroups of one half of 512 lanes, the tile's moments one after the other: what shards of more
    # than one round of tiles run, e.g. C3's 32768 chains on one GPU) forced at small and ragged populations, every np
    monkeypatch.setenv("SMMHIP_NORM_NARROW", "1")
    monkeypatch.setenv("SMMHIP_INLINE_WALK", "0")   # (the narrow kernel is the one without the walk)
    if npar == 2:
        prob, opts = cm.serial_normal(N=N, T=30, ns=300)
    else:
        prob, opts = cm.general_normal(npar, N, 30, ns=300, batch_size=npar)
    h, o = run_both(S, O, prob, opts, None)
    cm.assert_history_equal(h.history(), o.history())
    cm.assert_state_equal(h.state(), o.state())
    assert (h.history().exchanged != 0).any()


def banana10(S, N, T, mi=0.0, seed=3):
    # (started away from the optimum: the hotter chains, with their larger steps, get ahead of the colder ones, so that
    # `value_i - value_j > 0` — the exchange test at min_improve == 0 — is true for many pairs)
    npar = 10
    prob = S.Problem(init=np.full(npar, 1.2), lb=-2 * np.ones(npar), ub=2 * np.ones(npar), mom=np.zeros(npar), w=np.ones(npar), ns=1,
                     objective_id=A.SMM_OBJ_BANANA)
    opts = S.BGPOpts(N=N, maxiter=T, sigma=0.02 * cm.temps(N, 5), acc_tuner=np.geomspace(20, 1, N), min_improve=mi * np.ones(N), seed=seed)
    return prob, opts


@pytest.mark.parametrize("N,T", [(8192, 40), (5024, 30), (6000, 30), (4128, 300)])
def test_c4_key_form_against_oracle(S, O, N, T):
    # BASELINE config 4 as bench.py runs it (min_improve == 0): ONE launch per iteration, k_chain_iter<0, 16, 2, true> with the key walk
    # in its prologue — every workgroup walks its own cone of the pair list (smm_cone.hpp) where the population is whole
    # workgroups of 32 chains (8192, 5024, 4128; 6000 is not: the whole list in every workgroup); 300 iterations cross a window
    prob, opts = banana10(S, N, T)
    h, o = run_both(S, O, prob, opts, None)
    hh = h.history()
    cm.assert_history_equal(hh, o.history(), atol=1e-12)   # (parameters pass through 0: lb + x (ub - lb) cancels)
    cm.assert_state_equal(h.state(), o.state(), atol=1e-12)
    assert (hh.exchanged != 0).mean() > 0.01


def test_c4_cones_equal_the_whole_walk(S, monkeypatch, hooks):
    prob, opts = banana10(S, 8192, 30)
    a = S.hip_context(prob, opts)
    a.step(30)
    monkeypatch.setenv("SMMHIP_NO_CONE", "1")
    b = S.hip_context(prob, opts)
    b.step(30)
    cm.assert_history_equal(a.history(), b.history(), exact_floats=True)


def test_c4_a_cone_that_does_not_fit_falls_back(S, O):
    # a pair list of 31 levels in which the cone of workgroup 0 is 32, 64, 100, 100, ... pairs wide going back from the last level:
    # about 60 sub-levels of 64, more than the 32 a cone may have — the plan kernel says so (cone_ok = 0) and every workgroup of that
    # iteration walks the whole list.  Even iterations carry the random list (cones).
    N, T = 8192, 6
    prob, opts = banana10(S, N, T)
    tab = cm.random_tables(prob, opts, tries=24)
    levels, fresh, frontier = [], 32, list(range(32))
    for back in range(31):   # from the last level backwards
        lv = []
        for c in frontier[:100]:
            lv.append((c, fresh)); fresh += 1
        frontier = frontier + [p[1] for p in lv]
        levels.append(lv)
    pairs = [p for lv in reversed(levels) for p in lv]
    rest = list(range(fresh, N))   # the other chains among themselves, round after round (a few levels deep, no chain of the cone)
    half, rnd = len(rest) // 2, 0
    while len(pairs) < N:
        for x in range(half):
            if len(pairs) < N:
                pairs.append((rest[x], rest[half + (x + rnd) % half]))
        rnd += 1
    pt = np.array(pairs[:N], np.int32)
    pt = np.stack([pt.min(1), pt.max(1)], 1)
    tab.pairs[1::2] = pt[None]
    h, o = run_both(S, O, prob, opts, tab)
    assert (h.history().exchanged != 0).any()
    cm.assert_history_equal(h.history(), o.history(), atol=1e-12)


def test_window_boundaries(S, O):
    # look-ahead tables are produced window by window (256 iterations): cross two boundaries
    prob, opts = cm.serial_normal(N=40, T=600, ns=64)
    h, o = run_both(S, O, prob, opts, None)
    cm.assert_history_equal(h.history(), o.history())
    cm.assert_state_equal(h.state(), o.state())


def dense_problem(S, O, npar, nm, N, T, seed=3, explicit=True, **kw):
    from smm_jl_amd import Problem, BGPOpts
    rng = np.random.default_rng(seed)
    objp = None
    if explicit:
        objp = np.concatenate([rng.standard_normal(A.SMM_DENSE_D * npar) / np.sqrt(npar),
                               rng.standard_normal(nm * A.SMM_DENSE_D) / np.sqrt(A.SMM_DENSE_D)])
    prob = Problem(init=rng.uniform(-0.3, 0.3, npar), lb=-np.ones(npar), ub=np.ones(npar),
                   mom=rng.uniform(-0.5, 0.5, nm), w=rng.uniform(0.5, 2.0, nm), ns=1,
                   objective_id=A.SMM_OBJ_DENSE, obj_params=objp)
    opts = BGPOpts(N=kw.pop("N_local", N), maxiter=T, sigma=0.02 * cm.temps(N, 4), acc_tuner=np.geomspace(20, 1, N) if N > 1 else [2.0],
                   min_improve=np.zeros(N), N_global=N, seed=seed, **kw)
    return prob, opts


@pytest.mark.parametrize("npar,nm", [(1, 1), (3, 2), (6, 5), (17, 33), (50, 50), (64, 64)])
def test_dense_objective_eval_batch(S, O, npar, nm):
    # FP64 MFMA path against the oracle's fma chains (same summation order by contract)
    prob, opts = dense_problem(S, O, npar, nm, N=4, T=2)
    h, o = make_pair(S, O, prob, opts)
    rng = np.random.default_rng(1)
    for M in (1, 15, 16, 17, 200):
        p = rng.uniform(-1, 1, (npar, M))
        vh, mh, sh = h.eval_batch(p); vo, mo, so = o.eval_batch(p)
        # the whole objective is a numerical contract (fma chains of the products in MFMA order, the frozen tanh of include/smmhip.h): BIT-identical
        assert np.array_equal(mh, mo) and np.array_equal(vh, vo)
        assert np.array_equal(sh, so)


def test_dense_generated_matrices_match_oracle(S, O):
    prob, opts = dense_problem(S, O, 7, 9, N=4, T=2, explicit=False)
    h, o = make_pair(S, O, prob, opts)
    p = np.random.default_rng(2).uniform(-1, 1, (7, 40))
    # (tolerance from before round 5; the generated matrices are bit-identical now: test_failing_objective_and_batches_bit_identical)
    np.testing.assert_allclose(h.eval_batch(p)[1], o.eval_batch(p)[1], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("N", [1, 5, 16, 100])
def test_dense_bgp_small(S, O, N):
    prob, opts = dense_problem(S, O, 6, 5, N=N, T=30)
    h, o = run_both(S, O, prob, opts, None)
    cm.assert_history_equal(h.history(), o.history(), atol=1e-13)
    cm.assert_state_equal(h.state(), o.state(), atol=1e-13)


def test_c5_dense_4096_chains(S, O):
    # BASELINE config 5 shape: 50 params, 256-wide dense simulation, 4096 chains
    prob, opts = dense_problem(S, O, 50, 50, N=4096, T=20)
    h, o = make_pair(S, O, prob, opts, threads=16)
    h.step(20); o.step(20)
    # (the objective itself is bit-identical — its tanh is part of the numerical contract —; the proposals' normals go through log /
    # sincos — ocml's against libm's before round 5, the contract's own now (tests/test_gpu_bitexact.py holds array_equal) —: the tolerance of then)
    cm.assert_history_equal(h.history(), o.history(), atol=1e-13)
    cm.assert_state_equal(h.state(), o.state(), atol=1e-13)


def test_c5_bench_instance_against_oracle(S, O):
    # the instance bench.py --workload c5 times since round 5 (acc_tuner 60000 .. 3000: the cold chains accept 20-40 %, sigma is
    # stationary — VERDICT r4 "Next #4"; tools/exp/c5_instance.py), 4096 chains, 60 iterations, whole history against the oracle
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    prob, opts = bench.build_problem("c5", 4096, 4096, 0, 60, 0)
    h, o = make_pair(S, O, prob, opts, threads=16)
    h.step(60); o.step(60)
    cm.assert_history_equal(h.history(), o.history(), atol=1e-13)
    cm.assert_state_equal(h.state(), o.state(), atol=1e-13)
    acc = h.history().accepted[1:].mean()
    assert 0.2 < acc < 0.9, acc          # (not the 0.99 of the old instance)


@pytest.mark.parametrize("kind,npar,N,sig,smpl,bs", [("dense", 50, 100, 0.08, 100000, None), ("dense", 50, 37, 0.12, 100000, 25),
                                                      ("dense", 50, 16, 0.3, 40, None), ("norm", 18, 70, 0.12, 100000, None),
                                                      ("norm", 32, 9, 0.15, 100000, 16), ("dense", 64, 33, 0.06, 100000, None)])
def test_many_parameters_many_tries(S, O, monkeypatch, kind, npar, N, sig, smpl, bs, hooks):
    # proposals of 16 and more components whose tries run far past the pre-generated ones (sigma so wide that a try
    # seldom lands inside the box): the tile's lane segments share the open chains' further tries; same tries, same
    # order, same winner as the serial loop (mysample, AlgoBGP.jl:400-410) -- and the same hard error when smpl_iters
    # tries do not suffice (:409)
    if kind == "dense":
        prob, opts = dense_problem(S, O, npar, npar, N=N, T=25)
    else:
        prob, opts = cm.general_normal(npar, N=N, T=25, ns=100)
    opts.sigma[:] = sig * cm.temps(N, 2.0)
    opts.smpl_iters = smpl
    if bs:
        opts.batch_size = bs
    if kind == "norm" and npar == 18:
        monkeypatch.setenv("SMMHIP_TPW", "2")   # two tiles per workgroup: the rounds of shared tries are workgroup-wide
    h, o = make_pair(S, O, prob, opts)
    eh = eo = None
    try:
        h.step(25)
    except A.SMMHipError as e:
        eh = e
    try:
        o.step(25)
    except A.SMMHipError as e:
        eo = e
    assert (eh is None) == (eo is None) == (smpl > 40)
    if eh is None:
        cm.assert_history_equal(h.history(), o.history(), atol=1e-13)
        cm.assert_state_equal(h.state(), o.state(), atol=1e-13)
    else:   # the iterations before the failing one agree
        assert eh.code == eo.code == A.SMM_ERR_NO_DRAW_IN_SUPPORT
        assert "iteration 2" in str(eh) and "chain 1," in str(eh)   # (every chain fails: the first one is reported)
        n = o.state().iter
        assert h.state().iter == n + 1
        if n:
            cm.assert_history_equal(h.history(0, n), o.history(0, n), atol=1e-13)


def test_c5_dense_sharded_8(S, O):
    # ... sharded 8 ways (512 chains per shard), against the single-shard run
    prob, opts = dense_problem(S, O, 50, 50, N=4096, T=8)
    single = S.hip_context(prob, opts); single.step(8)
    ctxs = sharded_run(S, prob, opts, 8, 8)
    hs = single.history()
    for r, c in enumerate(ctxs):
        hr = c.history()
        for f in A.HistoryBuffers.FIELDS:
            assert np.array_equal(getattr(hr, f), getattr(hs, f)[..., r * 512:(r + 1) * 512], equal_nan=True), (f, r)


@pytest.mark.gpu
def test_noseed_eval_batch(S, O):
    # objfunc_norm with noseed=true: every evaluation has its own shock matrix (getSigma's repetitions)
    prob, opts = cm.serial_normal(N=3, T=2, ns=2000)
    h, o = make_pair(S, O, prob, opts, None)
    th = np.tile(np.array([[0.4], [-0.3]]), (1, 40))             # the same parameter vector 40 times
    vh, smh, sth = h.eval_batch_noseed(th, 777)
    vo, smo, sto = o.eval_batch_noseed(th, 777)
    assert np.array_equal(sth, sto) and (sth == 1).all()
    np.testing.assert_allclose(smh, smo, rtol=1e-12, atol=1e-14)   # device sincospi vs host sin/cos in Box-Muller
    np.testing.assert_allclose(vh, vo, rtol=1e-10)
    assert np.unique(np.round(smh[0], 12)).size == 40             # 40 different shock matrices ...
    v2, sm2, _ = h.eval_batch_noseed(th[:, :5], 777 + 10)
    assert np.array_equal(sm2, smh[:, 10:15])                      # ... keyed by base_seed + i
    assert abs(smh[0].mean() - 0.4) < 0.02 and abs(smh[0].std() - 1 / np.sqrt(2000)) < 0.01   # mean of ns draws of N(theta, 1)


def _random_chol(rng, npar, scale=1.0):
    A_ = rng.standard_normal((npar, npar))
    L = np.linalg.cholesky(A_ @ A_.T / npar + 0.5 * np.eye(npar)) * scale
    return np.tril(L)


@pytest.mark.parametrize("npar,per_chain,N,sig", [(6, False, 50, 0.05), (18, True, 40, 0.03), (2, False, 33, 0.3), (18, False, 24, 0.1)])
def test_cholesky_proposals_match_oracle(S, O, npar, per_chain, N, sig):
    # general Gaussian proposals x = mu01 + sigma_c * (L z) (north star "Cholesky apply"; VERDICT r1 missing #2): shared and
    # per-chain factors; sig = 0.1 (x temperatures up to 3) at np = 18 needs tries beyond the pre-generated ones (the in-kernel generator path)
    rng = np.random.default_rng(5)
    prob, opts = cm.general_normal(npar, N=N, T=25, ns=200)
    L = np.stack([_random_chol(rng, npar) for _ in range(N)]) if per_chain else _random_chol(rng, npar)
    opts.chol_L = np.ascontiguousarray(L)
    opts.sigma[:] = sig * cm.temps(N, 3.0)
    opts.smpl_iters = 100000
    h, o = run_both(S, O, prob, opts, None)
    hh = h.history()
    cm.assert_history_equal(hh, o.history())
    cm.assert_state_equal(h.state(), o.state())
    assert hh.accepted[1:].mean() > 0.02 and (hh.exchanged != 0).any()
    # injected normals as well (bit-exact arithmetic path)
    tab = cm.random_tables(prob, opts, tries=24)
    opts.sigma[:] = 0.004 * cm.temps(N, 3.0)
    h2, o2 = run_both(S, O, prob, opts, tab, T=10)
    cm.assert_history_equal(h2.history(), o2.history(), rtol=1e-12)


def test_cholesky_identity_equals_isotropic_kernel(S):
    # L = I is the reference's MvNormal(mu01, sigma): same history as without a factor
    prob, opts = cm.general_normal(5, N=30, T=30, ns=100)
    a = S.hip_context(prob, opts); a.step(30)
    opts.chol_L = np.eye(5)
    b = S.hip_context(prob, opts); b.step(30)
    cm.assert_history_equal(a.history(), b.history(), rtol=0, atol=0)
    prob2, opts2 = cm.general_normal(4, N=3, T=4, batch_size=2)
    opts2.chol_L = np.eye(4)
    with pytest.raises(A.SMMHipError) as e:
        S.hip_context(prob2, opts2)
    assert e.value.code == A.SMM_ERR_BAD_BATCH


def test_cholesky_proposal_covariance(S):
    # oracle-independent: with every proposal accepted (acc_tuner = 0 => prob = 1 > u) and bounds far away, the steps of a
    # chain in [0,1]-space are N(0, sigma^2 L L'): their sample covariance recovers L L'
    npar, N, T = 3, 64, 400
    L = np.array([[1.0, 0, 0], [0.8, 0.6, 0], [-0.5, 0.3, 0.4]])
    prob = S.Problem(init=np.zeros(npar), lb=-1e3 * np.ones(npar), ub=1e3 * np.ones(npar), mom=np.zeros(npar), w=np.ones(npar),
                     ns=8)
    sig = 1e-4
    opts = S.BGPOpts(N=N, maxiter=T, sigma=sig * np.ones(N), acc_tuner=np.zeros(N), min_improve=1e30 * np.ones(N),
                     sigma_update_steps=10 ** 6, chol_L=L, seed=11)
    h = S.hip_context(prob, opts)
    h.step(T)
    hh = h.history()
    assert hh.accepted.all() and (hh.exchanged == 0).all()
    steps = np.diff(hh.params, axis=0) / 2e3 / sig          # [T-1][np][N] in units of sigma, [0,1]-space
    X = steps.transpose(0, 2, 1).reshape(-1, npar)
    C_ = X.T @ X / len(X)
    np.testing.assert_allclose(C_, L @ L.T, atol=0.03)


@pytest.mark.parametrize("N,mi", [(9000, 0.0), (20000, 0.3), (32768, 0.0), (12000, -0.05)])
def test_key_exchange_kernel_equals_global_memory_walk(S, O, N, mi, monkeypatch, hooks):
    # k_exch_resolve_key (4-byte LDS slots: src + 16-bit order key, exact values only for undecided pairs) against the
    # global-memory level walk it replaces for 8192 < N_global <= 32768, and against the oracle; thresholds 0, > 0, < 0
    prob, opts = cm.serial_normal(N=N, T=8, ns=64, min_improve=mi)
    a = S.hip_context(prob, opts)
    a.step(8)
    monkeypatch.setenv("SMMHIP_KEY_EXCHANGE", "0")
    b = S.hip_context(prob, opts)
    b.step(8)
    monkeypatch.delenv("SMMHIP_KEY_EXCHANGE")
    cm.assert_history_equal(a.history(), b.history(), exact_floats=True)
    assert (a.history().exchanged != 0).mean() > 0.02
    o = O.OracleContext(prob, opts, S.Tables(Z=a.Z()), threads=_all_cores(O))
    o.step(8)
    cm.assert_history_equal(a.history(), o.history(), atol=1e-13)


def test_key_exchange_kernel_ties_failures_and_small_populations(S, O, monkeypatch, hooks):
    # forced for a small population (SMMHIP_BIG_EXCHANGE): equal values (iteration 1: every chain at the start value), the
    # failed-objective value -1.0 (negative order keys) and per-chain thresholds
    monkeypatch.setenv("SMMHIP_BIG_EXCHANGE", "1")
    N, T = 300, 25
    prob, opts = cm.serial_normal(N=N, T=T, ns=100, objective_id=A.SMM_OBJ_NORM_FAILBOX, obj_params=[0.4, 1.2], sigma0=0.3,
                                  min_improve=np.linspace(-0.2, 0.4, N))
    h, o = run_both(S, O, prob, opts, None)
    hh = h.history()
    assert (hh.status == -2).any() and (hh.exchanged != 0).any()
    cm.assert_history_equal(hh, o.history())
    cm.assert_state_equal(h.state(), o.state())


@pytest.mark.parametrize("N,npar,failbox", [(4096, 2, False), (1000, 2, True), (333, 1, False), (17, 2, True)])
def test_key_walk_equals_slot_walk(S, O, monkeypatch, N, npar, failbox, hooks):
    # k_chain_iter_norm's lean exchange walk (8-byte slots: 32-bit order key of the value, src, level of the last swap; padded
    # level plan) against the same kernel on 16-byte slots (SMMHIP_KEY_WALK=0) and, for the small ones, the oracle; failbox:
    # the -1.0 of a failed objective (negative keys) and, at iteration 1, every chain at the same value (equal keys: the
    # exact values decide)
    T = 40
    kw = dict(objective_id=A.SMM_OBJ_NORM_FAILBOX, obj_params=[0.4, 1.2], sigma0=0.3) if failbox else {}
    if npar == 2:
        prob, opts = cm.serial_normal(N=N, T=T, ns=300, **kw)
    else:
        prob, opts = cm.general_normal(1, N=N, T=T, ns=300)
    a = S.hip_context(prob, opts)
    a.step(T)
    monkeypatch.setenv("SMMHIP_KEY_WALK", "0")
    b = S.hip_context(prob, opts)
    b.step(T)
    monkeypatch.delenv("SMMHIP_KEY_WALK")
    cm.assert_history_equal(a.history(), b.history(), exact_floats=True)
    cm.assert_state_equal(a.state(), b.state(), rtol=0)
    hh = a.history()
    assert (hh.exchanged != 0).mean() > 0.02
    if failbox:
        assert (hh.status == -2).any()
    if N <= 1000:
        o = O.OracleContext(prob, opts, S.Tables(Z=a.Z()), threads=_all_cores(O))
        o.step(T)
        cm.assert_history_equal(hh, o.history())
        cm.assert_state_equal(a.state(), o.state())


def test_key_walk_deep_plan_falls_back(S, O):
    # an injected pair list in which one chain takes part in 40 pairs of an iteration: 40 dependency levels do not fit the
    # slot's 5 bits, the plan says so and that launch walks on 16-byte slots; iterations with an ordinary list use the lean walk
    N, T = 64, 12
    prob, opts = cm.serial_normal(N=N, T=T, ns=200)
    rng = np.random.default_rng(5)
    pairs = np.zeros((T, N, 2), np.int32)
    for t in range(T):
        for q in range(N):
            if t % 3 == 0 and q < 40:
                i, j = 0, 1 + (q % (N - 1))          # chain 0 in 40 pairs
            else:
                i, j = sorted(rng.choice(N, 2, replace=False))
            pairs[t, q] = (i, j)                     # 0-based, i < j (include/smmhip.h)
    tab = S.Tables(pairs=pairs)
    h, o = run_both(S, O, prob, opts, tab)
    hh = h.history()
    assert (hh.exchanged != 0).any()
    cm.assert_history_equal(hh, o.history())
    cm.assert_state_equal(h.state(), o.state())


@pytest.mark.parametrize("nan", [False, True])
def test_key_walk_special_values(S, O, monkeypatch, nan, hooks):
    # last accepted values that the order keys must not get wrong, planted through set_state: values that share the high word
    # of the double (equal keys: the exact values are read), -0.0 against +0.0 (equal), the -1.0 of a failed objective, Inf,
    # and (nan) NaN, which no key covers: the accept step raises the sticky flag and the launches walk on 16-byte slots
    N, T0, T1 = 96, 4, 14
    prob, opts = cm.serial_normal(N=N, T=T0 + T1, ns=200, acc_tuners=np.full(96, 400.0))
    a = S.hip_context(prob, opts)
    a.step(T0)
    st, hist = a.state(), a.history()
    v = st.la_value
    v[0:32] = 1.0 + np.arange(32) * 2.0 ** -45          # one high word, 32 low words
    v[32:36] = [-0.0, 0.0, -0.0, 0.0]
    v[36:40] = [-1.0, np.inf, -1.0, np.inf]
    v[40:48] = 3.0e-300                                   # equal values
    if nan:
        v[48:52] = np.nan
    runs = []
    for env in (None, "0"):
        if env is not None:
            monkeypatch.setenv("SMMHIP_KEY_WALK", env)
        b = S.hip_context(prob, opts)
        b.set_state(st, hist)
        b.step(T1)
        runs.append(b)
        if env is not None:
            monkeypatch.delenv("SMMHIP_KEY_WALK")
    cm.assert_history_equal(runs[0].history(), runs[1].history(), exact_floats=True)
    cm.assert_state_equal(runs[0].state(), runs[1].state(), rtol=0)
    o = O.OracleContext(prob, opts, S.Tables(Z=a.Z()))
    o.set_state(st, hist)
    o.step(T1)
    hh = runs[0].history()
    assert (hh.exchanged[T0:] != 0).any()
    cm.assert_history_equal(hh, o.history())
    cm.assert_state_equal(runs[0].state(), o.state())


@pytest.mark.parametrize("N,npar,mi,failbox", [(4096, 2, 0.05, False), (1000, 2, 0.5, True), (333, 1, 1e-3, False), (17, 2, 0.05, True),
                                               (64, 2, np.inf, False), (64, 2, np.nan, False), (5, 3, 0.01, False), (2, 2, 0.0005, False)])
def test_wide_walk_equals_slot_walk(S, O, monkeypatch, N, npar, mi, failbox, hooks):
    # one min_improve > 0 for all chains (the reference's default is 0.5, AlgoBGP.jl:522; Examples.jl:90 uses 0.05): the lean
    # walk on 16-byte slots {value, src | stamp} (k_chain_iter_norm_wide) against the same run on the older walk
    # (SMMHIP_KEY_WALK=0) and, for the small ones, the oracle.  Inf / NaN thresholds: nothing ever swaps.
    T = 40
    kw = dict(objective_id=A.SMM_OBJ_NORM_FAILBOX, obj_params=[0.4, 1.2], sigma0=0.3) if failbox else {}
    if npar == 2:
        prob, opts = cm.serial_normal(N=N, T=T, ns=300, min_improve=mi, **kw)
    else:
        prob, opts = cm.general_normal(npar, N=N, T=T, ns=300)
        opts.min_improve[:] = mi
    a = S.hip_context(prob, opts)
    a.step(T)
    monkeypatch.setenv("SMMHIP_KEY_WALK", "0")
    b = S.hip_context(prob, opts)
    b.step(T)
    monkeypatch.delenv("SMMHIP_KEY_WALK")
    cm.assert_history_equal(a.history(), b.history(), exact_floats=True)
    cm.assert_state_equal(a.state(), b.state(), rtol=0)
    hh = a.history()
    if np.isfinite(mi):
        assert (hh.exchanged != 0).any()
    else:
        assert not (hh.exchanged != 0).any()
    if failbox:
        assert (hh.status == -2).any()
    if N <= 1000:
        o = O.OracleContext(prob, opts, S.Tables(Z=a.Z()), threads=_all_cores(O))
        o.step(T)
        cm.assert_history_equal(hh, o.history())
        cm.assert_state_equal(a.state(), o.state())


@pytest.mark.parametrize("Ng,G", [(6000, 1), (7400, 2), (8192, 2)])
def test_wide_walk_standalone_and_sharded(S, O, monkeypatch, Ng, G, hooks):
    # the same form as the kernel of its own (k_exch_resolve_lean): single shards too large for the inline walk, and shards of
    # a sharded run (the walk reads the gathered records); 8192 chains do not fit the LDS on 16-byte slots and keep the older
    # kernel — all against the run with SMMHIP_KEY_WALK=0
    T = 12
    prob, opts = cm.serial_normal(N=Ng, T=T, ns=64, min_improve=0.02)
    if G == 1:
        a = S.hip_context(prob, opts); a.step(T)
        monkeypatch.setenv("SMMHIP_KEY_WALK", "0")
        b = S.hip_context(prob, opts); b.step(T)
        monkeypatch.delenv("SMMHIP_KEY_WALK")
        cm.assert_history_equal(a.history(), b.history(), exact_floats=True)
        cm.assert_state_equal(a.state(), b.state(), rtol=0)
        assert (a.history().exchanged != 0).any()
    else:
        single = S.hip_context(prob, opts); single.step(T)
        ctxs = sharded_run(S, prob, opts, G, T)
        hs = single.history()
        n = Ng // G
        for r, c in enumerate(ctxs):
            hr = c.history()
            for f in cm.INT_FIELDS:
                np.testing.assert_array_equal(getattr(hr, f), getattr(hs, f)[..., r * n:(r + 1) * n], err_msg=f)
            np.testing.assert_array_equal(hr.value, hs.value[:, r * n:(r + 1) * n])
        assert (hs.exchanged != 0).any()


@pytest.mark.parametrize("dist,mi", [(1, 0.02), (2, 0.05)])
@pytest.mark.parametrize("shape", ["norm64", "norm1000", "general4096", "norm5000", "norm20000", "sharded2", "values2", "perchain"])
def test_dist_fun_menu(S, O, dist, mi, shape):
    # opts["dist_fun"] other than the default `-` (AlgoBGP.jl:537,688; VERDICT r1 missing #6): |a - b| and (a - b) / |a| through every
    # exchange path that serves them — the inline walks of both chain kernels, the stand-alone level kernels in LDS (<= 4096,
    # <= 8192) and in global memory, both sharded forms — against the oracle (the key / lean / rows walks are for `-`)
    T = 14
    if shape == "general4096":
        prob, opts = cm.general_normal(6, N=4096, T=T, ns=64)
        opts.min_improve[:] = mi
    else:
        N = {"norm64": 64, "norm1000": 1000, "norm5000": 5000, "norm20000": 20000, "sharded2": 600, "values2": 600, "perchain": 300}[shape]
        prob, opts = cm.serial_normal(N=N, T=T, ns=64, min_improve=mi)
        if shape == "perchain":
            opts.min_improve[:] = np.linspace(0.0, 2 * mi, N)
    opts.dist_fun = dist
    o = O.OracleContext(prob, opts, threads=_all_cores(O))
    if shape in ("sharded2", "values2"):
        ctxs = (sharded_run if shape == "sharded2" else sharded_run_values)(S, prob, opts, 2, T)
        o = O.OracleContext(prob, opts, S.Tables(Z=ctxs[0].Z()), threads=_all_cores(O))
        o.step(T)
        ho = o.history()
        n = opts.N_global // 2
        for r, c in enumerate(ctxs):
            hr = c.history()
            for f in cm.INT_FIELDS:
                np.testing.assert_array_equal(getattr(hr, f), getattr(ho, f)[..., r * n:(r + 1) * n], err_msg=f)
            np.testing.assert_allclose(hr.value, ho.value[:, r * n:(r + 1) * n], rtol=1e-9)
        assert (ho.exchanged != 0).any()
        return
    h, o = run_both(S, O, prob, opts, None)
    hh = h.history()
    cm.assert_history_equal(hh, o.history())
    cm.assert_state_equal(h.state(), o.state())
    assert (hh.exchanged != 0).any()


def test_dist_fun_through_the_host_api(S):
    from smm_jl_amd import host as H
    import operator
    assert H._dist_fun_id(None) == H._dist_fun_id("-") == H._dist_fun_id(operator.sub) == A.SMM_DIST_MINUS
    assert H._dist_fun_id("absdiff") == A.SMM_DIST_ABSDIFF and H._dist_fun_id("reldiff") == A.SMM_DIST_RELDIFF
    with pytest.raises(NotImplementedError):
        H._dist_fun_id(lambda a, b: a * b)
    prob, opts = cm.serial_normal(N=8, T=3, ns=50)
    opts.dist_fun = 7
    with pytest.raises(A.SMMHipError) as e:
        S.hip_context(prob, opts)
    assert e.value.code == A.SMM_ERR_INVALID_ARG


@pytest.mark.parametrize("mi", [0.0, 0.3])
def test_first_launches_of_a_fresh_process(mi):
    # Found by tools/fuzz_parity.py: the values the exchange walk reads (KParams::vals, the lean walk's slots) used to be ONE array,
    # read by every workgroup in the prologue and rewritten by each workgroup's accept step — a workgroup that starts late
    # (cold instruction fetch on the first launches of a process; short kernels: few chains, few draws) found its neighbours' NEW
    # values in its walk's input and the redundant walks disagreed.  Now two arrays by iteration parity.  The failure needed a
    # fresh process, hence the child processes (5 of 6 failed before the fix).
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import smm_jl_amd as S, common as cm
        from oracle import oracle as O
        # (the case of the sweep: wide proposals, so that a chain's value changes a lot from one iteration to the next)
        prob = S.Problem(init=[1.4926310430330663], lb=[-3.4640012631720185], ub=[3.4640012631720185], mom=[-1.7242533461784755],
                         w=[np.nan], ns=4096)
        opts = S.BGPOpts(N=100, maxiter=14, sigma=0.3 * cm.temps(100, 8.0), acc_tuner=np.geomspace(10.0, 0.5, 100),
                         min_improve=np.full(100, %r), seed=673502443, batch_size=1, sigma_update_steps=3, N_global=100)
        h = S.hip_context(prob, opts)
        o = O.OracleContext(prob, opts, S.Tables(Z=h.Z()))
        h.step(14); o.step(14)
        cm.assert_history_equal(h.history(), o.history(), rtol=1e-9, atol=1e-12)
        print("fresh process ok")
    """) % (ROOT, os.path.join(ROOT, "tests"), mi)
    for _ in range(3):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "fresh process ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def _slice_state(st, lo, hi):
    """the chains lo..hi of a state (every per-chain array has the chain as its last axis)"""
    import copy
    out = copy.copy(st)
    for f in A.StateBuffers.FIELDS:
        setattr(out, f, getattr(st, f)[..., lo:hi])
    return out


def sharded_run_values(S, prob, opts_full, G, T):
    """the values form of the exchange phase (smm_bgp_export_values_dev / a2a_pack_dev / a2a_apply_dev): G contexts on one GPU
    emulate G ranks; the all-gather is a concatenation and the all-to-all a transposition of blocks here (the collectives
    themselves run under gloo in tests/test_dist_gloo.py)"""
    import torch
    from smm_jl_amd import BGPOpts
    N = opts_full.N_global // G
    ctxs = []
    for r in range(G):
        o = BGPOpts(N=N, maxiter=opts_full.maxiter, sigma=opts_full.sigma, acc_tuner=opts_full.acc_tuner,
                    min_improve=opts_full.min_improve, sigma_update_steps=opts_full.sigma_update_steps,
                    sigma_adjust_by=opts_full.sigma_adjust_by, smpl_iters=opts_full.smpl_iters,
                    batch_size=opts_full.batch_size, seed=opts_full.seed, chain_offset=r * N,
                    N_global=opts_full.N_global, dist_fun=opts_full.dist_fun, chol_L=opts_full.chol_L)
        ctxs.append(S.hip_context(prob, o))
    R, cap = ctxs[0].record_doubles(), ctxs[0].a2a_capacity()
    assert cap > 0
    vall = torch.empty((G, N), dtype=torch.float64, device="cuda")
    send = torch.zeros((G, G, cap, R), dtype=torch.float64, device="cuda")   # [rank][destination]
    torch.cuda.synchronize()   # (the fill runs on torch's stream, the contexts on their own)
    for _ in range(T):
        for r, c in enumerate(ctxs):
            c.local_step()
            c.export_values_dev(vall[r].data_ptr())
        for c in ctxs:
            c.sync()
        for r, c in enumerate(ctxs):
            c.a2a_pack_dev(vall.data_ptr(), send[r].data_ptr())
        for c in ctxs:
            c.sync()
        recv = send.transpose(0, 1).contiguous()                              # [rank][source]
        torch.cuda.synchronize()   # (torch's stream, not the contexts': the copy must have landed before a2a_apply reads it)
        for r, c in enumerate(ctxs):
            c.a2a_apply_dev(recv[r].data_ptr())
        for c in ctxs:
            c.sync()
    return ctxs


@pytest.mark.parametrize("G,N,T,npar", [(2, 32, 25, 2), (4, 400, 40, 2), (8, 4096, 12, 2), (4, 96, 20, 6), (3, 9000, 6, 2)])
def test_values_form_of_the_sharded_exchange_equals_single(S, O, G, N, T, npar):
    # 8 x 512: the lean resolve kernel; 3 x 3000: k_exch_resolve_rows; np = 6: the general chain kernel, longer records
    N -= N % G
    if npar == 2:
        prob, opts = cm.serial_normal(N=N, T=T, ns=64 if N > 100 else 300)
    else:
        prob, opts = cm.general_normal(npar, N=N, T=T, ns=100)
    single = S.hip_context(prob, opts)
    single.step(T)
    ctxs = sharded_run_values(S, prob, opts, G, T)
    hs = single.history()
    n = N // G
    assert (hs.exchanged != 0).any()
    for r, c in enumerate(ctxs):
        hr = c.history()
        for f in A.HistoryBuffers.FIELDS:
            assert np.array_equal(getattr(hr, f), getattr(hs, f)[..., r * n:(r + 1) * n], equal_nan=True), (f, r)
        cm.assert_state_equal(c.state(), _slice_state(single.state(), r * n, (r + 1) * n), rtol=0)


def test_values_form_block_overflow_is_a_hard_error(S, monkeypatch, hooks):
    # a (source, destination) block that needs more records than its capacity: sticky device error, reported at the next sync
    monkeypatch.setenv("SMMHIP_A2A_CAP", "1")
    prob, opts = cm.serial_normal(N=64, T=10, ns=64)
    with pytest.raises(A.SMMHipError) as e:
        sharded_run_values(S, prob, opts, 2, 10)
    assert e.value.code == A.SMM_ERR_EXCHANGE_CAPACITY


def test_lean_walk_across_plan_windows(S, O, monkeypatch, hooks):
    # 1000 chains over 600 iterations: the look-ahead plan is rebuilt twice (windows of 256 iterations), every switch settles the
    # open exchange through the stand-alone kernel; stepping in uneven pieces; lean walk against the 16-byte walk and the oracle
    N, T = 1000, 600
    prob, opts = cm.serial_normal(N=N, T=T, ns=64)
    a = S.hip_context(prob, opts)
    for n in (1, 254, 3, 255, 87):
        a.step(n)
    monkeypatch.setenv("SMMHIP_KEY_WALK", "0")
    b = S.hip_context(prob, opts)
    b.step(T)
    monkeypatch.delenv("SMMHIP_KEY_WALK")
    cm.assert_history_equal(a.history(), b.history(), exact_floats=True)
    cm.assert_state_equal(a.state(), b.state(), rtol=0)
    o = O.OracleContext(prob, opts, S.Tables(Z=a.Z()), threads=_all_cores(O))
    o.step(T)
    cm.assert_history_equal(a.history(), o.history(), atol=1e-13)
    cm.assert_state_equal(a.state(), o.state(), atol=1e-13)


@pytest.mark.parametrize("scout_after,gl", [(0, 16), (0, 8), (1, 16), (1000000, 16)])
def test_late_tries_scouted_by_lane_groups_equal_the_rounds(S, O, monkeypatch, hooks, scout_after, gl):
    # mysample's tries past the pre-generated ones (AlgoBGP.jl:400-410) in k_chain_iter: rounds of one try per lane segment, then — for
    # the chains still open — tries handed out by a counter per chain to groups of 8 / 16 lanes that give a try up at its first
    # group of pairs outside the box.  Whatever the split (all scouted, one round first, never scouted), whatever the group size:
    # the first successful try in order wins.  50 parameters, sigmas so wide that a proposal takes tens to hundreds of tries
    monkeypatch.setenv("SMMHIP_SCOUT_AFTER", str(scout_after))
    monkeypatch.setenv("SMMHIP_SCOUT_GL", str(gl))
    prob, opts = dense_problem(S, O, 50, 50, N=200, T=12)
    opts.sigma[:] = 0.05 * cm.temps(200, 2.0)
    opts.smpl_iters = 100000
    h, o = make_pair(S, O, prob, opts)
    h.step(12); o.step(12)
    cm.assert_history_equal(h.history(), o.history(), atol=1e-13)
    cm.assert_state_equal(h.state(), o.state(), atol=1e-13)


@pytest.mark.parametrize("N,no_cone", [(4096, False), (4096, True), (1000, False), (48, False), (2, False)])
def test_dense_tiles_walk_the_exchange_in_their_prologue(S, O, monkeypatch, hooks, N, no_cone):
    # BASELINE config 5 runs ONE launch per iteration: the dense objective's tiles walk the key exchange themselves (their cone where
    # the population is whole tiles of 16 chains; the whole list otherwise, or with the cones switched off), slots and lists UNDER the
    # tile's blocks in LDS.  Against the oracle and, to the bit, against the stand-alone resolution (SMMHIP_DENSE_KEYS=0)
    T = 14
    prob, opts = dense_problem(S, O, 50, 50, N=N, T=T)
    if no_cone:
        monkeypatch.setenv("SMMHIP_NO_CONE", "1")
    h, o = make_pair(S, O, prob, opts, threads=16)
    for n in (1, 5, 8):
        h.step(n); o.step(n)
    cm.assert_history_equal(h.history(), o.history(), atol=1e-13)
    cm.assert_state_equal(h.state(), o.state(), atol=1e-13)
    monkeypatch.setenv("SMMHIP_DENSE_KEYS", "0")
    c = S.hip_context(prob, opts)
    c.step(T)
    cm.assert_history_equal(h.history(), c.history(), exact_floats=True)
    assert N == 2 or (h.history().exchanged != 0).any()


@pytest.mark.parametrize("N,T", [(8208, 12), (20000, 8), (32768, 6)])
def test_large_shard_tiles_walk_their_own_cones(S, O, monkeypatch, hooks, N, T):
    # large single shards of objfunc_norm (8192 < N <= 32768: BASELINE config 3 on one GPU): the narrow chain kernel's tiles walk the
    # exchange over their own, LOCALLY numbered cones (smm_cone_big.hpp) instead of waiting for the one-workgroup resolution between two
    # launches.  Against the oracle and, to the bit, against that resolution (SMMHIP_CONE_BIG=0); uneven steps, a read-back in between
    prob, opts = cm.serial_normal(N=N, T=T, ns=48)
    h, o = make_pair(S, O, prob, opts, threads=16)
    for n in (1, 3, T - 4):
        h.step(n); o.step(n)
        if n == 3:
            cm.assert_state_equal(h.state(), o.state(), atol=1e-13)   # (simulated moments cross zero)
    cm.assert_history_equal(h.history(), o.history(), atol=1e-13)
    cm.assert_state_equal(h.state(), o.state(), atol=1e-13)
    monkeypatch.setenv("SMMHIP_CONE_BIG", "0")
    c = S.hip_context(prob, opts)
    c.step(T)
    cm.assert_history_equal(h.history(), c.history(), exact_floats=True)
    assert (h.history().exchanged != 0).mean() > 0.05
    # a run continued from an uploaded state (smm_set_state in the middle of the plan window) ends where the uninterrupted one does
    monkeypatch.delenv("SMMHIP_CONE_BIG")
    a = S.hip_context(prob, opts)
    a.step(T - 3)
    b = S.hip_context(prob, opts)
    b.set_state(a.state(), a.history(0, T - 3))
    b.step(3)
    cm.assert_history_equal(h.history(), b.history(), exact_floats=True)
    cm.assert_state_equal(h.state(), b.state(), rtol=0)


@pytest.mark.parametrize("cap", [1, 7])
def test_large_shard_plan_windows_are_planned_ahead(S, O, monkeypatch, hooks, cap):
    # the windows of exchange plans (with the tiles' cones) of large shards are planned AHEAD on a second stream, into the other of two
    # sets of tables, while the chain kernels of the current window run.  Short windows (test hook) so that a short run crosses many of
    # them: against the oracle and, to the bit, against the windows planned in place on the main stream (SMMHIP_PLAN_AHEAD=0); uneven
    # asynchronous steps with read-backs in between; a continuation from an uploaded state (its first window starts in the middle of
    # one of the uninterrupted run's: not the window planned ahead); two contexts side by side (each has its plan stream)
    N, T = 8208, 40
    prob, opts = cm.serial_normal(N=N, T=T, ns=32)
    monkeypatch.setenv("SMMHIP_PLAN_CAP", str(cap))
    h, o = make_pair(S, O, prob, opts, threads=16)
    g = S.hip_context(prob, opts)
    done = 0
    for n in (1, 5, 2, 9, 14, T - 31):
        h.step_async(n); g.step_async(n); done += n
        if n in (2, 14):
            cm.assert_history_equal(h.history(done - 2, done), g.history(done - 2, done), exact_floats=True)
    o.step(T)
    cm.assert_history_equal(h.history(), o.history(), atol=1e-13)
    cm.assert_state_equal(h.state(), o.state(), atol=1e-13)
    cm.assert_history_equal(h.history(), g.history(), exact_floats=True)
    assert (h.history().exchanged != 0).mean() > 0.05
    import ctypes as C
    hdr = np.zeros(9, np.uint32); pairs = np.zeros(2048, np.uint32); gl = np.zeros(512, np.uint16); info = np.zeros(4, np.int32)
    assert S._abi.load_hooks().smm_debug_cone(h._ctx, 0, 0, hdr.ctypes.data_as(C.c_void_p), pairs.ctypes.data_as(C.c_void_p), gl.ctypes.data_as(C.c_void_p),
                                              info.ctypes.data_as(C.c_void_p)) == 0
    assert info[1] <= cap and info[0] + info[1] - 1 == T        # (the last of the short windows)
    a = S.hip_context(prob, opts)
    a.step(T - 11)
    b = S.hip_context(prob, opts)
    b.set_state(a.state(), a.history(0, T - 11))   # from iteration T - 10 on
    b.step(4); b.step_async(7); b.sync()
    cm.assert_history_equal(h.history(), b.history(), exact_floats=True)
    cm.assert_state_equal(h.state(), b.state(), rtol=0)
    monkeypatch.setenv("SMMHIP_PLAN_AHEAD", "0")
    c = S.hip_context(prob, opts)
    c.step(T)
    cm.assert_history_equal(h.history(), c.history(), exact_floats=True)
    cm.assert_state_equal(h.state(), c.state(), rtol=0)


def test_large_shard_cone_that_does_not_fit_takes_the_resolution(S, O):
    # injected pair lists: odd iterations carry one chain of tile 0 at the end of a dependency chain 700 pairs long (a cone of more than
    # 384 pairs and 63 levels): the plan flags the iteration, the host sends it to k_exch_resolve_rows; even iterations walk cones
    N, T = 8208, 6
    prob, opts = cm.serial_normal(N=N, T=T, ns=32, sigma0=0.02)
    tab = cm.random_tables(prob, opts, tries=24)
    chain = [(100 + k, 101 + k) for k in range(700)] + [(3, 800)]        # ... -> (799, 800) -> (3, 800): chain 3's cone is all of it
    rest = [(1000 + 2 * (k % 3600), 1001 + 2 * (k % 3600)) for k in range(N - len(chain))]   # (the other chains among themselves)
    pt = np.array(chain + rest, np.int32)
    tab.pairs[1::2] = pt[None]
    h, o = run_both(S, O, prob, opts, tab)
    cm.assert_history_equal(h.history(), o.history(), atol=1e-13)
    assert (h.history().exchanged != 0).any()


@pytest.mark.parametrize("N", [8208, 32768])
def test_large_shard_cones_are_exactly_the_pairs_that_matter(S, hooks, N):
    # the plan side of smm_cone_big.hpp on its own (test build: smm_debug_cone): a tile's cone must hold exactly the pairs its 16 chains'
    # outcome depends on — found here by walking the injected pair list backwards on the CPU —, every chain's pairs in list order
    # across the sub-levels, the local numbers consistent with the gather list
    import ctypes as C
    prob, opts = cm.serial_normal(N=N, T=4, ns=8)
    tab = cm.random_tables(prob, opts, tries=8)
    a = S.hip_context(prob, opts, tab)
    a.step(2)
    lib = S._abi.load_hooks()
    hdr = np.zeros(9, np.uint32); pairs = np.zeros(2048, np.uint32); gl = np.zeros(512, np.uint16); info = np.zeros(4, np.int32)
    for tile in (0, 7, N // 16 - 1):
        for w in (1, 2):
            assert lib.smm_debug_cone(a._ctx, w, tile, hdr.ctypes.data_as(C.c_void_p), pairs.ctypes.data_as(C.c_void_p), gl.ctypes.data_as(C.c_void_p),
                                      info.ctypes.data_as(C.c_void_p)) == 0
            assert info[2] == 1 and info[3] == 16
            pl = tab.pairs[info[0] + w - 1]
            nsub, ngat = int(hdr[0] & 0xffff), int(hdr[0] >> 16)
            cnts = [int((hdr[1 + (s >> 2)] >> (8 * (s & 3))) & 0xff) for s in range(nsub)]
            loc = list(range(tile * 16, tile * 16 + 16)) + [int(x) for x in gl[:ngat]]
            assert len(set(loc)) == len(loc)
            cone = [(loc[(int(pairs[s * 64 + l]) & 0xffff) >> 3], loc[(int(pairs[s * 64 + l]) >> 16) >> 3], s) for s in range(nsub) for l in range(cnts[s])]
            need, exp = set(range(tile * 16, tile * 16 + 16)), set()
            for q in range(len(pl) - 1, -1, -1):
                i, j = int(pl[q, 0]), int(pl[q, 1])
                if i in need or j in need:
                    exp.add((i, j)); need.add(i); need.add(j)
            assert set((i, j) for (i, j, s) in cone) == exp and len(cone) == len(exp)
            posq = {(int(pl[q, 0]), int(pl[q, 1])): q for q in range(len(pl))}
            last = {}
            for (i, j, s) in cone:
                q = posq[(i, j)]
                for ch in (i, j):
                    assert ch not in last or (last[ch][0] < q and last[ch][1] < s)
                    last[ch] = (q, s)
