"""The p2p form of the sharded iteration (include/smmhip.h, smm.jl_amd/csrc/smm_p2p.hpp): every shard's accept step stores into
every shard's window, no collective.  Two layers:
  * several contexts of THIS process (windows attached by device pointer), stepped in lockstep: the data path — inline walk over
    the whole population out of the window, epilogue pushes, generic push kernel, resolve from the window — bit-exact against the
    single shard and the oracle;
  * several PROCESSES on the one GPU (windows attached through HIP IPC handles), free running: the transport itself — arrival
    counters, system-scope release/acquire, kernels of different processes waiting for each other."""
import os
import subprocess
import sys

import numpy as np
import pytest

import common as cm
from smm_jl_amd import _abi as A

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def shard_opts(opts_full, G, r):
    from smm_jl_amd import BGPOpts
    N = opts_full.N_global // G
    return BGPOpts(N=N, maxiter=opts_full.maxiter, sigma=opts_full.sigma, acc_tuner=opts_full.acc_tuner,
                   min_improve=opts_full.min_improve, sigma_update_steps=opts_full.sigma_update_steps,
                   sigma_adjust_by=opts_full.sigma_adjust_by, smpl_iters=opts_full.smpl_iters,
                   batch_size=opts_full.batch_size, seed=opts_full.seed, chain_offset=r * N,
                   N_global=opts_full.N_global, dist_fun=opts_full.dist_fun, chol_L=opts_full.chol_L)


def p2p_contexts(S, prob, opts_full, G, tables=None):
    ctxs = [S.hip_context(prob, shard_opts(opts_full, G, r), tables[r] if tables else None) for r in range(G)]
    wins = [c.p2p_init()[1] for c in ctxs]
    for r, c in enumerate(ctxs):
        for q in range(G):
            if q != r:
                c.p2p_attach(q, window=wins[q])
    return ctxs


def p2p_run_lockstep(ctxs, T, chunk=1, finish_every=None):
    """every shard enqueues `chunk` iterations, then all are waited for: with chunk == 1 every wait inside a kernel is already
    satisfied when the kernel starts (contexts of one process may share a hardware queue)"""
    done = 0
    while done < T:
        n = min(chunk, T - done)
        for c in ctxs:
            c.p2p_step(n)
        for c in ctxs:
            c.sync()
        done += n
        if finish_every and done % finish_every == 0 and done < T:
            for c in ctxs:
                c.p2p_finish()
            for c in ctxs:
                c.sync()
    for c in ctxs:
        c.p2p_finish()
    for c in ctxs:
        c.sync()


def assert_shards_equal_single(ctxs, single):
    hs, ss = single.history(), single.state()
    G = len(ctxs)
    n = hs.value.shape[1] // G
    for r, c in enumerate(ctxs):
        hr, st = c.history(), c.state()
        for f in A.HistoryBuffers.FIELDS:
            a, b = getattr(hr, f), getattr(hs, f)[..., r * n:(r + 1) * n]
            if not np.array_equal(a, b, equal_nan=True):
                bad = np.argwhere(~((a == b) | ((a != a) & (b != b))))
                raise AssertionError("history field %s of rank %d: %d entries differ, first at %s" % (f, r, len(bad), bad[0].tolist()))
        for f in A.StateBuffers.FIELDS:
            assert np.array_equal(getattr(st, f), getattr(ss, f)[..., r * n:(r + 1) * n], equal_nan=True), (f, r)
    assert (hs.exchanged != 0).any()


@pytest.mark.parametrize("G,N,T,fe", [(1, 48, 20, None), (2, 64, 30, None), (4, 64, 30, 7), (2, 5000, 12, None), (4, 8192, 8, None), (8, 640, 40, 11), (2, 64, 300, None)])
def test_p2p_inline_equals_single(S, O, G, N, T, fe):
    # objfunc_norm 2p/2m, min_improve == 0, N_global <= 8192: one launch per iteration and shard (k_chain_iter_norm_p2p)
    # (300 iterations: across a look-ahead window of 256)
    prob, opts = cm.serial_normal(N=N, T=T, ns=64 if N > 100 else 300)
    single = S.hip_context(prob, opts)
    single.step(T)
    ctxs = p2p_contexts(S, prob, opts, G)
    p2p_run_lockstep(ctxs, T, finish_every=fe)
    assert_shards_equal_single(ctxs, single)
    if N <= 100:
        o = O.OracleContext(prob, opts, S.Tables(Z=single.Z()))
        o.step(T)
        cm.assert_history_equal(single.history(), o.history())


def test_p2p_inline_nan_in_one_shard_is_reported_by_every_rank_in_the_same_iteration(S):
    # a NaN value in ONE shard's uploaded state.  The one-launch form has no second walk: every rank must report
    # SMM_ERR_HIP, for the same iteration, and promptly — the NaN fact travels in the slot word itself (P2P_KEY_NAN), so no rank can
    # validate a peer's slots and miss it (the window's separate NaN word may become visible later than the slots)
    import time
    G, N, T0 = 2, 64, 4
    prob, opts = cm.serial_normal(N=N, T=40, ns=64)
    ctxs = p2p_contexts(S, prob, opts, G)
    p2p_run_lockstep(ctxs, T0)
    st = [c.state() for c in ctxs]
    hs = [c.history() for c in ctxs]
    st[1].la_value[5] = np.nan          # rank 1 only
    for c, s_, h_ in zip(ctxs, st, hs):
        c.set_state(s_, h_)
    t0 = time.perf_counter()
    errs = [None] * G
    for _ in range(3):   # in lockstep, one iteration at a time (contexts of one process may share a hardware queue: p2p_run_lockstep)
        for c in ctxs:
            c.p2p_step(1)
        for r, c in enumerate(ctxs):
            try:
                c.sync()
            except A.SMMHipError as e:
                errs[r] = str(e)
        if any(errs):
            break
    assert time.perf_counter() - t0 < 3.0, "a rank sat in its time-out"
    assert all(errs), errs
    assert all("could not be resolved" in e for e in errs), errs
    its = [e.split("iteration ")[1].split()[0] for e in errs]
    assert its[0] == its[1], errs


@pytest.mark.parametrize("G,N,T,fe,failbox", [(4, 16384, 8, None, False), (8, 32768, 6, 4, False), (2, 20000, 6, None, False), (3, 9000, 8, 3, False),
                                              (4, 16384, 6, None, True), (8, 32768, 5, None, True), (2, 10000, 300, None, False)])
def test_p2p_rows_equals_single(S, G, N, T, fe, failbox):
    # objfunc_norm, min_improve == 0, 8192 < N_global <= 32768 (BASELINE configs[2]: 8 shards of 4096): two launches per iteration and
    # shard — k_exch_resolve_rows<., true> on the tagged slots of the window, k_chain_iter_norm_p2p without the walk.  failbox: the
    # first iteration leaves the value -1 with every chain that starts inside the box (a key that orders nothing: the iteration's
    # exchange is resolved by the fallback on the exact values, unpacked inside the kernel)
    # (8 x 4096 of 32768: the form that keeps the partners of the rank's own chains only, k_exch_resolve_rows<false, true, true>)
    kw = dict(objective_id=A.SMM_OBJ_NORM_FAILBOX, obj_params=[0.1, 0.3]) if failbox else {}
    prob, opts = cm.serial_normal(N=N, T=T, ns=64, **kw)
    single = S.hip_context(prob, opts)
    single.step(T)
    if failbox:
        assert (single.history().value[0] == -1.0).all()
    ctxs = p2p_contexts(S, prob, opts, G)
    p2p_run_lockstep(ctxs, T, finish_every=fe)
    assert_shards_equal_single(ctxs, single)


def test_p2p_rows_c3_real_workload_against_the_oracle(S, O):
    # VERDICT r3 "Next #6": the DEFAULT multi-GPU form of BASELINE configs[2] — 8 shards x 4096 chains, the rows form through the
    # windows — at the real workload (ns = 10000), 70 iterations (past the first look-ahead pieces of the rows plan), whole history
    # of every shard directly against the ORACLE (not against the single-shard HIP run)
    G, N, T = 8, 32768, 70
    prob, opts = cm.serial_normal(N=N, T=T, ns=10000)
    ctxs = p2p_contexts(S, prob, opts, G)
    p2p_run_lockstep(ctxs, T)
    o = O.OracleContext(prob, opts, S.Tables(Z=ctxs[0].Z()), threads=O.max_threads())
    o.step(T)
    ho, so = o.history(), o.state()
    n = N // G
    for r, c in enumerate(ctxs):
        hr, st = c.history(), c.state()
        for f in cm.INT_FIELDS:
            a, b = getattr(hr, f), getattr(ho, f)[..., r * n:(r + 1) * n]
            bad = np.argwhere(a != b)
            assert bad.size == 0, "%s of rank %d differs at %s (first of %d)" % (f, r, bad[0], len(bad))
        for f in cm.F64_FIELDS:
            np.testing.assert_allclose(getattr(hr, f), getattr(ho, f)[..., r * n:(r + 1) * n], rtol=1e-9, equal_nan=True, err_msg="%s rank %d" % (f, r))
        for f in ("la_status", "n_noex", "n_acc_noex", "best_id"):
            assert np.array_equal(getattr(st, f), getattr(so, f)[..., r * n:(r + 1) * n]), (f, r)
        for f in ("sigma", "accept_rate", "la_value", "la_params", "best_val"):
            np.testing.assert_allclose(getattr(st, f), getattr(so, f)[..., r * n:(r + 1) * n], rtol=1e-9, equal_nan=True, err_msg=f)
    assert 0.1 < (ho.exchanged != 0).mean() < 0.5


def test_p2p_rows_with_a_deep_plan_and_injected_tables(S):
    # rows form with injected randomness and a pair list that does not fit the rows plan in the odd iterations (forty pairs through
    # chain 0, one after the other): those iterations take the fallback inside k_exch_resolve_rows<., true>, the others the rows walk
    G, N, T = 3, 9000, 6
    prob, opts = cm.serial_normal(N=N, T=T, ns=32, min_improve=0.0)
    tab = cm.random_tables(prob, opts, tries=24)
    tab.pairs[1::2, :40, 0] = 0
    tab.pairs[1::2, :40, 1] = 1 + np.arange(40)[None, :] * 3
    single = S.hip_context(prob, opts, tab)
    single.step(T)
    n = N // G
    tabs = [S.Tables(probs_acc=tab.probs_acc[:, r * n:(r + 1) * n], prop_normals=tab.prop_normals[..., r * n:(r + 1) * n], pairs=tab.pairs, Z=tab.Z)
            for r in range(G)]
    ctxs = p2p_contexts(S, prob, opts, G, tabs)
    p2p_run_lockstep(ctxs, T)
    assert_shards_equal_single(ctxs, single)


@pytest.mark.parametrize("case", ["norm_16384", "norm_mi", "norm_mi_8192", "general_np6", "banana", "dense"])
def test_p2p_generic_equals_single(S, case):
    # everything the inline form does not cover: chain kernel into the own window + push kernel + resolve from the window
    G, T = 2, 10
    if case == "norm_16384":
        G, T = 4, 5
        prob, opts = cm.serial_normal(N=16384, T=T, ns=64)
    elif case == "norm_mi":
        prob, opts = cm.serial_normal(N=96, T=T, ns=200, min_improve=0.05)
    elif case == "norm_mi_8192":   # 2 x 4096 with a threshold: past the 16-byte lean walk's LDS, so the shards take the global-memory plan (round 5)
        T = 6
        prob, opts = cm.serial_normal(N=8192, T=T, ns=64, min_improve=0.05)
    elif case == "general_np6":
        prob, opts = cm.general_normal(6, 96, T, ns=200, batch_size=3)
    elif case == "banana":
        npar, N = 10, 256
        prob = S.Problem(init=np.zeros(npar), lb=-2 * np.ones(npar), ub=2 * np.ones(npar), mom=np.zeros(npar), w=np.ones(npar), ns=1,
                         objective_id=A.SMM_OBJ_BANANA)
        opts = S.BGPOpts(N=N, maxiter=T, sigma=0.01 * cm.temps(N, 4), acc_tuner=np.geomspace(2.0, 0.1, N), min_improve=np.zeros(N),
                         N_global=N, seed=3, smpl_iters=100000)
    else:
        npar = nm = 50; N = 64
        rng = np.random.default_rng(3)
        prob = S.Problem(init=rng.uniform(-0.3, 0.3, npar), lb=-np.ones(npar), ub=np.ones(npar), mom=rng.uniform(-0.5, 0.5, nm),
                         w=rng.uniform(0.5, 2.0, nm), ns=1, objective_id=A.SMM_OBJ_DENSE)
        opts = S.BGPOpts(N=N, maxiter=T, sigma=0.004 * cm.temps(N, 3), acc_tuner=np.geomspace(20, 1, N), min_improve=np.zeros(N),
                         N_global=N, seed=3, smpl_iters=100000)
    single = S.hip_context(prob, opts)
    single.step(T)
    ctxs = p2p_contexts(S, prob, opts, G)
    p2p_run_lockstep(ctxs, T)
    assert_shards_equal_single(ctxs, single)


def test_p2p_after_other_forms_and_restart(S):
    # the windows are (re)published whenever the context was stepped in another form or a state was uploaded
    G, T = 2, 24
    prob, opts = cm.serial_normal(N=64, T=T, ns=100)
    single = S.hip_context(prob, opts)
    single.step(T)
    ctxs = p2p_contexts(S, prob, opts, G)
    import torch
    R = ctxs[0].record_doubles()
    gathered = torch.zeros((G, 32, R), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    for it in range(6):   # three-phase form first
        for c in ctxs:
            c.local_step()
        for r, c in enumerate(ctxs):
            c.export_records_dev(gathered[r].data_ptr())
        for c in ctxs:
            c.sync()
        for c in ctxs:
            c.exchange_dev(gathered.data_ptr())
        for c in ctxs:
            c.sync()
    p2p_run_lockstep(ctxs, 8)
    saved = [(c.state(), c.history()) for c in ctxs]
    p2p_run_lockstep(ctxs, 3)                # ... thrown away by the upload:
    for c, (st, h) in zip(ctxs, saved):
        c.set_state(st, h)
    p2p_run_lockstep(ctxs, T - 14, chunk=1)
    assert_shards_equal_single(ctxs, single)


def test_p2p_argument_checks(S):
    prob, opts = cm.serial_normal(N=32, T=4, ns=50)
    c = S.hip_context(prob, shard_opts(opts, 2, 0))
    with pytest.raises(A.SMMHipError):
        c.p2p_step(1)                         # no window yet
    _, w = c.p2p_init()
    with pytest.raises(A.SMMHipError):
        c.p2p_step(1)                         # rank 1 not attached
    with pytest.raises(A.SMMHipError):
        c.p2p_attach(0, window=w)             # its own rank
    with pytest.raises(A.SMMHipError):
        c.p2p_attach(5, window=w)


WORKER = r"""
import os, sys, pickle, time
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import smm_jl_amd as S, common as cm
from test_gpu_p2p import shard_opts
rank, G, N, T, ns, d = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
prob, opts = cm.serial_normal(N=N, T=T, ns=ns)
c = S.hip_context(prob, shard_opts(opts, G, rank))
handle, _ = c.p2p_init()
def put(tag, data=b""):
    open(os.path.join(d, "%s_%d.tmp" % (tag, rank)), "wb").write(data); os.rename(os.path.join(d, "%s_%d.tmp" % (tag, rank)), os.path.join(d, "%s_%d" % (tag, rank)))
def get(tag, r):
    p = os.path.join(d, "%s_%d" % (tag, r)); t0 = time.time()
    while not os.path.exists(p):
        time.sleep(0.002)
        if time.time() - t0 > 120: raise SystemExit("rank %d: no %s from rank %d" % (rank, tag, r))
    return open(p, "rb").read()
put("handle", handle)
for r in range(G):
    if r != rank: c.p2p_attach(r, handle=get("handle", r))
put("mapped"); [get("mapped", r) for r in range(G)]
c.p2p_step(T // 2); c.p2p_step(T - T // 2)     # free running: kernels of different processes wait for each other on the device
t0 = time.perf_counter(); c.p2p_finish(); c.sync();
h, st = c.history(), c.state()
put("result", pickle.dumps(({{f: getattr(h, f) for f in h.FIELDS}}, {{f: getattr(st, f) for f in st.FIELDS}})))
[get("result", r) for r in range(G)]            # nobody unmaps a window a peer may still store into
"""


@pytest.mark.parametrize("G,N,ns", [(2, 2048, 1000), (4, 1024, 1000), (2, 512, 64), (2, 16384, 64), (4, 16384, 64), (8, 32768, 64)])
def test_p2p_processes_over_hip_ipc(S, tmp_path, G, N, ns):
    # G processes on the one GPU, 16 chains per workgroup: at most 256 workgroups in all, so that waiting kernels cannot keep
    # the kernels they wait for from starting.  (16384, and 32768 — BASELINE configs[2] as eight processes: the rows form — its chain kernels wait only for records whose slots have
    # arrived, the one waiting workgroup is k_exch_resolve_rows')
    import pickle
    T = 40
    prob, opts = cm.serial_normal(N=N, T=T, ns=ns)
    single = S.hip_context(prob, opts)
    single.step(T)
    hs, ss = single.history(), single.state()
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(G), str(N), str(T), str(ns), str(tmp_path)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(G)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, outs[r][-3000:])
    n = N // G
    for r in range(G):
        h, st = pickle.loads((tmp_path / ("result_%d" % r)).read_bytes())
        for f in A.HistoryBuffers.FIELDS:
            assert np.array_equal(h[f], getattr(hs, f)[..., r * n:(r + 1) * n], equal_nan=True), (f, r)
        for f in A.StateBuffers.FIELDS:
            assert np.array_equal(st[f], getattr(ss, f)[..., r * n:(r + 1) * n], equal_nan=True), (f, r)
    assert (hs.exchanged != 0).any()
