"""The error / rollback machine ENUMERATED, not sampled (VERDICT r5 "Next #6").  The reference aborts run! inside the failing iteration
(`error(...)` at AlgoBGP.jl:341 / :409); the library's contract (include/smmhip.h): the failing iteration completes, the error
surfaces at the next call that checks — with the failing iteration in its message —, `iter` stands at the failing iteration, the history
before it is the run's, and nothing is ever lost, whatever was enqueued around it.  Round 5's last sweep found such a loss by chance
(3 of ~1000 random cases: a hard error raised by a one-iteration launch AHEAD of a persistent launch on the stream, fixed in 608e3b4).
Here every sequence of up to three calls out of

    s1  step_async(1)            sn  step_async(5)  (>= 2: the persistent form)          sw  step_async(11)  (crosses a plan window: the test build's
    rb  read-back (state)        ss  state round trip (get_state -> set_state)                 SMMHIP_PLAN_CAP = 8 makes windows of 8 iterations)
    tp  toggle smm_set_persistent

is run, behind one settled iteration, with AlgoBGP.jl:409 injected (no draw in support: the proposal table of one iteration is 1e9) at the FIRST
and at the LAST iteration of every stepping call of the sequence — inside a launch, at its edges, in a launch ahead of others on the stream —, on
the persistent forms `loc` (objfunc_norm, 2 parameters), `tile` (6 parameters) and `gen` (banana) (the latter two: sequences of up to two calls), and
for a shard of two PROCESSES (`loc_shard`: calls s1 / sn / rendezvous).  Every case must end with THAT error at THAT iteration, `iter` there, and the
oracle's history in front of it.  Verified against a build with 608e3b4's fix taken out (check_device_error's "the word was raised before the first
persistent launch" branch disabled): 72 of the 605 `loc` cases then lose their error (`s1 sn`, `rb sn`, `ss sw`, ...: profiles/r06_error_enumeration.txt).
What the enumeration found on its first run (fixed in the same commit): every `... step ss` sequence lost its error in smm_set_state (a failure nobody had
been told of was cleared by the upload: the call now reports it, once); a chain without a draw could be reported as "negative objective" (the two kinds'
order in the error word; ERRK_* in smm_params.hpp); a shard's read-back behind a failing iteration turned the hard error into an internal one."""
import itertools
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest

import common as cm
from smm_jl_amd import _abi as A

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STEP = {"s1": 1, "sn": 5, "sw": 11}
OPS = ("s1", "sn", "sw", "rb", "ss", "tp")


def cases(ops, maxlen):
    """(sequence, failing iteration) pairs: the first and the last iteration of every stepping call (iteration 1 is the settled start)"""
    out = []
    for L in range(1, maxlen + 1):
        for seq in itertools.product(ops, repeat=L):
            cur, seen = 1, set()
            for op in seq:
                if op in STEP:
                    for tf in {cur + 1, cur + STEP[op]}:
                        if tf not in seen:
                            seen.add(tf); out.append((seq, tf))
                    cur += STEP[op]
    return out


def problem(form, T):
    import smm_jl_amd as S
    if form == "loc":
        prob, opts = cm.serial_normal(N=32, T=T, ns=16, sigma0=0.01)
    elif form == "tile":
        prob, opts = cm.general_normal(6, N=32, T=T, ns=16)
        opts.sigma[:] = 0.01
    else:
        npar, N = 3, 64
        prob = S.Problem(init=np.full(npar, 0.5), lb=-2 * np.ones(npar), ub=2 * np.ones(npar), mom=np.zeros(npar), w=np.ones(npar), ns=1, objective_id=A.SMM_OBJ_BANANA)
        opts = S.BGPOpts(N=N, maxiter=T, sigma=0.004 * cm.temps(N, 3), acc_tuner=np.geomspace(20, 1, N), min_improve=np.zeros(N), seed=5)
    return prob, opts


def _iteration_of(e):
    import re
    m = re.search(r"iteration (\d+)", str(e))
    return int(m.group(1)) if m else -1


def run_case(S, h, seq):
    """the sequence on a context that stands behind iteration 1; returns the error that surfaced (or None)"""
    persistent = True
    try:
        for op in seq:
            if op in STEP:
                h.step_async(STEP[op])
            elif op == "rb":
                h.state()
            elif op == "ss":
                st, hi = h.state(), h.history()
                h.set_state(st, hi)
            else:
                persistent = not persistent
                h.set_persistent(persistent)
        h.sync()
    except A.SMMHipError as e:
        return e
    return None


@pytest.mark.parametrize("form,maxlen", [("loc", 3), ("tile", 2), ("gen", 2)])
def test_every_short_call_sequence_with_a_hard_error_at_every_edge(S, O, hooks, monkeypatch, form, maxlen):
    monkeypatch.setenv("SMMHIP_PLAN_CAP", "8")
    T = 1 + 11 * maxlen + 2
    prob, opts = problem(form, T)
    base = cm.random_tables(prob, opts, tries=4, seed=3)
    want = {"loc": "loc", "tile": "tile_sim", "gen": "gen"}[form]
    expect = {}
    todo = cases(OPS, maxlen)
    failures, launches = [], 0
    for seq, tf in todo:
        tab = S.Tables(probs_acc=base.probs_acc, prop_normals=base.prop_normals.copy(), pairs=base.pairs, Z=base.Z)
        tab.prop_normals[tf - 1] = 1e9                 # AlgoBGP.jl:409 in iteration tf
        h = S.hip_context(prob, opts, tab)
        if not expect:
            assert h.describe()["persistent"] == want, h.describe()
        if tf not in expect:                           # the oracle's run up to the failing iteration (one per failing iteration)
            o = O.OracleContext(prob, opts, S.Tables(probs_acc=tab.probs_acc, prop_normals=tab.prop_normals, pairs=tab.pairs, Z=h.Z()))
            with pytest.raises(A.SMMHipError) as eo:
                o.step(T)
            assert eo.value.code == A.SMM_ERR_NO_DRAW_IN_SUPPORT and o.state().iter == tf - 1
            expect[tf] = o.history(0, T)
        h.step(1)
        e = run_case(S, h, seq)
        why = None
        if e is None:
            why = "the error was LOST (iter %d)" % h.state().iter
        elif e.code != A.SMM_ERR_NO_DRAW_IN_SUPPORT or _iteration_of(e) != tf:
            why = "another error: %s" % e
        elif h.state().iter != tf:
            why = "iter stands at %d" % h.state().iter
        else:
            hh, ho = h.history(0, T), expect[tf]
            for f in cm.INT_FIELDS:
                if not np.array_equal(getattr(hh, f)[:tf - 1], getattr(ho, f)[:tf - 1]):
                    why = "history field %s differs before the failing iteration" % f
            if why is None and not np.array_equal(hh.value[:tf - 1], ho.value[:tf - 1], equal_nan=True):
                why = "history values differ before the failing iteration"
            if why is None:
                with pytest.raises(A.SMMHipError):
                    h.step(1)                          # sticky until smm_set_state
        launches += h.persistent_info()[1]
        if why:
            failures.append("%s, failing iteration %d: %s" % (" ".join(seq), tf, why))
        del h
    assert not failures, "%d of %d cases:\n%s" % (len(failures), len(todo), "\n".join(failures[:20]))
    assert launches > len(todo) // 4          # (the persistent form did run in a good part of the cases)


@pytest.mark.parametrize("form", ["loc", "tile", "gen"])
def test_every_short_call_sequence_without_an_error(S, O, hooks, monkeypatch, form):
    # the control: the same calls, nothing injected — the run is the oracle's to the bit, whatever the calls' pattern
    monkeypatch.setenv("SMMHIP_PLAN_CAP", "8")
    T = 1 + 11 * 2
    prob, opts = problem(form, T)
    tab = cm.random_tables(prob, opts, tries=4, seed=3)
    for L in (1, 2):
        for seq in itertools.product(OPS, repeat=L):
            n = 1 + sum(STEP.get(op, 0) for op in seq)
            h = S.hip_context(prob, opts, tab)
            o = O.OracleContext(prob, opts, S.Tables(probs_acc=tab.probs_acc, prop_normals=tab.prop_normals, pairs=tab.pairs, Z=h.Z()))
            h.step(1)
            assert run_case(S, h, seq) is None, seq
            o.step(n)
            cm.assert_history_equal(h.history(), o.history(), exact_floats=True)
            cm.assert_state_equal(h.state(), o.state(), rtol=0)
            del h, o


WORKER = r"""
import os, sys, pickle, time, itertools
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import smm_jl_amd as S, common as cm
from smm_jl_amd import _abi as A
from test_gpu_p2p import shard_opts
from test_gpu_error_enumeration import cases, STEP
rank, G, d = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
def put(tag, data=b""):
    open(os.path.join(d, "%s_%d.tmp" % (tag, rank)), "wb").write(data); os.rename(os.path.join(d, "%s_%d.tmp" % (tag, rank)), os.path.join(d, "%s_%d" % (tag, rank)))
def get(tag, r):
    p = os.path.join(d, "%s_%d" % (tag, r)); t0 = time.time()
    while not os.path.exists(p):
        time.sleep(0.001)
        if time.time() - t0 > 120: raise SystemExit("rank %d: no %s from rank %d" % (rank, tag, r))
    return open(p, "rb").read()
def meet(tag):
    put(tag); [get(tag, r) for r in range(G)]
N, T = 64, 26
prob, opts = cm.serial_normal(N=N, T=T, ns=16, sigma0=0.01)
base = cm.random_tables(prob, opts, tries=4, seed=3)
n = N // G
out = []
for k, (seq, tf) in enumerate(cases(("s1", "sn", "rb"), 2)):
    normals = base.prop_normals.copy(); normals[tf - 1] = 1e9
    tab = S.Tables(probs_acc=np.ascontiguousarray(base.probs_acc[:, rank * n:(rank + 1) * n]), prop_normals=np.ascontiguousarray(normals[..., rank * n:(rank + 1) * n]),
                   pairs=base.pairs, Z=base.Z)
    c = S.hip_context(prob, shard_opts(opts, G, rank), tab)
    handle, _ = c.p2p_init()
    put("h%d" % k, handle)
    for r in range(G):
        if r != rank: c.p2p_attach(r, handle=get("h%d" % k, r))
    meet("m%d" % k)
    err = None
    try:
        c.p2p_step(1); c.p2p_finish(); c.sync(); meet("a%d" % k)
        for j, op in enumerate(seq):
            if op in STEP: c.p2p_step(STEP[op])
            else:
                c.p2p_finish(); c.sync(); meet("r%d_%d" % (k, j))     # (a barrier across the ranks belongs behind a finish: include/smmhip.h)
        c.p2p_finish(); c.sync()
    except A.SMMHipError as e:
        err = (e.code, str(e))
    it = c.state().iter if err is None or err[0] == A.SMM_ERR_NO_DRAW_IN_SUPPORT else -1
    hist = c.history(0, T) if it >= 0 else None
    out.append((seq, tf, err, it, None if hist is None else {{f: getattr(hist, f) for f in cm.INT_FIELDS + ("value",)}}, c.persistent_info()))
    meet("z%d" % k)           # nobody unmaps a window a peer may still store into
    del c
put("result", pickle.dumps(out))
"""


def test_a_shard_of_two_processes_every_short_call_sequence_with_a_hard_error(S, O, tmp_path):
    G = 2
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(G), str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(G)]
    outs = [p.communicate(timeout=400)[0] for p in procs]
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, outs[r][-3000:])
    res = [pickle.loads((tmp_path / ("result_%d" % r)).read_bytes()) for r in range(G)]
    N, T = 64, 26
    prob, opts = cm.serial_normal(N=N, T=T, ns=16, sigma0=0.01)
    base = cm.random_tables(prob, opts, tries=4, seed=3)
    n = N // G
    failures, persistent = [], 0
    for k, (seq, tf) in enumerate(cases(("s1", "sn", "rb"), 2)):
        normals = base.prop_normals.copy(); normals[tf - 1] = 1e9
        o = O.OracleContext(prob, opts, S.Tables(probs_acc=base.probs_acc, prop_normals=normals, pairs=base.pairs, Z=base.Z))
        with pytest.raises(A.SMMHipError):
            o.step(T)
        ho = o.history(0, T)
        for r in range(G):
            seq_r, tf_r, err, it, hist, pinfo = res[r][k]
            assert seq_r == seq and tf_r == tf
            persistent += pinfo[1]
            why = None
            if err is None:
                why = "the error was LOST"
            elif err[0] != A.SMM_ERR_NO_DRAW_IN_SUPPORT or _iteration_of(err[1]) != tf:
                why = "another error: %s" % (err,)
            elif it != tf:
                why = "iter stands at %d" % it
            else:
                for f in cm.INT_FIELDS + ("value",):
                    if not np.array_equal(hist[f][:tf - 1], getattr(ho, f)[:tf - 1, r * n:(r + 1) * n], equal_nan=True):
                        why = "history field %s differs before the failing iteration" % f
            if why:
                failures.append("rank %d: %s, failing iteration %d: %s" % (r, " ".join(seq), tf, why))
    assert not failures, "\n".join(failures[:20])
    assert persistent > 0
