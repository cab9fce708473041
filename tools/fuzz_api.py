"""Randomised sweep of API call sequences over contexts that run the persistent kernels: asynchronous steps of random lengths mixed
with synchronisations, state / history read-backs, objective batches, switching the persistent form off and on, and a state upload
(into a fresh context) — the final history and state against the oracle stepped straight through.
python tools/fuzz_api.py [cases] [seed]   (GPU box; test infrastructure)"""
import os, sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import smm_jl_amd as S, common as cm
from smm_jl_amd import _abi as A
from oracle import oracle as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
A_ = S._abi
A_.use_test_hooks(True)
for it in range(cases):
    os.environ.pop("SMMHIP_PLAN_CAP", None)
    if it % 4 == 2:   # large shards: plan windows (short ones: test hook) planned ahead on the second stream
        N, T = 16 * int(rng.integers(513, 700)), int(rng.integers(40, 140))
        os.environ["SMMHIP_PLAN_CAP"] = str(int(rng.choice([1, 3, 10, 50])))
        prob, opts = cm.serial_normal(N=N, T=T, ns=16, seed=int(rng.integers(1, 10 ** 6)))
    elif it % 4 == 3:
        N, npar = 32 * int(rng.integers(129, 200)), int(rng.choice([2, 5, 10]))
        T = int(rng.integers(60, 330))
        prob = S.Problem(init=np.full(npar, 0.8), lb=-2 * np.ones(npar), ub=2 * np.ones(npar), mom=np.zeros(npar), w=np.ones(npar), ns=1, objective_id=A.SMM_OBJ_BANANA)
        opts = S.BGPOpts(N=N, maxiter=T, sigma=0.01 * cm.temps(N, 3), acc_tuner=np.geomspace(20, 1, N), min_improve=np.zeros(N), seed=int(rng.integers(1, 10 ** 6)))
    else:
        N = int(rng.choice([16, 48, 200, 1000, 4096]))
        T = int(rng.integers(60, 700)) if N < 4096 else int(rng.integers(60, 300))
        prob, opts = cm.serial_normal(N=N, T=T, ns=int(rng.choice([16, 200])), seed=int(rng.integers(1, 10 ** 6)))
    h = S.hip_context(prob, opts)
    o = O.OracleContext(prob, opts, S.Tables(Z=h.Z()), threads=16)
    done, log = 0, []
    while done < T:
        op = rng.choice(["step", "step", "step", "sync", "state", "hist", "eval", "toggle", "upload"])
        log.append(op)
        if op == "step":
            n = int(min(T - done, rng.choice([1, 2, 3, 9, 40, 130, 290])))
            h.step_async(n); done += n
        elif op == "sync": h.sync()
        elif op == "state": h.state()
        elif op == "hist" and done > 2: h.history(max(0, done - 3), done)
        elif op == "eval": h.eval_batch(rng.uniform(-0.5, 0.5, (prob.np, 5)))
        elif op == "toggle": h.set_persistent(bool(rng.integers(0, 2)))
        elif op == "upload" and done > 0:
            st, hi = h.state(), h.history(0, done)
            h2 = S.hip_context(prob, opts)
            h2.set_state(st, hi)
            h = h2
    o.step(T)
    ok = True
    try:
        cm.assert_history_equal(h.history(), o.history(), atol=1e-12)
        cm.assert_state_equal(h.state(), o.state(), atol=1e-12)
    except AssertionError as e:
        ok = False; bad += 1
        print("CASE %d FAILED: %s\\n   ops: %s" % (it, str(e)[:300], " ".join(log)))
    print("case %3d N %5d np %2d T %3d: %s (%d calls; persistent launches %d, repairs %d)" % (it, N, prob.np, T, "ok" if ok else "FAILED", len(log), h.persistent_info()[1], h.persistent_info()[2]), flush=True)
    del h, o
print("%d of %d cases failed" % (bad, cases))
sys.exit(1 if bad else 0)
