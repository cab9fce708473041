"""Randomised sweep of the HARD-ERROR paths of the persistent chain kernel (AlgoBGP.jl:409 injected at a random iteration through the
proposal tables): random populations, step patterns (asynchronous steps with and without read-backs in between, across the 256-iteration
look-ahead windows), against the oracle — same error, same failing iteration, same history before it.
python tools/fuzz_errors.py [cases] [seed]   (GPU box; test infrastructure)"""
import os, sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import smm_jl_amd as S, common as cm
from smm_jl_amd import _abi as A
from oracle import oracle as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
only = int(sys.argv[3]) if len(sys.argv) > 3 else None
A.use_test_hooks(True)
for it in range(cases):
    os.environ.pop("SMMHIP_PLAN_CAP", None)
    N = int(rng.choice([16, 40, 64, 200, 400]))
    T = int(rng.integers(280, 700))
    tfail = int(rng.integers(5, T - 2))
    if it % 5 == 4:   # the persistent kernel of simulation-free objectives (banana, 4096 < N <= 8192 in whole workgroups of 32)
        N, npar = 32 * int(rng.integers(129, 140)), int(rng.choice([2, 3]))
        T = int(rng.integers(270, 330)); tfail = int(rng.integers(5, T - 2))
        prob = S.Problem(init=np.full(npar, 0.5), lb=-2 * np.ones(npar), ub=2 * np.ones(npar), mom=np.zeros(npar), w=np.ones(npar), ns=1, objective_id=A.SMM_OBJ_BANANA)
        opts = S.BGPOpts(N=N, maxiter=T, sigma=0.004 * cm.temps(N, 3), acc_tuner=np.geomspace(20, 1, N), min_improve=np.zeros(N), seed=int(rng.integers(1, 10 ** 6)))
        tab = cm.random_tables(prob, opts, tries=4, seed=int(rng.integers(1, 10 ** 6)))
    elif it % 5 == 3:   # large shards: the tiles walk their cones, the plan windows (short ones: test hook) are planned ahead on the second stream
        N, T = 16 * int(rng.integers(513, 600)), int(rng.integers(30, 90)); tfail = int(rng.integers(5, T - 2))
        os.environ["SMMHIP_PLAN_CAP"] = str(int(rng.choice([1, 4, 11, 256])))
        prob, opts = cm.serial_normal(N=N, T=T, ns=8, sigma0=0.01, seed=int(rng.integers(1, 10 ** 6)))
        tab = cm.random_tables(prob, opts, tries=8, seed=int(rng.integers(1, 10 ** 6)))
    else:
        prob, opts = cm.serial_normal(N=N, T=T, ns=int(rng.choice([16, 100])), sigma0=0.01, seed=int(rng.integers(1, 10 ** 6)))
        tab = cm.random_tables(prob, opts, tries=8, seed=int(rng.integers(1, 10 ** 6)))
    tab.prop_normals[tfail - 1] = 1e9
    h = S.hip_context(prob, opts, tab)
    o = O.OracleContext(prob, opts, S.Tables(probs_acc=tab.probs_acc, prop_normals=tab.prop_normals, pairs=tab.pairs, Z=h.Z()), threads=16)
    eh = None
    done = 0
    log = []
    try:
        while done < T:
            n = int(min(T - done, rng.choice([1, 2, 7, 50, 120, 300])))
            h.step_async(n)
            r1, r2 = rng.random(), rng.random()
            log.append("step_async(%d)%s%s -> %d" % (n, " sync" if r1 < 0.6 else "", " state" if r2 < 0.2 else "", done + n))
            if r1 < 0.6: h.sync()
            if r2 < 0.2: h.state()
            done += n
        h.sync()
    except A.SMMHipError as e:
        eh = e
    log.append("hip: %s | iter %d | info %s | %s" % (eh, h.state().iter, h.persistent_info(), h.describe()))
    eo = None
    try:
        o.step(T)
    except A.SMMHipError as e:
        eo = e
    import re
    if eh is not None:   # (a table of 8 tries may fail earlier all by itself: wherever the oracle fails is where the library must)
        tfail = int(re.search(r"iteration (\d+)", str(eh)).group(1))
    it_o = o.state().iter if eo is not None else -1
    ok = eh is not None and eo is not None and eh.code == eo.code and it_o + 1 == tfail and h.state().iter == tfail   # (the oracle stops IN iteration tfail: its counter stands at tfail - 1)
    if ok:
        hh, ho = h.history(0, T), o.history(0, T)
        for f in cm.INT_FIELDS:
            ok = ok and np.array_equal(getattr(hh, f)[:tfail - 1], getattr(ho, f)[:tfail - 1])
        ok = ok and np.allclose(hh.value[:tfail - 1], ho.value[:tfail - 1], rtol=1e-9, equal_nan=True)
    info = h.persistent_info()
    print("case %3d N %4d T %3d tfail %3d: %s (persistent launches %d, repairs %d)%s" % (it, N, T, tfail, "ok" if ok else "FAILED", info[1], info[2], "" if ok else "  oracle: iter %d %s | " % (it_o, str(eo)[:60]) + str(eh)[:100]), flush=True)
    if not ok or only == it:
        print("   seed-independent replay: N=%d T=%d tfail=%d prob/opts seed=%d | " % (N, T, tfail, opts.seed) + " ; ".join(log), flush=True)
    bad += 0 if ok else 1
    del h, o
print("%d of %d cases failed" % (bad, cases))
sys.exit(1 if bad else 0)
