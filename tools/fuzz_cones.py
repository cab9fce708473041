"""Randomised sweep of the large-shard exchange (8192 < N <= 32768: the tiles walk locally numbered cones, the plan windows are planned
ahead on a second stream): injected pair lists of random STRUCTURE per iteration — uniform pairs, pairs concentrated on a few hub
chains (deep dependency chains: cones that overflow their caps and send the iteration to the stand-alone resolution), one long chain
of pairs, pairs within the tiles only, repeated pairs — random plan window lengths (test hook), random asynchronous step patterns with
read-backs; against the oracle.
python tools/fuzz_cones.py [cases] [seed]   (GPU box; test infrastructure)"""
import os, sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import smm_jl_amd as S, common as cm
from smm_jl_amd import _abi as A
from oracle import oracle as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
A.use_test_hooks(True)
bad = 0
for it in range(cases):
    N = 16 * int(rng.integers(513, 1100 if it % 4 else 2049))
    T = int(rng.integers(8, 26))
    os.environ["SMMHIP_PLAN_CAP"] = str(int(rng.choice([1, 2, 3, 7, 256])))
    prob, opts = cm.serial_normal(N=N, T=T, ns=8, sigma0=0.02, seed=int(rng.integers(1, 10 ** 6)))
    tab = cm.random_tables(prob, opts, tries=16, seed=int(rng.integers(1, 10 ** 6)))
    K = tab.pairs.shape[1]
    kinds = []
    for t in range(T):
        kind = rng.choice(["uniform", "hub", "chain", "local", "repeat"])
        kinds.append(kind[0])
        if kind == "uniform": continue
        if kind == "hub":       # one end of most pairs among a few chains
            hubs = rng.choice(N, size=int(rng.integers(2, 40)), replace=False)
            i = np.where(rng.random(K) < 0.7, rng.choice(hubs, size=K), rng.integers(0, N, K)); j = rng.integers(0, N, K)
        elif kind == "chain":   # a long dependency chain somewhere, the rest uniform
            L = int(rng.integers(50, 900)); s0 = int(rng.integers(0, N - L - 2))
            i = rng.integers(0, N, K); j = rng.integers(0, N, K)
            at = int(rng.integers(0, K - L))
            i[at:at + L] = s0 + np.arange(L); j[at:at + L] = s0 + 1 + np.arange(L)
        elif kind == "local":   # partners within the own tile of 16 chains
            i = rng.integers(0, N, K); j = (i // 16) * 16 + rng.integers(0, 16, K)
        else:                   # the same few pairs over and over
            m = int(rng.integers(1, 200)); pi = rng.integers(0, N, m); pj = rng.integers(0, N, m)
            sel = rng.integers(0, m, K); i = pi[sel]; j = pj[sel]
        j = np.where(i == j, (j + 1) % N, j)
        lo, hi = np.minimum(i, j), np.maximum(i, j)
        tab.pairs[t, :, 0] = lo; tab.pairs[t, :, 1] = hi
    h = S.hip_context(prob, opts, tab)
    o = O.OracleContext(prob, opts, S.Tables(probs_acc=tab.probs_acc, prop_normals=tab.prop_normals, pairs=tab.pairs, Z=h.Z()), threads=16)
    done, ok, msg = 0, True, ""
    try:
        while done < T:
            n = int(min(T - done, rng.choice([1, 2, 3, 5, 11])))
            h.step_async(n); done += n
            if rng.random() < 0.3: h.sync()
            if rng.random() < 0.15: h.state()
            if rng.random() < 0.15: h.history(max(0, done - 2), done)
        o.step(T)
        cm.assert_history_equal(h.history(), o.history(), atol=1e-13)
        cm.assert_state_equal(h.state(), o.state(), atol=1e-13)
    except (AssertionError, A.SMMHipError) as e:
        ok = False; bad += 1; msg = str(e)[:300]
    ex = float((h.history().exchanged != 0).mean()) if ok else -1.0
    print("case %3d N %5d T %2d cap %3s kinds %s: %s exchanged %.3f %s" % (it, N, T, os.environ["SMMHIP_PLAN_CAP"], "".join(kinds), "ok" if ok else "FAILED", ex, msg), flush=True)
    del h, o
print("%d of %d cases failed" % (bad, cases))
sys.exit(1 if bad else 0)
