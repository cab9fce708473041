"""Randomised sweep of the p2p form of the sharded iteration (several contexts of one process, stepped in lockstep) against the single
shard: python tools/fuzz_p2p.py [cases] [seed]   (GPU box; test infrastructure).  Populations through all three forms: inline
(N_global <= 8192), rows (<= 32768), generic (thresholds, other objectives, larger populations)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import smm_jl_amd as S
import common as cm
from smm_jl_amd import _abi as A
from test_gpu_p2p import p2p_contexts, p2p_run_lockstep, assert_shards_equal_single

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(cases):
    G = int(rng.choice([1, 2, 2, 3, 4, 4, 8]))
    n = int(rng.choice([1, 5, 16, 17, 100, 512, 1000, 1024, 2048, 3000, 4096, 5000, 8192]))
    while G * n > 40000: n //= 2
    N = G * n
    T = int(rng.integers(3, 30)) if N < 10000 else int(rng.integers(3, 10))
    kind = str(rng.choice(["norm"] * 5 + ["norm_mi", "failbox", "np1", "np3", "np4", "banana"]))
    fe = int(rng.integers(2, T)) if rng.random() < 0.4 and T > 3 else None
    try:
        if kind == "banana":
            npar = int(rng.choice([2, 4, 10]))
            prob = S.Problem(init=np.full(npar, 1.0), lb=-2 * np.ones(npar), ub=2 * np.ones(npar), mom=np.zeros(npar), w=np.ones(npar), ns=1, objective_id=A.SMM_OBJ_BANANA)
            opts = S.BGPOpts(N=N, maxiter=T, sigma=0.02 * cm.temps(N, 4), acc_tuner=np.geomspace(2.0, 0.1, N) if N > 1 else np.array([1.0]), min_improve=np.zeros(N), N_global=N, seed=int(rng.integers(1, 99)), smpl_iters=100000)
        elif kind in ("np1", "np3", "np4"):
            npar = int(kind[2])
            prob, opts = cm.general_normal(npar, N, T, ns=int(rng.choice([17, 64, 300])), batch_size=npar, seed=int(rng.integers(1, 99)))
        else:
            kw = {}
            if kind == "failbox": kw = dict(objective_id=A.SMM_OBJ_NORM_FAILBOX, obj_params=[0.1, 0.3])
            prob, opts = cm.serial_normal(N=N, T=T, ns=int(rng.choice([17, 64, 300])), min_improve=0.05 if kind == "norm_mi" else 0.0, seed=int(rng.integers(1, 99)), **kw)
        single = S.hip_context(prob, opts)
        single.step(T)
        ctxs = p2p_contexts(S, prob, opts, G)
        p2p_run_lockstep(ctxs, T, finish_every=fe)
        hs = single.history()
        if N > 3 and kind not in ("banana", "failbox"):   # (those two may run without a single swap)
            assert (hs.exchanged != 0).any() or T < 3, "no exchange at all"
        G_ = len(ctxs)
        # (assert_shards_equal_single insists on exchanges having happened: not for every random case)
        n_ = hs.value.shape[1] // G_
        for r, c in enumerate(ctxs):
            hr, st, ss = c.history(), c.state(), single.state()
            for f in A.HistoryBuffers.FIELDS:
                assert np.array_equal(getattr(hr, f), getattr(hs, f)[..., r * n_:(r + 1) * n_], equal_nan=True), (f, r)
            for f in A.StateBuffers.FIELDS:
                assert np.array_equal(getattr(st, f), getattr(ss, f)[..., r * n_:(r + 1) * n_], equal_nan=True), (f, r)
        print("ok   case %d: %s G=%d n=%d T=%d finish_every=%s exchanged %.3f" % (it, kind, G, n, T, fe, (hs.exchanged != 0).mean()), flush=True)
    except Exception as e:
        bad += 1
        print("FAIL case %d: %s G=%d n=%d T=%d finish_every=%s: %s" % (it, kind, G, n, T, fe, repr(e)[:300]), flush=True)
print("%d cases, %d failures" % (cases, bad))
sys.exit(1 if bad else 0)
