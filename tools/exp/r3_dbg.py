import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import smm_jl_amd as S
import numpy as np, common as cm
import test_gpu_p2p as T
G, N, TT = int(sys.argv[1]), int(sys.argv[2]), 4
prob, opts = cm.serial_normal(N=N, T=TT, ns=64)
single = S.hip_context(prob, opts); single.step(TT)
ctxs = T.p2p_contexts(S, prob, opts, G)
T.p2p_run_lockstep(ctxs, TT)
hs = single.history(); n = N // G
ex = np.concatenate([c.history().exchanged for c in ctxs], axis=1)
val = np.concatenate([c.history().value for c in ctxs], axis=1)
cur = np.concatenate([c.history().curr_val for c in ctxs], axis=1)
d = np.argwhere(ex != hs.exchanged)
print("mismatches", len(d))
for t, c in d[:10]:
    print("t", t, "chain", c, "p2p exchanged", ex[t, c], "single", hs.exchanged[t, c], "curr p2p/single", cur[t, c].hex(), hs.curr_val[t, c].hex(), "value", val[t, c], hs.value[t, c])
    for who, e in (("p2p", ex), ("single", hs.exchanged)):
        p = e[t, c] - 1
        if p >= 0: print("   ", who, "partner", p, "its exchanged", e[t, p], "curr after", (cur if who == "p2p" else hs.curr_val)[t, p].hex(), "accepted-values t-1/t", hs.curr_val[t-1, p].hex())
print("all other fields equal:", all(np.array_equal(np.concatenate([getattr(c.history(), f) for c in ctxs], axis=-1), getattr(hs, f), equal_nan=True) for f in ("value", "curr_val", "best_val", "accepted", "params")))
