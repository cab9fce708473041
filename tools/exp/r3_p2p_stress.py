import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import smm_jl_amd as S
import numpy as np, common as cm
import test_gpu_p2p as T
from smm_jl_amd import _abi as A
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
# some allocation churn in front, like the other test files do
for i in range(3):
    prob, opts = cm.serial_normal(N=100 + 37 * i, T=20, ns=100); c = S.hip_context(prob, opts); c.step(20); del c
fails = 0
for rep in range(reps):
    for (G, N, TT) in [(2, 64, 300), (4, 128, 280)]:
        prob, opts = cm.serial_normal(N=N, T=TT, ns=64)
        single = S.hip_context(prob, opts); single.step(TT)
        ctxs = T.p2p_contexts(S, prob, opts, G)
        T.p2p_run_lockstep(ctxs, TT)
        hs = single.history(); n = N // G
        for f in A.HistoryBuffers.FIELDS:
            full = np.concatenate([getattr(c.history(), f) for c in ctxs], axis=-1)
            if not np.array_equal(full, getattr(hs, f), equal_nan=True):
                d = np.argwhere(~((full == getattr(hs, f)) | (np.isnan(full) & np.isnan(getattr(hs, f)))))
                print("rep", rep, (G, N, TT), "field", f, "first mismatch", d[0].tolist(), "count", len(d)); fails += 1
                break
        del ctxs, single
print("failures:", fails, "of", 2 * reps)
