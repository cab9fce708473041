import sys, os, time, ast
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import smm_jl_amd as S, common as cm
from smm_jl_amd import _abi as A
if os.environ.get("SMM_TEST_BUILD") == "hooks": A.use_test_hooks(True)
from test_gpu_p2p import p2p_contexts, shard_opts
for (G, N, T) in ast.literal_eval(sys.argv[1]):
    prob, opts = cm.serial_normal(N=N, T=T, ns=64)
    single = S.hip_context(prob, opts); single.step(T); hs = single.history()
    ctxs = p2p_contexts(S, prob, opts, G)
    try:
        for it in range(T):
            for r, c in enumerate(ctxs): c.p2p_step(1)
            for r, c in enumerate(ctxs): c.sync()
        for c in ctxs: c.p2p_finish()
        for c in ctxs: c.sync()
    except Exception as e:
        print(G, N, T, "it", it + 1, "FAILED", str(e)[:200])
    n = N // G
    for r, c in enumerate(ctxs):
        h = c.history()
        for f in A.HistoryBuffers.FIELDS:
            a, b = getattr(h, f), getattr(hs, f)[..., r * n:(r + 1) * n]
            bad = np.argwhere(~((a == b) | ((a != a) & (b != b))))
            if len(bad): print("G", G, "rank", r, f, len(bad), "differ, first", bad[0].tolist(), "iterations", sorted(set(bad[:, 0].tolist()))[:10])
    print(G, N, T, "done")
