import sys, os, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import smm_jl_amd as S, common as cm
from test_gpu_p2p import p2p_contexts, shard_opts
import ast
for (G, N, T) in ast.literal_eval(sys.argv[1]):
    prob, opts = cm.serial_normal(N=N, T=T, ns=64)
    ctxs = p2p_contexts(S, prob, opts, G)
    try:
        for it in range(T):
            for r, c in enumerate(ctxs):
                t0 = time.time(); c.p2p_step(1); c.sync(); dt = time.time() - t0
                print("G", G, "N", N, "it", it + 1, "rank", r, "%.3f s" % dt, flush=True)
                if dt > 0.5: print("G", G, "N", N, "it", it + 1, "rank", r, "took %.2f s" % dt)
        for c in ctxs: c.p2p_finish(); c.sync()
        print(G, N, T, "ok")
    except Exception as e:
        print(G, N, T, "it", it + 1, "rank", r, "FAILED", str(e)[:150])
