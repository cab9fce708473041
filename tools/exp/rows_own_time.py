"""eight shards of 4096 chains as eight contexts of ONE process on one GPU, stepped in lockstep (what each of 8 GPUs launches for C3):
run under rocprofv3 --kernel-trace --stats for the per-launch durations of k_chain_iter_norm_p2p_rows and k_exch_resolve_rows<., true, true>"""
import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import smm_jl_amd as S, common as cm
from test_gpu_p2p import p2p_contexts, p2p_run_lockstep
G, N, T = int(sys.argv[1]) if len(sys.argv) > 1 else 8, int(sys.argv[2]) if len(sys.argv) > 2 else 32768, 40
prob, opts = cm.serial_normal(N=N, T=T, ns=10000)
ctxs = p2p_contexts(S, prob, opts, G)
for it in range(T):          # one context at a time: nothing of another context runs next to the kernels being timed
    for c in ctxs:
        c.p2p_step(1); c.sync()
for c in ctxs: c.p2p_finish()
for c in ctxs: c.sync()
print("done", G, N, T)
