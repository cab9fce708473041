#!/bin/bash
cd $GRAFT_REPO_ROOT
for lib in libsmmhip.so libsmmhip_acq.so; do
  for i in 1 2 3; do
    SMM_TEST_LIB=$lib timeout 300 python - <<P 2>&1 | tail -2
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import smm_jl_amd as S
S._abi.LIB_PATH = os.path.join("smm.jl_amd/csrc", os.environ["SMM_TEST_LIB"])
import numpy as np, common as cm
from oracle import oracle as O
import test_gpu_p2p as T
bad = 0
for (G, N, TT, fe) in [(8, 640, 40, 11), (4, 8192, 8, None), (8, 640, 40, None), (8, 4096, 30, None)]:
    prob, opts = cm.serial_normal(N=N, T=TT, ns=64)
    single = S.hip_context(prob, opts); single.step(TT)
    ctxs = T.p2p_contexts(S, prob, opts, G)
    T.p2p_run_lockstep(ctxs, TT, finish_every=fe)
    try:
        T.assert_shards_equal_single(ctxs, single); print(os.environ["SMM_TEST_LIB"], G, N, "ok")
    except AssertionError as e:
        hs = single.history(); n = N // G
        for r, c in enumerate(ctxs):
            hr = c.history()
            d = np.argwhere(hr.exchanged != hs.exchanged[:, r*n:(r+1)*n])
            if len(d): print(os.environ["SMM_TEST_LIB"], G, N, "rank", r, "first exchanged mismatch at (t, c)", d[0], "of", len(d)); break
        else: print(os.environ["SMM_TEST_LIB"], G, N, "mismatch elsewhere", str(e)[:100])
P
  done
done
