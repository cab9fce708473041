"""What ONE rank of a sharded run spends on its look-ahead plan (the kernels that list its own tiles' cones over the population's pair list),
per window and per iteration: python tools/exp/shard_plan_time.py  (the test build: smm_debug_plan_window)"""
import ctypes as C
import os
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import smm_jl_amd as S, common as cm
from test_gpu_p2p import shard_opts
S._abi.use_test_hooks(True)
lib = S._abi.load_hooks()
lib.smm_debug_plan_window.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]
for G, N in ((1, 4096), (2, 8192), (4, 16384), (8, 32768)):
    for mi in (0.0, 0.05):
        prob, opts = cm.serial_normal(N=N, T=600, ns=10000, min_improve=mi)
        c = S.hip_context(prob, shard_opts(opts, G, 0) if G > 1 else opts)
        ms, w = C.c_double(0), C.c_int(0)
        out = []
        for rep in range(3):
            rc = lib.smm_debug_plan_window(c._ctx, 2 + rep, C.byref(ms), C.byref(w))
            assert rc == 0, rc
            out.append(ms.value)
        print("rank 0 of %d x %d chains (N_global %5d), min_improve %.2f: window of %3d iterations planned in %s ms -> %.2f us per iteration; persistent form available: %s"
              % (G, N // G, N, mi, w.value, " / ".join("%.3f" % x for x in out), min(out) * 1e3 / max(w.value, 1), c.persistent_info()[0]))
        del c
