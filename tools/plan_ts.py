"""phase stamps of k_exch_plan (workgroup 0 of the last launch): python tools/plan_ts.py [chains]"""
import ctypes as C, os, sys
import numpy as np
os.environ["SMMHIP_TS"] = "1"
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import smm_jl_amd as S, common as cm
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
prob, opts = cm.serial_normal(N=N, T=600)
ctx = S.hip_context(prob, opts)
ctx.step(300)
x = np.zeros(100, np.uint64)
S._abi.load().smm_debug_ts(ctx._ctx, x.ctypes.data_as(C.c_void_p), -1)
t = x[80:90].astype(np.float64) / 100.0
names = ["pairs sampled", "histogram..ranks", "levels (Jacobi sweeps)", "level sort + lean list", "cone: zero + seed", "cone: propagate", "cone: count", "cone: layout", "cone: scatter + gather"]
for i, n in enumerate(names):
    print("%-28s %8.2f us" % (n, t[i + 1] - t[i]))
print("%-28s %8.2f us" % ("total", t[9] - t[0]))
