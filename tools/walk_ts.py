import sys, os, ctypes as C
import numpy as np
os.environ["SMMHIP_TS"] = "1"
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import smm_jl_amd as S, common as cm
prob, opts = cm.serial_normal(N=4096, T=700)
ctx = S.hip_context(prob, opts)
ctx.step(150)
lib = S._abi.load()
buf = np.zeros((512, 8), np.uint64)
lib.smm_debug_ts(ctx._ctx, buf.ctypes.data_as(C.c_void_p), 512)
ts = buf.astype(np.float64) / 100.0
for nm, a, b in [("walk loads+stage", 0, 5), ("walk levels", 5, 6), ("level-2 + puts + barrier", 6, 1)]:
    d = ts[:, b] - ts[:, a]
    print("%-28s mean %.2f min %.2f max %.2f" % (nm, d.mean(), d.min(), d.max()))
