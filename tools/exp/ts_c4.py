"""phase stamps of k_chain_iter<0,16,2,true> (C4: the key form with cones): python tools/exp/ts_c4.py"""
import os, sys, ctypes as C
import numpy as np
os.environ["SMMHIP_TS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.argv = [sys.argv[0], "c4", "50"]
import runpy
g = runpy.run_path(os.path.join(ROOT, "tools", "run_objective.py"))
c, N = g["c"], g["N"]
import smm_jl_amd as S
nt = N // 16
buf = np.zeros((nt, 8), np.uint64)
S._abi.load().smm_debug_ts(c._ctx, buf.ctypes.data_as(C.c_void_p), nt)
ts = buf.astype(np.float64) / 100.0
ok = ts[:, 0] > 0
ts = ts[ok]; t0 = ts[:, 0].min()
print("tiles with stamps:", len(ts), " kernel span %.2f us" % (ts[:, 4].max() - t0))
names = [("start (after first WG)", None, 0), ("start -> staged (5)", 0, 5), ("staged -> after walk/loads (1)", 5, 1), ("-> settle (6)", 1, 6), ("-> proposal (2)", 6, 2),
         ("-> objective (3)", 2, 3), ("-> accept (7)", 3, 7), ("-> stores issued (4)", 7, 4)]
for n, a, b in names:
    d = (ts[:, b] - t0) if a is None else (ts[:, b] - ts[:, a])
    print("%-34s mean %6.2f  min %6.2f  max %6.2f" % (n, d.mean(), d.min(), d.max()))
