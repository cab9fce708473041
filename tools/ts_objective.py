"""phase stamps of k_chain_iter for BASELINE configs C4 / C5 (general kernel): python tools/ts_objective.py c4|c5"""
import os, sys, ctypes as C
import numpy as np
os.environ["SMMHIP_TS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.argv = [sys.argv[0], sys.argv[1], "50"]
import runpy
g = runpy.run_path(os.path.join(ROOT, "tools", "run_objective.py"))
c, N = g["c"], g["N"]
import smm_jl_amd as S
ct = 16 if sys.argv[1] == "c5" else 8
nwg = (N + ct - 1) // ct
buf = np.zeros((nwg, 8), np.uint64)
S._abi.load().smm_debug_ts(c._ctx, buf.ctypes.data_as(C.c_void_p), nwg)
ts = buf.astype(np.float64) / 100.0
t0 = ts[:, 0].min()
seq = [0, 1, 5, 6, 2, 3, 7, 4]
names = ["loads + LDS stage", "settle prev", "proposal", "barrier", "objective (sim / MFMA)", "objective finish + accept", "stores issued"]
dd = np.diff(ts[:, seq], axis=1)
print("tiles: %d; first start .. last end: %.2f us" % (nwg, ts[:, 4].max() - t0))
print("start of a tile after the first: mean %.2f max %.2f us" % ((ts[:, 0] - t0).mean(), (ts[:, 0] - t0).max()))
for i, n in enumerate(names):
    print("%-28s mean %7.2f  min %7.2f  max %7.2f us" % (n, dd[:, i].mean(), dd[:, i].min(), dd[:, i].max()))
tot = ts[:, 4] - ts[:, 0]
worst = np.argsort(tot)[-6:][::-1]
print("whole tile: mean %.2f  p50 %.2f  p90 %.2f  max %.2f us" % (tot.mean(), *np.percentile(tot, [50, 90]), tot.max()))
for wg in worst:
    print("  tile %4d: %6.2f us = " % (wg, tot[wg]) + "  ".join("%.2f" % x for x in dd[wg]))
