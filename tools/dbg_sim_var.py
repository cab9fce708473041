import sys, os, ctypes as C
import numpy as np
os.environ["SMMHIP_TS"] = "1"
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import smm_jl_amd as S, common as cm
prob, opts = cm.serial_normal(N=4096, T=700)
ctx = S.hip_context(prob, opts)
ctx.step(150)
lib = S._abi.load()
nwg = 512
buf = np.zeros((nwg, 8), np.uint64)
lib.smm_debug_ts(ctx._ctx, buf.ctypes.data_as(C.c_void_p), nwg)
ts = buf.astype(np.float64) / 100.0
t0 = ts[:, 0].min()
sim = ts[:, 3] - ts[:, 2]
start = ts[:, 0] - t0
simstart = ts[:, 2] - t0
print("sim duration by blockIdx%8:", [round(float(sim[i::8].mean()), 2) for i in range(8)])
print("sim duration by blockIdx//64:", [round(float(sim[i*64:(i+1)*64].mean()), 2) for i in range(8)])
print("sim duration, first 48 tiles:", np.round(sim[:48], 1).tolist())
print("sim start,    first 48 tiles:", np.round(simstart[:48], 1).tolist())
print("corr(sim duration, sim start) = %.2f" % np.corrcoef(sim, simstart)[0, 1])
o = np.argsort(simstart)
print("sim duration sorted by sim start (deciles):", [round(float(sim[o[i*51:(i+1)*51]].mean()), 2) for i in range(10)])
h = np.histogram(sim, bins=10)
print("histogram of sim duration:", h[0].tolist(), np.round(h[1], 1).tolist())
