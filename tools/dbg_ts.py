import sys, os, ctypes as C
import numpy as np
os.environ.setdefault("SMMHIP_TS", "1")   # "2": also a stamp per level of the inline walk (stretches the levels)
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import smm_jl_amd as S, common as cm
prob, opts = cm.serial_normal(N=4096, T=700, min_improve=float(os.environ.get("TS_MIN_IMPROVE", "0")))
ctx = S.hip_context(prob, opts)
ctx.step(150)
lib = S._abi.load()
nwg = int(os.environ.get("TS_NWG", "256"))   # tiles: 256 with k_chain_iter_norm (16-chain tiles), 512 with the general kernel
buf = np.zeros((nwg, 8), np.uint64)
lib.smm_debug_ts(ctx._ctx, buf.ctypes.data_as(C.c_void_p), nwg)
ts = buf.astype(np.float64) / 100.0  # wall_clock64 = 100 MHz -> us
t0 = ts[:, 0].min()
print("phase stamps relative to first WG start (us): mean / min / max over %d WGs" % nwg)
for i, name in enumerate(["start", "after loads", "after proposal+barrier", "after sim", "end"]):
    v = ts[:, i] - t0
    print("%-24s %7.2f %7.2f %7.2f" % (name, v.mean(), v.min(), v.max()))
seq = [0, 5, 1, 6, 2, 3, 7, 4]
names2 = ["start -> walk inputs staged in LDS", "walk levels (+ slot read)", "record load + settle + proposal", "barrier", "sim", "objective+accept", "stores issued"]
tt = ts[:, seq]
dd = np.diff(tt, axis=1)
print("fine phases (us) mean/min/max:")
for i, nme in enumerate(names2):
    print("%-34s %7.2f %7.2f %7.2f" % (nme, dd[:, i].mean(), dd[:, i].min(), dd[:, i].max()))
d = np.diff(ts[:, :5], axis=1)
print("per-WG phase durations (us) mean/min/max:")
for i, name in enumerate(["loads", "settle+proposal", "sim", "finish"]):
    print("%-18s %7.2f %7.2f %7.2f" % (name, d[:, i].mean(), d[:, i].min(), d[:, i].max()))
slow = np.argsort(-d[:, 1])[:8]
print("slowest proposal WGs:", [(int(i), round(float(d[i,1]),1)) for i in slow])
print("kernel span %.2f us" % (ts[:, 4].max() - t0))

x = np.zeros(100, np.uint64)
lib.smm_debug_ts(ctx._ctx, x.ctypes.data_as(C.c_void_p), -1)
xs = x.astype(np.float64) / 100.0
print("resolve kernel (thread 0): loads+init %.2f  levels %.2f  store %.2f us" % (xs[1]-xs[0], xs[3]-xs[1], xs[4]-xs[3]))
print("resolve: %d levels, %d shader cycles over %.2f us -> %.0f MHz" % (int(x[7]), int(x[6]), xs[4]-xs[0], x[6]/(xs[4]-xs[0])))

nl=int(x[7]); print("inline walk of WG 0: %d levels, %d with barriers; cycles per level:" % (nl, int(x[14])), np.diff(x[15:16+nl].astype(np.int64)).tolist())
print("cycles at level ends (thread 0, shader clock):", [int(v) for v in x[15:15+int(x[7])+1]])
print("ltail", int(x[14]))
print("level ends:", [int(v) for v in x[50:50+nl]])
print("per level:", np.diff(x[15:15+int(x[7])+1].astype(np.int64)).tolist())
print("walk of WG 0: %d shader cycles (s_memtime) in %.2f us (s_memrealtime, 100 MHz) -> %.0f MHz" % (int(x[42]) - int(x[40]), (int(x[43]) - int(x[41])) / 100.0, (int(x[42]) - int(x[40])) / max((int(x[43]) - int(x[41])) / 100.0, 1e-9)))
print("probe: 16 dependent ds_read_b32 of one wave: %d cycles -> %.0f per read" % (int(x[44]), int(x[44]) / 16.0))
