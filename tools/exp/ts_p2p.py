"""phase stamps of the chain kernel: single shard against the p2p form with one rank (python tools/exp/ts_p2p.py)"""
import sys, os, ctypes as C
import numpy as np
os.environ.setdefault("SMMHIP_TS", "1")
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import smm_jl_amd as S, common as cm
lib = S._abi.load()
def stamps(ctx, nwg=256):
    buf = np.zeros((nwg, 8), np.uint64)
    lib.smm_debug_ts(ctx._ctx, buf.ctypes.data_as(C.c_void_p), nwg)
    ts = buf.astype(np.float64) / 100.0
    seq = [0, 5, 1, 6, 2, 3, 7, 4]
    names2 = ["start -> staged", "walk (+slot read)", "record+settle+proposal", "barrier", "sim", "objective+accept", "stores (+push, arrive)"]
    dd = np.diff(ts[:, seq], axis=1)
    for i, nme in enumerate(names2):
        print("  %-26s %7.2f %7.2f %7.2f" % (nme, dd[:, i].mean(), dd[:, i].min(), dd[:, i].max()))
    print("  kernel span %.2f us" % (ts[:, 4].max() - ts[:, 0].min()))
prob, opts = cm.serial_normal(N=4096, T=700)
c = S.hip_context(prob, opts); c.step(150)
print("single shard:"); stamps(c)
c2 = S.hip_context(prob, opts); c2.p2p_init(); c2.p2p_step(150); c2.sync()
print("p2p, one rank:"); stamps(c2)
