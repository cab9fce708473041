"""in-kernel stamps of the stand-alone exchange kernel at a big population: python tools/exch_ts.py [N]"""
import os, sys, ctypes as C
import numpy as np
os.environ["SMMHIP_TS"] = "1"; os.environ["SMMHIP_INLINE_WALK"] = "0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import smm_jl_amd as S, common as cm
S._abi.use_test_hooks(True)   # (the SMMHIP_* seams below exist in the test build of the library only)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
prob, opts = cm.serial_normal(N=N, T=60, ns=64)
c = S.hip_context(prob, opts)
c.step(30)
x = np.zeros(100, np.uint64)
S._abi.load().smm_debug_ts(c._ctx, x.ctypes.data_as(C.c_void_p), -1)
xs = x.astype(np.float64) / 100.0
nl = int(x[7])
print("N=%d: stage %.2f us  levels %.2f us  partner pass %.2f us  output %.2f us  total %.2f us; key shift %d; %d levels"
      % (N, xs[1] - xs[0], xs[2] - xs[1], xs[3] - xs[2], xs[4] - xs[3], xs[4] - xs[0], int(x[14]), nl))
cyc = x[15:16 + nl].astype(np.int64)
print("staging cycles %d; cycles per level:" % cyc[0], np.diff(cyc).tolist())
