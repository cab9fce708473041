import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import smm_jl_amd as S, common as cm
from smm_jl_amd import _abi as A
if os.environ.get("SMM_TEST_BUILD") == "hooks": A.use_test_hooks(True)
import importlib.util
spec = importlib.util.spec_from_file_location("oracle", "/root/repo/oracle/oracle.py"); O = importlib.util.module_from_spec(spec); spec.loader.exec_module(O)
N, T = int(sys.argv[1]), int(sys.argv[2])
npar = 10
prob = S.Problem(init=np.full(npar, 1.2), lb=-2 * np.ones(npar), ub=2 * np.ones(npar), mom=np.zeros(npar), w=np.ones(npar), ns=1, objective_id=A.SMM_OBJ_BANANA)
opts = S.BGPOpts(N=N, maxiter=T, sigma=0.02 * cm.temps(N, 5), acc_tuner=np.geomspace(20, 1, N), min_improve=np.zeros(N), seed=3)
h = S.hip_context(prob, opts); h.step(T)
o = O.OracleContext(prob, opts, None); o.step(T)
hh, ho = h.history(), o.history()
for f in A.HistoryBuffers.FIELDS:
    a, b = getattr(hh, f), getattr(ho, f)
    bad = np.argwhere(~(np.isclose(a, b, rtol=1e-9, atol=0) | ((a != a) & (b != b))))
    if len(bad): print(f, len(bad), "differ; first", bad[0].tolist(), "iterations", sorted(set(bad[:, 0].tolist()))[:8], a[tuple(bad[0])], b[tuple(bad[0])])
print("done", N, T, "exchanged frac", (hh.exchanged != 0).mean())
