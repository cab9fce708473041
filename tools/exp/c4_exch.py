import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import smm_jl_amd as S, common as cm
from smm_jl_amd import _abi as A
npar, N, T = 10, 8192, 200
for init in (0.0, 0.5, 1.2):
    prob = S.Problem(init=np.full(npar, init), lb=-2 * np.ones(npar), ub=2 * np.ones(npar), mom=np.zeros(npar), w=np.ones(npar), ns=1, objective_id=A.SMM_OBJ_BANANA)
    opts = S.BGPOpts(N=N, maxiter=T, sigma=0.01 * cm.temps(N, 4), acc_tuner=np.geomspace(2.0, 0.1, N), min_improve=np.zeros(N), seed=3, smpl_iters=100000)
    h = S.hip_context(prob, opts); h.step(T); hh = h.history()
    print("init", init, "exchanged frac", (hh.exchanged != 0).mean(), "accepted", hh.accepted.mean(), "value median first/last", np.median(hh.value[0]), np.median(hh.curr_val[-1]))
