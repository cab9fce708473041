"""one BASELINE configuration other than the bench's, for profiling: python tools/run_objective.py c4|c5 [iters]
(c4: banana np = nm = 10, 8192 chains; c5: dense simulation np = nm = 50 on FP64 MFMA, 4096 chains)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import smm_jl_amd as S, common as cm
from smm_jl_amd import _abi as A

which = sys.argv[1]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 400
T = iters + 200
if which == "c4":
    npar, N = 10, 8192
    prob = S.Problem(init=np.zeros(npar), lb=-2 * np.ones(npar), ub=2 * np.ones(npar), mom=np.zeros(npar), w=np.ones(npar), ns=1,
                     objective_id=A.SMM_OBJ_BANANA)
    opts = S.BGPOpts(N=N, maxiter=T, sigma=0.01 * cm.temps(N, 4), acc_tuner=np.geomspace(2.0, 0.1, N), min_improve=np.zeros(N),
                     N_global=N, seed=3, smpl_iters=100000)
else:
    npar = nm = 50; N = 4096
    rng = np.random.default_rng(3)
    prob = S.Problem(init=rng.uniform(-0.3, 0.3, npar), lb=-np.ones(npar), ub=np.ones(npar), mom=rng.uniform(-0.5, 0.5, nm),
                     w=rng.uniform(0.5, 2.0, nm), ns=1, objective_id=A.SMM_OBJ_DENSE)
    opts = S.BGPOpts(N=N, maxiter=T, sigma=0.004 * cm.temps(N, 3), acc_tuner=3000.0 * np.geomspace(20, 1, N), min_improve=np.zeros(N),   # (bench.py: C5_ACC_SCALE)
                     N_global=N, seed=3, smpl_iters=100000)
c = S.hip_context(prob, opts)
c.step(200)
t0 = time.perf_counter(); c.step(iters); dt = time.perf_counter() - t0
print("%s: %d chains x %d iterations: %.1f us per iteration, %.1f M chain-evals/s" % (which, N, iters, dt / iters * 1e6, N * iters / dt / 1e6))
