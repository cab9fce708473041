"""chain-evals/s of the other device objectives (BASELINE configs C4, C5, and a user objective): not bench lines,
a table for DESIGN.md.  Run on the GPU box: python tools/bench_objectives.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import smm_jl_amd as S  # noqa: E402
import common as cm  # noqa: E402
from smm_jl_amd import _abi as A  # noqa: E402
from user_objective_src import AR1_SOURCE, PANEL_SOURCE  # noqa: E402


def rate(ctx, N, iters=200, reps=3):
    ctx.step(iters)
    best = 0.0
    for _ in range(reps):
        t0 = time.perf_counter()
        ctx.step(iters)
        best = max(best, N * iters / (time.perf_counter() - t0))
    return best


def main():
    T = 1000
    rows = []
    # C4: banana, 10 parameters, 8192 chains
    npar, N = 10, 8192
    prob = S.Problem(init=np.zeros(npar), lb=-2 * np.ones(npar), ub=2 * np.ones(npar), mom=np.zeros(npar), w=np.ones(npar), ns=1,
                     objective_id=A.SMM_OBJ_BANANA)
    opts = S.BGPOpts(N=N, maxiter=T, sigma=0.01 * cm.temps(N, 4), acc_tuner=np.geomspace(2.0, 0.1, N), min_improve=np.zeros(N),
                     N_global=N, seed=3, smpl_iters=100000)
    rows.append(("C4 banana np=10, N=8192", rate(S.hip_context(prob, opts), N)))
    # C5: dense simulation, np = nm = 50, 4096 chains
    npar = nm = 50; N = 4096
    rng = np.random.default_rng(3)
    prob = S.Problem(init=rng.uniform(-0.3, 0.3, npar), lb=-np.ones(npar), ub=np.ones(npar), mom=rng.uniform(-0.5, 0.5, nm),
                     w=rng.uniform(0.5, 2.0, nm), ns=1, objective_id=A.SMM_OBJ_DENSE2)
    opts = S.BGPOpts(N=N, maxiter=T, sigma=0.004 * cm.temps(N, 3), acc_tuner=np.geomspace(20, 1, N), min_improve=np.zeros(N),
                     N_global=N, seed=3, smpl_iters=100000)
    rows.append(("C5 dense2 (256x256 stage) np=nm=50 (FP64 MFMA), acc_tuner 20..1 (not the bench instance), N=4096", rate(S.hip_context(prob, opts), N)))
    opts = S.BGPOpts(N=N, maxiter=T, sigma=0.0004 * cm.temps(N, 3), acc_tuner=np.geomspace(20, 1, N), min_improve=np.zeros(N),
                     N_global=N, seed=3, smpl_iters=100000)
    rows.append(("C5 dense, 10x smaller proposal steps, N=4096", rate(S.hip_context(prob, opts), N)))
    # user objective: AR(1) with 400 simulated periods, one thread per chain
    oid = S.register_user_objective(AR1_SOURCE)
    N = 4096
    prob = S.Problem(init=[0.3, 1.0], lb=[-0.95, 0.1], ub=[0.95, 3.0], mom=[0.0, 0.12, 0.06], w=[0.05, 0.05, 0.05], ns=1,
                     objective_id=oid, obj_params=[400.0])
    opts = S.BGPOpts(N=N, maxiter=T, sigma=0.05 * cm.temps(N, 4.0), acc_tuner=np.geomspace(3.0, 0.5, N), min_improve=np.zeros(N),
                     seed=5, N_global=N)
    cu = S.hip_context(prob, opts)
    rows.append(("user objective AR(1) T=400, persistent loop (kernel compiled with it inside), N=4096", rate(cu, N)))
    assert cu.persistent_info()[1] >= 1
    cu = S.hip_context(prob, opts)
    cu.set_persistent(False)
    rows.append(("user objective AR(1) T=400, three launches per iteration, N=4096", rate(cu, N)))
    for T_ar in (40,):   # a cheap objective: what the loop itself costs
        prob2 = S.Problem(init=[0.3, 1.0], lb=[-0.95, 0.1], ub=[0.95, 3.0], mom=[0.0, 0.12, 0.06], w=[0.05, 0.05, 0.05], ns=1,
                          objective_id=oid, obj_params=[float(T_ar)])
        cu = S.hip_context(prob2, opts)
        rows.append(("user objective AR(1) T=%d, persistent loop, N=4096" % T_ar, rate(cu, N)))
        cu = S.hip_context(prob2, opts)
        cu.set_persistent(False)
        rows.append(("user objective AR(1) T=%d, three launches per iteration, N=4096" % T_ar, rate(cu, N)))
    # user objective, map-reduce form: a panel of 4096 AR(1) agents x 40 periods per evaluation, 256 lanes per chain
    oid = S.register_user_objective(PANEL_SOURCE, n_sums=3, lanes=256)
    prob = S.Problem(init=[0.3, 1.0], lb=[-0.95, 0.1], ub=[0.95, 3.0], mom=[0.0, 0.12, 0.06], w=[0.05, 0.05, 0.05], ns=1,
                     objective_id=oid, obj_params=[40.0, 4096.0])
    cu = S.hip_context(prob, opts)
    rows.append(("user map-reduce panel 4096x40, 256 lanes, persistent loop (%s), N=4096" % cu.describe()["persistent"], rate(cu, N, iters=100)))
    cu = S.hip_context(prob, opts)
    cu.set_persistent(False)
    rows.append(("user map-reduce panel 4096x40, 256 lanes, three launches per iteration, N=4096", rate(cu, N, iters=100)))
    for agents, lanes in ((256, 64), (1024, 256)):   # cheaper simulations: what the loop itself costs
        oid = S.register_user_objective(PANEL_SOURCE, n_sums=3, lanes=lanes)
        prob = S.Problem(init=[0.3, 1.0], lb=[-0.95, 0.1], ub=[0.95, 3.0], mom=[0.0, 0.12, 0.06], w=[0.05, 0.05, 0.05], ns=1,
                         objective_id=oid, obj_params=[40.0, float(agents)])
        cu = S.hip_context(prob, opts)
        rows.append(("user map-reduce panel %dx40, %d lanes, persistent loop, N=4096" % (agents, lanes), rate(cu, N, iters=100)))
        cu = S.hip_context(prob, opts)
        cu.set_persistent(False)
        rows.append(("user map-reduce panel %dx40, %d lanes, three launches per iteration, N=4096" % (agents, lanes), rate(cu, N, iters=100)))
    for name, r in rows:
        print("%-88s %8.1f M chain-evals/s  (%.1f us per iteration)" % (name, r / 1e6, 1e6 / (r / int(name.split("N=")[1]))))


if __name__ == "__main__":
    main()
