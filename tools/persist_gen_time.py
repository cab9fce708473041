"""The persistent chain kernel of simulation-free objectives (k_chain_persist_gen) against the one-launch-per-iteration kernel on
BASELINE config 4 (banana, 10 parameters, 8192 chains): us per iteration over K steps of 200, the kernel's in-kernel phase times
(SMMHIP_TS=1: accumulated wall-clock stamps of wave 0 of every workgroup), and a bit-exact comparison of the two histories.
  [PG_INSTANCE=bench] python tools/persist_gen_time.py [steps] [chains] [parameters]"""
import ctypes as C
import os
import sys
import time

import numpy as np

os.environ.setdefault("SMMHIP_TS", "1")
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import smm_jl_amd as S, common as cm
from smm_jl_amd import _abi as A

K = int(sys.argv[1]) if len(sys.argv) > 1 else 5
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
NP = int(sys.argv[3]) if len(sys.argv) > 3 else 10
IT = 200
lib = S._abi.load()
hist = {}
for on in (1, 0, 1):
    if os.environ.get("PG_INSTANCE") == "bench":   # bench.py's C4 instance (smm.jl_amd/workloads.py)
        from smm_jl_amd.workloads import build_problem
        prob, opts = build_problem("c4", N, N, 0, IT * (K + 1), 0)
    else:
      prob = S.Problem(init=np.full(NP, 1.2), lb=-2 * np.ones(NP), ub=2 * np.ones(NP), mom=np.zeros(NP), w=np.ones(NP), ns=1, objective_id=A.SMM_OBJ_BANANA)
      opts = S.BGPOpts(N=N, maxiter=IT * (K + 1), sigma=0.02 * cm.temps(N, 5), acc_tuner=np.geomspace(20, 1, N), min_improve=np.zeros(N), seed=3)
    ctx = S.hip_context(prob, opts)
    ctx.set_persistent(on)
    ctx.step(IT)
    t0 = time.perf_counter()
    for _ in range(K):
        ctx.step_async(IT)
    ctx.sync()
    dt = time.perf_counter() - t0
    avail, launches, repairs = ctx.persistent_info()
    print("persistent %d: %.2f us per iteration, %.1f M chain-evals/s   (launches of the persistent kernel %d, repairs %d)"
          % (on, dt / (K * IT) * 1e6, N * K * IT / dt / 1e6, launches, repairs))
    if on and launches:
        tiles = N // 32
        buf = np.zeros((tiles, 8), np.uint64)
        lib.smm_debug_ts(ctx._ctx, buf.ctypes.data_as(C.c_void_p), tiles)
        nit = int(buf[0, 6])
        ph = buf[:, :6].astype(np.float64).mean(axis=0) / 100.0 / nit   # the LAST launch's sums (100 MHz ticks) -> us per iteration
        print("   phases of the last launch, %d iterations (us per iteration, mean over workgroups):" % nit, " wait at the barrier (gather) %.2f | walk %.2f | barrier + donor record %.2f | proposal %.2f | "
              "objective, accept, publish %.2f | bookkeeping %.2f | sum %.2f" % (*ph, ph.sum()))
        per = buf[:, :6].astype(np.float64) / 100.0 / nit
        busy = per[:, 1:].sum(axis=1)
        print("   over the workgroups (min / mean / max): walk %.2f / %.2f / %.2f | everything but the wait %.2f / %.2f / %.2f | wait %.2f / %.2f / %.2f"
              % (per[:, 1].min(), per[:, 1].mean(), per[:, 1].max(), busy.min(), busy.mean(), busy.max(), per[:, 0].min(), per[:, 0].mean(), per[:, 0].max()))
    h = ctx.history()
    hist[on] = h
    del ctx
for f in cm.INT_FIELDS + cm.F64_FIELDS:
    assert np.array_equal(getattr(hist[1], f), getattr(hist[0], f), equal_nan=True), f
print("histories of the two forms: bit-identical (%d iterations x %d chains); exchanged %.3f, accepted %.3f"
      % (hist[1].value.shape[0], N, (hist[1].exchanged != 0).mean(), hist[1].accepted.mean()))
