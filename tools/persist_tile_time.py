"""The persistent tile kernel (k_chain_persist_tile: the dense objective of BASELINE config 5, objfunc_norm with more than two parameters)
against the one-launch-per-iteration kernels: us per iteration by block of 200 iterations, the kernel's in-kernel phase times
(SMMHIP_TS=1: accumulated wall-clock stamps of lane 0 of every tile), and a bit-exact comparison of the two histories.
  python tools/persist_tile_time.py [c5 | c5v1 | normP | c4user] [blocks] [chains]        (normP: objfunc_norm with P parameters, ns = 10000)"""
import ctypes as C
import os
import sys
import time

import numpy as np

os.environ.setdefault("SMMHIP_TS", "1")
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import smm_jl_amd as S, common as cm
import bench

BANANA_LANES = r"""
SMM_USER_PARTIAL(const double* theta, int np, const double* udata, int n_udata, int lane, int n_lanes, double* partial) { }
SMM_USER_FINISH(const double* theta, int np, const double* totals, int n_sums, const double* mom, const double* w, int nm,
                const double* udata, int n_udata, double* sim_moments, double* value, int* status)
{
    double v = 0.0;
    for (int i = 0; i + 1 < np; ++i) { const double a = theta[i], b = theta[i + 1], t1 = b - a * a, t2 = 1.0 - a; const double term = 100.0 * (t1 * t1) + t2 * t2; v = i == 0 ? term : v + term; }
    for (int k = 0; k < nm; ++k) sim_moments[k] = mom[k] + 2.2;
    *value = v; *status = 1;
}
"""
what = sys.argv[1] if len(sys.argv) > 1 else "c5"
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 5
N = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
IT = 200
lib = S._abi.load()
hist = {}
for on in (1, 0, 1):
    if what == "c4user":   # BASELINE config 4's instance with the banana written as a USER objective in map-reduce form (64 lanes per chain, nothing to
        # reduce): the tile kernel's 16-chain workgroups, two to a CU at 8192 chains — what a 16-chain form of k_chain_persist_gen would walk
        prob, opts = bench.build_problem("c4", N, N, 0, IT * blocks, 0)
        oid = S.register_user_objective(BANANA_LANES, n_sums=1, lanes=64)
        prob = S.Problem(init=prob.init, lb=prob.lb, ub=prob.ub, mom=prob.mom, w=prob.w, ns=1, objective_id=oid)
    elif what in ("c5", "c5v1"):
        prob, opts = bench.build_problem(what, N, N, 0, IT * blocks, 0)
    else:
        prob, opts = cm.general_normal(int(what[4:]), N=N, T=IT * blocks, ns=10000)
    ctx = S.hip_context(prob, opts)
    ctx.set_persistent(on)
    out = []
    for b in range(blocks):
        t0 = time.perf_counter()
        ctx.step_async(IT); ctx.sync()
        out.append((time.perf_counter() - t0) / IT * 1e6)
    avail, launches, repairs = ctx.persistent_info()
    print("%s persistent %d (%s): us per iteration by block of 200: %s | mean %.2f, %.1f M chain-evals/s   (launches of the persistent kernel %d, repairs %d)"
          % (what, on, ctx.describe()["persistent"], " ".join("%.1f" % x for x in out), np.mean(out), N / np.mean(out), launches, repairs))
    if on and launches:
        tiles = (N + 15) // 16
        buf = np.zeros((tiles, 8), np.uint64)
        lib.smm_debug_ts(ctx._ctx, buf.ctypes.data_as(C.c_void_p), tiles)
        nit = int(buf[0, 7])
        names = ["gather + rows + wait at B0", "walk", "donor + settle", "proposal", "objective", "moments + table", "lists + accept + publish"]
        per = buf[:, :7].astype(np.float64) / 100.0 / max(nit, 1)
        ph = per.mean(axis=0)
        print("   phases of the last launch, %d iterations (us per iteration, mean over tiles): " % nit + " | ".join("%s %.2f" % (n, v) for n, v in zip(names, ph)) +
              " | sum %.2f" % ph.sum())
        print("   over the tiles (min / mean / max): " + " | ".join("%s %.2f / %.2f / %.2f" % (n, per[:, i].min(), per[:, i].mean(), per[:, i].max()) for i, n in enumerate(names)))
    hist[on] = ctx.history()
    del ctx
for f in cm.INT_FIELDS + cm.F64_FIELDS:
    assert np.array_equal(getattr(hist[1], f), getattr(hist[0], f), equal_nan=True), f
print("histories of the two forms: bit-identical (%d iterations x %d chains); exchanged %.3f, accepted %.3f"
      % (hist[1].value.shape[0], N, (hist[1].exchanged != 0).mean(), hist[1].accepted.mean()))
