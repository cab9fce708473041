"""soak run: the four single-GPU bench workloads over many steps of 200 iterations (160 000 / 60 000 / 8 000 / 2 400 iterations): no time-out, no repair,\nfinite results.  python tools/soak.py   (GPU box; the histories take 63 / 110 / 25 / 8 GB of HBM)"""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import smm_jl_amd as S, common as cm, bench
CASES = (("c2", 4096, 800), ("c4", 8192, 300), ("c3", 32768, 40), ("c5", 4096, 12))
if len(sys.argv) > 2:   # one workload, its number of steps
    CASES = tuple((w, n, int(sys.argv[2])) for (w, n, s_) in CASES if w == sys.argv[1])
for wl, n, steps in CASES:
    T = 200 * steps
    prob, opts = bench.build_problem(wl, n, n, 0, T, 0)
    # (history of T iterations would not fit for the long runs: keep maxiter but the library allocates history [T][N][HW] -> limit T)
    import os
    if os.environ.get("SOAK_SEED"): opts.seed = int(os.environ["SOAK_SEED"])
    ctx = S.hip_context(prob, opts)
    if os.environ.get("SOAK_PERSIST") == "0": ctx.set_persistent(False)
    t0 = time.perf_counter()
    try:
        for s in range(steps):
            ctx.step_async(200)
            if s % 100 == 99 or len(sys.argv) > 3:
                ctx.sync()
                if len(sys.argv) > 3 and s % 20 == 19: print("  step", s + 1, flush=True)
        ctx.sync()
    except S._abi.SMMHipError as e:   # (the algorithm's own hard error — AlgoBGP.jl:409: sigma adapted until no draw falls into the box — ends a long run: C2's seed at 110224)
        print("%s: stopped by %s" % (wl, e), flush=True)
        T = ctx.state().iter
    dt = time.perf_counter() - t0
    info = ctx.persistent_info()
    st = ctx.state()
    print("%s: %d iterations in %.2f s = %.2f us per iteration; persistent launches %d repairs %d; iter %d; finite best %s"
          % (wl, T, dt, dt / T * 1e6, info[1], info[2], st.iter, bool(np.isfinite(st.best_val).all())), flush=True)
    del ctx
