import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, smm_jl_amd as S, common as cm
for npar in (3, 4):
    for fast in ("1", "0"):
        os.environ["SMMHIP_NORM_FAST"] = fast
        prob, opts = cm.general_normal(npar, N=4096, T=700, ns=10000)
        c = S.hip_context(prob, opts); c.step(100)
        t0 = time.perf_counter(); c.step(500); dt = time.perf_counter() - t0
        print("np=%d norm_fast=%s: %.1f us per iteration, %.1f M chain-evals/s" % (npar, fast, dt / 500 * 1e6, 4096 * 500 / dt / 1e6), flush=True)
