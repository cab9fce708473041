"""mysample's late tries in isolation: C5's problem with all sigmas scaled up (many tries per proposal from the first iteration on);
us per iteration for the forms of the late tries (see tools/c5_tail.py).  python tools/c5_tries.py scale iters [scout_after ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import smm_jl_amd as S, common as cm
import bench
S._abi.use_test_hooks(True)
scale = float(sys.argv[1]); iters = int(sys.argv[2])
for sa in [int(x) for x in sys.argv[3:]]:
    os.environ["SMMHIP_SCOUT_AFTER"] = str(sa)
    prob, opts = bench.build_problem("c5", 4096, 4096, 0, iters + 2, 0)
    opts.sigma = opts.sigma * scale
    ctx = S.hip_context(prob, opts)
    ctx.step(2)
    t0 = time.perf_counter(); ctx.step_async(iters); ctx.sync(); dt = time.perf_counter() - t0
    print("scale %.0f scout_after %8d: %.1f us per iteration" % (scale, sa, dt / iters * 1e6))
    del ctx
