"""C5 (dense objective, 50 parameters, 4096 chains): where a launch's time goes as sigma adapts — us per iteration by block of 200
iterations, for the forms of mysample's late tries (test build: SMMHIP_SCOUT_AFTER = rounds of one try per lane segment before the
remaining tries are scouted by groups of 8 lanes; a huge value = never).
  python tools/c5_tail.py [blocks] [scout_after ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import smm_jl_amd as S, common as cm
import bench

S._abi.use_test_hooks(True)
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 10
variants = [int(x) for x in sys.argv[2:]] or [1000000, 2]   # (SMMHIP_SCOUT_GL from the environment: 8 or 16 lanes per group)
for sa in variants:
    os.environ["SMMHIP_SCOUT_AFTER"] = str(sa)
    prob, opts = bench.build_problem("c5", 4096, 4096, 0, 200 * blocks, 0)
    ctx = S.hip_context(prob, opts)
    out = []
    for b in range(blocks):
        t0 = time.perf_counter()
        ctx.step_async(200); ctx.sync()
        out.append((time.perf_counter() - t0) / 200 * 1e6)
    h = ctx.history()
    print("scout_after %8d: us per iteration by block of 200: %s | mean %.1f | accepted %.3f  sigma of chain 0 / 4095: %.4f / %.4f"
          % (sa, " ".join("%.0f" % x for x in out), np.mean(out), h.accepted.mean(), ctx.state().sigma[0], ctx.state().sigma[-1]))
    del ctx
