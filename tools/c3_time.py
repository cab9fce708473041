import os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import smm_jl_amd as S, common as cm, bench
S._abi.use_test_hooks(True)
prob, opts = bench.build_problem("c3", 32768, 32768, 0, 800, 0)
ctx = S.hip_context(prob, opts)
ctx.step(200)
t0 = time.perf_counter(); ctx.step_async(400); ctx.sync(); dt = time.perf_counter() - t0
print("SMMHIP_DBG=%s CONE_BIG=%s: %.2f us per iteration" % (os.environ.get("SMMHIP_DBG"), os.environ.get("SMMHIP_CONE_BIG"), dt / 400 * 1e6))
