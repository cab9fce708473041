import os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import smm_jl_amd as S, common as cm, bench
S._abi.use_test_hooks(True)
prob, opts = bench.build_problem("c3", 32768, 32768, 0, 800, 0)
ctx = S.hip_context(prob, opts)
ctx.step(200)
t0 = time.perf_counter(); ctx.step_async(400); ctx.sync(); dt = time.perf_counter() - t0
print("%.2f us per iteration" % (dt / 400 * 1e6), flush=True)
import torch
torch.cuda.synchronize(); print("device synchronised", flush=True)
del ctx
print("context destroyed", flush=True)
