"""debug: what happens around a hard error (AlgoBGP.jl:409) that strikes in the middle of an asynchronous step (C2, seed of bench.py: iteration 110224)"""
import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import smm_jl_amd as S, common as cm, bench
from smm_jl_amd import _abi as A
prob, opts = bench.build_problem("c2", 4096, 4096, 0, 120000, 0)
ctx = S.hip_context(prob, opts)
if os.environ.get("SOAK_PERSIST") == "0": ctx.set_persistent(False)
ctx.step(110200)
print("at", ctx.state().iter, flush=True)
try:
    ctx.step_async(200)
    print("enqueued", flush=True)
    ctx.sync()
    print("synced without an error?", flush=True)
except A.SMMHipError as e:
    print("error:", e, flush=True)
print("state iter", ctx.state().iter, flush=True)
del ctx
print("context destroyed", flush=True)
