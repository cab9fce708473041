import sys, os, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import smm_jl_amd as S, common as cm
prob, opts = cm.serial_normal(N=4096, T=700)
ctx = S.hip_context(prob, opts)
ctx.step(100)
ctx.set_profiling(True)
ctx.step(200)
tm = ctx.timing()
print("DBG=%s chain_iter %.2f us  exch %.2f us  step %.2f ms" % (os.environ.get("SMMHIP_DBG"), tm.iter_kernel_ms*5, tm.exch_kernel_ms*5, tm.step_ms))
