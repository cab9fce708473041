"""host enqueue time of smm_bgp_step_async against the device time of the same step (is the loop launch-bound?)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import smm_jl_amd as S, common as cm
prob, opts = cm.serial_normal(N=4096, T=2400)
c = S.hip_context(prob, opts)
c.step(200)
for _ in range(3):
    t0 = time.perf_counter(); c.step_async(200); t1 = time.perf_counter(); c.sync(); t2 = time.perf_counter()
    print("enqueue of 200 iterations: %.2f ms (%.1f us per launch)   until the device is done: %.2f ms" % ((t1 - t0) * 1e3, (t1 - t0) * 1e6 / 200, (t2 - t0) * 1e3))
