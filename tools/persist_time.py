"""The persistent chain kernel against the one-launch-per-iteration kernel on the headline configuration (4096 chains, 2p/2m,
ns = 10000): us per iteration over K steps of 200, the persistent kernel's in-kernel phase times (SMMHIP_TS=1: accumulated
wall-clock stamps of every tile's control wave), and a bit-exact comparison of the two histories.
  python tools/persist_time.py [steps] [chains]"""
import ctypes as C
import os
import sys
import time

import numpy as np

os.environ.setdefault("SMMHIP_TS", "1")
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import smm_jl_amd as S, common as cm

K = int(sys.argv[1]) if len(sys.argv) > 1 else 5
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
IT = 200
if os.environ.get("SMM_TEST_BUILD") == "hooks":   # the test build: its seams (SMMHIP_PERSIST_LOC=1: the locally numbered kernel on this problem) are live
    S._abi.use_test_hooks(True)
    lib = S._abi.load_hooks()
else:
    lib = S._abi.load()
MI = float(os.environ.get("PT_MIN_IMPROVE", "0"))
hist = {}
for on in (1, 0, 1):
    prob, opts = cm.serial_normal(N=N, T=IT * (K + 1), min_improve=MI)
    ctx = S.hip_context(prob, opts)
    ctx.set_persistent(on)
    ctx.step(IT)
    t0 = time.perf_counter()
    for _ in range(K):
        ctx.step_async(IT)
    ctx.sync()
    dt = time.perf_counter() - t0
    avail, launches, repairs = ctx.persistent_info()
    print("persistent %d: %.2f us per iteration, %.1f M chain-evals/s   (launches of the persistent kernel %d, repairs %d)"
          % (on, dt / (K * IT) * 1e6, N * K * IT / dt / 1e6, launches, repairs))
    if on and launches:
        tiles = (N + 15) // 16
        buf = np.zeros((tiles, 8), np.uint64)
        lib.smm_debug_ts(ctx._ctx, buf.ctypes.data_as(C.c_void_p), tiles)
        nit = int(buf[0, 6])
        ph = buf[:, :6].astype(np.float64).mean(axis=0) / 100.0 / nit   # the LAST launch's sums (100 MHz ticks) -> us per iteration
        print("   phases of the last launch, %d iterations (us per iteration, mean over tiles):" % nit, " wait at the barrier (gather) %.2f | walk %.2f | record %.2f | settle+proposal %.2f | "
              "B2+simulation %.2f | accept+publish %.2f | sum %.2f" % (*ph, ph.sum()))
    h = ctx.history()
    hist[on] = h
    del ctx
for f in cm.INT_FIELDS + cm.F64_FIELDS:
    assert np.array_equal(getattr(hist[1], f), getattr(hist[0], f), equal_nan=True), f
print("histories of the two forms: bit-identical (%d iterations x %d chains); exchanged %.3f, accepted %.3f"
      % (hist[1].value.shape[0], N, (hist[1].exchanged != 0).mean(), hist[1].accepted.mean()))
