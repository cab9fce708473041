"""device time of the stand-alone exchange resolution kernel by population size (event brackets, net of the empty bracket)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["SMMHIP_INLINE_WALK"] = "0"
import smm_jl_amd as S, common as cm
S._abi.use_test_hooks(True)   # (the SMMHIP_* seams below exist in the test build of the library only)
MI = float(os.environ.get("EXCH_MIN_IMPROVE", "0"))   # one min_improve for all chains (0: the order-key forms)
for N in [int(a) for a in sys.argv[1:]] or (4096, 8192, 16384, 32768, 65000):
    prob, opts = cm.serial_normal(N=N, T=60, ns=64, min_improve=MI)
    c = S.hip_context(prob, opts)
    c.step(10)
    c.set_profiling(1)
    c.step(40)
    tm = c.timing()
    print("N=%6d  exchange %.1f us  chain kernel %.1f us" % (N, (tm.exch_kernel_ms - tm.null_bracket_ms) * 1e3 / 40,
                                                           (tm.iter_kernel_ms - tm.null_bracket_ms) * 1e3 / 40))
