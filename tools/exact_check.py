"""Is a run reproduced BIT FOR BIT?  The HIP path against the oracle on generated randomness (nothing injected): every field of the history and of
the state compared with array_equal — the elementary functions are part of the numerical contract (include/smmhip.h).
  python tools/exact_check.py      (GPU box; test infrastructure)"""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import smm_jl_amd as S, common as cm, bench
from smm_jl_amd import _abi as A
from oracle import oracle as O
from test_gpu_parity import dense_problem

def case(name, prob, opts, T, persistent=True):
    h = S.hip_context(prob, opts)
    if not persistent:
        h.set_persistent(False)
    o = O.OracleContext(prob, opts, S.Tables(Z=h.Z()), threads=16)
    h.step(T); o.step(T)
    hh, oh, hs, os_ = h.history(), o.history(), h.state(), o.state()
    bad = [f for f in cm.INT_FIELDS + cm.F64_FIELDS if not np.array_equal(getattr(hh, f), getattr(oh, f), equal_nan=True)]
    bad += ["state." + f for f in ("sigma", "accept_rate", "la_value", "la_params", "best_val") if not np.array_equal(getattr(hs, f), getattr(os_, f), equal_nan=True)]
    print("%-58s %s  (form %s, launches %d)" % (name, "BIT-IDENTICAL" if not bad else "differs in " + ", ".join(bad), h.describe()["persistent"], h.persistent_info()[1]), flush=True)
    return not bad

ok = True
p, o = cm.serial_normal(N=3, T=200, ns=500); ok &= case("C1 serialNormal 3 chains x 200", p, o, 200)
p, o = cm.serial_normal(N=4096, T=200, ns=10000); ok &= case("C2 4096 chains x 200, ns = 10000 (persistent)", p, o, 200)
p, o = cm.serial_normal(N=4096, T=60, ns=10000); ok &= case("C2 4096 chains x 60 (one launch per iteration)", p, o, 60, False)
p, o = cm.serial_normal(N=4096, T=100, ns=2000, min_improve=0.05); ok &= case("C2 size, min_improve 0.05 (16-byte slots)", p, o, 100)
p, o = bench.build_problem("c4", 8192, 8192, 0, 120, 0); ok &= case("C4 banana 10p, 8192 chains x 120", p, o, 120)
p, o = cm.general_normal(6, N=4096, T=60, ns=2000); ok &= case("objfunc_norm 6 parameters, 4096 chains x 60", p, o, 60)
p, o = cm.general_normal(18, N=333, T=60, ns=1000); ok &= case("objfunc_norm 18 parameters, 333 chains x 60", p, o, 60)
p, o = bench.build_problem("c5", 4096, 4096, 0, 80, 0); ok &= case("C5 dense 50p (bench instance), 4096 chains x 80", p, o, 80)
p, o = bench.build_problem("c3", 32768, 32768, 0, 12, 0); ok &= case("C3 32768 chains x 12, ns = 10000", p, o, 12)
print("all bit-identical" if ok else "NOT all bit-identical")
sys.exit(0 if ok else 1)
