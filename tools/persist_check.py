"""The persistent chain kernel against the one-launch-per-iteration path on a sweep of small problems (bit-identical histories and
states, no repairs, the persistent kernel really launched), uneven stepping included.  python tools/persist_check.py"""
import sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import smm_jl_amd as S, common as cm

def run(N, T, ns, steps, tables, seed, on):
    prob, opts = cm.serial_normal(N=N, T=T, ns=ns, seed=seed)
    tab = cm.random_tables(prob, opts, tries=6, seed=seed) if tables else None
    ctx = S.hip_context(prob, opts, tab)
    ctx.set_persistent(on)
    t0 = time.perf_counter()
    for n in steps:
        ctx.step(n)
    dt = time.perf_counter() - t0
    return ctx.history(), ctx.state(), ctx.persistent_info(), dt

bad = 0
for (N, T, ns, steps, tables) in [(17, 40, 200, [40], False), (17, 40, 200, [1, 5, 2, 20, 12], True), (64, 60, 1000, [60], False), (333, 50, 10000, [25, 25], False),
                                  (1000, 30, 300, [30], False), (4096, 300, 10000, [300], False), (100, 600, 64, [600], False), (48, 64, 10240, [3, 61], False),
                                  (2, 30, 100, [30], False), (16, 30, 100, [30], False), (33, 30, 513, [30], False)]:
    ha, sa, ia, ta = run(N, T, ns, steps, tables, 5, 1)
    hb, sb, ib, tb = run(N, T, ns, steps, tables, 5, 0)
    ok = True
    try:
        cm.assert_history_equal(ha, hb, exact_floats=True)
        cm.assert_state_equal(sa, sb, rtol=0)
    except AssertionError as e:
        ok = False
        print("   MISMATCH:", str(e)[:300])
    print("N=%5d T=%4d ns=%6d steps=%s tables=%d: %s  persistent launches %d repairs %d  (%.2fs vs %.2fs)  exchanged %.3f"
          % (N, T, ns, steps, tables, "identical" if ok else "DIFFERENT", ia[1], ia[2], ta, tb, (ha.exchanged != 0).mean()))
    bad += (not ok) or ia[2] != 0 or (ia[1] == 0 and N >= 2 and max(steps) > 2)
print("FAILED" if bad else "all good")
