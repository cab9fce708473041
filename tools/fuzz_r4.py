"""Randomised parity sweep of round 4's new paths against the oracle: the persistent chain kernel of simulation-free objectives
(banana, 4096 < N <= 8192 in whole workgroups of 32: random populations, parameters, step patterns, read-backs in between) and the
dense objective's tiles with the exchange in their prologue and wide sigmas (mysample's late tries scouted by lane groups).
python tools/fuzz_r4.py [cases] [seed]   (GPU box; test infrastructure, not part of the product)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import smm_jl_amd as S  # noqa: E402
import common as cm  # noqa: E402
from smm_jl_amd import _abi as A  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    for it in range(cases):
        kind = "banana" if it % 2 == 0 else "dense"
        if kind == "banana":
            npar = int(rng.choice([1, 2, 3, 5, 10, 12]))
            N = 32 * int(rng.integers(129, 257))
            T = int(rng.integers(4, 70))
            prob = S.Problem(init=rng.uniform(-1.5, 1.5, npar), lb=-2 * np.ones(npar), ub=2 * np.ones(npar), mom=rng.uniform(-1, 1, npar), w=np.ones(npar),
                             ns=1, objective_id=A.SMM_OBJ_BANANA)
            opts = S.BGPOpts(N=N, maxiter=T, sigma=float(rng.choice([0.005, 0.02, 0.08])) * cm.temps(N, float(rng.uniform(1, 6))),
                             acc_tuner=np.geomspace(float(rng.uniform(2, 30)), 1, N), min_improve=np.zeros(N), seed=int(rng.integers(1, 1 << 30)),
                             smpl_iters=int(rng.choice([50, 1000, 100000])), sigma_update_steps=int(rng.choice([3, 10, 1000])))
        else:
            npar = int(rng.choice([6, 17, 50]))
            nmd = int(rng.choice([2, 33, 50]))
            N = int(rng.choice([16, 48, 100, 1000, 1600, 4096]))
            T = int(rng.integers(4, 25))
            objp = np.concatenate([rng.standard_normal(A.SMM_DENSE_D * npar) / np.sqrt(npar), rng.standard_normal(nmd * A.SMM_DENSE_D) / np.sqrt(A.SMM_DENSE_D)])
            prob = S.Problem(init=rng.uniform(-0.3, 0.3, npar), lb=-np.ones(npar), ub=np.ones(npar), mom=rng.uniform(-0.5, 0.5, nmd), w=rng.uniform(0.5, 2.0, nmd),
                             ns=1, objective_id=A.SMM_OBJ_DENSE, obj_params=objp)
            opts = S.BGPOpts(N=N, maxiter=T, sigma=float(rng.choice([0.01, 0.04, 0.07])) * cm.temps(N, 2.0), acc_tuner=np.geomspace(20, 1, N),
                             min_improve=np.zeros(N), seed=int(rng.integers(1, 1 << 30)), smpl_iters=100000)
        h = S.hip_context(prob, opts)
        o = O.OracleContext(prob, opts, S.Tables(Z=h.Z()), threads=16)
        left, eh, eo = T, None, None
        try:
            while left > 0:
                n = int(min(left, rng.choice([1, 2, 3, 7, 20, 64])))
                h.step(n); o.step(n); left -= n
                if rng.random() < 0.3:
                    cm.assert_state_equal(h.state(), o.state(), atol=1e-12)
        except A.SMMHipError as e:
            eh = e
            try:
                o.step(n)
            except A.SMMHipError as e2:
                eo = e2
        ok = True
        try:
            if eh is not None:
                assert eo is not None and eo.code == eh.code, (eh, eo)
            else:
                cm.assert_history_equal(h.history(), o.history(), atol=1e-12)
                cm.assert_state_equal(h.state(), o.state(), atol=1e-12)
        except AssertionError as e:
            ok = False; bad += 1
            print("CASE %d FAILED: %s" % (it, str(e)[:300]))
        info = h.persistent_info()
        print("case %3d %-6s np %2d N %5d T %3d: %s  (persistent launches %d, repairs %d%s)" % (it, kind, npar, N, T, "ok" if ok else "FAILED", info[1], info[2],
                                                                                               ", hard error on both sides" if eh is not None else ""), flush=True)
        del h, o
    print("%d of %d cases failed" % (bad, cases))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
