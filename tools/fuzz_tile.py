"""Randomised parity sweep of k_chain_persist_tile (smm_chain_persist_tile.hpp) against the oracle AND against the one-launch-per-iteration
kernels (bit for bit): objfunc_norm with 3..40 parameters and the dense simulation with random np / nm, random populations (whole and
ragged tiles), thresholds 0 / > 0 / NaN, proposal batches, random step patterns with read-backs, injected tables now and then, short plan
windows now and then (the hooks build).
python tools/fuzz_tile.py [cases] [seed]   (GPU box; test infrastructure, not part of the product)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import smm_jl_amd as S  # noqa: E402
import common as cm  # noqa: E402
from smm_jl_amd import _abi as A  # noqa: E402
from oracle import oracle as O  # noqa: E402
from test_gpu_parity import dense_problem  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    for it in range(cases):
        dense = it % 3 == 2
        mi = float(rng.choice([0.0, 0.0, 0.002, 0.05, 0.5, np.nan]))
        T = int(rng.integers(4, 60))
        seed = int(rng.integers(1, 1 << 30))
        if dense:
            npar = int(rng.choice([3, 6, 17, 33, 50, 50, 64]))
            nm = int(rng.choice([2, 5, 33, 50, 50, 64]))
            N = 16 * int(rng.choice([1, 2, 3, 7, 16, int(rng.integers(1, 257))]))
            bs = None
            if rng.random() < 0.25:
                d = [k for k in range(1, npar) if npar % k == 0]
                bs = int(rng.choice(d)) if d else None
            v2 = rng.random() < 0.5        # SMM_OBJ_DENSE2: with the 256 x 256 stage (round 6)
            if v2:
                from test_dense2 import dense2_problem
                prob, opts = dense2_problem(npar, nm, N=N, T=T, seed=seed, **({"batch_size": bs} if bs else {}))
            else:
                prob, opts = dense_problem(S, O, npar, nm, N=N, T=T, seed=seed, **({"batch_size": bs} if bs else {}))
            if rng.random() < 0.5:
                opts.sigma = opts.sigma * float(rng.choice([3.0, 10.0]))   # late tries of mysample
                opts.smpl_iters = 100000
            note = "dense%s np %2d nm %2d bs %s" % ("2" if v2 else " ", npar, nm, bs)
            rtol, expect = 1e-9, "tile_dense2" if v2 else "tile_dense"
        else:
            npar = int(rng.choice([3, 4, 5, 6, 6, 18, 18, 32, 40]))
            N = int(rng.choice([2, 16, 17, 48, 333, 1000, 4096, int(rng.integers(3, 2049))]))
            if N > 1000:
                T = min(T, 25)
            ns = int(rng.choice([1, 64, 513, 1000, 10000])) if N <= 1000 else int(rng.choice([64, 513, 2000]))
            bs = None
            if rng.random() < 0.25:
                d = [k for k in range(1, npar) if npar % k == 0]
                bs = int(rng.choice(d)) if d else None
            prob, opts = cm.general_normal(npar, N=N, T=T, ns=ns, seed=seed, batch_size=bs)
            note = "norm  np %2d ns %5d bs %s" % (npar, ns, bs)
            rtol, expect = 1e-9, "tile_sim"
        opts.min_improve[:] = mi
        opts.sigma_update_steps = int(rng.choice([3, 10, 1000]))
        tab = cm.random_tables(prob, opts, tries=int(rng.choice([2, 7, 24])), seed=seed + 1, Z=not dense) if (rng.random() < 0.2 and N <= 512) else None
        try:
            h = S.hip_context(prob, opts, tab)
        except A.SMMHipError as e:   # (a tile that does not fit the LDS is refused at creation: documented, not a parity case)
            print("case %3d %s N %5d: refused at creation (%s)" % (it, note, N, str(e)[:80]), flush=True)
            continue
        c = S.hip_context(prob, opts, tab)
        c.set_persistent(False)
        t = tab if tab is not None else S.Tables()
        o = O.OracleContext(prob, opts, S.Tables(probs_acc=t.probs_acc, prop_normals=t.prop_normals, pairs=t.pairs, Z=h.Z()), threads=16)
        form = h.describe()["persistent"]
        ok, left, eh = True, T, None
        try:
            try:
                while left > 0:
                    n = int(min(left, rng.choice([1, 2, 3, 7, 20, 64])))
                    h.step(n); left -= n
                    if rng.random() < 0.3:
                        h.state()
            except A.SMMHipError as e:
                eh = e
            done = T - left if eh is None else None
            ec = eo = None
            try:
                c.step(T)
            except A.SMMHipError as e:
                ec = e
            try:
                o.step(T)
            except A.SMMHipError as e:
                eo = e
            if eh is not None or ec is not None or eo is not None:
                assert eh is not None and ec is not None and eo is not None and eh.code == ec.code == eo.code, (eh, ec, eo)
            else:
                assert done == T
                cm.assert_history_equal(h.history(), c.history(), exact_floats=True)
                cm.assert_state_equal(h.state(), c.state(), rtol=0)
                cm.assert_history_equal(h.history(), o.history(), rtol=rtol, atol=1e-12)
                cm.assert_state_equal(h.state(), o.state(), rtol=rtol, atol=1e-12)
            # (64 + 64 parameters / moments, or 24 injected tries of 32 parameters: the tile's blocks alone fill the LDS — the per-iteration kernels then)
            assert form == expect or (form == "none" and (max(npar, nm if dense else npar) > 56 or (tab is not None and npar >= 18))), (form, expect)
        except AssertionError as e:
            ok = False; bad += 1
            print("CASE %d FAILED: %s" % (it, str(e)[:400]))
        info = h.persistent_info()
        print("case %3d %s N %5d T %3d mi %s%s: %s  (form %s, launches %d, repairs %d%s)" % (
            it, note, N, T, mi, " tables" if tab is not None else "", "ok" if ok else "FAILED", form, info[1], info[2],
            ", hard error on all sides" if eh is not None else ""), flush=True)
        del h, c, o
    print("%d of %d cases failed" % (bad, cases))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
