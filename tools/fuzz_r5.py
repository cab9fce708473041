"""Randomised parity sweep of round 5's new paths against the oracle (single shards) and against the single shard (shards):
  loc    k_chain_persist_loc: objfunc_norm with np = 1 / 2, thresholds 0 / > 0 / NaN, any population up to 4096, random step patterns, read-backs,
         injected tables now and then, the failing objective
  gen    k_chain_persist_gen at populations of whole groups of 32 up to 8192 (banana), and with a USER objective compiled into it
  shard  2 / 4 / 8 PROCESSES over HIP IPC on the one GPU (all tiles resident), thresholds, the big plan's local tables now and then
python tools/fuzz_r5.py [cases] [seed]   (GPU box; test infrastructure, not part of the product)"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import smm_jl_amd as S  # noqa: E402
import common as cm  # noqa: E402
from smm_jl_amd import _abi as A  # noqa: E402
from oracle import oracle as O  # noqa: E402
from user_objective_src import AR1_SOURCE  # noqa: E402


def run_steps(h, o, T, rng):
    left, eh, eo = T, None, None
    try:
        while left > 0:
            n = int(min(left, rng.choice([1, 2, 3, 7, 20, 64, 300])))
            h.step(n); o.step(n); left -= n
            if rng.random() < 0.3:
                cm.assert_state_equal(h.state(), o.state(), atol=1e-12)
    except A.SMMHipError as e:
        eh = e
        try:
            o.step(n)
        except A.SMMHipError as e2:
            eo = e2
    return eh, eo


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    oid = S.register_user_objective(AR1_SOURCE)
    O.register_user_objective(AR1_SOURCE, oid)
    for it in range(cases):
        kind = ("loc", "gen", "loc", "user", "shard")[it % 5]
        note = ""
        if kind == "shard":
            import test_gpu_p2p_persist as TP
            import pathlib
            G = int(rng.choice([2, 4, 8]))
            n = 16 * int(rng.integers(1, 256 // G + 1))
            N, T = G * n, int(rng.integers(6, 50))
            mi = float(rng.choice([0.0, 0.0, 0.05, 0.5]))
            big = rng.random() < 0.3
            env = dict(SMM_TEST_BUILD="hooks", SMMHIP_BIG_EXCHANGE="1") if big else None
            ns = int(rng.choice([64, 300, 2000]))
            with tempfile.TemporaryDirectory() as d:
                ok = True
                try:
                    res = TP._run(pathlib.Path(d), G, N, T, ns, mi, "plain", env)
                    TP._check(S, O, res, G, N, T, ns, mi, oracle=N <= 2048)
                except AssertionError as e:
                    ok = False; bad += 1
                    print("CASE %d FAILED: %s" % (it, str(e)[:400]))
            print("case %3d shard  G %d x %4d T %3d ns %5d mi %.2f%s: %s" % (it, G, n, T, ns, mi, " big plan" if big else "", "ok" if ok else "FAILED"), flush=True)
            continue
        if kind == "loc":
            npar = int(rng.choice([1, 2, 2]))
            N = int(rng.choice([2, 17, 64, 333, 1000, 2048, 4096, int(rng.integers(3, 4097))]))
            T = int(rng.integers(4, 90)) if N <= 1000 else int(rng.integers(4, 30))
            ns = int(rng.choice([1, 64, 513, 1000, 10000, 10240]))
            mi = float(rng.choice([0.0, 0.0, 0.002, 0.05, 0.5, np.nan]))
            if npar == 2:
                kw = dict(objective_id=A.SMM_OBJ_NORM_FAILBOX, obj_params=[0.4, 1.2], sigma0=0.3) if rng.random() < 0.25 else {}
                prob, opts = cm.serial_normal(N=N, T=T, ns=ns, min_improve=mi, seed=int(rng.integers(1, 1 << 30)), **kw)
            else:
                prob, opts = cm.general_normal(1, N=N, T=T, ns=ns, seed=int(rng.integers(1, 1 << 30)))
                opts.min_improve[:] = mi
            opts.sigma_update_steps = int(rng.choice([3, 10, 1000]))
            tab = cm.random_tables(prob, opts, tries=int(rng.choice([2, 7, 24])), seed=int(rng.integers(1, 1 << 30))) if (rng.random() < 0.25 and N <= 1000) else None
            note = "np %d ns %5d mi %s%s" % (npar, ns, mi, " tables" if tab is not None else "")
        elif kind == "gen":
            npar = int(rng.choice([1, 2, 3, 5, 10, 16]))
            N = 32 * int(rng.integers(1, 257))
            T = int(rng.integers(4, 70))
            prob = S.Problem(init=rng.uniform(-1.5, 1.5, npar), lb=-2 * np.ones(npar), ub=2 * np.ones(npar), mom=rng.uniform(-1, 1, npar), w=np.ones(npar),
                             ns=1, objective_id=A.SMM_OBJ_BANANA)
            opts = S.BGPOpts(N=N, maxiter=T, sigma=float(rng.choice([0.005, 0.02, 0.08])) * cm.temps(N, float(rng.uniform(1, 6))),
                             acc_tuner=np.geomspace(float(rng.uniform(2, 30)), 1, N), min_improve=np.zeros(N), seed=int(rng.integers(1, 1 << 30)),
                             smpl_iters=int(rng.choice([50, 1000, 100000])), sigma_update_steps=int(rng.choice([3, 10, 1000])))
            tab = None
            note = "np %d" % npar
        else:
            N = 32 * int(rng.integers(1, 65))
            T = int(rng.integers(4, 50))
            fail = None if rng.random() < 0.5 else float(rng.uniform(0.5, 0.9))
            prob = S.Problem(init=[0.3, 1.0], lb=[-0.95, 0.1], ub=[0.95, 3.0], mom=[0.0, 0.12, 0.06], w=[0.05, 0.05, 0.05], ns=1,
                             objective_id=oid, obj_params=[float(rng.choice([10.0, 100.0, 400.0]))] + ([fail] if fail is not None else []))
            opts = S.BGPOpts(N=N, maxiter=T, sigma=float(rng.choice([0.02, 0.05, 0.2])) * cm.temps(N, 4.0), acc_tuner=np.geomspace(3.0, 0.5, N),
                             min_improve=np.zeros(N), seed=int(rng.integers(1, 1 << 30)), N_global=N)
            tab = None
            note = "fail above %s" % fail
        h = S.hip_context(prob, opts, tab)
        t = tab if tab is not None else S.Tables()
        o = O.OracleContext(prob, opts, S.Tables(probs_acc=t.probs_acc, prop_normals=t.prop_normals, pairs=t.pairs, Z=h.Z()), threads=16)
        eh, eo = run_steps(h, o, T, rng)
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
        print("case %3d %-5s N %5d T %3d %s: %s  (persistent: available %s, launches %d, repairs %d%s)" % (
            it, kind, N, T, note, "ok" if ok else "FAILED", info[0], info[1], info[2], ", hard error on both sides" if eh is not None else ""), flush=True)
        del h, o
    print("%d of %d cases failed" % (bad, cases))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
