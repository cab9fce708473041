"""Randomised parity sweep: libsmmhip (default path) against the oracle on random problem shapes.
python tools/fuzz_parity.py [cases] [seed] [big|long]   (GPU box; test infrastructure, not part of the product; `big`: populations of
12000 .. 32768 chains — the stand-alone exchange kernels for 4 and 8 GPUs — with a small objective; `long`: 260 .. 1500 iterations)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import smm_jl_amd as S  # noqa: E402
S._abi.use_test_hooks(True)   # (the SMMHIP_* seams below exist in the test build of the library only)
import common as cm  # noqa: E402
from smm_jl_amd import _abi as A  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    big = len(sys.argv) > 3 and sys.argv[3] == "big"
    long_runs = len(sys.argv) > 3 and sys.argv[3] == "long"
    bad = 0
    only = int(os.environ.get("FUZZ_ONLY", "-1"))   # re-run one case of a sweep (same cases / seed arguments)
    for it in range(cases):
        npar = int(rng.choice([1, 2, 2, 3, 4, 6, 9, 18]))
        N = int(rng.choice([1, 2, 3, 5, 8, 9, 16, 17, 40, 100, 255, 256, 257, 600, 1500, 4096, 4100, 8192, 9001]))
        T = int(rng.integers(3, 50))
        ns = int(rng.choice([1, 17, 64, 511, 512, 513, 1000, 4096, 4097, 10000]))
        if big:
            npar = int(rng.choice([1, 2, 2, 3]))
            N = int(rng.choice([12000, 16384, 20001, 24576, 24577, 30000, 32768]))
            T = int(rng.integers(3, 9))
            ns = int(rng.choice([1, 17, 64]))
        if long_runs:   # several look-ahead windows (256 iterations each), sigma adaptation over many periods
            N = int(rng.choice([3, 16, 100, 600, 4096]))
            T = int(rng.integers(300, 1500)) if N < 4096 else int(rng.integers(260, 400))
            ns = int(rng.choice([1, 17, 64]))
        divs = [d for d in range(1, npar + 1) if npar % d == 0]
        bs = int(rng.choice(divs))
        half = rng.uniform(1.0, 5.0, npar)
        init = rng.uniform(-0.5, 0.5, npar) * half
        mom = rng.uniform(-0.5, 0.5, npar) * half
        w = rng.uniform(0.5, 2.0, npar)
        if rng.random() < 0.2:
            w[rng.integers(npar)] = np.nan
        # the objective: mostly objfunc_norm; now and then banana, the dense simulation (FP64 MFMA), the fault-injecting
        # variant of objfunc_norm (status -2 records), or Cholesky-shaped proposals
        kind = str(rng.choice(["norm"] * 6 + ["banana", "dense", "failbox", "chol"])) if not big else "norm"
        chol = None
        if kind == "banana":
            prob = S.Problem(init=np.zeros(npar), lb=-2 * np.ones(npar), ub=2 * np.ones(npar), mom=np.zeros(npar), w=np.ones(npar), ns=1,
                             objective_id=A.SMM_OBJ_BANANA)
        elif kind == "dense":
            npar = int(rng.choice([3, 6, 17, 50]))
            nmd = int(rng.choice([2, 5, 33, 50]))
            N = min(N, 1500)
            divs_d = [d for d in range(1, npar + 1) if npar % d == 0]
            bs = int(rng.choice(divs_d))
            objp = np.concatenate([rng.standard_normal(A.SMM_DENSE_D * npar) / np.sqrt(npar),
                                   rng.standard_normal(nmd * A.SMM_DENSE_D) / np.sqrt(A.SMM_DENSE_D)])
            prob = S.Problem(init=rng.uniform(-0.3, 0.3, npar), lb=-np.ones(npar), ub=np.ones(npar), mom=rng.uniform(-0.5, 0.5, nmd),
                             w=rng.uniform(0.5, 2.0, nmd), ns=1, objective_id=A.SMM_OBJ_DENSE, obj_params=objp)
        elif kind == "failbox" and npar == 2:
            prob = S.Problem(init=init, lb=-half, ub=half, mom=mom, w=w, ns=ns, objective_id=A.SMM_OBJ_NORM_FAILBOX,
                             obj_params=[0.4 * half[0], 0.6 * half[1]])
        else:
            prob = S.Problem(init=init, lb=-half, ub=half, mom=mom, w=w, ns=ns)
            if kind == "chol":
                Am = rng.standard_normal((npar, npar))
                chol = np.tril(np.linalg.cholesky(Am @ Am.T / npar + 0.5 * np.eye(npar)))
                bs = npar
        mi = float(rng.choice([0.0, 0.0, -0.1, 0.3])) if rng.random() < 0.6 else rng.uniform(-0.2, 0.5, N)
        opts = S.BGPOpts(N=N, maxiter=T, sigma=float(rng.choice([0.02, 0.05, 0.3])) * cm.temps(N, float(rng.choice([1.5, 3.0, 8.0]))),
                         acc_tuner=np.geomspace(10.0, 0.5, N) if N > 1 else np.array([2.0]),
                         min_improve=np.broadcast_to(np.asarray(mi, float), (N,)).copy(), seed=int(rng.integers(1, 1 << 30)),
                         batch_size=bs, sigma_update_steps=int(rng.choice([3, 10])), N_global=N,
                         dist_fun=int(rng.choice([0, 0, 0, 1, 2])), chol_L=chol)
        desc = "case %d: %s np=%d N=%d T=%d ns=%d bs=%d dist_fun=%d mi=%s" % (it, kind, npar, N, T, ns, bs, opts.dist_fun, "per chain" if np.ndim(mi) else mi)
        # now and then as G shards on this one GPU (the three host protocols of the sharded exchange: tests/test_gpu_parity.py)
        G = int(rng.choice([1, 1, 1, 2, 4, 8])) if N >= 16 else 1
        mode = str(rng.choice(["records", "fused", "values"]))
        if G > 1 and N % G:
            G = 1
        if G > 1:
            desc += " sharded %d ways (%s)" % (G, mode)
        if only >= 0 and it != only:   # FUZZ_ONLY=<case>: the same draws, nothing run
            if G == 1:
                rng.integers(1, T)
            continue
        try:
            if G > 1:
                import test_gpu_parity as tg
                run = {"records": tg.sharded_run, "fused": tg.sharded_run_fused, "values": tg.sharded_run_values}[mode]
                ctxs = run(S, prob, opts, G, T)
                o = O.OracleContext(prob, opts, S.Tables(Z=ctxs[0].Z()), threads=O.max_threads())
                o.step(T)
                ho, n = o.history(), N // G
                for r, c in enumerate(ctxs):
                    hr = c.history()
                    for f in cm.INT_FIELDS:
                        np.testing.assert_array_equal(getattr(hr, f), getattr(ho, f)[..., r * n:(r + 1) * n], err_msg="shard %d %s" % (r, f))
                    for f in cm.F64_FIELDS:
                        np.testing.assert_allclose(getattr(hr, f), getattr(ho, f)[..., r * n:(r + 1) * n], rtol=1e-9, atol=1e-12, err_msg="shard %d %s" % (r, f))
                print("ok  ", desc)
                continue
            h = S.hip_context(prob, opts)
            o = O.OracleContext(prob, opts, S.Tables(Z=h.Z()), threads=O.max_threads() if big else 1)
            split = int(rng.integers(1, T))
            h.step(split); h.step(T - split)
            o.step(T)
            cm.assert_history_equal(h.history(), o.history(), rtol=1e-9, atol=1e-12)   # bookkeeping exact; floats 1e-9 / 1e-12
            cm.assert_state_equal(h.state(), o.state(), rtol=1e-9, atol=1e-12)
            print("ok  ", desc)
        except Exception as e:  # noqa: BLE001
            msg = " | ".join(x for x in str(e).splitlines() if x.strip())[:400]
            both_fail = False
            if "no draw in support" in msg or "non-negative" in msg:
                try:
                    o2 = O.OracleContext(prob, opts, S.Tables(Z=S.hip_context(prob, opts).Z()))
                    o2.step(T)
                except Exception as e2:  # noqa: BLE001
                    both_fail = str(e2).splitlines()[0][:40] == msg[:40]
            if both_fail:
                print("ok  ", desc, "(both stop with:", msg[:70] + ")")
            else:
                bad += 1
                print("FAIL", desc, "->", msg)
                if only >= 0 and G == 1 and "h" in dir() and "split" in dir():   # where the two part, and whether the 16-byte-slot walks agree with the oracle
                    for env in ("the failing run itself", None, "0"):
                        if env == "0":
                            os.environ["SMMHIP_KEY_WALK"] = env
                        if env is None or env == "0":
                            h2 = S.hip_context(prob, opts)
                            o2 = O.OracleContext(prob, opts, S.Tables(Z=h2.Z()))
                            h2.step(split); h2.step(T - split); o2.step(T)
                        else:
                            h2, o2 = h, o
                        hh, ho = h2.history(), o2.history()
                        print("  SMMHIP_KEY_WALK=%s split=%d" % (env, split))
                        for f in cm.INT_FIELDS + ("curr_val",):
                            d = np.argwhere(getattr(hh, f) != getattr(ho, f))
                            print("    %-10s %d mismatches, first (iteration-1, chain) %s" % (f, len(d), d[:4].tolist()))
                        d = np.argwhere(hh.exchanged != ho.exchanged)
                        if len(d):   # the first iteration whose exchange differs: both partner rows and the pairs in question
                            t = int(d[0][0])
                            cs = sorted(set(int(c) for tt, c in d if tt == t))
                            print("    iteration %d: chains (0-based) %s: hip partners %s, oracle partners %s" % (t + 1, cs, hh.exchanged[t, cs].tolist(), ho.exchanged[t, cs].tolist()))
                            for q, (i, j) in enumerate(O.gen_pairs(opts.seed, t + 1, N)):
                                if i in cs or j in cs:
                                    print("      pair %d: (%d, %d)  values after the exchange hip %r %r  oracle %r %r  min_improve_i %r" % (
                                        q, i, j, hh.curr_val[t, i], hh.curr_val[t, j], ho.curr_val[t, i], ho.curr_val[t, j], opts.min_improve[i]))
    print("%d cases, %d failures" % (cases, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
