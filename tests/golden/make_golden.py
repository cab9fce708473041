#!/usr/bin/env python
"""Generates the golden fixtures in this directory with the build's CPU oracle (oracle/smm_oracle.c).

PARITY UNPINNED against the reference: it is Julia (no interpreter in the build image), its proposals
use a non-seedable RandomDevice and its tests hold no known-answer vector for this path (SURVEY.md
§8c).  The fixtures therefore pin the oracle's own restatement of the reference semantics — inputs
(all randomness injected as tables) and expected outputs — so that the oracle on another machine,
and the HIP path on the GPU, can be checked against a committed record.

    python tests/golden/make_golden.py        # rewrites tests/golden/*.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import common as cm  # noqa: E402
import smm_jl_amd as S  # noqa: E402
from oracle import oracle as O  # noqa: E402
from smm_jl_amd import _abi as A  # noqa: E402


def hist_dict(h, prefix="h_"):
    return {prefix + f: getattr(h, f) for f in A.HistoryBuffers.FIELDS}


def state_dict(s, prefix="s_"):
    d = {prefix + f: getattr(s, f) for f in A.StateBuffers.FIELDS}
    d[prefix + "iter"] = s.iter
    return d


def tables_dict(t):
    return {"t_probs_acc": t.probs_acc, "t_prop_normals": t.prop_normals, "t_pairs": t.pairs, "t_Z": t.Z}


def run_case(name, prob, opts, tab, T, extra=None):
    o = O.OracleContext(prob, opts, tab)
    o.step(T)
    d = {}
    d.update(tables_dict(tab))
    d.update(hist_dict(o.history()))
    d.update(state_dict(o.state()))
    if extra:
        d.update(extra)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
    print(name, "written")


def main():
    rng = np.random.default_rng(20260928)

    # G1: objfunc_norm (ObjExamples.jl:59-116) on a small committed shock matrix, a grid of parameter vectors
    ns = 64
    Z = rng.standard_normal((2, ns))
    prob, opts = cm.serial_normal(N=3, T=1, ns=ns)
    o = O.OracleContext(prob, opts, S.Tables(Z=Z))
    g1, g2 = np.meshgrid(np.linspace(-3, 3, 7), np.linspace(-20, 20, 9))
    params = np.stack([g1.ravel(), g2.ravel()])
    v, sm, st = o.eval_batch(params)
    probu, _ = cm.serial_normal(N=3, T=1, ns=ns, w=(np.nan, np.nan))  # the weight-less branch, :96-97
    vu, smu, _ = O.OracleContext(probu, opts, S.Tables(Z=Z)).eval_batch(params)
    np.savez_compressed(os.path.join(HERE, "g1_objfunc_norm.npz"), Z=Z, params=params, value=v, sim_moments=sm, status=st,
                        value_unweighted=vu)
    print("g1_objfunc_norm written")

    # G1b: the library's default shock matrix (seed 12, ns = 10000): a fingerprint
    Zd = O.gen_Z(12, 2, 10000)
    np.savez_compressed(os.path.join(HERE, "g1b_default_Z.npz"), head=Zd[:, :32], col_sums=Zd.sum(1),
                        strided=Zd[:, ::997])
    print("g1b_default_Z written")

    # G2: a full C1 trajectory (serialNormal: N = 3, T = 200) with every random number injected
    prob, opts = cm.serial_normal(N=3, T=200, ns=500)
    run_case("g2_c1_trajectory", prob, opts, cm.random_tables(prob, opts, tries=24, seed=1), 200)

    # G3: exchange resolution where chains occur in several pairs of one iteration (order dependence)
    N, T = 6, 8
    prob, opts = cm.serial_normal(N=N, T=T, ns=200, acc_tuners=[1.0] * N, min_improve=0.0)
    tab = cm.random_tables(prob, opts, tries=24, seed=2)
    pairs = np.zeros((T, N, 2), np.int32)
    pairs[:] = [[0, 1], [0, 2], [0, 3], [1, 2], [2, 5], [0, 5]]
    tab.pairs = pairs
    tab.probs_acc[:] *= 0.05  # accept almost everything: distinct values on all chains
    run_case("g3_exchange_order", prob, opts, tab, T)

    # G4a: objective "exceptions" (status -2) inside a run
    prob, opts = cm.serial_normal(N=8, T=60, ns=200, objective_id=A.SMM_OBJ_NORM_FAILBOX, obj_params=[-0.2, 0.1])
    run_case("g4a_failbox", prob, opts, cm.random_tables(prob, opts, tries=24, seed=3), 60)

    # G4b: sigma adaptation crossing the 0.234 threshold, 4 parameters updated in batches of 2, min_improve > 0
    prob, opts = cm.general_normal(4, N=10, T=50, ns=128, batch_size=2, sigma_update_steps=5, sigma_adjust_by=0.1)
    opts.min_improve[:] = 0.01
    run_case("g4b_sigma_batches", prob, opts, cm.random_tables(prob, opts, tries=8, seed=4), 50,
             extra={"p_init": prob.init, "p_lb": prob.lb, "p_ub": prob.ub, "p_mom": prob.mom, "p_w": prob.w})

    # G5: banana generalised to 10 dimensions (BASELINE config 4 shape, small)
    npar, N, T = 10, 16, 30
    prob = S.Problem(init=np.zeros(npar), lb=-2 * np.ones(npar), ub=2 * np.ones(npar), mom=np.zeros(npar),
                     w=np.ones(npar), ns=1, objective_id=A.SMM_OBJ_BANANA)
    opts = S.BGPOpts(N=N, maxiter=T, sigma=0.02 * cm.temps(N, 5), acc_tuner=np.geomspace(20, 1, N), min_improve=np.zeros(N))
    run_case("g5_banana10", prob, opts, cm.random_tables(prob, opts, tries=6, seed=5), T)


if __name__ == "__main__":
    main()
