"""Which forms a context is given at creation (smm_describe) — the table VERDICT r4 "Next #7" asked for: (objective, N, N_global, parameters,
thresholds, ...) -> per-iteration chain kernel, where exchangeMoves! (AlgoBGP.jl:647-716) is walked, the stand-alone resolution, the
persistent form, the look-ahead plan.  The selection itself lives in smm_ctx_create (smmhip.hip); this file is its specification: a change of
the selection logic that moves a row shows up here, on purpose or not."""
import numpy as np
import pytest

import common as cm
from smm_jl_amd import _abi as A
from test_gpu_p2p import shard_opts

pytestmark = pytest.mark.gpu


def norm(N, mi=0.0, **kw):
    return cm.serial_normal(N=N, T=8, ns=64, min_improve=mi, **kw)


def banana(S, N, npar=10):
    prob = S.Problem(init=np.full(npar, 1.2), lb=-2 * np.ones(npar), ub=2 * np.ones(npar), mom=np.zeros(npar), w=np.ones(npar), ns=1, objective_id=A.SMM_OBJ_BANANA)
    return prob, S.BGPOpts(N=N, maxiter=8, sigma=0.02 * cm.temps(N, 5), acc_tuner=np.geomspace(20, 1, N), min_improve=np.zeros(N), seed=3)


def dense(S, N):
    rng = np.random.default_rng(3)
    prob = S.Problem(init=rng.uniform(-0.3, 0.3, 50), lb=-np.ones(50), ub=np.ones(50), mom=rng.uniform(-0.5, 0.5, 50), w=rng.uniform(0.5, 2.0, 50), ns=1,
                     objective_id=A.SMM_OBJ_DENSE)
    return prob, S.BGPOpts(N=N, maxiter=8, sigma=0.004 * cm.temps(N, 3), acc_tuner=np.geomspace(20, 1, N), min_improve=np.zeros(N), seed=3, smpl_iters=100000)


# (what, chain kernel, walk, exchange, persistent, plan)
TABLE = [
    ("C1: serialNormal, 3 chains",                              lambda S: norm(3),                  "iter_norm", "inline_lean", "lean", "loc", "lds"),
    ("C2: 4096 chains (the headline)",                          lambda S: norm(4096),               "iter_norm", "inline_lean", "lean", "loc", "lds"),
    ("C2 with the reference's default threshold 0.5",           lambda S: norm(4096, 0.5),          "iter_norm_wide", "inline_lean_wide", "lean", "loc_wide", "lds"),
    ("per-chain thresholds (round 6: the persistent form walks them)", lambda S: norm(64, np.linspace(0, 0.5, 64)), "iter_norm_any", "inline_slots", "lvl", "loc_wide", "lds"),
    ("per-chain thresholds, one of them negative",              lambda S: norm(64, np.linspace(-0.1, 0.5, 64)), "iter_norm_any", "inline_slots", "lvl", "none", "lds"),
    ("4097 chains: more tiles than compute units",              lambda S: norm(4112),               "iter_norm_narrow", "standalone", "lean", "none", "lds"),
    ("8192 chains",                                             lambda S: norm(8192),               "iter_norm_narrow", "standalone", "lean", "none", "lds"),
    ("C3: 32768 chains on one GPU",                             lambda S: norm(32768),              "iter_norm_narrow_cone", "cone_local", "rows", "none", "big_ahead"),
    ("40000 chains",                                            lambda S: norm(40000),              "iter_norm_narrow", "standalone", "lvl_big", "none", "big"),
    ("four parameters (general_normal)",                        lambda S: cm.general_normal(4, N=64, T=8, ns=64), "iter_norm", "inline_lean", "lean", "tile_sim", "lds"),
    ("six parameters (the reference's snorm_standard)",         lambda S: cm.general_normal(6, N=64, T=8, ns=64), "iter<sim,8>", "inline_lean16", "lean", "tile_sim", "lds"),
    ("C4: banana, 8192 chains",                                 lambda S: banana(S, 8192),          "iter<gen,16,2>", "inline_keys_cone", "lean", "gen", "lds"),
    ("banana, 2048 chains",                                     lambda S: banana(S, 2048),          "iter<gen,8>", "inline_lean16", "lean", "gen", "lds"),
    ("banana, 1000 chains (not whole groups of 32)",            lambda S: banana(S, 1000),          "iter<gen,8>", "inline_lean16", "lean", "none", "lds"),
    ("C5: dense 50 parameters, 4096 chains",                    lambda S: dense(S, 4096),           "iter<dense,16>", "inline_keys_under_tile", "lean", "tile_dense", "lds"),
]


@pytest.mark.parametrize("row", TABLE, ids=[r[0] for r in TABLE])
def test_forms_of_single_shards(S, row):
    what, build, chain, walk, exch, pers, plan = row
    prob, opts = build(S)
    d = S.hip_context(prob, opts).describe()
    assert (d["chain"], d["walk"], d["exchange"], d["persistent"], d["plan"]) == (chain, walk, exch, pers, plan), (what, d)


SHARDS = [
    ("2 x 2048 (N_global 4096)",                 4096, 2, 0.0,  "loc_shard", "lds"),
    ("2 x 4096 (N_global 8192)",                 8192, 2, 0.0,  "loc_shard", "lds"),
    ("2 x 4096 with the default threshold",      8192, 2, 0.5,  "loc_wide_shard_bigplan", "big"),
    ("2 x 2048 with a threshold",                4096, 2, 0.05, "loc_wide_shard", "lds"),
    ("8 x 4096 (C3 across a node)",              32768, 8, 0.0, "loc_shard_bigplan", "big"),
    ("4 x 8192: more tiles than compute units",  32768, 4, 0.0, "none", "big"),
    ("2 x 1000: not whole tiles",                2000, 2, 0.0,  "none", "lds"),
]


@pytest.mark.parametrize("row", SHARDS, ids=[r[0] for r in SHARDS])
def test_forms_of_shards(S, row):
    what, N, G, mi, pers, plan = row
    prob, opts = norm(N, mi)
    d = S.hip_context(prob, shard_opts(opts, G, G - 1)).describe()
    assert (d["persistent"], d["plan"]) == (pers, plan), (what, d)


def test_form_of_a_user_objective(S):
    from test_user_objective import ar1_problem, AR1_SOURCE, PANEL_SOURCE, panel_problem
    oid = S.register_user_objective(AR1_SOURCE)
    d = S.hip_context(*ar1_problem(S, oid, N=256, T=8)).describe()
    assert (d["chain"], d["persistent"]) == ("user_3launches", "gen_user"), d
    oid2 = S.register_user_objective(PANEL_SOURCE, n_sums=3, lanes=64)     # the map-reduce form: compiled into the persistent TILE kernel (round 6)
    d = S.hip_context(*panel_problem(S, oid2, N=64, T=8)).describe()
    assert (d["chain"], d["persistent"]) == ("user_lanes_3launches", "tile_user"), d
    oid3 = S.register_user_objective(PANEL_SOURCE, n_sums=3, lanes=1024)   # more lanes than a tile has: its own launches
    d = S.hip_context(*panel_problem(S, oid3, N=64, T=8)).describe()
    assert d["persistent"] == "none", d
