"""The drop-in boundary: libsmmhip.so loads, exports every symbol include/smmhip.h declares, and the
ctypes mirror has the C layout.  No compute calls: there is no GPU in this tier."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import pytest

from smm_jl_amd import _abi as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "smmhip.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(smm_[a-zA-Z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = A.load()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "libsmmhip.so does not export %s" % n
    assert sorted(s[0] for s in A.SYMBOLS) == names  # the ctypes table mirrors the header exactly
    assert lib.smm_abi_version() == 3


def test_ctypes_layout_matches_the_header():
    prog = r'''
#include <stdio.h>
#include <stddef.h>
#include "smmhip.h"
#define P(T) printf(#T " %zu\n", sizeof(T))
#define O(T, f) printf(#T "." #f " %zu\n", offsetof(T, f))
int main(void) {
  P(smm_problem_t); P(smm_bgp_opts_t); P(smm_tables_t); P(smm_history_t); P(smm_state_t); P(smm_timing_t);
  O(smm_problem_t, init); O(smm_problem_t, n_obj_params);
  O(smm_bgp_opts_t, sigma); O(smm_bgp_opts_t, sigma_adjust_by); O(smm_bgp_opts_t, seed); O(smm_bgp_opts_t, device);
  O(smm_tables_t, pairs); O(smm_tables_t, Z); O(smm_history_t, status); O(smm_state_t, best_id); O(smm_timing_t, chain_evals);
  printf("SMM_REDUCE_LANES %d\n", SMM_REDUCE_LANES);
  return 0; }
'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "p.c"), "w").write(prog)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "p.c"), "-o", os.path.join(d, "p")])
        out = dict(l.rsplit(" ", 1) for l in subprocess.check_output([os.path.join(d, "p")]).decode().strip().splitlines())
    for T in (A.smm_problem_t, A.smm_bgp_opts_t, A.smm_tables_t, A.smm_history_t, A.smm_state_t, A.smm_timing_t):
        assert int(out[T.__name__]) == C.sizeof(T), T.__name__
    for key, v in out.items():
        if "." in key:
            t, f = key.split(".")
            assert getattr(getattr(A, t), f).offset == int(v), key
    assert int(out["SMM_REDUCE_LANES"]) == A.SMM_REDUCE_LANES


def test_no_cpu_fallback_without_a_device():
    """the product fails loudly when there is no HIP device (this tier has none)"""
    import smm_jl_amd as S
    import common as cm
    lib = A.load()
    if lib.smm_device_count() > 0:
        pytest.skip("a GPU is present")
    prob, opts = cm.serial_normal(N=3, T=2)
    with pytest.raises(A.SMMHipError) as e:
        S.hip_context(prob, opts)
    assert e.value.code == A.SMM_ERR_NO_DEVICE and "no CPU fallback" in str(e.value)


def test_invalid_arguments_are_rejected_before_touching_the_device():
    import smm_jl_amd as S
    import common as cm
    prob, opts = cm.general_normal(4, N=3, T=4, batch_size=3)
    with pytest.raises(A.SMMHipError) as e:
        S.hip_context(prob, opts)
    assert e.value.code == A.SMM_ERR_BAD_BATCH
    prob = S.Problem(init=[0.0, 0.0], lb=[-1, -1], ub=[1, 1], mom=[0.0], w=[1.0])  # objfunc_norm needs np == nm
    _, opts = cm.serial_normal(N=3, T=2)
    with pytest.raises(A.SMMHipError) as e:
        S.hip_context(prob, opts)
    assert e.value.code == A.SMM_ERR_INVALID_ARG


def test_zero_initialised_structs_and_bad_tables_are_rejected():
    """a C or Julia caller with a zero-initialised opts struct, NULL vectors or an out-of-range injected pair list
    gets SMM_ERR_INVALID_ARG with a message, not a segfault or a silently different algorithm (exchanges from iteration 1)"""
    import numpy as np
    import smm_jl_amd as S
    import common as cm
    lib = A.load()

    def create(prob, opts, tab=None):
        ps, os_ = prob.struct(), opts.struct(prob.np)
        ts = tab.struct() if tab is not None else None
        return ps, os_, ts

    def rc_of(ps, os_, ts=None):
        ctx = C.c_void_p()
        rc = lib.smm_ctx_create(C.byref(ps), C.byref(os_), C.byref(ts) if ts is not None else None, C.byref(ctx))
        assert not ctx.value
        return rc, lib.smm_last_error(None).decode()

    prob, opts = cm.serial_normal(N=3, T=4)
    for field, value, word in (("exchange_from_iter", 0, "exchange_from_iter"), ("exchange_from_iter", 1, "exchange_from_iter"),
                               ("smpl_iters", 0, "smpl_iters"), ("sigma_update_steps", 0, "sigma_update_steps")):
        ps, os_, _ = create(prob, opts)
        setattr(os_, field, value)
        rc, msg = rc_of(ps, os_)
        assert rc == A.SMM_ERR_INVALID_ARG and word in msg, (field, rc, msg)
    for field in ("sigma", "acc_tuner", "min_improve"):
        ps, os_, _ = create(prob, opts)
        setattr(os_, field, None)
        rc, msg = rc_of(ps, os_)
        assert rc == A.SMM_ERR_INVALID_ARG and "NULL" in msg, field
    for field in ("init", "lb", "ub", "mom", "w"):
        ps, os_, _ = create(prob, opts)
        setattr(ps, field, None)
        rc, msg = rc_of(ps, os_)
        assert rc == A.SMM_ERR_INVALID_ARG and "NULL" in msg, field
    zero = A.smm_bgp_opts_t()      # all zeros
    ps, _, _ = create(prob, opts)
    assert rc_of(ps, zero)[0] in (A.SMM_ERR_INVALID_ARG, A.SMM_ERR_BAD_BATCH)
    for bad in ([[0, 3]], [[2, 1]], [[1, 1]], [[-1, 2]]):
        pairs = np.tile(np.array([[0, 1], [0, 2], [1, 2]], np.int32), (4, 1, 1))
        pairs[2, 1] = bad[0]
        ps, os_, ts = create(prob, opts, S.Tables(pairs=pairs))
        rc, msg = rc_of(ps, os_, ts)
        assert rc == A.SMM_ERR_INVALID_ARG and "pairs" in msg, bad


def test_product_does_not_reference_the_oracle():
    """only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may touch oracle/"""
    pkg = os.path.join(ROOT, "smm.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in txt.lower().replace("# oracle", ""), os.path.join(dirpath, f)
    out = subprocess.check_output(["ldd", A.LIB_PATH]).decode()
    assert "smm_oracle" not in out


def test_the_shipped_library_has_no_test_seams():
    """VERDICT r2 #6 / ADVICE: the switches that force a kernel or shrink a capacity live in libsmmhip_hooks.so only.  The shipped
    library reads three environment variables, all diagnostics (SMMHIP_TS, SMMHIP_DBG, and SMMHIP_VERBOSE: why a user objective's
    persistent kernel did not compile).  (Round 5: the library carries its device headers as text — hiprtc compiles them with a user's
    objective inside —, so the public header's own macros SMMHIP_H / SMMHIP_ABI_VERSION appear in the file too: not variables.)"""
    blob = open(A.LIB_PATH, "rb").read()
    names = sorted(set(m.decode() for m in re.findall(rb"SMMHIP_[A-Z0-9_]+", blob)) - {"SMMHIP_H", "SMMHIP_ABI_VERSION"})
    assert names == ["SMMHIP_DBG", "SMMHIP_TS", "SMMHIP_VERBOSE"], names
    lib = A.load()
    lib.smm_debug_has_test_hooks.restype = C.c_int
    assert lib.smm_debug_has_test_hooks() == 0
    hooks = A.load_hooks()
    assert hooks.smm_debug_has_test_hooks() == 1
    for n in declared_symbols():
        assert hasattr(hooks, n), "libsmmhip_hooks.so does not export %s" % n
    seams = sorted(set(m.decode() for m in re.findall(rb"SMMHIP_[A-Z0-9_]+", open(A.HOOKS_LIB_PATH, "rb").read())))
    assert "SMMHIP_INLINE_WALK" in seams and "SMMHIP_A2A_CAP" in seams
