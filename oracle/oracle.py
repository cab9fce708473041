"""Loader of the CPU oracle (oracle/smm_oracle.c). TEST INFRASTRUCTURE ONLY — imported by
tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke(), never by smm.jl_amd/."""
import ctypes as C
import os
import subprocess

import numpy as np

import smm_jl_amd as S
from smm_jl_amd import _abi as A

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsmm_oracle.so")

_ORC_ONLY = [
    ("orc_bgp_export_records", None, [C.c_void_p, A.c_double_p]),
    ("orc_bgp_exchange", C.c_int, [C.c_void_p, A.c_double_p]),
    ("orc_bgp_resolve_values", C.c_int, [C.c_void_p, A.c_double_p, A.c_int32_p, A.c_int32_p]),
    ("orc_set_mode", None, [C.c_void_p, C.c_int, C.c_int]),
    ("orc_gen_Z", None, [C.c_uint64, C.c_int, C.c_int, A.c_double_p]),
    ("orc_gen_pairs", None, [C.c_uint64, C.c_int32, C.c_int32, A.c_int32_p]),
    ("orc_philox4x32_10", None, [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("orc_max_threads", C.c_int, []),
    ("orc_gen_dense", None, [C.c_uint64, C.c_int, C.c_int, A.c_double_p]),
    ("orc_gen_dense2", None, [C.c_uint64, C.c_int, C.c_int, A.c_double_p]),
    ("orc_tanh", None, [A.c_double_p, A.c_double_p, C.c_int]),
    ("orc_math", None, [C.c_int, A.c_double_p, A.c_double_p, C.c_int]),
    ("orc_set_user_objective", None, [C.c_int, C.c_void_p]),
    ("orc_set_user_objective_lanes", None, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
]
_SHARED = ["smm_ctx_create", "smm_ctx_destroy", "smm_last_error", "smm_bgp_step", "smm_bgp_local_step",
           "smm_bgp_record_doubles", "smm_eval_batch", "smm_eval_batch_noseed", "smm_get_history", "smm_get_state", "smm_set_state", "smm_get_Z"]

_lib = None


def build(force=False):
    if force or not os.path.exists(LIB_PATH) or \
            os.path.getmtime(LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "smm_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libsmm_oracle.so"], stdout=subprocess.DEVNULL)


def load():
    global _lib
    if _lib is None:
        build()
        lib = C.CDLL(LIB_PATH)
        A.bind(lib, [s for s in A.SYMBOLS if s[0] in _SHARED], "smm_", "orc_")
        A.bind(lib, _ORC_ONLY)
        _lib = lib
    return _lib


class OracleContext(S.BGPContext):
    """the oracle behind the product's context interface: the same method bodies drive liboracle's `orc_` twins of the ABI
    entry points (the redirection lives here, in test infrastructure: the product's BGPContext only knows libsmmhip.so)"""
    _p = "orc_"

    def __init__(self, problem, opts, tables=None, threads=1, regen_z=False):
        self._lib = load()
        self._create(problem, opts, tables)
        self._lib.orc_set_mode(self._ctx, threads, int(regen_z))

    def set_mode(self, threads=1, regen_z=False):
        self._lib.orc_set_mode(self._ctx, threads, int(regen_z))

    def export_records(self):
        rec = np.empty((self.N, self.record_doubles()))
        self._lib.orc_bgp_export_records(self._ctx, A.dptr(rec))
        return rec

    def exchange(self, gathered):
        g = A.f64(gathered)
        self._check(self._lib.orc_bgp_exchange(self._ctx, A.dptr(g)))

    def resolve_values(self, vals_all):
        """(src, partner) of every chain from every chain's value: the exchange walk alone"""
        v = A.f64(vals_all)
        src = np.empty(v.shape[0], np.int32); partner = np.empty(v.shape[0], np.int32)
        self._check(self._lib.orc_bgp_resolve_values(self._ctx, A.dptr(v), src.ctypes.data_as(A.c_int32_p), partner.ctypes.data_as(A.c_int32_p)))
        return src, partner


def gen_Z(seed, nm, ns):
    Z = np.empty((nm, ns))
    load().orc_gen_Z(seed, nm, ns, A.dptr(Z))
    return Z


def gen_pairs(seed, t, Ng):
    K = Ng - 1 if Ng < 3 else Ng
    p = np.empty((max(K, 0), 2), np.int32)
    if K > 0:
        load().orc_gen_pairs(seed, t, Ng, p.ctypes.data_as(A.c_int32_p))
    return p


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr); k = (C.c_uint32 * 2)(*key); o = (C.c_uint32 * 4)()
    load().orc_philox4x32_10(c, k, o)
    return list(o)


def max_threads():
    return load().orc_max_threads()


def gen_dense(seed, np_, nm):
    out = np.empty(A.SMM_DENSE_D * np_ + nm * A.SMM_DENSE_D)
    load().orc_gen_dense(seed, np_, nm, A.dptr(out))
    return out


def gen_dense2(seed, np_, nm):
    out = np.empty(A.SMM_DENSE_D * np_ + A.SMM_DENSE_D * A.SMM_DENSE_D + nm * A.SMM_DENSE_D)
    load().orc_gen_dense2(seed, np_, nm, A.dptr(out))
    return out


def contract_math(what, x):
    """the contract's elementary functions (smm_oracle.c): what = "log" | "exp" | "sin2pi" | "cos2pi"""
    x = np.ascontiguousarray(x, np.float64)
    y = np.empty_like(x)
    load().orc_math({"log": 0, "exp": 1, "sin2pi": 2, "cos2pi": 3}[what], A.dptr(x), A.dptr(y), x.size)
    return y


def dense_tanh(x):
    """the dense objective's tanh as include/smmhip.h freezes it (smm_oracle.c: smm_tanh)"""
    x = np.ascontiguousarray(x, np.float64)
    y = np.empty_like(x)
    load().orc_tanh(A.dptr(x), A.dptr(y), x.size)
    return y


_user_libs = []


def register_user_objective(source, objective_id, workdir=None, n_sums=None, lanes=256):
    """Build the SAME objective source the device compiles (include/smmhip.h: SMM_USER_OBJECTIVE, or with n_sums the
    map-reduce pair SMM_USER_PARTIAL / SMM_USER_FINISH) for the host with gcc (-ffp-contract=off like the device
    build) and hook it into the oracle under the device's handle."""
    import tempfile
    d = workdir or tempfile.mkdtemp(prefix="smm_user_obj_")
    src = os.path.join(d, "user_objective_%d.c" % objective_id)
    so = os.path.join(d, "user_objective_%d.so" % objective_id)
    with open(src, "w") as f:
        f.write("#include <math.h>\n#include <stddef.h>\n#define SMM_USER_OBJECTIVE void smm_user_objective\n"
                "#define SMM_USER_PARTIAL void smm_user_partial\n#define SMM_USER_FINISH void smm_user_finish\n"
                "#define SMM_NSUMS %d\n" % (n_sums or 1) + source)
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-o", so, src, "-lm"])
    lib = C.CDLL(so)
    _user_libs.append(lib)
    if n_sums is None:
        load().orc_set_user_objective(int(objective_id), C.cast(lib.smm_user_objective, C.c_void_p))
    else:
        load().orc_set_user_objective_lanes(int(objective_id), C.cast(lib.smm_user_partial, C.c_void_p),
                                            C.cast(lib.smm_user_finish, C.c_void_p), int(n_sums), int(lanes))
    return so
