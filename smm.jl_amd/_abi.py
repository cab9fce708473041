"""ctypes mirror of include/smmhip.h and the loader of libsmmhip.so.

The library is the product: there is NO CPU fallback.  If the shared object is missing or
does not export every symbol of the header, importing the backend raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libsmmhip.so")
# the same sources with the test seams compiled in (-DSMM_TEST_HOOKS: forcing an exchange kernel, switching a fast path off, ...):
# loaded by tests/ and tools/ only, never by the product path
HOOKS_LIB_PATH = os.path.join(_HERE, "csrc", "libsmmhip_hooks.so")

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
c_uint8_p = C.POINTER(C.c_uint8)
c_int8_p = C.POINTER(C.c_int8)

# smm_status_t
SMM_OK = 0
SMM_ERR_INVALID_ARG = -1
SMM_ERR_NO_DEVICE = -2
SMM_ERR_NEGATIVE_OBJECTIVE = -3
SMM_ERR_NO_DRAW_IN_SUPPORT = -4
SMM_ERR_BAD_BATCH = -5
SMM_ERR_MAXITER = -6
SMM_ERR_HIP = -7
SMM_ERR_STATE = -8
SMM_ERR_EXCHANGE_CAPACITY = -9

# smm_objective_t
SMM_OBJ_NORM = 0
SMM_OBJ_BANANA = 1
SMM_OBJ_NORM_FAILBOX = 2
SMM_OBJ_DENSE = 3
SMM_OBJ_DENSE2 = 5   # spec v2 of the dense simulation: with the 256 x 256 stage (BASELINE config 5 as worded)
SMM_DIST_MINUS, SMM_DIST_ABSDIFF, SMM_DIST_RELDIFF = 0, 1, 2   # smm_dist_fun_t
SMM_OBJ_USER_BASE = 1000   # objective ids >= this are handles of smm_register_user_objective
SMM_DENSE_D = 256

SMM_REDUCE_LANES = 512
SMM_P2P_HANDLE_BYTES = 64


class smm_problem_t(C.Structure):
    _fields_ = [
        ("np", C.c_int32), ("nm", C.c_int32), ("ns", C.c_int32), ("objective_id", C.c_int32),
        ("init", c_double_p), ("lb", c_double_p), ("ub", c_double_p),
        ("mom", c_double_p), ("w", c_double_p), ("obj_params", c_double_p),
        ("n_obj_params", C.c_int32), ("reserved", C.c_int32),
    ]


class smm_bgp_opts_t(C.Structure):
    _fields_ = [
        ("N", C.c_int32), ("maxiter", C.c_int32),
        ("sigma", c_double_p), ("acc_tuner", c_double_p), ("min_improve", c_double_p),
        ("sigma_update_steps", C.c_int32), ("smpl_iters", C.c_int32),
        ("sigma_adjust_by", C.c_double),
        ("batch_size", C.c_int32), ("exchange_from_iter", C.c_int32),
        ("seed", C.c_uint64),
        ("chain_offset", C.c_int32), ("N_global", C.c_int32), ("device", C.c_int32), ("chol_per_chain", C.c_int32),
        ("chol_L", c_double_p),
        ("dist_fun", C.c_int32), ("reserved", C.c_int32),
    ]


class smm_tables_t(C.Structure):
    _fields_ = [
        ("probs_acc", c_double_p), ("prop_normals", c_double_p),
        ("prop_tries", C.c_int32), ("n_pairs", C.c_int32),
        ("pairs", c_int32_p), ("Z", c_double_p),
    ]


class smm_history_t(C.Structure):
    _fields_ = [
        ("value", c_double_p), ("prob", c_double_p), ("curr_val", c_double_p), ("best_val", c_double_p),
        ("params", c_double_p), ("sim_moments", c_double_p),
        ("best_id", c_int32_p), ("exchanged", c_int32_p),
        ("accepted", c_uint8_p), ("status", c_int8_p),
    ]


class smm_state_t(C.Structure):
    _fields_ = [
        ("iter", C.c_int32), ("reserved", C.c_int32),
        ("sigma", c_double_p), ("accept_rate", c_double_p),
        ("la_value", c_double_p), ("la_prob", c_double_p), ("la_params", c_double_p), ("la_sim_moments", c_double_p),
        ("la_status", c_int8_p), ("n_noex", c_int32_p), ("n_acc_noex", c_int32_p),
        ("best_val", c_double_p), ("best_id", c_int32_p),
    ]


class smm_timing_t(C.Structure):
    _fields_ = [
        ("step_ms", C.c_double), ("iter_kernel_ms", C.c_double), ("exch_kernel_ms", C.c_double),
        ("chain_evals", C.c_int64), ("iters", C.c_int32), ("reserved", C.c_int32),
        ("null_bracket_ms", C.c_double),
    ]


# every symbol declared in include/smmhip.h: (name, restype, argtypes)
SYMBOLS = [
    ("smm_abi_version", C.c_int, []),
    ("smm_register_user_objective", C.c_int, [C.c_char_p, C.POINTER(C.c_int32)]),
    ("smm_register_user_objective_lanes", C.c_int, [C.c_char_p, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]),
    ("smm_device_count", C.c_int, []),
    ("smm_ctx_create", C.c_int, [C.POINTER(smm_problem_t), C.POINTER(smm_bgp_opts_t), C.POINTER(smm_tables_t),
                                 C.POINTER(C.c_void_p)]),
    ("smm_ctx_destroy", None, [C.c_void_p]),
    ("smm_last_error", C.c_char_p, [C.c_void_p]),
    ("smm_bgp_step", C.c_int, [C.c_void_p, C.c_int32]),
    ("smm_bgp_step_async", C.c_int, [C.c_void_p, C.c_int32]),
    ("smm_sync", C.c_int, [C.c_void_p]),
    ("smm_bgp_local_step", C.c_int, [C.c_void_p]),
    ("smm_bgp_record_doubles", C.c_int, [C.c_void_p]),
    ("smm_bgp_export_records_dev", C.c_int, [C.c_void_p, C.c_void_p]),
    ("smm_bgp_exchange_dev", C.c_int, [C.c_void_p, C.c_void_p]),
    ("smm_bgp_a2a_capacity", C.c_int, [C.c_void_p]),
    ("smm_bgp_export_values_dev", C.c_int, [C.c_void_p, C.c_void_p]),
    ("smm_bgp_a2a_pack_dev", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ("smm_bgp_a2a_apply_dev", C.c_int, [C.c_void_p, C.c_void_p]),
    ("smm_bgp_sharded_step", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ("smm_bgp_sharded_finish", C.c_int, [C.c_void_p, C.c_void_p]),
    ("smm_bgp_p2p_init", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    ("smm_bgp_p2p_attach", C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    ("smm_bgp_p2p_step", C.c_int, [C.c_void_p, C.c_int32]),
    ("smm_bgp_p2p_finish", C.c_int, [C.c_void_p]),
    ("smm_stream", C.c_void_p, [C.c_void_p]),
    ("smm_eval_batch", C.c_int, [C.c_void_p, c_double_p, C.c_int32, c_double_p, c_double_p, c_int8_p]),
    ("smm_eval_batch_noseed", C.c_int, [C.c_void_p, c_double_p, C.c_int32, C.c_uint64, c_double_p, c_double_p, c_int8_p]),
    ("smm_get_history", C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(smm_history_t)]),
    ("smm_get_state", C.c_int, [C.c_void_p, C.POINTER(smm_state_t)]),
    ("smm_set_state", C.c_int, [C.c_void_p, C.POINTER(smm_state_t), C.POINTER(smm_history_t)]),
    ("smm_get_timing", C.c_int, [C.c_void_p, C.POINTER(smm_timing_t)]),
    ("smm_get_Z", C.c_int, [C.c_void_p, c_double_p]),
    ("smm_set_profiling", C.c_int, [C.c_void_p, C.c_int32]),
    ("smm_set_persistent", C.c_int, [C.c_void_p, C.c_int32]),
    ("smm_get_persistent", C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("smm_describe", C.c_int, [C.c_void_p, C.c_char_p, C.c_int32]),
]


class SMMHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("smmhip error %d: %s" % (code, msg))
        self.code = code


def bind(lib, symbols=SYMBOLS, prefix_from="smm_", prefix_to=None):
    """attach restype/argtypes; raises AttributeError if a symbol is missing"""
    for name, res, args in symbols:
        n = name if prefix_to is None else prefix_to + name[len(prefix_from):]
        fn = getattr(lib, n)
        fn.restype = res
        fn.argtypes = args
    return lib


_lib = None


def load():
    """dlopen libsmmhip.so (built by __graft_entry__.build() / csrc/Makefile). Fails loudly."""
    global _lib
    if _lib is None:
        path = LIB_PATH
        if not os.path.exists(path):
            raise ImportError(
                "libsmmhip.so not found at %s — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback." % path)
        _lib = bind(C.CDLL(path, mode=C.RTLD_GLOBAL))
        if _lib.smm_abi_version() != 3:
            raise ImportError("libsmmhip.so ABI version mismatch")
    return _lib


_hooks_lib = None


def load_hooks():
    """the test build (libsmmhip_hooks.so); tests/ and tools/ make it current with use_test_hooks()"""
    global _hooks_lib
    if _hooks_lib is None:
        if not os.path.exists(HOOKS_LIB_PATH):
            raise ImportError("libsmmhip_hooks.so not found at %s — `make -C smm.jl_amd/csrc all`" % HOOKS_LIB_PATH)
        lib = bind(C.CDLL(HOOKS_LIB_PATH, mode=C.RTLD_GLOBAL))   # (GLOBAL like the shipped one: one HIP runtime per process; both are linked -Bsymbolic)
        lib.smm_debug_has_test_hooks.restype = C.c_int
        if lib.smm_abi_version() != 3 or lib.smm_debug_has_test_hooks() != 1:
            raise ImportError("libsmmhip_hooks.so: wrong ABI version or built without -DSMM_TEST_HOOKS")
        _hooks_lib = lib
    return _hooks_lib


def use_test_hooks(on=True):
    """contexts created from now on come from the test build (on) or the shipped library (off); returns the library"""
    global _lib
    _lib = load_hooks() if on else None
    return load()


def dptr(a):
    return None if a is None else a.ctypes.data_as(c_double_p)


def f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        assert a.shape == tuple(shape), (a.shape, shape)
    return a


class HistoryBuffers:
    """numpy-backed smm_history_t for nt iterations of N chains"""

    def __init__(self, nt, N, np_, nm):
        self.value = np.empty((nt, N)); self.prob = np.empty((nt, N))
        self.curr_val = np.empty((nt, N)); self.best_val = np.empty((nt, N))
        self.params = np.empty((nt, np_, N)); self.sim_moments = np.empty((nt, nm, N))
        self.best_id = np.empty((nt, N), np.int32); self.exchanged = np.empty((nt, N), np.int32)
        self.accepted = np.empty((nt, N), np.uint8); self.status = np.empty((nt, N), np.int8)

    FIELDS = ("value", "prob", "curr_val", "best_val", "params", "sim_moments", "best_id", "exchanged", "accepted",
              "status")

    def struct(self):
        h = smm_history_t()
        for f, t in smm_history_t._fields_:
            setattr(h, f, getattr(self, f).ctypes.data_as(t))
        return h


class StateBuffers:
    def __init__(self, N, np_, nm):
        self.iter = 0
        self.sigma = np.empty(N); self.accept_rate = np.empty(N)
        self.la_value = np.empty(N); self.la_prob = np.empty(N)
        self.la_params = np.empty((np_, N)); self.la_sim_moments = np.empty((nm, N))
        self.la_status = np.empty(N, np.int8)
        self.n_noex = np.empty(N, np.int32); self.n_acc_noex = np.empty(N, np.int32)
        self.best_val = np.empty(N); self.best_id = np.empty(N, np.int32)

    FIELDS = ("sigma", "accept_rate", "la_value", "la_prob", "la_params", "la_sim_moments", "la_status", "n_noex",
              "n_acc_noex", "best_val", "best_id")

    def struct(self):
        s = smm_state_t()
        s.iter = self.iter
        for f, t in smm_state_t._fields_:
            if f in ("iter", "reserved"):
                continue
            setattr(s, f, getattr(self, f).ctypes.data_as(t))
        return s
