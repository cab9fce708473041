"""Thin object wrapper over the C ABI of include/smmhip.h (libsmmhip.so).
There is deliberately no CPU code path in here."""
import ctypes as C

import numpy as np

from . import _abi as A


class Problem:
    """Flattened MProb (mprob.jl:29-53): what smm_problem_t carries."""

    def __init__(self, init, lb, ub, mom, w=None, ns=10000, objective_id=A.SMM_OBJ_NORM, obj_params=None):
        self.init = A.f64(init); self.lb = A.f64(lb); self.ub = A.f64(ub)
        self.mom = A.f64(mom)
        self.w = A.f64(np.full(len(self.mom), np.nan) if w is None else w)
        self.np = len(self.init); self.nm = len(self.mom); self.ns = int(ns)
        self.objective_id = int(objective_id)
        self.obj_params = None if obj_params is None else A.f64(obj_params)
        assert len(self.lb) == self.np and len(self.ub) == self.np and len(self.w) == self.nm

    def struct(self):
        p = A.smm_problem_t()
        p.np, p.nm, p.ns, p.objective_id = self.np, self.nm, self.ns, self.objective_id
        p.init, p.lb, p.ub, p.mom, p.w = map(A.dptr, (self.init, self.lb, self.ub, self.mom, self.w))
        p.obj_params = A.dptr(self.obj_params)
        p.n_obj_params = 0 if self.obj_params is None else len(self.obj_params)
        return p


class BGPOpts:
    """Flattened opts Dict of MAlgoBGP (AlgoBGP.jl:505-537). Per-chain vectors are GLOBAL
    (length N_global); the context owns chains [chain_offset, chain_offset+N)."""

    def __init__(self, N, maxiter, sigma, acc_tuner, min_improve, sigma_update_steps=10, sigma_adjust_by=0.01,
                 smpl_iters=1000, batch_size=None, exchange_from_iter=2, seed=12, chain_offset=0, N_global=None,
                 device=0, chol_L=None, dist_fun=0):
        self.N = int(N); self.maxiter = int(maxiter)
        self.N_global = int(N if N_global is None else N_global)
        self.sigma = A.f64(sigma, (self.N_global,)); self.acc_tuner = A.f64(acc_tuner, (self.N_global,))
        self.min_improve = A.f64(min_improve, (self.N_global,))
        self.sigma_update_steps = int(sigma_update_steps); self.sigma_adjust_by = float(sigma_adjust_by)
        self.smpl_iters = int(smpl_iters); self.batch_size = batch_size
        self.exchange_from_iter = int(exchange_from_iter); self.seed = int(seed)
        self.chain_offset = int(chain_offset); self.device = int(device)
        # general Gaussian proposals: a lower-triangular factor [np][np] (shared) or [N_global][np][np] (per chain), see smmhip.h
        self.chol_L = None if chol_L is None else A.f64(chol_L)
        self.dist_fun = int(dist_fun)   # smm_dist_fun_t: 0 `-` (AlgoBGP.jl:537), 1 |a - b|, 2 (a - b) / |a|

    def struct(self, np_):
        o = A.smm_bgp_opts_t()
        o.N, o.maxiter = self.N, self.maxiter
        o.sigma, o.acc_tuner, o.min_improve = map(A.dptr, (self.sigma, self.acc_tuner, self.min_improve))
        o.sigma_update_steps, o.smpl_iters = self.sigma_update_steps, self.smpl_iters
        o.sigma_adjust_by = self.sigma_adjust_by
        o.batch_size = np_ if self.batch_size is None else int(self.batch_size)
        o.exchange_from_iter = self.exchange_from_iter
        o.seed = self.seed
        o.chain_offset, o.N_global, o.device = self.chain_offset, self.N_global, self.device
        o.chol_L = A.dptr(self.chol_L)
        o.dist_fun = self.dist_fun
        o.chol_per_chain = 0 if self.chol_L is None or self.chol_L.ndim == 2 else 1
        if self.chol_L is not None:
            want = (np_, np_) if self.chol_L.ndim == 2 else (self.N_global, np_, np_)
            if self.chol_L.shape != want:
                raise ValueError("chol_L must be [np][np] or [N_global][np][np]")
        return o


class Tables:
    """Injected randomness (smm_tables_t). All optional."""

    def __init__(self, probs_acc=None, prop_normals=None, pairs=None, Z=None):
        self.probs_acc = None if probs_acc is None else A.f64(probs_acc)          # [T][N]
        self.prop_normals = None if prop_normals is None else A.f64(prop_normals)  # [T][K][np][N]
        self.pairs = None if pairs is None else np.ascontiguousarray(pairs, dtype=np.int32)  # [T][K][2]
        self.Z = None if Z is None else A.f64(Z)                                   # [nm][ns]

    def check(self, problem, opts):
        """the C ABI reads T x ... entries out of these host arrays (include/smmhip.h, smm_tables_t): shapes that do not cover
        opts.maxiter iterations of this shard would be a host over-read, so they are refused here"""
        T, N, npar = opts.maxiter, opts.N, problem.np
        if self.probs_acc is not None and self.probs_acc.shape != (T, N):
            raise ValueError("Tables.probs_acc must be [maxiter][N] = %s, got %s" % ((T, N), self.probs_acc.shape))
        if self.prop_normals is not None and (self.prop_normals.ndim != 4 or self.prop_normals.shape[0] != T or
                                              self.prop_normals.shape[1] < 1 or self.prop_normals.shape[2:] != (npar, N)):
            raise ValueError("Tables.prop_normals must be [maxiter][tries][np][N] = (%d, K, %d, %d), got %s"
                             % (T, npar, N, self.prop_normals.shape))
        if self.pairs is not None and (self.pairs.ndim != 3 or self.pairs.shape[0] != T or self.pairs.shape[2] != 2):
            raise ValueError("Tables.pairs must be [maxiter][n_pairs][2] with maxiter = %d, got %s" % (T, self.pairs.shape))
        if self.Z is not None and self.Z.shape != (problem.nm, problem.ns):
            raise ValueError("Tables.Z must be [nm][ns] = %s, got %s" % ((problem.nm, problem.ns), self.Z.shape))

    def covers_iterations(self):
        """iterations the per-iteration tables were made for (None: nothing injected per iteration)"""
        ts = [a.shape[0] for a in (self.probs_acc, self.prop_normals, self.pairs) if a is not None]
        return min(ts) if ts else None

    def struct(self):
        t = A.smm_tables_t()
        t.probs_acc = A.dptr(self.probs_acc)
        t.prop_normals = A.dptr(self.prop_normals)
        t.prop_tries = 0 if self.prop_normals is None else self.prop_normals.shape[1]
        t.pairs = None if self.pairs is None else self.pairs.ctypes.data_as(A.c_int32_p)
        t.n_pairs = 0 if self.pairs is None else self.pairs.shape[1]
        t.Z = A.dptr(self.Z)
        return t


class BGPContext:
    """One device context of libsmmhip.so = the chains of one MAlgoBGP shard."""

    _p = "smm_"   # symbol prefix of the C ABI

    def __init__(self, problem, opts, tables=None):
        self._lib = A.load()
        self._create(problem, opts, tables)

    def _create(self, problem, opts, tables):
        self.problem, self.opts = problem, opts
        self.tables = tables
        self._ctx = C.c_void_p()
        ps, os_ = problem.struct(), opts.struct(problem.np)
        if tables is not None:
            tables.check(problem, opts)
        ts = tables.struct() if tables is not None else None
        rc = self._fn("ctx_create")(C.byref(ps), C.byref(os_), C.byref(ts) if ts is not None else None,
                                    C.byref(self._ctx))
        if rc != 0:
            msg = self._fn("last_error")(None)
            raise A.SMMHipError(rc, msg.decode() if msg else "ctx_create failed")
        self.N, self.np, self.nm = opts.N, problem.np, problem.nm

    def _fn(self, name):
        return getattr(self._lib, self._p + name)

    def _check(self, rc):
        if rc != 0:
            msg = self._fn("last_error")(self._ctx)
            raise A.SMMHipError(rc, msg.decode() if msg else "")

    def close(self):
        if self._ctx:
            self._fn("ctx_destroy")(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- the path ---------------------------------------------------------------------
    def step(self, n_iters=1):
        self._check(self._fn("bgp_step")(self._ctx, int(n_iters)))

    def step_async(self, n_iters=1):
        self._check(self._fn("bgp_step_async")(self._ctx, int(n_iters)))

    def sync(self):
        self._check(self._fn("sync")(self._ctx))

    def local_step(self):
        self._check(self._fn("bgp_local_step")(self._ctx))

    def record_doubles(self):
        return self._fn("bgp_record_doubles")(self._ctx)

    def export_records_dev(self, ptr):
        self._check(self._fn("bgp_export_records_dev")(self._ctx, C.c_void_p(ptr)))

    def exchange_dev(self, ptr):
        self._check(self._fn("bgp_exchange_dev")(self._ctx, C.c_void_p(ptr)))

    # the values form of the exchange phase (include/smmhip.h): device pointers
    def a2a_capacity(self):
        return self._fn("bgp_a2a_capacity")(self._ctx)

    def export_values_dev(self, ptr):
        self._check(self._fn("bgp_export_values_dev")(self._ctx, C.c_void_p(ptr)))

    def a2a_pack_dev(self, vals_all_ptr, send_ptr):
        self._check(self._fn("bgp_a2a_pack_dev")(self._ctx, C.c_void_p(vals_all_ptr), C.c_void_p(send_ptr)))

    def a2a_apply_dev(self, recv_ptr):
        self._check(self._fn("bgp_a2a_apply_dev")(self._ctx, C.c_void_p(recv_ptr)))

    def sharded_step(self, prev_ptr, next_ptr):
        """fused sharded iteration (include/smmhip.h): prev/next are device pointers of [N_global][RW] buffers"""
        self._check(self._fn("bgp_sharded_step")(self._ctx, C.c_void_p(prev_ptr or 0), C.c_void_p(next_ptr)))

    def sharded_finish(self, ptr):
        self._check(self._fn("bgp_sharded_finish")(self._ctx, C.c_void_p(ptr or 0)))

    # the p2p form of the sharded iteration (include/smmhip.h): windows mapped by every rank, no collective
    def p2p_init(self):
        """allocate this rank's window; returns (ipc handle bytes, device pointer)"""
        h = C.create_string_buffer(A.SMM_P2P_HANDLE_BYTES)
        w = C.c_void_p()
        self._check(self._fn("bgp_p2p_init")(self._ctx, h, C.byref(w)))
        return bytes(h.raw), int(w.value)

    def p2p_attach(self, rank, handle=None, window=None):
        """rank's window: by IPC handle (another process) or device pointer (a context of this process)"""
        hb = C.create_string_buffer(handle, A.SMM_P2P_HANDLE_BYTES) if handle is not None else None
        self._check(self._fn("bgp_p2p_attach")(self._ctx, int(rank), hb, C.c_void_p(window) if window is not None else None))

    def p2p_step(self, n_iters=1):
        self._check(self._fn("bgp_p2p_step")(self._ctx, int(n_iters)))

    def p2p_finish(self):
        self._check(self._fn("bgp_p2p_finish")(self._ctx))

    def stream(self):
        return self._fn("stream")(self._ctx)

    def eval_batch(self, params):
        params = A.f64(params)
        assert params.shape[0] == self.np
        M = params.shape[1]
        value = np.empty(M); simM = np.empty((self.nm, M)); status = np.empty(M, np.int8)
        self._check(self._fn("eval_batch")(self._ctx, A.dptr(params), M, A.dptr(value), A.dptr(simM),
                                           status.ctypes.data_as(A.c_int8_p)))
        return value, simM, status

    def eval_batch_noseed(self, params, base_seed):
        """objfunc_norm with noseed=true (ObjExamples.jl:71-75): evaluation i draws its own shocks (base_seed + i)"""
        params = A.f64(params)
        assert params.shape[0] == self.np
        M = params.shape[1]
        value = np.empty(M); simM = np.empty((self.nm, M)); status = np.empty(M, np.int8)
        self._check(self._fn("eval_batch_noseed")(self._ctx, A.dptr(params), M, C.c_uint64(int(base_seed)), A.dptr(value),
                                                  A.dptr(simM), status.ctypes.data_as(A.c_int8_p)))
        return value, simM, status

    # --- read back --------------------------------------------------------------------
    def history(self, t0=0, t1=None):
        t1 = self.state().iter if t1 is None else t1
        hb = A.HistoryBuffers(t1 - t0, self.N, self.np, self.nm)
        hs = hb.struct()
        self._check(self._fn("get_history")(self._ctx, t0, t1, C.byref(hs)))
        return hb

    def state(self):
        sb = A.StateBuffers(self.N, self.np, self.nm)
        ss = sb.struct()
        self._check(self._fn("get_state")(self._ctx, C.byref(ss)))
        sb.iter = ss.iter
        return sb

    def set_state(self, sb, hb):
        ss, hs = sb.struct(), hb.struct()
        self._check(self._fn("set_state")(self._ctx, C.byref(ss), C.byref(hs)))

    def timing(self):
        t = A.smm_timing_t()
        self._check(self._fn("get_timing")(self._ctx, C.byref(t)))
        return t

    def set_profiling(self, on=True):
        """0/False off; 1/True event brackets; 2 per-kernel begin/end timestamps (see include/smmhip.h)"""
        self._check(self._fn("set_profiling")(self._ctx, int(on)))

    def set_persistent(self, on=True):
        """the persistent form of step() (one launch per look-ahead window; include/smmhip.h): on by default where a context qualifies"""
        self._check(self._fn("set_persistent")(self._ctx, int(bool(on))))

    def describe(self):
        """the forms this context was given at creation, as a dict (smm_describe: chain / walk / exchange / persistent / plan / window)"""
        buf = C.create_string_buffer(256)
        self._check(self._fn("describe")(self._ctx, buf, 256))
        return dict(kv.split("=", 1) for kv in buf.value.decode().split())

    def persistent_info(self):
        """(would the next step take the persistent form, launches of it so far, repairs so far)"""
        a, l, r = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        self._check(self._fn("get_persistent")(self._ctx, C.byref(a), C.byref(l), C.byref(r)))
        return bool(a.value), int(l.value), int(r.value)

    def Z(self):
        z = np.empty((self.nm, self.problem.ns))
        self._check(self._fn("get_Z")(self._ctx, A.dptr(z)))
        return z


def register_user_objective(source, n_sums=None, lanes=256):
    """Compile a user objective for the device and return its objective_id handle (include/smmhip.h).
    n_sums=None: `source` defines SMM_USER_OBJECTIVE(...), evaluated by one thread per chain.
    n_sums=k:    map-reduce form — `source` defines SMM_USER_PARTIAL(...) and SMM_USER_FINISH(...); `lanes` threads
                 evaluate one chain and their k partial sums are reduced in a fixed order.
    Raises with the compiler log if the source does not compile."""
    lib = A.load()
    oid = C.c_int32(0)
    if n_sums is None:
        rc = lib.smm_register_user_objective(source.encode(), C.byref(oid))
    else:
        rc = lib.smm_register_user_objective_lanes(source.encode(), int(n_sums), int(lanes), C.byref(oid))
    if rc != 0:
        raise RuntimeError("smm_register_user_objective failed (%d): %s" % (rc, lib.smm_last_error(None).decode()))
    return int(oid.value)


def hip_context(problem, opts, tables=None):
    """A BGPContext on libsmmhip.so. Raises if the library is missing."""
    return BGPContext(problem, opts, tables)
