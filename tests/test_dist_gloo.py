"""The N > 1 path on CPU: world_size-2 (and 3) `gloo` process groups drive smm_jl_amd.dist.ShardedBGP
with the oracle standing in for the device engine.  What is exercised is the host logic that runs
unchanged on the GPUs: the sharding by chain blocks, the all-gather of last-accepted records, the
replicated exchange resolution, and that the sharded result equals the single-process one."""
import contextlib
import os
import socket
import time
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleShardEngine:
    """ShardedBGP protocol on CPU tensors, backed by oracle/smm_oracle.c (test infrastructure)"""

    def __init__(self, ctx):
        self.ctx = ctx
        self.N = ctx.N
        self.R = ctx.record_doubles()

    def new_tensor(self, shape):
        return torch.zeros(shape, dtype=torch.float64)

    def local_step(self):
        self.ctx.local_step()

    def export_records(self, out):
        out.copy_(torch.from_numpy(self.ctx.export_records()))

    def exchange(self, gathered):
        self.ctx.exchange(gathered.reshape(-1, self.R).numpy())

    def sync(self):
        pass

    def stream_ctx(self):
        return contextlib.nullcontext()


class OracleFusedEngine(OracleShardEngine):
    """the two-enqueue protocol of ShardedBGP (fused_step / fused_finish: alternating gather buffers, in-place
    all-gather of this rank's slice) on top of the oracle's three phases"""

    def __init__(self, ctx, rank):
        super().__init__(ctx)
        self.rank = rank
        self.open = False

    def fused_step(self, prev, nxt):
        if self.open:
            self.ctx.exchange(prev.reshape(-1, self.R).numpy())
        self.ctx.local_step()
        nxt[self.rank].copy_(torch.from_numpy(self.ctx.export_records()))
        self.open = True

    def fused_finish(self, gathered):
        if self.open:
            self.ctx.exchange(gathered.reshape(-1, self.R).numpy())
        self.open = False


class OracleValuesEngine(OracleShardEngine):
    """the values form of the exchange phase (include/smmhip.h: export_values / a2a_pack / a2a_apply) on top of the oracle:
    the walk over the gathered values says which records travel; the blocks are laid out as the library's kernels lay
    them out (block b of the send buffer: the records the chains of rank b continue from, in the order of those chains)"""

    def __init__(self, ctx, rank, world, cap=None):
        super().__init__(ctx)
        self.rank, self.world = rank, world
        n = self.N
        self.cap = cap if cap is not None else min(n, 2 * ((n + world - 1) // world) + 64)

    def a2a_capacity(self):
        return self.cap

    def export_values(self, out):
        out.copy_(torch.from_numpy(self.ctx.export_records()[:, 0].copy()))

    def a2a_pack(self, vals_all, send):
        n, me = self.N, self.rank
        self.vals_all = vals_all.numpy().copy()
        self.src, self.partner = self.ctx.resolve_values(self.vals_all)
        rec = self.ctx.export_records()
        g = np.arange(self.world * n)
        for b in range(self.world):
            sel = (g // n == b) & (self.partner != 0) & (self.src // n == me)
            rows = self.src[sel] - me * n
            assert len(rows) <= self.cap, "block overflow"
            send[b, :len(rows)] = torch.from_numpy(rec[rows])

    def a2a_apply(self, recv):
        n, me = self.N, self.rank
        full = np.zeros((self.world * n, self.R))
        full[:, 0] = self.vals_all                      # the walk is repeated from the values; only donor rows are read beyond them
        mine = np.arange(me * n, (me + 1) * n)
        for a in range(self.world):
            sel = (self.partner[mine] != 0) & (self.src[mine] // n == a)
            full[self.src[mine][sel]] = recv[a, :int(sel.sum())].numpy()
        self.ctx.exchange(full)


class OracleP2PEngine:
    """the p2p form of ShardedBGP on CPU: the oracle evaluates the local chains, the "windows" are POSIX shared-memory segments that
    every rank maps by name (the role of the HIP IPC handle) and writes its records into, tagged with the iteration; a rank reads its
    OWN window once every row carries the tag it waits for — the protocol of smm.jl_amd/csrc/smm_p2p.hpp without a GPU"""

    def __init__(self, ctx, rank, world):
        from multiprocessing import shared_memory
        self._shm = shared_memory
        self.ctx, self.rank, self.world = ctx, rank, world
        self.N, self.R = ctx.N, ctx.record_doubles()
        self.device = torch.device("cpu")
        self.mine = None
        self.peers = {}
        self.iter = 0

    def _view(self, seg):
        return np.ndarray((2, self.world * self.N, self.R + 1), dtype=np.float64, buffer=seg.buf)   # [parity][chain][record | tag]

    def p2p_init(self):
        nbytes = 2 * self.world * self.N * (self.R + 1) * 8
        self.mine = self._shm.SharedMemory(create=True, size=nbytes)
        self._view(self.mine)[:] = 0.0
        self.peers[self.rank] = self.mine
        return self.mine.name.encode(), 0

    def p2p_attach(self, rank, handle=None, window=None):
        if os.environ.get("SMM_TEST_P2P_FAIL_RANK") == str(self.rank):
            raise RuntimeError("cannot map the window of rank %d (injected)" % rank)
        self.peers[rank] = self._shm.SharedMemory(name=handle.decode())

    def p2p_step(self, n):
        n_loc, me = self.N, self.rank
        for _ in range(n):
            self.ctx.local_step()
            self.iter += 1
            rec = self.ctx.export_records()
            b, tag = self.iter & 1, float(self.iter)
            for r in range(self.world):                       # the accept step's stores into every rank's window
                w = self._view(self.peers[r])
                w[b, me * n_loc:(me + 1) * n_loc, :self.R] = rec
                w[b, me * n_loc:(me + 1) * n_loc, self.R] = tag     # (the tag last: a row with the right tag is complete)
            own = self._view(self.mine)
            t0 = time.time()
            while not (own[b, :, self.R] == tag).all():        # every rank's rows of this iteration are in MY window
                time.sleep(0.0005)
                assert time.time() - t0 < 60, "a peer never stored its records"
            self.ctx.exchange(np.ascontiguousarray(own[b, :, :self.R]))

    def p2p_finish(self):
        pass

    def sync(self):
        pass

    def close(self):
        for r, seg in self.peers.items():
            seg.close()
        self.mine.unlink()


def _worker(rank, world, port, N, T, q, fused=False):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import common as cm
    from oracle import oracle as O
    from smm_jl_amd.dist import ShardedBGP
    from smm_jl_amd import _abi as A
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = N // world
        prob, opts = cm.serial_normal(N=N, T=T, ns=100, N_local=n, chain_offset=rank * n)
        octx = O.OracleContext(prob, opts)
        if fused == "p2p":
            sh = ShardedBGP(OracleP2PEngine(octx, rank, world), protocol="p2p")
        elif fused == "values":
            sh = ShardedBGP(OracleValuesEngine(octx, rank, world), protocol="values")
        else:
            sh = ShardedBGP(OracleFusedEngine(octx, rank) if fused else OracleShardEngine(octx))
            assert sh.fused == fused
        assert sh.world == world and sh.rank == rank
        sh.step(T // 2); sh.step(T - T // 2)
        sh.sync()
        hs = sh.e.ctx.history()
        # single-process reference of the whole population
        prob1, opts1 = cm.serial_normal(N=N, T=T, ns=100)
        one = O.OracleContext(prob1, opts1); one.step(T)
        h1 = one.history()
        ok = all(np.array_equal(getattr(hs, f), getattr(h1, f)[..., rank * n:(rank + 1) * n], equal_nan=True)
                 for f in A.HistoryBuffers.FIELDS)
        nx = int((hs.exchanged != 0).sum())
        remote = int(((hs.exchanged != 0) & ((hs.exchanged - 1) // n != rank)).sum())  # partners on other ranks
        q.put((rank, ok, nx, remote))
        if fused == "p2p":
            dist.barrier()      # nobody unlinks a window a peer may still read
            sh.e.close()
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.parametrize("world,fused", [(2, False), (3, False), (2, True), (3, True), (2, "values"), (3, "values"), (2, "p2p"), (3, "p2p")])
def test_sharded_gloo_equals_single_process(world, fused):
    N, T = 12 * world, 30
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, T, q, fused)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(world))
    assert all(r[1] for r in res), res
    assert sum(r[2] for r in res) > 0 and sum(r[3] for r in res) > 0  # exchanges happened, some across ranks


def _worker_attach_fails(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import common as cm
    from oracle import oracle as O
    from smm_jl_amd.dist import ShardedBGP
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["SMM_TEST_P2P_FAIL_RANK"] = "1"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        prob, opts = cm.serial_normal(N=8 * world, T=4, ns=50, N_local=8, chain_offset=rank * 8)
        e = OracleP2PEngine(O.OracleContext(prob, opts), rank, world)
        try:
            ShardedBGP(e, protocol="p2p")
            q.put((rank, "no error"))
        except RuntimeError as err:
            q.put((rank, str(err)))
        dist.barrier()          # every rank is still in step with the others: nobody was left at a collective
        e.mine.close(); e.mine.unlink()
    finally:
        dist.destroy_process_group()


def test_p2p_attach_failure_raises_on_every_rank():
    # one rank cannot map a peer's window: ALL ranks raise (none is left waiting at a collective the failing one never joins)
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_attach_fails, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert "could not map" in res[0] and "another rank" in res[0]
    assert "could not map" in res[1] and "injected" in res[1]
