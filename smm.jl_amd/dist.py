"""Sharding of the BGP chains over ranks (one process per GPU).

Chains are independent inside an iteration (next_eval, AlgoBGP.jl:272-294); the only coupling
is exchangeMoves! (AlgoBGP.jl:647-716), which needs every chain's last accepted record.  Each
rank owns a contiguous block of chains; per iteration there is ONE collective — an all-gather of
the fixed-size last-accepted records over RCCL/xGMI — after which every rank resolves the
identical pair list redundantly and applies the swaps that touch its own chains.

For long records (50 parameters + 50 moments are 816 bytes per chain) there is the
VALUES form (`protocol="values"`): an all-gather of 8 bytes per chain, the replicated resolution, and an
all-to-all of fixed-size blocks that carries each swapped record to the owner of the chain that continues
from it — two collectives, about a quarter of a record per chain on the wire instead of a whole one.

The P2P form (`protocol="p2p"`) has NO collective in the iteration at all: every rank owns a window of device memory that
all other ranks map (HIP IPC), a chain's accept step stores its record into every rank's window over the point-to-point
xGMI links and counts itself in, the next iteration's kernel waits on its own window's counters (include/smmhip.h,
smm.jl_amd/csrc/smm_p2p.hpp).  torch.distributed is used once, to hand the IPC handles round.
"""
import torch
import torch.distributed as dist


class HipShardEngine:
    """adapter of a libsmmhip BGPContext to the ShardedBGP protocol (device tensors)"""

    def __init__(self, ctx, device):
        self.ctx = ctx
        self.device = torch.device(device)
        self.N = ctx.N
        self.R = ctx.record_doubles()
        # all torch work (the RCCL all-gather) is ordered on the library's own HIP stream
        self.stream = torch.cuda.ExternalStream(ctx.stream(), device=self.device)

    def new_tensor(self, shape):
        return torch.empty(shape, dtype=torch.float64, device=self.device)

    def local_step(self):
        self.ctx.local_step()

    def export_records(self, out):
        self.ctx.export_records_dev(out.data_ptr())

    def exchange(self, gathered):
        self.ctx.exchange_dev(gathered.data_ptr())

    # values form (include/smmhip.h)
    def a2a_capacity(self):
        return self.ctx.a2a_capacity()

    def export_values(self, out):
        self.ctx.export_values_dev(out.data_ptr())

    def a2a_pack(self, vals_all, send):
        self.ctx.a2a_pack_dev(vals_all.data_ptr(), send.data_ptr())

    def a2a_apply(self, recv):
        self.ctx.a2a_apply_dev(recv.data_ptr())

    # fused form (two enqueues per iteration, see include/smmhip.h): used by ShardedBGP when the engine offers it
    def fused_step(self, prev, nxt):
        self.ctx.sharded_step(prev.data_ptr() if prev is not None else 0, nxt.data_ptr())

    def fused_finish(self, gathered):
        self.ctx.sharded_finish(gathered.data_ptr() if gathered is not None else 0)

    # p2p form
    def p2p_init(self):
        return self.ctx.p2p_init()

    def p2p_attach(self, rank, handle=None, window=None):
        self.ctx.p2p_attach(rank, handle=handle, window=window)

    def p2p_step(self, n):
        self.ctx.p2p_step(n)

    def p2p_finish(self):
        self.ctx.p2p_finish()

    def sync(self):
        self.ctx.sync()

    def stream_ctx(self):
        return torch.cuda.stream(self.stream)


class ShardedBGP:
    """computeNextIteration! (AlgoBGP.jl:589-640) over world_size shards."""

    def __init__(self, engine, group=None, protocol="records"):
        """protocol: "records" (all-gather of the last-accepted records) or "values" (all-gather of the values + all-to-all of
        the swapped records: for long records)"""
        if protocol not in ("records", "values", "p2p"):
            raise ValueError("protocol: 'records', 'values' or 'p2p'")
        self.e = engine
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.protocol = protocol
        if protocol == "p2p":
            # hand the window handles round once; afterwards the iteration needs no collective
            handle, _ = engine.p2p_init()
            if self.world > 1:
                handles = [None] * self.world
                dist.all_gather_object(handles, handle, group=group)
                err = None
                try:
                    for r, h in enumerate(handles):
                        if r != self.rank:
                            engine.p2p_attach(r, handle=h)
                except Exception as e:   # noqa: BLE001 -- reported below, on EVERY rank
                    err = e
                # every window is mapped everywhere before anybody stores into one — and if one rank could not map one, all ranks
                # raise together (a rank that left alone would leave the others at a collective it never joins)
                be = str(dist.get_backend(group))
                flag = torch.tensor([0 if err else 1], dtype=torch.int32, device=engine.device if "nccl" in be else "cpu")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
                if not bool(flag.item()):
                    raise RuntimeError("p2p: a rank could not map a peer's window%s" % (": %s" % err if err else " (another rank)"))
            self.fused = False
            return
        if protocol == "values":
            cap = engine.a2a_capacity()
            if cap <= 0:
                raise ValueError("the values form needs equal shards")
            self.vloc = engine.new_tensor((engine.N,))
            self.vall = engine.new_tensor((self.world * engine.N,))
            self.send = engine.new_tensor((self.world, cap, engine.R))
            self.recv = engine.new_tensor((self.world, cap, engine.R))
            self.fused = False
            return
        self.local = engine.new_tensor((engine.N, engine.R))
        self.gathered = engine.new_tensor((self.world, engine.N, engine.R))  # == [N_global][R] in global chain order
        self.fused = hasattr(engine, "fused_step")
        self.inplace_ok = self._probe_inplace() if (self.fused and self.world > 1) else False
        if self.fused:   # two gather buffers alternate: iteration t reads donors from one, writes its records to the other
            self.gbuf = [self.gathered, engine.new_tensor((self.world, engine.N, engine.R))]
            self.gcur = None   # index of the buffer holding the gathered records of the last iteration

    def step(self, n_iters=1):
        e = self.e
        if self.protocol == "p2p":
            e.p2p_step(n_iters)
            return
        with e.stream_ctx():
            if self.protocol == "values":
                for _ in range(n_iters):
                    e.local_step()
                    e.export_values(self.vloc)
                    if self.world == 1:
                        self.vall.copy_(self.vloc)
                    else:
                        dist.all_gather_into_tensor(self.vall, self.vloc, group=self.group)
                    e.a2a_pack(self.vall, self.send)
                    if self.world == 1:
                        self.recv.copy_(self.send)
                    else:
                        dist.all_to_all_single(self.recv.view(-1), self.send.view(-1), group=self.group)
                    e.a2a_apply(self.recv)
                return
            if self.fused:
                for _ in range(n_iters):
                    nxt = 0 if self.gcur is None else self.gcur ^ 1
                    e.fused_step(self.gbuf[self.gcur] if self.gcur is not None else None, self.gbuf[nxt])
                    if self.world > 1:
                        self._all_gather(self.gbuf[nxt])
                    self.gcur = nxt
                return
            for _ in range(n_iters):
                e.local_step()
                if self.world == 1:
                    e.export_records(self.gathered[0])
                else:
                    e.export_records(self.local)
                    dist.all_gather_into_tensor(self.gathered.view(-1), self.local.view(-1), group=self.group)
                e.exchange(self.gathered)

    def _probe_inplace(self):
        """ONE probe at construction: does the backend accept the aliased form (ncclAllGather with sendbuff == recvbuff +
        rank * count; RCCL does, gloo does not)?  The answer is a property of the backend, so it is settled here, on a
        scratch tensor, and never revisited: a failing collective during the run is an error, not a mode switch."""
        t = self.e.new_tensor((self.world, 2))
        try:
            with self.e.stream_ctx():
                dist.all_gather_into_tensor(t.view(-1), t[self.rank].view(-1), group=self.group)
            ok = True
        except (RuntimeError, ValueError):
            ok = False
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=t.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)   # all ranks take the same path
        return bool(flag.item())

    def _all_gather(self, buf):
        """in place where the backend allows it: this rank's slice is already where the collective wants it; otherwise
        through a staging copy of the slice.  Errors propagate."""
        if self.inplace_ok:
            dist.all_gather_into_tensor(buf.view(-1), buf[self.rank].view(-1), group=self.group)
        else:
            self.local.copy_(buf[self.rank])
            dist.all_gather_into_tensor(buf.view(-1), self.local.view(-1), group=self.group)

    def sync(self):
        """settle the last iteration into the context (history/state readable afterwards) and wait for the device"""
        if self.protocol == "p2p":
            self.e.p2p_finish()
            self.e.sync()
            # nobody publishes again (the next step's first publication rewrites the windows' parity of the last iteration with a new
            # epoch) before EVERY rank has read the last iteration out of its window: a rank still in its finish would otherwise find
            # its words replaced — a time-out in the tagged forms, silently other records in the generic one (include/smmhip.h)
            if self.world > 1:
                dist.barrier(group=self.group)
            return
        if self.fused and self.gcur is not None:
            with self.e.stream_ctx():
                self.e.fused_finish(self.gbuf[self.gcur])
            self.gcur = None
        self.e.sync()
