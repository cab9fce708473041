#!/usr/bin/env python
"""bench.py — chain-evals/sec of the BGP hot path on MI355X.

Workload (BASELINE.json configs[1], "C2"): serialNormal objective (2 params / 2 moments,
ns = 10000 simulated draws per moment), 4096 BGP chains per GPU, FP64.  One "step" = one job
of 200 iterations over all chains (= the metric's "4096 chains x 200 iters"); successive steps
continue the same chains.  value = N_global * 200 * steps / wall, inputs resident in HBM.
N > 1: one process per GPU (torch.distributed / RCCL), chains sharded, weak scaling
(4096 chains per GPU), one all-gather of last-accepted records per iteration.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CHAINS_PER_GPU = 4096
ITERS_PER_STEP = 200
NS = 10000
# algorithmic work per chain evaluation: 2*nm*ns FP64 adds + ~100 for
# proposal/objective/accept; 128 B of HBM traffic (state read 40 B + history record 88 B)
FLOP_PER_EVAL = 2 * 2 * NS + 100
BYTES_PER_EVAL = 128
PEAK_FP64_ADD_TFLOPS = 256 * 4 * 16 * 2.4e9 / 1e12  # 39.3: 256 CU x 4 SIMD x 16 f64 lanes/clk x 2.4 GHz (adds cannot be FMA'd)
PEAK_HBM_GBS = 8000.0


def kernel_source_hash():
    """sha256 (16 hex digits) over the device sources: ties a committed PMC summary to the build it was taken from"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "smm.jl_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic():
    """HBM bytes per k_chain_iter_norm launch from the newest committed rocprofv3 PMC passes (profiles/): FETCH_SIZE and
    WRITE_SIZE are KB per launch; gfx950's FETCH_SIZE tallies wide coalesced reads at half their size
    (MI355X_MICROARCH.md, HBM) so it is doubled.  Counters cannot be read from inside the timed process, so the number
    comes from the profile file; `stale` says whether that profile was taken from other device sources than this build
    (the summary carries the source hash).  (None, ...) when no profile is committed."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.txt")))
    if not files:
        return None, None, None
    fetch = write = None
    src_hash = None
    for line in open(files[-1]):
        m = re.match(r"#\s*kernel_source_sha16=([0-9a-f]+)", line)
        if m:
            src_hash = m.group(1)
        if "k_chain_iter_norm<2, true>" in line or ("k_chain_iter<1," in line and fetch is None):
            m = re.search(r"(FETCH_SIZE|WRITE_SIZE)\s+launches=\s*\d+\s+mean_per_launch=\s*([0-9.]+)", line)
            if m and m.group(1) == "FETCH_SIZE":
                fetch = float(m.group(2))
            elif m:
                write = float(m.group(2))
    if fetch is None or write is None:
        return None, None, None
    return (2.0 * fetch + write) * 1024.0, os.path.relpath(files[-1], ROOT), (src_hash != kernel_source_hash())


def host_cores():
    """cores this process may really use: affinity mask and cgroup quota, not the machine total"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(p))))
    except Exception:
        pass
    return n


def cpu_baseline(threads):
    """the oracle (C port of the reference path, oracle/smm_oracle.c) on the host cores the cgroup grants.
    value: the FULL workload of the metric (4096 chains x 200 iterations) with the shock matrix drawn once and cached -- an
    upper bound for any CPU run of the reference path, which redraws its 2 x 10000 normals inside every evaluation
    (ObjExamples.jl:74-79).  regen_value: a bounded sample with those draws regenerated per evaluation by the port's
    generator (Philox + Box-Muller, both outputs used): a lower bound (Julia's ziggurat randn is several times cheaper)."""
    import common as cm
    from oracle import oracle as O
    n, t = CHAINS_PER_GPU, ITERS_PER_STEP
    reps_max = 400
    prob, opts = cm.serial_normal(N=n, T=t * reps_max)
    o = O.OracleContext(prob, opts, threads=threads, regen_z=False)
    o.step(t)                                   # warm-up job
    reps, t0 = 0, time.perf_counter()
    while reps < reps_max - 1 and (reps == 0 or time.perf_counter() - t0 < 10.0):   # whole 4096 x 200 jobs for ~10 s
        o.step(t); reps += 1
    dt = (time.perf_counter() - t0) / reps
    n2, t2 = 512, 20
    prob2, opts2 = cm.serial_normal(N=n2, T=t2 * 200)
    o2 = O.OracleContext(prob2, opts2, threads=threads, regen_z=True)
    reps2, t0 = 0, time.perf_counter()
    while reps2 < 199 and (reps2 == 0 or time.perf_counter() - t0 < 8.0):
        o2.step(t2); reps2 += 1
    dt2 = (time.perf_counter() - t0) / reps2
    return {"value": n * t / dt, "unit": "chain-evals/s", "cores": threads, "kind": "port",
            "sample": "the whole workload: %d chains x %d iterations, OpenMP over chains (%d threads), AVX2, shock matrix "
                      "cached (no per-evaluation RNG: an upper bound for the reference's CPU path); %d such jobs, %.2f s each" % (n, t, threads, reps, dt),
            "regen_value": n2 * t2 / dt2,
            "regen_sample": "%d chains x %d iterations with the 2x10000 normals of every evaluation regenerated "
                            "(ObjExamples.jl:74-79) by the port's Philox/Box-Muller: a lower bound; %d repetitions, %.2f s each" % (n2, t2, reps2, dt2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-unfused", action="store_true", help="skip the reference timing of the unfused kernels (profiling runs)")
    ap.add_argument("--chains", type=int, default=CHAINS_PER_GPU)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                     % (args.gpus, args.gpus))
        args.gpus = world

    import numpy as np
    import torch
    import torch.distributed as dist
    import smm_jl_amd as S
    import common as cm

    torch.cuda.set_device(local_rank)
    if world > 1 or os.environ.get("SMM_BENCH_FORCE_SHARDED") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    n_loc = args.chains
    n_glob = n_loc * world
    K, W = args.steps, args.warmup
    T = ITERS_PER_STEP * (K + W + 2)   # + the two profiled steps
    prob, opts = cm.serial_normal(N=n_glob, T=T, N_local=n_loc, chain_offset=rank * n_loc, device=local_rank)
    ctx = S.hip_context(prob, opts)

    def barrier():
        if world > 1:
            dist.barrier()

    force_sharded = os.environ.get("SMM_BENCH_FORCE_SHARDED") == "1"  # exercise the N>1 code path on one GPU
    if world == 1 and not force_sharded:
        def run_step():
            ctx.step_async(ITERS_PER_STEP)
        sync = ctx.sync
    else:
        from smm_jl_amd.dist import HipShardEngine, ShardedBGP
        # SMM_BENCH_PROTOCOL=values: the two-collective form for long records (all-gather of values + all-to-all of the swapped
        # records); the default is the one all-gather of records per iteration
        sh = ShardedBGP(HipShardEngine(ctx, torch.device("cuda", local_rank)), protocol=os.environ.get("SMM_BENCH_PROTOCOL", "records"))

        def run_step():
            sh.step(ITERS_PER_STEP)
        sync = sh.sync

    for _ in range(W):
        run_step()
    sync(); torch.cuda.synchronize(); barrier()
    t0 = time.perf_counter()
    for _ in range(K):
        run_step()
    sync(); torch.cuda.synchronize(); barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    evals = n_glob * ITERS_PER_STEP * K
    value = evals / dt

    # roofline of the dominant kernel (k_chain_iter), HIP events on the library's stream
    roof = None
    if world == 1 and not force_sharded:
        # (a) the kernels' own start/stop events (hipExtLaunchKernelGGL on the library's stream): dispatch begin to
        #     end as the command processor stamps it -- the same quantity rocprofv3 --kernel-trace reports
        ctx.set_profiling(2)
        ctx.step(ITERS_PER_STEP)
        tm = ctx.timing()
        k_us = tm.iter_kernel_ms * 1e3 / ITERS_PER_STEP
        x_us = tm.exch_kernel_ms * 1e3 / ITERS_PER_STEP   # the chains are past iteration 1: every iteration exchanges
        # (b) event brackets around the kernels minus the measured empty-bracket overhead: the kernels' net cost on
        #     the stream (excludes the part of dispatch/drain that overlaps with the neighbours)
        ctx.set_profiling(1)
        ctx.step(ITERS_PER_STEP)
        tb = ctx.timing()
        ctx.set_profiling(0)
        null_us = tb.null_bracket_ms * 1e3 / ITERS_PER_STEP
        k_net_us = tb.iter_kernel_ms * 1e3 / ITERS_PER_STEP - null_us
        x_net_us = tb.exch_kernel_ms * 1e3 / ITERS_PER_STEP - null_us
        # the same chain kernel without the exchange walk in its prologue (a second context with the stand-alone
        # resolve kernel): shows what the fused launch consists of
        unfused = None
        if os.environ.get("SMMHIP_INLINE_WALK") != "0" and not args.no_unfused:
            os.environ["SMMHIP_INLINE_WALK"] = "0"
            try:
                prob2, opts2 = cm.serial_normal(N=n_glob, T=2 * ITERS_PER_STEP, device=local_rank)
                c2 = S.hip_context(prob2, opts2)
            finally:
                del os.environ["SMMHIP_INLINE_WALK"]
            c2.step(ITERS_PER_STEP)
            c2.set_profiling(2)
            c2.step(ITERS_PER_STEP)
            t2 = c2.timing()
            c2.set_profiling(0)
            ku = t2.iter_kernel_ms * 1e3 / ITERS_PER_STEP
            unfused = {"chain_kernel_us": ku, "resolve_kernel_us": t2.exch_kernel_ms * 1e3 / ITERS_PER_STEP,
                       "chain_kernel_frac": n_loc * FLOP_PER_EVAL / (ku * 1e-6) / 1e12 / PEAK_FP64_ADD_TFLOPS,
                       "note": "SMMHIP_INLINE_WALK=0: k_chain_iter without the exchange walk + k_exch_resolve_lvl as its own "
                               "launch (the configuration of the earlier round-1 profiles)"}
            del c2
        flops = n_loc * FLOP_PER_EVAL
        byts = n_loc * BYTES_PER_EVAL
        ach = flops / (k_us * 1e-6) / 1e12
        hbm = byts / (k_us * 1e-6) / 1e9
        traffic, traffic_src, traffic_stale = pmc_traffic()
        roof = {"bound": "valu_fp64", "kernel": "k_chain_iter_norm<2, true>", "achieved": ach, "peak": PEAK_FP64_ADD_TFLOPS,
                "unit": "TFLOP/s", "frac": ach / PEAK_FP64_ADD_TFLOPS, "traffic": traffic, "traffic_stale": traffic_stale,
                "traffic_note": "HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 from %s; algorithmic: %d B"
                                % (traffic_src, byts),
                "avg_kernel_us": k_us, "avg_exchange_us": x_us,
                "net_kernel_us": k_net_us, "net_exchange_us": x_net_us, "event_bracket_overhead_us": null_us,
                "timing_note": "avg_*: per-kernel start/stop events (dispatch duration, what rocprofv3 reports; used for "
                               "'achieved'); net_*: event brackets minus the empty-bracket overhead",
                "kernel_contents": "one launch per iteration: exchangeMoves! of the previous iteration (level walk by every "
                                   "workgroup, LDS/latency bound, no flops) + next_eval of 4096 chains (16 per workgroup)",
                "unfused": unfused,
                "profiled_step_ms": tm.step_ms,
                "note": "2p/2m objfunc_norm is FP64-add bound (313 flop/B, SURVEY.md 8d): peak = 256CU x 4SIMD x 16 lanes "
                        "x 2.4GHz adds/s (FMA peak 78.6 TF is unreachable: no multiplies in the algorithm)",
                "hbm": {"bound": "hbm", "achieved": hbm, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": hbm / PEAK_HBM_GBS, "traffic": traffic,
                        "note": "algorithmic 128 B per chain-eval; small by construction"}}
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(host_cores())

    if rank == 0:
        out = {"metric": "chain-evals/sec (whole node), serialNormal 2p/2m, 4096 chains x 200 iters",
               "value": value, "unit": "chain-evals/s", "n_gpus": world, "steps": K, "warmup": W,
               "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f64", "data": "synthetic",
               "config": {"workload": "serialNormal objfunc_norm 2 params / 2 moments, ns=10000, %d BGP chains per GPU "
                                      "(%d total) x %d iterations per step (BASELINE configs[1])"
                                      % (n_loc, n_glob, ITERS_PER_STEP),
                          "chains_per_gpu": n_loc, "iters_per_step": ITERS_PER_STEP, "ns": NS,
                          "exchange": "every iteration >= 2, N pairs" + (", RCCL all-gather" if world > 1 else "")},
               "roofline": roof, "cpu_baseline": cpu}
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
