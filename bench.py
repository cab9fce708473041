#!/usr/bin/env python
"""bench.py — chain-evals/sec of the BGP hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c4|c5] [--protocol auto|p2p|records|values]

Workload (default: BASELINE.json configs[1], "C2"): serialNormal objective (2 params / 2 moments, ns = 10000 simulated draws per
moment), 4096 BGP chains per GPU, FP64.  One "step" = one job of 200 iterations over all chains (= the metric's "4096 chains x
200 iters"); successive steps continue the same chains.  value = N_global * 200 * steps / wall, inputs resident in HBM.

N > 1: one process per GPU, chains sharded in contiguous blocks, weak scaling (4096 chains per GPU).  `python bench.py --gpus N`
launches its N ranks itself (torch.distributed.run, 127.0.0.1); started under a launcher (RANK/WORLD_SIZE set) it is one rank.
The exchange step between the shards (exchangeMoves!, AlgoBGP.jl:647-716):
  p2p      every rank's accept step stores its records into every rank's window over xGMI (HIP IPC), no collective (default
           wherever its self-check against the record form passes)
  records  one RCCL all-gather of the last-accepted records per iteration
  values   RCCL all-gather of the values + all-to-all of the swapped records (long records)
Other BASELINE configurations (--workload): c3 = 32768 chains in 8 temperature levels (N_global fixed, split over the GPUs),
c4 = banana 10 params, 8192 chains, c5 = dense simulation 50 params on FP64 MFMA, 4096 chains.

On a 1-GPU lease the multi-process form can still be exercised: --same-device puts all N ranks on GPU 0 (gloo bootstrap, p2p
transport; RCCL refuses duplicate devices), with chains_per_gpu / N chains each so that every rank's kernels are resident at once.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ITERS_PER_STEP = 200
NS = 10000
PEAK_FP64_ADD_TFLOPS = 256 * 4 * 16 * 2.4e9 / 1e12  # 39.3: 256 CU x 4 SIMD x 16 f64 lanes/clk x 2.4 GHz (adds cannot be FMA'd)
PEAK_FP64_MFMA_TFLOPS = 78.6                         # dense FP64 matrix peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0

# per workload: chains per GPU (weak) or in total (fixed), algorithmic work per chain evaluation and the roofline that bounds it
WORKLOADS = {
    # 2*nm*ns FP64 adds + ~100 for proposal/objective/accept; 128 B of HBM traffic (state read 40 B + history record 88 B)
    "c2": dict(chains=4096, total=False, flop=2 * 2 * NS + 100, bytes=128, bound="valu_fp64", peak=PEAK_FP64_ADD_TFLOPS, unit="TFLOP/s",
               kernel="k_chain_persist_loc<2, false, false, false>",
               label="serialNormal objfunc_norm 2 params / 2 moments, ns=10000 (BASELINE configs[1])"),
    "c3": dict(chains=32768, total=True, flop=2 * 2 * NS + 100, bytes=128, bound="valu_fp64", peak=PEAK_FP64_ADD_TFLOPS, unit="TFLOP/s",
               kernel="k_chain_iter_norm_narrow_cone<2>",
               label="serialNormal objfunc_norm 2p/2m, ns=10000, 32768 chains = 8 temperature levels x 4096 (BASELINE configs[2])"),
    # ~80 flop vs (2*10 + 10 + 8) * 8 = 304 B per chain evaluation: an HBM / latency stream
    "c4": dict(chains=8192, total=False, flop=80, bytes=304, bound="hbm", peak=PEAK_HBM_GBS, unit="GB/s", kernel="k_chain_persist_gen",
               label="banana / Rosenbrock 10 params / 10 moments, 8192 chains (BASELINE configs[3])"),
    # BASELINE configs[4] AS WORDED (SMM_OBJ_DENSE2, round 6): x = B theta (256 x np), h1 = tanh x, g = A2 h1 (256 x 256), h2 = tanh g, y = A h2 (nm x 256) on
    # v_mfma_f64_16x16x4.  Per 16-chain tile: 16 row tiles x (13 + 64 + 16) = 1488 credited MFMAs (np = 50 padded to 52, nm = 50 to 64; the kernel EXECUTES
    # 16 x (16 + 64 + 16) = 1536: the first product runs in groups of four fragments), 2048 flop each => 190 464 flop per chain evaluation;
    # the un-padded algorithm is 2*256*50 + 2*256*256 + 2*50*256 = 182 272
    "c5": dict(chains=4096, total=False, flop=1488 * 2048 // 16, useful_flop=2 * 256 * 50 + 2 * 256 * 256 + 2 * 50 * 256, bytes=(3 * 50 + 50 + 8) * 8, bound="mfma",
               peak=PEAK_FP64_MFMA_TFLOPS, unit="TFLOP/s", kernel="k_chain_persist_tile<2, false>", mfma_per_tile=(1488, 1536),
               label="synthetic dense simulation, 50 params -> 256 hidden units (tanh) -> 256 x 256 matvec -> 256 hidden units (tanh) -> 50 moments on FP64 MFMA, "
                     "4096 chains (BASELINE configs[4] as worded: a 256x256 matvec per evaluation; SMM_OBJ_DENSE2)"),
    # the instance of rounds 2-5 (SMM_OBJ_DENSE, no 256 x 256 stage): 16 x (13 + 16) = 464 credited MFMAs per tile, 59 392 flop per evaluation; un-padded 51 200
    "c5v1": dict(chains=4096, total=False, flop=464 * 2048 // 16, useful_flop=2 * 256 * 50 + 2 * 50 * 256, bytes=(3 * 50 + 50 + 8) * 8, bound="mfma",
                 peak=PEAK_FP64_MFMA_TFLOPS, unit="TFLOP/s", kernel="k_chain_persist_tile<2, false>", mfma_per_tile=(464, 512),
                 label="synthetic dense simulation WITHOUT the 256 x 256 stage (SMM_OBJ_DENSE, the instance of rounds 2-5): 50 params -> 256 hidden units (tanh) -> 50 "
                       "moments on FP64 MFMA, 4096 chains"),
}


def kernel_source_hash():
    """sha256 (16 hex digits) over the device sources: ties a committed profile to the build it was taken from"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "smm.jl_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def newest_profile(pattern):
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    return files[-1] if files else None


def profile_tag(workload):
    """file name infix of the committed profiles of a workload: r03_pmc_summary.txt (C2), r03_c4_pmc_summary.txt, ..."""
    return "" if workload == "c2" else workload + "_"   # (no committed profile of c3: its lines carry no rocprof / traffic figures)


# the persistent launches + the single iterations at window boundaries, per workload
CHAIN_KERNELS = {"c2": ("k_chain_persist_loc<2, false, false, false>", "k_chain_persist_loc<2, false, true, false>", "k_chain_persist_norm<2>", "k_chain_iter_norm<2, true>", "k_chain_iter_norm<2, false>"),
                 "c4": ("k_chain_persist_gen", "k_chain_iter<0, 16, 2, true>"),
                 "c5": ("k_chain_persist_tile<2, false>", "k_chain_iter<2, 16, 1, true>", "k_chain_iter<2, 16, 1, false>"),
                 "c5v1": ("k_chain_persist_tile<2, false>", "k_chain_iter<2, 16, 1, true>", "k_chain_iter<2, 16, 1, false>")}


def _profile_iterations(summary_path, which):
    import re
    if not summary_path or not os.path.exists(summary_path):
        return None
    m = re.search(r"iterations_%s=(\d+)" % which, open(summary_path).read())
    return int(m.group(1)) if m else None


def pmc_traffic(kernel, workload="c2"):
    """HBM bytes per launch of the dominant kernel from the newest committed rocprofv3 PMC passes (profiles/): FETCH_SIZE and
    WRITE_SIZE are KB per launch; gfx950's FETCH_SIZE tallies wide coalesced reads at half their size (MI355X_MICROARCH.md, HBM)
    so it is doubled.  Counters cannot be read from inside the timed process, so the number comes from the profile file; `stale`
    says whether that profile was taken from other device sources than this build (the summary carries the source hash).
    The persistent chain kernel (C2) covers many iterations per launch: its figure is per ITERATION — the totals over all chain
    kernels of the profiled command / its iterations."""
    import re
    f = newest_profile("r[0-9][0-9]_%spmc_summary.txt" % profile_tag(workload))
    if not f:
        return None, None, None
    fetch = write = src_hash = None
    per_iter = kernel.startswith("k_chain_persist")
    iters = _profile_iterations(f, "pmc") if per_iter else None
    tot = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
    if workload != "c2":   # (the C4 / C5 summaries carry no hash line of their own: the C2 bundle of the same round does)
        import re as _re
        f0 = f.replace("_%spmc" % profile_tag(workload), "_pmc")
        if os.path.exists(f0):
            m0 = _re.search(r"kernel_source_sha16=([0-9a-f]+)", open(f0).read())
            src_hash = m0.group(1) if m0 else None
    for line in open(f):
        m = re.match(r"#\s*kernel_source_sha16=([0-9a-f]+)", line)
        if m:
            src_hash = m.group(1)
        if per_iter:
            if any(k in line for k in CHAIN_KERNELS.get(workload, ())):
                m = re.search(r"(FETCH_SIZE|WRITE_SIZE)\s+launches=\s*\d+\s+mean_per_launch=\s*[0-9.]+\s+total=\s*([0-9.]+)", line)
                if m:
                    tot[m.group(1)] += float(m.group(2))
        elif kernel in line:
            m = re.search(r"(FETCH_SIZE|WRITE_SIZE)\s+launches=\s*\d+\s+mean_per_launch=\s*([0-9.]+)", line)
            if m and m.group(1) == "FETCH_SIZE" and fetch is None:
                fetch = float(m.group(2))
            elif m and m.group(1) == "WRITE_SIZE" and write is None:
                write = float(m.group(2))
    if per_iter:
        if not iters or tot["FETCH_SIZE"] == 0.0:
            return None, None, None
        fetch, write = tot["FETCH_SIZE"] / iters, tot["WRITE_SIZE"] / iters
    if fetch is None or write is None:
        return None, None, None
    return (2.0 * fetch + write) * 1024.0, os.path.relpath(f, ROOT), (src_hash != kernel_source_hash())


def rocprof_kernel_us(kernel, workload="c2"):
    """average duration of the dominant kernel in the newest committed rocprofv3 --kernel-trace --stats summary (profiles/), with
    the staleness of that file against this build's device sources (a `# kernel_source_sha16=` line next to it).  The persistent
    chain kernel: per ITERATION — the total duration of all chain kernels of the profiled command / its iterations."""
    import csv
    f = newest_profile("r[0-9][0-9]_%skernel_stats.csv" % profile_tag(workload))
    if not f:
        return None, None, None
    tag = f.replace("_%skernel_stats.csv" % profile_tag(workload), "_pmc_summary.txt")
    us = None
    per_iter = kernel.startswith("k_chain_persist")
    total_ns = 0.0
    for row in csv.DictReader(l for l in open(f) if not l.startswith("#")):
        name = row.get("Name") or row.get("KernelName") or ""
        if per_iter:
            if any(k in name for k in CHAIN_KERNELS.get(workload, ())) and "TotalDurationNs" in row:
                total_ns += float(row["TotalDurationNs"])
        elif kernel in name and "AverageNs" in row:
            us = float(row["AverageNs"]) / 1e3
            break
    if per_iter:
        own = f.replace("kernel_stats.csv", "pmc_summary.txt")   # (the workload's own summary says how many iterations its commands ran)
        iters = _profile_iterations(own, "kernel_trace") or _profile_iterations(tag, "kernel_trace")
        us = total_ns / 1e3 / iters if iters and total_ns else None
    stale = None
    if os.path.exists(tag):
        import re
        m = re.search(r"kernel_source_sha16=([0-9a-f]+)", open(tag).read())
        stale = (m.group(1) != kernel_source_hash()) if m else None
    return us, os.path.relpath(f, ROOT), stale


def mfma_counter(kernel, workload):
    """SQ_INSTS_VALU_MFMA_MOPS_F64 per launch of the dominant kernel from the newest committed PMC summary of the workload"""
    import re
    f = newest_profile("r[0-9][0-9]_%spmc_summary.txt" % profile_tag(workload))
    if not f:
        return None, None
    if kernel.startswith("k_chain_persist"):   # per ITERATION: the chain kernels' totals / the iterations of the counter passes' command
        iters, tot = _profile_iterations(f, "pmc"), 0.0
        for line in open(f):
            if any(k in line for k in CHAIN_KERNELS.get(workload, ())) and "SQ_INSTS_VALU_MFMA_MOPS_F64" in line:
                m = re.search(r"total=\s*([0-9.]+)", line)
                if m:
                    tot += float(m.group(1))
        return (tot / iters, os.path.relpath(f, ROOT)) if iters and tot else (None, None)
    for line in open(f):
        if kernel in line and "SQ_INSTS_VALU_MFMA_MOPS_F64" in line:
            m = re.search(r"mean_per_launch=\s*([0-9.]+)", line)
            if m:
                return float(m.group(1)), os.path.relpath(f, ROOT)
    return None, None


def profiled_clock(kernel, workload):
    """effective shader clock (MHz) under the dominant kernel in the newest committed GRBM_GUI_ACTIVE pass (tools/clock_from_pmc.py lines in the
    workload's pmc summary): cycles the GPU was busy / the dispatches' duration — what the chip really clocked at under this load, profiler attached"""
    import re
    f = newest_profile("r[0-9][0-9]_%spmc_summary.txt" % profile_tag(workload))
    if not f:
        return None, None
    if not kernel.startswith("k_chain_persist"):
        return None, None   # (a dispatch of one iteration is too short: the counter's window is a few microseconds wider than the dispatch's timestamps)
    names = CHAIN_KERNELS.get(workload, (kernel,))
    cyc = ns = 0.0
    for line in open(f):
        if "GRBM_GUI_ACTIVE" in line and "effective_clock_MHz" in line and any(k in line for k in names):
            m = re.search(r"cycles=\s*([0-9.]+)\s+duration_ns=\s*([0-9.]+)", line)
            if m:
                cyc += float(m.group(1)); ns += float(m.group(2))
    return (cyc / 8.0 / ns * 1e3, os.path.relpath(f, ROOT)) if ns > 0 else (None, None)   # (the counter is summed over the 8 XCDs)


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
    from smm_jl_amd import workloads as cm
    from oracle import oracle as O
    n, t = 4096, ITERS_PER_STEP
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
            "sample": "the whole C2 workload: %d chains x %d iterations, OpenMP over chains (%d threads), AVX2, shock matrix "
                      "cached (no per-evaluation RNG: an upper bound for the reference's CPU path); %d such jobs, %.2f s each" % (n, t, threads, reps, dt),
            "regen_value": n2 * t2 / dt2,
            "regen_sample": "%d chains x %d iterations with the 2x10000 normals of every evaluation regenerated "
                            "(ObjExamples.jl:74-79) by the port's Philox/Box-Muller: a lower bound; %d repetitions, %.2f s each" % (n2, t2, reps2, dt2)}


def build_problem(workload, n_loc, n_glob, rank, T, device):
    """Problem / BGPOpts of one shard of the workload: smm.jl_amd/workloads.py (the package's own builders; tests/ and tools/ use the same)"""
    from smm_jl_amd import workloads
    return workloads.build_problem(workload, n_loc, n_glob, rank, T, device)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args):
    """`python bench.py --gpus N` outside a launcher: start the N ranks (one per GPU) and pass rank 0's line through"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // args.gpus)))
    return subprocess.call(cmd, env=env)


def dry_launch(rank, world, local_rank):
    """no GPU: the ranks meet over gloo and rank 0 reports who came up (the launcher's test)"""
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo")
        seen = [None] * world
        dist.all_gather_object(seen, (rank, local_rank, os.getpid()))
        dist.destroy_process_group()
    else:
        seen = [(rank, local_rank, os.getpid())]
    if rank == 0:
        print(json.dumps({"dry_launch": True, "n_gpus": world, "ranks": sorted(r for r, _, _ in seen),
                          "local_ranks": sorted(l for _, l, _ in seen), "pids": len({p for _, _, p in seen})}))


def _all_ok(torch, dist, world, ok, same_device):
    """every rank takes the same branch: the minimum of the ranks' verdicts (a collective every rank reaches, whatever happened to it)"""
    if world <= 1:
        return ok
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cpu" if same_device else "cuda")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(flag.item())


def p2p_self_check(S, cm, torch, dist, rank, world, device, same_device, n_loc=64, ns=500, T=24):
    """a short sharded run through the p2p windows and — where RCCL can run — through the record all-gather, ON THE FORM THE TIMED RUN
    WILL USE (the same N_global and chains per rank, ns = 10000, more iterations than the rows plan's first look-ahead pieces):
    identical histories or the bench takes the collective form.  Every rank executes the same sequence of collectives whatever
    happens to it: failures are caught per protocol, the barrier is always reached, the verdicts are reduced before anybody goes on."""
    import numpy as np
    from smm_jl_amd.dist import HipShardEngine, ShardedBGP
    hist, why = {}, None
    protos = ("p2p",) if same_device else ("p2p", "records")
    for proto in protos:
        ok, err, ctx, sh = True, None, None, None
        try:
            prob, opts = cm.serial_normal(N=n_loc * world, T=T, ns=ns, N_local=n_loc, chain_offset=rank * n_loc, device=device)
            ctx = S.hip_context(prob, opts)
            sh = ShardedBGP(HipShardEngine(ctx, torch.device("cuda", device)), protocol=proto)
            sh.step(T)
            sh.sync()
            h = ctx.history()
            hist[proto] = [getattr(h, f).copy() for f in h.FIELDS]
        except Exception as e:   # noqa: BLE001 -- any failure of a transport is a verdict, not a crash
            ok, err = False, "%s failed: %s" % (proto, str(e)[:200])
        if world > 1:
            dist.barrier()   # nobody drops its window while a peer may still store into it
        del sh, ctx
        if not _all_ok(torch, dist, world, ok, same_device):
            return False, (err or "%s failed on another rank" % proto)
    if same_device:
        return True, "p2p ran on %d chains per rank x %d iterations, ns = %d (same-device form: no RCCL to compare with; tests/test_gpu_p2p.py holds the bit-exact comparison)" % (n_loc, T, ns)
    same = all(np.array_equal(a, b, equal_nan=True) for a, b in zip(hist["p2p"], hist["records"]))
    if not _all_ok(torch, dist, world, same, same_device):
        return False, "p2p and the record all-gather disagree" if not same else "p2p and the record all-gather disagree on another rank"
    return True, "p2p == records, bit-exact, on the timed run's form: %d chains per rank (%d in all) x %d iterations, ns = %d" % (n_loc, n_loc * world, T, ns)


def cross_rank_check(ctx, torch, dist, rank, world, n_loc, same_device, last=8):
    """after the timed run, outside the clock: do the shards agree on the exchange?  The `exchanged` rows of the last iterations are
    all-gathered; a chain marked with partner p must find p marked in the same iteration (p may sit on any rank), and nobody is its
    own partner.  Returns (ok, note)."""
    import numpy as np
    it = ctx.state().iter
    t0 = max(0, it - last)
    ex = np.ascontiguousarray(ctx.history(t0, it).exchanged.astype(np.int32))   # [last][n_loc], partner + 1 (global chain id) or 0
    if world > 1:
        mine = torch.from_numpy(ex).to("cpu" if same_device else "cuda")
        allx = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allx, mine)
        full = np.concatenate([a.cpu().numpy() for a in allx], axis=1)       # [last][N_global]
    else:
        full = ex
    bad = 0
    for r in range(full.shape[0]):
        row = full[r]
        idx = np.nonzero(row)[0]
        partners = row[idx] - 1
        bad += int(np.count_nonzero(row[partners] == 0)) + int(np.count_nonzero(partners == idx))
    frac = float(np.count_nonzero(full)) / max(1, full.size)
    ok = bad == 0
    return ok, "last %d iterations: %.3f of the chains exchanged, %d inconsistent partner marks across %d rank(s)" % (full.shape[0], frac, bad, world)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps of 200 iterations per repetition (default: c2 20, c5 8 = the steady state past 1600 iterations, else 5)")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reps", type=int, default=None, help="repetitions of the K timed steps, each bracketed by barrier + synchronize (default: as many as make the "
                                                            "timed regions add up to >= 0.6 s, between 3 and 12, within 32 GB of history); value = the MEDIAN repetition")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c2")
    ap.add_argument("--protocol", choices=["auto", "p2p", "records", "values"], default=os.environ.get("SMM_BENCH_PROTOCOL", "auto"))
    ap.add_argument("--chains", type=int, default=None, help="chains per GPU (default: the workload's)")
    ap.add_argument("--same-device", action="store_true", help="all ranks on GPU 0 (1-GPU lease: gloo bootstrap, p2p transport)")
    ap.add_argument("--dry-launch", action="store_true", help="bring the ranks up and report them; no GPU work")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-unfused", action="store_true", help="skip the reference timings on other contexts (the one-launch-per-iteration path, the unfused kernels): profiling runs")
    args = ap.parse_args()

    under_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not under_launcher:
        sys.exit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if under_launcher and args.gpus != world:
        args.gpus = world
    if args.dry_launch:
        return dry_launch(rank, world, local_rank)

    import numpy as np
    import torch
    import torch.distributed as dist
    import smm_jl_amd as S
    from smm_jl_amd import workloads as cm

    W = WORKLOADS[args.workload]
    device = 0 if args.same_device else local_rank
    torch.cuda.set_device(device)
    force_sharded = os.environ.get("SMM_BENCH_FORCE_SHARDED") == "1"  # exercise the N>1 code path on one GPU
    sharded = world > 1 or force_sharded
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if args.same_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("cpu:gloo,cuda:nccl", device_id=torch.device("cuda", device))

    if W["total"]:
        n_glob = args.chains * world if args.chains else W["chains"]
        n_loc = n_glob // world
    else:
        n_loc = args.chains or (W["chains"] // world if args.same_device else W["chains"])
        n_glob = n_loc * world
    if args.steps is None:
        args.steps = 8 if args.workload in ("c5", "c5v1") else 20 if args.workload == "c2" else 5   # (c2: the driver's own --steps 20; 12 repetitions = 0.6 s of timed region)
    K, Wm = args.steps, args.warmup

    # the self-check runs the form the timed run will use: the same N_global and chains per rank (at least 1024: the large-population
    # kernels at 4096 per rank), ns of the workload, 72 iterations (past the first look-ahead pieces of the rows plan)
    chk = dict(n_loc=n_loc, ns=NS, T=72) if args.workload in ("c2", "c3") else dict()   # (c4 / c5: the generic form, checked on a small serialNormal population)
    protocol, proto_note = args.protocol, None
    if sharded and args.same_device and protocol in ("auto", "p2p"):
        _, proto_note = p2p_self_check(S, cm, torch, dist, rank, world, device, True, **chk)
        protocol = "p2p"
    elif sharded and protocol == "auto":
        ok, proto_note = p2p_self_check(S, cm, torch, dist, rank, world, device, False, **chk)
        protocol = "p2p" if ok else "records"
    elif not sharded:
        protocol = None

    # history capacity: every repetition's K steps + warm-up + the two profiled steps; the repetitions bounded by 32 GB of history rows
    prob0, _ = build_problem(args.workload, n_loc, n_glob, rank, 1, device)
    row_bytes = n_loc * ((8 + prob0.np + prob0.nm + 1) & ~1) * 8
    # (a single shard repeats the SAME K steps from the state behind the warm-up: one repetition's worth of history; shards continue their chains)
    reps_cap = max(1, min(args.reps or 12, int(32e9 // (row_bytes * ITERS_PER_STEP * K)))) if sharded else (args.reps or 12)
    T = ITERS_PER_STEP * ((reps_cap if sharded else 1) * K + Wm + 2)

    def barrier():
        if world > 1:
            dist.barrier()

    def timed_run(proto):
        """warm-up + K timed steps on fresh contexts; returns (ctx, run_step, sync, seconds)"""
        prob, opts = build_problem(args.workload, n_loc, n_glob, rank, T, device)
        ctx = S.hip_context(prob, opts)
        if not sharded:
            def run_step():
                ctx.step_async(ITERS_PER_STEP)
            sync = ctx.sync
        else:
            from smm_jl_amd.dist import HipShardEngine, ShardedBGP
            sh = ShardedBGP(HipShardEngine(ctx, torch.device("cuda", device)), protocol=proto)

            def run_step():
                sh.step(ITERS_PER_STEP)
            sync = sh.sync
        for _ in range(Wm):
            run_step()
        sync(); torch.cuda.synchronize(); barrier()
        # every repetition times THE SAME K steps (the iterations behind the warm-up): the state after the warm-up is read back once and uploaded
        # again before each further repetition, outside the clock (a single shard; the generator is counter-based, so the iterations are the
        # same to the bit).  Continuing the chains instead measures another workload every time: over tens of thousands of iterations sigma keeps
        # adapting, more proposals leave the box, and a step of 200 iterations drifts from 2.55 to 2.80 ms (round 6, 12 x 20 steps in a row)
        restore = (ctx.state(), ctx.history()) if not sharded else None
        times, want = [], (args.reps or 3)
        while len(times) < min(want, reps_cap):
            if restore is not None and times:
                ctx.set_state(*restore); ctx.sync(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(K):
                run_step()
            sync(); torch.cuda.synchronize(); barrier()
            dt = time.perf_counter() - t0
            if world > 1:   # (outside the clock: the MAX over the ranks, which also makes every rank take the same number of repetitions)
                tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.same_device else "cuda")
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dt = float(tt.item())
            times.append(dt)
            if args.reps is None and len(times) == 1:
                want = max(3, min(12, int(0.6 / dt) + 1))
        return ctx, run_step, sync, times

    # the p2p windows meet real links here for the first time in earnest: a failure (a peer's stores never arrive: SMM_ERR_HIP after
    # ~4 s) must not kill the bench — every rank learns of it, the contexts are made anew, and the run is repeated on the collective form
    err = None
    try:
        ctx, run_step, sync, times = timed_run(protocol)
    except Exception as e:   # noqa: BLE001
        if not (sharded and protocol == "p2p"):
            raise
        err = str(e)[:200]
    if sharded and protocol == "p2p" and world > 1:
        if not _all_ok(torch, dist, world, err is None, args.same_device):
            if args.same_device:
                raise RuntimeError("the p2p run failed and RCCL cannot run with all ranks on one device: %s" % (err or "another rank"))
            proto_note = "%s; TIMED RUN on p2p failed (%s): repeated on the record all-gather" % (proto_note, err or "on another rank")
            protocol = "records"
            barrier()
            ctx, run_step, sync, times = timed_run(protocol)
    elif err is not None:
        raise RuntimeError(err)
    dt = sorted(times)[len(times) // 2]   # the median repetition (each already the max over the ranks)
    # outside the clock: do the shards agree on the exchange of the last iterations?
    xok, xnote = cross_rank_check(ctx, torch, dist, rank, world, n_loc, args.same_device) if args.workload in ("c2", "c3", "c4", "c5", "c5v1") else (True, None)
    if not xok:
        raise RuntimeError("cross-rank check failed: " + xnote)
    evals = n_glob * ITERS_PER_STEP * K
    value = evals / dt

    # ---- roofline of the dominant kernel: the kernels' own start/stop events (hipExtLaunchKernelGGL on the library's stream):
    # dispatch begin to end as the command processor stamps it -- the same quantity rocprofv3 --kernel-trace reports
    ctx.set_profiling(2)
    run_step(); sync()
    tm = ctx.timing()
    ctx.set_profiling(0)
    barrier()
    k_us = tm.iter_kernel_ms * 1e3 / ITERS_PER_STEP
    x_us = tm.exch_kernel_ms * 1e3 / ITERS_PER_STEP
    step_us = tm.step_ms * 1e3 / ITERS_PER_STEP
    roof = None
    if k_us > 0:
        work = n_loc * (W["flop"] if W["bound"] != "hbm" else W["bytes"])
        ach = work / (k_us * 1e-6) / (1e12 if W["bound"] != "hbm" else 1e9)
        hbm = n_loc * W["bytes"] / (k_us * 1e-6) / 1e9
        norm_p2p = sharded and protocol == "p2p" and args.workload in ("c2", "c3")
        pinfo = ctx.persistent_info()
        shard_persist = norm_p2p and pinfo[1] > 0     # the shard ran the persistent form: the ring across the windows (smm_chain_persist_loc.hpp)
        kernel = (W["kernel"] if not sharded else "k_chain_persist_loc<2, false, true, false>" if shard_persist
                  else "k_chain_iter_norm_p2p<2>" if norm_p2p and n_glob <= 8192      # the walk inline
                  else "k_chain_iter_norm_p2p_rows<2>" if norm_p2p and n_glob <= 32768                       # + k_exch_resolve_rows<., true>
                  else W["kernel"].replace("true", "false"))
        traffic, traffic_src, traffic_stale = pmc_traffic(kernel, args.workload)
        prof_us, prof_src, prof_stale = rocprof_kernel_us(kernel, args.workload)
        roof = {"bound": W["bound"], "kernel": kernel, "achieved": ach, "peak": W["peak"], "unit": W["unit"], "frac": ach / W["peak"],
                "frac_rocprof": (work / (prof_us * 1e-6) / (1e12 if W["bound"] != "hbm" else 1e9) / W["peak"]) if prof_us else None,
                "rocprof_kernel_us": prof_us, "rocprof_source": prof_src, "rocprof_stale": prof_stale,
                "traffic": traffic, "traffic_stale": traffic_stale,
                "traffic_note": ("HBM bytes per %s = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 from %s; algorithmic: %d B"
                                 % ("ITERATION (all chain kernels of the profiled command / its iterations)" if kernel.startswith("k_chain_persist") else "launch",
                                    traffic_src, n_loc * W["bytes"])),
                "avg_kernel_us": k_us, "avg_exchange_us": x_us, "profiled_iteration_us": step_us,
                "profiled_other_us": max(0.0, step_us - k_us - x_us),
                "unit_of_work": ("one ITERATION of all chains: the persistent kernel covers up to a look-ahead window of iterations per launch, so every per-launch "
                                 "figure of this object (achieved, traffic, avg_kernel_us, rocprof_kernel_us) is the launches' total / the iterations they cover"
                                 if kernel.startswith("k_chain_persist") else "one launch = one iteration of all chains"),
                "timing_note": "rank 0, one profiled step: avg_kernel_us = the chain kernel's own start/stop events (dispatch duration, what "
                               "rocprofv3 reports; used for 'achieved' and 'frac'), avg_exchange_us = the stand-alone exchange resolution "
                               "where there is one, profiled_iteration_us = the PROFILED step's events / 200 (start/stop events on every kernel slow the "
                               "stream down: the unprofiled figure is ms_per_step / iters_per_step), profiled_other_us = its rest (launch boundaries, "
                               "event handling, push / collective).  frac_rocprof = the same work over the committed rocprofv3 average (the profiler lowers the clock)",
                "algorithmic_per_launch": {"flop": n_loc * W["flop"], "hbm_bytes": n_loc * W["bytes"]},
                "hbm": {"bound": "hbm", "achieved": hbm, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": hbm / PEAK_HBM_GBS,
                        "note": "algorithmic %d B per chain-eval" % W["bytes"]}}
        mhz, mhz_src = profiled_clock(kernel, args.workload)
        if mhz and W["bound"] != "hbm":
            # the paper roofs are quoted at 2.4 GHz; under FP64 load the chip clocks lower (DVFS): the same roof at the clock the profiler saw
            roof["clock"] = {"effective_MHz": mhz, "source": mhz_src, "peak_at_clock": W["peak"] * mhz / 2400.0,
                             "frac_rocprof_at_clock": (roof["frac_rocprof"] * 2400.0 / mhz) if roof["frac_rocprof"] else None,
                             "note": "GRBM_GUI_ACTIVE / dispatch duration of the chain kernels in the committed counter pass (profiler attached: compare with "
                                     "frac_rocprof, not with frac); peak_at_clock = peak x effective / 2400"}
        if W["bound"] == "mfma":
            mops, msrc = mfma_counter(kernel, args.workload)
            roof["useful"] = {"flop_per_launch": n_loc * W["useful_flop"], "achieved": n_loc * W["useful_flop"] / (k_us * 1e-6) / 1e12,
                              "frac": n_loc * W["useful_flop"] / (k_us * 1e-6) / 1e12 / W["peak"],
                              "note": "the un-padded algorithm: %d flop per evaluation" % W["useful_flop"]}
            roof["mfma_counter"] = None if mops is None else {
                "SQ_INSTS_VALU_MFMA_MOPS_F64_per_launch": mops, "flop_per_launch": mops * 512.0, "source": msrc,
                "achieved_at_avg_kernel_us": mops * 512.0 / (k_us * 1e-6) / 1e12,
                "achieved_at_rocprof_kernel_us": (mops * 512.0 / (prof_us * 1e-6) / 1e12) if prof_us else None,
                "flop_per_launch_counted_over_assumed": mops * 512.0 / work,
                "executed_over_credited_by_construction": W["mfma_per_tile"][1] / W["mfma_per_tile"][0]}
            roof["note"] = ("achieved = CREDITED MFMA flop (%d v_mfma_f64_16x16x4 per 16 chains: np / nm padded to 52 / 64; the kernel executes %d — the first product runs in "
                            "groups of four fragments, straight-line code, the padded ones are NOT credited) over the chain kernel's duration; mfma_counter = the same from "
                            "SQ_INSTS_VALU_MFMA_MOPS_F64 x 512 of the committed PMC pass.  " % W["mfma_per_tile"]
                            + ("The objective is BASELINE configs[4] as worded since round 6: x = B theta (256 x np), h1 = tanh x, g = A2 h1 (256 x 256; A2 streamed from L2 in fragment "
                               "order, 512 KB per tile and evaluation), h2 = tanh g, y = A h2 (nm x 256): 182 272 useful flop per evaluation (rounds 2-5 measured the instance "
                               "without the 256 x 256 stage: --workload c5v1).  " if args.workload == "c5" else
                               "The objective is x = B theta (256 x np), h = tanh x, y = A h (nm x 256): 51 200 useful flop per evaluation, NOT BASELINE's 256 x 256 matvec (--workload c5 is).  ")
                            + "The persistent tile kernel (smm_chain_persist_tile.hpp): one launch per look-ahead window, every figure per ITERATION; the hidden layers' tanh is part "
                              "of the numerical contract (include/smmhip.h: one exponential, one division; device and oracle bit-identical).  The instance (acc_tuner x 3000: sigma "
                              "stationary) is quoted at >= 1600 iterations (--steps 8, the default for this workload)")
        if args.workload in ("c2", "c3"):
            roof["note"] = ("2p/2m objfunc_norm is FP64-add bound (313 flop/B, SURVEY.md 8d): peak = 256CU x 4SIMD x 16 lanes x 2.4GHz adds/s "
                            "(FMA peak 78.6 TF is unreachable: no multiplies in the algorithm)")
    # the one-launch-per-iteration path on the same problem (what the persistent kernel replaced): smm_set_persistent(0)
    if roof is not None and not sharded and args.workload in ("c2", "c4") and kernel.startswith("k_chain_persist") and not args.no_unfused:
        prob1, opts1 = build_problem(args.workload, n_loc, n_glob, rank, 3 * ITERS_PER_STEP, device)
        c1 = S.hip_context(prob1, opts1)
        c1.set_persistent(False)
        c1.step(ITERS_PER_STEP)
        t1 = time.perf_counter()
        c1.step_async(ITERS_PER_STEP); c1.sync()
        w1 = time.perf_counter() - t1
        c1.set_profiling(2)
        c1.step(ITERS_PER_STEP)
        k1 = c1.timing().iter_kernel_ms * 1e3 / ITERS_PER_STEP
        roof["one_launch_per_iteration"] = {"kernel": "k_chain_iter_norm<2, true>" if args.workload == "c2" else "k_chain_iter<0, 16, 2, true>",
                                            "avg_kernel_us": k1, "us_per_iteration": w1 / ITERS_PER_STEP * 1e6,
                                            "chain_evals_per_s": n_loc * ITERS_PER_STEP / w1,
                                            "frac": (n_loc * (W["flop"] if W["bound"] != "hbm" else W["bytes"]) / (k1 * 1e-6) / (1e12 if W["bound"] != "hbm" else 1e9) / W["peak"]) if k1 > 0 else None,
                                            "note": "the same context with smm_set_persistent(0): one launch per iteration, the exchange walk in its prologue"}
        del c1
    if roof is not None and kernel.startswith("k_chain_persist"):
        info = ctx.persistent_info()
        roof["persistent"] = {"launches": info[1], "repairs": info[2]}
    # the same chain kernel without the exchange walk in its prologue (single shard, C2): what the fused launch consists of
    if roof is not None and not sharded and args.workload == "c2" and not args.no_unfused and os.path.exists(S._abi.HOOKS_LIB_PATH):
        # (a seam of the TEST build of the library, libsmmhip_hooks.so: the shipped one has no switch for it)
        os.environ["SMMHIP_INLINE_WALK"] = "0"
        try:
            S._abi.use_test_hooks(True)
            prob2, opts2 = cm.serial_normal(N=n_glob, T=2 * ITERS_PER_STEP, device=device)
            c2 = S.hip_context(prob2, opts2)
        finally:
            del os.environ["SMMHIP_INLINE_WALK"]
            S._abi.use_test_hooks(False)
        c2.step(ITERS_PER_STEP)
        c2.set_profiling(2)
        c2.step(ITERS_PER_STEP)
        t2 = c2.timing()
        ku = t2.iter_kernel_ms * 1e3 / ITERS_PER_STEP
        if ku > 0:
            roof["unfused"] = {"chain_kernel_us": ku, "resolve_kernel_us": t2.exch_kernel_ms * 1e3 / ITERS_PER_STEP,
                               "chain_kernel_frac": n_loc * W["flop"] / (ku * 1e-6) / 1e12 / PEAK_FP64_ADD_TFLOPS,
                               "note": "the chain kernel without the exchange walk + the stand-alone resolve kernel as its own launch"}
        del c2

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:   # (at N = 1 only: the contract's cpu_baseline leg)
        cpu = cpu_baseline(host_cores())
    barrier()

    if rank == 0:
        exch = "every iteration >= 2, N pairs"
        if sharded:
            exch += {"p2p": "; shards exchange through windows over xGMI (HIP IPC stores, no collective)",
                     "records": "; RCCL all-gather of the records", "values": "; RCCL all-gather of values + all-to-all of swapped records"}[protocol]
        out = {"metric": "chain-evals/sec (whole node), serialNormal 2p/2m, 4096 chains x 200 iters" if args.workload == "c2"
                         else "chain-evals/sec (whole node), %s" % args.workload,
               "value": value, "unit": "chain-evals/s", "n_gpus": world, "steps": K, "warmup": Wm,
               "ms_per_step": dt / K * 1e3,
               "repetitions": {"n": len(times), "ms_per_step_min": min(times) / K * 1e3, "ms_per_step_median": dt / K * 1e3, "ms_per_step_max": max(times) / K * 1e3,
                               "ms_per_step_all": [round(x / K * 1e3, 4) for x in times], "timed_region_s_total": sum(times), "value_best": evals / min(times),
                               "note": "each repetition = EXACTLY --steps steps bracketed by barrier + synchronize (max over the ranks); value and ms_per_step are the MEDIAN repetition's; "
                                       "a single shard repeats the SAME steps (the state behind the warm-up uploaded again before each repetition, outside the clock), shards continue their chains"},
               "iterations_run": ITERS_PER_STEP * (Wm + (len(times) if sharded else 1) * K + 1),   # warm-up + the repetitions' iterations + the profiled step: where the timed context stands
               "higher_is_better": True, "scaling": "strong" if (W["total"] or (world > 1 and args.same_device and not args.chains)) else "weak", "vs_baseline": None,
               "dtype": "f64", "data": "synthetic",
               "config": {"workload": "%s, %d BGP chains per GPU (%d total) x %d iterations per step" % (W["label"], n_loc, n_glob, ITERS_PER_STEP),
                          "chains_per_gpu": n_loc, "chains_total": n_glob, "iters_per_step": ITERS_PER_STEP, "ns": NS if args.workload in ("c2", "c3") else None,
                          "exchange": exch, "protocol": protocol, "protocol_check": proto_note, "cross_rank_check": xnote,
                          "shard_form": (None if not sharded else
                                         ("persistent: one launch per look-ahead window and rank, the ring of tagged slots in every rank's window (%d launches, %d repairs on rank 0)"
                                          % (ctx.persistent_info()[1], ctx.persistent_info()[2])) if ctx.persistent_info()[1] > 0
                                         else "per-iteration launches (the windows' inline / rows / generic forms)"),
                          "world_seen": (dist.get_world_size() if dist.is_initialized() else 1),
                          "backend": (str(dist.get_backend()) if dist.is_initialized() else None),
                          "same_device": bool(args.same_device), "forced_sharded": force_sharded},
               "vs_cpu_baseline": (None if cpu is None else {"cached_shocks_upper_bound": value / cpu["value"], "regenerated_draws_lower_bound": value / cpu["regen_value"],
                                                              "note": "value / the C port of the reference path on this box's host cores (cpu_baseline; parity unpinned: not the Julia "
                                                                      "reference).  vs_baseline stays null: BASELINE.md holds no published number for this metric"}),
               "roofline": roof, "cpu_baseline": cpu}
        print(json.dumps(out))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
