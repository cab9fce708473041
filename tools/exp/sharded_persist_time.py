"""The persistent form of a SHARD (k_chain_persist_loc<., ., true>) as G processes on the ONE GPU (HIP IPC windows, all 256 tiles resident):
us per iteration of the free-running ranks and the in-kernel phase times of the control waves (SMMHIP_TS=1), next to the single shard's.
  python tools/exp/sharded_persist_time.py [G ...]     (N_global = 4096, ns = 10000; min_improve from PT_MIN_IMPROVE)"""
import os
import pickle
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
WORKER = r"""
import os, sys, pickle, time, ctypes as C
import numpy as np
os.environ["SMMHIP_TS"] = "1"
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import smm_jl_amd as S, common as cm
from test_gpu_p2p import shard_opts
rank, G, N, d, mi = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], float(sys.argv[5])
IT, K = 200, 3
prob, opts = cm.serial_normal(N=N, T=1 + IT * (K + 1), ns=10000, min_improve=mi)
c = S.hip_context(prob, shard_opts(opts, G, rank) if G > 1 else opts)
def put(tag, data=b""):
    open(os.path.join(d, "%s_%d.tmp" % (tag, rank)), "wb").write(data); os.rename(os.path.join(d, "%s_%d.tmp" % (tag, rank)), os.path.join(d, "%s_%d" % (tag, rank)))
def get(tag, r):
    p = os.path.join(d, "%s_%d" % (tag, r)); t0 = time.time()
    while not os.path.exists(p):
        time.sleep(0.001)
        if time.time() - t0 > 120: raise SystemExit("rank %d: no %s from rank %d" % (rank, tag, r))
    return open(p, "rb").read()
if G > 1:
    handle, _ = c.p2p_init()
    put("handle", handle)
    for r in range(G):
        if r != rank: c.p2p_attach(r, handle=get("handle", r))
    put("mapped"); [get("mapped", r) for r in range(G)]
    step, fin = c.p2p_step, c.p2p_finish
else:
    step, fin = c.step_async, (lambda: None)
step(1); step(IT); c.sync()
put("warm"); [get("warm", r) for r in range(G)]
t0 = time.perf_counter()
for _ in range(K): step(IT)
c.sync()
dt = time.perf_counter() - t0
tiles = (N // G + 15) // 16
buf = np.zeros((tiles, 8), np.uint64)
S._abi.load().smm_debug_ts(c._ctx, buf.ctypes.data_as(C.c_void_p), tiles)
nit = max(int(buf[0, 6]), 1)
ph = buf[:, :6].astype(np.float64).mean(axis=0) / 100.0 / nit
fin(); c.sync()
put("result", pickle.dumps((dt / (K * IT) * 1e6, ph, nit, c.persistent_info())))
[get("result", r) for r in range(G)]
"""
mi = float(os.environ.get("PT_MIN_IMPROVE", "0"))
for G in [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8]:
    with tempfile.TemporaryDirectory() as d:
        script = os.path.join(d, "w.py")
        open(script, "w").write(WORKER.format(root=ROOT))
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs = [subprocess.Popen([sys.executable, script, str(r), str(G), "4096", d, repr(mi)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(G)]
        outs = [p.communicate(timeout=600)[0] for p in procs]
        for r, p in enumerate(procs):
            if p.returncode != 0:
                print("rank %d failed:\n%s" % (r, outs[r][-2000:]))
                sys.exit(1)
        res = [pickle.loads(open(os.path.join(d, "result_%d" % r), "rb").read()) for r in range(G)]
    us = max(r[0] for r in res)
    ph, nit, info = res[0][1], res[0][2], res[0][3]
    print("%d rank(s) x %4d chains on one GPU, min_improve %.2f: %.2f us per iteration (slowest rank), %.1f M chain-evals/s in all; rank 0: %d launches of the persistent form, %d repairs"
          % (G, 4096 // G, mi, us, 4096 / us, info[1], info[2]))
    print("   rank 0, last launch (%d iterations), us per iteration, mean over its tiles: wait at the barrier (gather) %.2f | walk %.2f | record %.2f | proposal %.2f | "
          "simulation %.2f | accept+publish %.2f | sum %.2f" % (nit, *ph, ph.sum()))
