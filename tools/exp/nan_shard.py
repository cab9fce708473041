"""ADVICE r5 (medium), by hand: two ranks as processes on one GPU, a NaN uploaded into ONE shard's state, then a step of many iterations —
where does the time go?  python tools/exp/nan_shard.py"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
W = r'''
import os, sys, time
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import smm_jl_amd as S, common as cm
from smm_jl_amd import _abi as A
from test_gpu_p2p import shard_opts
rank, G, d = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
N, T, ns = 1024, 30, 100
prob, opts = cm.serial_normal(N=N, T=T, ns=ns)
c = S.hip_context(prob, shard_opts(opts, G, rank))
handle, _ = c.p2p_init()
def put(tag, data=b""):
    open(os.path.join(d, "%%s_%%d.tmp" %% (tag, rank)), "wb").write(data); os.rename(os.path.join(d, "%%s_%%d.tmp" %% (tag, rank)), os.path.join(d, "%%s_%%d" %% (tag, rank)))
def get(tag, r):
    p = os.path.join(d, "%%s_%%d" %% (tag, r))
    while not os.path.exists(p): time.sleep(0.002)
    return open(p, "rb").read()
put("handle", handle)
for r in range(G):
    if r != rank: c.p2p_attach(r, handle=get("handle", r))
put("mapped"); [get("mapped", r) for r in range(G)]
T0 = time.time()
def say(m): print("rank %%d  %%7.2f s  %%s" %% (rank, time.time() - T0, m), flush=True)
c.p2p_step(1); c.p2p_step(10); c.p2p_finish(); c.sync(); put("mid"); [get("mid", r) for r in range(G)]
say("11 iterations done: %%s" %% (c.persistent_info(),))
st0, h0 = c.state(), c.history()
if rank == G - 1: st0.la_value[3] = np.nan
c.set_state(st0, h0)
put("up"); [get("up", r) for r in range(G)]
say("state uploaded")
try:
    c.p2p_step(19); say("p2p_step(19) returned")
    c.p2p_finish(); say("p2p_finish returned")
    c.sync(); say("sync returned")
except A.SMMHipError as e:
    say("error: %%s | %%s" %% (e, c.persistent_info()))
put("result"); [get("result", r) for r in range(G)]
''' % (ROOT, ROOT)
d = tempfile.mkdtemp()
open(os.path.join(d, "w.py"), "w").write(W)
env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
ps = [subprocess.Popen([sys.executable, os.path.join(d, "w.py"), str(r), "2", d], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
for p in ps:
    print(p.communicate(timeout=400)[0][-3000:])
