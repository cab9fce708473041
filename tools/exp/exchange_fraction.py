import sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, smm_jl_amd as S, common as cm
prob, opts = cm.serial_normal(N=4096, T=1200)
c = S.hip_context(prob, opts); c.step(1200)
h = c.history()
ex = (h.exchanged != 0)
for a,b in ((1,50),(50,200),(200,600),(600,1200)):
    e = ex[a:b]
    per_tile = e.reshape(e.shape[0], 256, 16).sum(axis=2)
    print("iters %4d-%4d: exchanged fraction %.3f; per 16-chain tile: mean %.2f, P(0) %.3f, P(<=2) %.3f, P(<=4) %.3f, max %d; accept rate %.3f"
          % (a, b, e.mean(), per_tile.mean(), (per_tile==0).mean(), (per_tile<=2).mean(), (per_tile<=4).mean(), per_tile.max(), h.accepted[a:b].mean()))
