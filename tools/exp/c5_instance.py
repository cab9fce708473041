import sys, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import smm_jl_amd as S, common as cm
from smm_jl_amd import _abi as A
from oracle import oracle as O
O.load()
OBJ = A.SMM_OBJ_DENSE2 if "v1" not in sys.argv else A.SMM_OBJ_DENSE
def run(N, T, accs, wscale=1.0, sig0=0.004):
    npar=nm=50
    rng=np.random.default_rng(3)
    prob=S.Problem(init=rng.uniform(-0.3,0.3,npar), lb=-np.ones(npar), ub=np.ones(npar), mom=rng.uniform(-0.5,0.5,nm), w=wscale*rng.uniform(0.5,2.0,nm), ns=1, objective_id=OBJ)
    opts=S.BGPOpts(N=N, maxiter=T, sigma=sig0*cm.temps(N,3), acc_tuner=accs, min_improve=np.zeros(N), seed=3, smpl_iters=100000)
    o=O.OracleContext(prob, opts, S.Tables(), threads=O.max_threads())
    o.step(T)
    h=o.history(); st=o.state()
    acc=h.accepted.astype(float)
    blocks=[acc[i:i+T//4].mean(axis=0) for i in range(0,T,T//4)]
    print("  acc by quarter: cold %s  hot %s | all %.3f | sigma cold %.4f hot %.4f | value cold %.3g hot %.3g | exch %.3f" % (
        ["%.2f"%b[:N//4].mean() for b in blocks], ["%.2f"%b[-N//4:].mean() for b in blocks], acc.mean(), st.sigma[0], st.sigma[-1], h.value[-1,0], h.value[-1,-1], (h.exchanged!=0).mean()))

N=64; T=2000
for f in (1000, 3000):
    print("acc_tuner x", f); run(N,T,f*np.geomspace(20,1,N))
