import sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
import smm_jl_amd as S, common as cm
from oracle import oracle as O
from test_gpu_parity import dense_problem, make_pair
for npar, nm in [(1,1),(3,2),(6,5),(17,33),(50,50),(64,64)]:
    prob, opts = dense_problem(S, O, npar, nm, N=4, T=2)
    h, o = make_pair(S, O, prob, opts)
    rng = np.random.default_rng(1)
    for M in (1, 16, 200, 5000):
        p = rng.uniform(-1, 1, (npar, M))
        vh, mh, sh = h.eval_batch(p); vo, mo, so = o.eval_batch(p)
        print(npar, nm, M, "moments equal:", np.array_equal(mh, mo), "values equal:", np.array_equal(vh, vo), "max rel", np.max(np.abs(mh-mo)/(np.abs(mo)+1e-300)))
