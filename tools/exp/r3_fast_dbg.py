import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import smm_jl_amd as S
import numpy as np, common as cm
from oracle import oracle as O
for (N, T, ns) in [(64, 300, 64), (3, 300, 300), (4096, 40, 64)]:
    prob, opts = cm.serial_normal(N=N, T=T, ns=ns)
    h = S.hip_context(prob, opts); h.step(T)
    o = O.OracleContext(prob, opts, S.Tables(Z=h.Z())); o.step(T)
    hh, ho = h.history(), o.history()
    bad = np.argwhere(hh.exchanged != ho.exchanged)
    badv = np.argwhere(~np.isclose(hh.value, ho.value, rtol=1e-9, equal_nan=True))
    print(N, T, "exchanged mismatches", len(bad), bad[:3].tolist(), "value mismatches", len(badv), badv[:3].tolist())
