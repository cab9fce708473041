"""the locally numbered cones of large single shards (smm_cone_big.hpp: k_cone_chains, k_cone_tiles) against the injected pair list
walked backwards on the CPU: same pairs, list order kept along every chain (test build: smm_debug_cone).  python tools/cone_check.py [chains]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import smm_jl_amd as S, common as cm
S._abi.use_test_hooks(True)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8208
T = 6
prob, opts = cm.serial_normal(N=N, T=T, ns=32)
tab = cm.random_tables(prob, opts, tries=8)
a = S.hip_context(prob, opts, tab); a.step(3)
lib = S._abi.load_hooks()
hdr = np.zeros(9, np.uint32); pairs = np.zeros(2048, np.uint32); gl = np.zeros(512, np.uint16); info = np.zeros(4, np.int32)
for tile in (0, 1, N // 16 - 1):
    for w in (1, 2):
        rc = lib.smm_debug_cone(a._ctx, w, tile, hdr.ctypes.data_as(C.c_void_p), pairs.ctypes.data_as(C.c_void_p), gl.ctypes.data_as(C.c_void_p), info.ctypes.data_as(C.c_void_p))
        assert rc == 0, rc
        t = info[0] + w                      # iteration whose exchange this is
        pl = tab.pairs[t - 1]                # [K][2]
        nsub, ngat = int(hdr[0] & 0xffff), int(hdr[0] >> 16)
        cnts = [int((hdr[1 + (s >> 2)] >> (8 * (s & 3))) & 0xff) for s in range(nsub)]
        loc = list(range(tile * 16, tile * 16 + 16)) + [int(x) for x in gl[:ngat]]
        cone = []
        for s in range(nsub):
            for l in range(cnts[s]):
                pw = int(pairs[s * 64 + l]); cone.append((loc[(pw & 0xffff) >> 3], loc[(pw >> 16) >> 3], s))
        # expected cone on the CPU: walk the list backwards from the tile's chains
        need = set(range(tile * 16, tile * 16 + 16)); exp = []
        for q in range(len(pl) - 1, -1, -1):
            i, j = int(pl[q, 0]), int(pl[q, 1])
            if i in need or j in need:
                exp.append((i, j)); need.add(i); need.add(j)
        got = set((i, j) for (i, j, s) in cone)
        print("tile %d w %d (iteration %d, window %d+%d, ok %d): nsub %d ngat %d pairs %d | expected pairs %d (superset walk) | got - exp %d, exp - got %d"
              % (tile, w, t, info[0], info[1], info[2], nsub, ngat, len(cone), len(exp), len(got - set(exp)), len(set(exp) - got)))
        # order: along every chain the cone's pairs must come in list order
        posq = {(int(pl[q, 0]), int(pl[q, 1])): q for q in range(len(pl))}
        last = {}
        bad = 0
        for (i, j, s) in cone:
            q = posq.get((i, j), -1)
            if q < 0: bad += 1; continue
            for ch in (i, j):
                if ch in last and (last[ch][0] > q or last[ch][1] >= s): bad += 1
                last[ch] = (q, s)
        print("   order violations / unknown pairs:", bad)
