// look-ahead kernels: k_pregen_rng (randomness blocks) and k_exch_plan (pair list, ranks, dependency levels) — part of libsmmhip (included by smmhip.hip inside its anonymous namespace; gfx950 device code).
#pragma once
// ------------------------------------------------------------------------------------------
// k_pregen_rng: the state-independent randomness of iterations t0 .. t0+W-1 as per-chain blocks
//   rb[w][c] = { u, z[try][k] }:  u = the MH uniform (probs_acc = rand(n), AlgoBGP.jl:85),
//   z = standard normals of mysample's first tries (rand(RAND,d), :404) — injected or generated.
// one thread per (iteration, try, parameter pair, chain).
// ------------------------------------------------------------------------------------------
// (grid: x over the (try, parameter pair, chain) triples of one iteration — consecutive threads write consecutive 16-byte pieces of a
// chain's block — y = the iteration of the window: the index arithmetic stays in 32 bits; as one linear 64-bit index its four
// divisions were most of the kernel: 482 us per 256 iterations of C4's 8192 chains)
__global__ void k_pregen_rng(const KParams P, const int t0, const int W, double* __restrict__ rb) {
    const int N = P.N, np = P.np, TR = P.rb_tries;
    const int Q = (np + 1) / 2;
    const uint32_t per_iter = (uint32_t)TR * (uint32_t)Q * (uint32_t)N;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per_iter) return;
    const int q = (int)(i % (uint32_t)Q);
    const uint32_t rest = i / (uint32_t)Q;
    const int r = (int)(rest % (uint32_t)TR);
    const int c = (int)(rest / (uint32_t)TR);
    const int w = (int)blockIdx.y;
    const int t = t0 + w;
    const uint32_t gc = (uint32_t)(P.offset + c);
    double* blk = rb + ((size_t)w * N + c) * P.RBW;
    if (t > 1) {  // iteration 1 proposes the initial value (:426-427)
        double z0, z1 = 0.0;
        if (P.user_ntab) {
            const size_t base = (((size_t)(t - 1) * TR + r) * np) * N + c;
            z0 = P.user_ntab[base + (size_t)(2 * q) * N];
            if (2 * q + 1 < np) z1 = P.user_ntab[base + (size_t)(2 * q + 1) * N];
        } else {
            rng_prop_normal2(P.seed, gc, (uint32_t)t, (uint32_t)r, (uint32_t)q, z0, z1);
        }
        blk[1 + r * np + 2 * q] = z0;
        if (2 * q + 1 < np) blk[1 + r * np + 2 * q + 1] = z1;
    }
    if (r == 0 && q == 0) blk[0] = P.user_utab ? P.user_utab[(size_t)(t - 1) * N + c] : rng_u(P.seed, gc, (uint32_t)t);
}

// ------------------------------------------------------------------------------------------
// k_exch_plan: one workgroup per iteration t = t0 + blockIdx.x.  Samples the exchange pair list
// (sample(props, K, replace=false), AlgoBGP.jl:653-656) and derives the dependency structure of
// the ordered walk (:662-691): for pair q = (i,j), r_i / r_j = number of earlier pairs touching
// chain i / chain j (counting sort of the 2K endpoints by chain: LDS atomics + block scan).
// plan[t-t0][q] = i | j<<16 | r_i<<32 | r_j<<48, plan_mi[t-t0][q] = min_improve[i].
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(XWG) void k_exch_plan(const KParams P, const int t0, unsigned long long* __restrict__ plan,
                                                   double* __restrict__ plan_mi, uint32_t* __restrict__ lv_pairs,
                                                   double* __restrict__ lv_mi, uint32_t* __restrict__ lv_off,
                                                   uint32_t* __restrict__ lv_pairs_p, uint32_t* __restrict__ lv_offp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char xsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = t0 + blockIdx.x;
    const int Ng = P.Ng, K = P.plan_K;
    uint32_t* cnt = (uint32_t*)xsm;          // [Ng+2]  histogram -> cursor (later: level histogram)
    uint32_t* ep = cnt + Ng + 2;             // [2K]  list positions bucketed by chain (later: levels)
    uint16_t* pi = (uint16_t*)(ep + 2 * K);  // [K]
    uint16_t* pj = pi + K;                   // [K]
    uint32_t* wsum = (uint32_t*)(pj + K);    // [32] (pi,pj: 4K bytes from a 4-byte aligned base)
    unsigned long long* out = plan + (size_t)blockIdx.x * K;
    double* out_mi = plan_mi + (size_t)blockIdx.x * K;

    if (P.ts && blockIdx.x == 0 && tid == 0) P.ts[(size_t)8 * 60000 + 80] = wall_clock64();
    for (int c = tid; c < Ng; c += XWG) cnt[c] = 0;
    if (P.pairtab) {
        for (int q = tid; q < K; q += XWG) {
            pi[q] = (uint16_t)P.pairtab[((size_t)(t - 1) * K + q) * 2];
            pj[q] = (uint16_t)P.pairtab[((size_t)(t - 1) * K + q) * 2 + 1];
        }
    } else {
        PairPerm pp;
        pp.init(P.seed, (uint32_t)t, (uint64_t)Ng * (uint64_t)(Ng - 1) / 2);
        for (int q = tid; q < K; q += XWG) {
            int32_t i, j;
            pair_unrank(pp.eval((uint64_t)q), i, j);
            pi[q] = (uint16_t)i;
            pj[q] = (uint16_t)j;
        }
    }
    __syncthreads();
    if (P.ts && blockIdx.x == 0 && tid == 0) P.ts[(size_t)8 * 60000 + 81] = wall_clock64();
    for (int q = tid; q < K; q += XWG) {  // histogram of endpoints
        atomicAdd(&cnt[pi[q]], 1u);
        atomicAdd(&cnt[pj[q]], 1u);
    }
    __syncthreads();
    {   // exclusive scan of cnt[0..Ng) -> bucket start
        constexpr int PER = XLDS_MAX / XWG;
        const int per = (Ng + XWG - 1) / XWG;
        const int c0 = tid * per;
        uint32_t loc[PER];
        uint32_t sum = 0;
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int c = c0 + u;
            const uint32_t v = (u < per && c < Ng) ? cnt[c] : 0u;
            loc[u] = sum;
            sum += v;
        }
        uint32_t incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t base = 0;
        for (int w = 0; w < wave; ++w) base += wsum[w];
        const uint32_t excl = base + incl - sum;
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int c = c0 + u;
            if (u < per && c < Ng) cnt[c] = excl + loc[u];
        }
    }
    __syncthreads();
    for (int q = tid; q < K; q += XWG) {  // scatter (order inside a bucket is arbitrary)
        ep[atomicAdd(&cnt[pi[q]], 1u)] = (uint32_t)q;
        ep[atomicAdd(&cnt[pj[q]], 1u)] = (uint32_t)q;
    }
    __syncthreads();  // now cnt[c] == end of chain c's bucket
    // rank = number of smaller list positions in the bucket; kept in registers for the level pass
    constexpr int MAXPP = XLDS_MAX / XWG;
    uint16_t rri[MAXPP], rrj[MAXPP];
#pragma unroll
    for (int m = 0; m < MAXPP; ++m) {
        const int q = tid + m * XWG;
        rri[m] = 0; rrj[m] = 0;
        if (q < K) {
            const uint32_t i = pi[q], j = pj[q];
            uint32_t b = i ? cnt[i - 1] : 0u, e = cnt[i], ri = 0, rj = 0;
            for (uint32_t x = b; x < e; ++x) ri += (ep[x] < (uint32_t)q) ? 1u : 0u;
            b = j ? cnt[j - 1] : 0u; e = cnt[j];
            for (uint32_t x = b; x < e; ++x) rj += (ep[x] < (uint32_t)q) ? 1u : 0u;
            out[q] = (unsigned long long)i | ((unsigned long long)j << 16) | ((unsigned long long)ri << 32) |
                     ((unsigned long long)rj << 48);
            out_mi[q] = P.min_improve_g[i];  // the threshold of the pair's colder chain, AlgoBGP.jl:688
            rri[m] = (uint16_t)ri; rrj[m] = (uint16_t)rj;
        }
    }
    __syncthreads();
    if (P.ts && blockIdx.x == 0 && tid == 0) P.ts[(size_t)8 * 60000 + 82] = wall_clock64();
    // ---- dependency levels: level(q) = 1 + max(level of q's predecessor on chain i, on chain j) ----
    // buckets re-written in rank order, so that the predecessor of rank r is the entry of rank r-1
#pragma unroll
    for (int m = 0; m < MAXPP; ++m) {
        const int q = tid + m * XWG;
        if (q < K) {
            const uint32_t i = pi[q], j = pj[q];
            ep[(i ? cnt[i - 1] : 0u) + rri[m]] = (uint32_t)q;
            ep[(j ? cnt[j - 1] : 0u) + rrj[m]] = (uint32_t)q;
        }
    }
    __syncthreads();
    int prei[MAXPP], prej[MAXPP];
    bool lasti[MAXPP], lastj[MAXPP];   // the pair is the last one of its chain i / j (the cones start there, at the end of the kernel)
#pragma unroll
    for (int m = 0; m < MAXPP; ++m) {
        const int q = tid + m * XWG;
        prei[m] = -1; prej[m] = -1; lasti[m] = false; lastj[m] = false;
        if (q < K) {
            const uint32_t i = pi[q], j = pj[q];
            if (rri[m]) prei[m] = (int)ep[(i ? cnt[i - 1] : 0u) + rri[m] - 1];
            if (rrj[m]) prej[m] = (int)ep[(j ? cnt[j - 1] : 0u) + rrj[m] - 1];
            lasti[m] = (i ? cnt[i - 1] : 0u) + rri[m] + 1u == cnt[i];
            lastj[m] = (j ? cnt[j - 1] : 0u) + rrj[m] + 1u == cnt[j];
        }
    }
    __syncthreads();  // cnt / ep are free from here on
    // (two copies of the levels, read from one and written to the other: ONE barrier per sweep; a pair's own level stays in a register)
    uint16_t* lvl = (uint16_t*)ep;          // [K]
    uint16_t* lvl_b = lvl + K;               // [K]  (ep holds 2K words)
    uint32_t* lhist = cnt;                   // [nlev+1] <= Ng+2 entries
    for (int q = tid; q < K; q += XWG) { lvl[q] = 0; lvl_b[q] = 0; }
    __syncthreads();
    {
        uint16_t own[MAXPP];
#pragma unroll
        for (int m = 0; m < MAXPP; ++m) own[m] = 0;
        const uint16_t* cur = lvl;
        uint16_t* nxt = lvl_b;
        int changed = 1;
        while (changed) {  // Jacobi sweeps: converges after (number of levels) sweeps
            int mine = 0;
#pragma unroll
            for (int m = 0; m < MAXPP; ++m) {
                const int q = tid + m * XWG;
                if (q < K) {
                    const uint32_t a = prei[m] >= 0 ? cur[prei[m]] : 0u, b = prej[m] >= 0 ? cur[prej[m]] : 0u;
                    const bool known = (prei[m] < 0 || a) && (prej[m] < 0 || b);
                    const uint16_t nl = known ? (uint16_t)(1u + (a > b ? a : b)) : (uint16_t)0;
                    if (nl != own[m]) mine = 1;
                    own[m] = nl;
                    nxt[q] = nl;
                }
            }
            changed = __syncthreads_or(mine);
            const uint16_t* x = cur; cur = nxt; nxt = (uint16_t*)x;
        }
        // (converged: the last sweep changed nothing, both copies hold the levels; everything below reads lvl)
    }
    if (P.ts && blockIdx.x == 0 && tid == 0) P.ts[(size_t)8 * 60000 + 83] = wall_clock64();
    // counting sort of the pairs by level
    for (int c = tid; c < Ng + 2; c += XWG) lhist[c] = 0;
    __syncthreads();
    for (int q = tid; q < K; q += XWG) atomicAdd(&lhist[lvl[q]], 1u);  // lhist[l] = size of level l (1-based), lhist[0] = 0
    __syncthreads();
    uint32_t* wsum2 = wsum + 16;
    __shared__ uint32_t s_nlev;
    if (tid == 0) s_nlev = 0;
    __syncthreads();
    {
        uint32_t mx = 0;
        for (int q = tid; q < K; q += XWG) mx = lvl[q] > mx ? lvl[q] : mx;
        atomicMax(&s_nlev, mx);
    }
    __syncthreads();
    const int nlev = (int)s_nlev;
    {   // exclusive scan of lhist[0..nlev] -> first position of level l (stored at lhist[l-1] after the shift below)
        constexpr int PER = XLDS_MAX / XWG + 1;
        const int n = nlev + 1;
        const int per = (n + XWG - 1) / XWG;
        const int c0 = tid * per;
        uint32_t loc[PER];
        uint32_t sum = 0;
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int c = c0 + u;
            const uint32_t v = (u < per && c < n) ? lhist[c] : 0u;
            loc[u] = sum;
            sum += v;
        }
        uint32_t incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        if (lane == 63) wsum2[wave] = incl;
        __syncthreads();
        uint32_t base = 0;
        for (int w = 0; w < wave; ++w) base += wsum2[w];
        const uint32_t excl = base + incl - sum;
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int c = c0 + u;
            if (u < per && c < n) lhist[c] = excl + loc[u];  // = number of pairs in levels < c  (level c starts here)
        }
    }
    __syncthreads();
    uint32_t* o_off = lv_off + (size_t)blockIdx.x * (K + 2);
    // o_off[l] = end of the l-th level (0-based) = start of 1-based level l+2
    for (int l = tid; l < nlev; l += XWG) o_off[l] = (l + 2 <= nlev) ? lhist[l + 2] : (uint32_t)K;
    if (tid == 0) o_off[K + 1] = (uint32_t)nlev;
    // the same list for the lean walk (smm_walk_lean.hpp): every level padded to whole waves (so that no lane has to ask
    // whether it has a pair), pair words that are LDS offsets of the chains' slots
    __shared__ uint32_t s_lst[LV_MAXLEV + 2], s_pst[LV_MAXLEV + 2];   // first (padded) position of 1-based level c
    const bool lean = lv_pairs_p != nullptr && nlev <= LV_MAXLEV;
    uint32_t* o_pp = lv_pairs_p ? lv_pairs_p + (size_t)blockIdx.x * P.plan_Kp : nullptr;
    if (lv_offp && tid == 0) {
        uint32_t* o = lv_offp + (size_t)blockIdx.x * LV_OFFP;
        o[33] = (uint32_t)nlev; o[34] = lean ? 1u : 0u;
        if (lean) {
            uint32_t acc = 0;
            for (int c = 1; c <= nlev; ++c) {
                s_lst[c] = lhist[c]; s_pst[c] = acc; o[c - 1] = acc;
                acc += ((c + 1 <= nlev ? lhist[c + 1] : (uint32_t)K) - lhist[c] + 63u) & ~63u;
            }
            s_pst[nlev + 1] = acc; o[nlev] = acc;
        }
    }
    __syncthreads();
    const uint32_t sc8 = (uint32_t)P.lean_unit;   // a word holds 8 i (byte offsets of the 8-byte slots) or, for more than 8190 chains, 4 i
    if (lean) {
        const uint32_t Ng4 = (uint32_t)((Ng + 3) & ~3);
        // two slots behind the chains' whose keys say "no swap"; wide form: one slot, value 0, as both sides
        const uint32_t dummy = P.lean_wide ? (sc8 * Ng4) * 0x10001u : (sc8 * Ng4) | ((sc8 * (Ng4 + 1u)) << 16);
        for (uint32_t q = tid; q < s_pst[nlev + 1]; q += XWG) o_pp[q] = dummy;
    }
    __syncthreads();
    uint32_t* o_pairs = lv_pairs + (size_t)blockIdx.x * K;
    double* o_mi = lv_mi + (size_t)blockIdx.x * K;
    uint16_t ci[MAXPP], cj[MAXPP], lvq[MAXPP];
#pragma unroll
    for (int m = 0; m < MAXPP; ++m) {
        const int q = tid + m * XWG;
        ci[m] = 0; cj[m] = 0; lvq[m] = 0;
        if (q < K) {
            const uint32_t lv = lvl[q];
            const uint32_t pos = atomicAdd(&lhist[lv], 1u);
            const uint32_t i = pi[q], j = pj[q];
            o_pairs[pos] = i | (j << 16);
            o_mi[pos] = P.min_improve_g[i];
            if (lean) o_pp[s_pst[lv] + (pos - s_lst[lv])] = (sc8 * i) | ((sc8 * j) << 16);
            ci[m] = (uint16_t)i; cj[m] = (uint16_t)j; lvq[m] = (uint16_t)lv;
        }
    }
    if (P.ts && blockIdx.x == 0 && tid == 0) P.ts[(size_t)8 * 60000 + 84] = wall_clock64();
    // ---- the cones (smm_cone.hpp): for every workgroup of the chain kernel the pairs its chains' outcome depends on ----
    // A pair is in a workgroup's cone when it is the last pair of one of its chains, or the predecessor (on either chain) of a pair of
    // the cone: one bit per workgroup and pair, seeded at the chains' last pairs and OR-ed into the predecessors level by level from
    // the last level down.  Then, per workgroup: its pairs counted by level, a level's pairs laid out in sub-levels of 64 words (the
    // consumer pads the last one with dummy pairs itself: a byte per sub-level says how many words count), the pairs scattered.  No
    // pass has a barrier per level except the propagation, and every thread's pairs are spread over the whole list (q = tid + 1024 m):
    // the pairs of the first levels, which are in hundreds of cones, are not one thread's.  As many workgroups per pass as their bits
    // fit the LDS next to the counters (8192 pairs: 128).
    if (P.cone_ok == nullptr) return;
    const int tiles = P.cone_tiles;
    uint32_t* o_ok = (uint32_t*)P.cone_ok + blockIdx.x;
    if (!lean) {   // (a deep plan: the walk's own fallback serves the iteration)
        if (tid == 0) *o_ok = 0u;
        return;
    }
    const int LS = nlev + 1;
    int wpp = (int)(((size_t)150 * 1024) / ((size_t)K * 4 + (size_t)32 * LS * 4 + 32 * 4));   // words of bits per pair and pass
    if (wpp > (tiles + 31) / 32) wpp = (tiles + 31) / 32;
    if (wpp < 1) { if (tid == 0) *o_ok = 0u; return; }
    __shared__ uint32_t s_bad;
    if (tid == 0) s_bad = 0u;
    uint32_t* need = (uint32_t*)xsm;                 // [K][wpp]
    uint32_t* lcnt = need + (size_t)K * wpp;         // [32 wpp][LS]: pairs of (workgroup, level); then: where the level's words start
    uint32_t* gcnt = lcnt + (size_t)32 * wpp * LS;   // [32 wpp]: chains of other workgroups in the cone (the persistent kernel's gather list)
    uint16_t* o_gl = P.cone_gather ? P.cone_gather + (size_t)blockIdx.x * tiles * CONE_GCAP : nullptr;
    uint32_t* o_hdr = (uint32_t*)P.cone_hdr + (size_t)blockIdx.x * tiles * CONE_HDRW;
    uint32_t* o_cp = (uint32_t*)P.cone_pairs + (size_t)blockIdx.x * tiles * (CONE_LEVELS * 64);
    const uint32_t ct = (uint32_t)P.cone_ct;
    for (int b0 = 0; b0 < tiles; b0 += 32 * wpp) {   // workgroups b0 .. b0 + 32 wpp - 1
        const int nb = min(32 * wpp, tiles - b0);
        __syncthreads();   // (first pass: everything in LDS is dead — pairs and levels are in registers)
        for (int x = tid; x < K * wpp + 32 * wpp * LS + 32 * wpp; x += XWG) need[x] = 0u;
        __syncthreads();
        auto seed = [&](const uint32_t c, const int q) {
            const uint32_t b = (c - (uint32_t)P.offset) / ct - (uint32_t)b0;   // (single shard: offset 0, whole tiles)
            if (b < (uint32_t)nb) atomicOr(&need[(size_t)q * wpp + (b >> 5)], 1u << (b & 31u));
        };
#pragma unroll
        for (int m = 0; m < MAXPP; ++m) {
            const int q = tid + m * XWG;
            if (q < K) {
                if (lasti[m]) seed(ci[m], q);
                if (lastj[m]) seed(cj[m], q);
            }
        }
        if (P.ts && blockIdx.x == 0 && tid == 0) P.ts[(size_t)8 * 60000 + 85] = wall_clock64();
        __syncthreads();
        for (int l = nlev; l >= 2; --l) {
#pragma unroll
            for (int m = 0; m < MAXPP; ++m) {
                const int q = tid + m * XWG;
                if (q < K && lvq[m] == l) {
                    // (the pair's words read together — 16 bytes at a time where the row is whole pieces —, then OR-ed into the predecessors' rows)
                    uint32_t* ri = prei[m] >= 0 ? &need[(size_t)prei[m] * wpp] : nullptr;
                    uint32_t* rj = prej[m] >= 0 ? &need[(size_t)prej[m] * wpp] : nullptr;
                    if ((wpp & 3) == 0) {
                        for (int w = 0; w < wpp; w += 8) {
                            const uint4 v0 = *(const uint4*)&need[(size_t)q * wpp + w];
                            const uint4 v1 = w + 4 < wpp ? *(const uint4*)&need[(size_t)q * wpp + w + 4] : make_uint4(0u, 0u, 0u, 0u);
                            const uint32_t v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
                            for (int x = 0; x < 8; ++x)
                                if (v[x]) {
                                    if (ri) atomicOr(&ri[w + x], v[x]);
                                    if (rj) atomicOr(&rj[w + x], v[x]);
                                }
                        }
                    } else
                        for (int w = 0; w < wpp; ++w) {
                            const uint32_t v = need[(size_t)q * wpp + w];
                            if (v) {
                                if (ri) atomicOr(&ri[w], v);
                                if (rj) atomicOr(&rj[w], v);
                            }
                        }
                }
            }
            __syncthreads();
        }
        if (P.ts && blockIdx.x == 0 && tid == 0) P.ts[(size_t)8 * 60000 + 86] = wall_clock64();
#pragma unroll
        for (int m = 0; m < MAXPP; ++m) {   // count
            const int q = tid + m * XWG;
            if (q < K)
                for (int w = 0; w < wpp; ++w) {
                    uint32_t bits = need[(size_t)q * wpp + w];
                    while (bits) {
                        const uint32_t b = (uint32_t)w * 32u + (uint32_t)__builtin_ctz(bits);
                        bits &= bits - 1u;
                        atomicAdd(&lcnt[b * LS + lvq[m]], 1u);
                    }
                }
        }
        __syncthreads();
        if (P.ts && blockIdx.x == 0 && tid == 0) P.ts[(size_t)8 * 60000 + 87] = wall_clock64();
        for (int b = tid; b < nb; b += XWG) {   // sub-levels: counts out, starting words in
            uint32_t sub = 0;
            uint32_t hw[CONE_HDRW];
#pragma unroll
            for (int x = 0; x < CONE_HDRW; ++x) hw[x] = 0u;
            for (int l = 1; l <= nlev; ++l) {
                const uint32_t n = lcnt[b * LS + l];
                lcnt[b * LS + l] = sub * 64u;
                for (uint32_t x = 0; x < n; x += 64u, ++sub)
                    if (sub < (uint32_t)CONE_LEVELS) {
                        const uint32_t cnt_s = n - x < 64u ? n - x : 64u;
#pragma unroll
                        for (int y = 0; y < CONE_LEVELS / 4; ++y)
                            if ((int)(sub >> 2) == y) hw[1 + y] |= cnt_s << (8u * (sub & 3u));
                    }
            }
            hw[0] = sub < (uint32_t)CONE_LEVELS ? sub : (uint32_t)CONE_LEVELS;
            if (sub > (uint32_t)CONE_LEVELS) atomicOr(&s_bad, 1u);
#pragma unroll
            for (int x = 0; x < CONE_HDRW; ++x) o_hdr[(size_t)(b0 + b) * CONE_HDRW + x] = hw[x];
        }
        __syncthreads();
        if (P.ts && blockIdx.x == 0 && tid == 0) P.ts[(size_t)8 * 60000 + 88] = wall_clock64();
#pragma unroll
        for (int m = 0; m < MAXPP; ++m) {   // scatter
            const int q = tid + m * XWG;
            if (q < K) {
                const uint32_t word = (sc8 * ci[m]) | ((sc8 * cj[m]) << 16);
                // (a pair of the first levels is in a hundred cones and its thread walks them one after the other: four cones per trip — their
                // counters' atomics in flight together, then the stores — instead of one round trip to the LDS per cone)
                const bool gi = o_gl && rri[m] == 0, gj = o_gl && rrj[m] == 0;
                const uint32_t ti = ((uint32_t)ci[m] - (uint32_t)P.offset) / ct, tj = ((uint32_t)cj[m] - (uint32_t)P.offset) / ct;
                for (int w = 0; w < wpp; ++w) {
                    uint32_t bits = need[(size_t)q * wpp + w];
                    while (bits) {
                        uint32_t bb[4], dst[4], g1[4], g2[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            bb[u] = 0xffffffffu;
                            if (bits) { bb[u] = (uint32_t)w * 32u + (uint32_t)__builtin_ctz(bits); bits &= bits - 1u; }
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            dst[u] = 0u; g1[u] = 0xffffffffu; g2[u] = 0xffffffffu;
                            if (bb[u] != 0xffffffffu) {
                                dst[u] = atomicAdd(&lcnt[bb[u] * LS + lvq[m]], 1u);
                                // a chain is in the cones its FIRST pair is in (the bits are monotone along a chain: an earlier pair of the
                                // chain carries every bit of a later one) — that is where its initial slot is needed
                                if (gi && ti != (uint32_t)b0 + bb[u]) g1[u] = atomicAdd(&gcnt[bb[u]], 1u);
                                if (gj && tj != (uint32_t)b0 + bb[u]) g2[u] = atomicAdd(&gcnt[bb[u]], 1u);
                            }
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (bb[u] != 0xffffffffu) {
                                if (dst[u] < (uint32_t)(CONE_LEVELS * 64)) o_cp[(size_t)(b0 + bb[u]) * (CONE_LEVELS * 64) + dst[u]] = word;
                                if (g1[u] < (uint32_t)CONE_GCAP) o_gl[(size_t)(b0 + bb[u]) * CONE_GCAP + g1[u]] = ci[m];
                                if (g2[u] < (uint32_t)CONE_GCAP) o_gl[(size_t)(b0 + bb[u]) * CONE_GCAP + g2[u]] = cj[m];
                            }
                    }
                }
            }
        }
        if (o_gl) {
            __syncthreads();
            for (int b = tid; b < nb; b += XWG) {
                const uint32_t g = gcnt[b];
                if (g > (uint32_t)CONE_GCAP) atomicOr(&s_bad, 1u);
                o_hdr[(size_t)(b0 + b) * CONE_HDRW] |= (g < (uint32_t)CONE_GCAP ? g : (uint32_t)CONE_GCAP) << 16;
            }
        }
    }
    __syncthreads();
    if (P.ts && blockIdx.x == 0 && tid == 0) P.ts[(size_t)8 * 60000 + 89] = wall_clock64();
    if (tid == 0) *o_ok = s_bad ? 0u : 1u;
}
// LDS of the cone passes (k_exch_plan's request is the larger of this and its own)
__host__ __device__ inline size_t plan_cone_bytes() { return (size_t)152 * 1024; }
