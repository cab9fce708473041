// mysample / proposal (AlgoBGP.jl:400-410, 424-471) for a tile of CT chains by ALL the tile's waves — shared by the per-iteration chain
// kernel (k_chain_iter, smm_chain.hpp: proposal batches of at least SMM_COOP_MIN_BATCH components) and the persistent tile kernel
// (k_chain_persist_tile, smm_chain_persist_tile.hpp) — part of libsmmhip (included by smmhip.hip inside its anonymous namespace;
// gfx950 device code).
#pragma once
// ------------------------------------------------------------------------------------------
// Many components (the whole proposal was 10.5 us per iteration at 50 parameters: one lane per chain and try walking all the
// components, then the redraw loop of mysample): every wave of the tile works, a chain is served by 64 * waves / CT lanes of ONE wave,
// a lane by the component pairs q = sl, sl + LPC, ... (one generator call per pair and try).  The tries are taken in order, each one
// tested by all the chain's lanes at once (a segment of the wave's ballot); the first one inside the unit box wins: same tries, same
// order, same winner as the serial form.
// ------------------------------------------------------------------------------------------
struct CoopProp {
    const double* rec; int RW;          // [CT][RW]: the records the chains continue from (parameters at 3 ..)
    double* m01; int m01w;              // [CT][m01w]: scratch, mapto_01 of the old parameters
    double* theta; int np;              // [CT][np]: the proposals (out)
    double* h; int HW;                  // [CT][HW]: scratch rows (heads, candidates at H_PARAMS ..)
    const double* rb; int RBW;          // [CT][RBW]: u, z[try][np]
    const double* cs; int csw;          // [CT][csw]: sigma at CS_SIGMA
    const double *lb, *ub;              // [np]
    unsigned long long* round_word;     // a word every tile of the workgroup sees
    unsigned long long* err;
    uint64_t seed;
    int offset, N, bs, rb_tries, user_n, smpl_iters, scout_after, scout_gl;
};
struct CoopSyncThreads { __device__ __forceinline__ void operator()() const { __syncthreads(); } };

// tid: the lane of the tile, nwv: waves per tile (LPC = 64 nwv / CT lanes per chain, a power of two), lead: one thread of the workgroup
template <int CT, class BAR>
__device__ __forceinline__ void coop_mysample(const CoopProp X, const int t, const int tile, const int tid, const int nwv, const bool lead, const BAR bar) {
    const int np = X.np, bs = X.bs, RW = X.RW, HW = X.HW, RBW = X.RBW, N = X.N;
    const int lane = tid & 63;
    const int max_tries = X.user_n ? min(X.rb_tries, X.smpl_iters) : X.smpl_iters;
    const int LPC = 64 * nwv / CT;                   // lanes per chain: 4 .. 64, a power of two
    const int cc = tid / LPC, sl = tid % LPC;
    const int cg = tile * CT + cc;
    const bool vld = cg < N;
    const int sh = (lane / LPC) * LPC;
    const unsigned long long seg = (LPC == 64 ? ~0ull : ((1ull << LPC) - 1ull)) << sh;   // this chain's lanes in the wave
    const double* rcc = X.rec + cc * RW;
    double* m01c = X.m01 + cc * X.m01w;
    double* thc = X.theta + cc * np;
    const double sgc = X.cs[cc * X.csw + CS_SIGMA];
    const uint32_t gcc = (uint32_t)(X.offset + cg);
    if (vld)
        for (int k = sl; k < np; k += LPC) {   // mapto_01 (mprob.jl:248) once per chain and parameter
            const double lbk = X.lb[k];
            m01c[k] = (rcc[3 + k] - lbk) / (X.ub[k] - lbk);
        }
    // (a word every tile of the workgroup sees: the number of the last round of shared tries somebody asked for)
    unsigned long long* round_word = X.round_word;
    if (lead) *round_word = 0ull;
    unsigned long long round_id = 0ull;
    bar();   // (a lane reads the m01 of its pairs, which other lanes of the chain may have written)
    // one try of chain `u` (its lanes: this half-wave or whatever segment serves it), components of the batch
    // [b0, b0 + bs): the point into `out`, true when inside the unit box
    auto one_try = [&](const int u, const uint32_t gu, const double sgu, const int rr, const int b0, double* out) -> bool {
        const double* m01u = X.m01 + u * X.m01w;
        const double* zzu = X.rb + u * RBW + 1;
        bool okl = true;
        for (int q = sl; 2 * q < b0 + bs; q += LPC) {
            if (2 * q + 1 < b0) continue;
            double z0, z1;
            if (rr < X.rb_tries) { z0 = zzu[rr * np + 2 * q]; z1 = 2 * q + 1 < np ? zzu[rr * np + 2 * q + 1] : 0.0; }
            else { const double2 zz2 = rng_prop_normal2_outofline(X.seed, gu, (uint32_t)t, (uint32_t)rr, (uint32_t)q); z0 = zz2.x; z1 = zz2.y; }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int k = 2 * q + e;
                if (k < b0 || k >= b0 + bs) continue;
                const double lbk = X.lb[k];
                const double span = X.ub[k] - lbk;
                const double step = sgu * (e ? z1 : z0);   // MvNormal(mu01, sigma): x = mu + sigma*z
                const double x = m01u[k] + step;
                if (!(x >= 0.0 && x <= 1.0)) okl = false;  // inclusive bounds, :405
                const double sc = x * span;
                out[k] = sc + lbk;   // mapto_ab, mprob.jl:271
            }
        }
        return okl;
    };
    const int n_pre = min(X.rb_tries, max_tries);   // tries whose normals are in the randomness block
    for (int b0 = 0; b0 < np; b0 += bs) {
        bool done = !vld;
        for (int rr = 0; rr < n_pre && __ballot(!done) != 0ull; ++rr) {   // mysample, :400-410, try rr
            const bool okl = done || one_try(cc, gcc, sgc, rr, b0, thc);   // (kept if this try wins or is the last one)
            const unsigned long long m = __ballot(okl);
            if ((m & seg) == seg) done = true;
        }
        // Later tries come from the generator (~1.4 us each: Philox4x32-10 + Box-Muller per component pair), and a
        // launch lasts as long as its unluckiest chain — with 50 parameters and adapted sigmas regularly 5-15 tries,
        // now and then 50 (C5: 26 us per launch early in a run, 36 us on average, spikes of 130).  So the tile's CT
        // lane segments ("slots") all work for the chains still open: open chain number i of n gets the slots
        // i, i + n, i + 2n, ..., each evaluating one further try, and the lowest successful try wins — the tries,
        // their order and the winner are those of the serial loop.  Scratch: the (still unused) history rows of the
        // tile: slot s keeps its candidate in row s, chain u its success mask in the head of row u.
        int base = n_pre;
        for (int rounds = 0;; ++rounds) {
            unsigned long long* head = (unsigned long long*)(X.h + cc * HW);   // [0]: open, [1]: successful offsets
            ++round_id;
            if (sl == 0) { head[0] = done ? 0ull : 1ull; head[1] = 0ull; if (!done) *round_word = round_id; }
            bar();
            if (*round_word != round_id || base >= max_tries || rounds >= X.scout_after) break;   // (uniform over the workgroup: nobody open, no try left, or the stubborn chains' turn below)
            const unsigned long long open = __ballot(lane < CT && *(const unsigned long long*)(X.h + lane * HW) != 0ull);
            const int n_open = __popcll(open);
            const int per = n_open ? CT / n_open : 0;      // tries per open chain in this round (>= 1)
            const int off = n_open ? cc / n_open : 0;
            unsigned long long mm = open;
            for (int i = n_open ? cc % n_open : 0; i > 0; --i) mm &= mm - 1ull;
            const int u = mm ? __ffsll((long long)mm) - 1 : 0;
            const int rr = base + off;
            const bool active = n_open && off < per && rr < max_tries;
            double* cand = X.h + cc * HW + H_PARAMS;
            bool okl = true;
            if (active) okl = one_try(u, (uint32_t)(X.offset + tile * CT + u), X.cs[u * X.csw + CS_SIGMA], rr, b0, cand);
            const unsigned long long m = __ballot(okl);
            if (active && (m & seg) == seg && sl == 0) atomicOr((unsigned long long*)(X.h + u * HW) + 1, 1ull << off);
            bar();
            if (active) {
                const unsigned long long won = ((const unsigned long long*)(X.h + u * HW))[1];
                if (won ? (__ffsll((long long)won) - 1 == off) : (rr == max_tries - 1)) {   // the winner (or the last try of all)
                    double* thu = X.theta + u * np;
                    for (int q = sl; 2 * q < b0 + bs; q += LPC)
                        for (int e = 0; e < 2; ++e) {
                            const int k = 2 * q + e;
                            if (k >= b0 && k < b0 + bs) thu[k] = cand[k];
                        }
                }
            }
            if (!done && head[1] != 0ull) done = true;
            bar();
            base += max(per, 1);
        }
        // The chains still open after those rounds are the stubborn ones (adapted sigmas, 50 parameters: one try in hundreds or
        // thousands is inside the box late in a run — C5: 38 us per iteration in the first 200 of 2000, 244 in the last, at 16
        // tries per 1.5 us and tile).  A try that fails fails EARLY, so the remaining tries are scouted by groups of 16 lanes, 16
        // pairs at a time, and given up at the first group of pairs with a component outside the box: 32 tries in flight
        // per tile instead of 16, most of them one trip long.  The groups TAKE their tries — a counter per chain hands them
        // out in order — from whichever chain of the tile still has tries worth making (below its lowest successful one),
        // so that the tile's unluckiest chain ends up with all 32 groups; a try that gets through all its pairs enters the
        // chain's minimum.  The lowest successful try wins — the tries, their order and the winner are the serial loop's —
        // and is then evaluated once more, in full, by the chain's own lanes (one_try: the same arithmetic as ever).
        // Scratch: the head of the chain's (still unused) history row — [0]: open, [1]: lowest successful try, [3]: next try
        // to hand out ([2] of row 0 is the workgroup's round word); double 4 of rows 0 .. CT / 8 - 1: the open chains' numbers, a byte each.
        {
            unsigned long long* head = (unsigned long long*)(X.h + cc * HW);
            ++round_id;
            const bool more = !done && base < max_tries;
            if (sl == 0) { head[0] = more ? 1ull : 0ull; head[1] = ~0ull; head[3] = (unsigned long long)base; if (more) *round_word = round_id; }
            bar();
            if (*round_word == round_id) {   // (uniform over the workgroup: somebody is open)
                const unsigned long long open = __ballot(lane < CT && *(const unsigned long long*)(X.h + lane * HW) != 0ull);
                const int n_open = __popcll(open);
                // (a BYTE per open chain, eight to double 4 of each of the first CT / 8 rows — a word no head uses: sixteen ints in
                // a row of HW = 10, one parameter and one moment, reached into row 1's head words; nothing any
                // wave reads in its ballot above lies there)
                static_assert(CT <= 64 && CT % 8 == 0, "olist: a byte per chain of the tile");
                auto olist = [&](const int idx) -> unsigned char* { return (unsigned char*)(X.h + (idx >> 3) * HW + 4) + (idx & 7); };
                if (tid < CT && ((open >> tid) & 1ull)) *olist(__popcll(open & ((1ull << tid) - 1ull))) = (unsigned char)tid;
                bar();
                if (n_open) {
                    const int GLr = 64 * nwv >= 128 ? X.scout_gl : 4;   // lanes of a group (one try at a time; the slim launch: one wave per tile)
                    const int G = tid / GLr, gj = tid - G * GLr;
                    const int g_lead = lane & ~(GLr - 1);
                    const unsigned long long gseg = ((1ull << GLr) - 1ull) << g_lead;   // the group's lanes in the wave
                    const int qlo = b0 >> 1, qhi = (min(b0 + bs, np) + 1) >> 1;   // the pairs with a component of this batch
                    const unsigned long long cap = (unsigned long long)max_tries;
                    int oi = G % n_open;        // where the group looks first
                    int u = 0, q0 = qlo;
                    unsigned long long rr = 0ull;
                    uint32_t gu = 0u;
                    double sgu = 0.0;
                    const double* m01u = X.m01;
                    bool have = false, quit = false;
                    while (__ballot(!quit) != 0ull) {
                        if (!quit && !have) {   // the group's next try: from a chain that has tries below its lowest successful one
                            int found = -1;
                            unsigned long long r0 = 0ull;
                            if (gj == 0) {
                                for (int sft = 0; sft < n_open && found < 0; ++sft) {
                                    const int idx = oi + sft < n_open ? oi + sft : oi + sft - n_open;
                                    const int v = (int)*olist(idx);
                                    unsigned long long* hv = (unsigned long long*)(X.h + v * HW);
                                    const unsigned long long lim = min(__hip_atomic_load(hv + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), cap);
                                    if (__hip_atomic_load(hv + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < lim) {
                                        r0 = __hip_atomic_fetch_add(hv + 3, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                        if (r0 < lim) { found = v; oi = idx; }
                                    }
                                }
                            }
                            found = __shfl(found, g_lead, 64);
                            r0 = (unsigned long long)(unsigned)__shfl((int)(unsigned)r0, g_lead, 64) | ((unsigned long long)(unsigned)__shfl((int)(unsigned)(r0 >> 32), g_lead, 64) << 32);
                            if (found < 0) quit = true;
                            else {
                                u = found; rr = r0; q0 = qlo; have = true;
                                gu = (uint32_t)(X.offset + tile * CT + u);
                                sgu = X.cs[u * X.csw + CS_SIGMA];
                                m01u = X.m01 + u * X.m01w;
                            }
                        }
                        bool okp = true;
                        const int q = q0 + gj;
                        if (have && q < qhi) {
                            const double2 zz2 = rng_prop_normal2_outofline(X.seed, gu, (uint32_t)t, (uint32_t)rr, (uint32_t)q);
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                const int k = 2 * q + e;
                                if (k >= b0 && k < b0 + bs && k < np) {
                                    const double step = sgu * (e ? zz2.y : zz2.x);   // MvNormal(mu01, sigma): x = mu + sigma*z
                                    const double x = m01u[k] + step;
                                    if (!(x >= 0.0 && x <= 1.0)) okp = false;         // inclusive bounds, :405
                                }
                            }
                        }
                        const bool gok = (__ballot(okp) & gseg) == gseg;   // all pairs of the trip inside the box
                        if (have) {
                            if (gok && q0 + GLr >= qhi) {   // the try's last pairs: a candidate for the chain's first successful try
                                if (gj == 0) atomicMin((unsigned long long*)(X.h + u * HW) + 1, rr);
                                have = false;
                            } else if (gok) q0 += GLr;
                            else have = false;
                        }
                    }
                }
                bar();
                if (more) {   // the chain's own lanes: its winning try in full (or, when none got through, the last one: :409 below)
                    const unsigned long long won = head[1];
                    (void)one_try(cc, gcc, sgc, won != ~0ull ? (int)won : max_tries - 1, b0, thc);
                    if (won != ~0ull) done = true;
                }
            }
            bar();
        }
        if (!done && sl == 0) {   // :409
            const unsigned long long key = ((unsigned long long)t << 34) | ((unsigned long long)gcc << 2) | (unsigned)ERRK_NO_DRAW;
            atomicMin(X.err, key);
        }
    }
}
