// the tiles' cones of the exchange for LARGE single shards (8192 < N <= 32768 chains: BASELINE config 3 on one GPU), with the cone's
// chains numbered LOCALLY: plan side (k_cone_chains, k_cone_tiles) and consumer (exchange_walk_cone_local) — part of libsmmhip
// (included by smmhip.hip inside its anonymous namespace; gfx950 device code).
#pragma once
// ------------------------------------------------------------------------------------------
// exchangeMoves! (AlgoBGP.jl:647-716) over 32768 chains was ONE workgroup walking all 32768 pairs between two chain kernels
// (k_exch_resolve_rows: 21.7 of an iteration's 92 us, 255 CUs idle).  What a tile of 16 chains needs of that walk is its
// cone — a chain's last pair, that pair's predecessors on either chain, theirs, ...: ~100 pairs over ~100 chains, known from the pair
// list alone (smm_cone.hpp) — but the per-iteration and persistent kernels index the walk's slots by CHAIN (8 bytes x N in LDS: 262 KB
// here).  So the plan numbers a cone's chains locally: the tile's own 16 are 0..15, the others 16, 17, ... in the order the plan meets
// them (the tile's gather list), and the pair words hold the LDS offsets of those local slots.  A tile then stages ~100 slots and
// ~100 pair words, wave 0 walks them (lean_walk_levels, no barriers), and the chain kernel of 2048 tiles needs no exchange kernel
// between its launches.
//   k_cone_chains  one workgroup per iteration of the look-ahead window, behind k_exch_plan_big (whose output is the pair list in level
//                  order): the levels one after the other — their pairs share no chain — with every chain's last pair so far in LDS:
//                  each pair's predecessor on either chain, every chain's last pair (a pair's name is its position in that list);
//   k_cone_tiles   a WAVE per tile: the cone by a breadth-first walk back over the predecessor links (the pairs met in a small hash
//                  table of the wave's own), counted by level, laid out in sub-levels of 64 words (the consumer pads the last one of a
//                  level itself), the chains numbered through a second table.  6.8 KB of LDS per wave: the walk is a chain of ~15
//                  dependent L2 round trips per tile, and what hides them is the number of waves in flight.
// Formats as smm_cone.hpp's (cone_hdr: sub-levels | gathered chains << 16, a count byte per sub-level; cone_pairs: 32 x 64 words;
// cone_gather: up to 512 chain ids), cone_ok[w] = 0 when some tile's cone does not fit (more than 384 pairs, 32 sub-levels, 63 levels
// or 512 other chains): the host — it reads a window's flags once, behind the plan kernels — resolves that iteration with the stand-alone
// kernel (sampled lists stay far below the caps).
// ------------------------------------------------------------------------------------------
constexpr int CONEB_PAIRS = 384;     // pairs of a cone (16 chains: ~100, 209 the largest seen at 4096 chains)
constexpr int CONEB_CHASH = 512;     // slots of a wave's table of chains (every pair of a cone brings at most ONE chain its successor on
                                     // the walk back did not have: <= 16 + 384 entries)
constexpr int CONEB_WAVES = 4;       // tiles per workgroup of k_cone_tiles

// scratch per iteration of the window, by POSITION in the level-ordered pair list (lv_pairs): the predecessor of the pair on its first /
// second chain (position, -1: none), every chain's last pair, the pairs' levels (a byte each)
// (one 16-byte record per pair — {pair word, predecessor on its first chain, on its second, level} —: the walk below is bound by the
// number of scattered L2 transactions, and a pair costs one)
__host__ __device__ inline size_t cone_big_scratch_words(int Ng, int K) { return 4 * (size_t)K + (size_t)Ng; }
struct ConeBigScratch {
    uint4* rec; int32_t* lastpair;
    __device__ void carve(uint32_t* base, int Ng, int K) { rec = (uint4*)base; lastpair = (int32_t*)(base + 4 * (size_t)K); }
};

// one workgroup per iteration of the window: the levels in order (their pairs share no chain), the chains' last pairs so far in LDS
__global__ __launch_bounds__(XWG) void k_cone_chains(const KParams P, const uint32_t* __restrict__ lv_pairs, const uint32_t* __restrict__ lv_off,
                                                     uint32_t* __restrict__ cone_scratch) {
    extern __shared__ __attribute__((aligned(16))) int32_t cc_last[];   // [Ng]
    const int tid = threadIdx.x;
    const int Ng = P.Ng, K = P.plan_K;
    const uint32_t* __restrict__ g_pairs = lv_pairs + (size_t)blockIdx.x * K;
    const uint32_t* __restrict__ g_off = lv_off + (size_t)blockIdx.x * (K + 2);
    ConeBigScratch C;
    C.carve(cone_scratch + (size_t)blockIdx.x * cone_big_scratch_words(Ng, K), Ng, K);
    const int nlev = (int)g_off[K + 1];
    if (tid == 0) ((uint32_t*)P.cone_ok)[blockIdx.x] = nlev < 64 ? 1u : 0u;   // (k_cone_tiles counts levels in 64 lanes)
    for (int c = tid; c < Ng; c += XWG) cc_last[c] = -1;
    __syncthreads();
    uint32_t b = 0;
    for (int l = 0; l < nlev; ++l) {   // level l + 1
        const uint32_t e = g_off[l];
        for (uint32_t pos = b + tid; pos < e; pos += XWG) {
            const uint32_t w = g_pairs[pos], i = w & 0xffffu, j = w >> 16;
            C.rec[pos] = make_uint4(w, (uint32_t)cc_last[i], (uint32_t)cc_last[j], (uint32_t)(l + 1));
            cc_last[i] = (int32_t)pos; cc_last[j] = (int32_t)pos;
        }
        b = e;
        __syncthreads();
    }
    for (int c = tid; c < Ng; c += XWG) C.lastpair[c] = cc_last[c];
}

// (a wave's LDS: table of chains, level counters, the cone's list, and ONE BIT per pair of the iteration for "met": 4 KB at 32768 pairs —
// a table of the pairs met instead, 2 KB, let 20 instead of 16 tiles run per CU, but its look-ups were nested compare-and-swap loops under
// divergent control and the kernel is bound by instruction issue: EXPERIMENTS.md §R4.11)
__host__ __device__ inline size_t cone_tiles_met_words(int K) { return ((size_t)K + 127) / 128 * 4; }   // (whole 16-byte pieces)
__host__ __device__ inline size_t cone_tiles_wave_bytes(int K) { return cone_tiles_met_words(K) * 4 + CONEB_PAIRS * 2 + CONEB_CHASH * 4 + 64 * 4 * 3 + 16; }
__host__ __device__ inline size_t cone_tiles_lds_bytes(int K) { return CONEB_WAVES * ((cone_tiles_wave_bytes(K) + 15) & ~(size_t)15); }

__global__ __launch_bounds__(64 * CONEB_WAVES) void k_cone_tiles(const KParams P, const int n_iters, const uint32_t* __restrict__ cone_scratch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char cb_smem[];
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    const int Ng = P.Ng, K = P.plan_K, tiles = P.cone_tiles;
    // workgroups go to the 8 XCDs round robin: all tiles of an iteration on ONE of them, so that the iteration's records (16 bytes x K:
    // 512 KB at 32768 pairs) stay in that XCD's L2 while its tiles walk them
    const int G = (tiles + CONEB_WAVES - 1) / CONEB_WAVES;      // workgroups per iteration
    const int L = (int)blockIdx.x, xcd = L & 7, k = L >> 3;
    const int w = xcd + 8 * (k / G);
    const int tile = (k % G) * CONEB_WAVES + wave;
    if (w >= n_iters || tile >= tiles) return;   // (no workgroup barrier below: every wave is on its own)
    ConeBigScratch C;
    C.carve((uint32_t*)cone_scratch + (size_t)w * cone_big_scratch_words(Ng, K), Ng, K);
    unsigned char* base = cb_smem + (size_t)wave * ((cone_tiles_wave_bytes(K) + 15) & ~(size_t)15);
    uint32_t* chash = (uint32_t*)base;                                 // [CONEB_CHASH]: (chain + 1) << 16 | local number (0: free)
    uint32_t* lcnt = chash + CONEB_CHASH;                              // [64]: pairs of level l; then the level's cursor
    uint32_t* lsub = lcnt + 64;                                        // [64]: first sub-level of level l
    uint32_t* subc = lsub + 64;                                        // [64]: counts of the sub-levels (32 used)
    uint32_t* misc = subc + 64;                                        // [0]: gathered chains
    uint32_t* met = misc + 4;                                          // [K / 32]: a bit per pair (position) met
    const int metw = (int)cone_tiles_met_words(K);
    uint16_t* list = (uint16_t*)(met + metw);                          // [CONEB_PAIRS]: the cone's pairs (positions)
    uint32_t* o_hdr = (uint32_t*)P.cone_hdr + ((size_t)w * tiles + tile) * CONE_HDRW;
    uint32_t* o_cp = (uint32_t*)P.cone_pairs + ((size_t)w * tiles + tile) * (CONE_LEVELS * 64);
    uint16_t* o_gl = (uint16_t*)P.cone_gather + ((size_t)w * tiles + tile) * CONE_GCAP;
    const int c0 = P.offset + tile * P.cone_ct;   // (a shard lists the cones of its OWN tiles over the population's pair list: smm_chain_persist_loc.hpp)
    auto wave_sync = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); };
    for (int x = lane; x < metw / 4; x += 64) ((uint4*)met)[x] = make_uint4(0u, 0u, 0u, 0u);
    for (int x = lane; x < CONEB_CHASH; x += 64) chash[x] = 0u;
    lcnt[lane] = 0u; subc[lane] = 0u;
    if (lane == 0) misc[0] = 0u;
    wave_sync();
    auto meet = [&](const uint32_t pos) -> bool {   // true: met for the first time
        const uint32_t bit = 1u << (pos & 31u);
        return (atomicOr(&met[pos >> 5], bit) & bit) == 0u;
    };
    // ---- the cone: the chains' last pairs, then back over the predecessor links ----
    int n = 0;
    bool bad = K > 65534;
    {
        int q = -1;
        if (lane < P.cone_ct && c0 + lane < P.offset + P.N) q = C.lastpair[c0 + lane];
        const bool fresh = q >= 0 && meet((uint32_t)q);
        const unsigned long long m = __ballot(fresh);
        if (fresh) list[n + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)q;
        n += __popcll(m);
    }
    wave_sync();
    bool deep = false;   // (a pair of level 64 or more)
    for (int head = 0; head < n && !bad;) {
        const int batch = min(64, n - head);   // (what the list holds now; what this step appends is the next steps')
        int pa = -1, pb = -1;
        if (lane < batch) {   // (every pair of the list is read here exactly once, unless the list runs over: its level is counted on the way)
            const uint4 rq = C.rec[list[head + lane]];
            pa = (int)rq.y; pb = (int)rq.z;
            if (rq.w < 64u) atomicAdd(&lcnt[rq.w], 1u); else deep = true;
        }
        head += batch;
        // (appended in the order: every lane's first predecessor, then every lane's second)
        const bool fa = pa >= 0 && meet((uint32_t)pa), fb = pb >= 0 && meet((uint32_t)pb);
        const unsigned long long ma = __ballot(fa), mb = __ballot(fb), below = (1ull << lane) - 1ull;
        const int na = __popcll(ma), aa = n + __popcll(ma & below), ab = n + na + __popcll(mb & below);
        if (fa && aa < CONEB_PAIRS) list[aa] = (uint16_t)pa;
        if (fb && ab < CONEB_PAIRS) list[ab] = (uint16_t)pb;
        n += na + __popcll(mb);
        if (n > CONEB_PAIRS) { bad = true; n = CONEB_PAIRS; }   // (the list must not run over: stop)
        wave_sync();
    }
    // ---- by level (counted above): sub-levels of 64 words ----
    if (deep) bad = true;
    wave_sync();
    int nsub = 0;
    {
        const uint32_t cnt = lcnt[lane];                    // lane l: level l (level 0 does not exist)
        const uint32_t ns = (cnt + 63u) / 64u;
        uint32_t incl = ns;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(incl, off, 64); if (lane >= off) incl += o; }
        const uint32_t first = incl - ns;
        lsub[lane] = first;
        nsub = __shfl((int)incl, 63, 64);
        for (uint32_t s = 0; s < ns; ++s)
            if (first + s < 64u) subc[first + s] = s + 1u < ns ? 64u : cnt - 64u * s;
        lcnt[lane] = 0u;                                    // (from here on: the level's cursor)
    }
    if (nsub > CONE_LEVELS) { bad = true; nsub = CONE_LEVELS; }
    wave_sync();
    // ---- the tile's own chains are 0 .. cone_ct - 1 ----
    auto cslot = [&](const uint32_t chain) { return (chain * 2654435761u) >> 23; };   // 9 bits
    static_assert(CONEB_CHASH == 512 && CONEB_PAIRS + 16 < CONEB_CHASH, "cslot: 9 bits");
    if (lane < P.cone_ct && c0 + lane < P.offset + P.N) {   // (a shard's ragged last tile: the chains behind it are somebody else's)
        const uint32_t chain = (uint32_t)(c0 + lane);
        uint32_t h = cslot(chain);
        while (atomicCAS(&chash[h], 0u, ((chain + 1u) << 16) | (uint32_t)lane) != 0u) h = (h + 1u) & (CONEB_CHASH - 1);
    }
    wave_sync();
    // ---- the pairs into their sub-levels, their chains numbered as they are met ----
    for (int x0 = 0; x0 < n; x0 += 64) {
        const bool has = x0 + lane < n;
        uint32_t lv = 0u, ci = 0u, cj = 0u;
        if (has) { const uint4 rq = C.rec[list[x0 + lane]]; lv = rq.w; ci = rq.x & 0xffffu; cj = rq.x >> 16; }
        uint32_t li[2] = {0u, 0u};
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const uint32_t chain = side ? cj : ci;
            uint32_t h = cslot(chain);
            bool won = false;
            if (has) {
                for (;;) {   // (a chain being numbered in this very step shows the number 0xffff until its lane has written it)
                    const uint32_t old = atomicCAS(&chash[h], 0u, ((chain + 1u) << 16) | 0xffffu);
                    if (old == 0u) { won = true; break; }
                    if ((old >> 16) == chain + 1u) break;
                    h = (h + 1u) & (CONEB_CHASH - 1);
                }
            }
            if (won) {
                const uint32_t g = atomicAdd(&misc[0], 1u);
                chash[h] = ((chain + 1u) << 16) | (((uint32_t)P.cone_ct + g) & 0xffffu);
                if (g < (uint32_t)CONE_GCAP) o_gl[g] = (uint16_t)chain;
            }
            wave_sync();
            if (has) li[side] = chash[h] & 0xffffu;
        }
        if (has && lv < 64u) {
            const uint32_t pos = atomicAdd(&lcnt[lv], 1u);
            const uint32_t word = lsub[lv] * 64u + pos;
            if (word < (uint32_t)(CONE_LEVELS * 64)) o_cp[word] = (8u * li[0]) | ((8u * li[1]) << 16);
        }
    }
    wave_sync();
    const uint32_t ngat = misc[0];
    if (ngat > (uint32_t)CONE_GCAP) bad = true;
    if (lane < CONE_HDRW) {
        uint32_t hw;
        if (lane == 0) hw = (uint32_t)nsub | ((ngat < (uint32_t)CONE_GCAP ? ngat : (uint32_t)CONE_GCAP) << 16);
        else { const int s = 4 * (lane - 1); hw = subc[s] | (subc[s + 1] << 8) | (subc[s + 2] << 16) | (subc[s + 3] << 24); }
        o_hdr[lane] = hw;
    }
    if (__ballot(bad) != 0ull && lane == 0) ((uint32_t*)P.cone_ok)[w] = 0u;
}
