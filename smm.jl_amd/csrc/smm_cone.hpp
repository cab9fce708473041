// a workgroup's own part of the exchange walk ("cone"), consumer side: k_chain_iter's inline key walk (single shards of 4096 < N <=
// 8192 chains, BASELINE config 4) — part of libsmmhip (included by smm_chain.hpp inside smmhip.hip's anonymous namespace; gfx950
// device code).  The plan side is at the end of k_exch_plan (smm_lookahead.hpp).
#pragma once
// ------------------------------------------------------------------------------------------
// exchangeMoves! (AlgoBGP.jl:647-716) is an ordered walk over N_global pairs, and the key form of k_chain_iter walked ALL of it in
// every workgroup (8192 pairs in ~13 levels, four of them wider than the workgroup: ~9 us of a 22 us iteration, behind 100 KB of
// staging) to learn where its 32 chains end up.  Their outcome depends on few pairs: a chain's last pair, that pair's two
// predecessors (the previous pair of either chain), theirs, ... — about 200 pairs, known from the pair list alone, i.e. ahead of
// time: k_exch_plan lists them per workgroup in sub-levels of at most 64 pairs.  The prologue then
//   * fetches its list (a wave per sub-level, no dependence on anything) and, one round trip later, the initial slots
//     {order_key32(value), chain} of the chains those pairs name, out of the array the accept step wrote (slot8[]) — a few KB
//     instead of 100;
//   * lets wave 0 walk the sub-levels alone (lean_walk_levels on 64 lanes: no barriers; a lone wave needs ~190 cycles per level
//     whatever its width — the dependency depth is what the exchange costs, not the population).
// Whenever a cone does not fit CONE_LEVELS sub-levels the whole iteration is walked the old way (cone_ok[w] == 0: every workgroup
// decides the same).  min_improve == 0, dist_fun = -, like the key walk itself.
// ------------------------------------------------------------------------------------------
// false (uniform; nothing done that the key walk's staging does not overwrite): not this iteration
// (gc2 / src2: a second chain whose record source a lane wants — the dense kind's loads are spread over all lanes of the tile; -1: none)
template <int NT, int US>
__device__ inline bool exchange_walk_tile_cone(const KParams& P, const int tx, unsigned char* lds, const int tid, const bool valid, const int gc,
                                               unsigned long long& xr, const int wg, const int ts_tile, const int gc2 = -1, uint32_t* src2 = nullptr) {
    static_assert((NT == 1024 || NT == 512) && CONE_LEVELS == 32, "a wave per sub-level, two or four rounds");
    constexpr int NWV = NT / 64, RND = CONE_LEVELS / NWV;
    const int Ng = P.Ng;
    const int w = tx - P.plan_t0;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t Ng4 = (uint32_t)((Ng + 3) & ~3);
    const uint32_t pbase = 8u * (Ng4 + 4u);               // LDS offset of the pair words (the key walk's layout)
    // everything that depends on nothing is requested first
    const uint32_t okw = P.cone_ok[w];
    const uint32_t wflags = P.walk_flags[0];
    const uint32_t* __restrict__ g_hdr = P.cone_hdr + ((size_t)w * P.cone_tiles + wg) * CONE_HDRW;
    const uint32_t hv = g_hdr[min(lane, CONE_HDRW - 1)];   // lane 0: sub-levels; lanes 1..8: their counts, a byte each
    const uint32_t* __restrict__ g_cp = P.cone_pairs + ((size_t)w * P.cone_tiles + wg) * (CONE_LEVELS * 64);
    uint32_t pw[RND];
#pragma unroll
    for (int r = 0; r < RND; ++r) pw[r] = g_cp[(wv + NWV * r) * 64 + lane];   // sub-levels wv, wv + 16 (8), ... (anything past the list: not used)
    uint2 own = make_uint2(0u, 0u);
    const int c_own = wg * P.cone_ct + tid;
    if (tid < P.cone_ct) own = P.slot8[c_own];
    if (okw == 0u || wflags != 0u || (uint32_t)(size_t)lds != 0u) return false;
    const int nsub = __builtin_amdgcn_readlane((int)hv, 0) & 0xffff;   // (high half: the persistent kernel's gather count)
    uint2* slot = (uint2*)lds;
    const uint32_t dummy = ((8u >> US) * Ng4) | (((8u >> US) * (Ng4 + 1u)) << 16);   // the two slots behind the chains': keys 1 < 2, "no swap"
#pragma unroll
    for (int r = 0; r < RND; ++r) {
        const int s = wv + NWV * r;
        if (s < nsub) {   // (wave-uniform)
            const uint32_t cnt_s = ((uint32_t)__builtin_amdgcn_readlane((int)hv, 1 + (s >> 2)) >> (8 * (s & 3))) & 0xffu;
            const bool has = (uint32_t)lane < cnt_s;
            uint32_t ai, aj;
            lean_decode<US>(pw[r], ai, aj);                // LDS byte offsets of the two slots = 8 x chain
            uint2 si = make_uint2(0u, 0u), sj = si;
            if (has) { si = P.slot8[ai >> 3]; sj = P.slot8[aj >> 3]; }
            *(uint32_t*)(lds + pbase + 4u * (uint32_t)(s * 64 + lane)) = has ? pw[r] : dummy;
            if (has) { *(uint2*)(lds + ai) = si; *(uint2*)(lds + aj) = sj; }   // (a chain named twice gets the same slot twice)
        }
    }
    if (tid < P.cone_ct) slot[c_own] = own;                // (chains without a pair: their slot is the result)
    if (tid == NT - 1) { slot[Ng4] = make_uint2(1u, 0u); slot[Ng4 + 1] = make_uint2(2u, 0u); }
    __syncthreads();
    if (P.ts && tid == 0) P.ts[(size_t)ts_tile * 8 + 5] = wall_clock64();   // staged
    if (tid < 64) lean_walk_levels<64, US>(P.vals, 1, pbase, (uint32_t)(64 * lane), nsub, tid, 0);
    __syncthreads();   // (the walk was wave 0's alone; the control waves of the tiles read now)
    if (valid) {
        const uint32_t meta = slot[gc].y;
        const uint32_t partner = lean_partner<US>(lds, pbase, meta, (uint32_t)gc);
        xr = (unsigned long long)(meta & 0xffffu) | ((unsigned long long)partner << 32);
    }
    if (gc2 >= 0) *src2 = slot[gc2].y & 0xffffu;
    __syncthreads();   // (from here on the tile's blocks may overwrite the pair list)
    return true;
}

// ------------------------------------------------------------------------------------------
// The same for LARGE single shards (8192 < N <= 32768 chains), where 8 bytes of LDS per chain of the population do not fit: the plan
// numbers a cone's chains LOCALLY (smm_cone_big.hpp: k_cone_chains, k_cone_tiles) — the tile's own 16 are 0..15, the others 16, 17, ...
// in the order of its gather list — and the pair words hold the LDS offsets of those local slots.
// ------------------------------------------------------------------------------------------
constexpr int CONEB_LOCAL = 16 + CONE_GCAP + 2;                    // local slots: own chains, gathered chains, the dummy pair's two
constexpr uint32_t CONEB_PBASE = ((8u * CONEB_LOCAL + 127u) & ~127u);   // LDS offset of the pair words behind the local slots
// consumer: the prologue of a chain kernel's tile (NT lanes).  LDS from address 0: local slots | pair words | the gather list; nothing
// of the tile may have been written yet.  xr = src | (partner + 1) << 32 for the chain (valid, gc) of the control wave's lanes.
template <int NT>
__device__ inline bool exchange_walk_cone_local(const KParams& P, const int tx, unsigned char* lds, const int tid, const bool valid, const int cl,
                                                unsigned long long& xr, const int tile) {
    constexpr int NWV = NT / 64, RND = CONE_LEVELS / NWV;
    const int w = tx - P.plan_t0;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t okw = P.cone_ok[w];
    const uint32_t wflags = P.walk_flags[0];
    const uint32_t* __restrict__ g_hdr = P.cone_hdr + ((size_t)w * P.cone_tiles + tile) * CONE_HDRW;
    const uint32_t hv = g_hdr[min(lane, CONE_HDRW - 1)];
    const uint32_t* __restrict__ g_cp = P.cone_pairs + ((size_t)w * P.cone_tiles + tile) * (CONE_LEVELS * 64);
    const uint16_t* __restrict__ g_gl = P.cone_gather + ((size_t)w * P.cone_tiles + tile) * CONE_GCAP;
    const int c0 = tile * P.cone_ct;
    // (the first two sub-levels of every wave and the first NT entries of the gather list are requested TOGETHER with the header, not
    // behind it: a cone of 16 chains is ~10 sub-levels and ~100 chains — one round trip less on the tile's serial path)
    uint32_t pw[RND];
#pragma unroll
    for (int r = 0; r < RND; ++r) pw[r] = r < 2 ? g_cp[(wv + NWV * r) * 64 + lane] : 0u;
    uint32_t g_first = tid < CONE_GCAP ? (uint32_t)g_gl[tid] : 0u;
    uint2 own = make_uint2(0u, 0u);
    if (tid < P.cone_ct && c0 + tid < P.N) own = P.slot8[c0 + tid];
    const uint32_t hw0 = (uint32_t)__builtin_amdgcn_readlane((int)hv, 0);
    const int nsub = (int)(hw0 & 0xffffu), ngat = (int)(hw0 >> 16);
#pragma unroll
    for (int r = 2; r < RND; ++r) { const int s = wv + NWV * r; if (s < nsub) pw[r] = g_cp[s * 64 + lane]; }
    if (okw == 0u || wflags != 0u || (uint32_t)(size_t)lds != 0u) return false;
    uint2* slot = (uint2*)lds;
    uint16_t* s_gl = (uint16_t*)(lds + CONEB_PBASE + CONE_LEVELS * 64 * 4);
    for (int e = tid; e < ngat; e += NT) {   // the cone's other chains: their slots as the accept step wrote them
        const uint32_t g = e == tid ? g_first : (uint32_t)g_gl[e];
        s_gl[e] = (uint16_t)g;
        slot[P.cone_ct + e] = P.slot8[g];
    }
    const uint32_t nloc = (uint32_t)(P.cone_ct + ngat);
    const uint32_t dummy = (8u * nloc) | ((8u * (nloc + 1u)) << 16);   // the two slots behind the cone's: keys 1 < 2, "no swap"
#pragma unroll
    for (int r = 0; r < RND; ++r) {
        const int s = wv + NWV * r;
        if (s < nsub) {
            const uint32_t cnt_s = ((uint32_t)__builtin_amdgcn_readlane((int)hv, 1 + (s >> 2)) >> (8 * (s & 3))) & 0xffu;
            *(uint32_t*)(lds + CONEB_PBASE + 4u * (uint32_t)(s * 64 + lane)) = (uint32_t)lane < cnt_s ? pw[r] : dummy;
        }
    }
    if (tid < P.cone_ct) slot[tid] = own;
    if (tid == NT - 1) { slot[nloc] = make_uint2(1u, 0u); slot[nloc + 1u] = make_uint2(2u, 0u); }
    __syncthreads();
    if (tid < 64) {   // (the walk is this wave's instruction stream — ~65 instructions per sub-level —, and its SIMD is shared with the simulating waves of
                      // the CU's other workgroup: ahead of them while it lasts)
        __builtin_amdgcn_s_setprio(3);
        lean_walk_levels<64, 0>(P.vals, 1, CONEB_PBASE, (uint32_t)(64 * lane), nsub, tid, 0);
        __builtin_amdgcn_s_setprio(1);
    }
    __syncthreads();
    if (valid) {
        const uint32_t meta = slot[cl].y;
        uint32_t partner = lean_partner<0>(lds, CONEB_PBASE, meta, (uint32_t)cl);   // 1 + the partner's LOCAL number (0: none)
        if (partner) partner = 1u + (partner - 1u < (uint32_t)P.cone_ct ? (uint32_t)c0 + partner - 1u : (uint32_t)s_gl[partner - 1u - (uint32_t)P.cone_ct]);
        xr = (unsigned long long)(meta & 0xffffu) | ((unsigned long long)partner << 32);
    }
    __syncthreads();   // (from here on the tile's blocks may overwrite the walk's)
    return true;
}
__host__ __device__ inline size_t cone_local_lds_bytes() { return (size_t)CONEB_PBASE + CONE_LEVELS * 64 * 4 + CONE_GCAP * 2; }
