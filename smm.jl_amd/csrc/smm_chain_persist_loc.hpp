// the persistent chain kernel of the bundled objective on LOCALLY NUMBERED cones: k_chain_persist_loc — part of libsmmhip (included by
// smmhip.hip inside its anonymous namespace, behind smm_chain_persist.hpp whose helpers it shares; gfx950 device code).
#pragma once
// ------------------------------------------------------------------------------------------
// k_chain_persist_norm (smm_chain_persist.hpp, round 4) indexes everything the exchange touches by the chain's number in the POPULATION:
// 8 bytes of walk slot and 16 of gathered parameters per chain of N_global in every tile's LDS (96 of its 142 KB at 4096 chains).
// That is why it was the benchmark's kernel and nothing else: no room for the 16-byte slots a threshold needs (min_improve > 0 is the
// reference's DEFAULT, AlgoBGP.jl:522), no meaning at all for a shard of a larger population (8 x 4096: 262 KB of slots).  Here a
// tile numbers its cone's chains itself — its own 16 are 0..15, the chains of its gather list 16, 17, ... in list order — exactly
// as smm_cone_big.hpp's plan does for the large single shards:
//   * LDS per tile: 530 slots of 16 bytes, their parameters, two pair lists, two gather lists, a small hash: 70 KB whatever N_global;
//   * the plan's tables are taken as they are: k_exch_plan's (pair words on population offsets; N_global <= 8192) are re-numbered
//     by the waves that brought them, under the simulation (a hash of the gather list built by wave 3: nothing of it is on the path
//     between two publications); k_cone_tiles' (N_global > 8192) are local already;
//   * WIDE = one min_improve > 0 for all chains (dist_fun = -): 16-byte slots {value, local | stamp << 16}, the test is the
//     subtraction itself (lean_walk_levels<., ., true>), the gathered value is the record's own self-validating copy;
//   * SH = a shard of a sharded run (one process per GPU, smm_p2p.hpp's windows): the ring lives in every rank's window; a tile
//     publishes its chains' WHOLE records into its own rank's window and, into every PEER's (fire-and-forget stores over xGMI), only
//     what every walk consumes — the parameters and the value, NP + 1 self-validating granules = 48 bytes per chain and peer at two
//     parameters (round 5 pushed the 8-byte slot and all RW granules: 136) —; a tile gathers its cone from its own window (the walk
//     slot's key is derived from the gathered value), and only an EXCHANGED chain fetches the rest of its donor's record (prob,
//     status, simulated moments: swap_ev_ij!, AlgoBGP.jl:734-749) from the donor's OWNER's window by LDS-DMA, under the simulation;
//     progress is announced to all; launches of different ranks meet in a start barrier built from one word per rank (a rank
//     arrives at launch e only when its launch e - 1 has ended: nothing of the ring's last use is still being read).
// The launch's FIRST walk (an exchange the previous kernel left pending) is fed like every other: the tiles publish the records
// they start from as "iteration 0" of the launch, so there is no plain-memory path for it and a shard needs none for its peers.
// Everything else — roles of the waves, barriers, ring, tags, overrun guard, numerical contract, error convention — is
// smm_chain_persist.hpp's; results are bit-identical to every other form.
// Reference semantics: AlgoBGP.jl:589-640 (computeNextIteration!), :647-716 (exchangeMoves!), :734-749 (swap_ev_ij!).
// ------------------------------------------------------------------------------------------
constexpr int PL_LOCN = NORM_CT + CONE_GCAP + 2;                          // local slots: own chains, gathered chains, the dummy pair's
constexpr uint32_t PL_PBASE = ((16u * PL_LOCN + 127u) & ~127u);          // LDS offset of the pair lists (slots: 16 bytes reserved each, from address 0)
constexpr int PL_HASH = 1024;                                             // slots of the re-numbering table (<= 512 entries)

// the ring and its control words in a window (one per rank; a single shard has one of its own)
struct PrWin { uint32_t ctl, arrive, fin, progress, slot, rec; size_t total; };
__host__ __device__ inline PrWin pr_win_layout(const int Ng, const int RW, const int G, const int tiles_rank) {
    PrWin L;
    size_t o = 0;
    L.ctl = (uint32_t)o; o += 128;                                            // word 0: the epoch of a launch somebody gave up on
    L.arrive = (uint32_t)o; o += (size_t)128 * P2P_MAXG;                      // rank r's word: the last launch it has arrived at
    L.fin = (uint32_t)o; o += (size_t)128 * P2P_MAXG;                         // rank r: {error word of its last launch (u64), epoch (u32)}
    L.progress = (uint32_t)o; o += (((size_t)G * tiles_rank * 4) + 127) & ~(size_t)127;
    L.slot = (uint32_t)o; o += ((size_t)PR_K * ((size_t)Ng + 4) * 8 + 127) & ~(size_t)127;
    L.rec = (uint32_t)o; o += ((size_t)PR_K * (size_t)Ng * RW * 16 + 127) & ~(size_t)127;
    L.total = o;
    return L;
}

struct PersistLocArgs {
    const uint32_t* cone_hdr; const uint32_t* cone_pairs; const uint16_t* cone_gather; const uint32_t* cone_ok;
    unsigned char* win[P2P_MAXG];     // the ranks' windows (a single shard: win[0] = self)
    unsigned char* self;              // this rank's window
    uint32_t o_ctl, o_arrive, o_progress, o_slot, o_rec;
    double* cs; const double* rec_in; double* rec_out; double* vals_out; uint2* slot8_out; uint32_t* walk_flags;
    double* hrec; unsigned long long* err; unsigned long long* ts;
    const double *Z, *lb, *ub, *mom, *w, *objp;
    const double* rb;                 // randomness blocks of injected tables (null: drawn in the kernel)
    int N, Ng, offset, G, rank, ns, zstride, plan_t0, exch_from, sigma_update_steps, smpl_iters, t0, t1;
    int rb_t0, RBW, rb_tries, user_n, failbox;
    int ring_k, slow_tile, slow_ticks, walk_first;
    int tables_local;                 // the plan's pair words name local slots already (k_cone_tiles); else population offsets in units of 1 << unit_sh bytes
    int unit_sh;
    uint32_t epoch;
    double sigma_adjust_by, thr;
    uint64_t seed;
    unsigned long long tmo;           // ticks a spin may last
    const double* mi_g;               // WIDE: min_improve of every chain of the population (AlgoBGP.jl:522; the pair (i, j) is tested against chain i's, :688)
};
__host__ __device__ inline size_t persist_loc_smem_bytes(const int np) {
    const size_t hw = (size_t)((H_PARAMS + 2 * np + 1) & ~1);
    const size_t dbl = (size_t)NORM_CT * np + (size_t)np * 8 * NORM_CT + (size_t)NORM_CT * persist_line(np) + 2 * 64 * (size_t)(1 + 2 * np) +
                       2 * NORM_CT * hw + (size_t)PR_ZR * 64 + 16 + 16 + 2 * 64 * 2 + (size_t)PL_LOCN * np + (size_t)PL_LOCN;   // (+ a threshold per local slot: the wide walk's)
    return (size_t)PL_PBASE + 2 * (size_t)CONE_LEVELS * 64 * 4 + 2 * (size_t)CONE_GCAP * 2 + 4 * 16 * 4 + 2 * (size_t)PL_HASH * 4 + dbl * 8;
}

template <int NP>
struct PersistLocLds {
    using L = NormLayout<NP>;
    static constexpr int CT = NORM_CT, RW = L::RW, HW = L::HW, LW = PR_STW + RW, RNGW = 1 + 2 * NP;
    uint32_t pbase, gbase, hbase, tbase;
    uint32_t* s_hdr;
    double *s_theta, *s_part, *s_st, *s_rng, *s_hrow, *s_xrow, *s_z0, *s_const, *s_gth, *s_thr;
    uint4* s_donor;
    unsigned long long* s_ts;
    unsigned* s_arrived; int* s_minprog; unsigned* s_abort; unsigned* s_xmask; int* s_glready; int* s_pub; int* s_hready;
    __device__ inline PersistLocLds(unsigned char* lds) {
        pbase = PL_PBASE;
        gbase = pbase + 2u * CONE_LEVELS * 64 * 4;
        hbase = gbase + 2u * CONE_GCAP * 2;
        tbase = hbase + 4u * 16 * 4;
        s_hdr = (uint32_t*)(lds + hbase);
        s_theta = (double*)(lds + tbase + 2u * PL_HASH * 4);
        s_part = s_theta + CT * NP;
        s_st = s_part + NP * 8 * CT;
        s_rng = s_st + CT * LW;
        s_hrow = s_rng + 2 * 64 * RNGW;
        s_xrow = s_hrow + CT * HW;
        s_z0 = s_xrow + CT * HW;
        s_const = s_z0 + PR_ZR * 64;
        s_ts = (unsigned long long*)(s_const + 16);
        s_arrived = (unsigned*)(s_ts + 8);
        s_minprog = (int*)(s_arrived + 1);
        s_abort = s_arrived + 2;
        s_xmask = s_arrived + 3;
        s_glready = (int*)(s_arrived + 4);    // the exchange whose gather list has landed
        s_pub = (int*)(s_arrived + 5);        // the iteration this tile has published
        s_hready = (int*)(s_arrived + 6);     // the exchange whose re-numbering table is built
        s_donor = (uint4*)(s_const + 16 + 16);
        s_gth = s_const + 16 + 16 + 2 * 64 * 2;   // [PL_LOCN][NP]: the parameters of the cone's chains' last accepted records, by LOCAL number
        s_thr = s_gth + PL_LOCN * NP;             // [PL_LOCN]: min_improve of the chain AT that local position (the wide walk's per-position thresholds)
    }
};

// the re-numbering table of one exchange: population chain -> local number (16 + position in the gather list)
__device__ inline uint32_t pl_hslot(const uint32_t chain) { return (chain * 2654435761u) >> 22; }   // 10 bits
static_assert(PL_HASH == 1024, "pl_hslot: 10 bits");
__device__ inline void pl_hash_put(uint32_t* tab, const uint32_t chain, const uint32_t local) {
    uint32_t h = pl_hslot(chain);
    while (atomicCAS(&tab[h], 0u, ((chain + 1u) << 16) | local) != 0u) h = (h + 1u) & (PL_HASH - 1);
}
__device__ inline uint32_t pl_hash_get(const uint32_t* tab, const uint32_t chain) {
    uint32_t h = pl_hslot(chain);
    for (int probe = 0; probe < PL_HASH; ++probe) {
        const uint32_t v = tab[h];
        if ((v >> 16) == chain + 1u) return v & 0xffffu;
        if (v == 0u) break;
        h = (h + 1u) & (PL_HASH - 1);
    }
    return 0xffffu;   // (not in the cone's gather list: the plan and the kernel disagree — the caller reports it)
}

// GUARD of the lean walk on local numbers: the exact value of local chain s (a key tie) out of the ring
struct PersistLocWalkValues {
    PrWait W; const uint4* ring; const uint16_t* gl; uint32_t c0g; uint32_t tag; int RW; int NPV; int t;
    __device__ __forceinline__ double value(const uint32_t s) const {
        const uint32_t g = s < (uint32_t)NORM_CT ? c0g + s : (uint32_t)gl[s - NORM_CT];
        return pr_tie_value(W, ring, nullptr, tag, RW, NPV, t, g);
    }
};
// (the ring's words travel at the system scope — sc0 sc1: pr_store8 / pr_store_ll / pr_dma16 and the loads of smm_chain_persist.hpp —
// whether the tiles are one device's or the ranks': a remote store lands in this device's memory behind its L2)

// progress of the slowest tile of ALL ranks in this launch; a word of a LATER launch counts as "through" (its rank has left this
// launch behind: nothing of it is waited for any more), of an earlier one as "not started"
__device__ inline int pl_min_progress(const uint32_t* pr_progress, const uint32_t epoch, const int tiles, const int lane) {
    uint32_t m = 0xfffu;
    for (int b = lane; b < tiles; b += 64) {
        const uint32_t w = pr_load4_sys(pr_progress + b);
        const int d = (int)(((w >> 12) - epoch) << 12) >> 12;   // (20-bit epochs, wrap-safe)
        const uint32_t rel = d == 0 ? (w & 0xfffu) : (d > 0 ? 0xfffu : 0u);
        m = min(m, rel);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, off, 64));
    return (int)m;
}
__device__ __attribute__((noinline)) void pl_wait_progress(const PrWait W, const uint32_t* pr_progress, int* s_minprog, const int need, const int tiles, const int lane,
                                                           const int t, const int chain) {
    unsigned spins = 0;
    const unsigned long long w0 = wall_clock64();
    while (__hip_atomic_load(s_minprog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need && *W.s_abort == 0u) {
        const int m = pl_min_progress(pr_progress, W.epoch, tiles, lane);
        if (lane == 0) __hip_atomic_store(s_minprog, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((++spins & 15u) == 0u && pr_give_up(W, w0)) { if (lane == 0) pr_abort(W, t, chain); break; }
        __builtin_amdgcn_s_sleep(4);
    }
}
// out of line: the gather of the wide form — three self-validating pieces of the same chain's record, looked at again together
struct PlGather3 { uint4 q0, q1, q2; };
__device__ __attribute__((noinline)) PlGather3 pl_wait_gather3(const PrWait W, const uint4* p0, const uint4* p1, const uint4* p2, const uint32_t tag, const int t, const int g) {
    PlGather3 o;
    o.q0 = make_uint4(0u, 0u, 0u, 0u); o.q1 = o.q0; o.q2 = o.q0;
    if (*W.s_abort) return o;
    const unsigned long long w0 = wall_clock64();
    unsigned spins = 0;
    for (;;) {
        __builtin_amdgcn_s_sleep(2);
        p2p_u32x4 q0, q1, q2;
        asm volatile("global_load_dwordx4 %0, %3, off sc0 sc1\n\tglobal_load_dwordx4 %1, %4, off sc0 sc1\n\tglobal_load_dwordx4 %2, %5, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(q0), "=&v"(q1), "=&v"(q2) : "v"(p0), "v"(p1), "v"(p2) : "memory");
        o.q0 = make_uint4(q0.x, q0.y, q0.z, q0.w); o.q1 = make_uint4(q1.x, q1.y, q1.z, q1.w); o.q2 = make_uint4(q2.x, q2.y, q2.z, q2.w);
        if (p2p_ll_ok(o.q0, tag) && p2p_ll_ok(o.q1, tag) && p2p_ll_ok(o.q2, tag)) break;
        if ((++spins & 63u) == 0u && pr_give_up(W, w0)) { pr_abort(W, t, g); break; }
    }
    return o;
}

// (PCT: thresholds by chain — WIDE single shards only; a form of its own so that ONE threshold for all chains pays nothing for it: 11.84 -> 12.06 us per iteration when
// the per-position read was always there, 12.27 as a run-time branch)
template <int NP, bool WIDE, bool SH, bool PCT = false>
__global__ __launch_bounds__(NORM_WG, 4) void k_chain_persist_loc(const PersistLocArgs A) {
    static_assert(!PCT || (WIDE && !SH), "thresholds by chain: the wide walk of a single shard");
    static_assert(NP == 1 || NP == 2, "one moment per half of the workgroup");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    using LY = PersistLocLds<NP>;
    constexpr int CT = NORM_CT, RW = LY::RW, HW = LY::HW, LW = LY::LW, RNGW = LY::RNGW;
    constexpr int NPC = RW / 2, NPH = HW / 2;   // 16-byte pieces of a record / a history row
    constexpr uint32_t SB = WIDE ? 16u : 8u;    // bytes of a walk slot
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = (int)blockIdx.x, tiles = (int)gridDim.x;
    const int N = A.N;
    const LY Y(lds);
    const uint32_t epoch = A.epoch;
    const int t0 = A.t0, t1 = A.t1;
    unsigned char* const mine = A.self;
    uint32_t* const pr_ctl = (uint32_t*)(mine + A.o_ctl);
    uint32_t* const pr_progress = (uint32_t*)(mine + A.o_progress);
    const PrWait W{A.err, pr_ctl, Y.s_abort, A.epoch, A.tmo};
    const int rmask = A.ring_k - 1;
    const int tiles_all = SH ? tiles * A.G : tiles;
    const uint32_t c0g = (uint32_t)(A.offset + tile * CT);   // the tile's first chain in the population
    const bool exch_any = A.Ng > 1;
    auto exch_on = [&](const int tx) { return exch_any && tx >= A.exch_from; };   // AlgoBGP.jl:637

    if (error_before(*(const volatile unsigned long long*)A.err, t0)) return;   // an EARLIER launch raised a hard error

    // ---- once per launch: the tile's chain state and records, the constants, wave 0's shocks into LDS ----
    if (tid < 64) {
        const int cl = lane >> 2, r = lane & 3, c = tile * CT + cl;
        if (c < N) {
            const double2* g_cs = (const double2*)(A.cs + (size_t)c * CSW);
            const double2* g_rec = (const double2*)(A.rec_in + (size_t)c * RW);
            double2* st2 = (double2*)(Y.s_st + cl * LW);
            for (int i = r; i < 6; i += 4) st2[i] = g_cs[i];
            for (int i = r; i < NPC; i += 4) st2[PR_STW / 2 + i] = g_rec[i];
            if (r == 0) {
                const double v0 = g_rec[0].x;
                // (a NaN value — only an uploaded state can hold one, smm_set_state — orders under no key: the shard's form is chosen from
                // what every rank knows, never from a shard's own values, so the launch itself says it; the ranks agree on the word at their
                // next rendezvous and replay the step on the per-iteration forms, which resolve or report such a state: include/smmhip.h)
                if (SH && v0 != v0) pr_report(A.err, 3, t0, (int)c0g + cl);
                if constexpr (WIDE) ((uint4*)lds)[cl] = make_uint4((uint32_t)__double2loint(v0), (uint32_t)__double2hiint(v0), (uint32_t)cl, 0u);
                else ((uint2*)lds)[cl] = make_uint2(order_key32(v0), (uint32_t)cl);
                for (int k = 0; k < NP; ++k) Y.s_gth[cl * NP + k] = A.rec_in[(size_t)c * RW + 3 + k];
                if constexpr (PCT) Y.s_thr[cl] = A.mi_g[c0g + (uint32_t)cl];
            }
        }
#pragma unroll
        for (int u = 0; u < PR_ZR; ++u) Y.s_z0[u * 64 + lane] = (lane + u * WG < A.ns) ? A.Z[lane + (size_t)u * WG] : 0.0;
    }
    if (tid >= 64 && tid < 64 + NP) {
        const int k = tid - 64;
        Y.s_const[k] = A.lb[k]; Y.s_const[NP + k] = A.ub[k]; Y.s_const[2 * NP + k] = A.mom[k]; Y.s_const[3 * NP + k] = A.w[k];
    }
    if (tid == 128) {
        Y.s_const[4 * NP] = A.failbox ? A.objp[0] : 1.0; Y.s_const[4 * NP + 1] = A.failbox ? A.objp[1] : 0.0;
        Y.s_const[4 * NP + 2] = (double)A.ns;
        *Y.s_arrived = 0u; *Y.s_minprog = 0; *Y.s_abort = 0u; *Y.s_xmask = 0u; *Y.s_glready = t0 - 2; *Y.s_pub = t0 - 2; *Y.s_hready = t0 - 2;
    }
    if (tid >= 192 && tid < 200) Y.s_ts[tid - 192] = 0ull;
    if (wave == 3 && lane < CONE_HDRW) {
        if (A.walk_first) Y.s_hdr[((t0 - 1) & 3) * 16 + lane] = A.cone_hdr[((size_t)(t0 - 1 - A.plan_t0) * tiles + tile) * CONE_HDRW + lane];
        if (t0 < t1 && exch_on(t0)) Y.s_hdr[(t0 & 3) * 16 + lane] = A.cone_hdr[((size_t)(t0 - A.plan_t0) * tiles + tile) * CONE_HDRW + lane];
    }
    if constexpr (SH) {
        // the launches of the ranks meet: this rank's kernel of launch `epoch` runs, so its launch epoch - 1 has ended and nothing of
        // the ring's last use is being read here any more; nobody stores into anybody's ring before every rank has said so
        if (tile == 0 && wave == 2 && lane < A.G) {
            unsigned char* w = nullptr;
#pragma unroll
            for (int p = 0; p < P2P_MAXG; ++p) w = lane == p ? A.win[p] : w;
            __hip_atomic_store((uint32_t*)(w + A.o_arrive + 128 * (size_t)A.rank), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (wave == 2) {
            const unsigned long long w0 = wall_clock64();
            unsigned spins = 0;
            bool there = lane >= A.G;
            while (__ballot(!there) != 0ull) {
                if (!there) there = (int)(__hip_atomic_load((const uint32_t*)(mine + A.o_arrive + 128 * (size_t)lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) >= 0;
                if ((++spins & 63u) == 0u && (wall_clock64() - w0 > A.tmo)) {
                    if (lane == 0) { pr_report(A.err, 3, t0, (int)c0g); __hip_atomic_store(pr_ctl, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
        }
    }
    PR_BARRIER();
    if (tid == 0 && SH && pr_load4_sys(pr_ctl) == epoch) *Y.s_abort = 1u;
    // the lists of exchange tx by LDS-DMA into their parity's buffers (pairs: waves 8..15, a KB each; gather list: wave 3) ...
    auto request_lists = [&](const int tx) {
        const size_t tb = (size_t)(tx - A.plan_t0) * tiles + tile;
        const uint32_t hw1 = Y.s_hdr[(tx & 3) * 16];
        const int nsub1 = (int)(hw1 & 0xffffu), ngat1 = (int)(hw1 >> 16);
        const uint32_t b = (uint32_t)(tx & 1);
        if (wave >= 8 && 4 * (wave - 8) < nsub1)
            lds_dma16((const uint4*)(A.cone_pairs + tb * (CONE_LEVELS * 64)) + (tid - 512), Y.pbase + b * (CONE_LEVELS * 64 * 4) + (uint32_t)(wave - 8) * 1024u);
        if (wave == 3 && 8 * lane < ngat1)
            lds_dma16((const uint4*)(A.cone_gather + tb * CONE_GCAP) + lane, Y.gbase + b * (CONE_GCAP * 2));
    };
    // ... wave 3, once its gather list has landed: the re-numbering table (population chain -> local number) of plans that name
    // population offsets, and the dummy pair's slots behind the cone's
    auto build_table = [&](const int tx, const int lane) {
        const int ngat = (int)(Y.s_hdr[(tx & 3) * 16] >> 16);
        if (!A.tables_local) {
            uint32_t* tab = (uint32_t*)(lds + Y.tbase) + (tx & 1) * PL_HASH;
            for (int x = lane; x < PL_HASH / 4; x += 64) ((uint4*)tab)[x] = make_uint4(0u, 0u, 0u, 0u);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const uint16_t* gl = (const uint16_t*)(lds + Y.gbase) + (tx & 1) * CONE_GCAP;
            for (int e = lane; e < ngat; e += 64) pl_hash_put(tab, (uint32_t)gl[e], (uint32_t)(CT + e));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    };
    // ... waves 8..15, once their piece of the pair list has landed (and the table stands): local slot offsets, and the tail of
    // every sub-level padded with dummy pairs that never swap
    auto fix_lists = [&](const int tx, const int t_report, const int lane) {
        const uint32_t* hd = Y.s_hdr + (tx & 3) * 16;
        const int nsub1 = (int)(hd[0] & 0xffffu);
        const uint32_t nloc = (uint32_t)CT + (hd[0] >> 16);
        const uint32_t dummy = WIDE ? (16u * nloc) * 0x10001u : (8u * nloc) | ((8u * (nloc + 1u)) << 16);
        uint32_t* pw = (uint32_t*)(lds + Y.pbase) + (tx & 1) * (CONE_LEVELS * 64);
        const uint32_t* tab = (const uint32_t*)(lds + Y.tbase) + (tx & 1) * PL_HASH;
        bool bad = false;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int s = 4 * (wave - 8) + q;
            if (s < nsub1) {
                const uint32_t cnt = (hd[1 + (s >> 2)] >> (8 * (s & 3))) & 0xffu;
                uint32_t w = dummy;
                if ((uint32_t)lane < cnt) {
                    const uint32_t pw0 = pw[s * 64 + lane];
                    uint32_t li, lj;
                    if (A.tables_local) { li = (pw0 & 0xffffu) >> 3; lj = pw0 >> 19; }
                    else {
                        const uint32_t ci = (pw0 & 0xffffu) >> A.unit_sh, cj = (pw0 >> 16) >> A.unit_sh;
                        li = ci - c0g < (uint32_t)CT ? ci - c0g : pl_hash_get(tab, ci);
                        lj = cj - c0g < (uint32_t)CT ? cj - c0g : pl_hash_get(tab, cj);
                        if (li == 0xffffu || lj == 0xffffu) { bad = true; li = nloc; lj = nloc; }
                    }
                    w = (SB * li) | ((SB * lj) << 16);
                }
                pw[s * 64 + lane] = w;
            }
        }
        if (__ballot(bad) != 0ull && lane == 0) pr_report(A.err, 3, t_report, (int)c0g);
    };
    // the cone's initial slots out of ring entry `rel` (tags of launch iteration rel), past the caches: lanes of waves 4..7
    auto gather = [&](const int tx, const int rel, const int t_report, const int tid) {
        const int ngat = (int)(Y.s_hdr[(tx & 3) * 16] >> 16);
        const unsigned long long* rs = (const unsigned long long*)(mine + A.o_slot) + (size_t)(rel & rmask) * (A.Ng + 4);
        const uint32_t want = pr_tag16(epoch, rel) << 16;
        const uint16_t* gl = (const uint16_t*)(lds + Y.gbase) + (tx & 1) * CONE_GCAP;
        const uint4* rr = (const uint4*)(mine + A.o_rec) + (size_t)(rel & rmask) * A.Ng * RW;
        const uint32_t tag = pr_tag32(epoch, rel);
        for (int e = tid - 256; e < ngat; e += 256) {
            const int g = (int)gl[e];
            const uint32_t loc = (uint32_t)(CT + e);
            if constexpr (WIDE || SH) {
                // the record's parameters and its value (the ring's doubles 0 .. NP), self-validating, requested together
                // (a shard's peers push exactly these: the 8-byte slot's key is derived from the value here)
                p2p_u32x4 q0, q1, q2;
                asm volatile("global_load_dwordx4 %0, %3, off sc0 sc1\n\tglobal_load_dwordx4 %1, %4, off sc0 sc1\n\tglobal_load_dwordx4 %2, %5, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                             : "=&v"(q0), "=&v"(q1), "=&v"(q2) : "v"(rr + (size_t)g * RW), "v"(rr + (size_t)g * RW + (NP - 1)), "v"(rr + (size_t)g * RW + NP) : "memory");
                uint4 u0 = make_uint4(q0.x, q0.y, q0.z, q0.w), u1 = make_uint4(q1.x, q1.y, q1.z, q1.w), u2 = make_uint4(q2.x, q2.y, q2.z, q2.w);
                if (__builtin_expect(!(p2p_ll_ok(u0, tag) && p2p_ll_ok(u1, tag) && p2p_ll_ok(u2, tag)), 0)) {
                    const PlGather3 w3 = pl_wait_gather3(W, rr + (size_t)g * RW, rr + (size_t)g * RW + (NP - 1), rr + (size_t)g * RW + NP, tag, t_report, g);
                    u0 = w3.q0; u1 = w3.q1; u2 = w3.q2;
                }
                if constexpr (PCT) Y.s_thr[loc] = A.mi_g[g];
                if constexpr (WIDE) ((uint4*)lds)[loc] = make_uint4(u2.x, u2.z, loc, 0u);
                else ((uint2*)lds)[loc] = make_uint2(order_key32(p2p_ll_double(u2)), loc);
                Y.s_gth[loc * NP] = p2p_ll_double(u0);
                if constexpr (NP > 1) Y.s_gth[loc * NP + 1] = p2p_ll_double(u1);
            } else {
                unsigned long long v;
                p2p_u32x4 q0, q1;
                asm volatile("global_load_dwordx2 %0, %3, off sc0 sc1\n\tglobal_load_dwordx4 %1, %4, off sc0 sc1\n\tglobal_load_dwordx4 %2, %5, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                             : "=&v"(v), "=&v"(q0), "=&v"(q1) : "v"(rs + g), "v"(rr + (size_t)g * RW), "v"(rr + (size_t)g * RW + (NP - 1)) : "memory");
                uint4 u0 = make_uint4(q0.x, q0.y, q0.z, q0.w), u1 = make_uint4(q1.x, q1.y, q1.z, q1.w);
                if (__builtin_expect((((uint32_t)(v >> 32)) & 0xffff0000u) != want || !(p2p_ll_ok(u0, tag) && p2p_ll_ok(u1, tag)), 0)) {
                    const PrGather w3 = pr_wait_gather(W, rs + g, rr + (size_t)g * RW, rr + (size_t)g * RW + (NP - 1), want, tag, t_report, g);
                    v = w3.v; u0 = w3.q0; u1 = w3.q1;
                }
                ((uint2*)lds)[loc] = make_uint2((uint32_t)v, loc);
                Y.s_gth[loc * NP] = p2p_ll_double(u0);
                if constexpr (NP > 1) Y.s_gth[loc * NP + 1] = p2p_ll_double(u1);
            }
        }
        if (tid == 256) {   // the dummy pair's slots behind the cone's: keys 1 < 2 / value 0 on both sides — "no swap"
            const uint32_t nloc = (uint32_t)(CT + ngat);
            if constexpr (PCT) Y.s_thr[nloc] = 0.0;   // (0 - 0 > 0 is false)
            if constexpr (WIDE) ((uint4*)lds)[nloc] = make_uint4(0u, 0u, 0u, 0u);
            else { ((uint2*)lds)[nloc] = make_uint2(1u, 0u); ((uint2*)lds)[nloc + 1u] = make_uint2(2u, 0u); }
        }
    };
    // a chain's walk slot and self-validating record as iteration `rel` of the launch, into every rank's ring (lanes of the control wave)
    auto publish = [&](const int rel, const int c_glob, const int r, const double (&nr)[RW]) {
        const unsigned long long sw = (unsigned long long)order_key32(nr[0]) | ((unsigned long long)((uint32_t)c_glob | (pr_tag16(epoch, rel) << 16)) << 32);
        double ro[8];   // the ring's order
#pragma unroll
        for (int f = 0; f < 8; ++f) ro[f] = 0.0;
#pragma unroll
        for (int f = 0; f < RW; ++f) ro[pr_ring_index<NP>(f)] = nr[f];
        const double2 pv = sel4(r, make_double2(ro[0], ro[1]), make_double2(ro[2], ro[3]), make_double2(ro[4], ro[5]), make_double2(ro[6], ro[7]));
        const uint32_t tag = pr_tag32(epoch, rel);
        const size_t so = (size_t)A.o_slot + ((size_t)(rel & rmask) * (A.Ng + 4) + (size_t)c_glob) * 8;
        const size_t ro_ = (size_t)A.o_rec + ((size_t)(rel & rmask) * A.Ng + (size_t)c_glob) * RW * 16 + (size_t)r * 32;
        if constexpr (SH) {
            // the whole record into this rank's own window; into the peers' the ring's doubles 0 .. NP only (parameters, value):
            // lane r holds doubles 2r and 2r + 1
            constexpr int need = NP + 1;
            if (r < NPC) pr_store_ll(mine + ro_, pv, tag);
#pragma unroll
            for (int p = 0; p < P2P_MAXG; ++p) {
                if (p < A.G && p != A.rank) {
                    unsigned char* w = A.win[p];
                    if (2 * r + 1 < need) pr_store_ll(w + ro_, pv, tag);
                    else if (2 * r < need) pr_store_ll1(w + ro_, pv.x, tag);
                }
            }
            (void)sw; (void)so;
        } else {
            if (!WIDE && r == 0) pr_store8(mine + so, sw);
            if (r < NPC) pr_store_ll(mine + ro_, pv, tag);
        }
    };
    auto announce = [&](const int rel) {   // lane 0 of the control wave: every read of the ring's entry rel - 1 is done
        const uint32_t word = pr_progress_word(epoch, rel);
        const size_t po = (size_t)A.o_progress + 4 * (size_t)((SH ? A.rank * tiles : 0) + tile);
        if constexpr (SH) {
#pragma unroll
            for (int p = 0; p < P2P_MAXG; ++p)
                if (p < A.G) __hip_atomic_store((uint32_t*)(A.win[p] + po), word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else __hip_atomic_store((uint32_t*)(mine + po), word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };

    if (A.walk_first) {
        // the previous kernel — of any form — left the exchange of its last iteration to this one: its lists, and the records the
        // launch starts from published as the launch's iteration 0 (every rank's tiles do: the first walk gathers like any other)
        if (wave == 3 || wave >= 8) {
            request_lists(t0 - 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (wave == 3) build_table(t0 - 1, lane);
        }
        if (wave == 2 && lane == 0 && A.cone_ok[t0 - 1 - A.plan_t0] == 0u) pr_report(A.err, 3, t0, (int)c0g);
        if (wave == 0) {
            const int cl = lane >> 2, r = lane & 3, c = tile * CT + cl;
            if (c < N) {
                double nr[RW];
#pragma unroll
                for (int f = 0; f < RW; ++f) nr[f] = Y.s_st[cl * LW + PR_STW + f];
                publish(0, (int)c0g + cl, r, nr);
            }
        }
        PR_BARRIER();
        if (wave >= 8) fix_lists(t0 - 1, t0, lane);
        if (wave >= 4 && wave < 8) gather(t0 - 1, 0, t0, tid);
    }

    if (wave != 0) {
        // =====================================================================================================================
        // the WORKER waves 1..15: simulation (all), randomness (1), progress (2), gather list + table + history (3), gather (4..7), pair lists (8..15)
        // =====================================================================================================================
        const int h = wave >> 3, wih = wave & 7;
        const bool simw = h < NP;
        const bool rng_here = A.rb == nullptr;
        double z[PR_ZR];
        int nfull = 0;
        bool extra = false;
        if (simw) {
            const int l = wih * 64 + lane;
            const double* zr = A.Z + (size_t)h * A.zstride + l;
#pragma unroll
            for (int u = 0; u < PR_ZR; ++u) z[u] = (l + u * WG < A.ns) ? zr[(size_t)u * WG] : 0.0;
            nfull = A.ns / WG;
            extra = l < A.ns - nfull * WG;
        } else {
#pragma unroll
            for (int u = 0; u < PR_ZR; ++u) z[u] = 0.0;
        }
        auto make_rng = [&](const int tn) {
            const int c1 = tile * CT + (lane >> 2);
            if (c1 >= N) return;
            const uint32_t g1 = c0g + (uint32_t)(lane >> 2);
            double* o = Y.s_rng + ((tn & 1) * 64 + lane) * RNGW;
            if (rng_here) {
                o[0] = rng_u(A.seed, g1, (uint32_t)tn);                               // probs_acc[iter], AlgoBGP.jl:85
#pragma unroll
                for (int q = 0; 2 * q < NP; ++q) {                                    // rand(RAND, d) of try r, :404
                    double z0, z1;
                    rng_prop_normal2(A.seed, g1, (uint32_t)tn, (uint32_t)(lane & 3), (uint32_t)q, z0, z1);
                    o[1 + 2 * q] = z0;
                    if (2 * q + 1 < NP) o[1 + 2 * q + 1] = z1;
                }
            } else {
                const double* g_rb = A.rb + ((size_t)(tn - A.rb_t0) * N + c1) * A.RBW;
                const int rr = lane & 3;
                o[0] = g_rb[0];
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    o[1 + k] = rr < A.rb_tries ? g_rb[1 + rr * NP + k] : 0.0;
                    o[1 + NP + k] = NORM_NR + rr < A.rb_tries ? g_rb[1 + (NORM_NR + rr) * NP + k] : 0.0;
                }
            }
        };
        auto store_rows = [&](const double* s_row, const int trow, const unsigned mask) {
            for (int e = lane; e < CT * NPH; e += 64) {
                const int cl = e / NPH, i = e - cl * NPH, c = tile * CT + cl;
                if (c < N && ((mask >> cl) & 1u)) ((double2*)(A.hrec + ((size_t)(trow - 1) * N + c) * HW))[i] = ((const double2*)(s_row + cl * HW))[i];
            }
        };
        if (wave == 1) make_rng(t0);
        for (int t = t0; t <= t1; ++t) {
            const int rel = t - t0 + 1;
            PR_BARRIER();   // BA
            uint32_t nhdr = 0u;
            const bool want_hdr = wave == 3 && lane < CONE_HDRW && t + 1 < t1 && exch_on(t + 1);
            if (wave == 3) {
                if (t > t0) {
                    store_rows(Y.s_hrow, t - 1, 0xffffu);
                    store_rows(Y.s_xrow, t - 2, *Y.s_xmask);
                }
                if (want_hdr) nhdr = A.cone_hdr[((size_t)(t + 1 - A.plan_t0) * tiles + tile) * CONE_HDRW + lane];
            }
            if (t < t1 && exch_on(t)) request_lists(t);
            PR_BARRIER();   // BB
            if (simw) {
                if (nfull == PR_ZR - 1) persist_simulate<NP, true>(z, nfull, extra, Y.s_theta, Y.s_part, h, wih);
                else persist_simulate<NP, false>(z, nfull, extra, Y.s_theta, Y.s_part, h, wih);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                const int lane3 = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
                if (lane3 == 0) __hip_atomic_fetch_add(Y.s_arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            const bool lists = t < t1 && exch_on(t);
            if (wave == 3 || wave >= 8) {
                if (wave == 3 && want_hdr) Y.s_hdr[((t + 1) & 3) * 16 + lane] = nhdr;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA has landed
                if (wave == 3 && lane == 0) __hip_atomic_store(Y.s_glready, t, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (wave == 3 && lists) {
                    build_table(t, lane);
                    if (lane == 0) __hip_atomic_store(Y.s_hready, t, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                if (wave >= 8 && lists) {
                    if (!A.tables_local) while (__hip_atomic_load(Y.s_hready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != t) __builtin_amdgcn_s_sleep(1);
                    fix_lists(t, t + 1, lane);
                }
            }
            if (wave == 1 && t < t1) make_rng(t + 1);
            if (wave == 2) {
                const int m = pl_min_progress(pr_progress, epoch, tiles_all, lane);
                if (lane == 0) {
                    __hip_atomic_store(Y.s_minprog, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (pr_load4_sys(pr_ctl) == epoch) *Y.s_abort = 1u;
                    if (lists && A.cone_ok[t - A.plan_t0] == 0u) pr_report(A.err, 3, t + 1, (int)c0g);
                }
                if constexpr (SH) {   // somebody of this rank gave up: the other ranks need not wait out their time
                    if (*Y.s_abort && lane < A.G) {
                        unsigned char* w = nullptr;
#pragma unroll
                        for (int p = 0; p < P2P_MAXG; ++p) w = lane == p ? A.win[p] : w;
                        __hip_atomic_store((uint32_t*)(w + A.o_ctl), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                }
            }
            if (wave >= 4 && wave < 8 && lists) {
                while (__hip_atomic_load(Y.s_glready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != t) __builtin_amdgcn_s_sleep(1);
                while (__hip_atomic_load(Y.s_pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != t) __builtin_amdgcn_s_sleep(1);
                __builtin_amdgcn_s_sleep(PR_GATHER_DELAY);
                gather(t, rel, t + 1, wave * 64 + lane);
            }
        }
        PR_BARRIER();   // the last epilogue is done
        if (wave == 3) { store_rows(Y.s_hrow, t1, 0xffffu); store_rows(Y.s_xrow, t1 - 1, *Y.s_xmask); }
        if (wave == 1) {   // the result blocks where the next launch (of any form) expects them
            const int cl = lane >> 2, r = lane & 3, c = tile * CT + cl;
            if (c < N) {
                const double2* st2 = (const double2*)(Y.s_st + cl * LW);
                double2* g_cs = (double2*)(A.cs + (size_t)c * CSW);
                for (int i = r; i < 6; i += 4) g_cs[i] = i == 2 ? make_double2(st2[2].x, 0.0) : st2[i];
                for (int i = r; i < NPC; i += 4) ((double2*)(A.rec_out + (size_t)c * RW))[i] = st2[PR_STW / 2 + i];
                if (r == 0) {
                    const double v = Y.s_st[cl * LW + PR_STW];
                    A.vals_out[c] = v;
                    if (A.slot8_out) { A.slot8_out[c] = make_uint2(order_key32(v), (uint32_t)c); if (v != v) atomicOr(A.walk_flags, 1u); }
                }
            }
        }
        return;
    }

    // =========================================================================================================================
    // the CONTROL wave (smm_chain_persist.hpp): four lanes per chain; between two publications: the walk, the proposal, its share of
    // the simulation, the objective, the accept step — everything else behind the publication
    // =========================================================================================================================
    const int nfull0 = A.ns / WG;
    for (int t = t0; t <= t1; ++t) {
        const int rel = t - t0 + 1;
        const bool exch = t == t0 ? A.walk_first != 0 : exch_on(t - 1);
        PR_BARRIER();   // BA
        {
            const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            const int cl = lane >> 2, r = lane & 3;
            const int c = tile * CT + cl;
            const bool valid = c < N;
            double* st = Y.s_st + cl * LW;
            unsigned long long ts1 = 0;
            if (A.ts && lane == 0) { ts1 = wall_clock64(); if (t != t0) Y.s_ts[0] += ts1 - Y.s_ts[7]; }
            // ---- the walk over the cone's sub-levels, on local slots: this wave alone, no barriers ----
            uint32_t src = (uint32_t)cl;    // local number of the chain whose record this chain continues from
            int partner = 0;                // 1 + the partner's number in the population
            const uint32_t lbase = Y.pbase + (uint32_t)((t - 1) & 1) * (CONE_LEVELS * 64 * 4);
            const uint16_t* gl = (const uint16_t*)(lds + Y.gbase) + ((t - 1) & 1) * CONE_GCAP;
            if (exch) {
                const int nsub = (int)(Y.s_hdr[((t - 1) & 3) * 16] & 0xffffu);
                if constexpr (WIDE) {
                    lean_walk_levels<64, 0, true, WalkNoGuard, PCT>(nullptr, 1, lbase, (uint32_t)(64 * lane), nsub, lane, 0, A.thr, WalkNoGuard(),
                                                                    (uint32_t)((unsigned char*)Y.s_thr - lds));
                } else {
                    const PersistLocWalkValues values{W, (const uint4*)(mine + A.o_rec) + (size_t)((rel - 1) & rmask) * A.Ng * RW, gl, c0g, pr_tag32(epoch, rel - 1), RW, NP, t};
                    lean_walk_levels<64, 0, false, PersistLocWalkValues>(nullptr, 1, lbase, (uint32_t)(64 * lane), nsub, lane, 0, 0.0, values);
                }
                if (valid) {
                    const uint32_t kmeta = WIDE ? ((const uint4*)lds)[cl].z : ((const uint2*)lds)[cl].y;
                    src = kmeta & 0xffffu;
                    if (kmeta >> 16) {   // set_exchanged!, :747-748
                        const uint32_t pl = lean_partner<0, WIDE ? 4 : 3>(lds, lbase, kmeta, (uint32_t)cl) - 1u;
                        partner = 1 + (int)(pl < (uint32_t)CT ? c0g + pl : (uint32_t)gl[pl - CT]);
                    }
                }
            }
            const bool donor = valid && src != (uint32_t)cl;
            const uint32_t src_g = src < (uint32_t)CT ? c0g + src : (uint32_t)gl[src - CT];
            // the donor's whole record (swap_ev_ij!, :734-749), requested now and looked at behind the simulation (LDS-DMA, past the caches)
            if (donor) {
                const unsigned char* dwin = mine;
                if constexpr (SH) {   // (the donor's OWNER holds the whole record: equal shards of A.N chains)
                    const int owner = (int)src_g / A.N;
#pragma unroll
                    for (int p = 0; p < P2P_MAXG; ++p) dwin = owner == p ? A.win[p] : dwin;
                }
                const uint4* g_ll = (const uint4*)(dwin + A.o_rec) + ((size_t)((rel - 1) & rmask) * A.Ng + src_g) * RW;
                const uint32_t dbase = (uint32_t)((unsigned char*)Y.s_donor - lds);
                pr_dma16(g_ll + r, dbase);
                if (4 + r < RW) pr_dma16(g_ll + 4 + r, dbase + 64 * 16);
            }
            unsigned long long ts2 = 0;
            if (A.ts && lane == 0) ts2 = wall_clock64();
            // ---- proposal: lane r evaluates try r; the chain's first try inside the unit box wins (mysample, :400-410) ----
            double mu01[NP], th[NP], th_old[NP];
            const double sigma = st[CS_SIGMA];
            const double* o = Y.s_rng + ((t & 1) * 64 + lane) * RNGW;
            bool found = !valid;
            {
                double x[NP];
                bool ok = valid;
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    const double lbk = Y.s_const[k], ubk = Y.s_const[NP + k];
                    th_old[k] = donor ? Y.s_gth[src * NP + k] : st[PR_STW + 3 + k];
                    th[k] = valid ? th_old[k] : 0.0;
                    mu01[k] = (th_old[k] - lbk) / (ubk - lbk);   // mapto_01, mprob.jl:248
                    const double step = sigma * o[1 + k];          // MvNormal(mu01, sigma): x = mu + sigma*z
                    x[k] = mu01[k] + step;
                    if (!(x[k] >= 0.0 && x[k] <= 1.0)) ok = false;   // inclusive bounds, :405
                }
                if (r >= A.smpl_iters || (A.user_n && r >= A.rb_tries)) ok = false;
                const unsigned long long m = __ballot(ok);
                const unsigned quad = (unsigned)(m >> (lane & ~3)) & 0xfu;
                if (quad) {
                    const int rwin = __builtin_ctz(quad);
#pragma unroll
                    for (int k = 0; k < NP; ++k) {
                        const double lbk = Y.s_const[k];
                        const double sc = x[k] * (Y.s_const[NP + k] - lbk);
                        const double thk = sc + lbk;   // mapto_ab, mprob.jl:271
                        th[k] = quad_bcast_dyn(thk, lane, rwin);
                    }
                    found = true;
                }
            }
            if (__builtin_expect(__any(!found), 0)) {
                const PrTries TR{A.rb, A.seed, A.rb_t0, A.N, A.RBW, A.rb_tries, A.user_n, A.smpl_iters, A.offset};
                const PrTh<NP> lt = persist_late_tries<NP>(TR, o, t, c, valid, lane, found, sigma, mu01[0], mu01[NP - 1], Y.s_const, th[0], th[NP - 1]);
#pragma unroll
                for (int k = 0; k < NP; ++k) th[k] = lt.th[k];
                found = lt.found;
                if (!found && r == 0) pr_report(A.err, ERRK_NO_DRAW, t, (int)c0g + cl);   // :409
            }
            if (r == 0) {
#pragma unroll
                for (int k = 0; k < NP; ++k) Y.s_theta[cl * NP + k] = th[k];
                st[CS_PARTNER] = (double)partner;
                st[CS_WASX] = (double)src;   // (free during the launch: the LOCAL number the record comes from)
            }
            if (A.ts && lane == 0) {
                const unsigned long long ts4 = wall_clock64();
                Y.s_ts[1] += ts2 - ts1; Y.s_ts[3] += ts4 - ts2; Y.s_ts[6] = ts4;
            }
        }
        PR_BARRIER();   // BB
        {
            if (nfull0 == PR_ZR - 1) persist_simulate_lds<NP, true>(Y.s_z0, A.ns, nfull0, Y.s_theta, Y.s_part);
            else persist_simulate_lds<NP, false>(Y.s_z0, A.ns, nfull0, Y.s_theta, Y.s_part);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            const int lane_a = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            if (lane_a == 0) __hip_atomic_fetch_add(Y.s_arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        {
            const unsigned want = (unsigned)(8 * NP) * (unsigned)rel;
            while (__hip_atomic_load(Y.s_arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < want) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        const int cl = lane >> 2, r = lane & 3;
        const int c = tile * CT + cl;
        const bool valid = c < N;
        double* st = Y.s_st + cl * LW;
        unsigned long long ts5 = 0;
        if (A.ts && lane == 0) ts5 = wall_clock64();
        if (valid) {
            // ---- the record the chain continues from (classic order: value, prob, status, theta, sim_moments) ----
            const int src = (int)st[CS_WASX];
            const bool donor = src != cl;
            double rc2[RW];
#pragma unroll
            for (int f = 0; f < RW; ++f) rc2[f] = st[PR_STW + f];
            if (donor) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA has landed
                const uint32_t tag = pr_tag32(epoch, rel - 1);
                double rr[8];
#pragma unroll
                for (int f = 0; f < 8; ++f) rr[f] = 0.0;
                bool ok = true;
#pragma unroll
                for (int f = 0; f < RW; ++f) {
                    const uint4 q = Y.s_donor[(f >> 2) * 64 + 4 * cl + (f & 3)];
                    ok = ok && p2p_ll_ok(q, tag);
                    rr[f] = p2p_ll_double(q);
                }
                if (__builtin_expect(!ok, 0)) {   // (the gather validated the parameters only: the other pieces of the same publication may still be on their way)
                    const uint16_t* gl = (const uint16_t*)(lds + Y.gbase) + ((t - 1) & 1) * CONE_GCAP;
                    const uint32_t src_g = (uint32_t)src < (uint32_t)CT ? c0g + (uint32_t)src : (uint32_t)gl[src - CT];
                    const unsigned char* dwin = mine;
                    if constexpr (SH) {
                        const int owner = (int)src_g / A.N;
#pragma unroll
                        for (int p = 0; p < P2P_MAXG; ++p) dwin = owner == p ? A.win[p] : dwin;
                    }
                    const uint4* g_ll = (const uint4*)(dwin + A.o_rec) + ((size_t)((rel - 1) & rmask) * A.Ng + src_g) * RW;
#pragma unroll
                    for (int f = 0; f < RW; f += 2) {
                        const PrLL2 w2 = pr_wait_ll2(W, g_ll + f, g_ll + f + 1, tag, t, (int)c0g + cl);
                        rr[f] = p2p_ll_double(w2.q0); rr[f + 1] = p2p_ll_double(w2.q1);
                    }
                }
#pragma unroll
                for (int f = 0; f < RW; ++f) rc2[f] = rr[pr_ring_index<NP>(f)];
            }
            if (lane == 0) announce(rel);   // every read of the ring's last entry is done
            // ---- objective value (ObjExamples.jl:79-110), doAcceptReject! (:324-392) ----
            const double atun = st[CS_ATUN];
            const double uu = Y.s_rng[((t & 1) * 64 + lane) * RNGW];
            double th2[NP], sm[NP];
#pragma unroll
            for (int k = 0; k < NP; ++k) th2[k] = Y.s_theta[cl * NP + k];
            double value;
            int status;
            if (th2[0] >= Y.s_const[4 * NP] && th2[0] <= Y.s_const[4 * NP + 1]) {   // NORM_FAILBOX's "exception": mprob.jl:183-186
#pragma unroll
                for (int k = 0; k < NP; ++k) sm[k] = NAN;
                value = -1.0;   // Eval() default, Eval.jl:84
                status = -2;
            } else {
                double mk = 0.0, vk = 0.0;
                if (r < NP) {
                    double tot = Y.s_part[(r * 8 + 0) * CT + cl];
#pragma unroll
                    for (int wv = 1; wv < 8; ++wv) tot = tot + Y.s_part[(r * 8 + wv) * CT + cl];
                    mk = tot / Y.s_const[4 * NP + 2];
                    double d = mk - Y.s_const[2 * NP + r];
                    const double wk = Y.s_const[3 * NP + r];
                    if (!isnan(wk)) d = d / wk;
                    vk = d * d;
                }
                double vsum = 0.0;
                {
                    const double m0 = quad_bcast<0>(mk), v0 = quad_bcast<0>(vk);
                    sm[0] = m0; vsum = v0;
                    if constexpr (NP > 1) { const double m1 = quad_bcast<1>(mk), v1 = quad_bcast<1>(vk); sm[1] = m1; vsum = vsum + v1; }
                }
                value = vsum / (double)NP;
                status = 1;
            }
            const double old = rc2[0];
            double prob;
            bool acc;
            if (status < 0) {   // :336-338
                prob = 0.0; acc = false;
            } else {
                if (!(value >= 0.0) && r == 0) pr_report(A.err, ERRK_NEGATIVE, t, (int)c0g + cl);   // :341
                const double e = pr_exp(atun * (old - value));
                prob = (e != e) ? e : (e < 1.0 ? e : 1.0);   // minimum([1.0,e]), NaN propagates (:344)
                if (!isfinite(prob)) { prob = 0.0; acc = false; status = -1; }   // :350-353
                else if (!isfinite(old)) { prob = 1.0; acc = true; }             // :355-359
                else { status = 1; acc = prob > uu; }                            // strict >, :362-367
            }
            const double accd = acc ? 1.0 : 0.0;
            const double v = acc ? value : old;
            double nr[RW];   // the chain's last accepted record (lastAccepted :209-215) = input of the exchange step
            nr[0] = v; nr[1] = acc ? prob : rc2[1]; nr[2] = acc ? (double)status : rc2[2];
#pragma unroll
            for (int k = 0; k < NP; ++k) { nr[3 + k] = acc ? th2[k] : rc2[3 + k]; nr[3 + NP + k] = acc ? sm[k] : rc2[3 + NP + k]; }
            if (RW > 3 + 2 * NP) nr[RW - 1] = 0.0;
            // ---- publish: the walk slot and the self-validating record of iteration t into the ring(s) ----
            if (t < t1) {
                if (__builtin_expect(rel > rmask && __hip_atomic_load(Y.s_minprog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < rel - rmask, 0))
                    pl_wait_progress(W, pr_progress, Y.s_minprog, rel - rmask, tiles_all, lane, t, (int)c0g + cl);
#ifdef SMM_TEST_HOOKS
                if (tile == A.slow_tile) { const unsigned long long w0 = wall_clock64(); while (wall_clock64() - w0 < (unsigned long long)A.slow_ticks) __builtin_amdgcn_s_sleep(8); }
#endif
                publish(rel, (int)c0g + cl, r, nr);
                if (lane == 0) __hip_atomic_store(Y.s_pub, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (A.ts && lane == 0) { const unsigned long long ts6 = wall_clock64(); Y.s_ts[4] += ts5 - Y.s_ts[6]; Y.s_ts[5] += ts6 - ts5; Y.s_ts[6] = ts6; }
            // ================= behind the publication =================
            const int partner = (int)st[CS_PARTNER];
            const double sig = st[CS_SIGMA];
            int nn = (int)st[CS_NNOEX], na = (int)st[CS_NACC];
            double bp = st[CS_BEST], bpid = st[CS_BESTID];
            if (partner != 0) {
                // set_eval!(ci, ej) of swap_ev_ij! as a history record (:231-243)
                const double dv = rc2[0];
                if (dv < st[CS_BESTP]) { bp = dv; bpid = (double)(t - 1); }
                else { bp = st[CS_BESTP]; bpid = st[CS_BESTPID]; }
                if (r == 0) {
                    double* hx = Y.s_xrow + cl * HW;
                    hx[H_VALUE] = dv; hx[H_PROB] = rc2[1]; hx[H_CURR] = dv; hx[H_BEST] = bp; hx[H_BESTID] = bpid;
                    hx[H_EXCH] = (double)partner; hx[H_ACC] = 1.0; hx[H_STATUS] = rc2[2];
#pragma unroll
                    for (int k = 0; k < 2 * NP; ++k) hx[H_PARAMS + k] = rc2[3 + k];
                    if (HW > H_PARAMS + 2 * NP) hx[HW - 1] = 0.0;
                }
            } else { nn += 1; na += (int)st[CS_LACC]; }   // set_acceptRate!, :253-257 (exchanged iterations do not count)
            double nsig = sig;
            const bool upd = (t % A.sigma_update_steps) == 0;
            if (upd || t == t1) {
                const double rate = (double)(na + (acc ? 1 : 0)) / (double)(nn + 1);   // set_acceptRate!, :253-257
                if (upd) nsig = (rate > 0.234) ? sig * (1.0 + A.sigma_adjust_by) : sig * (1.0 - A.sigma_adjust_by);   // :381-390
                if (r == 0) st[CS_RATE] = rate;
            }
            double bestv, bestid;
            const double currv = acc ? value : old;
            if (value < bp) { bestv = value; bestid = (double)t; }
            else { bestv = bp; bestid = bpid; }
            if (r == 0) {
                // the tile's own slots and parameters of the next walk (a donor may be a chain of the same tile)
                if constexpr (WIDE) ((uint4*)lds)[cl] = make_uint4((uint32_t)__double2loint(v), (uint32_t)__double2hiint(v), (uint32_t)cl, 0u);
                else ((uint2*)lds)[cl] = make_uint2(order_key32(v), (uint32_t)cl);
#pragma unroll
                for (int k = 0; k < NP; ++k) Y.s_gth[cl * NP + k] = nr[3 + k];
                st[CS_SIGMA] = nsig; st[CS_NNOEX] = (double)nn; st[CS_NACC] = (double)na; st[CS_LACC] = accd;
                st[CS_BEST] = bestv; st[CS_BESTID] = bestid; st[CS_BESTP] = bp; st[CS_BESTPID] = bpid;
#pragma unroll
                for (int f = 0; f < RW; ++f) st[PR_STW + f] = nr[f];
                double* hv = Y.s_hrow + cl * HW;
                hv[H_VALUE] = value; hv[H_PROB] = prob; hv[H_CURR] = currv; hv[H_BEST] = bestv; hv[H_BESTID] = bestid;
                hv[H_EXCH] = 0.0; hv[H_ACC] = accd; hv[H_STATUS] = (double)status;
#pragma unroll
                for (int k = 0; k < NP; ++k) { hv[H_PARAMS + k] = th2[k]; hv[H_PARAMS + NP + k] = sm[k]; }
                if (HW > H_PARAMS + 2 * NP) hv[HW - 1] = 0.0;
            }
        }
        {
            const unsigned long long xm = __ballot(valid && r == 0 && (int)st[CS_PARTNER] != 0);
            if (lane == 0) {
                unsigned m = 0u;
#pragma unroll
                for (int q = 0; q < CT; ++q) m |= (unsigned)((xm >> (4 * q)) & 1ull) << q;
                *Y.s_xmask = m;
            }
        }
        if (A.ts && lane == 0) { const unsigned long long ts7 = wall_clock64(); Y.s_ts[2] += ts7 - Y.s_ts[6]; Y.s_ts[7] = ts7; }
    }
    PR_BARRIER();
    {
        const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        if (A.ts && lane < 7) A.ts[(size_t)tile * 8 + lane] = lane < 6 ? Y.s_ts[lane] : (unsigned long long)(t1 - t0 + 1);
    }
}
