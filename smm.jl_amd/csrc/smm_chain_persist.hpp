// what the PERSISTENT chain kernels share — the ring's tags, stores and loads, the out-of-line waits, the simulation with the lane's
// shocks in registers, mysample's late tries — part of libsmmhip (included by smmhip.hip inside its anonymous namespace; gfx950 device
// code).  The kernels: smm_chain_persist_loc.hpp (objfunc_norm, locally numbered cones: single shards and shards of a sharded run,
// thresholds) and smm_chain_persist_gen.hpp (objectives without a simulation).
#pragma once
// ------------------------------------------------------------------------------------------
// ONE launch for a whole run of iterations (up to the end of the look-ahead windows): next_eval for every chain and exchangeMoves!
// between two iterations (AlgoBGP.jl:589-640, 647-716) without a kernel boundary in between.  Results are bit-identical to the
// per-iteration kernels' (same numerical contract, same arithmetic, same order of everything that has an order).
//
// What a kernel boundary per iteration cost the headline configuration (4096 chains, one 1024-lane tile of 16 chains per CU): the
// boundary itself (~1.5 us), every workgroup re-staging all 4096 walk slots and the whole pair list (~1.6 us), the walk over all
// 13 levels with barriers (~2.6 us), the chain state and 160 KB of shocks re-fetched per launch — 7.3 of 14.4 us next to 7.4 us
// of simulation.  In the persistent kernels, per workgroup (= tile = CU) and iteration:
//   * the chain state, the last accepted records and the lane's SHOCKS (20 doubles: draws l, l + 512, ... of the half's moment)
//     stay in LDS / registers for the whole launch; the simulation issues no load at all;
//   * the tile waits only for ITS CONE of the exchange (the ~100 pairs / ~100 chains of other tiles its 16 chains' outcome depends on,
//     listed ahead of time by the plan kernels): after its accept step a tile publishes, per chain, one 8-byte walk slot
//     {order_key32(value), chain | tag << 16} and the self-validating record (smm_p2p.hpp's LL granules) into a ring of PR_K
//     iterations, with write-through stores; the next iteration's prologue gathers the cone's slots past the caches (one lane per
//     chain, waves 4..7), looks again where a tag is still the old one, lets wave 0 walk the cone's sub-levels alone, and fetches a
//     donor's record only for a chain that was exchanged;
//   * the next iteration's lists arrive by LDS-DMA under the simulation, its randomness is drawn by wave 1 behind its share of the
//     simulation, and the gathering waves start looking for the other tiles' slots as soon as this tile has published;
//   * the waves are ROLE-SPECIALISED (two loops in one kernel, the same number of workgroup barriers in each): wave 0 is the control
//     wave (its shocks come out of LDS for its share of the simulation); history rows leave through LDS and are stored by a worker
//     wave; what happens once per launch or almost never is out of line.
// Nobody waits for an acknowledgement, no grid barrier, no atomics on the way.  Why the ring cannot be overrun: a tile publishes
// iteration i into entry i mod PR_K only after every tile has announced (progress word) that it has finished the prologue reads of
// iteration i - PR_K + 1 — never failing in practice, holding under any skew (tests: a delayed workgroup, a ring of 2).  Deadlock
// freedom: the slowest tile waits only for publications of iterations that every other tile has passed already, and those stay in
// the ring until it has read them.  All tiles must be RESIDENT (one per CU: the host checks the occupancy against the grid); every
// spin has a time-out, and a tile that gives up raises the launch's abort word so that nobody else waits out theirs.
//
// Errors.  A hard error (AlgoBGP.jl:341,409) is reported like everywhere; tiles run on to the end of the launch (a tile that stopped
// would starve the others), and the HOST, when it finds the error word set after such launches, restores the state it saved before
// the first of them and repeats those iterations on the one-launch-per-iteration path, which stops at the failing iteration with
// the library's documented state (smmhip.hip, persist_repair).
// ------------------------------------------------------------------------------------------
constexpr int PR_K = 8;          // iterations in the ring
constexpr int PR_ZR = 20;        // shocks per lane held in registers: ns <= 512 * PR_ZR
constexpr int PR_MAX_ITERS = 4000;   // iterations per launch (12 bits of the tags count them)
constexpr int PR_GATHER_DELAY = 8;   // s_sleep units (64 clocks) between this tile's publication and the gather's first look
constexpr int PR_STW = 12;       // doubles of chain state in front of the record in a tile's LDS line
constexpr unsigned long long PERSIST_TMO_FIRST = 40000000ull;   // 0.4 s of the 100 MHz wall clock: the spins of a context's first launches (smmhip.hip, launch_chain_persist)

__host__ __device__ inline int persist_line(int np) { return PR_STW + ((3 + 2 * np + 1) & ~1); }
__host__ __device__ inline size_t persist_ring_slot_bytes(int Ng) { return (size_t)PR_K * (((size_t)Ng + 4) * 8); }
__host__ __device__ inline size_t persist_ring_rec_bytes(int Ng, int RW) { return (size_t)PR_K * (size_t)Ng * RW * 16; }

// tags: never 0 (the ring starts zeroed and is zeroed again whenever the 7 epoch bits of the slot tag wrap)
__device__ inline uint32_t pr_tag16(const uint32_t epoch, const int rel) { return 0x8000u | ((epoch & 0x7fu) << 8) | ((uint32_t)rel & 0xffu); }
__device__ inline uint32_t pr_tag32(const uint32_t epoch, const int rel) { return 0x80000000u | ((epoch & 0x7ffffu) << 12) | ((uint32_t)rel & 0xfffu); }
__device__ inline uint32_t pr_progress_word(const uint32_t epoch, const int rel) { return (epoch << 12) | (uint32_t)rel; }

// The ring is written with write-through stores and read past the caches.  Scope: for the tiles of ONE device the agent scope (sc1) would
// do, but the same words also travel between the ranks' windows (smm_chain_persist_loc.hpp): the system scope (sc0 sc1) for all — measured
// equal on one device (EXPERIMENTS.md R4.3).
#define PR_SC "sc0 sc1"
__device__ inline unsigned long long pr_load8_sys(const void* p) {
    unsigned long long v;
    asm volatile("global_load_dwordx2 %0, %1, off " PR_SC "\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ inline uint32_t pr_load4_sys(const void* p) { return __hip_atomic_load((const uint32_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ inline uint4 pr_load16_sys(const void* p) {
    p2p_u32x4 q;
    asm volatile("global_load_dwordx4 %0, %1, off " PR_SC "\n\ts_waitcnt vmcnt(0)" : "=v"(q) : "v"(p) : "memory");
    return make_uint4(q.x, q.y, q.z, q.w);
}
__device__ inline void pr_load16x2_sys(const void* p0, const void* p1, uint4& a, uint4& b) {
    p2p_u32x4 q0, q1;
    asm volatile("global_load_dwordx4 %0, %2, off " PR_SC "\n\tglobal_load_dwordx4 %1, %3, off " PR_SC "\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(q0), "=&v"(q1) : "v"(p0), "v"(p1) : "memory");
    a = make_uint4(q0.x, q0.y, q0.z, q0.w); b = make_uint4(q1.x, q1.y, q1.z, q1.w);
}
__device__ inline void pr_store8(void* p, const unsigned long long v) {
    asm volatile("global_store_dwordx2 %0, %1, off " PR_SC :: "v"(p), "v"(v) : "memory");
}
// one LDS-DMA instruction past the caches: 16 bytes per active lane from gsrc (per lane) to LDS byte address lds_dst (wave-uniform) + 16 * lane
__device__ inline void pr_dma16(const void* gsrc, const uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off " PR_SC "\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// 16 bytes of payload as 32 self-validating bytes at p (smm_p2p.hpp's granules; s_nop: the store-data hazard of inline asm stores)
__device__ inline void pr_store_ll(void* p, const double2 v, const uint32_t tag) {
    const unsigned long long a = __builtin_bit_cast(unsigned long long, v.x), b = __builtin_bit_cast(unsigned long long, v.y);
    const p2p_u32x4 q0 = {(unsigned)a, tag, (unsigned)(a >> 32), tag}, q1 = {(unsigned)b, tag, (unsigned)(b >> 32), tag};
    asm volatile("global_store_dwordx4 %0, %1, off " PR_SC "\n\tglobal_store_dwordx4 %2, %3, off " PR_SC "\n\ts_nop 1"
                 :: "v"(p), "v"(q0), "v"((unsigned char*)p + 16), "v"(q1) : "memory");
}

// ... one granule: 8 bytes of payload as 16 self-validating bytes
__device__ inline void pr_store_ll1(void* p, const double v, const uint32_t tag) {
    const unsigned long long a = __builtin_bit_cast(unsigned long long, v);
    const p2p_u32x4 q0 = {(unsigned)a, tag, (unsigned)(a >> 32), tag};
    asm volatile("global_store_dwordx4 %0, %1, off " PR_SC "\n\ts_nop 1" :: "v"(p), "v"(q0) : "memory");
}

// The ring holds a record's doubles in its own order — parameters first, so that the gather fetches them together with the walk
// slot: theta[NP], value, prob, status, sim_moments[NP], pad — one uint4 {lo, tag, hi, tag} per double.
template <int NP> __device__ constexpr int pr_ring_index(const int classic) {   // classic: value, prob, status, theta[NP], sim_moments[NP], pad
    return classic < 3 ? NP + classic : classic < 3 + NP ? classic - 3 : classic;
}

// (what the out-of-line helpers need travels BY VALUE: a reference to the kernel's argument block would make the compiler copy the
// whole block into scratch memory and read every field from there)
struct PrWait { unsigned long long* err; uint32_t* pr_ctl; unsigned* s_abort; uint32_t epoch; unsigned long long tmo; /* ticks a spin may last */ };
__device__ inline void pr_report(unsigned long long* err, int kind, int t, int chain) {
    const unsigned long long key = ((unsigned long long)t << 34) | ((unsigned long long)chain << 2) | (unsigned)kind;
    atomicMin(err, key);
}
__device__ inline bool pr_give_up(const PrWait W, const unsigned long long w0) {   // time-out, or another tile has given up
    return wall_clock64() - w0 > W.tmo || pr_load4_sys(W.pr_ctl) == W.epoch;
}
__device__ inline void pr_abort(const PrWait W, int t, int chain) {
    pr_report(W.err, 3, t, chain);
    __hip_atomic_store(W.pr_ctl, W.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    *W.s_abort = 1u;
}
// out of line: a tagged slot that is not there yet (its stores are on their way) — look again, past the caches
__device__ __attribute__((noinline)) unsigned long long pr_wait_slot(const PrWait W, const unsigned long long* p, const uint32_t want, const int t, const int g) {
    unsigned long long v = 0ull;
    if (*W.s_abort) return v;
    const unsigned long long w0 = wall_clock64();
    unsigned spins = 0;
    do {
        __builtin_amdgcn_s_sleep(1);
        v = pr_load8_sys(p);
        if ((++spins & 63u) == 0u && pr_give_up(W, w0)) { pr_abort(W, t, g); break; }
    } while ((((uint32_t)(v >> 32)) & 0xffff0000u) != want);
    return v;
}
// ... a slot and two self-validating pieces of the same chain's record (the gather): all three are looked at again together
struct PrGather { unsigned long long v; uint4 q0, q1; };
__device__ __attribute__((noinline)) PrGather pr_wait_gather(const PrWait W, const unsigned long long* ps, const uint4* p0, const uint4* p1, const uint32_t want,
                                                             const uint32_t tag, const int t, const int g) {
    PrGather o;
    o.v = 0ull; o.q0 = make_uint4(0u, 0u, 0u, 0u); o.q1 = o.q0;
    if (*W.s_abort) return o;
    const unsigned long long w0 = wall_clock64();
    unsigned spins = 0;
    for (;;) {
        __builtin_amdgcn_s_sleep(2);
        unsigned long long v;
        p2p_u32x4 q0, q1;
        asm volatile("global_load_dwordx2 %0, %3, off " PR_SC "\n\tglobal_load_dwordx4 %1, %4, off " PR_SC "\n\tglobal_load_dwordx4 %2, %5, off " PR_SC "\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(v), "=&v"(q0), "=&v"(q1) : "v"(ps), "v"(p0), "v"(p1) : "memory");
        o.v = v; o.q0 = make_uint4(q0.x, q0.y, q0.z, q0.w); o.q1 = make_uint4(q1.x, q1.y, q1.z, q1.w);
        if ((((uint32_t)(v >> 32)) & 0xffff0000u) == want && p2p_ll_ok(o.q0, tag) && p2p_ll_ok(o.q1, tag)) break;
        if ((++spins & 63u) == 0u && pr_give_up(W, w0)) { pr_abort(W, t, g); break; }
    }
    return o;
}
// ... two self-validating pieces of a record
struct PrLL2 { uint4 q0, q1; };
__device__ __attribute__((noinline)) PrLL2 pr_wait_ll2(const PrWait W, const uint4* p0, const uint4* p1, const uint32_t tag, const int t, const int g) {
    PrLL2 o;
    o.q0 = make_uint4(0u, 0u, 0u, 0u); o.q1 = o.q0;
    if (*W.s_abort) return o;
    uint4 q0, q1;
    const unsigned long long w0 = wall_clock64();
    unsigned spins = 0;
    do {
        __builtin_amdgcn_s_sleep(1);
        pr_load16x2_sys(p0, p1, q0, q1);
        if ((++spins & 63u) == 0u && pr_give_up(W, w0)) { pr_abort(W, t, g); break; }
    } while (!(p2p_ll_ok(q0, tag) && p2p_ll_ok(q1, tag)));
    o.q0 = q0; o.q1 = q1;
    return o;
}
// the exact value of chain s after the last iteration (a key tie in the walk): double 0 of its record — self-validating in the ring,
// plain in the launch's input records (first iteration).  Out of line, arguments by value (an object with an out-of-line member would
// be built in scratch memory every iteration).
__device__ __attribute__((noinline)) double pr_tie_value(const PrWait W, const uint4* ring, const double* plain, const uint32_t tag, const int RW, const int NPV, const int t,
                                                         const uint32_t s) {
    if (plain) return plain[(size_t)s * RW];
    const uint4* a = ring + (size_t)s * RW + NPV;   // (the ring's order: the value behind the parameters)
    uint4 q = pr_load16_sys(a);
    if (!p2p_ll_ok(q, tag)) {
        const unsigned long long w0 = wall_clock64();
        unsigned spins = 0;
        do {
            __builtin_amdgcn_s_sleep(1);
            q = pr_load16_sys(a);
            if ((++spins & 63u) == 0u && pr_give_up(W, w0)) { pr_abort(W, t, (int)s); break; }
        } while (!p2p_ll_ok(q, tag));
    }
    return p2p_ll_double(q);
}
struct PersistWalkValues {   // GUARD of the lean walk (smm_walk_lean.hpp)
    PrWait W; const uint4* ring; const double* plain; uint32_t tag; int RW; int NPV; int t;
    __device__ __forceinline__ double value(const uint32_t s) const { return pr_tie_value(W, ring, plain, tag, RW, NPV, t, s); }
};

// exp() out of line: inlined, its polynomial's nine 64-bit coefficients are hoisted out of the iteration loop into registers, spilled
// to scratch memory there, and fetched back from it in every accept step (the same code as the inlined one: identical results)
__device__ __attribute__((noinline)) double pr_exp(const double x) { return smm_exp(x); }   // (the contract exponential, smm_rng.hpp)

// the slowest tile's progress in this launch (lanes of one wave; words of another launch count as "not started")
__device__ inline int pr_min_progress(const uint32_t* pr_progress, const uint32_t epoch, const int tiles, const int lane) {
    uint32_t m = 0xfffu;
    for (int b = lane; b < tiles; b += 64) {
        const uint32_t w = pr_load4_sys(pr_progress + b);
        const uint32_t rel = (w >> 12) == (epoch & 0xfffffu) ? (w & 0xfffu) : 0u;
        m = min(m, rel);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, off, 64));
    return (int)m;
}
// out of line: the ring entry a tile is about to overwrite has not been read by everybody yet (never in practice)
__device__ __attribute__((noinline)) void pr_wait_progress(const PrWait W, const uint32_t* pr_progress, int* s_minprog, const int need, const int tiles, const int lane,
                                                           const int t, const int chain) {
    unsigned spins = 0;
    const unsigned long long w0 = wall_clock64();
    while (__hip_atomic_load(s_minprog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need && *W.s_abort == 0u) {
        const int m = pr_min_progress(pr_progress, W.epoch, tiles, lane);
        if (lane == 0) __hip_atomic_store(s_minprog, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((++spins & 15u) == 0u && pr_give_up(W, w0)) { if (lane == 0) pr_abort(W, t, chain); break; }
        __builtin_amdgcn_s_sleep(4);
    }
}

// the simulation for a tile of 16 chains with the lane's shocks in registers: half h (512 lanes) sums the draws of moment h; lane l
// takes the draws l, l + 512, ... in that order (numerical contract); two passes of 8 chains (the shocks cost nothing to re-use).
// s_part [NP][8][16].  FULL: ns > 512 (PR_ZR - 1) — every lane has the first PR_ZR - 1 draws, some the last one (the headline's
// ns = 10000): straight-line code; otherwise a scalar branch per draw (whose merges cost register copies).
template <bool FULL>
__device__ __forceinline__ void persist_add_draw(const double zu, const int u, const int nfull, const bool extra, const double (&mu)[8], double (&acc)[8]) {
    const bool all = FULL ? u < PR_ZR - 1 : u < nfull;   // (u is a literal once the callers' loops are unrolled)
    if (all) {
#pragma unroll
        for (int c = 0; c < 8; ++c) { const double x = zu + mu[c]; acc[c] = acc[c] + x; }
    } else if ((FULL || u == nfull) && extra) {   // the ragged last row
#pragma unroll
        for (int c = 0; c < 8; ++c) { const double x = zu + mu[c]; acc[c] = acc[c] + x; }
    }
}
template <int NP, bool FULL>
__device__ __forceinline__ void persist_simulate(const double (&z)[PR_ZR], const int nfull, const bool extra, const double* s_theta, double* s_part,
                                                 const int h, const int wih) {
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        double acc[8], mu[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) { mu[c] = s_theta[(pass * 8 + c) * NP + h]; acc[c] = 0.0; }
#pragma unroll
        for (int u = 0; u < PR_ZR; ++u) persist_add_draw<FULL>(z[u], u, nfull, extra, mu, acc);
        const int lane2 = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        const double tot = wave_reduce_transposed<8>(acc, lane2);
        if ((lane2 & 7) == 0) s_part[(h * 8 + wih) * NORM_CT + pass * 8 + (lane2 >> 3)] = tot;
    }
}
// ... with the lane's shocks in LDS (the control wave, half 0, wave 0 of it: s_z0[u][64]), fetched half at a time: its registers belong
// to the serial parts of the iteration
template <int NP, bool FULL>
__device__ __forceinline__ void persist_simulate_lds(const double* s_z0, const int ns, const int nfull, const double* s_theta, double* s_part) {
    constexpr int HZ = PR_ZR / 2;
    static_assert(PR_ZR % 2 == 0, "two halves");
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        const bool extra = lane < ns - nfull * WG;
        double acc[8], mu[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) { mu[c] = s_theta[(pass * 8 + c) * NP]; acc[c] = 0.0; }
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            double z[HZ];
#pragma unroll
            for (int k = 0; k < HZ; ++k) z[k] = s_z0[(hh * HZ + k) * 64 + lane];
#pragma unroll
            for (int k = 0; k < HZ; ++k) persist_add_draw<FULL>(z[k], hh * HZ + k, nfull, extra, mu, acc);
        }
        const double tot = wave_reduce_transposed<8>(acc, lane);
        if ((lane & 7) == 0) s_part[pass * 8 + (lane >> 3)] = tot;
    }
}

// the workgroup barrier of role-specialised waves (every wave executes the same NUMBER of them per iteration, from different code).
// Bare: the wave's LDS operations are complete (lgkmcnt), its GLOBAL ones stay in flight — a fence would wait for the write-through
// publication, the LDS-DMA of the next lists and the history stores at every barrier, i.e. put their latency on the critical path.
// Whoever hands global data over waits for it itself (vmcnt(0) behind the LDS-DMA).
#define PR_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// out of line (once per launch / almost never): proposal tries past the first four of mysample (AlgoBGP.jl:400-410) for the chains of the
// control wave that have not found a point inside the box yet; the same loop as k_chain_iter_norm's
struct PrTries { const double* rb; uint64_t seed; int rb_t0, N, RBW, rb_tries, user_n, smpl_iters, goff /* first chain of the shard in the population */; };
template <int NP>
struct PrTh { double th[NP]; bool found; };
template <int NP>
__device__ __attribute__((noinline)) PrTh<NP> persist_late_tries(const PrTries A, const double* o, const int t, const int c, const bool valid, const int lane,
                                                                 bool found, const double sigma, const double mu0, const double mu1, const double* s_const,
                                                                 const double th0, const double th1) {
    const int r = lane & 3;
    double mu01[NP], th[NP];
    mu01[0] = mu0; th[0] = th0;
    if constexpr (NP > 1) { mu01[1] = mu1; th[1] = th1; }
    const bool rng_here = A.rb == nullptr;
    const int rb_tries = rng_here ? NORM_NR : A.rb_tries;
    const int max_tries = A.user_n ? min(A.rb_tries, A.smpl_iters) : A.smpl_iters;
    const double* g_rb = A.rb + ((size_t)(t - A.rb_t0) * A.N + (valid ? c : 0)) * A.RBW;
    for (int j0 = NORM_NR; __any(!found) && j0 < max_tries; j0 += NORM_NR) {
        const int j = j0 + r;
        double x[NP], zz[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) { x[k] = 0.0; zz[k] = 0.0; }
        bool ok = !found && j < max_tries;
        if (ok) {
            if (j0 == NORM_NR && !rng_here) {
#pragma unroll
                for (int k = 0; k < NP; ++k) zz[k] = o[1 + NP + k];
            } else if (j < rb_tries) {
#pragma unroll
                for (int k = 0; k < NP; ++k) zz[k] = g_rb[1 + j * NP + k];
            }
            if (j >= rb_tries) {   // past the pre-generated tries: the in-kernel generator (never with injected normals)
#pragma unroll
                for (int q = 0; 2 * q < NP; ++q) {
                    double z0, z1;
                    rng_prop_normal2(A.seed, (uint32_t)(A.goff + c), (uint32_t)t, (uint32_t)j, (uint32_t)q, z0, z1);
                    zz[2 * q] = z0;
                    if (2 * q + 1 < NP) zz[2 * q + 1] = z1;
                }
            }
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const double step = sigma * zz[k];   // MvNormal(mu01, sigma): x = mu + sigma*z
                x[k] = mu01[k] + step;
                if (!(x[k] >= 0.0 && x[k] <= 1.0)) ok = false;   // inclusive bounds, :405
            }
        }
        const unsigned long long m = __ballot(ok);
        const unsigned quad = (unsigned)(m >> (lane & ~3)) & 0xfu;
        if (!found && quad) {
            const int rwin = __builtin_ctz(quad);
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const double lbk = s_const[k];
                const double sc = x[k] * (s_const[NP + k] - lbk);
                const double thk = sc + lbk;   // mapto_ab, mprob.jl:271
                th[k] = quad_bcast_dyn(thk, lane, rwin);
            }
            found = true;
        }
    }
    PrTh<NP> out;
#pragma unroll
    for (int k = 0; k < NP; ++k) out.th[k] = th[k];
    out.found = found;
    return out;
}
