// the PERSISTENT chain kernel of the bundled objective (objfunc_norm, np == nm <= 2, one proposal batch, single shard of at most
// one tile per CU): k_chain_persist_norm — part of libsmmhip (included by smmhip.hip inside its anonymous namespace; gfx950 device
// code).
#pragma once
// ------------------------------------------------------------------------------------------
// ONE launch for a whole run of iterations (up to the end of the look-ahead windows): next_eval for every chain and exchangeMoves!
// between two iterations (AlgoBGP.jl:589-640, 647-716) without a kernel boundary in between.  Results are bit-identical to
// k_chain_iter_norm's (same numerical contract, same arithmetic, same order of everything that has an order).
//
// What a kernel boundary per iteration cost the headline configuration (4096 chains, one 1024-lane tile of 16 chains per CU): the
// boundary itself (~1.5 us), every workgroup re-staging all 4096 walk slots and the whole pair list (~1.6 us), the walk over all
// 13 levels with barriers (~2.6 us), the chain state and 160 KB of shocks re-fetched per launch — 7.3 of 14.4 us next to 7.4 us
// of simulation.  Here, per workgroup (= tile = CU) and iteration:
//   * the chain state, the last accepted records and the lane's SHOCKS (20 doubles: draws l, l + 512, ... of the half's moment)
//     stay in LDS / registers for the whole launch; the simulation issues no load at all;
//   * the tile waits only for ITS CONE of the exchange (smm_cone.hpp: the ~100 pairs / ~100 chains of other tiles its 16 chains'
//     outcome depends on, listed ahead of time by k_exch_plan): after its accept step a tile publishes, per chain, one 8-byte walk
//     slot {order_key32(value), chain | tag << 16} and the self-validating record (smm_p2p.hpp's LL granules) into a ring of PR_K
//     iterations in device memory, with write-through stores; the next iteration's prologue gathers the cone's slots past the caches
//     (one lane per chain, waves 4..7: the control wave's own stores must not sit in front of these loads in its memory queue),
//     looks again where a tag is still the old one, lets wave 0 walk the cone's sub-levels alone, and fetches a donor's record only
//     for a chain that was exchanged;
//   * the next iteration's lists (pairs, gather list) arrive by LDS-DMA under the simulation, its randomness is drawn by wave 1
//     behind its share of the simulation, and the gathering waves start looking for the other tiles' slots as soon as their own
//     share of the simulation is done — under the control wave's accept step;
//   * the waves are ROLE-SPECIALISED (two loops in one kernel, the same number of workgroup barriers in each): wave 0 is the control
//     wave and nothing else competes for its registers (its shocks come out of LDS for its share of the simulation); history rows
//     leave through LDS and are stored by a worker wave; what happens once per launch or almost never (the launch's first and last
//     iteration, late proposal tries, waiting) is out of line.
// Nobody waits for an acknowledgement, no grid barrier, no atomics on the way.  Why the ring cannot be overrun: a tile publishes
// iteration i into entry i mod PR_K only after every tile has announced (progress word, one write-through store per tile and
// iteration, read by an otherwise idle wave under the simulation) that it has finished the prologue reads of iteration i - PR_K + 1
// — with PR_K = 8 the test never fails in practice (tiles are never more than ~2 iterations apart: each needs ~40 % of all tiles'
// last results), so it costs nothing on the critical path, and it holds under any skew (tests: a delayed workgroup).  Deadlock
// freedom: the slowest tile waits only for publications of iterations that every other tile has passed already, and those stay in
// the ring until it has read them.  All tiles must be RESIDENT (one per CU: the host checks the occupancy against the grid); every
// spin has a time-out, and a tile that gives up raises the launch's abort word so that nobody else waits their four seconds.
//
// Errors.  A hard error (AlgoBGP.jl:341,409) is reported like everywhere (report_error); tiles run on to the end of the launch
// (a tile that stopped would starve the others), and the HOST, when it finds the error word set after a launch of this kernel,
// restores the state it saved before the first such launch since the last check and repeats those iterations on the one-launch-
// per-iteration path, which stops at the failing iteration with the library's documented state (smmhip.hip, persist_repair).
// ------------------------------------------------------------------------------------------
constexpr int PR_K = 8;          // iterations in the ring
constexpr int PR_ZR = 20;        // shocks per lane held in registers: ns <= 512 * PR_ZR
constexpr int PR_MAX_ITERS = 4000;   // iterations per launch (12 bits of the tags count them)
#ifndef SMM_EXP_PR_DELAY
#define SMM_EXP_PR_DELAY 8
#endif
constexpr int PR_GATHER_DELAY = SMM_EXP_PR_DELAY;   // s_sleep units (64 clocks) between this tile's publication and the gather's first look
constexpr int PR_STW = 12;       // doubles of chain state in front of the record in a tile's LDS line
constexpr unsigned long long PERSIST_TMO_FIRST = 40000000ull;   // 0.4 s of the 100 MHz wall clock: the spins of a context's first launches (smmhip.hip, launch_chain_persist)

// LDS of a tile: [walk slots: 8 bytes per chain of the population + 4] [pair lists x 2] [gather lists x 2] [headers x 4] and, as doubles:
// theta[16][NP] part[NP][8][16] lines[16][LW] rng[2][64][1 + 2 NP] hrow[16][HW] xrow[16][HW] z0[PR_ZR][64] const[16] misc[16] donor[2][64] (uint4) gth[Ng4][NP]
__host__ __device__ inline int persist_line(int np) { return PR_STW + ((3 + 2 * np + 1) & ~1); }
__host__ __device__ inline size_t persist_smem_bytes(int Ng, int np) {
    const size_t slots = (size_t)(((Ng + 3) & ~3) + 4) * 8;
    const size_t lists = 2 * (size_t)CONE_LEVELS * 64 * 4 + 2 * (size_t)CONE_GCAP * 2 + 4 * 16 * 4;
    const size_t hw = (size_t)((H_PARAMS + 2 * np + 1) & ~1);
    const size_t dbl = (size_t)NORM_CT * np + (size_t)np * 8 * NORM_CT + (size_t)NORM_CT * persist_line(np) + 2 * 64 * (size_t)(1 + 2 * np) +
                       2 * NORM_CT * hw + (size_t)PR_ZR * 64 + 16 + 16 + 2 * 64 * 2 + (size_t)((Ng + 3) & ~3) * np;   // (PersistLds)
    return slots + lists + dbl * 8;
}
__host__ __device__ inline size_t persist_ring_slot_bytes(int Ng) { return (size_t)PR_K * (((size_t)Ng + 4) * 8); }
__host__ __device__ inline size_t persist_ring_rec_bytes(int Ng, int RW) { return (size_t)PR_K * (size_t)Ng * RW * 16; }

// tags: never 0 (the ring starts zeroed and is zeroed again whenever the 7 epoch bits of the slot tag wrap)
__device__ inline uint32_t pr_tag16(const uint32_t epoch, const int rel) { return 0x8000u | ((epoch & 0x7fu) << 8) | ((uint32_t)rel & 0xffu); }
__device__ inline uint32_t pr_tag32(const uint32_t epoch, const int rel) { return 0x80000000u | ((epoch & 0x7ffffu) << 12) | ((uint32_t)rel & 0xfffu); }
__device__ inline uint32_t pr_progress_word(const uint32_t epoch, const int rel) { return (epoch << 12) | (uint32_t)rel; }

// The ring is written with write-through stores and read past the caches.  Scope: the tiles are workgroups of ONE device, so the
// agent scope (sc1) is enough; SMM_EXP_PR_SYS=1 makes it the system scope (sc0 sc1) of the p2p windows, for comparison.
#ifndef SMM_EXP_PR_SYS
#define SMM_EXP_PR_SYS 1   // (round 5: the same words also travel between the ranks' windows, smm_chain_persist_loc.hpp — one scope for all; measured equal, EXPERIMENTS.md R4.3)
#endif
#if SMM_EXP_PR_SYS
#define PR_SC "sc0 sc1"
#else
#define PR_SC "sc1"
#endif
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

// The ring holds a record's doubles in its own order — parameters first, so that the gather fetches them together with the walk
// slot: theta[NP], value, prob, status, sim_moments[NP], pad — one uint4 {lo, tag, hi, tag} per double.
template <int NP> __device__ constexpr int pr_ring_index(const int classic) {   // classic: value, prob, status, theta[NP], sim_moments[NP], pad
    return classic < 3 ? NP + classic : classic < 3 + NP ? classic - 3 : classic;
}

// what the kernel needs, and nothing else (the whole KParams block as a kernel argument cost the control wave hundreds of scalar
// spills: every field the loop touches is loop-invariant and wants a register)
struct PersistArgs {
    const uint32_t* cone_hdr; const uint32_t* cone_pairs; const uint16_t* cone_gather; const uint32_t* cone_ok;
    uint2* pr_slot; uint4* pr_rec; uint32_t* pr_progress; uint32_t* pr_ctl;
    double* cs; const double* rec_in; double* rec_out; double* vals_out; uint2* slot8_out; uint32_t* walk_flags;
    double* hrec; unsigned long long* err; unsigned long long* ts;
    const double *Z, *lb, *ub, *mom, *w, *objp;
    const double* rb;                 // randomness blocks of injected tables (null: drawn in the kernel)
    int N, Ng, ns, zstride, plan_t0, exch_from, sigma_update_steps, smpl_iters, t0, t1;
    int rb_t0, RBW, rb_tries, user_n, failbox;
    int ring_k;                       // entries of the ring in use (a power of two <= PR_K)
    int slow_tile, slow_ticks;        // test build: this tile's control wave idles so many wall-clock ticks before it publishes (skew)
    int walk_first;                   // the exchange of iteration t0 - 1 is still to be applied: the first iteration walks it on the launch's input records
    unsigned long long tmo;           // ticks a spin may last (PERSIST_TMO_FIRST until a launch of the context has come through, then P2P_TIMEOUT_TICKS)
    uint32_t epoch;
    double sigma_adjust_by;
    uint64_t seed;
};

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
__device__ __attribute__((noinline)) double pr_exp(const double x) { return exp(x); }

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

template <int NP>
struct PersistLds {   // where things are in a tile's LDS (byte offsets from its start; the walk slots start at 0)
    using L = NormLayout<NP>;
    static constexpr int CT = NORM_CT, RW = L::RW, HW = L::HW, LW = PR_STW + RW, RNGW = 1 + 2 * NP;
    uint32_t pbase, gbase, hbase;
    uint2* slots; uint32_t* s_hdr;
    double *s_theta, *s_part, *s_st, *s_rng, *s_hrow, *s_xrow, *s_z0, *s_const, *s_gth;
    uint4* s_donor;
    unsigned long long* s_ts;
    unsigned* s_arrived; int* s_minprog; unsigned* s_abort; unsigned* s_xmask; int* s_glready; int* s_pub;
    __device__ inline PersistLds(unsigned char* lds, const int Ng4) {
        pbase = 8u * (uint32_t)(Ng4 + 4);
        gbase = pbase + 2u * CONE_LEVELS * 64 * 4;
        hbase = gbase + 2u * CONE_GCAP * 2;
        slots = (uint2*)lds;
        s_hdr = (uint32_t*)(lds + hbase);
        s_theta = (double*)(lds + hbase + 4 * 16 * 4);
        s_part = s_theta + CT * NP;
        s_st = s_part + NP * 8 * CT;
        s_rng = s_st + CT * LW;
        s_hrow = s_rng + 2 * 64 * RNGW;
        s_xrow = s_hrow + CT * HW;
        s_z0 = s_xrow + CT * HW;
        s_const = s_z0 + PR_ZR * 64;          // lb[NP] ub[NP] mom[NP] w[NP] failbox[2] ns
        s_ts = (unsigned long long*)(s_const + 16);   // [8] + 8 words of flags = 16 doubles
        s_arrived = (unsigned*)(s_ts + 8);
        s_minprog = (int*)(s_arrived + 1);
        s_abort = s_arrived + 2;
        s_xmask = s_arrived + 3;              // bit cl: chain cl's row of the last iteration is rewritten (s_xrow)
        s_glready = (int*)(s_arrived + 4);    // the exchange whose gather list has landed
        s_pub = (int*)(s_arrived + 5);        // the iteration this tile has published (the others publish at about the same time)
        s_donor = (uint4*)(s_const + 16 + 16);   // [2][64]: the donors' records as they land (LDS-DMA of the control wave: lane L -> entries L and 64 + L)
        s_gth = s_const + 16 + 16 + 2 * 64 * 2;   // [Ng4][NP]: the parameters of the gathered chains' last accepted records (the donors' come from here)
    }
};

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

template <int NP>
__global__ __launch_bounds__(NORM_WG, 4) void k_chain_persist_norm(const PersistArgs A) {
    static_assert(NP == 1 || NP == 2, "one moment per half of the workgroup");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    using LY = PersistLds<NP>;
    constexpr int CT = NORM_CT, RW = LY::RW, HW = LY::HW, LW = LY::LW, RNGW = LY::RNGW;
    constexpr int NPC = RW / 2, NPH = HW / 2;   // 16-byte pieces of a record / a history row
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = (int)blockIdx.x, tiles = (int)gridDim.x;
    const int N = A.N, Ng4 = (A.Ng + 3) & ~3;
    const LY Y(lds, Ng4);
    const uint32_t epoch = A.epoch;
    const int t0 = A.t0, t1 = A.t1;
    const PrWait W{A.err, A.pr_ctl, Y.s_abort, A.epoch, A.tmo};
    const int rmask = A.ring_k - 1;   // (the ring's depth: PR_K; the test build can make it smaller)
    const bool exch_any = A.Ng > 1;
    auto exch_on = [&](const int tx) { return exch_any && tx >= A.exch_from; };   // AlgoBGP.jl:637

    // an EARLIER launch raised a hard error: nothing is stored any more (every tile decides the same: errors of this launch's own
    // iterations do not count here)
    if (error_before(*(const volatile unsigned long long*)A.err, t0)) return;

    // ---- once per launch: the tile's chain state and records, the constants, wave 0's shocks into LDS, the lists of the pending exchange ----
    if (tid < 64) {
        const int cl = lane >> 2, r = lane & 3, c = tile * CT + cl;
        if (c < N) {
            const double2* g_cs = (const double2*)(A.cs + (size_t)c * CSW);
            const double2* g_rec = (const double2*)(A.rec_in + (size_t)c * RW);
            double2* st2 = (double2*)(Y.s_st + cl * LW);
            for (int i = r; i < 6; i += 4) st2[i] = g_cs[i];
            for (int i = r; i < NPC; i += 4) st2[PR_STW / 2 + i] = g_rec[i];
            if (r == 0) {
                Y.slots[c] = make_uint2(order_key32(g_rec[0].x), (uint32_t)c);   // the tile's own slots of the first walk
                for (int k = 0; k < NP; ++k) Y.s_gth[c * NP + k] = A.rec_in[(size_t)c * RW + 3 + k];   // (a donor may be a chain of the same tile)
            }
        }
#pragma unroll
        for (int u = 0; u < PR_ZR; ++u) Y.s_z0[u * 64 + lane] = (lane + u * WG < A.ns) ? A.Z[lane + (size_t)u * WG] : 0.0;   // (wave 0: half 0, lanes 0..63)
    }
    if (tid >= 64 && tid < 64 + NP) {
        const int k = tid - 64;
        Y.s_const[k] = A.lb[k]; Y.s_const[NP + k] = A.ub[k]; Y.s_const[2 * NP + k] = A.mom[k]; Y.s_const[3 * NP + k] = A.w[k];
    }
    if (tid == 128) {
        Y.s_const[4 * NP] = A.failbox ? A.objp[0] : 1.0; Y.s_const[4 * NP + 1] = A.failbox ? A.objp[1] : 0.0;   // (an empty interval: no "exception")
        Y.s_const[4 * NP + 2] = (double)A.ns;
        Y.slots[Ng4] = make_uint2(1u, 0u); Y.slots[Ng4 + 1] = make_uint2(2u, 0u);
        *Y.s_arrived = 0u; *Y.s_minprog = 0; *Y.s_abort = 0u; *Y.s_xmask = 0u; *Y.s_glready = t0 - 1; *Y.s_pub = t0 - 1;
    }
    if (tid >= 192 && tid < 200) Y.s_ts[tid - 192] = 0ull;
    // headers of the pending exchange t0 - 1 (walked by the first iteration) and of exchange t0
    if (wave == 3 && lane < CONE_HDRW) {
        if (A.walk_first) Y.s_hdr[((t0 - 1) & 3) * 16 + lane] = A.cone_hdr[((size_t)(t0 - 1 - A.plan_t0) * tiles + tile) * CONE_HDRW + lane];
        if (t0 < t1 && exch_on(t0)) Y.s_hdr[(t0 & 3) * 16 + lane] = A.cone_hdr[((size_t)(t0 - A.plan_t0) * tiles + tile) * CONE_HDRW + lane];
    }
    PR_BARRIER();
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
    // ... and, once the wave's own piece has landed, the tail of its sub-levels: dummy pairs that never swap
    auto pad_lists = [&](const int tx) {
        const uint32_t* hd = Y.s_hdr + (tx & 3) * 16;
        const int nsub1 = (int)(hd[0] & 0xffffu);
        const uint32_t dummy = (8u * (uint32_t)Ng4) | ((8u * (uint32_t)(Ng4 + 1)) << 16);
        uint32_t* pw = (uint32_t*)(lds + Y.pbase) + (tx & 1) * (CONE_LEVELS * 64);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int s = 4 * (wave - 8) + q;
            if (s < nsub1) {
                const uint32_t cnt = (hd[1 + (s >> 2)] >> (8 * (s & 3))) & 0xffu;
                if ((uint32_t)lane >= cnt) pw[s * 64 + lane] = dummy;
            }
        }
    };
    if (A.walk_first) {   // (the previous kernel — of any form — left the exchange of its last iteration to this one)
        if (wave == 3 || wave >= 8) {
            request_lists(t0 - 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (wave >= 8) pad_lists(t0 - 1);
        }
        if (wave == 2 && lane == 0 && A.cone_ok[t0 - 1 - A.plan_t0] == 0u) pr_report(A.err, 3, t0, tile * CT);
        PR_BARRIER();
        if (wave >= 4 && wave < 8) {   // its cone's initial slots from the launch's input records (plain memory: complete at the launch)
            const int ngat = (int)(Y.s_hdr[((t0 - 1) & 3) * 16] >> 16);
            const uint16_t* gl = (const uint16_t*)(lds + Y.gbase) + ((t0 - 1) & 1) * CONE_GCAP;
            for (int e = tid - 256; e < ngat; e += 256) {
                const int g = (int)gl[e];
                const double* gr = A.rec_in + (size_t)g * RW;
                Y.slots[g] = make_uint2(order_key32(gr[0]), (uint32_t)g);
#pragma unroll
                for (int k = 0; k < NP; ++k) Y.s_gth[g * NP + k] = gr[3 + k];
            }
        }
    }

    if (wave != 0) {
        // =====================================================================================================================
        // the WORKER waves 1..15: simulation (all), randomness (1), progress (2), gather list + history (3), gather (4..7), pair lists (8..15)
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
        // the randomness of iteration tn into its parity's rows: wave 1, lane = the control wave's lane (chain, try)
        auto make_rng = [&](const int tn) {
            const int c1 = tile * CT + (lane >> 2);
            if (c1 >= N) return;
            double* o = Y.s_rng + ((tn & 1) * 64 + lane) * RNGW;
            if (rng_here) {
                o[0] = rng_u(A.seed, (uint32_t)c1, (uint32_t)tn);                     // probs_acc[iter], AlgoBGP.jl:85
#pragma unroll
                for (int q = 0; 2 * q < NP; ++q) {                                    // rand(RAND, d) of try r, :404
                    double z0, z1;
                    rng_prop_normal2(A.seed, (uint32_t)c1, (uint32_t)tn, (uint32_t)(lane & 3), (uint32_t)q, z0, z1);
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
        // history rows out of LDS (16 chains x NPH pieces of 16 bytes), by wave 3
        auto store_rows = [&](const double* s_row, const int trow, const unsigned mask) {
            for (int e = lane; e < CT * NPH; e += 64) {
                const int cl = e / NPH, i = e - cl * NPH, c = tile * CT + cl;
                if (c < N && ((mask >> cl) & 1u)) ((double2*)(A.hrec + ((size_t)(trow - 1) * N + c) * HW))[i] = ((const double2*)(s_row + cl * HW))[i];
            }
        };
        if (wave == 1) make_rng(t0);
        for (int t = t0; t <= t1; ++t) {
            const int rel = t - t0 + 1;                       // iteration of this launch, from 1
            PR_BARRIER();   // BA: the cone's slots are staged, this iteration's lists and randomness are in LDS, the last epilogue is done
            uint32_t nhdr = 0u;
            const bool want_hdr = wave == 3 && lane < CONE_HDRW && t + 1 < t1 && exch_on(t + 1);
            if (wave == 3) {
                if (t > t0) {   // the last iteration's history rows, and the rows of iteration t-2 of the chains it found exchanged (set_eval! of swap_ev_ij!): complete since the barrier
                    store_rows(Y.s_hrow, t - 1, 0xffffu);
                    store_rows(Y.s_xrow, t - 2, *Y.s_xmask);
                }
                if (want_hdr) nhdr = A.cone_hdr[((size_t)(t + 1 - A.plan_t0) * tiles + tile) * CONE_HDRW + lane];
            }
            // the lists of the NEXT exchange (t), landing under the simulation (not by the gathering waves: the DMA would sit in front of
            // the gather's loads in their memory queue)
            if (t < t1 && exch_on(t)) request_lists(t);
            PR_BARRIER();   // BB: the proposals are in LDS
            // ---- simulation: every lane, its resident shocks x 16 chains ----
            if (simw) {
                if (nfull == PR_ZR - 1) persist_simulate<NP, true>(z, nfull, extra, Y.s_theta, Y.s_part, h, wih);
                else persist_simulate<NP, false>(z, nfull, extra, Y.s_theta, Y.s_part, h, wih);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                const int lane3 = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
                if (lane3 == 0) __hip_atomic_fetch_add(Y.s_arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            // ---- behind the simulation (lane ids derived anew: nothing lane-dependent lives across it) ----
            const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            const int tid = wave * 64 + lane;
            if (wave == 3 || wave >= 8) {
                if (wave == 3 && want_hdr) Y.s_hdr[((t + 1) & 3) * 16 + lane] = nhdr;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA has landed
                if (wave >= 8 && t < t1 && exch_on(t)) pad_lists(t);
                if (wave == 3 && lane == 0) __hip_atomic_store(Y.s_glready, t, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (wave == 1 && t < t1) make_rng(t + 1);
            if (wave == 2) {
                const int m = pr_min_progress(A.pr_progress, epoch, tiles, lane);
                if (lane == 0) {
                    __hip_atomic_store(Y.s_minprog, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (pr_load4_sys(A.pr_ctl) == epoch) *Y.s_abort = 1u;
                    if (t < t1 && exch_on(t) && A.cone_ok[t - A.plan_t0] == 0u) pr_report(A.err, 3, t + 1, tile * CT);   // (a cone does not fit its caps: the host repeats the step)
                }
            }
            if (wave >= 4 && wave < 8 && t < t1 && exch_on(t)) {
                // ---- gather for the NEXT iteration's walk: the cone's initial slots out of the ring, past the caches, under the control
                // wave's accept step; every word says which iteration it is from, a lane that finds an older one looks again ----
                while (__hip_atomic_load(Y.s_glready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != t) __builtin_amdgcn_s_sleep(1);
                // (no point in looking before anybody can have published: the tiles run in step, so this tile's own publication is the
                // clock — and failed looks are not free: thousands of lanes re-reading scattered words load every CU's memory queue)
                while (__hip_atomic_load(Y.s_pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != t) __builtin_amdgcn_s_sleep(1);
                __builtin_amdgcn_s_sleep(PR_GATHER_DELAY);
                const int ngat = (int)(Y.s_hdr[(t & 3) * 16] >> 16);
                const unsigned long long* rs = (const unsigned long long*)A.pr_slot + (size_t)(rel & rmask) * (A.Ng + 4);
                const uint32_t want = pr_tag16(epoch, rel) << 16;
                const uint16_t* gl = (const uint16_t*)(lds + Y.gbase) + (t & 1) * CONE_GCAP;
                const uint4* rr = (const uint4*)A.pr_rec + (size_t)(rel & rmask) * A.Ng * RW;
                const uint32_t tag = pr_tag32(epoch, rel);
                for (int e = tid - 256; e < ngat; e += 256) {
                    const int g = (int)gl[e];
                    // the slot and the record's parameters (the ring's first NP doubles), requested together
                    unsigned long long v;
                    p2p_u32x4 q0, q1;
                    asm volatile("global_load_dwordx2 %0, %3, off " PR_SC "\n\tglobal_load_dwordx4 %1, %4, off " PR_SC "\n\tglobal_load_dwordx4 %2, %5, off " PR_SC "\n\ts_waitcnt vmcnt(0)"
                                 : "=&v"(v), "=&v"(q0), "=&v"(q1) : "v"(rs + g), "v"(rr + (size_t)g * RW), "v"(rr + (size_t)g * RW + (NP - 1)) : "memory");
                    uint4 u0 = make_uint4(q0.x, q0.y, q0.z, q0.w), u1 = make_uint4(q1.x, q1.y, q1.z, q1.w);
                    if (__builtin_expect((((uint32_t)(v >> 32)) & 0xffff0000u) != want || !(p2p_ll_ok(u0, tag) && p2p_ll_ok(u1, tag)), 0)) {
                        const PrGather w3 = pr_wait_gather(W, rs + g, rr + (size_t)g * RW, rr + (size_t)g * RW + (NP - 1), want, tag, t + 1, g);
                        v = w3.v; u0 = w3.q0; u1 = w3.q1;
                    }
                    Y.slots[g] = make_uint2((uint32_t)v, (uint32_t)(v >> 32) & 0xffffu);
                    Y.s_gth[g * NP] = p2p_ll_double(u0);
                    if constexpr (NP > 1) Y.s_gth[g * NP + 1] = p2p_ll_double(u1);
                }
            }
        }
        PR_BARRIER();   // the last epilogue is done
        if (wave == 3) { store_rows(Y.s_hrow, t1, 0xffffu); store_rows(Y.s_xrow, t1 - 1, *Y.s_xmask); }
        if (wave == 1) {   // the result blocks where the next launch (of any form) expects them: out of the chains' lines
            const int cl = lane >> 2, r = lane & 3, c = tile * CT + cl;
            if (c < N) {
                const double2* st2 = (const double2*)(Y.s_st + cl * LW);
                double2* g_cs = (double2*)(A.cs + (size_t)c * CSW);
                for (int i = r; i < 6; i += 4) g_cs[i] = i == 2 ? make_double2(st2[2].x, 0.0) : st2[i];   // (the WASX field carried the record's source during the launch)
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
    // the CONTROL wave: four lanes per chain with identical state (lane r of a quad evaluates proposal try r).
    // Per iteration, on the critical path between two publications: the walk, the proposal (parameters of the chain's own record or,
    // for an exchanged chain, of its donor's — gathered with the slots), its share of the simulation, the objective and the accept
    // step up to the publication.  Everything else — the donor's whole record (requested behind the walk, looked at only here),
    // settling iteration t-1, counters, sigma, best values, the chain's line, the history rows — runs BEHIND the publication, in the
    // shadow of the time its stores need to become visible to the other tiles.
    // (every section derives its lane ids anew — mbcnt needs no input register — so that nothing lane-dependent stays live across the
    // register-hungry simulation: the compiler spilled such values to scratch memory, and scratch reloads are memory operations)
    // =========================================================================================================================
    const int nfull0 = A.ns / WG;
    for (int t = t0; t <= t1; ++t) {
        const int rel = t - t0 + 1;
        const bool first = t == t0;
        const bool exch = first ? A.walk_first != 0 : exch_on(t - 1);
        PR_BARRIER();   // BA
        {
            const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            const int cl = lane >> 2, r = lane & 3;
            const int c = tile * CT + cl;                         // (single shard: global == local chain id)
            const bool valid = c < N;
            double* st = Y.s_st + cl * LW;
            unsigned long long ts1 = 0;
            if (A.ts && lane == 0) { ts1 = wall_clock64(); if (!first) Y.s_ts[0] += ts1 - Y.s_ts[7]; }   // (the wait for the gather at the barrier)
            // ---- the walk over the cone's sub-levels: this wave alone, no barriers (its LDS operations complete in order) ----
            uint32_t src = (uint32_t)c;
            int partner = 0;
            const uint32_t lbase = Y.pbase + (uint32_t)((t - 1) & 1) * (CONE_LEVELS * 64 * 4);
            if (exch) {
                const int nsub = (int)(Y.s_hdr[((t - 1) & 3) * 16] & 0xffffu);
                const PersistWalkValues values{W, (const uint4*)A.pr_rec + (size_t)((rel - 1) & rmask) * A.Ng * RW, first ? A.rec_in : nullptr, pr_tag32(epoch, rel - 1), RW, NP, t};
                lean_walk_levels<64, 0, false, PersistWalkValues>(nullptr, 1, lbase, (uint32_t)(64 * lane), nsub, lane, 0, 0.0, values);
                if (valid) {
                    const uint32_t kmeta = Y.slots[c].y;
                    src = kmeta & 0xffffu;
                    if (kmeta >> 16) partner = (int)lean_partner<0>(lds, lbase, kmeta, (uint32_t)c);   // set_exchanged!, :747-748
                }
            }
            const bool donor = valid && src != (uint32_t)c;
            // the donor's whole record (swap_ev_ij!, :734-749), requested now and looked at behind the simulation
            // (by LDS-DMA: lane r of the quad brings the record's doubles r and 4 + r — ring order, a uint4 each — to entries lane and 64 + lane)
            if (donor && !first) {
                const uint4* g_ll = (const uint4*)A.pr_rec + ((size_t)((rel - 1) & rmask) * A.Ng + src) * RW;
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
                    th_old[k] = donor ? Y.s_gth[src * NP + k] : st[PR_STW + 3 + k];   // the record the chain continues from: its own, or its donor's
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
                const PrTries TR{A.rb, A.seed, A.rb_t0, A.N, A.RBW, A.rb_tries, A.user_n, A.smpl_iters, 0};
                const PrTh<NP> lt = persist_late_tries<NP>(TR, o, t, c, valid, lane, found, sigma, mu01[0], mu01[NP - 1], Y.s_const, th[0], th[NP - 1]);
#pragma unroll
                for (int k = 0; k < NP; ++k) th[k] = lt.th[k];
                found = lt.found;
                if (!found && r == 0) pr_report(A.err, 2, t, c);   // :409
            }
            if (r == 0) {
#pragma unroll
                for (int k = 0; k < NP; ++k) Y.s_theta[cl * NP + k] = th[k];
                st[CS_PARTNER] = (double)partner;
                st[CS_WASX] = (double)src;   // (free during the launch: where the record comes from, for the section behind the simulation)
            }
            if (A.ts && lane == 0) {
                const unsigned long long ts4 = wall_clock64();
                Y.s_ts[1] += ts2 - ts1; Y.s_ts[3] += ts4 - ts2; Y.s_ts[6] = ts4;
            }
        }
        PR_BARRIER();   // BB
        // ---- this wave's share of the simulation: its shocks come out of LDS (they must not occupy registers during the serial parts) ----
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
            const bool donor = src != c;
            double rc2[RW];
#pragma unroll
            for (int f = 0; f < RW; ++f) rc2[f] = st[PR_STW + f];
            if (donor) {
                if (__builtin_expect(first, 0)) {   // the launch's input records (plain)
                    const double2* g_rec = (const double2*)(A.rec_in + (size_t)src * RW);
#pragma unroll
                    for (int i = 0; i < NPC; ++i) { const double2 q = g_rec[i]; rc2[2 * i] = q.x; rc2[2 * i + 1] = q.y; }
                } else {
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
                    if (__builtin_expect(!ok, 0)) {   // (cannot be: the gather has validated this chain's slot and parameters; late stores of the SAME publication?)
                        const uint4* g_ll = (const uint4*)A.pr_rec + ((size_t)((rel - 1) & rmask) * A.Ng + src) * RW;
#pragma unroll
                        for (int f = 0; f < RW; f += 2) {
                            const PrLL2 w2 = pr_wait_ll2(W, g_ll + f, g_ll + f + 1, tag, t, c);
                            rr[f] = p2p_ll_double(w2.q0); rr[f + 1] = p2p_ll_double(w2.q1);
                        }
                    }
#pragma unroll
                    for (int f = 0; f < RW; ++f) rc2[f] = rr[pr_ring_index<NP>(f)];
                }
            }
            // every read of the ring's last entry is done: say so (the publication of iteration rel + PR_K - 1 waits for it)
            if (lane == 0) __hip_atomic_store(A.pr_progress + tile, pr_progress_word(epoch, rel), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
                if (!(value >= 0.0) && r == 0) pr_report(A.err, 1, t, c);   // :341
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
            // ---- publish: the walk slot and the self-validating record of iteration t into the ring (write-through stores) ----
            if (t < t1) {
                // entry rel mod PR_K still holds iteration rel - PR_K: has everybody read it?  (always, in practice)
                if (__builtin_expect(rel > rmask && __hip_atomic_load(Y.s_minprog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < rel - rmask, 0))
                    pr_wait_progress(W, A.pr_progress, Y.s_minprog, rel - rmask, tiles, lane, t, c);
#ifdef SMM_TEST_HOOKS
                if (tile == A.slow_tile) { const unsigned long long w0 = wall_clock64(); while (wall_clock64() - w0 < (unsigned long long)A.slow_ticks) __builtin_amdgcn_s_sleep(8); }
#endif
                if (r == 0)
                    pr_store8((uint2*)A.pr_slot + (size_t)(rel & rmask) * (A.Ng + 4) + c,
                              (unsigned long long)order_key32(v) | ((unsigned long long)((uint32_t)c | (pr_tag16(epoch, rel) << 16)) << 32));
                unsigned char* g_ll = (unsigned char*)A.pr_rec + ((size_t)(rel & rmask) * A.Ng + c) * RW * 16;
                double ro[8];   // the ring's order
#pragma unroll
                for (int f = 0; f < 8; ++f) ro[f] = 0.0;
#pragma unroll
                for (int f = 0; f < RW; ++f) ro[pr_ring_index<NP>(f)] = nr[f];
                const double2 pv = sel4(r, make_double2(ro[0], ro[1]), make_double2(ro[2], ro[3]), make_double2(ro[4], ro[5]), make_double2(ro[6], ro[7]));
                if (r < NPC) pr_store_ll(g_ll + (size_t)r * 32, pv, pr_tag32(epoch, rel));
                if (lane == 0) __hip_atomic_store(Y.s_pub, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (A.ts && lane == 0) { const unsigned long long ts6 = wall_clock64(); Y.s_ts[4] += ts5 - Y.s_ts[6]; Y.s_ts[5] += ts6 - ts5; Y.s_ts[6] = ts6; }
            // ================= behind the publication =================
            // ---- settle iteration t-1 (as k_chain_iter_norm's prologue; F_CLOSE_PREV always: the host starts this kernel behind a closed iteration) ----
            const int partner = (int)st[CS_PARTNER];
            const double sig = st[CS_SIGMA];
            int nn = (int)st[CS_NNOEX], na = (int)st[CS_NACC];
            double bp = st[CS_BEST], bpid = st[CS_BESTID];
            if (partner != 0) {
                // set_eval!(ci, ej) of swap_ev_ij! as a history record: the chain's record of iteration t-1 is the donor's last
                // accepted one (accepted = true, the donor's prob/status), curr = donor value, best against iteration t-2 (:231-243)
                const double dv = rc2[0];
                if (dv < st[CS_BESTP]) { bp = dv; bpid = (double)(t - 1); }
                else { bp = st[CS_BESTP]; bpid = st[CS_BESTPID]; }
                if (r == 0) {   // (the row goes out through LDS: wave 3 stores it behind the next barrier)
                    double* hx = Y.s_xrow + cl * HW;
                    hx[H_VALUE] = dv; hx[H_PROB] = rc2[1]; hx[H_CURR] = dv; hx[H_BEST] = bp; hx[H_BESTID] = bpid;
                    hx[H_EXCH] = (double)partner; hx[H_ACC] = 1.0; hx[H_STATUS] = rc2[2];
#pragma unroll
                    for (int k = 0; k < 2 * NP; ++k) hx[H_PARAMS + k] = rc2[3 + k];
                    if (HW > H_PARAMS + 2 * NP) hx[HW - 1] = 0.0;
                }
            } else { nn += 1; na += (int)st[CS_LACC]; }   // set_acceptRate!, :253-257 (exchanged iterations do not count)
            // ---- the rest of doAcceptReject! and set_eval! (:220-245) for iteration t ----
            double nsig = sig;
            const bool upd = (t % A.sigma_update_steps) == 0;
            if (upd || t == t1) {   // (the rate is looked at where sigma is adapted, and by whoever reads the state after the launch)
                const double rate = (double)(na + (acc ? 1 : 0)) / (double)(nn + 1);   // set_acceptRate!, :253-257
                if (upd) nsig = (rate > 0.234) ? sig * (1.0 + A.sigma_adjust_by) : sig * (1.0 - A.sigma_adjust_by);   // :381-390
                if (r == 0) st[CS_RATE] = rate;
            }
            double bestv, bestid;
            const double currv = acc ? value : old;
            if (value < bp) { bestv = value; bestid = (double)t; }
            else { bestv = bp; bestid = bpid; }
            if (r == 0) {
                Y.slots[c] = make_uint2(order_key32(v), (uint32_t)c);   // the tile's own slots of the next walk
#pragma unroll
                for (int k = 0; k < NP; ++k) Y.s_gth[c * NP + k] = nr[3 + k];   // (... and parameters: a donor may be a chain of the same tile)
                // ---- the chain's line for the next iteration ----
                st[CS_SIGMA] = nsig; st[CS_NNOEX] = (double)nn; st[CS_NACC] = (double)na; st[CS_LACC] = accd;
                st[CS_BEST] = bestv; st[CS_BESTID] = bestid; st[CS_BESTP] = bp; st[CS_BESTPID] = bpid;
#pragma unroll
                for (int f = 0; f < RW; ++f) st[PR_STW + f] = nr[f];
                // ---- the history row, through LDS (wave 3 stores it behind the next barrier) ----
                double* hv = Y.s_hrow + cl * HW;
                hv[H_VALUE] = value; hv[H_PROB] = prob; hv[H_CURR] = currv; hv[H_BEST] = bestv; hv[H_BESTID] = bestid;
                hv[H_EXCH] = 0.0; hv[H_ACC] = accd; hv[H_STATUS] = (double)status;
#pragma unroll
                for (int k = 0; k < NP; ++k) { hv[H_PARAMS + k] = th2[k]; hv[H_PARAMS + NP + k] = sm[k]; }
                if (HW > H_PARAMS + 2 * NP) hv[HW - 1] = 0.0;
            }
        }
        {   // which chains' rows of iteration t-1 are rewritten (s_xrow)
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
    PR_BARRIER();   // the last epilogue is done (the workers store the last history rows and the result blocks)
    {
        const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        if (A.ts && lane < 7) A.ts[(size_t)tile * 8 + lane] = lane < 6 ? Y.s_ts[lane] : (unsigned long long)(t1 - t0 + 1);
    }
}
