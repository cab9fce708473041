// the chain kernel of the bundled objective (objfunc_norm, np == nm <= 4, one proposal batch): k_chain_iter_norm — part of
// libsmmhip (included by smmhip.hip inside its anonymous namespace; gfx950 device code).
#pragma once
// ------------------------------------------------------------------------------------------
// k_chain_iter_norm<NP, WALK>: the same iteration as k_chain_iter<1, ...> (next_eval for every chain, AlgoBGP.jl:272-294,
// with exchangeMoves! of the previous iteration in its prologue), arranged for the configuration the headline metric is
// quoted on.  Results are bit-identical to the general kernel's (same numerical contract, same arithmetic).
//
//   * ONE tile of 16 chains per workgroup of 1024 lanes (one workgroup per CU at 4096 chains).  The two halves of the
//     workgroup take the moments k = h, h+2, ...: every shock is loaded once per CU and used for 16 chains from a register
//     (32 FP64 adds per 8-byte L2 read; the 8-chain tiles of the general kernel read the matrix twice per CU and the two
//     tiles of a CU finish 4 us apart).  Measured in isolation (tools/sim_bench.hip): 7.0 us against 8.9 us per launch.
//   * the serial bracket lives in registers: a chain is served by FOUR adjacent lanes of the control wave that hold
//     identical state (same loads: the four requests merge), evaluate four proposal tries side by side and store different
//     16-byte pieces of the result blocks straight from registers.  Nothing is staged in LDS except what has to survive
//     the register-hungry simulation (one 16-double line per chain).
// LDS: [exchange walk: chain slots | pair list] theta[16][NP] part[NP][8][16] park[16][PARKW] arrived.
// ------------------------------------------------------------------------------------------
constexpr int NORM_CT = 16;           // chains per tile
constexpr int NORM_NR = 4;            // lanes per chain in the control wave
constexpr int NORM_WG = 2 * WG;       // 1024 lanes
constexpr int NORM_ZU = 4;   // shock rows per chunk (the 16 means sit in SGPRs: 16 accumulators + two chunk buffers = 64 VGPRs)

template <int ZK>
__device__ inline void sim_load_chunk_n(const ZBuf& zb, const KParams& P, int k, int ch, double (&z)[ZK]) {
    const int row0 = (k * P.zstride + ch * (ZK * WG)) * (int)sizeof(double);
#pragma unroll
    for (int u = 0; u < ZK; ++u)
        z[u] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(zb.rsrc, zb.lane_off, row0 + u * WG * (int)sizeof(double), 0));
}

template <int NP>
struct NormLayout {
    static constexpr int RW = (3 + 2 * NP + 1) & ~1;
    static constexpr int HW = (H_PARAMS + 2 * NP + 1) & ~1;
    static constexpr int PARKW = 8 + RW;   // sigma, acc_tuner, u, best, best_id, n_noex, n_acc, partner, record[RW]
    static constexpr size_t doubles = (size_t)NORM_CT * NP + (size_t)NP * 8 * NORM_CT + (size_t)NORM_CT * PARKW + 2 + 64 * (1 + NP) + 2;
};
__host__ __device__ inline size_t norm_tile_doubles(int np) {
    const int RW = (3 + 2 * np + 1) & ~1;
    return (size_t)NORM_CT * np + (size_t)np * 8 * NORM_CT + (size_t)NORM_CT * (8 + RW) + 2 + 64 * (1 + np) + 2;
}

// one of four values by the lane's position in its quad
// (two levels of selects on the bits of r: a ternary chain over double2 becomes a table in scratch memory)
__device__ inline double sel4d(int r, double a, double b, double c, double d) {
    const double x = (r & 1) ? b : a, y = (r & 1) ? d : c;
    return (r & 2) ? y : x;
}
__device__ inline double2 sel4(int r, const double2& a, const double2& b, const double2& c, const double2& d) {
    return make_double2(sel4d(r, a.x, b.x, c.x, d.x), sel4d(r, a.y, b.y, c.y, d.y));
}
// value held by lane q of this lane's quad (DPP quad_perm broadcast, no LDS)
template <int Q>
__device__ inline double quad_bcast(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const int lo = __builtin_amdgcn_mov_dpp((int)(unsigned)u, Q * 0x55, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(unsigned)(u >> 32), Q * 0x55, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
__device__ inline double quad_bcast_dyn(double v, int lane, int q) {   // q: wave-varying source position
    return __shfl(v, (lane & ~3) + q, 64);
}

// the simulation for a tile of 16 chains: half h (512 lanes, l = lane of the half) sums the draws of the moments
// k = h, h+2, ...; lane l takes the draws l, l+512, ... in that order (numerical contract).  s_part [NP][8][16].
template <int NP, int HALVES = 2>   // HALVES: halves of 512 lanes in the workgroup — two (moments h, h + 2, ...) or one (every moment, one after the other)
__device__ inline void simulate_tile16(const KParams& P, const ZBuf& zb0, const double* s_theta, double* s_part, const int h, const int wih,
                                       double (&zc)[NORM_ZU]) {
    constexpr int ZU = NORM_ZU;
    // lane of the half, derived anew from mbcnt and the scalar wave id (nothing of the prologue stays live in a vector register)
    const int l = wih * 64 + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    ZBuf zb;
    zb.rsrc = zb0.rsrc;
    zb.lane_off = l * (int)sizeof(double);
    constexpr int CT = NORM_CT;
    const int ns = P.ns;
    const int nch = (ns + ZU * WG - 1) / (ZU * WG);
    const int last_draws = ns - (nch - 1) * (ZU * WG);
    const bool ragged = last_draws < ZU * WG;
#pragma clang loop unroll(disable)
    for (int k = h; k < NP; k += HALVES) {
        double acc[CT], mu[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            mu[c] = s_theta[c * NP + k];   // (in a scalar register pair it was slower: EXPERIMENTS.md R4.5)
            acc[c] = 0.0;
        }
        auto add_full = [&](const double (&z)[ZU]) {
#pragma unroll
            for (int u = 0; u < ZU; ++u) {
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    const double x = z[u] + mu[c];
                    acc[c] = acc[c] + x;
                }
            }
        };
        auto add_last = [&](const double (&z)[ZU]) {
            if (!ragged) { add_full(z); return; }
            const int ll = wih * 64 + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));   // (not hoisted: see above)
#pragma unroll
            for (int u = 0; u < ZU; ++u) {
                if (ll + u * WG < last_draws) {
#pragma unroll
                    for (int c = 0; c < CT; ++c) {
                        const double x = z[u] + mu[c];
                        acc[c] = acc[c] + x;
                    }
                }
            }
        };
        const int knext = (k + HALVES < NP) ? k + HALVES : k;   // last moment of the half: a harmless reload
        double zn[ZU];
        int ch = 0;
#pragma clang loop unroll(disable)
        for (; ch + 2 <= nch; ch += 2) {
            // a wave that is ahead yields to the ones behind (the arbiter prefers the oldest wave: left alone, the four waves of
            // a SIMD finish one after the other and the last one issues FP64 on its own, at 57 % of the rate)
            if (ch == 0) __builtin_amdgcn_s_setprio(3); else if (ch == 2) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1);
            sim_load_chunk_n<ZU>(zb, P, k, ch + 1, zn);
            add_full(zc);
            const bool last = (ch + 2 == nch);
            sim_load_chunk_n<ZU>(zb, P, last ? knext : k, last ? 0 : ch + 2, zc);
            if (last) add_last(zn); else add_full(zn);
        }
        if (ch < nch) {
            __builtin_amdgcn_s_setprio(0);
            sim_load_chunk_n<ZU>(zb, P, knext, 0, zn);
            add_last(zc);
#pragma unroll
            for (int u = 0; u < ZU; ++u) zc[u] = zn[u];
        }
        // lane id derived anew (mbcnt needs no input register): the loop above has no register to spare for it
        const int lane2 = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        const double tot = wave_reduce_transposed<CT>(acc, lane2);
        if ((lane2 & 3) == 0) s_part[(k * 8 + wih) * CT + (lane2 >> 2)] = tot;
    }
}

// one LDS-DMA instruction: 16 bytes per lane from gsrc (per lane) to LDS byte address lds_dst (wave-uniform) + 16 * lane.
// hipcc does not count it: whoever reads the data waits for vmcnt(0) itself (cdna_hip_programming.md, "LDS-DMA recipe").
__device__ inline void lds_dma16(const void* gsrc, const uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// (the lean exchange walk itself: smm_walk_lean.hpp)
// false (nothing done): this iteration's plan does not fit the form (more than 31 levels), a NaN value is among the chains',
// or the dynamic LDS does not start at address 0 — the caller runs the 16-byte walk.  The result is in LDS (slots at lds);
// only wave 0 may read it without a barrier.
__device__ inline bool exchange_walk_lean(const KParams& P, const int tx, unsigned char* lds, const int tid, const int ts_tile) {
    constexpr int NT = NORM_WG;
    const int Ng = P.Ng;
    const int w = tx - P.plan_t0;
    const uint32_t* __restrict__ g_offp = P.lv_offp + (size_t)w * LV_OFFP;
    const uint4* __restrict__ g_pairs = (const uint4*)(P.lv_pairs_p + (size_t)w * P.plan_Kp);
    const int lane = tid & 63;
    const uint32_t Ng4 = (uint32_t)((Ng + 3) & ~3);
    const uint32_t pbase = 8u * (Ng4 + 4u);   // LDS offset of the pair words
    // one round trip of 16-byte loads: the slots of the chains 4 tid .. 4 tid + 3, the pair words 4 tid .. and 4 (tid + NT) ..
    // (the arrays are padded; words past the padded length are never looked at)
    // (everything is requested before anything is looked at: a load issued behind the first wait is a round trip of its own)
    const uint32_t wflags = P.walk_flags[tid & 3];   // (word 0 is looked at; a per-lane address keeps it a vector load among the others)
    const uint32_t ov = g_offp[min(lane, LV_OFFP - 1)];   // lane l: first word of level l; lane 33: levels; lane 34: the plan fits
    // (round 3, VERDICT r2 #4a) slots and pair list straight into LDS by LDS-DMA, no VGPR hop, no ds_write pass: staging 1.67 ->
    // 1.59 us, kernel -0.03..0.1 us (A/B on one box: small, never negative).  A wave's share
    // is contiguous in memory and in LDS alike (the image is lane-linear): two instructions for its 256 chains' slots, two for its
    // pair words.  Whole 1 KB pieces only: populations that are a multiple of 128 chains (others take the register path below).
    if ((Ng & 127) == 0 && (uint32_t)(size_t)lds == 0u) {
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        const uint4* gs = (const uint4*)P.slot8;
        if (wv * 256 < Ng) lds_dma16(gs + wv * 128 + lane, (uint32_t)(wv * 2048));
        if (wv * 256 + 128 < Ng) lds_dma16(gs + wv * 128 + 64 + lane, (uint32_t)(wv * 2048 + 1024));
        if (4 * (wv * 64) < P.plan_Kp) lds_dma16(g_pairs + tid, pbase + (uint32_t)(wv * 1024));
        if (4 * (wv * 64 + NT) < P.plan_Kp) lds_dma16(g_pairs + tid + NT, pbase + (uint32_t)((wv + 16) * 1024));
        const int nlev_d = __builtin_amdgcn_readlane((int)ov, 33);
        const bool bad = __builtin_amdgcn_readlane((int)ov, 34) == 0 || __builtin_amdgcn_readlane((int)wflags, 0) != 0;
        const int ltail_d = lean_walk_tail(ov, nlev_d, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) ((uint4*)lds)[2 * (Ng4 / 4)] = make_uint4(1u, 0u, 2u, 0u);   // the two dummy slots: keys 1 < 2, "no swap"
        __syncthreads();
        if (bad) return false;
        if (P.ts && tid == 0) P.ts[(size_t)ts_tile * 8 + 5] = wall_clock64();   // staged
        lean_walk_levels<NORM_WG, 0>(P.vals, 1, pbase, ov, nlev_d, tid, ltail_d);
        return true;
    }
    uint4 s0 = make_uint4(0u, 0u, 0u, 0u), s1 = s0;
    if (4 * tid < Ng) { s0 = ((const uint4*)P.slot8)[2 * tid]; s1 = ((const uint4*)P.slot8)[2 * tid + 1]; }
    uint4 p0 = make_uint4(0u, 0u, 0u, 0u), p1 = p0;
    if (4 * tid < P.plan_Kp) p0 = g_pairs[tid];
    if (4 * (tid + NT) < P.plan_Kp) p1 = g_pairs[tid + NT];
    const int nlev = __builtin_amdgcn_readlane((int)ov, 33);
    if (__builtin_amdgcn_readlane((int)ov, 34) == 0 || __builtin_amdgcn_readlane((int)wflags, 0) != 0 || (uint32_t)(size_t)lds != 0u) return false;
    if (4 * tid < Ng) { ((uint4*)lds)[2 * tid] = s0; ((uint4*)lds)[2 * tid + 1] = s1; }
    if (4 * tid < P.plan_Kp) ((uint4*)(lds + pbase))[tid] = p0;
    if (4 * (tid + NT) < P.plan_Kp) ((uint4*)(lds + pbase))[tid + NT] = p1;
    if (tid == 0) ((uint4*)lds)[2 * (Ng4 / 4)] = make_uint4(1u, 0u, 2u, 0u);   // the two dummy slots: keys 1 < 2, "no swap"
    const int ltail = lean_walk_tail(ov, nlev, lane);
    __syncthreads();
    if (P.ts && tid == 0) P.ts[(size_t)ts_tile * 8 + 5] = wall_clock64();   // staged
    if (P.ts && tid == 0 && blockIdx.x == 0) { P.ts[(size_t)8 * 60000 + 40] = clock64(); P.ts[(size_t)8 * 60000 + 41] = wall_clock64(); }
    lean_walk_levels<NORM_WG, 0>(P.vals, 1, pbase, ov, nlev, tid, ltail);
    if (P.ts && tid == 0 && blockIdx.x == 0) {
        P.ts[(size_t)8 * 60000 + 42] = clock64(); P.ts[(size_t)8 * 60000 + 43] = wall_clock64();
        P.ts[(size_t)8 * 60000 + 14] = (unsigned long long)ltail; P.ts[(size_t)8 * 60000 + 7] = (unsigned long long)nlev;
    }
    return true;
}

// the wide form (one min_improve > 0 or NaN for all chains): 16-byte slots {value, src | stamp << 16, -} built here from the
// chains' values (8 bytes per chain from memory, as for the keys).  false: the plan has more than 31 levels.
__device__ inline bool exchange_walk_lean_wide(const KParams& P, const int tx, unsigned char* lds, const int tid, const int ts_tile) {
    constexpr int NT = NORM_WG;
    const int Ng = P.Ng;
    const int w = tx - P.plan_t0;
    const uint32_t* __restrict__ g_offp = P.lv_offp + (size_t)w * LV_OFFP;
    const uint4* __restrict__ g_pairs = (const uint4*)(P.lv_pairs_p + (size_t)w * P.plan_Kp);
    const int lane = tid & 63;
    const uint32_t Ng4 = (uint32_t)((Ng + 3) & ~3);
    const uint32_t pbase = 16u * (Ng4 + 1u);   // LDS offset of the pair words
    const uint32_t ov = g_offp[min(lane, LV_OFFP - 1)];   // lane l: first word of level l; lane 33: levels; lane 34: the plan fits
    constexpr int PT = XLVL_MAX / NT;                     // chains per lane: tid, tid + NT, ... (16-byte LDS writes at a 16-byte lane stride)
    double v_[PT];
#pragma unroll
    for (int r = 0; r < PT; ++r) {
        const int g = tid + r * NT;
        v_[r] = g < Ng ? P.vals[g] : 0.0;
    }
    uint4 p0 = make_uint4(0u, 0u, 0u, 0u), p1 = p0;
    if (4 * tid < P.plan_Kp) p0 = g_pairs[tid];
    if (4 * (tid + NT) < P.plan_Kp) p1 = g_pairs[tid + NT];
    const int nlev = __builtin_amdgcn_readlane((int)ov, 33);
    if (__builtin_amdgcn_readlane((int)ov, 34) == 0 || (uint32_t)(size_t)lds != 0u) return false;
#pragma unroll
    for (int r = 0; r < PT; ++r) {
        const int g = tid + r * NT;
        if (g < Ng) ((uint4*)lds)[g] = make_uint4((uint32_t)__double2loint(v_[r]), (uint32_t)__double2hiint(v_[r]), (uint32_t)g, 0u);
    }
    if (4 * tid < P.plan_Kp) ((uint4*)(lds + pbase))[tid] = p0;
    if (4 * (tid + NT) < P.plan_Kp) ((uint4*)(lds + pbase))[tid + NT] = p1;
    if (tid == 0) ((uint4*)lds)[Ng4] = make_uint4(0u, 0u, 0u, 0u);   // the dummy pair's slot: 0 - 0 > min_improve is false
    const int ltail = lean_walk_tail(ov, nlev, lane);
    __syncthreads();
    if (P.ts && tid == 0) P.ts[(size_t)ts_tile * 8 + 5] = wall_clock64();   // staged
    if (P.lean_unit == 16) lean_walk_levels<NORM_WG, 0, true>(nullptr, 0, pbase, ov, nlev, tid, ltail, P.mi_value);
    else lean_walk_levels<NORM_WG, 1, true>(nullptr, 0, pbase, ov, nlev, tid, ltail, P.mi_value);
    return true;
}

// the lean walk of a SHARD (the inline p2p form, smm_p2p.hpp): the slots of all N_global <= 8192 chains come from this rank's
// window, where every rank's accept step of iteration tx stored them.  Nobody polls anything first: plan and slots are requested
// in one batch and every slot says which iteration it is from (p2p_tag); a lane that finds an older one looks again, past the
// caches, until it is there.  On a key tie the exact values are read from their self-validating copies (P2PWalkValues).
// false: timed out, the plan does not fit the form, or a NaN value.
template <int NMAX>   // the largest population the staging loops serve: XLVL_MAX (4096) or XLDS_MAX (8192)
__device__ inline bool exchange_walk_lean_p2p(const KParams& P, const int tx, unsigned char* lds, const int tid, const int ts_tile) {
    constexpr int NT = NORM_WG;
    constexpr int SR = NMAX / (4 * NT);                                       // rounds of four slots per lane
    constexpr int PR = (NMAX + 64 * LV_MAXLEV + 4 * NT - 1) / (4 * NT);        // rounds of four pair words per lane
    const int Ng = P.Ng;
    const int w = tx - P.plan_t0;
    const uint32_t* __restrict__ g_offp = P.lv_offp + (size_t)w * LV_OFFP;
    const uint4* __restrict__ g_pairs = (const uint4*)(P.lv_pairs_p + (size_t)w * P.plan_Kp);
    const int lane = tid & 63;
    const uint32_t Ng4 = (uint32_t)((Ng + 3) & ~3);
    const uint32_t pbase = 8u * (Ng4 + 4u);   // LDS offset of the pair words
    const unsigned char* mine = P.p2p_self;
    const uint4* g_slots = (const uint4*)(mine + p2p_slot_off(P, tx & 1));
    const uint32_t ov = g_offp[min(lane, LV_OFFP - 1)];   // lane l: first word of level l; lane 33: levels; lane 34: the plan fits
    // (the NaN word carries the publication epoch of the NaN it reports: a word left by an earlier publication — a state since rolled back — means nothing)
    const uint32_t wflags = *(const uint32_t*)(mine + 128 * (size_t)P2P_MAXG + 0 * (size_t)(tid & 1)) == P.p2p_epoch ? 1u : 0u;
    uint4 p_[PR];
#pragma unroll
    for (int r = 0; r < PR; ++r) {
        const int q4 = tid + r * NT;
        p_[r] = 4 * q4 < P.plan_Kp ? g_pairs[q4] : make_uint4(0u, 0u, 0u, 0u);
    }
    // the slots, past the caches (p2p_load16_sys), all in flight together with the plan words above
    uint4 s_[2 * SR];
    {
        const int qc0 = min(tid, (Ng - 1) / 4), qc1 = min(tid + NT, (Ng - 1) / 4);   // (lanes past the end read the last piece: not used)
        if constexpr (SR == 1) p2p_load16x2_sys(g_slots + 2 * qc0, g_slots + 2 * qc0 + 1, s_[0], s_[1]);
        else p2p_load16x4_sys(g_slots + 2 * qc0, g_slots + 2 * qc0 + 1, g_slots + 2 * qc1, g_slots + 2 * qc1 + 1, s_[0], s_[1], s_[2], s_[3]);
    }
    const uint32_t want_hi = p2p_tag(P, tx) << 16;
    bool timed_out = false, gave_up = false;   // (gave_up: the slots never came — timed out, or the run has failed already)
#pragma unroll
    for (int r = 0; r < SR; ++r) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int g = 4 * (tid + r * NT) + 2 * hh;   // the two slots of this 16-byte piece: g, g + 1 (Ng is what it is: the pad slots carry nothing)
            uint4& q = s_[2 * r + hh];
            auto ok = [&]() { return (g >= Ng || (q.y & 0xffff0000u) == want_hi) && (g + 1 >= Ng || (q.w & 0xffff0000u) == want_hi); };
            if (__builtin_expect(!ok(), 0)) {   // somebody's stores are still on their way
                const unsigned long long t0 = wall_clock64();
                do {
                    __builtin_amdgcn_s_sleep(1);
                    q = p2p_load16_sys(g_slots + (g >> 1));
                    if (const int o = p2p_spin_over(P, t0, tx)) { timed_out = timed_out || o == 1; gave_up = true; break; }
                } while (!ok());
            }
        }
    }
    (void)gave_up;
    if (timed_out) report_error(P, 3, tx + 1, P.offset);   // (everybody goes on to the barriers below; the failure is reported)
    const int nlev = __builtin_amdgcn_readlane((int)ov, 33);
    bool form_ok = __builtin_amdgcn_readlane((int)ov, 34) != 0 && __builtin_amdgcn_readfirstlane((int)wflags) == 0 && (uint32_t)(size_t)lds == 0u;
    {   // a NaN value: its slot says so itself (P2P_KEY_NAN arrives with the tag that validated the word; the window's NaN word, read at
        // the top, may lag behind the slots).  Every wave must decide the same: a byte per wave in the two spare slots behind the dummy pair's
        bool nan_seen = false;
#pragma unroll
        for (int r = 0; r < SR; ++r) {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int g = 4 * (tid + r * NT) + 2 * hh;
                const uint4 q = s_[2 * r + hh];
                nan_seen = nan_seen || (g < Ng && q.x == P2P_KEY_NAN) || (g + 1 < Ng && q.z == P2P_KEY_NAN);
            }
        }
        const bool wnan = __ballot(nan_seen) != 0ull;
        if (lane == 0) lds[8u * (Ng4 + 2u) + (uint32_t)(tid >> 6)] = wnan ? (unsigned char)1 : (unsigned char)0;
    }
#pragma unroll
    for (int r = 0; r < SR; ++r) {
        const int q = tid + r * NT;
        if (4 * q < Ng) {
            ((uint4*)lds)[2 * q] = make_uint4(s_[2 * r].x, s_[2 * r].y & 0xffffu, s_[2 * r].z, s_[2 * r].w & 0xffffu);
            ((uint4*)lds)[2 * q + 1] = make_uint4(s_[2 * r + 1].x, s_[2 * r + 1].y & 0xffffu, s_[2 * r + 1].z, s_[2 * r + 1].w & 0xffffu);
        }
    }
#pragma unroll
    for (int r = 0; r < PR; ++r) {
        const int q4 = tid + r * NT;
        if (4 * q4 < P.plan_Kp) ((uint4*)(lds + pbase))[q4] = p_[r];
    }
    if (tid == 0) ((uint4*)lds)[2 * (Ng4 / 4)] = make_uint4(1u, 0u, 2u, 0u);   // the two dummy slots: keys 1 < 2, "no swap"
    const int ltail = lean_walk_tail(ov, nlev, lane);
    __syncthreads();
    if (P.ts && tid == 0) P.ts[(size_t)ts_tile * 8 + 5] = wall_clock64();   // staged
    {
        const uint4 nf = *(const uint4*)(lds + 8u * (Ng4 + 2u));   // the 16 waves' bytes
        if ((nf.x | nf.y | nf.z | nf.w) != 0u) form_ok = false;
    }
    if (!form_ok) return false;   // (wave-uniform AND the same in every wave: the plan is what it is for the whole launch, the NaN bytes are read behind the barrier)
    const P2PWalkValues values{P, tx};
    if (NMAX <= XLVL_MAX || P.lean_unit == 8) lean_walk_levels<NORM_WG, 0, false, P2PWalkValues>(nullptr, 1, pbase, ov, nlev, tid, ltail, 0.0, values);
    else lean_walk_levels<NORM_WG, 1, false, P2PWalkValues>(nullptr, 1, pbase, ov, nlev, tid, ltail, 0.0, values);
    return true;
}

template <int NP, bool P2P, bool BIG>
__device__ inline void epilogue_norm(const KParams& P, const int t, double* __restrict__ rec_out, const double* s_theta, const double* s_part,
                                     const double* s_park, const int tile, const int tid);

// (WIDE: the walk's form for one min_improve > 0 shared by all chains — a kernel of its own, k_chain_iter_norm_wide, so that
// the registers of the min_improve == 0 kernel stay what they are)
// (LEAN: the walk is one of the lean forms and nothing else — the host launches k_chain_iter_norm_any where a plan of more
// than 31 levels or a NaN value may turn up (injected pair lists, uploaded states; it knows both): with both walks in one
// kernel the headline kernel spilled 8 scalar registers in its latency-bound prologue and took 0.3 us longer)
// (P2P: a shard of the p2p form, smm_p2p.hpp — records, values and walk slots of ALL chains live in this rank's window, the accept
// step stores its results into every rank's window and arrives; WALK then means "when the launch says so", F_WALK_INLINE)
// (CONEL: no walk of the whole list, but — when the launch says F_WALK_INLINE — the tile's own, locally numbered cone: smm_cone.hpp)
template <int NP, bool WALK, bool WIDE, bool LEAN, bool P2P = false, bool P2P_BIG = true, int HALVES = 2, bool CONEL = false>
__device__ __forceinline__ void chain_iter_norm_body(const KParams& P, const int t, const double* __restrict__ rec_in_arg,
                                                     double* __restrict__ rec_out, const int flags) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    using L = NormLayout<NP>;
    constexpr int CT = NORM_CT, RW = L::RW, HW = L::HW, PARKW = L::PARKW;
    constexpr int NPC = RW / 2, NPH = HW / 2;   // 16-byte pieces of a record / a history row
    const int tid = (int)threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave of the workgroup (scalar)
    const int h = wave >> 3;                                      // half of the workgroup
    const int l = tid & (WG - 1), lane = tid & 63;
    const int tile = (int)blockIdx.x;
    const int N = P.N;
    double* s_theta = smem + P.tile_off;                  // [CT][NP]
    double* s_part = s_theta + CT * NP;                   // [NP][8][CT]
    double* s_park = s_part + NP * 8 * CT;                // [CT][PARKW]
    unsigned* s_arrived = (unsigned*)(s_park + CT * PARKW);   // + 1: the poisoned flag
    double* s_rng = s_park + CT * PARKW + 2;               // [64][1 + NP]: u and the normals of try r, made by wave 1
    unsigned* s_rng_ready = (unsigned*)(s_rng + 64 * (1 + NP));
    const bool ctl = tid < 64;
    const int cl = lane >> 2, r = lane & 3;               // control wave: chain of the tile, position in its quad
    const int c = tile * CT + cl;
    const bool valid = ctl && c < N;
    const int gc = P.offset + c;
    const int goff = (flags & F_GLOBAL_REC) ? 0 : P.offset;   // rec_in indexed by global chain id (all-gathered buffer)?
    const double* __restrict__ rec_in = rec_in_arg;   // (P2P: the records come out of this rank's window, below)
    const bool walk_now = WALK && (!P2P || (flags & F_WALK_INLINE));
    TS_MARK(0);

    // ---- global reads that do not depend on the exchange, all issued before anything waits ----
    double za[NORM_ZU];
    ZBuf zb;
    const bool simw = h < NP;   // NP == 1: the second half has no moment
    if (simw) { zb.init(P, l); sim_load_chunk_n<NORM_ZU>(zb, P, h, 0, za); }
    unsigned long long err_word = ERR_NONE;
    double2 csq[6];             // the chain state block, fields 0..11
    double u = 0.0, zA[NP], zB[NP];
    unsigned long long xr = (unsigned long long)(unsigned)gc;
#pragma unroll
    for (int k = 0; k < NP; ++k) { zA[k] = 0.0; zB[k] = 0.0; }
    // (a plain load, looked at after the walk: a volatile one is a system-scope round trip the control wave waits for at once,
    // before it has requested anything else — and the staging barrier waits for the control wave)
    if (ctl) err_word = *(const unsigned long long*)P.err;
    // this iteration's randomness (the MH uniform, the normals of the tries this lane evaluates): generated right here unless
    // tables are injected — the counter generator needs nothing but (seed, chain, iteration, try), the control wave would
    // otherwise just wait for the exchange inputs, and 144 bytes per chain and iteration need not be written and read back
    const bool rng_here = !P.user_ntab && !P.user_utab;
    const int rb_tries = rng_here ? NORM_NR : P.rb_tries;   // tries held in registers or memory; later ones come from the generator
    if (valid) {
        const double2* g_cs = (const double2*)(P.cs + (size_t)c * CSW);
#pragma unroll
        for (int i = 0; i < 6; ++i) csq[i] = g_cs[i];
        if (t > 1 && !rng_here) {
            const double* g_rb = P.rb + ((size_t)(t - P.rb_t0) * N + c) * P.RBW;
            u = g_rb[0];
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                if (r < P.rb_tries) zA[k] = g_rb[1 + r * NP + k];
                if (NORM_NR + r < P.rb_tries) zB[k] = g_rb[1 + (NORM_NR + r) * NP + k];
            }
        }
        if (!walk_now && !(CONEL && (flags & F_WALK_INLINE)) && (flags & F_HAS_PENDING)) xr = P.xres[gc];
    }
    if constexpr (CONEL) {
        __builtin_amdgcn_s_setprio(1);   // (ahead of the waves of the next window's plan kernels, which run beside this kernel: smmhip.hip, plan_window_into)
        // exchangeMoves! of iteration t-1 (AlgoBGP.jl:647-716) over the tile's cone, while those loads are in flight; its slots and pair
        // words lie UNDER the tile's blocks: nothing of the tile has been written yet
        if (flags & F_WALK_INLINE)
            if (!exchange_walk_cone_local<(WG * HALVES)>(P, t - 1, (unsigned char*)smem, tid, valid, cl, xr, tile) && tid == 0) report_error(P, 3, t, gc);
    }
    // (LDS survives from workgroup to workgroup: the hand-over flags are reset before anybody can look at them — the walk's first
    // barrier, or the one below, orders the reset)
    if (tid == 64) { *s_arrived = 0u; *s_rng_ready = 0u; }
    if (!walk_now) __syncthreads();

    uint32_t kmeta = 0u;   // lean walk: src | stamp << 16 of the chain's slot (the partner is looked up while the record is on its way)
    if (walk_now) {
        // exchangeMoves! of iteration t-1 (AlgoBGP.jl:647-716), by all lanes of the workgroup, while those loads are in flight
        if constexpr (LEAN) {
            bool lean;
            if constexpr (P2P) lean = exchange_walk_lean_p2p<P2P_BIG ? XLDS_MAX : XLVL_MAX>(P, t - 1, (unsigned char*)smem, tid, tile);
            else if constexpr (WIDE) lean = exchange_walk_lean_wide(P, t - 1, (unsigned char*)smem, tid, tile);
            else lean = exchange_walk_lean(P, t - 1, (unsigned char*)smem, tid, tile);
            if (!lean) {   // (cannot happen: the host launches k_chain_iter_norm_any wherever it can; loud if it does)
                if (tid == 0) report_error(P, 3, t, gc);
            } else if (valid) {
                kmeta = WIDE ? ((const uint4*)smem)[gc].z : ((const uint2*)smem)[gc].y;
                xr = (unsigned long long)(kmeta & 0xffffu);
            }
        } else {
            exchange_walk_fast<NORM_WG, false>(P, t - 1, (unsigned char*)smem, tid, tile);
            if (valid) {
                const XSlot sv = ((const XSlot*)smem)[gc];
                xr = (unsigned long long)sv.src | ((unsigned long long)sv.partner << 32);
            }
        }
    }
    // This iteration's randomness, by wave 1 (lane = the control wave's lane: chain, try): it has nothing to do from here to the
    // simulation, and the control wave needs the numbers only once the record it continues from has arrived (~1 us from now).
    if (rng_here && t > 1 && tid >= 64 && tid < 128) {
        const int c1 = tile * CT + (lane >> 2);
        if (c1 < N) {
            const uint32_t g1 = (uint32_t)(P.offset + c1);
            double* o = s_rng + lane * (1 + NP);
            o[0] = rng_u(P.seed, g1, (uint32_t)t);                     // probs_acc[iter], AlgoBGP.jl:85
#pragma unroll
            for (int q = 0; 2 * q < NP; ++q) {                          // rand(RAND, d) of try r, :404
                double z0, z1;
                rng_prop_normal2(P.seed, g1, (uint32_t)t, (uint32_t)(lane & 3), (uint32_t)q, z0, z1);
                o[1 + 2 * q] = z0;
                if (2 * q + 1 < NP) o[1 + 2 * q + 1] = z1;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_store(s_rng_ready, (unsigned)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    TS_MARK(1);

    // ---- control wave: the record the chain continues from, settle iteration t-1, propose (AlgoBGP.jl:424-471) ----
    asm volatile("" : "+v"(err_word));   // (looked at only here: the compiler would otherwise wait for it right behind the load)
    const bool poisoned = error_before(err_word, t);   // an earlier iteration raised a hard error: nothing is stored any more
    if (ctl) {
        if (tid == 0) s_park[CT * PARKW + 1] = poisoned ? 1.0 : 0.0;
        int partner = (int)(xr >> 32);
        double mu01[NP], th[NP];
        const double sigma = csq[0].x;
        if (rng_here && t > 1) {   // wave 1's numbers (the stamp is the iteration: nothing to reset between launches... LDS is per launch anyway)
            while (__hip_atomic_load(s_rng_ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != (unsigned)t) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const double* o = s_rng + lane * (1 + NP);
            u = o[0];
#pragma unroll
            for (int k = 0; k < NP; ++k) zA[k] = o[1 + k];
        }
        {
            // its own record, or its donor's (swap_ev_ij!, :734-749): the one dependent memory level of the iteration
            double rc[RW];
            if (valid) {
                const int s = (int)(unsigned)(xr & 0xffffffffu) - goff;
                if constexpr (P2P) {   // the self-validating record of iteration t-1 out of this rank's window (its own chain's, or the donor's)
                    const uint4* g_ll = (const uint4*)(P.p2p_self + p2p_llrec_off(P, (t - 1) & 1) + (size_t)s * RW * 16);
                    const uint32_t tag = p2p_tag(P, t - 1);
                    uint4 q[RW];
                    static_assert(RW % 2 == 0, "records are an even number of doubles");
#pragma unroll
                    for (int i = 0; i + 4 <= RW; i += 4) p2p_load16x4_sys(g_ll + i, g_ll + i + 1, g_ll + i + 2, g_ll + i + 3, q[i], q[i + 1], q[i + 2], q[i + 3]);
                    if constexpr (RW % 4 != 0) p2p_load16x2_sys(g_ll + RW - 2, g_ll + RW - 1, q[RW - 2], q[RW - 1]);
                    bool ok = true;
#pragma unroll
                    for (int i = 0; i < RW; ++i) ok = ok && p2p_ll_ok(q[i], tag);
                    if (__builtin_expect(!ok, 0)) {   // still on its way: look again, past the caches
                        const unsigned long long t0 = wall_clock64();
                        do {
                            __builtin_amdgcn_s_sleep(1);
                            ok = true;
#pragma unroll
                            for (int i = 0; i < RW; ++i) { q[i] = p2p_load16_sys(g_ll + i); ok = ok && p2p_ll_ok(q[i], tag); }
                            if (const int o = p2p_spin_over(P, t0, t - 1)) { if (o == 1) report_error(P, 3, t, gc); break; }
                        } while (!ok);
                    }
#pragma unroll
                    for (int i = 0; i < RW; ++i) rc[i] = p2p_ll_double(q[i]);
                } else {
                    const double2* g_rec = (const double2*)(rec_in + (size_t)s * RW);
#pragma unroll
                    for (int i = 0; i < NPC; ++i) { const double2 q = g_rec[i]; rc[2 * i] = q.x; rc[2 * i + 1] = q.y; }
                }
                if (WALK && LEAN && (kmeta >> 16))   // set_exchanged!, :747-748: from the pair word the swap stamped into the slot
                    partner = (P2P && P2P_BIG && P.lean_unit != 8) ? (int)lean_partner<1>((const unsigned char*)smem, 8u * ((uint32_t)((P.Ng + 3) & ~3) + 4u), kmeta, (uint32_t)gc)
                              : !WIDE ? (int)lean_partner<0>((const unsigned char*)smem, 8u * ((uint32_t)((P.Ng + 3) & ~3) + 4u), kmeta, (uint32_t)gc)
                              : P.lean_unit == 16 ? (int)lean_partner<0, 4>((const unsigned char*)smem, 16u * ((uint32_t)((P.Ng + 3) & ~3) + 1u), kmeta, (uint32_t)gc)
                                                  : (int)lean_partner<1, 4>((const unsigned char*)smem, 16u * ((uint32_t)((P.Ng + 3) & ~3) + 1u), kmeta, (uint32_t)gc);
            } else {
#pragma unroll
                for (int f = 0; f < RW; ++f) rc[f] = 0.0;
            }
            int nn = (int)csq[1].x, na = (int)csq[1].y;
            double bp = csq[3].x, bpid = csq[3].y;
            if (valid && t > 1) {
                bool exch_prev = false;
                if (partner != 0) {
                    // set_eval!(ci, ej) of swap_ev_ij! as a history record: the chain's record of iteration t-1 is the donor's last
                    // accepted one (accepted = true, the donor's prob/status), curr = donor value, best against iteration t-2 (:231-243)
                    exch_prev = true;
                    const double value = rc[0];
                    if (value < csq[4].x) { bp = value; bpid = (double)(t - 1); }
                    else { bp = csq[4].x; bpid = csq[4].y; }
                    if (!poisoned) {
                        double hv[HW];
                        hv[H_VALUE] = value; hv[H_PROB] = rc[1]; hv[H_CURR] = value; hv[H_BEST] = bp; hv[H_BESTID] = bpid;
                        hv[H_EXCH] = (double)partner; hv[H_ACC] = 1.0; hv[H_STATUS] = rc[2];
#pragma unroll
                        for (int k = 0; k < 2 * NP; ++k) hv[H_PARAMS + k] = rc[3 + k];
                        if (HW > H_PARAMS + 2 * NP) hv[HW - 1] = 0.0;
                        double2* g_h = (double2*)(P.hrec + ((size_t)(t - 2) * N + c) * HW);
#pragma unroll
                        for (int j = 0; 4 * j < NPH; ++j) {
                            const int i = 4 * j + r;
                            const double2 v = sel4(r, make_double2(hv[8 * j], hv[8 * j + 1]),
                                                   make_double2(hv[(8 * j + 2) % HW], hv[(8 * j + 3) % HW]),
                                                   make_double2(hv[(8 * j + 4) % HW], hv[(8 * j + 5) % HW]),
                                                   make_double2(hv[(8 * j + 6) % HW], hv[(8 * j + 7) % HW]));
                            if (i < NPH) g_h[i] = v;
                        }
                    }
                } else if (csq[2].y != 0.0) {   // sharded three-phase path: k_exch_apply already rewrote record and history
                    exch_prev = true;
                }
                if ((flags & F_CLOSE_PREV) && !exch_prev) { nn += 1; na += (int)csq[2].x; }   // set_acceptRate!, :253-257
            }
            // park what the epilogue needs (one line per chain; nothing of it stays in registers across the simulation)
            if (r == 0) {
                double* pk = s_park + cl * PARKW;
                pk[0] = sigma; pk[1] = csq[5].x; pk[2] = u; pk[3] = bp; pk[4] = bpid; pk[5] = (double)nn; pk[6] = (double)na;
                pk[7] = (double)partner;
#pragma unroll
                for (int f = 0; f < RW; ++f) pk[8 + f] = rc[f];
            }
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                th[k] = !valid ? 0.0 : (t == 1 ? P.init[k] : rc[3 + k]);
                mu01[k] = (rc[3 + k] - P.lb[k]) / (P.ub[k] - P.lb[k]);   // mapto_01, mprob.jl:248
            }
        }
        // proposal: lane r evaluates try j0 + r; the chain's first try inside the unit box wins (mysample, :400-410)
        if (t > 1) {
            const int max_tries = P.user_n ? min(P.rb_tries, P.smpl_iters) : P.smpl_iters;
            bool found = !valid;
            const double* g_rb = P.rb + ((size_t)(t - P.rb_t0) * N + (valid ? c : 0)) * P.RBW;
            for (int j0 = 0; __any(!found) && j0 < max_tries; j0 += NORM_NR) {
                const int j = j0 + r;
                double x[NP], z[NP];
#pragma unroll
                for (int k = 0; k < NP; ++k) { x[k] = 0.0; z[k] = 0.0; }
                bool ok = !found && j < max_tries;
                if (ok) {
                    if (j0 == 0) {
#pragma unroll
                        for (int k = 0; k < NP; ++k) z[k] = zA[k];
                    } else if (j0 == NORM_NR && !rng_here) {
#pragma unroll
                        for (int k = 0; k < NP; ++k) z[k] = zB[k];
                    } else if (j < rb_tries) {
#pragma unroll
                        for (int k = 0; k < NP; ++k) z[k] = g_rb[1 + j * NP + k];
                    }
                    if (j >= rb_tries) {   // past the pre-generated tries: the in-kernel generator (never with injected normals)
#pragma unroll
                        for (int q = 0; 2 * q < NP; ++q) {
                            const double2 zz2 = rng_prop_normal2_outofline(P.seed, (uint32_t)gc, (uint32_t)t, (uint32_t)j, (uint32_t)q);
                            z[2 * q] = zz2.x;
                            if (2 * q + 1 < NP) z[2 * q + 1] = zz2.y;
                        }
                    }
#pragma unroll
                    for (int k = 0; k < NP; ++k) {
                        const double step = sigma * z[k];   // MvNormal(mu01, sigma): x = mu + sigma*z
                        x[k] = mu01[k] + step;
                        if (!(x[k] >= 0.0 && x[k] <= 1.0)) ok = false;   // inclusive bounds, :405
                    }
                }
                const unsigned long long m = __ballot(ok);
                const unsigned quad = (unsigned)(m >> (lane & ~3)) & 0xfu;
                if (!found && quad) {   // (found and quad are the same in the four lanes of a chain: its shuffles run with the whole quad active)
                    const int rwin = __builtin_ctz(quad);
#pragma unroll
                    for (int k = 0; k < NP; ++k) {
                        const double lbk = P.lb[k];
                        const double sc = x[k] * (P.ub[k] - lbk);
                        const double thk = sc + lbk;   // mapto_ab, mprob.jl:271
                        th[k] = quad_bcast_dyn(thk, lane, rwin);
                    }
                    found = true;
                }
            }
            if (!found && r == 0) report_error(P, ERRK_NO_DRAW, t, gc);   // :409
        }
        if (r == 0) {
#pragma unroll
            for (int k = 0; k < NP; ++k) s_theta[cl * NP + k] = th[k];
        }
    }
    TS_MARK(6);
    __syncthreads();
    TS_MARK(2);

    // ---- simulation: every lane, ns draws x its half's moments x 16 chains ----
    if (simw) simulate_tile16<NP, HALVES>(P, zb, s_theta, s_part, h, wave & 7, za);
    // No workgroup barrier: only the control wave consumes the partial sums.  Every wave announces its partials with one
    // LDS add and is done; the control wave waits for the announcements of the waves that had a moment.
    // (lane ids are derived anew from mbcnt and the scalar wave id: no register of the prologue stays live across the simulation)
    const int tid2 = wave * 64 + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (simw && (tid2 & 63) == 0) __hip_atomic_fetch_add(s_arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (tid2 >= 64) return;
    {
        const unsigned want = (unsigned)(8 * (NP < HALVES ? NP : HALVES));
        while (__hip_atomic_load(s_arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < want) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    if (P.ts && tid2 == 0) P.ts[(size_t)tile * 8 + 3] = wall_clock64();
    // Epilogue by the control wave: everything it needs comes from LDS (parked by the prologue)
    epilogue_norm<NP, P2P, !WALK>(P, t, rec_out, s_theta, s_part, s_park, tile, tid2);
}
template <int NP, bool WALK>
__global__ __launch_bounds__(NORM_WG, 4) void k_chain_iter_norm(const KParams P, const int t, const double* __restrict__ rec_in,
                                                                 double* __restrict__ rec_out, const int flags) {
    chain_iter_norm_body<NP, WALK, false, true>(P, t, rec_in, rec_out, flags);
}
// the kernel without the walk on workgroups of ONE half (512 lanes: the tile's moments one after the other, each over the same 512
// lanes in the same order — the numerical contract does not change): two workgroups share a CU, and one's serial prologue and
// epilogue (3.6 of a tile's 10.9 us) run under the other's simulation.  For shards of more than one round of tiles (> 4096 chains).
template <int NP>
__global__ __launch_bounds__(NORM_WG / 2, 4) void k_chain_iter_norm_narrow(const KParams P, const int t, const double* __restrict__ rec_in,
                                                                           double* __restrict__ rec_out, const int flags) {
    chain_iter_norm_body<NP, false, false, true, false, true, 1>(P, t, rec_in, rec_out, flags);
}
// ... with the exchange of the last iteration walked by every tile over its own cone (large single shards: smm_cone_big.hpp)
template <int NP>
__global__ __launch_bounds__(NORM_WG / 2, 4) void k_chain_iter_norm_narrow_cone(const KParams P, const int t, const double* __restrict__ rec_in,
                                                                                double* __restrict__ rec_out, const int flags) {
    chain_iter_norm_body<NP, false, false, true, false, true, 1, true>(P, t, rec_in, rec_out, flags);
}
template <int NP>
__global__ __launch_bounds__(NORM_WG, 4) void k_chain_iter_norm_wide(const KParams P, const int t, const double* __restrict__ rec_in,
                                                                      double* __restrict__ rec_out, const int flags) {
    chain_iter_norm_body<NP, true, true, true>(P, t, rec_in, rec_out, flags);
}
// the walk on 16-byte slots {value, src, partner} (exchange_walk_fast): any thresholds, any plan depth, NaN values
template <int NP>
__global__ __launch_bounds__(NORM_WG, 4) void k_chain_iter_norm_any(const KParams P, const int t, const double* __restrict__ rec_in,
                                                                     double* __restrict__ rec_out, const int flags) {
    chain_iter_norm_body<NP, true, false, false>(P, t, rec_in, rec_out, flags);
}

// a shard of the p2p form: the lean key walk inline when the launch says so (N_global <= 8192), results into every rank's window
template <int NP, bool BIG>
__global__ __launch_bounds__(NORM_WG, 4) void k_chain_iter_norm_p2p(const KParams P, const int t, const double* __restrict__ rec_in,
                                                                     double* __restrict__ rec_out, const int flags) {
    chain_iter_norm_body<NP, true, false, true, true, BIG>(P, t, rec_in, rec_out, flags);
}

// a shard of the rows form (8192 < N_global <= 32768): no walk in this kernel at all — k_exch_resolve_rows<., true> has resolved the
// exchange from the window's 4-byte slots, which this kernel's accept step stores
template <int NP>
__global__ __launch_bounds__(NORM_WG, 4) void k_chain_iter_norm_p2p_rows(const KParams P, const int t, const double* __restrict__ rec_in,
                                                                          double* __restrict__ rec_out, const int flags) {
    chain_iter_norm_body<NP, false, false, true, true, true>(P, t, rec_in, rec_out, flags);
}

// ... on half-size workgroups, two to a CU, where the shard is more than one round of tiles (like k_chain_iter_norm_narrow)
template <int NP>
__global__ __launch_bounds__(NORM_WG / 2, 4) void k_chain_iter_norm_p2p_rows_narrow(const KParams P, const int t, const double* __restrict__ rec_in,
                                                                                    double* __restrict__ rec_out, const int flags) {
    chain_iter_norm_body<NP, false, false, true, true, true, 1>(P, t, rec_in, rec_out, flags);
}

// objective value (ObjExamples.jl:79-110), doAcceptReject! (:324-392), set_eval! (:220-245) and the result blocks
// (BIG: the kernel without an inline walk — the one large populations run: their stand-alone resolution takes its initial slots
// from the accept step when the host says so, KParams::slots17_out)
template <int NP, bool P2P, bool BIG>
__device__ inline void epilogue_norm(const KParams& P, const int t, double* __restrict__ rec_out, const double* s_theta,
                                     const double* s_part, const double* s_park, const int tile, const int tid) {
    using L = NormLayout<NP>;
    constexpr int CT = NORM_CT, RW = L::RW, HW = L::HW, PARKW = L::PARKW;
    constexpr int NPC = RW / 2, NPH = HW / 2;
    const int lane = tid & 63, cl = lane >> 2, r = lane & 3;
    const int c = tile * CT + cl, N = P.N;
    const int gc = P.offset + c;
    if (c >= N) return;
    const double* pk = s_park + cl * PARKW;
    if (s_park[CT * PARKW + 1] != 0.0) return;   // poisoned (parked by the prologue)

    const double sig = pk[0], atun = pk[1], uu = pk[2], bp = pk[3], bpid = pk[4];
    const int nn = (int)pk[5], na = (int)pk[6];
    double rc[RW], th[NP], sm[NP];
#pragma unroll
    for (int f = 0; f < RW; ++f) rc[f] = pk[8 + f];
#pragma unroll
    for (int k = 0; k < NP; ++k) th[k] = s_theta[cl * NP + k];
    double value;
    int status;
    if (P.obj == SMM_OBJ_NORM_FAILBOX && P.objp && th[0] >= P.objp[0] && th[0] <= P.objp[1]) {   // "exception": mprob.jl:183-186
#pragma unroll
        for (int k = 0; k < NP; ++k) sm[k] = NAN;
        value = -1.0;   // Eval() default, Eval.jl:84
        status = -2;
    } else {
        // lane r finishes moment r: wave totals left to right, mean, weighted deviation; the squares are added in moment order
        double mk = 0.0, vk = 0.0;
        if (r < NP) {
            double tot = s_part[(r * 8 + 0) * CT + cl];
#pragma unroll
            for (int wv = 1; wv < 8; ++wv) tot = tot + s_part[(r * 8 + wv) * CT + cl];
            mk = tot / (double)P.ns;
            double d = mk - P.mom[r];
            const double wk = P.w[r];
            if (!isnan(wk)) d = d / wk;
            vk = d * d;
        }
        double vsum = 0.0;
        {
            const double m0 = quad_bcast<0>(mk), v0 = quad_bcast<0>(vk);
            sm[0] = m0; vsum = v0;
            if constexpr (NP > 1) { const double m1 = quad_bcast<1>(mk), v1 = quad_bcast<1>(vk); sm[1] = m1; vsum = vsum + v1; }
            if constexpr (NP > 2) { const double m2 = quad_bcast<2>(mk), v2 = quad_bcast<2>(vk); sm[2] = m2; vsum = vsum + v2; }
            if constexpr (NP > 3) { const double m3 = quad_bcast<3>(mk), v3 = quad_bcast<3>(vk); sm[3] = m3; vsum = vsum + v3; }
        }
        value = vsum / (double)NP;
        status = 1;
    }
    const double old = rc[0];
    double prob;
    bool acc;
    if (t == 1) {   // :326-332
        prob = 1.0; acc = true; status = 1;
    } else if (status < 0) {   // :336-338
        prob = 0.0; acc = false;
    } else {
        if (!(value >= 0.0) && r == 0) report_error(P, ERRK_NEGATIVE, t, gc);   // :341
        const double e = smm_exp(atun * (old - value));   // (the contract exponential, smm_rng.hpp)
        prob = (e != e) ? e : (e < 1.0 ? e : 1.0);   // minimum([1.0,e]), NaN propagates (:344)
        if (!isfinite(prob)) { prob = 0.0; acc = false; status = -1; }   // :350-353
        else if (!isfinite(old)) { prob = 1.0; acc = true; }             // :355-359
        else { status = 1; acc = prob > uu; }                            // strict >, :362-367
    }
    TS_MARK(7);
    const double rate = (double)(na + (acc ? 1 : 0)) / (double)(nn + 1);   // set_acceptRate!, :253-257
    double nsig = sig;
    if (t > 1 && (t % P.sigma_update_steps) == 0)   // :381-390
        nsig = (rate > 0.234) ? sig * (1.0 + P.sigma_adjust_by) : sig * (1.0 - P.sigma_adjust_by);
    double bestv, currv, bestid;   // set_eval!, :220-245
    if (t == 1) { bestv = value; currv = value; bestid = 1.0; }
    else {
        currv = acc ? value : old;
        if (value < bp) { bestv = value; bestid = (double)t; }
        else { bestv = bp; bestid = bpid; }
    }
    const int pb = t & 1;
    if (r == 0) {
        const double v = acc ? value : old;
        if constexpr (P2P) {   // value and walk slot into every rank's window (its own included)
#pragma unroll
            for (int p = 0; p < P2P_MAXG; ++p)
                if (p < P.p2p_G) {
                    unsigned char* w = P.p2p_win[p];
                    const unsigned long long vb = __builtin_bit_cast(unsigned long long, v);
                    const p2p_u32x4 qv = {(unsigned)vb, p2p_tag(P, t), (unsigned)(vb >> 32), p2p_tag(P, t)};
                    p2p_store16u((uint4*)(w + p2p_llval_off(P, pb)) + gc, qv);
                    if constexpr (BIG) p2p_store4((uint32_t*)(w + p2p_slot4_off(P, pb)) + gc, p2p_slot4_word(P, v, t));   // (rows form: the kernel without the walk)
                    else p2p_store8((uint2*)(w + p2p_slot_off(P, pb)) + gc, p2p_slot_word(P, v, (uint32_t)gc, t));
                    if (v != v) __hip_atomic_fetch_max((uint32_t*)(w + 128 * (size_t)P2P_MAXG), P.p2p_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
        } else {
            P.vals_out[c] = v;
            if constexpr (BIG) {
                if (P.slots17_out) {   // the chain's initial slots of k_exch_resolve_rows / _key (what k_exch_keys would make of vals[c])
                    P.slots17_out[c] = (uint32_t)gc | (order_key17(v) << 15);
                    if (v != v) atomicOr(P.nan_flags_out, 1u);
                }
            }
            if (P.slot8_out) {   // the chain's slot at the start of the next exchange walk (exchange_walk_lean)
                P.slot8_out[c] = make_uint2(order_key32(v), (uint32_t)gc);
                if (v != v) atomicOr(P.walk_flags, 1u);
            }
        }
    }
    // ---- result blocks, straight from registers: lane r of the quad stores the 16-byte pieces r, r+4, ... ----
    {
        double2* g_cs = (double2*)(P.cs + (size_t)c * CSW);
        const double accd = acc ? 1.0 : 0.0;
        g_cs[r] = sel4(r, make_double2(nsig, rate), make_double2((double)nn, (double)na), make_double2(accd, 0.0), make_double2(bestv, bestid));
        if (r < 2) g_cs[4 + r] = r == 0 ? make_double2(bp, bpid) : make_double2(atun, pk[7]);
        // the chain's last accepted record (lastAccepted :209-215) = input of the exchange step
        double nr[RW];
        nr[0] = acc ? value : rc[0]; nr[1] = acc ? prob : rc[1]; nr[2] = acc ? (double)status : rc[2];
#pragma unroll
        for (int k = 0; k < NP; ++k) { nr[3 + k] = acc ? th[k] : rc[3 + k]; nr[3 + NP + k] = acc ? sm[k] : rc[3 + NP + k]; }
        if (RW > 3 + 2 * NP) nr[RW - 1] = 0.0;
#pragma unroll
        for (int j = 0; 4 * j < NPC; ++j) {
            const int i = 4 * j + r;
            const double2 v = sel4(r, make_double2(nr[8 * j], nr[8 * j + 1]), make_double2(nr[(8 * j + 2) % RW], nr[(8 * j + 3) % RW]),
                                   make_double2(nr[(8 * j + 4) % RW], nr[(8 * j + 5) % RW]), make_double2(nr[(8 * j + 6) % RW], nr[(8 * j + 7) % RW]));
            if constexpr (P2P) {   // into every rank's window, global chain order
#pragma unroll
                for (int p = 0; p < P2P_MAXG; ++p)
                    if (p < P.p2p_G && i < NPC) p2p_store_ll(P.p2p_win[p] + p2p_llrec_off(P, pb) + ((size_t)gc * NPC + i) * 32, v, p2p_tag(P, t));
            } else {
                if (i < NPC) ((double2*)(rec_out + (size_t)c * RW))[i] = v;
            }
        }
        double hv[HW];
        hv[H_VALUE] = value; hv[H_PROB] = prob; hv[H_CURR] = currv; hv[H_BEST] = bestv; hv[H_BESTID] = bestid;
        hv[H_EXCH] = 0.0; hv[H_ACC] = accd; hv[H_STATUS] = (double)status;
#pragma unroll
        for (int k = 0; k < NP; ++k) { hv[H_PARAMS + k] = th[k]; hv[H_PARAMS + NP + k] = sm[k]; }
        if (HW > H_PARAMS + 2 * NP) hv[HW - 1] = 0.0;
        double2* g_h = (double2*)(P.hrec + ((size_t)(t - 1) * N + c) * HW);
#pragma unroll
        for (int j = 0; 4 * j < NPH; ++j) {
            const int i = 4 * j + r;
            const double2 v = sel4(r, make_double2(hv[8 * j], hv[8 * j + 1]), make_double2(hv[(8 * j + 2) % HW], hv[(8 * j + 3) % HW]),
                                   make_double2(hv[(8 * j + 4) % HW], hv[(8 * j + 5) % HW]), make_double2(hv[(8 * j + 6) % HW], hv[(8 * j + 7) % HW]));
            if (i < NPH) g_h[i] = v;
        }
    }
    TS_MARK(4);
}
