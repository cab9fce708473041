// the lean exchange walk (exchangeMoves!, AlgoBGP.jl:647-716, for one min_improve >= 0 shared by all chains): shared by
// k_chain_iter_norm (in its prologue, N_global <= 4096) and k_exch_resolve_lean (stand-alone, N_global <= 8192) — part of
// libsmmhip (included by smmhip.hip inside its anonymous namespace; gfx950 device code).
#pragma once
// ------------------------------------------------------------------------------------------
// The level walk written for the number of INSTRUCTIONS per level.
// Measured (SMMHIP_TS=2, s_memtime per level, k_chain_iter_norm at 4096 chains): a level costs what one wave needs from
// barrier to barrier, and that is not the LDS round trip (71 cycles for a dependent ds_read_b32 in that kernel) but the wave's
// own instruction stream — ~8 cycles per instruction and ~25 per taken branch for a lone wave: 556 cycles for the ~65
// instructions of a level in the straightforward loop (fetch the pair word, test for "no pair", take it apart, address
// arithmetic, test for undecided keys, carry the partner), whatever the slot size.  So the plan and the slot format take the
// work out of the loop:
//   * the plan (k_exch_plan) pads every level to whole waves with dummy pairs that never swap: no lane asks whether it has a
//     pair; a wave without pairs skips the level on a scalar compare;
//   * a pair word holds the LDS byte offsets of its two slots (halved when 16 bits would not reach the last slot);
//   * a slot is 8 bytes: {32-bit order key of the value (order_key32), src | stamp << 16}.  `value_i - value_j > 0` is ONE
//     unsigned compare of the keys; equal keys (same high word of the doubles: rare) read the exact values from memory;
//   * set_exchanged! (:747-748) costs one add per trip: a swap stamps 1 + the position of its pair word into both slots, and
//     whoever needs the partner afterwards reads that pair word (lean_partner);
//   * the lane's next pair word is requested together with the slots: one LDS round trip per level.
// LDS: slots uint2[Ng4 + 4] (Ng4 = N_global rounded up to 4; the two slots behind the chains' are the dummy pair's) | pair
// words u32[plan_Kp].  The slots must start at LDS address 0 (the callers check).
// min_improve > 0 (the reference's default is 0.5, AlgoBGP.jl:522) — the WIDE form: a key cannot decide `v_i - v_j > m`, so a
// slot is 16 bytes {value, src | stamp << 16, -} and the test is the subtraction and the compare themselves (NaN values and a
// NaN threshold need nothing special: the compare is false).  The dummy pair is (d, d) with d one slot behind the chains',
// value 0: `0 - 0 > m` is false for m >= 0 or NaN (a chain's own slot would not do: another wave's swap could change it between
// the two reads).  LDS: slots uint4[Ng4 + 1] | pair words u32[plan_Kp]; up to ~7400 chains fit the 160 KB.
// ------------------------------------------------------------------------------------------
__host__ __device__ inline int lean_walk_Kp(int K) { return (K + 64 * LV_MAXLEV + 3) & ~3; }
__host__ __device__ inline size_t lean_walk_bytes(int Ng, int K) { return (size_t)(((Ng + 3) & ~3) + 4) * 8 + (size_t)lean_walk_Kp(K) * 4; }
__host__ __device__ inline int lean_walk_unit(int Ng) { return 8 * (((Ng + 3) & ~3) + 2) <= 65536 ? 8 : 4; }   // KParams::lean_unit
__host__ __device__ inline size_t lean_wide_bytes(int Ng, int K) { return (size_t)(((Ng + 3) & ~3) + 1) * 16 + (size_t)lean_walk_Kp(K) * 4; }
__host__ __device__ inline int lean_wide_unit(int Ng) { return 16 * (((Ng + 3) & ~3) + 1) <= 65536 ? 16 : 8; }    // KParams::lean_unit, wide form (8: up to 8188 chains)
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
// k_chain_iter's inline walk on this form: 16-byte slots whatever the threshold, two zero-valued dummy slots (the plan may be the
// key form's, whose dummy pair names two slots)
__host__ __device__ inline size_t tile_lean_slot_bytes(int Ng) { return (size_t)(((Ng + 3) & ~3) + 2) * 16; }
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

// LDS byte offsets of the two slots of a pair word (US = 1: the word holds them halved, KParams::lean_unit == 4)
template <int US>
__device__ inline void lean_decode(const uint32_t pw, uint32_t& ai, uint32_t& aj) {
    if constexpr (US == 0) { ai = pw & 0xffffu; aj = pw >> 16; }
    else { ai = (pw & 0xffffu) << US; aj = (pw >> 16) << US; }
}
// 1 + the chain's last exchange partner (0: none) from its slot's second word after the walk
// (SL: log2 of the slot size, 3 or 4 for the wide form)
template <int US, int SL = 3>
__device__ inline uint32_t lean_partner(const unsigned char* lds, const uint32_t pbase, const uint32_t meta, const uint32_t g) {
    const uint32_t stamp = meta >> 16;
    if (stamp == 0u) return 0u;
    const uint32_t pw = *(const uint32_t*)(lds + pbase + 4u * (stamp - 1u));
    const uint32_t i = (pw & 0xffffu) >> (SL - US), j = (pw >> 16) >> (SL - US);
    return (i == g ? j : i) + 1u;
}

// (GUARD: where the exact value of chain s comes from on a key tie — vsrc[s * vstride] by default; the p2p form reads a
// self-validating copy out of its window, GUARD::value)
struct WalkNoGuard {};
template <class GUARD>
__device__ inline double walk_value(const GUARD& g, const double* __restrict__ vsrc, const int vstride, const uint32_t s) {
    if constexpr (std::is_same<GUARD, WalkNoGuard>::value) return vsrc[(size_t)s * vstride];
    else return g.value(s);
}
// PCT (the persistent forms' wide walk, round 6): the threshold is the COLDER chain's own (min_improve[i], AlgoBGP.jl:522, :688) — a double per slot
// POSITION at LDS offset thr_base + 8 * position, read in the same LDS round trip as the slots (positions do not travel with the records a swap exchanges)
// one pair, on its own (further words of a level wider than the workgroup)
template <int US, bool WIDE = false, class GUARD = WalkNoGuard, bool PCT = false>
__device__ inline void lean_pair(const double* __restrict__ vsrc, const int vstride, const uint32_t pw, const uint32_t stamp, const double thr = 0.0,
                                 const GUARD& guard = GUARD(), const uint32_t thr_base = 0u) {
    uint32_t ai, aj;
    lean_decode<US>(pw, ai, aj);
    if constexpr (WIDE) {
        u32x4_t si, sj;
        double thr_l = thr;
        if constexpr (PCT) asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %4\n\tds_read_b64 %2, %5\n\ts_waitcnt lgkmcnt(0)" : "=&v"(si), "=&v"(sj), "=&v"(thr_l) : "v"(ai), "v"(aj), "v"(thr_base + (ai >> 1)) : "memory");
        else asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(si), "=&v"(sj) : "v"(ai), "v"(aj) : "memory");
        const double vi = __hiloint2double((int)si.y, (int)si.x), vj = __hiloint2double((int)sj.y, (int)sj.x);
        if (vi - vj > thr_l) {   // dist_fun = -, AlgoBGP.jl:688
            const u32x4_t ni = {sj.x, sj.y, (sj.z & 0xffffu) | stamp, 0u}, nj = {si.x, si.y, (si.z & 0xffffu) | stamp, 0u};
            asm volatile("ds_write_b128 %0, %2\n\tds_write_b128 %1, %3" :: "v"(ai), "v"(aj), "v"(ni), "v"(nj) : "memory");
        }
        return;
    }
    u32x2_t si, sj;
    asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(si), "=&v"(sj) : "v"(ai), "v"(aj) : "memory");
    bool swap = si.x > sj.x;
    if (si.x == sj.x) swap = walk_value(guard, vsrc, vstride, si.y & 0xffffu) - walk_value(guard, vsrc, vstride, sj.y & 0xffffu) > 0.0;
    if (swap) {
        const u32x2_t ni = {sj.x, (sj.y & 0xffffu) | stamp}, nj = {si.x, (si.y & 0xffffu) | stamp};
        asm volatile("ds_write_b64 %0, %2\n\tds_write_b64 %1, %3" :: "v"(ai), "v"(aj), "v"(ni), "v"(nj) : "memory");
    }
}
// the levels: `ov` holds in lane l the first word of level l (lane nlev: the padded length); pair words at LDS offset pbase;
// exact values of chain s at vsrc[s * vstride].  Every level ends with a workgroup barrier, except the narrow tail (every
// remaining level at most one wave wide), which wave 0 walks alone: only wave 0 may read the result without a barrier.
// the first level of the narrow tail (call it before the barrier that ends the staging: it needs no LDS data)
__device__ inline int lean_walk_tail(const uint32_t ov, const int nlev, const int lane) {
    const uint32_t nx = (uint32_t)__shfl_down((int)ov, 1, 64);
    const unsigned long long wide = __ballot(lane < nlev && nx - ov > 64u);
    return wide ? 64 - __builtin_clzll(wide) : 0;
}
template <int NT, int US, bool WIDE = false, class GUARD = WalkNoGuard, bool PCT = false>
__device__ inline void lean_walk_levels(const double* __restrict__ vsrc, const int vstride, const uint32_t pbase, const uint32_t ov,
                                        const int nlev, const int tid, const int ltail, const double thr = 0.0, const GUARD& guard = GUARD(),
                                        const uint32_t thr_base = 0u) {
    const uint32_t wbase = (uint32_t)__builtin_amdgcn_readfirstlane(tid & ~63);   // this wave's first word within a trip
    const uint32_t tid4p = pbase + 4u * (uint32_t)tid;
    const uint32_t tidst = ((uint32_t)tid + 1u) << 16;                           // this lane's stamp at position 0
    uint32_t st = 0u, st1 = (uint32_t)__builtin_amdgcn_readlane((int)ov, 1);     // first words of the levels l, l+1
    uint32_t pw = 0u;
    if (nlev > 0) asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(pw) : "v"(tid4p) : "memory");   // (garbage past the level: not used)
    // (written for the common case to fall through: every taken branch costs a lone wave ~25 cycles)
    auto level = [&](const int l, const bool lone) {
        const uint32_t st2 = (uint32_t)__builtin_amdgcn_readlane((int)ov, l + 2);   // (l + 2 <= 33; past the last level: anything)
        const uint32_t width = st1 - st;
        if (__builtin_expect(wbase < width, 1)) {
            uint32_t nptr;                                // this lane's word of the next level (garbage past it: not used)
            asm volatile("v_add_u32 %0, %1, %2" : "=v"(nptr) : "s"(st1 << 2), "v"(tid4p));
            uint32_t stamp;                               // in a vector register: v_and_or_b32 takes one scalar operand, the mask
            asm volatile("v_add_u32 %0, %1, %2" : "=v"(stamp) : "s"(st << 16), "v"(tidst));
            uint32_t ai, aj;
            lean_decode<US>(pw, ai, aj);
            uint32_t pwn;
            if constexpr (WIDE) {
                u32x4_t si, sj;
                double thr_l = thr;
                if constexpr (PCT)   // (the threshold of position i rides in the same LDS round trip; a compile-time form: as a run-time branch it cost the one-threshold walk 0.2 us)
                    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b32 %2, %6\n\tds_read_b64 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(si), "=&v"(sj), "=&v"(pwn), "=&v"(thr_l) : "v"(ai), "v"(aj), "v"(nptr), "v"(thr_base + (ai >> 1)) : "memory");
                else
                    asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %4\n\tds_read_b32 %2, %5\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(si), "=&v"(sj), "=&v"(pwn) : "v"(ai), "v"(aj), "v"(nptr) : "memory");
                const double vi = __hiloint2double((int)si.y, (int)si.x), vj = __hiloint2double((int)sj.y, (int)sj.x);
                if (vi - vj > thr_l) {   // dist_fun = -, AlgoBGP.jl:688; swap_ev_ij!, :739-744; the stamp stands for set_exchanged!, :747-748
                    const u32x4_t ni = {sj.x, sj.y, (sj.z & 0xffffu) | stamp, 0u}, nj = {si.x, si.y, (si.z & 0xffffu) | stamp, 0u};
                    asm volatile("ds_write_b128 %0, %2\n\tds_write_b128 %1, %3" :: "v"(ai), "v"(aj), "v"(ni), "v"(nj) : "memory");
                }
            } else {
                u32x2_t si, sj;
                asm volatile("ds_read_b64 %0, %3\n\tds_read_b64 %1, %4\n\tds_read_b32 %2, %5\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(si), "=&v"(sj), "=&v"(pwn) : "v"(ai), "v"(aj), "v"(nptr) : "memory");
                bool swap = si.x > sj.x;
                const bool tie = si.x == sj.x;
                if (__builtin_expect(__ballot(tie) != 0ull, 0)) {   // the keys do not decide: the exact values (dist_fun = -, AlgoBGP.jl:688)
                    if (tie) swap = walk_value(guard, vsrc, vstride, si.y & 0xffffu) - walk_value(guard, vsrc, vstride, sj.y & 0xffffu) > 0.0;
                }
                if (swap) {   // swap_ev_ij!, :739-744; the stamp stands for set_exchanged!, :747-748
                    const u32x2_t ni = {sj.x, (sj.y & 0xffffu) | stamp}, nj = {si.x, (si.y & 0xffffu) | stamp};
                    asm volatile("ds_write_b64 %0, %2\n\tds_write_b64 %1, %3" :: "v"(ai), "v"(aj), "v"(ni), "v"(nj) : "memory");
                }
            }
            pw = pwn;
            if (__builtin_expect(width > (uint32_t)NT, 0)) {   // a level wider than the workgroup: its further words, one by one
                for (uint32_t o = (uint32_t)NT; wbase + o < width; o += (uint32_t)NT) {
                    uint32_t pwx;
                    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(pwx) : "v"(tid4p + 4u * (st + o)) : "memory");
                    lean_pair<US, WIDE, GUARD, PCT>(vsrc, vstride, pwx, ((st + o) << 16) + tidst, thr, guard, thr_base);
                }
            }
        } else if (wbase < st2 - st1) {   // idle in this level, not in the next: its word
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(pw) : "v"(tid4p + 4u * st1) : "memory");
        }
        st = st1; st1 = st2;
        if (!lone) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __syncthreads(); }
    };
    {   // (two levels per trip of the loop, by hand: the compiler does not unroll across the barrier)
        int l = 0;
#pragma clang loop unroll(disable)
        for (; l + 2 <= ltail; l += 2) { level(l, false); level(l + 1, false); }
        if (l < ltail) level(l, false);
    }
    if (ltail < nlev && tid < 64) {   // the narrow tail: wave 0 alone, no barriers (its own LDS operations complete in order)
        int l = ltail;
#pragma clang loop unroll(disable)
        for (; l + 2 <= nlev; l += 2) { level(l, true); level(l + 1, true); }
        if (l < nlev) level(l, true);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}
