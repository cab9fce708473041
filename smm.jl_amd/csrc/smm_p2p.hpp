// the p2p form of the sharded iteration: windows, self-validating records, arrival counters, publish / push / unpack kernels — part
// of libsmmhip (included by smmhip.hip inside its anonymous namespace; gfx950 device code).
#pragma once
// ------------------------------------------------------------------------------------------
// exchangeMoves! (AlgoBGP.jl:647-716) couples the chains of all shards through each chain's last accepted record.  The xGMI
// fabric of an MI355X node is point-to-point (every GPU has a direct link to every other), so the all-gather of those records
// needs no collective call at all: every rank owns a WINDOW of device memory that all other ranks map (HIP IPC between
// processes, plain pointers between contexts of one process), and a chain's accept step stores its new record, value and walk
// slot straight into every rank's window — fire-and-forget stores over the links.  The next iteration's kernel reads its own
// window.  No host-enqueued collective, no RCCL kernel, no extra launch.
//
// Three forms share the windows:
//   * INLINE (k_chain_iter_norm_p2p: objfunc_norm, np == nm <= 4, min_improve == 0, N_global <= 8192): ONE launch per iteration
//     and NOBODY WAITS FOR AN ACKNOWLEDGEMENT.  Everything a reader takes out of a window says which iteration it is from:
//       slot   uint2  {order_key32(value), chain | tag << 16}                 (the walk strips the tag while staging)
//       llval  uint4  {value.lo, tag, value.hi, tag}                          (the exact value, read on a key tie only)
//       llrec  per 16 bytes of record two uint4 {d0, tag, d1, tag}, {d2, tag, d3, tag}
//     8-byte stores are single-copy atomic, so a word with the right tag IS that iteration's word (the LL protocol of the
//     collective libraries).  A reader that finds an older tag looks again with loads that no cache serves (sc0 sc1).
//     The writer stores and ends: no s_waitcnt for the acknowledgements of remote stores in the tail of every launch, no atomic.
//   * ROWS (k_chain_iter_norm_p2p_rows + k_exch_resolve_rows<., true>: the same objectives at 8192 < N_global <= 32768 — four and
//     eight shards of 4096): TWO launches per iteration, still nobody waiting in a kernel of its own.  The accept step stores a
//     4-byte slot4 word per chain, order_key17(value) << 15 | tag15 (the chain is the word's position); the one resolving workgroup
//     reads them past the caches, validates word by word, walks, and falls back — inside the launch — to the exact values where a
//     key says "NaN" or the plan does not fit the rows.  The chain kernel has no walk at all and reads the resolution's result.
//   * GENERIC (every other objective / population): the chain kernel writes plain records and values into its OWN window, a push
//     kernel copies them to every other window and counts itself in (one atomic per unit and rank, after the acknowledgements);
//     k_p2p_wait waits on this rank's counters in front of the stand-alone exchange resolution.
// k_p2p_unpack turns the self-validating form into the plain one where the inline form needs a stand-alone resolution (once per
// look-ahead window, and when the run is settled).
//
// Window (identical layout on every rank; Ng = N_global, RW = doubles per record), two parities each — iteration t reads parity
// (t-1) & 1 and writes parity t & 1; a rank that runs ahead cannot overwrite what a slower one still reads: before its kernel t+1
// stores anything it has seen every chain's slot of iteration t, and a chain's slot is stored after its tile's prologue reads:
//   arrived[P2P_MAXG]  u64, 128 bytes apart   arrivals from source rank r (generic form)
//   nan                u32                    a NaN value entered the population in publication epoch <word> (order keys do not cover it; an older epoch's word means nothing)
//   rec [2][Ng][RW], val [2][Ng + 4]          plain records / values
//   slot[2][Ng + 4]    uint2                  tagged walk slots
//   slot4[2][Ng + 4]   u32                    tagged 17-bit keys (rows form: 8192 < N_global <= 32768)
//   llrec[2][Ng][2 RW], llval[2][Ng] uint4    self-validating records / values
// ------------------------------------------------------------------------------------------
struct P2PLayout { size_t arrived, nan, rec[2], val[2], slot[2], llrec[2], llval[2], slot4[2], total; };
__host__ __device__ inline P2PLayout p2p_layout(const int Ng, const int RW) {
    P2PLayout L;
    L.arrived = 0;
    L.nan = (size_t)128 * P2P_MAXG;
    size_t o = L.nan + 128;
    for (int b = 0; b < 2; ++b) { L.rec[b] = o; o += ((size_t)Ng * RW * 8 + 127) & ~(size_t)127; }
    for (int b = 0; b < 2; ++b) { L.val[b] = o; o += ((size_t)(Ng + 4) * 8 + 127) & ~(size_t)127; }
    for (int b = 0; b < 2; ++b) { L.slot[b] = o; o += ((size_t)(Ng + 4) * 8 + 127) & ~(size_t)127; }
    for (int b = 0; b < 2; ++b) { L.llrec[b] = o; o += ((size_t)Ng * RW * 16 + 127) & ~(size_t)127; }
    for (int b = 0; b < 2; ++b) { L.llval[b] = o; o += ((size_t)(Ng + 4) * 16 + 127) & ~(size_t)127; }
    for (int b = 0; b < 2; ++b) { L.slot4[b] = o; o += ((size_t)(Ng + 4) * 4 + 127) & ~(size_t)127; }
    L.total = o;
    return L;
}
// device side: the offsets from the kernel arguments (selects on constant indices: no address arithmetic, no scratch)
__device__ inline size_t p2p_rec_off(const KParams& P, const int b) { return b ? P.p2p_off[1] : P.p2p_off[0]; }
__device__ inline size_t p2p_val_off(const KParams& P, const int b) { return b ? P.p2p_off[3] : P.p2p_off[2]; }
__device__ inline size_t p2p_slot_off(const KParams& P, const int b) { return b ? P.p2p_off[5] : P.p2p_off[4]; }
__device__ inline size_t p2p_llrec_off(const KParams& P, const int b) { return b ? P.p2p_off[7] : P.p2p_off[6]; }
__device__ inline size_t p2p_llval_off(const KParams& P, const int b) { return b ? P.p2p_off[9] : P.p2p_off[8]; }
__device__ inline size_t p2p_slot4_off(const KParams& P, const int b) { return b ? P.p2p_off[11] : P.p2p_off[10]; }
// the tag of iteration t: never 0 (a fresh window is zeroed); different from the tag of iteration t-2, whose words the stores of
// iteration t replace (same parity), and from every word an EARLIER PUBLICATION left behind — a run that was settled, rolled back
// by an uploaded state and published again writes the same iteration numbers a second time, and a reader must not take the old
// words for the new ones: the epoch (publications so far, the same on every rank) is part of the tag
__host__ __device__ inline uint32_t p2p_tag_of(const int t, const uint32_t epoch) { return 0x8000u | ((epoch & 0x7ffu) << 4) | ((uint32_t)t & 0xfu); }
__device__ inline uint32_t p2p_tag(const KParams& P, const int t) { return p2p_tag_of(t, P.p2p_epoch); }
// (a NaN value travels IN the slot word — the reserved key P2P_KEY_NAN, which no other value has —, so a reader learns of it with the very
// word it validates: the window's NaN word is stored separately and may become visible after the slot)
constexpr uint32_t P2P_KEY_NAN = 0xffffffffu;
__device__ inline unsigned long long p2p_slot_word(const KParams& P, const double v, const uint32_t gchain, const int t) {
    return (unsigned long long)(v != v ? P2P_KEY_NAN : order_key32(v)) | ((unsigned long long)(gchain | (p2p_tag(P, t) << 16)) << 32);
}
// The rows form (8192 < N_global <= 32768; k_exch_resolve_rows<., true>) keeps a chain's walk slot in FOUR bytes: the 17-bit order key of
// its value (smm_params.hpp; the last bucket also says "NaN": the reader then resolves that iteration on the exact values) over a
// 15-bit tag — half the bytes the one resolving workgroup has to fetch past the caches.  The chain is the word's position.
__device__ inline uint32_t p2p_tag15(const KParams& P, const int t) { return 0x4000u | ((P.p2p_epoch & 0x3ffu) << 4) | ((uint32_t)t & 0xfu); }
__device__ inline uint32_t p2p_slot4_word(const KParams& P, const double v, const int t) {
    return ((v != v ? XKEY17_TOP : order_key17(v)) << 15) | p2p_tag15(P, t);
}
__device__ inline void p2p_store4(void* p, const uint32_t v) { __hip_atomic_store((uint32_t*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// chains per arrival unit (generic form): the push kernel arrives once per workgroup of this share
constexpr int P2P_UNIT = 16;
__host__ __device__ inline int p2p_units(const int N) { return (N + P2P_UNIT - 1) / P2P_UNIT; }
constexpr unsigned long long P2P_TIMEOUT_TICKS = 400000000ull;   // 4 s of the 100 MHz wall clock: a peer is gone, not late
// a spin for tagged words of iteration t_data is over — 1: timed out (the caller reports it); 2: the run has FAILED and those words will never come:
// the launch that should have stored them was poisoned (an error of an EARLIER iteration: it saw the word at its entry and stored nothing — error
// convention, include/smmhip.h), or it gave up itself (kind 3 IN that iteration: a form that could not resolve its exchange returns without storing; a hard
// error of the algorithm, kinds 1 / 2, does not count: that iteration completes and its words arrive).  Nothing is reported then — the first error stands —
// and a failed run drains at once instead of waiting 4 s in every remaining launch of its step (a NaN uploaded into a shard of N_global <= 8192,
// reported in its first exchange: 168 s for a step of 19 iterations before)
__device__ inline int p2p_spin_over(const KParams& P, const unsigned long long t0, const int t_data) {
    if (wall_clock64() - t0 > P2P_TIMEOUT_TICKS) return 1;
    const unsigned long long e = __hip_atomic_load(P.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (e == ERR_NONE) return 0;
    const int it = (int)(e >> 34), kind = (int)(e & 3ull);
    return (it < t_data || (it == t_data && kind == 3)) ? 2 : 0;
}

// Stores into a window are system-scope stores (sc0 sc1: written through, acknowledged once visible to every agent), so
// "s_waitcnt vmcnt(0)" after them is the release (generic form).  A system-scope release fence would write back the whole L2 per
// tile and launch: measured 50 us instead of 14 per launch.
typedef unsigned int p2p_u32x4 __attribute__((ext_vector_type(4)));
__device__ inline void p2p_store16u(void* p, const p2p_u32x4 q) {
    // (s_nop 1: a VALU write of the data registers of a store of more than 8 bytes needs 2 wait states on gfx940+, and the
    // compiler's hazard recognizer does not look into inline asm — without it the next select overwrote the data: found by the
    // 8-rank test, where the unrolled peer loop puts a select right behind every store)
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p), "v"(q) : "memory");
}
__device__ inline void p2p_store16(void* p, const double2 v) {   // 16 plain bytes
    const unsigned long long a = __builtin_bit_cast(unsigned long long, v.x), b = __builtin_bit_cast(unsigned long long, v.y);
    const p2p_u32x4 q = {(unsigned)a, (unsigned)(a >> 32), (unsigned)b, (unsigned)(b >> 32)};
    p2p_store16u(p, q);
}
__device__ inline void p2p_store8(void* p, const unsigned long long v) {
    __hip_atomic_store((unsigned long long*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// 16 bytes of payload as 32 self-validating bytes at p
__device__ inline void p2p_store_ll(void* p, const double2 v, const uint32_t tag) {
    const unsigned long long a = __builtin_bit_cast(unsigned long long, v.x), b = __builtin_bit_cast(unsigned long long, v.y);
    const p2p_u32x4 q0 = {(unsigned)a, tag, (unsigned)(a >> 32), tag}, q1 = {(unsigned)b, tag, (unsigned)(b >> 32), tag};
    p2p_store16u(p, q0);
    p2p_store16u((unsigned char*)p + 16, q1);
}
// A load that no cache serves (system scope): how a self-validating word is read — a line fetched earlier in the same launch,
// before the word's new contents had landed, must not be served again.
__device__ inline uint4 p2p_load16_sys(const void* p) {
    p2p_u32x4 q;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(q) : "v"(p) : "memory");
    return make_uint4(q.x, q.y, q.z, q.w);
}
// four / eight of them in flight together (one wait: an asm load's destination counts as written at the end of its statement)
__device__ inline void p2p_load16x4_sys(const void* p0, const void* p1, const void* p2, const void* p3, uint4& a, uint4& b, uint4& c, uint4& d) {
    p2p_u32x4 q0, q1, q2, q3;
    asm volatile("global_load_dwordx4 %0, %4, off sc0 sc1\n\tglobal_load_dwordx4 %1, %5, off sc0 sc1\n\t"
                 "global_load_dwordx4 %2, %6, off sc0 sc1\n\tglobal_load_dwordx4 %3, %7, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
    a = make_uint4(q0.x, q0.y, q0.z, q0.w); b = make_uint4(q1.x, q1.y, q1.z, q1.w);
    c = make_uint4(q2.x, q2.y, q2.z, q2.w); d = make_uint4(q3.x, q3.y, q3.z, q3.w);
}
__device__ inline void p2p_load16x2_sys(const void* p0, const void* p1, uint4& a, uint4& b) {
    p2p_u32x4 q0, q1;
    asm volatile("global_load_dwordx4 %0, %2, off sc0 sc1\n\tglobal_load_dwordx4 %1, %3, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(q0), "=&v"(q1) : "v"(p0), "v"(p1) : "memory");
    a = make_uint4(q0.x, q0.y, q0.z, q0.w); b = make_uint4(q1.x, q1.y, q1.z, q1.w);
}
__device__ inline bool p2p_ll_ok(const uint4 q, const uint32_t tag) { return q.y == tag && q.w == tag; }
__device__ inline double p2p_ll_double(const uint4 q) { return __hiloint2double((int)q.z, (int)q.x); }
// the exact value of chain g after iteration t out of this rank's window (any lane on its own, e.g. the tie branch of the walk)
__device__ inline double p2p_ll_value(const KParams& P, const int t, const uint32_t g) {
    const uint4* a = (const uint4*)(P.p2p_self + p2p_llval_off(P, t & 1)) + g;
    const uint32_t tag = p2p_tag(P, t);
    uint4 q = p2p_load16_sys(a);   // (past the caches, like every read of a window: see p2p_load16_sys)
    if (!p2p_ll_ok(q, tag)) {
        const unsigned long long t0 = wall_clock64();
        do {
            __builtin_amdgcn_s_sleep(1);
            q = p2p_load16_sys(a);   // (past the caches: the stale line must not be served again)
            if (const int o = p2p_spin_over(P, t0, t)) { if (o == 1) report_error(P, 3, t + 1, (int)g); break; }
        } while (!p2p_ll_ok(q, tag));
    }
    return p2p_ll_double(q);
}
struct P2PWalkValues {   // GUARD of the lean walk (smm_walk_lean.hpp)
    const KParams& P; int t;
    __device__ inline double value(const uint32_t s) const { return p2p_ll_value(P, t, s); }
};

// ---- generic form: arrival counters ----
// Every wave for itself: lanes < G poll this rank's arrival counters until all of them have reached P.p2p_want, then the wave
// drops what its caches may hold of the windows.  0: complete; 1: timed out (the caller reports it); 2: the run has failed
// already (nothing to report, nothing to wait for: a failed run drains without waiting 4 s in every launch).
__device__ inline int p2p_wait_arrivals(const KParams& P, const int lane) {
    int rc = 0;
    if (P.p2p_want != 0ull && *(const volatile unsigned long long*)P.err != ERR_NONE) rc = 2;
    else if (P.p2p_want != 0ull && lane < P.p2p_G) {
        const unsigned long long* a = (const unsigned long long*)(P.p2p_self + 128 * (size_t)lane);
        const unsigned long long t0 = wall_clock64();
        unsigned spins = 0;
        while (__hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < P.p2p_want) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 255u) == 0u && wall_clock64() - t0 > P2P_TIMEOUT_TICKS) { rc = 1; break; }
        }
    }
    const bool bad1 = __ballot(rc == 1) != 0ull, bad2 = __ballot(rc == 2) != 0ull;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    return bad2 ? 2 : (bad1 ? 1 : 0);
}
// one arrival of this rank at every rank (lanes < G of one wave); the caller's waves have waited for their stores' acknowledgements
__device__ inline void p2p_arrive(const KParams& P, const int lane) {
    unsigned char* w = nullptr;   // lane p: rank p's window (a select chain: no dynamic index into the kernel arguments)
#pragma unroll
    for (int p = 0; p < P2P_MAXG; ++p) w = lane == p ? P.p2p_win[p] : w;
    if (lane < P.p2p_G)
        __hip_atomic_fetch_add((unsigned long long*)(w + 128 * (size_t)P.p2p_rank), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// k_p2p_push: this rank's records after iteration t into every rank's window, one workgroup per unit of P2P_UNIT chains.
//   LL = false (generic form): plain records and values (and tagged slots), then one arrival per workgroup and rank.  FROM_CTX:
//        the source is the context's own record array (the first publication after stepping in another form), else this rank's
//        own window, where the chain kernel has just written them.
//   LL = true (first publication of the inline form, always FROM_CTX): tagged slots and self-validating values and records;
//        nobody counts anything.
template <bool FROM_CTX, bool LL>
__global__ __launch_bounds__(256) void k_p2p_push(const KParams P, const int t, const double* __restrict__ rec_src) {
    const int b = t & 1;   // the state after iteration t, into parity t & 1
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int c0 = (int)blockIdx.x * P2P_UNIT, n = min(P2P_UNIT, P.N - c0);
    const int RW = P.RW;
    unsigned char* mine = P.p2p_self;
    const double* rs = FROM_CTX ? rec_src + (size_t)c0 * RW : (const double*)(mine + p2p_rec_off(P, b)) + (size_t)(P.offset + c0) * RW;
    const double* vs = (const double*)(mine + p2p_val_off(P, b)) + P.offset + c0;
    const uint32_t tag = p2p_tag(P, t);
#pragma unroll
    for (int p = 0; p < P2P_MAXG; ++p) {
        if (p >= P.p2p_G) break;
        unsigned char* w = P.p2p_win[p];
        const bool own = !FROM_CTX && p == P.p2p_rank;   // (its records and values are where they belong already)
        if (LL) {
            unsigned char* rd = w + p2p_llrec_off(P, b) + (size_t)(P.offset + c0) * RW * 16;
            for (int i = tid; i < n * RW / 2; i += 256) p2p_store_ll(rd + (size_t)i * 32, make_double2(rs[2 * i], rs[2 * i + 1]), tag);
        } else if (!own) {
            double* rd = (double*)(w + p2p_rec_off(P, b)) + (size_t)(P.offset + c0) * RW;
            for (int i = tid; i < n * RW; i += 256) p2p_store8(rd + i, __builtin_bit_cast(unsigned long long, rs[i]));
        }
        if (tid < n) {
            const double v = FROM_CTX ? rs[(size_t)tid * RW] : vs[tid];   // (FROM_CTX: the record's own value column)
            const unsigned long long vb = __builtin_bit_cast(unsigned long long, v);
            if (LL) {
                const p2p_u32x4 q = {(unsigned)vb, tag, (unsigned)(vb >> 32), tag};
                p2p_store16u((uint4*)(w + p2p_llval_off(P, b)) + P.offset + c0 + tid, q);
            } else if (!own) p2p_store8((double*)(w + p2p_val_off(P, b)) + P.offset + c0 + tid, vb);
            p2p_store8((uint2*)(w + p2p_slot_off(P, b)) + P.offset + c0 + tid, p2p_slot_word(P, v, (uint32_t)(P.offset + c0 + tid), t));
            if (LL) p2p_store4((uint32_t*)(w + p2p_slot4_off(P, b)) + P.offset + c0 + tid, p2p_slot4_word(P, v, t));
            if (v != v) { __hip_atomic_fetch_max((uint32_t*)(w + 128 * (size_t)P2P_MAXG), P.p2p_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
        }
    }
    if (LL) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave's own stores are out before the workgroup meets ...
    __syncthreads();                                   // ... so that the arrival below cannot overtake any of them
    if (tid < 64) p2p_arrive(P, lane);
}
// the wait alone (one wave): in front of a stand-alone exchange resolution (generic form)
__global__ __launch_bounds__(64) void k_p2p_wait(const KParams P, const int t) {
    if (p2p_wait_arrivals(P, (int)threadIdx.x) == 1 && threadIdx.x == 0) report_error(P, 3, t, P.offset);
}
// k_p2p_unpack (inline form): the self-validating records and values of ALL chains after iteration t, out of this rank's window,
// into its plain arrays (what the stand-alone exchange resolution and k_flush read) — waiting, chain by chain, until they are there
__global__ __launch_bounds__(256) void k_p2p_unpack(const KParams P, const int t) {
    const int g = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (g >= P.Ng) return;
    const int b = t & 1, RW = P.RW;
    const uint32_t tag = p2p_tag(P, t);
    unsigned char* mine = P.p2p_self;
    ((double*)(mine + p2p_val_off(P, b)))[g] = p2p_ll_value(P, t, (uint32_t)g);
    const uint4* src = (const uint4*)(mine + p2p_llrec_off(P, b) + (size_t)g * RW * 16);
    double* dst = (double*)(mine + p2p_rec_off(P, b)) + (size_t)g * RW;
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < RW; ++i) {
        uint4 q = p2p_load16_sys(src + i);
        while (!p2p_ll_ok(q, tag)) {
            __builtin_amdgcn_s_sleep(1);
            q = p2p_load16_sys(src + i);
            if (const int o = p2p_spin_over(P, t0, t)) { if (o == 1) report_error(P, 3, t, g); return; }
        }
        dst[i] = p2p_ll_double(q);
    }
}

// this rank's OWN records after iteration t out of their self-validating form in its window into a plain array [N][RW] (the hand-over
// to the persistent form, smm_chain_persist_loc.hpp): they are the previous launch's own stores — complete, nothing to wait for
__global__ __launch_bounds__(256) void k_p2p_own_records(const KParams P, const int t, double* __restrict__ dst) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= P.N * P.RW) return;
    const uint4* src = (const uint4*)(P.p2p_self + p2p_llrec_off(P, t & 1) + (size_t)P.offset * P.RW * 16);
    dst[i] = p2p_ll_double(p2p_load16_sys(src + i));
}
