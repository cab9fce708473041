// the p2p form of the sharded iteration: windows, arrival counters, the generic publish / push kernels — part of libsmmhip
// (included by smmhip.hip inside its anonymous namespace; gfx950 device code).
#pragma once
// ------------------------------------------------------------------------------------------
// exchangeMoves! (AlgoBGP.jl:647-716) couples the chains of all shards through each chain's last accepted record.  The xGMI
// fabric of an MI355X node is point-to-point (every GPU has a direct link to every other), so the all-gather of those records
// needs no collective call at all: every rank owns a WINDOW of device memory that all other ranks map (HIP IPC between
// processes, plain pointers between contexts of one process), and a chain's accept step stores its new record, value and walk
// slot straight into every rank's window — fire-and-forget stores over the links — followed by a system-scope release and ONE
// atomic add per tile on an arrival counter in each window.  The next iteration's kernel polls its OWN window's counters
// (one per source rank, monotone: `units` arrivals per iteration and source), then walks and reads donor records from its own
// memory.  No host-enqueued collective, no RCCL kernel, no extra launch: an iteration of a shard is one launch, like the single
// shard's, plus the flight time of the last tile's stores.
//
// Window (identical layout on every rank; Ng = N_global, RW = doubles per record):
//   arrived[P2P_MAXG]  u64, 128 bytes apart   arrivals from source rank r
//   nan                u32                    a NaN value entered the population (order keys do not cover it: sticky)
//   rec [2][Ng][RW]    last accepted records after the accept step of iteration t, at parity t & 1, global chain order
//   val [2][Ng + 4]    their values (the exchange walk's exact input)
//   slot[2][Ng + 4]    uint2 {order_key32(value), chain}: the lean walk's initial slots
// Two parities: iteration t reads parity (t-1) & 1 and writes parity t & 1.  A rank that runs ahead cannot overwrite what a slower
// one still reads: its kernel t+1 stores only after it has seen every rank's arrivals of iteration t, and a rank arrives only
// after all its tiles are past their prologue reads.
// ------------------------------------------------------------------------------------------
struct P2PLayout {
    size_t arrived, nan, rec[2], val[2], slot[2], total;
    // (selects, not indexed loads: the struct lives in registers)
    __host__ __device__ size_t rec_at(int b) const { return b ? rec[1] : rec[0]; }
    __host__ __device__ size_t val_at(int b) const { return b ? val[1] : val[0]; }
    __host__ __device__ size_t slot_at(int b) const { return b ? slot[1] : slot[0]; }
};
__host__ __device__ inline P2PLayout p2p_layout(const int Ng, const int RW) {
    P2PLayout L;
    L.arrived = 0;
    L.nan = (size_t)128 * P2P_MAXG;
    size_t o = L.nan + 128;
    for (int b = 0; b < 2; ++b) { L.rec[b] = o; o += ((size_t)Ng * RW * 8 + 127) & ~(size_t)127; }
    for (int b = 0; b < 2; ++b) { L.val[b] = o; o += ((size_t)(Ng + 4) * 8 + 127) & ~(size_t)127; }
    for (int b = 0; b < 2; ++b) { L.slot[b] = o; o += ((size_t)(Ng + 4) * 8 + 127) & ~(size_t)127; }
    L.total = o;
    return L;
}
// chains per arrival unit: a unit is a tile of k_chain_iter_norm (the chain kernels that push from their epilogue arrive once
// per tile; the generic push kernel once per workgroup of the same share)
constexpr int P2P_UNIT = 16;
__host__ __device__ inline int p2p_units(const int N) { return (N + P2P_UNIT - 1) / P2P_UNIT; }
// device side: the offsets from the kernel arguments (selects on constant indices: no address arithmetic, no scratch)
__device__ inline size_t p2p_rec_off(const KParams& P, const int b) { return b ? P.p2p_off[1] : P.p2p_off[0]; }
__device__ inline size_t p2p_val_off(const KParams& P, const int b) { return b ? P.p2p_off[3] : P.p2p_off[2]; }
__device__ inline size_t p2p_slot_off(const KParams& P, const int b) { return b ? P.p2p_off[5] : P.p2p_off[4]; }
// A slot in a window carries the iteration it belongs to in the upper half of its second word ({order_key32(value), chain |
// tag << 16}; the walk wants that half zero and strips it while staging): an 8-byte store is single-copy atomic, so a reader
// that finds the tag of the iteration it wants has that iteration's slot — it need not look at the arrival counters first.
__host__ __device__ inline uint32_t p2p_tag(const int t) { return 0x8000u | ((uint32_t)t & 0x7fffu); }   // (never 0: a fresh window is zeroed)
__device__ inline unsigned long long p2p_slot_word(const double v, const uint32_t gchain, const int t) {
    return (unsigned long long)order_key32(v) | ((unsigned long long)(gchain | (p2p_tag(t) << 16)) << 32);
}
constexpr unsigned long long P2P_TIMEOUT_TICKS = 400000000ull;   // 4 s of the 100 MHz wall clock: a peer is gone, not late

// Memory ordering without fences.  A system-scope release fence writes back the whole L2 (buffer_wbl2) and a system-scope acquire
// invalidates it — per tile and launch that is most of an iteration (measured: 50 us instead of 14 per launch).  Neither is
// needed here: the windows are UNCACHED memory (hipDeviceMallocUncached: no level of any device's cache hierarchy keeps a line of
// them), every store into a window is a system-scope store (sc0 sc1: acknowledged once it is visible to every agent), so
// "s_waitcnt vmcnt(0)" after the stores IS the release, and a reader that has seen the counters reads memory itself.
__device__ inline void p2p_store16(void* p, const double2 v) {   // 16 bytes into a window
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const unsigned long long a = __builtin_bit_cast(unsigned long long, v.x), b = __builtin_bit_cast(unsigned long long, v.y);
    const u32x4 q = {(unsigned)a, (unsigned)(a >> 32), (unsigned)b, (unsigned)(b >> 32)};
    // (s_nop 1: a VALU write of the data registers of a store of more than 8 bytes needs 2 wait states on gfx940+, and the
    // compiler's hazard recognizer does not look into inline asm — without it the next select overwrote the data: found by the
    // 8-rank test, where the unrolled peer loop puts a select right behind every store)
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p), "v"(q) : "memory");
}
__device__ inline void p2p_store8(void* p, const unsigned long long v) {
    __hip_atomic_store((unsigned long long*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// Every wave for itself: lanes < G poll this rank's arrival counters until all of them have reached P.p2p_want, then the wave
// drops what its caches may hold of the windows (a system-scope acquire: lines of a window read earlier in this launch, before
// their new contents had landed, must not be served again — this runs only where somebody actually has to wait).
// 0: complete; 1: timed out (the caller reports it); 2: the run has failed already (nothing to report, nothing to wait for).
__device__ inline int p2p_wait_arrivals(const KParams& P, const int lane) {
    int rc = 0;
    // (a run that has already failed — a peer timed out, a hard error — drains without waiting 4 s in every launch)
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
// `n` arrivals of this rank at every rank (lanes < G of one wave).  The caller's stores into the windows must be COMPLETE:
//   * WAIT = true: they are this wave's own, and the wave waits for their acknowledgements first;
//   * WAIT = false: they were made by an earlier launch on the same stream.  That is how the chain kernel arrives: its accept
//     step stores and is done (no wait for acknowledgements in the tail of every launch: ~2 us), and the NEXT launch's first
//     instructions count those stores in (F_P2P_ARRIVE) — the slots carry their iteration tag and need no counter, the counters
//     guard the records, which nobody reads before his walk is over.
template <bool WAIT>
__device__ inline void p2p_arrive(const KParams& P, const int lane, const unsigned long long n = 1ull) {
    if constexpr (WAIT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned char* w = nullptr;   // lane p: rank p's window (a select chain: no dynamic index into the kernel arguments)
#pragma unroll
    for (int p = 0; p < P2P_MAXG; ++p) w = lane == p ? P.p2p_win[p] : w;
    if (lane < P.p2p_G)
        __hip_atomic_fetch_add((unsigned long long*)(w + 128 * (size_t)P.p2p_rank), n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// k_p2p_push: the generic form of the push, for chain kernels that wrote their results into this rank's OWN window only (every
// objective, every population).  One workgroup per unit of P2P_UNIT chains: copies the unit's records, values and slots of
// parity b to every other rank's window and arrives everywhere.  FROM_CTX: the source is the context's own arrays (the first
// publication after stepping in another form); else this rank's window.  (Nobody waits in here: whoever reads the windows next
// — the chain kernel that walks inline, or k_p2p_wait in front of a stand-alone resolution — waits for the arrivals.  Every wait
// therefore stands at the START of an iteration's work, which keeps contexts of one process that share a hardware queue live.)
template <bool FROM_CTX>
__global__ __launch_bounds__(256) void k_p2p_push(const KParams P, const int t, const double* __restrict__ rec_src) {
    const int b = t & 1;   // the state after iteration t, into parity t & 1
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int c0 = (int)blockIdx.x * P2P_UNIT, n = min(P2P_UNIT, P.N - c0);
    const int RW = P.RW;
    unsigned char* mine = P.p2p_self;
    const double* rs = FROM_CTX ? rec_src + (size_t)c0 * RW : (const double*)(mine + p2p_rec_off(P, b)) + (size_t)(P.offset + c0) * RW;
    const double* vs = (const double*)(mine + p2p_val_off(P, b)) + P.offset + c0;
#pragma unroll
    for (int p = 0; p < P2P_MAXG; ++p) {
        if (p >= P.p2p_G) break;
        unsigned char* w = P.p2p_win[p];
        const bool own = !FROM_CTX && p == P.p2p_rank;   // (its records and values are where they belong already)
        double* rd = (double*)(w + p2p_rec_off(P, b)) + (size_t)(P.offset + c0) * RW;
        if (!own)
            for (int i = tid; i < n * RW; i += 256) p2p_store8(rd + i, __builtin_bit_cast(unsigned long long, rs[i]));
        if (tid < n) {
            const double v = FROM_CTX ? rs[(size_t)tid * RW] : vs[tid];   // (FROM_CTX: the record's own value column)
            if (!own) p2p_store8((double*)(w + p2p_val_off(P, b)) + P.offset + c0 + tid, __builtin_bit_cast(unsigned long long, v));
            p2p_store8((uint2*)(w + p2p_slot_off(P, b)) + P.offset + c0 + tid, p2p_slot_word(v, (uint32_t)(P.offset + c0 + tid), t));
            if (v != v) { __hip_atomic_fetch_or((uint32_t*)(w + 128 * (size_t)P2P_MAXG), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave's own stores are out before the workgroup meets ...
    __syncthreads();                                   // ... so that the arrival below cannot overtake any of them
    if (tid < 64) p2p_arrive<false>(P, lane);   // (every wave waited for its own stores in front of the barrier)
}
// the wait alone (one wave): in front of a stand-alone exchange resolution that reads what chain kernels pushed from their epilogue
// (owed: arrivals of this rank that a chain kernel left to the next launch)
__global__ __launch_bounds__(64) void k_p2p_wait(const KParams P, const int t, const int owed) {
    if (owed) p2p_arrive<false>(P, (int)threadIdx.x, (unsigned long long)owed);
    if (p2p_wait_arrivals(P, (int)threadIdx.x) == 1 && threadIdx.x == 0) report_error(P, 3, t, P.offset);
}
