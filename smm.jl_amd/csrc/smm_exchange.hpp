// stand-alone exchangeMoves! kernels: k_exch_resolve_lds / _lvl / _lvl_soa / _lvl_big / _any, k_exch_plan_big, k_exch_apply — part of libsmmhip (included by smmhip.hip inside its anonymous namespace; gfx950 device code).
#pragma once
#define XTS(i) do { if (P.ts && tid == 0) P.ts[(size_t)8 * 60000 + (i)] = wall_clock64(); } while (0)
#ifdef SMM_TEST_HOOKS   // (the ticket kernel is a test reference: not in the shipped library)
// ------------------------------------------------------------------------------------------
// k_exch_resolve_lds: exchangeMoves! (AlgoBGP.jl:647-716) for N_global <= XLDS_MAX, one workgroup,
// all state in LDS.  The reference walks the K sampled pairs in order and swaps the two chains'
// last accepted records when value_i - value_j > min_improve_i (:688).  Pairs that share no chain
// commute, so the list is executed as a data-flow graph: ticket[c] counts the executed pairs of
// chain c and pair q = (i,j) runs exactly when ticket[i]==r_i && ticket[j]==r_j (all of its
// predecessors on both chains ran, none of its successors did); then it publishes ticket+1 on both
// chains (release/acquire at workgroup scope).  Critical path = longest dependency chain of the
// list (~log N) x one LDS round trip.  Output xres[g] = src | partner<<32: whose record chain g
// ends up with, and its last exchange partner (1-based, 0 = none).
// gathered: last accepted records of all chains, [Ng][RW].
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(XWG) void k_exch_resolve_lds(const KParams P, const int t, const double* __restrict__ gathered) {
    extern __shared__ __attribute__((aligned(16))) unsigned char xsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int Ng = P.Ng, RW = P.RW, K = P.plan_K;
    double* val = (double*)xsm;               // [Ng]
    uint32_t* ticket = (uint32_t*)(val + Ng);  // [Ng]
    uint16_t* src = (uint16_t*)(ticket + Ng);  // [Ng]
    uint16_t* partner = src + Ng;              // [Ng]
    const unsigned long long* __restrict__ plan = P.plan + (size_t)(t - P.plan_t0) * K;
    const double* __restrict__ plan_mi = P.plan_mi + (size_t)(t - P.plan_t0) * K;

    XTS(0);
    // this thread's pairs (list positions tid, tid+1024, ...): plan words and thresholds up front
    constexpr int MAXPP = XLDS_MAX / XWG;
    unsigned long long pws[MAXPP];
    double mis[MAXPP];
#pragma unroll
    for (int m = 0; m < MAXPP; ++m) {
        const int qq = tid + m * XWG;
        pws[m] = (qq < K) ? plan[qq] : 0ull;
        mis[m] = (qq < K) ? plan_mi[qq] : 0.0;
    }
    const double* __restrict__ vsrc = gathered ? gathered : P.vals;  // single shard: the compact value array
    const int vstride = gathered ? RW : 1;
    for (int g = tid; g < Ng; g += XWG) {
        val[g] = vsrc[(size_t)g * vstride];
        ticket[g] = 0;
        src[g] = (uint16_t)g;
        partner[g] = 0;
    }
    __syncthreads();
    XTS(1);
    int q = tid, m = 0;
    unsigned long long pw = pws[0];
    double mi = mis[0];
    unsigned spins = 0;
    while (true) {
        bool progressed = false;
        if (q < K) {
            const uint32_t i = (uint32_t)(pw & 0xffff), j = (uint32_t)((pw >> 16) & 0xffff);
            const uint32_t ri = (uint32_t)((pw >> 32) & 0xffff), rj = (uint32_t)(pw >> 48);
            const uint32_t ti = __hip_atomic_load(&ticket[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint32_t tj = __hip_atomic_load(&ticket[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (ti == ri && tj == rj) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                const double vi = val[i], vj = val[j];
                if (dist_fun_eval(P.dist_fun, vi, vj) > mi) {   // dist_fun (default -), :688
                    val[i] = vj; val[j] = vi;               // swap_ev_ij!, :739-744
                    const uint16_t si = src[i];
                    src[i] = src[j]; src[j] = si;
                    partner[i] = (uint16_t)(j + 1); partner[j] = (uint16_t)(i + 1);  // set_exchanged!, :747-748
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __hip_atomic_store(&ticket[i], ti + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_store(&ticket[j], tj + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                q += XWG;
                ++m;
#pragma unroll
                for (int k = 1; k < MAXPP; ++k)
                    if (m == k) { pw = pws[k]; mi = mis[k]; }
                progressed = true;
            }
        }
        if (__all(q >= K)) break;
        if (!__any(progressed)) {
            if (++spins > XSPIN_LIMIT) {  // cannot happen: the smallest pending list position is always runnable
                if (lane == 0) report_error(P, 3, t, 0);
                break;
            }
            if (!(P.dbg & 16)) __builtin_amdgcn_s_sleep(1);
        }
    }
    XTS(2);
    __syncthreads();
    XTS(3);
    for (int g = tid; g < Ng; g += XWG) P.xres[g] = (unsigned long long)src[g] | ((unsigned long long)partner[g] << 32);
    XTS(4);
}
#endif

// k_exch_resolve_lvl: the same result for N_global <= XLVL_MAX, executed level by level: the plan
// groups the pair list by dependency level (k_exch_plan); the pairs of one level touch pairwise
// disjoint chains, so a level is one parallel step and the walk needs (number of levels ~ log N)
// barriers.  Plan, values and thresholds are staged in LDS with coalesced loads up front.
template <int LWG>
__global__ __launch_bounds__(LWG) void k_exch_resolve_lvl(const KParams P, const int t, const double* __restrict__ gathered) {
    extern __shared__ __attribute__((aligned(16))) unsigned char xsm[];
    const int tid = threadIdx.x;
    const int Ng = P.Ng, RW = P.RW, K = P.plan_K;
    const int w = t - P.plan_t0;
    XSlot* slot = (XSlot*)xsm;                  // [Ng]
    double* mi = (double*)(slot + Ng);          // [K]
    uint32_t* pairs = (uint32_t*)(mi + K);      // [K]
    const uint32_t* __restrict__ g_off = P.lv_off + (size_t)w * (K + 2);   // level ends: wave-uniform scalar loads
    const uint32_t* __restrict__ g_pairs = P.lv_pairs + (size_t)w * K;
    const double* __restrict__ g_mi = P.lv_mi + (size_t)w * K;
    XTS(0);
    const unsigned long long cyc0 = clock64();
    // ONE round trip of global loads: values, plan and level ends are all requested before the first wait
    // (the values were written by other XCDs a moment ago and come from memory-side cache, ~1 us away;
    // a load-store loop would pay that latency once per trip).
    constexpr int PT = XLVL_MAX / LWG;
    const double* __restrict__ vsrc = gathered ? gathered : P.vals;  // single shard: the compact value array
    const int vstride = gathered ? RW : 1;
    const int lane = tid & 63;
    double v_[PT], mq_[PT];
    uint32_t pq_[PT];
#pragma unroll
    for (int r = 0; r < PT; ++r) {
        const int g = tid + r * LWG;
        v_[r] = g < Ng ? vsrc[(size_t)g * vstride] : 0.0;
    }
#pragma unroll
    for (int r = 0; r < PT; ++r) {
        const int q = tid + r * LWG;
        pq_[r] = q < K ? g_pairs[q] : 0u;
        mq_[r] = q < K ? g_mi[q] : 0.0;
    }
    const uint32_t ev = g_off[min(lane, K)];   // lane l: end of level l (entries past the last level are unused)
    const int nlev = (int)g_off[K + 1];
#pragma unroll
    for (int r = 0; r < PT; ++r) {
        const int g = tid + r * LWG;
        if (g < Ng) {
            XSlot s_;
            s_.val = v_[r];
            s_.src = (uint32_t)g;
            s_.partner = 0;
            slot[g] = s_;
        }
    }
#pragma unroll
    for (int r = 0; r < PT; ++r) {
        const int q = tid + r * LWG;
        if (q < K) { pairs[q] = pq_[r]; mi[q] = mq_[r]; }
    }
    for (int q = tid + PT * LWG; q < K; q += LWG) { pairs[q] = g_pairs[q]; mi[q] = g_mi[q]; }  // K > XLVL_MAX: injected long pair lists
    // Level ends: lane l of every wave holds the end of level l (one coalesced load, read back with
    // v_readlane).  The level loop stays ROLLED on purpose: the kernel runs once per iteration on a CU whose
    // instruction cache has been flushed by the chain kernel in between, so every byte of straight-line code
    // is an instruction-fetch miss.
    auto level_end = [&](int l) -> uint32_t {
        const int lc = min(l, nlev - 1);
        return lc < 64 ? (uint32_t)__builtin_amdgcn_readlane((int)ev, lc) : g_off[lc];
    };
    const int ltail = nlev;
    __syncthreads();
    XTS(1);
    uint32_t b = 0;
    int lvc = 0;
    unsigned long long* lts = (unsigned long long*)(pairs + K + (K & 1));   // [64] level stamps (debug)
    if (P.ts && tid == 0) lts[63] = clock64() - cyc0;
    uint32_t e = nlev > 0 ? level_end(0) : 0u;
    uint32_t e2 = nlev > 0 ? level_end(1) : 0u;
    uint32_t pw = (b + tid < e) ? pairs[b + tid] : 0u;
    double m = (b + tid < e) ? mi[b + tid] : 0.0;
    // (the loop twice: the default dist_fun `-` pays nothing for the menu)
    auto levels = [&](auto gen) {
        const int dk = decltype(gen)::value ? P.dist_fun : 0;
#pragma clang loop unroll(disable)
        for (int l = 0; l < ltail; ++l) {
            const uint32_t e3 = level_end(l + 2);
            // this thread's first pair of the next level (LDS) is fetched while this level runs
            const uint32_t pw2 = (e + tid < e2) ? pairs[e + tid] : 0u;
            const double m2 = (e + tid < e2) ? mi[e + tid] : 0.0;
            for (uint32_t pos = b + tid; pos < e && !(P.dbg & 32); pos += LWG) {
                if (pos != b + tid) { pw = pairs[pos]; m = mi[pos]; }
                const uint32_t i = pw & 0xffffu, j = pw >> 16;
                const XSlot si = slot[i], sj = slot[j];
                if (dist_fun_eval(dk, si.val, sj.val) > m) {   // dist_fun (default -), AlgoBGP.jl:688
                    XSlot ni, nj;                           // swap_ev_ij!, :739-744; set_exchanged!, :747-748
                    ni.val = sj.val; ni.src = sj.src; ni.partner = j + 1;
                    nj.val = si.val; nj.src = si.src; nj.partner = i + 1;
                    slot[i] = ni;
                    slot[j] = nj;
                }
            }
            b = e; e = e2; e2 = e3; pw = pw2; m = m2;
            if (!(P.dbg & 64)) __syncthreads();
            if (P.ts && tid == 0) { lts[lvc & 31] = clock64() - cyc0; ++lvc; }
        }
    };
    if (P.dist_fun != 0) levels(std::true_type{}); else levels(std::false_type{});
    if (P.ts && tid == 0) {
        P.ts[(size_t)8 * 60000 + 15] = lts[63];
        for (int l = 0; l < min(lvc, 32); ++l) P.ts[(size_t)8 * 60000 + 16 + l] = lts[l];
        P.ts[(size_t)8 * 60000 + 14] = (unsigned long long)ltail;
    }
    XTS(3);
    for (int g = tid; g < Ng; g += LWG) P.xres[g] = (unsigned long long)slot[g].src | ((unsigned long long)slot[g].partner << 32);
    XTS(4);
    if (P.ts && tid == 0) { P.ts[(size_t)8 * 60000 + 6] = clock64() - cyc0; P.ts[(size_t)8 * 60000 + 7] = (unsigned long long)nlev; }
}

// k_exch_resolve_lvl_soa: the level walk for XLVL_MAX < N_global <= XLDS_MAX (e.g. 2 GPUs x 4096 chains).  Same plan,
// same arithmetic; the chain slots are split (8-byte value, 4-byte src | partner << 16) so that 8192 chains and
// their pair list take 128 KB of LDS.  Thresholds: one scalar when min_improve is uniform, else from the plan.
template <int LWG>
__device__ inline void resolve_lvl_soa_body(const KParams& P, const int t, const double* __restrict__ gathered, unsigned char* xsm) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int Ng = P.Ng, RW = P.RW, K = P.plan_K;
    const int w = t - P.plan_t0;
    double* val = (double*)xsm;                 // [Ng]
    uint32_t* sp = (uint32_t*)(val + Ng);       // [Ng]
    uint32_t* pairs = sp + Ng;                  // [K]
    const uint32_t* __restrict__ g_off = P.lv_off + (size_t)w * (K + 2);
    const uint32_t* __restrict__ g_pairs = P.lv_pairs + (size_t)w * K;
    const double* __restrict__ g_mi = P.lv_mi + (size_t)w * K;
    const bool mi_u = P.mi_uniform != 0;
    const double mi_v = P.mi_value;
    const double* __restrict__ vsrc = gathered ? gathered : P.vals;
    const int vstride = gathered ? RW : 1;
    constexpr int PT = XLDS_MAX / LWG;
    double v_[PT];
    uint32_t pq_[PT];
#pragma unroll
    for (int r = 0; r < PT; ++r) {
        const int g = tid + r * LWG;
        v_[r] = g < Ng ? vsrc[(size_t)g * vstride] : 0.0;
    }
#pragma unroll
    for (int r = 0; r < PT; ++r) {
        const int q = tid + r * LWG;
        pq_[r] = q < K ? g_pairs[q] : 0u;
    }
    const uint32_t ev = g_off[min(lane, K)];
    const int nlev = (int)g_off[K + 1];
#pragma unroll
    for (int r = 0; r < PT; ++r) {
        const int g = tid + r * LWG;
        if (g < Ng) { val[g] = v_[r]; sp[g] = (uint32_t)g; }
    }
#pragma unroll
    for (int r = 0; r < PT; ++r) {
        const int q = tid + r * LWG;
        if (q < K) pairs[q] = pq_[r];
    }
    auto level_end = [&](int l) -> uint32_t {
        const int lc = min(l, nlev - 1);
        return lc < 64 ? (uint32_t)__builtin_amdgcn_readlane((int)ev, lc) : g_off[lc];
    };
    __syncthreads();
    uint32_t b = 0;
    uint32_t e = nlev > 0 ? level_end(0) : 0u;
    uint32_t e2 = nlev > 0 ? level_end(1) : 0u;
    uint32_t pw = (b + tid < e) ? pairs[b + tid] : 0u;
    double m = mi_u ? mi_v : ((b + tid < e) ? g_mi[b + tid] : 0.0);
    // (the loop twice: the default dist_fun `-` pays nothing for the menu)
    auto levels = [&](auto gen) {
        const int dk = decltype(gen)::value ? P.dist_fun : 0;
#pragma clang loop unroll(disable)
        for (int l = 0; l < nlev; ++l) {
            const uint32_t e3 = level_end(l + 2);
            const uint32_t pw2 = (e + tid < e2) ? pairs[e + tid] : 0u;
            const double m2 = mi_u ? mi_v : ((e + tid < e2) ? g_mi[e + tid] : 0.0);
            for (uint32_t pos = b + tid; pos < e; pos += LWG) {
                if (pos != b + tid) { pw = pairs[pos]; m = mi_u ? mi_v : g_mi[pos]; }
                const uint32_t i = pw & 0xffffu, j = pw >> 16;
                const double vi = val[i], vj = val[j];
                const uint32_t si = sp[i], sj = sp[j];
                if (dist_fun_eval(dk, vi, vj) > m) {    // dist_fun (default -), AlgoBGP.jl:688
                    val[i] = vj; val[j] = vi;               // swap_ev_ij!, :739-744; set_exchanged!, :747-748
                    sp[i] = (sj & 0xffffu) | ((j + 1) << 16);
                    sp[j] = (si & 0xffffu) | ((i + 1) << 16);
                }
            }
            b = e; e = e2; e2 = e3; pw = pw2; m = m2;
            __syncthreads();
        }
    };
    if (P.dist_fun != 0) levels(std::true_type{}); else levels(std::false_type{});
    for (int g = tid; g < Ng; g += LWG) {
        const uint32_t s_ = sp[g];
        P.xres[g] = (unsigned long long)(s_ & 0xffffu) | ((unsigned long long)(s_ >> 16) << 32);
    }
}

template <int LWG>
__global__ __launch_bounds__(LWG) void k_exch_resolve_lvl_soa(const KParams P, const int t, const double* __restrict__ gathered) {
    extern __shared__ __attribute__((aligned(16))) unsigned char xsm[];
    resolve_lvl_soa_body<LWG>(P, t, gathered, xsm);
}

// k_exch_resolve_lean: the lean walk of smm_walk_lean.hpp as a kernel of its own (one workgroup): one min_improve >= 0 for every
// chain, N_global <= 8192 (~7400 when it is not 0: 16-byte slots) — the sharded path at 1 and 2 GPUs x 4096 chains, single shards whose chain kernel does not walk
// inline (8192 chains; objectives other than objfunc_norm).  Same result as k_exch_resolve_lvl_soa, which it falls back to
// (as a function, in the same launch) when this iteration's plan has more than 31 levels or a chain value is NaN.
__global__ __launch_bounds__(XWG) void k_exch_resolve_lean(const KParams P, const int t, const double* __restrict__ gathered) {
    extern __shared__ __attribute__((aligned(16))) unsigned char xsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int Ng = P.Ng;
    const int w = t - P.plan_t0;
    const uint32_t* __restrict__ g_offp = P.lv_offp + (size_t)w * LV_OFFP;
    const uint4* __restrict__ g_pairs = (const uint4*)(P.lv_pairs_p + (size_t)w * P.plan_Kp);
    const double* __restrict__ vsrc = gathered ? gathered : P.vals;
    const int vstride = gathered ? P.RW : 1;
    const uint32_t Ng4 = (uint32_t)((Ng + 3) & ~3);
    const uint32_t pbase = 8u * (Ng4 + 4u);
    constexpr int PT = XLDS_MAX / XWG;                 // chains per lane
    constexpr int PR = (XLDS_MAX + 64 * LV_MAXLEV + 4 * XWG - 1) / (4 * XWG);   // rounds of 16-byte loads for the pair words
    XTS(0);
    const uint32_t ov = g_offp[min(lane, LV_OFFP - 1)];   // lane l: first word of level l; lane 33: levels; lane 34: the plan fits
    double v_[PT];
#pragma unroll
    for (int r = 0; r < PT; ++r) {   // one round trip: the chains' values (a strided read of the gathered records) and the pair words
        const int g = tid + r * XWG;
        v_[r] = g < Ng ? vsrc[(size_t)g * vstride] : 0.0;
    }
    uint4 p_[PR];
#pragma unroll
    for (int r = 0; r < PR; ++r) {
        const int q4 = tid + r * XWG;
        p_[r] = 4 * q4 < P.plan_Kp ? g_pairs[q4] : make_uint4(0u, 0u, 0u, 0u);
    }
    if (P.lean_wide) {   // one min_improve > 0 (or NaN): 16-byte slots of values (smm_walk_lean.hpp)
        const int nlev = __builtin_amdgcn_readlane((int)ov, 33);
        if (__builtin_amdgcn_readlane((int)ov, 34) == 0 || (uint32_t)(size_t)xsm != 0u) { resolve_lvl_soa_body<XWG>(P, t, gathered, xsm); return; }
        const uint32_t pbw = 16u * (Ng4 + 1u);
        uint4* slot = (uint4*)xsm;
#pragma unroll
        for (int r = 0; r < PT; ++r) {
            const int g = tid + r * XWG;
            if (g < Ng) slot[g] = make_uint4((uint32_t)__double2loint(v_[r]), (uint32_t)__double2hiint(v_[r]), (uint32_t)g, 0u);
        }
#pragma unroll
        for (int r = 0; r < PR; ++r) {
            const int q4 = tid + r * XWG;
            if (4 * q4 < P.plan_Kp) ((uint4*)(xsm + pbw))[q4] = p_[r];
        }
        if (tid == 0) slot[Ng4] = make_uint4(0u, 0u, 0u, 0u);   // the dummy pair's slot: 0 - 0 > min_improve is false
        const int ltail = lean_walk_tail(ov, nlev, lane);
        __syncthreads();
        XTS(1);
        if (P.lean_unit == 16) lean_walk_levels<XWG, 0, true>(nullptr, 0, pbw, ov, nlev, tid, ltail, P.mi_value);
        else lean_walk_levels<XWG, 1, true>(nullptr, 0, pbw, ov, nlev, tid, ltail, P.mi_value);
        __syncthreads();
        XTS(2);
#pragma unroll
        for (int r = 0; r < PT; ++r) {
            const int g = tid + r * XWG;
            if (g < Ng) {
                const uint32_t meta = slot[g].z;
                const uint32_t partner = P.lean_unit == 16 ? lean_partner<0, 4>(xsm, pbw, meta, (uint32_t)g) : lean_partner<1, 4>(xsm, pbw, meta, (uint32_t)g);
                P.xres[g] = (unsigned long long)(meta & 0xffffu) | ((unsigned long long)partner << 32);
            }
        }
        XTS(4);
        return;
    }
    uint32_t* s_nan = (uint32_t*)(xsm + 8u * (Ng4 + 2u));   // (a spare slot behind the dummy pair's)
    if (tid == 0) *s_nan = 0u;
    __syncthreads();
    bool nan = false;
#pragma unroll
    for (int r = 0; r < PT; ++r) nan = nan || v_[r] != v_[r];
    if (nan) *s_nan = 1u;
    const int nlev = __builtin_amdgcn_readlane((int)ov, 33);
    const bool fits = __builtin_amdgcn_readlane((int)ov, 34) != 0 && (uint32_t)(size_t)xsm == 0u;
    __syncthreads();
    const bool has_nan = *s_nan != 0u;
    __syncthreads();
    if (!fits || has_nan) { resolve_lvl_soa_body<XWG>(P, t, gathered, xsm); return; }
    uint2* slot = (uint2*)xsm;
#pragma unroll
    for (int r = 0; r < PT; ++r) {
        const int g = tid + r * XWG;
        if (g < Ng) slot[g] = make_uint2(order_key32(v_[r]), (uint32_t)g);
    }
#pragma unroll
    for (int r = 0; r < PR; ++r) {
        const int q4 = tid + r * XWG;
        if (4 * q4 < P.plan_Kp) ((uint4*)(xsm + pbase))[q4] = p_[r];
    }
    if (tid == 0) { slot[Ng4] = make_uint2(1u, 0u); slot[Ng4 + 1] = make_uint2(2u, 0u); }   // the dummy pair's slots: keys 1 < 2, "no swap"
    const int ltail = lean_walk_tail(ov, nlev, lane);
    __syncthreads();
    XTS(1);
    if (P.lean_unit == 8) lean_walk_levels<XWG, 0>(vsrc, vstride, pbase, ov, nlev, tid, ltail);
    else lean_walk_levels<XWG, 1>(vsrc, vstride, pbase, ov, nlev, tid, ltail);
    __syncthreads();
    XTS(2);
#pragma unroll
    for (int r = 0; r < PT; ++r) {
        const int g = tid + r * XWG;
        if (g < Ng) {
            const uint32_t meta = slot[g].y;
            const uint32_t partner = P.lean_unit == 8 ? lean_partner<0>(xsm, pbase, meta, (uint32_t)g) : lean_partner<1>(xsm, pbase, meta, (uint32_t)g);
            P.xres[g] = (unsigned long long)(meta & 0xffffu) | ((unsigned long long)partner << 32);
        }
    }
    XTS(4);
    if (P.ts && tid == 0) P.ts[(size_t)8 * 60000 + 7] = (unsigned long long)nlev;
}

// ------------------------------------------------------------------------------------------
// Large populations (8192 < N_global <= 65535, e.g. 8 GPUs x 4096 chains): the same level plan and
// level-synchronous walk with their working sets in global memory (the LDS of one CU is too small).
// k_exch_plan_big: one workgroup per iteration, scratch [blockIdx] in global memory; speed is not
// critical (runs ahead of the dependent loop, one window at a time).
// ------------------------------------------------------------------------------------------
struct BigPlanScratch {  // per workgroup
    uint32_t *cnt, *pw, *lvl;
    __host__ __device__ static size_t words(int Ng, int K) { return (size_t)(Ng + 4) + (size_t)K * 2; }
    __device__ void carve(uint32_t* base, int Ng, int K) { cnt = base; pw = cnt + Ng + 4; lvl = pw + K; }
};
// its LDS: the level every chain was last met at (16 bits: there are at most K <= 65535 levels), the pair counts of the first
// PLANBIG_HIST levels, a claim table of H words (H a power of two: what is left of 156 KiB, at most 16384)
constexpr int PLANBIG_HIST = 1024;
__host__ __device__ inline uint32_t plan_big_claims(int Ng) {
    const size_t left = (size_t)156 * 1024 - (((size_t)Ng * 2 + 15) & ~(size_t)15) - (size_t)PLANBIG_HIST * 4;
    uint32_t H = 16384;
    while ((size_t)H * 4 > left) H >>= 1;
    return H;
}
__host__ __device__ inline size_t plan_big_lds_bytes(int Ng) {
    return (((size_t)Ng * 2 + 15) & ~(size_t)15) + (size_t)PLANBIG_HIST * 4 + (size_t)plan_big_claims(Ng) * 4;
}

__device__ inline uint32_t block_excl_scan_step(uint32_t v, uint32_t* wsum, int tid, uint32_t& total) {
    // exclusive scan of one value per thread over the 1024-thread block
    const int lane = tid & 63, wave = tid >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    __syncthreads();
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    total = 0;
    for (int w = 0; w < XWG / 64; ++w) total += wsum[w];
    return base + incl - v;
}

// in-place exclusive scan of a[0..n) (n arbitrary), returns nothing; all threads must call
__device__ inline void block_excl_scan(uint32_t* a, int n, uint32_t* wsum, int tid) {
    uint32_t carry = 0;
    for (int b = 0; b < n; b += XWG) {
        const int i = b + tid;
        const uint32_t v = i < n ? a[i] : 0u;
        uint32_t total;
        const uint32_t ex = block_excl_scan_step(v, wsum, tid, total);
        if (i < n) a[i] = carry + ex;
        carry += total;
        __syncthreads();
    }
}

// The levels (pair q's level = 1 + the later of the levels its two chains were last met at) are found in list order, a batch of
// XWG consecutive pairs at a time, a lane per pair: a pair may take its level once no EARLIER pair of the batch that shares a chain
// with it is still waiting — every waiting pair puts its batch position into the claim slots of its two chains (LDS atomic min; the
// slots are hashed, a false conflict only costs a round), and whoever holds both of its slots is ready.  The earliest waiting pair
// always is, and a batch of 1024 random pairs out of 32768 chains is done in 3-4 rounds: O(K) work and ~100 barriers per iteration,
// where the Jacobi sweeps over predecessor links this replaces cost depth x K global accesses (1.7 ms per iteration at 32768).
__global__ __launch_bounds__(XWG) void k_exch_plan_big(const KParams P, const int t0, uint32_t* __restrict__ scratch,
                                                       uint32_t* __restrict__ lv_pairs, double* __restrict__ lv_mi,
                                                       uint32_t* __restrict__ lv_off, uint32_t* __restrict__ lv_rows,
                                                       uint32_t* __restrict__ lv_rowinfo) {
    extern __shared__ __align__(16) unsigned char pb_smem[];
    __shared__ uint32_t wsum[XWG / 64];
    __shared__ uint32_t s_nlev, s_pend[2];
    const int tid = threadIdx.x;
    const int t = t0 + blockIdx.x;
    const int Ng = P.Ng, K = P.plan_K;
    BigPlanScratch S;
    S.carve(scratch + (size_t)blockIdx.x * BigPlanScratch::words(Ng, K), Ng, K);
    uint16_t* last = (uint16_t*)pb_smem;
    uint32_t* hist = (uint32_t*)(pb_smem + (((size_t)Ng * 2 + 15) & ~(size_t)15));
    uint32_t* claim = hist + PLANBIG_HIST;
    const uint32_t Hm = plan_big_claims(Ng) - 1u;
    for (int c = tid; c < Ng + 4; c += XWG) S.cnt[c] = 0;
    for (int c = tid; c < (Ng + 1) / 2; c += XWG) ((uint32_t*)last)[c] = 0u;
    for (int c = tid; c < PLANBIG_HIST; c += XWG) hist[c] = 0u;
    for (uint32_t c = tid; c <= Hm; c += XWG) claim[c] = 0xffffffffu;
    if (tid == 0) { s_nlev = 0; s_pend[0] = 0; s_pend[1] = 0; }
    PairPerm pp;
    if (!P.pairtab) pp.init(P.seed, (uint32_t)t, (uint64_t)Ng * (uint64_t)(Ng - 1) / 2);
    __syncthreads();
    uint32_t mx = 0, rnd = 0;
    for (int q0 = 0; q0 < K; q0 += XWG) {
        const int q = q0 + tid;
        bool pend = q < K;
        uint32_t i = 0, j = 0;
        if (pend) {
            if (P.pairtab) {
                i = (uint32_t)P.pairtab[((size_t)(t - 1) * K + q) * 2];
                j = (uint32_t)P.pairtab[((size_t)(t - 1) * K + q) * 2 + 1];
            } else {
                int32_t a, b;
                pair_unrank(pp.eval((uint64_t)q), a, b);
                i = (uint32_t)a; j = (uint32_t)b;
            }
            S.pw[q] = i | (j << 16);
        }
        const uint32_t hi = i & Hm, hj = j & Hm;
        for (;;) {
            if (pend) { atomicMin(&claim[hi], (uint32_t)tid); atomicMin(&claim[hj], (uint32_t)tid); }
            __syncthreads();
            if (pend && claim[hi] == (uint32_t)tid && claim[hj] == (uint32_t)tid) {
                const uint32_t a = last[i], b = last[j];
                const uint32_t lv = 1u + (a > b ? a : b);
                last[i] = (uint16_t)lv; last[j] = (uint16_t)lv;
                const uint32_t r = lv < (uint32_t)PLANBIG_HIST ? atomicAdd(&hist[lv], 1u) : atomicAdd(&S.cnt[lv], 1u);
                S.lvl[q] = lv | (r << 16);       // r: its place among the pairs of its level (any order: they share no chain)
                mx = lv > mx ? lv : mx;
                claim[hi] = 0xffffffffu; claim[hj] = 0xffffffffu;
                pend = false;
            }
            if (pend) s_pend[rnd & 1] = 1u;
            if (tid == 0) s_pend[(rnd + 1) & 1] = 0u;   // (last read before this round's first barrier)
            __syncthreads();
            const bool more = s_pend[rnd & 1] != 0u;
            ++rnd;
            if (!more) break;
        }
    }
    atomicMax(&s_nlev, mx);
    __syncthreads();
    const int nlev = (int)s_nlev;
    for (int l = tid; l < PLANBIG_HIST && l <= nlev; l += XWG) S.cnt[l] = hist[l];
    __syncthreads();
    block_excl_scan(S.cnt, nlev + 1, wsum, tid);  // cnt[l] = pairs in levels < l (1-based l)
    uint32_t* o_off = lv_off + (size_t)blockIdx.x * (K + 2);
    for (int l = tid; l < nlev; l += XWG) o_off[l] = (l + 2 <= nlev) ? S.cnt[l + 2] : (uint32_t)K;
    if (tid == 0) o_off[K + 1] = (uint32_t)nlev;
    // the same list for k_exch_resolve_rows: every level padded to whole rows of 1024 words with dummy pairs that never swap (no
    // lane asks whether it has a pair; a lane's words are tid, tid + 1024, ...: a plain stride to fetch ahead)
    __shared__ uint32_t s_ls[LV_MAXLEV + 2], s_rs[LV_MAXLEV + 2];   // compact start / first row of 1-based level c
    __shared__ uint32_t s_rows_ok;
    if (tid == 0) {
        uint32_t ok = lv_rows != nullptr && nlev <= LV_MAXLEV ? 1u : 0u, rows = 0;
        unsigned long long endmask = 0ull;
        if (ok) {
            for (int c = 1; c <= nlev; ++c) {
                s_ls[c] = S.cnt[c]; s_rs[c] = rows;
                rows += ((c + 1 <= nlev ? S.cnt[c + 1] : (uint32_t)K) - S.cnt[c] + (uint32_t)XWG - 1u) / (uint32_t)XWG;
                if (rows >= 1u && rows <= 64u) endmask |= 1ull << (rows - 1u);
            }
            s_rs[nlev + 1] = rows;
            if (rows > (uint32_t)P.rows_cap || rows > 63u) ok = 0u;
        }
        s_rows_ok = ok;
        if (lv_rowinfo) {
            uint32_t* o = lv_rowinfo + (size_t)blockIdx.x * 4;
            o[0] = rows; o[1] = (uint32_t)endmask; o[2] = (uint32_t)(endmask >> 32); o[3] = ok;
        }
    }
    __syncthreads();
    const bool rows_ok = s_rows_ok != 0u;
    uint32_t* o_rows = lv_rows ? lv_rows + (size_t)blockIdx.x * P.rows_cap * XWG : nullptr;
    if (rows_ok) {
        const uint32_t dummy = (uint32_t)Ng | ((uint32_t)(Ng + 1) << 16);   // the two slots behind the chains': their keys say "no swap"
        for (uint32_t q = tid; q < s_rs[nlev + 1] * (uint32_t)XWG; q += XWG) o_rows[q] = dummy;
    }
    __syncthreads();
    uint32_t* o_pairs = lv_pairs + (size_t)blockIdx.x * K;
    double* o_mi = lv_mi + (size_t)blockIdx.x * K;
    for (int l = tid; l < PLANBIG_HIST && l <= nlev; l += XWG) hist[l] = S.cnt[l];   // (first position of level l)
    __syncthreads();
    for (int q = tid; q < K; q += XWG) {
        const uint32_t lr = S.lvl[q], lv = lr & 0xffffu, r = lr >> 16;
        const uint32_t pos = (lv < (uint32_t)PLANBIG_HIST ? hist[lv] : S.cnt[lv]) + r;
        const uint32_t w = S.pw[q];
        o_pairs[pos] = w;
        o_mi[pos] = P.min_improve_g[w & 0xffffu];
        if (rows_ok) o_rows[s_rs[lv] * (uint32_t)XWG + r] = w;
    }
}

// k_exch_resolve_lvl_big: level-synchronous walk with the 16-byte chain slots {value, src, partner} in global memory
// (one 16-byte load or store per chain access: a single CU's address path is what bounds this kernel).  The slots
// are shared by the waves of ONE workgroup only: plain loads and stores with the workgroup barrier between levels
// are coherent (all waves of a workgroup sit on one CU and share its write-through L1), so a level costs L1/L2
// round trips, not the memory round trips of agent-scope accesses (those made a level ~12 us).
__global__ __launch_bounds__(XWG) void k_exch_resolve_lvl_big(const KParams P, const int t, const double* __restrict__ gathered) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int Ng = P.Ng, RW = P.RW, K = P.plan_K;
    const int w = t - P.plan_t0;
    const uint32_t* __restrict__ g_off = P.lv_off + (size_t)w * (K + 2);
    const uint32_t* __restrict__ g_pairs = P.lv_pairs + (size_t)w * K;
    const double* __restrict__ g_mi = P.lv_mi + (size_t)w * K;
    XSlot* slot = (XSlot*)P.xslot;
    const bool mi_u = P.mi_uniform != 0;
    const double mi_v = P.mi_value;
    const uint32_t ev = g_off[min(lane, K)];   // lane l: end of level l
    const int nlev = (int)g_off[K + 1];
    for (int g = tid; g < Ng; g += XWG) {
        XSlot s_;
        s_.val = gathered[(size_t)g * RW]; s_.src = (uint32_t)g; s_.partner = 0;
        slot[g] = s_;
    }
    auto level_end = [&](int l) -> uint32_t {
        const int lc = min(l, nlev - 1);
        return lc < 64 ? (uint32_t)__builtin_amdgcn_readlane((int)ev, lc) : g_off[lc];
    };
    __syncthreads();
    uint32_t b = 0;
    constexpr int BATCH = 8;  // pairs of one level are independent: their loads are issued together
    // (the loop twice: the default dist_fun `-` pays nothing for the menu)
    auto levels = [&](auto gen) {
        const int dk = decltype(gen)::value ? P.dist_fun : 0;
#pragma clang loop unroll(disable)
        for (int l = 0; l < nlev; ++l) {
            const uint32_t e = level_end(l);
            for (uint32_t p0 = b + tid; p0 < e; p0 += XWG * BATCH) {
                uint32_t pw[BATCH];
                double m[BATCH];
                XSlot si[BATCH], sj[BATCH];
    #pragma unroll
                for (int u = 0; u < BATCH; ++u) {
                    const uint32_t pos = p0 + u * XWG;
                    pw[u] = pos < e ? g_pairs[pos] : 0u;
                    m[u] = mi_u ? mi_v : (pos < e ? g_mi[pos] : 0.0);
                }
    #pragma unroll
                for (int u = 0; u < BATCH; ++u) {   // (pair word 0 = chains (0,0): never swaps, see the guard below)
                    si[u] = slot[pw[u] & 0xffffu];
                    sj[u] = slot[pw[u] >> 16];
                }
    #pragma unroll
                for (int u = 0; u < BATCH; ++u) {
                    const uint32_t pos = p0 + u * XWG;
                    const uint32_t i = pw[u] & 0xffffu, j = pw[u] >> 16;
                    if (pos < e && dist_fun_eval(dk, si[u].val, sj[u].val) > m[u]) {  // dist_fun (default -), AlgoBGP.jl:688
                        XSlot ni, nj;                               // swap_ev_ij!, :739-744; set_exchanged!, :747-748
                        ni.val = sj[u].val; ni.src = sj[u].src; ni.partner = j + 1;
                        nj.val = si[u].val; nj.src = si[u].src; nj.partner = i + 1;
                        slot[i] = ni;
                        slot[j] = nj;
                    }
                }
            }
            b = e;
            __syncthreads();
        }
    };
    if (P.dist_fun != 0) levels(std::true_type{}); else levels(std::false_type{});
    for (int g = tid; g < Ng; g += XWG) {
        const XSlot s_ = slot[g];
        P.xres[g] = (unsigned long long)s_.src | ((unsigned long long)s_.partner << 32);
    }
}

// k_exch_resolve_any: the same result for any N_global, state in global memory, executed in
// barrier-separated dependency rounds: every pending pair bids (atomicMin of its list position) on
// both of its chains; a pair that wins both bids has no pending predecessor and is executed.
__global__ __launch_bounds__(XWG) void k_exch_resolve_any(const KParams P, const int t, const double* __restrict__ gathered) {
    const int tid = threadIdx.x;
    const int Ng = P.Ng, RW = P.RW;
    const int K = P.pairtab ? P.n_pairs_tab : n_exchange_pairs(Ng);
    for (int g = tid; g < Ng; g += XWG) {
        P.xval[g] = gathered[(size_t)g * RW];
        P.xsrc[g] = g;
        P.xpartner[g] = 0;
        P.xnext[g] = 0x7fffffff;
    }
    if (P.pairtab) {
        for (int q = tid; q < K; q += XWG) {
            P.xpairs[2 * q] = P.pairtab[((size_t)(t - 1) * K + q) * 2];
            P.xpairs[2 * q + 1] = P.pairtab[((size_t)(t - 1) * K + q) * 2 + 1];
        }
    } else {
        PairPerm pp;
        pp.init(P.seed, (uint32_t)t, (uint64_t)Ng * (uint64_t)(Ng - 1) / 2);
        for (int q = tid; q < K; q += XWG) {
            int32_t i, j;
            pair_unrank(pp.eval((uint64_t)q), i, j);
            P.xpairs[2 * q] = i;
            P.xpairs[2 * q + 1] = j;
        }
    }
    __syncthreads();
    int remaining = 1;
    while (remaining) {
        for (int q = tid; q < K; q += XWG) {
            const int i = P.xpairs[2 * q];
            if (i < 0) continue;  // executed
            const int j = P.xpairs[2 * q + 1];
            atomicMin(&P.xnext[i], q);
            atomicMin(&P.xnext[j], q);
        }
        __syncthreads();
        int mine = 0;
        for (int q = tid; q < K; q += XWG) {
            const int i = P.xpairs[2 * q];
            if (i < 0) continue;
            const int j = P.xpairs[2 * q + 1];
            if (__hip_atomic_load(&P.xnext[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == q &&
                __hip_atomic_load(&P.xnext[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == q) {
                const double vi = P.xval[i], vj = P.xval[j];
                if (dist_fun_eval(P.dist_fun, vi, vj) > P.min_improve_g[i]) {  // dist_fun (default -), :688
                    P.xval[i] = vj; P.xval[j] = vi;   // swap_ev_ij!, :739-744
                    const int si = P.xsrc[i];
                    P.xsrc[i] = P.xsrc[j]; P.xsrc[j] = si;
                    P.xpartner[i] = j + 1; P.xpartner[j] = i + 1;  // set_exchanged!, :747-748
                }
                P.xnext[i] = 0x7fffffff; P.xnext[j] = 0x7fffffff;
                P.xpairs[2 * q] = -1 - i;  // mark executed
            } else {
                mine = 1;
            }
        }
        remaining = __syncthreads_or(mine);
    }
    for (int g = tid; g < Ng; g += XWG)
        P.xres[g] = (unsigned long long)(unsigned)P.xsrc[g] | ((unsigned long long)(unsigned)P.xpartner[g] << 32);
}

// k_exch_apply (sharded path): set_eval!(ci, ej) + set_exchanged! of swap_ev_ij! (AlgoBGP.jl:734-749)
// for the local chains, reading the donor records from the all-gathered buffer [Ng][RW];
// rec = this shard's own post-accept records [N][RW], updated in place.
__global__ void k_exch_apply(const KParams P, const int t, const double* __restrict__ gathered, double* __restrict__ rec) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= P.N) return;
    if (*(const volatile unsigned long long*)P.err != ERR_NONE) return;   // the run stopped at the failing iteration
    const int N = P.N, RW = P.RW, HW = P.HW;
    const unsigned long long xr = P.xres[P.offset + c];
    const int partner = (int)(xr >> 32);
    if (partner == 0) return;
    const int s = (int)(unsigned)(xr & 0xffffffffu);
    const double* __restrict__ donor = gathered + (size_t)s * RW;
    double* csb = P.cs + (size_t)c * CSW;
    double* hrec = P.hrec + ((size_t)(t - 1) * N + c) * HW;
    const double value = donor[0];
    double bestv, bestid;
    if (value < csb[CS_BESTP]) { bestv = value; bestid = (double)t; }
    else { bestv = csb[CS_BESTP]; bestid = csb[CS_BESTPID]; }
    csb[CS_BEST] = bestv; csb[CS_BESTID] = bestid; csb[CS_WASX] = 1.0;
    hrec[H_VALUE] = value; hrec[H_PROB] = donor[1]; hrec[H_CURR] = value; hrec[H_BEST] = bestv;
    hrec[H_BESTID] = bestid; hrec[H_EXCH] = (double)partner; hrec[H_ACC] = 1.0; hrec[H_STATUS] = donor[2];
    for (int k = 0; k < P.np + P.nm; ++k) hrec[H_PARAMS + k] = donor[3 + k];
    for (int f = 0; f < RW; ++f) rec[(size_t)c * RW + f] = donor[f];
}

// ------------------------------------------------------------------------------------------
// k_exch_resolve_key: exchangeMoves! for 8192 < N_global <= 32768 (the 4- and 8-GPU populations) with the walk's state in
// LDS.  A chain's slot is 4 bytes — src (16 bits: whose record sits here) and a 16-bit ORDER KEY of that record's value — so
// 32768 chains take 128 KB.  The key is a bucket number of the value's high word (exponent | 20 mantissa bits, monotone for
// values >= 0) on a fixed scale (order_key16).  A pair whose two buckets decide the test `value_i - value_j > min_improve_i` (AlgoBGP.jl:688) for
// every pair of values in them is resolved from LDS alone; an undecided pair reads the two exact values from memory.  Swaps
// exchange the 4-byte slots and set the pair's bit in an LDS bitmap; the last exchange partner of every chain
// (set_exchanged!, :747-748) is recovered afterwards with one LDS atomic max per swapped endpoint (ordered by the position in
// the level order, which respects every chain's own order).  Same plan, same result as the other kernels.
// (The level walk of k_exch_resolve_lvl_big keeps 16-byte slots in global memory: one CU's address path bounds it at ~100 us
// for 32768 chains.)
// ------------------------------------------------------------------------------------------
constexpr int XKEY_MAX = 32768;
// up to XKEY_PARTNER_MAX chains the last partner of every chain has its own 2-byte LDS array, written in the swap itself
constexpr int XKEY_PARTNER_MAX = 24576;
__host__ __device__ inline size_t resolve_key_bytes(int Ng, int K) {
    return (size_t)Ng * 4 + (Ng <= XKEY_PARTNER_MAX ? (size_t)(Ng + 4) * 2 : (((size_t)K + 63) / 64) * 8) + 256;
}

// (order_key16 and the bounds of its buckets: smm_params.hpp)

// k_exch_keys: by the whole chip, before the one resolving workgroup starts: the value column of the gathered records
// ([Ng][RW], or the compact array of a single shard with RW = 1) as a compact array, and every chain's initial 4-byte slot
// (src = itself | key << 16).  A strided read and 32768 key computations by ONE workgroup cost more than the walk itself.
// (slots17_out / nan_flags: for k_exch_resolve_rows — key17 << 15 | src, and two words used in turn by iteration parity: this
// launch raises word t & 1 when a value is NaN and clears the other one for the next iteration)
__global__ void k_exch_keys(const double* __restrict__ gathered, const int RW, const int Ng, double* __restrict__ vals_out,
                            uint32_t* __restrict__ slots_out, uint32_t* __restrict__ slots17_out, uint32_t* __restrict__ nan_flags, const int t) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g == 0 && nan_flags) nan_flags[(t + 1) & 1] = 0u;
    if (g >= Ng) return;
    const double v = gathered[(size_t)g * RW];
    if (vals_out != gathered) vals_out[g] = v;
    slots_out[g] = (uint32_t)g | (order_key16(v) << 16);
    if (slots17_out) {
        slots17_out[g] = (uint32_t)g | (order_key17(v) << 15);
        if (v != v) atomicOr(&nan_flags[t & 1], 1u);
    }
}

template <bool PLDS>   // PLDS: partners in LDS (N_global <= XKEY_PARTNER_MAX); else the bitmap + partner pass
__device__ inline void resolve_key_body(const KParams& P, const int t, const double* __restrict__ vals, const uint32_t* __restrict__ slots0,
                                        unsigned char* xsm) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Ng = P.Ng, K = P.plan_K;
    const int w = t - P.plan_t0;
    uint32_t* state = (uint32_t*)xsm;                                   // [Ng] src | key << 16
    unsigned long long* bits = (unsigned long long*)(state + Ng + (Ng & 1));   // [(K+63)/64] swapped pairs, by level-order position
    uint16_t* partner = (uint16_t*)bits;                                // PLDS: [Ng] last exchange partner + 1 instead
    const int nwords = PLDS ? 0 : (K + 63) / 64;
    const uint32_t* __restrict__ g_off = P.lv_off + (size_t)w * (K + 2);
    const uint32_t* __restrict__ g_pairs = P.lv_pairs + (size_t)w * K;
    const double* __restrict__ g_mi = P.lv_mi + (size_t)w * K;
    const bool mi_u = P.mi_uniform != 0;
    const double mi_v = P.mi_value;
    const uint32_t ev = g_off[min(lane, K)];   // lane l: end of level l
    const int nlev = (int)g_off[K + 1];
    constexpr int PT = XKEY_MAX / XWG;
    XTS(0);
    const unsigned long long cyc0 = clock64();
    // This thread's pairs are the list positions tid, tid + 1024, ... (a level is a contiguous range of positions).  Pair n sits
    // in register q[n mod 3], requested three pairs ahead: the register a pair is consumed from is refilled at once with the
    // pair three further on, so no load is ever waited for inside a level (a queue that shifts would wait at every shift).
    uint32_t pn = (uint32_t)tid;                 // this lane's next position
    uint32_t q0 = pn < (uint32_t)K ? g_pairs[pn] : 0u;
    uint32_t q1 = pn + XWG < (uint32_t)K ? g_pairs[pn + XWG] : 0u;
    uint32_t q2 = pn + 2 * XWG < (uint32_t)K ? g_pairs[pn + 2 * XWG] : 0u;
    if (slots0 == nullptr) {   // nobody made the initial slots (the accept step wrote the rows walk's only, and this is its rare fallback): from the values
        for (int g = tid; g < Ng; g += XWG) state[g] = (uint32_t)g | (order_key16(vals[g]) << 16);
    } else {
        typedef unsigned int u32x4s_t __attribute__((ext_vector_type(4)));
        const u32x4s_t* __restrict__ s4 = (const u32x4s_t*)slots0;     // the initial slots, made by k_exch_keys: 16 bytes per lane
        u32x4s_t* d4 = (u32x4s_t*)state;
        constexpr int P4 = PT / 4;
        u32x4s_t v_[P4];
#pragma unroll
        for (int r = 0; r < P4; ++r) {
            const int g4 = tid + r * XWG;
            v_[r] = 4 * g4 + 3 < Ng ? s4[g4] : u32x4s_t{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int r = 0; r < P4; ++r) {
            const int g4 = tid + r * XWG;
            if (4 * g4 + 3 < Ng) d4[g4] = v_[r];
        }
        for (int g = (Ng & ~3) + tid; g < Ng; g += XWG) state[g] = slots0[g];   // a ragged tail
    }
    for (int q = tid; q < nwords; q += XWG) bits[q] = 0ull;
    if constexpr (PLDS)
        for (int g = tid; g < (Ng + 1) / 2; g += XWG) ((uint32_t*)partner)[g] = 0u;
    auto level_end = [&](int l) -> uint32_t {
        const int lc = min(l, nlev - 1);
        return lc < 64 ? (uint32_t)__builtin_amdgcn_readlane((int)ev, lc) : g_off[lc];
    };
    __syncthreads();
    XTS(1);
    if (P.ts && tid == 0) P.ts[(size_t)8 * 60000 + 15] = clock64() - cyc0;
    uint32_t e = 0;
#pragma clang loop unroll(disable)
    for (int l = 0; l < nlev; ++l) {
        e = level_end(l);
        // A wave's lanes hold consecutive positions, but a level may end inside them: the lower lanes are then one pair ahead.
        // Trips are counted from the position of the lane furthest behind (lane 63); a lane takes part in the trip that
        // covers its own position, so that bit k of the ballot below always belongs to position pb + k.
        const uint32_t pb0 = (uint32_t)__builtin_amdgcn_readlane((int)pn, 63) - 63u;
        auto trip = [&](uint32_t& q, const uint32_t pb) {
            const bool live = pn == pb + (uint32_t)lane && pn < e;
            const uint32_t pw = q;
            const double m = mi_u ? mi_v : (live ? g_mi[pn] : 0.0);
            const uint32_t i = pw & 0xffffu, j = pw >> 16;
            const uint32_t si = live ? state[i] : 0u, sj = live ? state[j] : 0u;
            const uint32_t ki = si >> 16, kj = sj >> 16;
            bool sure, swap;
            if (ki == 0xffffu || kj == 0xffffu) {
                sure = false; swap = false;
            } else if (m == 0.0) {            // distinct buckets order the values themselves: v_i > v_j <=> v_i - v_j > 0
                sure = ki != kj; swap = ki > kj;
            } else {                          // every (v_i, v_j) of the two buckets on the same side of the threshold?
                const bool yes = (order_key_lo(ki) - order_key_hi(kj)) > m, no = (order_key_hi(ki) - order_key_lo(kj)) <= m;   // rounding is monotone: bounds carry over
                sure = yes || no; swap = yes;
            }
            if (live && !sure) {              // the exact values (AlgoBGP.jl:688)
                const double vi = vals[si & 0xffffu], vj = vals[sj & 0xffffu];
                swap = vi - vj > m;
            }
            swap = swap && live;
            if (swap) {                       // swap_ev_ij!, :739-744
                state[i] = sj; state[j] = si;
                if constexpr (PLDS) { partner[i] = (uint16_t)(j + 1); partner[j] = (uint16_t)(i + 1); }   // set_exchanged!, :747-748
            }
            if constexpr (!PLDS) {
                const unsigned long long mask = __ballot(swap);
                if (mask && lane == 0) {      // bits pb .. pb+63 of the bitmap (two words; other waves share them)
                    const uint32_t wi = pb >> 6, sh = pb & 63u;
                    atomicOr(&bits[wi], mask << sh);
                    if (sh && (int)wi + 1 < nwords) atomicOr(&bits[wi + 1], mask >> (64u - sh));
                }
            }
            if (live) {                       // this lane's pair is done: its register takes the pair three further on
                pn += XWG;
                q = pn + 2 * XWG < (uint32_t)K ? g_pairs[pn + 2 * XWG] : 0u;
            }
        };
        for (uint32_t pb = pb0; pb < e; pb += XWG) {
            const uint32_t nm = ((pb - (uint32_t)(wave * 64)) / XWG) % 3u;   // wave-uniform
            if (nm == 0) trip(q0, pb);
            else if (nm == 1) trip(q1, pb);
            else trip(q2, pb);
        }
        __syncthreads();
        if (P.ts && tid == 0 && l < 40) P.ts[(size_t)8 * 60000 + 16 + l] = clock64() - cyc0;
    }
    XTS(2);
    if (P.ts && tid == 0) { P.ts[(size_t)8 * 60000 + 7] = (unsigned long long)nlev; P.ts[(size_t)8 * 60000 + 14] = (unsigned long long)XKEY_SHIFT; }
    // ---- result: src from the slots; the last exchange partner from the bitmap ----
    uint32_t src_[PT];
#pragma unroll
    for (int r = 0; r < PT; ++r) {
        const int g = tid + r * XWG;
        src_[r] = g < Ng ? (state[g] & 0xffffu) : 0u;
    }
    if constexpr (PLDS) {
        XTS(3);
#pragma unroll
        for (int r = 0; r < PT; ++r) {
            const int g = tid + r * XWG;
            if (g < Ng) P.xres[g] = (unsigned long long)src_[r] | ((unsigned long long)partner[g] << 32);
        }
    } else {
        uint32_t pr_[PT];
#pragma unroll
        for (int r = 0; r < PT; ++r) {       // the pair list once more, one round trip
            const int p0 = tid + r * XWG;
            pr_[r] = p0 < K ? g_pairs[p0] : 0u;
        }
        __syncthreads();
        uint32_t* last = state;                   // [Ng] max over the chain's swapped pairs of (position + 1) << 16 | (partner + 1)
#pragma unroll
        for (int r = 0; r < PT; ++r) {
            const int g = tid + r * XWG;
            if (g < Ng) last[g] = 0u;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < PT; ++r) {
            const int p0 = tid + r * XWG;
            if (p0 < K && ((bits[p0 >> 6] >> (p0 & 63)) & 1ull)) {
                const uint32_t i = pr_[r] & 0xffffu, j = pr_[r] >> 16;
                atomicMax(&last[i], ((uint32_t)(p0 + 1) << 16) | (j + 1));   // set_exchanged!, :747-748: the later pair wins
                atomicMax(&last[j], ((uint32_t)(p0 + 1) << 16) | (i + 1));
            }
        }
        __syncthreads();
        XTS(3);
#pragma unroll
        for (int r = 0; r < PT; ++r) {
            const int g = tid + r * XWG;
            if (g < Ng) P.xres[g] = (unsigned long long)src_[r] | ((unsigned long long)(last[g] & 0xffffu) << 32);
        }
    }
    XTS(4);
    if (P.ts && tid == 0) P.ts[(size_t)8 * 60000 + 6] = clock64() - cyc0;
}

template <bool PLDS>
__global__ __launch_bounds__(XWG) void k_exch_resolve_key(const KParams P, const int t, const double* __restrict__ vals,
                                                          const uint32_t* __restrict__ slots0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char xsm[];
    resolve_key_body<PLDS>(P, t, vals, slots0, xsm);
}

// ------------------------------------------------------------------------------------------
// k_exch_resolve_rows: exchangeMoves! for 8192 < N_global <= 32768 with min_improve == 0 for every chain (the 4- and 8-GPU
// populations), written like the lean walk of smm_walk_lean.hpp for the number of instructions per pair — with 32768 pairs on one
// CU the walk is bound by the instructions its 16 waves can issue (k_exch_resolve_key: ~40 per wave and 64 pairs).
//   * the plan (k_exch_plan_big) pads every level to whole rows of 1024 words with dummy pairs that never swap: every lane has a
//     pair in every row, a lane's words are tid, tid + 1024, ... (all of them — at most 63 — fetched into registers at the start),
//     and one bit per row says whether a barrier follows (the last row of a level);
//   * a chain slot is 4 bytes in LDS, order_key17(value) << 15 | src: `value_i - value_j > 0` is one unsigned compare of the two
//     slots whenever their keys differ (one XOR and one compare tell), else the exact values are read from memory; a swap
//     exchanges the two words;
//   * set_exchanged! (AlgoBGP.jl:747-748): a wave's ballot of the swaps of a row is one 8-byte LDS word of its own (no atomics);
//     the last partner of every chain is recovered afterwards (one LDS max per swapped endpoint, the later pair wins).
// Falls back to k_exch_resolve_key's body when the plan does not fit (more than 31 levels or 63 rows) or a value is NaN.
// LDS: slots u32[Ng + 2 (+ pad)] | swap ballots u64[rows_cap][16].
// ------------------------------------------------------------------------------------------
constexpr int XROWS_MAX = 63;
// (n_own > 0: the OWN form's partner array over the shard's chains instead of the ballots)
__host__ __device__ inline size_t resolve_rows_bytes(int Ng, int K, int rows_cap, int n_own = 0) {
    const size_t a = (((size_t)Ng + 2 + 3) & ~(size_t)3) * 4 +
                     (Ng <= XKEY_PARTNER_MAX ? ((size_t)Ng + 4) * 2 : std::max((size_t)rows_cap * 16 * 8, ((size_t)n_own + 4) * 2));
    const size_t b = resolve_key_bytes(Ng, K);
    return a > b ? a : b;
}
// WIN (the p2p form of a sharded run, smm_p2p.hpp): the initial slots are the tagged 8-byte slots every rank's accept step has
// stored into THIS rank's window — read past the caches and validated word by word (a slot that has not landed yet is read
// again), so that nobody waits for anybody in a kernel of its own and no pre-pass makes keys; exact values (ties, the
// fallback) are the window's self-validating values.  A value whose key says "negative, infinite or NaN" sends the whole
// iteration to the fallback (k_exch_resolve_key's body on the exact values, unpacked here).
// OWN (a shard of a population too large for PLDS, e.g. 4096 of 32768 chains): a rank needs the last partner of ITS chains only — a
// 2-byte array over them fits next to the slots, the swap writes it for the endpoints in the shard's range, and the second pass over
// the rows (a third of the kernel) is gone.
template <bool PLDS, bool WIN = false, bool OWN = false>   // PLDS: the last partner of every chain in a 2-byte LDS array of its own, written by the swap (N_global <= XKEY_PARTNER_MAX)
__global__ __launch_bounds__(XWG) void k_exch_resolve_rows(const KParams P, const int t, const double* __restrict__ vals,
                                                           const uint32_t* __restrict__ slots16, const uint32_t* __restrict__ slots17,
                                                           uint32_t* __restrict__ nan_flags) {
    extern __shared__ __attribute__((aligned(16))) unsigned char xsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Ng = P.Ng;
    const int w = t - P.plan_t0;
    const uint32_t* __restrict__ info = P.lv_rowinfo + (size_t)w * 4;
    XTS(0);
    const int nrows = (int)info[0];
    const unsigned long long endmask = (unsigned long long)info[1] | ((unsigned long long)info[2] << 32);
    if ((uint32_t)(size_t)xsm != 0u) {   // the walk's LDS addresses count from 0: a static __shared__ object in this kernel breaks it (loud)
        if (tid == 0) report_error(P, 3, t + 1, 0);
        return;
    }
    if constexpr (!WIN) {
        if (tid == 0) nan_flags[(t + 1) & 1] = 0u;   // (the next iteration's word: its accept step or k_exch_keys raises it)
        if (info[3] == 0u || nan_flags[t & 1] != 0u) {
            resolve_key_body<PLDS>(P, t, vals, slots16, xsm);
            return;
        }
    }
    uint32_t* slot = (uint32_t*)xsm;
    const uint32_t sl_words = ((uint32_t)Ng + 2u + 3u) & ~3u;
    unsigned long long* bits = (unsigned long long*)(xsm + 4u * sl_words);   // !PLDS: [nrows][16] ballots of the swaps
    uint16_t* partner = (uint16_t*)(xsm + 4u * sl_words);                     // PLDS: [Ng] last exchange partner + 1
    const uint32_t pbase = 4u * sl_words;
    constexpr int PT = XKEY_MAX / XWG;   // chains per lane
    // this lane's pair words: row r sits in register q[r mod 3], requested three rows ahead
    // (inline asm: the compiler, which sees loads in the rare tie branch of a row too, would wait for ALL loads in flight at the head
    // of the loop — 14 round trips to the L2 in 42 rows; here the wait is for the oldest of the three only.  Fetches past the
    // last row are clamped to it: always three in flight, so that the count is right.)
    const int rl = nrows > 0 ? nrows - 1 : 0;
    // (a buffer load: the row is a scalar offset, the lane a constant vector offset — no address arithmetic per fetch)
    typedef int i32x4_t __attribute__((ext_vector_type(4)));
    const unsigned long long rows_base = (unsigned long long)(P.lv_rows + (size_t)w * P.rows_cap * XWG);
    const i32x4_t rows_rsrc = {(int)(unsigned)rows_base, (int)(unsigned)((rows_base >> 32) & 0xffffu), (int)((unsigned)P.rows_cap * XWG * 4u), 0x00020000};
    const uint32_t lane_off = 4u * (uint32_t)tid;
    auto fetch = [&](uint32_t& q, const int rr) {
        asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(q) : "v"(lane_off), "s"(rows_rsrc), "s"(min(rr, rl) * (XWG * 4)) : "memory");
    };
    uint32_t q0, q1, q2;
    if constexpr (WIN) {
        // the tagged 4-byte slots (order_key17(value) << 15 | tag15, smm_p2p.hpp) of this rank's window, four per 16-byte piece: all of
        // a lane's pieces requested together, then looked at one by one
        constexpr int PP = PT / 4;
        const uint4* g_slots = (const uint4*)(P.p2p_self + p2p_slot4_off(P, t & 1));
        const uint32_t want = p2p_tag15(P, t);
        const int last_piece = (Ng - 1) / 4;
        p2p_u32x4 s_[PP];
#pragma unroll
        for (int r = 0; r < PP; ++r) {
            const uint4* a = g_slots + min(tid + r * XWG, last_piece);   // (lanes past the end read the last piece: not used)
            asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(s_[r]) : "v"(a) : "memory");
        }
        static_assert(PP == 8, "the wait below names eight pieces");
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(s_[0]), "+v"(s_[1]), "+v"(s_[2]), "+v"(s_[3]), "+v"(s_[4]), "+v"(s_[5]), "+v"(s_[6]), "+v"(s_[7]) :: "memory");
        bool odd = false;
        auto slot17 = [&](const uint32_t w4, const uint32_t g) -> uint32_t {   // key17 << 15 | chain; the last bucket may be a NaN
            if ((w4 >> 15) == XKEY17_TOP) odd = true;
            return (w4 & ~0x7fffu) | g;
        };
#pragma unroll
        for (int r = 0; r < PP; ++r) {
            const int pc = tid + r * XWG, g = 4 * pc;
            if (g < Ng) {
                p2p_u32x4 q = s_[r];
                auto ok = [&]() {
                    return (q.x & 0x7fffu) == want && (g + 1 >= Ng || (q.y & 0x7fffu) == want) && (g + 2 >= Ng || (q.z & 0x7fffu) == want) &&
                           (g + 3 >= Ng || (q.w & 0x7fffu) == want);
                };
                if (__builtin_expect(!ok(), 0)) {   // somebody's stores are still on their way
                    const unsigned long long t0 = wall_clock64();
                    do {
                        __builtin_amdgcn_s_sleep(1);
                        const uint4 q4 = p2p_load16_sys(g_slots + pc);
                        q = p2p_u32x4{q4.x, q4.y, q4.z, q4.w};
                        if (const int o = p2p_spin_over(P, t0, t)) { if (o == 1) report_error(P, 3, t + 1, g); odd = true; break; }
                    } while (!ok());
                }
                const p2p_u32x4 w4 = {slot17(q.x, (uint32_t)g), g + 1 < Ng ? slot17(q.y, (uint32_t)g + 1u) : 0u,
                                      g + 2 < Ng ? slot17(q.z, (uint32_t)g + 2u) : 0u, g + 3 < Ng ? slot17(q.w, (uint32_t)g + 3u) : 0u};
                ((p2p_u32x4*)xsm)[pc] = w4;
            }
        }
        // (no static LDS in this kernel: the walk addresses the dynamic block from 0.  Every wave leaves its verdict in the words
        // behind the slots, which nothing uses before the walk)
        uint32_t* oddw = (uint32_t*)(xsm + 4u * ((((uint32_t)Ng + 2u + 3u) & ~3u)));
        const bool wave_odd = __ballot(odd) != 0ull;
        if (lane == 0) oddw[wave] = wave_odd ? 1u : 0u;
        __syncthreads();
        uint32_t any_odd = 0u;
#pragma unroll
        for (int wv = 0; wv < XWG / 64; ++wv) any_odd |= oddw[wv];
        __syncthreads();   // (read by everybody before the fallback or the walk reuse the words)
        if (info[3] == 0u || any_odd != 0u) {
            // the fallback on exact values: unpacked into the window's plain value array (this workgroup is its only reader)
            double* pv = (double*)(P.p2p_self + p2p_val_off(P, t & 1));
            for (int g = tid; g < Ng; g += XWG) pv[g] = p2p_ll_value(P, t, (uint32_t)g);
            __syncthreads();
            resolve_key_body<PLDS>(P, t, pv, nullptr, xsm);
            return;
        }
    }
    fetch(q0, 0); fetch(q1, 1); fetch(q2, 2);
    if constexpr (!WIN) {
        typedef unsigned int u32x4s_t __attribute__((ext_vector_type(4)));
        const u32x4s_t* __restrict__ s4 = (const u32x4s_t*)slots17;     // made by k_exch_keys: 16 bytes per lane and round
        constexpr int P4 = PT / 4;
        u32x4s_t v_[P4];
#pragma unroll
        for (int r = 0; r < P4; ++r) {
            const int g4 = tid + r * XWG;
            v_[r] = 4 * g4 < Ng ? s4[g4] : u32x4s_t{0u, 0u, 0u, 0u};   // (the array is padded to a multiple of 4)
        }
#pragma unroll
        for (int r = 0; r < P4; ++r) {
            const int g4 = tid + r * XWG;
            if (4 * g4 < Ng) ((u32x4s_t*)slot)[g4] = v_[r];
        }
    }
    __syncthreads();
    if (tid == 0) { slot[Ng] = 1u << 15; slot[Ng + 1] = 2u << 15; }   // the dummy pair's slots: keys 1 < 2, "no swap" (behind the staged words)
    static_assert(!(PLDS && OWN), "one partner array or the other");
    if constexpr (PLDS)
        for (int g = tid; g < (Ng + 1) / 2; g += XWG) ((uint32_t*)partner)[g] = 0u;
    if constexpr (OWN)
        for (int g = tid; g < (P.N + 1) / 2; g += XWG) ((uint32_t*)partner)[g] = 0u;
    const uint32_t own0 = (uint32_t)P.offset, ownN = (uint32_t)P.N;
    __syncthreads();
    XTS(1);
    auto row = [&](const uint32_t pw, const int r) {
        const uint32_t ai = (pw & 0xffffu) << 2, aj = (pw >> 16) << 2;
        uint32_t si, sj;
        asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(si), "=&v"(sj) : "v"(ai), "v"(aj) : "memory");
        bool swap = si > sj;
        const bool tie = (si ^ sj) < 0x8000u;
        if (__builtin_expect(__ballot(tie) != 0ull, 0)) {   // the keys do not decide: the exact values (dist_fun = -, AlgoBGP.jl:688)
            if (tie) {
                if constexpr (WIN) swap = p2p_ll_value(P, t, si & 0x7fffu) - p2p_ll_value(P, t, sj & 0x7fffu) > 0.0;
                else swap = vals[si & 0x7fffu] - vals[sj & 0x7fffu] > 0.0;
            }
        }
        if (swap) {   // swap_ev_ij!, :739-744
            asm volatile("ds_write_b32 %0, %2\n\tds_write_b32 %1, %3" :: "v"(ai), "v"(aj), "v"(sj), "v"(si) : "memory");
            if constexpr (PLDS) {   // set_exchanged!, :747-748
                const uint32_t pi = pbase + (ai >> 1), pj = pbase + (aj >> 1);
                asm volatile("ds_write_b16 %0, %2\n\tds_write_b16 %1, %3" :: "v"(pi), "v"(pj), "v"((pw >> 16) + 1u), "v"((pw & 0xffffu) + 1u) : "memory");
            }
            if constexpr (OWN) {    // ... for the endpoints that are this shard's
                const uint32_t ci = (pw & 0xffffu) - own0, cj = (pw >> 16) - own0;
                if (ci < ownN) partner[ci] = (uint16_t)((pw >> 16) + 1u);
                if (cj < ownN) partner[cj] = (uint16_t)((pw & 0xffffu) + 1u);
            }
        }
        if constexpr (!PLDS && !OWN) {
            const unsigned long long m = __ballot(swap);
            if (lane == 0) bits[r * 16 + wave] = m;
        }
        if ((endmask >> r) & 1ull) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }   // the level ends here
    };
    {
        int r = 0;
#pragma clang loop unroll(disable)
        for (; r + 3 <= nrows; r += 3) {
            asm volatile("s_waitcnt vmcnt(2)" : "+v"(q0) :: "memory");
            row(q0, r);
            fetch(q0, r + 3);
            asm volatile("s_waitcnt vmcnt(2)" : "+v"(q1) :: "memory");
            row(q1, r + 1);
            fetch(q1, r + 4);
            asm volatile("s_waitcnt vmcnt(2)" : "+v"(q2) :: "memory");
            row(q2, r + 2);
            fetch(q2, r + 5);
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(q0), "+v"(q1), "+v"(q2) :: "memory");
        if (r < nrows) row(q0, r);
        if (r + 1 < nrows) row(q1, r + 1);
    }
    __syncthreads();
    XTS(2);
    // ---- result: src from the slots; the last exchange partner from the swap's own array, or from the ballots ----
    // (the bulk passes over the chains in 16-byte LDS operations, four consecutive chains per lane and round — as 4-byte ones they
    // were a third of the kernel: 1024 LDS instructions per pass)
    typedef unsigned int u32x4r_t __attribute__((ext_vector_type(4)));
    constexpr int PT4 = PT / 4;
    u32x4r_t src4[PT4];
#pragma unroll
    for (int r = 0; r < PT4; ++r) {
        const int g4 = tid + r * XWG;
        src4[r] = 4 * g4 < Ng ? ((const u32x4r_t*)slot)[g4] & 0x7fffu : u32x4r_t{0u, 0u, 0u, 0u};
    }
    auto put4 = [&](const int g4, const u32x4r_t sv, const uint32_t p0, const uint32_t p1, const uint32_t p2, const uint32_t p3) {
        const int g = 4 * g4;   // chains g .. g + 3 (the last group of a population that is no multiple of 4 is ragged)
        if (g + 3 < Ng) {
            ((u32x4r_t*)P.xres)[2 * g4] = u32x4r_t{sv.x, p0, sv.y, p1};
            ((u32x4r_t*)P.xres)[2 * g4 + 1] = u32x4r_t{sv.z, p2, sv.w, p3};
        } else {
            if (g < Ng) P.xres[g] = (unsigned long long)sv.x | ((unsigned long long)p0 << 32);
            if (g + 1 < Ng) P.xres[g + 1] = (unsigned long long)sv.y | ((unsigned long long)p1 << 32);
            if (g + 2 < Ng) P.xres[g + 2] = (unsigned long long)sv.z | ((unsigned long long)p2 << 32);
        }
    };
    if constexpr (PLDS) {
        XTS(3);
#pragma unroll
        for (int r = 0; r < PT4; ++r) {
            const int g4 = tid + r * XWG;
            if (4 * g4 < Ng) {
                const uint2 pp = ((const uint2*)partner)[g4];   // four 2-byte entries (the array is padded)
                put4(g4, src4[r], pp.x & 0xffffu, pp.x >> 16, pp.y & 0xffffu, pp.y >> 16);
            }
        }
    } else if constexpr (OWN) {
        XTS(3);
        auto own_partner = [&](const uint32_t g) -> uint32_t { return g - own0 < ownN ? (uint32_t)partner[g - own0] : 0u; };   // (other shards' chains: nobody here asks)
#pragma unroll
        for (int r = 0; r < PT4; ++r) {
            const int g4 = tid + r * XWG;
            if (4 * g4 < Ng) {
                const uint32_t g = 4u * (uint32_t)g4;
                put4(g4, src4[r], own_partner(g), own_partner(g + 1u), own_partner(g + 2u), own_partner(g + 3u));
            }
        }
    } else {
        __syncthreads();
        uint32_t* last = slot;   // [Ng] 1 + the chain's partner in its last swapped pair
#pragma unroll
        for (int r = 0; r < PT4; ++r) {
            const int g4 = tid + r * XWG;
            if (4 * g4 < Ng) ((u32x4r_t*)last)[g4] = u32x4r_t{0u, 0u, 0u, 0u};
        }
        __syncthreads();
        // the rows once more (from the L2), in order: the pairs of a level touch disjoint chains, so plain stores do, with a
        // barrier where a level ends — a later pair overwrites an earlier one (LDS atomics: 9.5 us for this pass at 32768 chains)
        // (lane l holds its wave's ballot of row l; a row's ballot becomes the exec mask of the two stores: no test per lane)
        const unsigned long long mrow = lane < nrows ? bits[lane * 16 + wave] : 0ull;
        const uint32_t mlo = (uint32_t)mrow, mhi = (uint32_t)(mrow >> 32);
        auto prow = [&](const uint32_t pw, const int r) {
            const unsigned long long m = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)mlo, r) |
                                         ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)mhi, r) << 32);
            const uint32_t ai = (pw & 0xffffu) << 2, aj = (pw >> 16) << 2;
            unsigned long long saved;
            asm volatile("s_and_saveexec_b64 %0, %1\n\tds_write_b32 %2, %4\n\tds_write_b32 %3, %5\n\ts_mov_b64 exec, %0"   // set_exchanged!, :747-748
                         : "=&s"(saved) : "s"(m), "v"(ai), "v"(aj), "v"((pw >> 16) + 1u), "v"((pw & 0xffffu) + 1u) : "memory", "scc");
            if ((endmask >> r) & 1ull) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
        };
        {   // (eight rows ahead: a row of this pass is too short for three to cover the way from the L2)
            constexpr int D = 8;
            uint32_t pq[D];
#pragma unroll
            for (int d = 0; d < D; ++d) fetch(pq[d], d);
            int r = 0;
#pragma clang loop unroll(disable)
            for (; r + D <= nrows; r += D) {
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    asm volatile("s_waitcnt vmcnt(7)" : "+v"(pq[d]) :: "memory");
                    prow(pq[d], r + d);
                    fetch(pq[d], r + d + D);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(pq[0]), "+v"(pq[1]), "+v"(pq[2]), "+v"(pq[3]), "+v"(pq[4]), "+v"(pq[5]), "+v"(pq[6]), "+v"(pq[7]) :: "memory");
#pragma unroll
            for (int d = 0; d < D; ++d)
                if (r + d < nrows) prow(pq[d], r + d);
        }
        __syncthreads();
        XTS(3);
#pragma unroll
        for (int r = 0; r < PT4; ++r) {
            const int g4 = tid + r * XWG;
            if (4 * g4 < Ng) {
                const u32x4r_t l4 = ((const u32x4r_t*)last)[g4];
                put4(g4, src4[r], l4.x, l4.y, l4.z, l4.w);
            }
        }
    }
    XTS(4);
    if (P.ts && tid == 0) P.ts[(size_t)8 * 60000 + 7] = (unsigned long long)nrows;
}

// ------------------------------------------------------------------------------------------
// The values form of the sharded exchange (for long records): only the chains' VALUES travel to every rank (an
// all-gather of 8 bytes per chain); every rank resolves exchangeMoves! from them, and the record a chain continues from
// travels to that chain's owner alone — an all-to-all of fixed-size blocks of `cap` records per (source, destination) pair
// (no host round trip for counts; a block that overflows raises the sticky device error).  Ranks own equal blocks of N chains.
//   k_a2a_index: one workgroup per peer rank p.  As DESTINATION p: the chains of rank p whose record sits with me, in chain
//       order -> the rows of my send block p.  As SOURCE p: my chains whose record sits with rank p, in chain order -> where
//       each finds its row in the received buffer (block p, the same order: both sides count the same chains).
//       The block to myself goes the same way (the collective copies it), so the apply step reads donors from one place.
//   k_a2a_pack: the rows into the send buffer [G][cap][RW].
//   k_a2a_apply: k_exch_apply with the donor rows taken from the received buffer.
// ------------------------------------------------------------------------------------------
__host__ __device__ inline int a2a_capacity(int N, int G) {
    const int c = 2 * ((N + G - 1) / G) + 64;
    return c < N ? c : N;
}
__device__ inline uint32_t block_excl_count(const bool sel, uint32_t* wsum, const int tid, uint32_t& total) {
    const unsigned long long m = __ballot(sel);
    const int lane = tid & 63, wave = tid >> 6;
    const uint32_t in_wave = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    __syncthreads();
    if (lane == 0) wsum[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t base = 0;
    total = 0;
    for (int w = 0; w < XWG / 64; ++w) { const uint32_t v = wsum[w]; if (w < wave) base += v; total += v; }
    return base + in_wave;
}
__global__ __launch_bounds__(XWG) void k_a2a_index(const KParams P, const int t, const int G, const int cap, int32_t* __restrict__ send_idx,
                                                   int32_t* __restrict__ send_cnt, int32_t* __restrict__ rowidx) {
    __shared__ uint32_t wsum[XWG / 64];
    const int tid = threadIdx.x, p = blockIdx.x;
    const int N = P.N, me = P.offset / P.N;
    // (one read for the whole workgroup: lanes that saw different answers would part ways in front of the barriers below)
    __shared__ int s_stop;
    if (tid == 0) s_stop = *(const volatile unsigned long long*)P.err != ERR_NONE;
    __syncthreads();
    if (s_stop) return;
    for (int part = 0; part < 2; ++part) {
        uint32_t base = 0;
        for (int i0 = 0; i0 < N; i0 += XWG) {
            const int i = i0 + tid;
            const int g = (part == 0 ? p : me) * N + i;          // part 0: a chain of rank p; part 1: one of mine
            unsigned long long xr = 0ull;
            if (i < N) xr = P.xres[g];
            const int s = (int)(unsigned)(xr & 0xffffffffu);
            const bool sel = i < N && (xr >> 32) != 0ull && s / N == (part == 0 ? me : p);
            uint32_t total;
            const uint32_t pos = base + block_excl_count(sel, wsum, tid, total);
            if (sel) {
                if (pos < (uint32_t)cap) {
                    if (part == 0) send_idx[(size_t)p * cap + pos] = s - me * N;
                    else rowidx[i] = p * cap + (int)pos;
                } else {
                    atomicMin(P.err, ((unsigned long long)t << 34) | ((unsigned long long)g << 2));   // kind 0: a block overflowed
                }
            }
            base += total;
        }
        if (part == 0 && tid == 0) send_cnt[p] = (int32_t)(base < (uint32_t)cap ? base : (uint32_t)cap);
    }
}
__global__ void k_a2a_pack(const KParams P, const int G, const int cap, const int32_t* __restrict__ send_idx, const int32_t* __restrict__ send_cnt,
                           const double* __restrict__ rec, double* __restrict__ send) {
    const int RW = P.RW;
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // element of the send buffer [G][cap][RW]
    if (e >= (size_t)G * cap * RW) return;
    const int f = (int)(e % RW);
    const size_t row = e / RW;
    const int b = (int)(row / cap), pos = (int)(row % cap);
    if (pos >= send_cnt[b]) return;
    send[e] = rec[(size_t)send_idx[row] * RW + f];
}
__global__ void k_a2a_apply(const KParams P, const int t, const double* __restrict__ recv, const int32_t* __restrict__ rowidx,
                            double* __restrict__ rec) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= P.N) return;
    if (*(const volatile unsigned long long*)P.err != ERR_NONE) return;   // the run stopped at the failing iteration
    const int N = P.N, RW = P.RW, HW = P.HW;
    const unsigned long long xr = P.xres[P.offset + c];
    const int partner = (int)(xr >> 32);
    if (partner == 0) return;
    const double* __restrict__ donor = recv + (size_t)rowidx[c] * RW;
    double* csb = P.cs + (size_t)c * CSW;
    double* hrec = P.hrec + ((size_t)(t - 1) * N + c) * HW;
    const double value = donor[0];
    double bestv, bestid;
    if (value < csb[CS_BESTP]) { bestv = value; bestid = (double)t; }
    else { bestv = csb[CS_BESTP]; bestid = csb[CS_BESTPID]; }
    csb[CS_BEST] = bestv; csb[CS_BESTID] = bestid; csb[CS_WASX] = 1.0;
    hrec[H_VALUE] = value; hrec[H_PROB] = donor[1]; hrec[H_CURR] = value; hrec[H_BEST] = bestv;
    hrec[H_BESTID] = bestid; hrec[H_EXCH] = (double)partner; hrec[H_ACC] = 1.0; hrec[H_STATUS] = donor[2];
    for (int k = 0; k < P.np + P.nm; ++k) hrec[H_PARAMS + k] = donor[3 + k];
    for (int f = 0; f < RW; ++f) rec[(size_t)c * RW + f] = donor[f];
}
