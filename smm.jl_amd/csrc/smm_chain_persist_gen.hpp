// the PERSISTENT chain kernel of simulation-free objectives (BASELINE config 4: banana, 10 parameters, 8192 chains): k_chain_persist_gen —
// part of libsmmhip (included by smmhip.hip inside its anonymous namespace; gfx950 device code).
#pragma once
// ------------------------------------------------------------------------------------------
// The loop of smm_chain_persist.hpp (one launch per look-ahead window, workgroups coupled by a ring of tagged words, every workgroup
// waits only for its cone of the exchange) for objectives WITHOUT a simulation: an iteration of such a problem is pure latency —
// publication -> visibility -> walk -> the donor's record -> proposal -> objective -> accept — and the one-launch-per-iteration
// kernel (k_chain_iter<0, 16, 2, true>) put a kernel boundary, 100 KB of staging and a block load / store round on top of it:
// 20 us per iteration at 8192 chains for ~80 flop per chain.
//   * one workgroup = 32 chains (the cone unit of the per-iteration key walk, P.cone_ct: 256 workgroups at 8192 chains, one per CU),
//     1024 lanes by role: waves 0 and 1 are the CONTROL waves (16 chains each, four lanes per chain, lane r of a quad takes the
//     component pairs r, r + 4, ...), wave 0 walks the workgroup's cone; wave 2 makes the next iteration's randomness (the MH uniform
//     and the normals of the first tries: in the kernel — no k_pregen_rng window — unless tables are injected) and reads the tiles'
//     progress; wave 3 brings the gather list and stores the history rows out of LDS; waves 4..11 gather (a lane per chain of the
//     cone: ~200-400); waves 8..15 bring the next pair list by LDS-DMA and pad it first;
//   * the ring holds the walk slot {order_key32(value), chain | tag << 16} and the self-validating record (value, prob, status,
//     parameters, moments: RW doubles, a uint4 each); the gather takes the slots only — with 10 parameters the donors' parameters are
//     fetched behind the walk, by the exchanged chains alone;
//   * the chains' state blocks and records live in LDS for the whole launch (the records in two parities: what the chain continues
//     from / its last accepted record after the accept step); settling the previous iteration, counters, sigma, best values and the
//     history rows run behind the publication.
// np, nm <= 16 in one proposal batch, isotropic proposals, banana, min_improve == 0,
// dist_fun = -, 4096 < N <= 8192 in whole workgroups of 32 (the key walk's population: its 8-byte slots are addressed in halved
// units).  Results are bit-identical to k_chain_iter<0, 16, 2, true>'s.  Errors, ring overrun guard, time-outs, repair: as in
// smm_chain_persist.hpp.
// ------------------------------------------------------------------------------------------
constexpr int PG_CT = 32;          // chains per workgroup
constexpr int PG_MAXP = 16;        // parameters (= moments)
constexpr int PG_TRIES = 2;        // proposal tries whose normals are made ahead (the per-iteration path's rb_tries for np > 8)

struct PersistGenArgs {
    const uint32_t* cone_hdr; const uint32_t* cone_pairs; const uint16_t* cone_gather; const uint32_t* cone_ok;
    uint2* pr_slot; uint4* pr_rec; uint32_t* pr_progress; uint32_t* pr_ctl;
    double* cs; const double* rec_in; double* rec_out; double* vals_out; uint2* slot8_out; uint32_t* walk_flags;
    double* hrec; unsigned long long* err; unsigned long long* ts;
    const double *lb, *ub, *mom, *w;
    const double* rb;                 // randomness blocks of injected tables (null: drawn in the kernel)
    int N, Ng, np, nm, RW, HW, plan_t0, exch_from, sigma_update_steps, smpl_iters, t0, t1;
    int rb_t0, RBW, rb_tries, user_n;
    int ring_k, slow_tile, slow_ticks, walk_first;
    unsigned long long tmo;           // ticks a spin may last
    uint32_t epoch;
    double sigma_adjust_by;
    uint64_t seed;
    const double* udata; int n_udata; // a user objective's own data (SMM_GEN_USER: the kernel compiled with the user's source, below)
};

__host__ __device__ inline int persist_gen_rngw(int np) { return (1 + PG_TRIES * np + 1) & ~1; }
__host__ __device__ inline size_t persist_gen_smem_bytes(int Ng, int np, int RW, int HW) {
    const size_t slots = (size_t)(((Ng + 3) & ~3) + 4) * 8;
    const size_t lists = 2 * (size_t)CONE_LEVELS * 64 * 4 + 2 * (size_t)CONE_GCAP * 2 + 4 * 16 * 4;
    const size_t dbl = (size_t)PG_CT * PG_MAXP + (size_t)PG_CT * PR_STW + 2 * (size_t)PG_CT * RW + 2 * (size_t)PG_CT * persist_gen_rngw(np) + 2 * (size_t)PG_CT * HW +
                       5 * PG_MAXP + 16;
    const size_t land = 2 * (size_t)((RW + 3) / 4) * 64 * 16;   // the donors' records as they arrive (LDS-DMA): [control wave][piece][lane] x 16 B
    return slots + lists + dbl * 8 + land;
}

// A USER OBJECTIVE in this loop (round 5; mprob.jl:159,182 — "bring your own objective"): the library compiles THIS FILE once more with
// hiprtc, together with the user's source (smm_register_user_objective: SMM_USER_OBJECTIVE(theta, np, mom, w, nm, udata, n_udata,
// sim_moments, value, status)), as SMM_GEN_USER — the kernel is then smm_user_persist_kernel, lane 0 of every chain's quad calls the
// user's function on the proposal in LDS (one thread per evaluation, as the stand-alone smm_user_eval_kernel does), the chain's simulated
// moments are the function's, and a failing evaluation (status < 0) is the rejection of mprob.jl:183-186 / AlgoBGP.jl:336-338.
__host__ __device__ inline size_t persist_gen_user_bytes() { return (size_t)PG_CT * PG_MAXP * 8 + (size_t)PG_CT * 8 + (size_t)PG_CT * 8; }
#ifdef SMM_GEN_USER
extern "C" __global__ __launch_bounds__(1024, 4) void smm_user_persist_kernel(const PersistGenArgs A) {
#else
__global__ __launch_bounds__(1024, 4) void k_chain_persist_gen(const PersistGenArgs A) {
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = (int)blockIdx.x, tiles = (int)gridDim.x;
    const int N = A.N, Ng4 = (A.Ng + 3) & ~3, np = A.np, nm = A.nm, RW = A.RW, HW = A.HW;
    const int RNGW = persist_gen_rngw(np);
    const int NPC = (RW + 3) / 4;                     // 16-byte pieces of a record per lane of a chain's quad
    const int NPH = HW / 2;
    const int US = 8 * (Ng4 + 2) <= 65536 ? 0 : 1;   // pair words hold the slots' byte offsets, halved where 16 bits would not reach (smm_walk_lean.hpp)
    // ---- LDS ----
    const uint32_t pbase = 8u * (uint32_t)(Ng4 + 4);
    const uint32_t gbase = pbase + 2u * CONE_LEVELS * 64 * 4;
    const uint32_t hbase = gbase + 2u * CONE_GCAP * 2;
    uint2* slots = (uint2*)lds;
    uint32_t* s_hdr = (uint32_t*)(lds + hbase);
    double* s_theta = (double*)(lds + hbase + 4 * 16 * 4);     // [32][PG_MAXP]: the proposals
    double* s_cs = s_theta + PG_CT * PG_MAXP;                   // [32][PR_STW]: state block fields 0..11
    double* s_rec = s_cs + PG_CT * PR_STW;                      // [2][32][RW]: the record the chain continues from (parity of t) / its last accepted record after t (the other)
    double* s_rng = s_rec + 2 * PG_CT * RW;                     // [2][32][RNGW]: u, normals [try][np], by iteration parity
    double* s_hrow = s_rng + 2 * PG_CT * RNGW;                  // [32][HW]
    double* s_xrow = s_hrow + PG_CT * HW;                       // [32][HW]
    double* s_const = s_xrow + PG_CT * HW;                      // lb[16] ub[16] mom[16] w[16] mom + 2.2 [16]
    unsigned long long* s_ts = (unsigned long long*)(s_const + 5 * PG_MAXP);   // [8]
    unsigned* s_flags = (unsigned*)(s_ts + 8);
    int* s_minprog = (int*)(s_flags + 0);
    unsigned* s_abort = s_flags + 1;
    unsigned* s_xmask = s_flags + 2;
    int* s_glready = (int*)(s_flags + 3);
    unsigned* s_pub = s_flags + 4;                              // publications of this workgroup's control waves so far (two per iteration)
    unsigned* s_read = s_flags + 5;                             // ... and their completed reads of the ring's last entry
    uint4* s_land = (uint4*)(s_ts + 16);                        // [2][NPC][64]
#ifdef SMM_GEN_USER
    double* s_usm = (double*)(lds + persist_gen_smem_bytes(A.Ng, np, RW, HW));   // [32][PG_MAXP]: the user's simulated moments of this iteration's proposals
    double* s_uval = s_usm + PG_CT * PG_MAXP;                   // [32]: ... values
    double* s_ust = s_uval + PG_CT;                             // [32]: ... status (as a double)
#endif
    const uint32_t epoch = A.epoch;
    const int t0 = A.t0, t1 = A.t1;
    const PrWait W{A.err, A.pr_ctl, s_abort, A.epoch, A.tmo};
    const int rmask = A.ring_k - 1;
    const bool exch_any = A.Ng > 1;
    auto exch_on = [&](const int tx) { return exch_any && tx >= A.exch_from; };   // AlgoBGP.jl:637
    const bool rng_here = A.rb == nullptr;

    if (error_before(*(const volatile unsigned long long*)A.err, t0)) return;

    // ---- once per launch: the workgroup's chain state and records, the constants, the lists of the pending exchange ----
    if (tid < 128) {
        const int cl = tid >> 2, r = tid & 3, c = tile * PG_CT + cl;
        if (c < N) {
            const double2* g_cs = (const double2*)(A.cs + (size_t)c * CSW);
            const double2* g_rec = (const double2*)(A.rec_in + (size_t)c * RW);
            for (int i = r; i < 6; i += 4) ((double2*)(s_cs + cl * PR_STW))[i] = g_cs[i];
            for (int i = r; i < RW / 2; i += 4) ((double2*)(s_rec + ((t0 & 1) * PG_CT + cl) * RW))[i] = g_rec[i];
            if (r == 0) slots[c] = make_uint2(order_key32(g_rec[0].x), (uint32_t)c);   // the workgroup's own slots of the first walk
        }
    }
    if (tid >= 128 && tid < 128 + PG_MAXP) {
        const int k = tid - 128;
        s_const[k] = k < np ? A.lb[k] : 0.0; s_const[PG_MAXP + k] = k < np ? A.ub[k] : 1.0;
        s_const[2 * PG_MAXP + k] = k < nm ? A.mom[k] : 0.0; s_const[3 * PG_MAXP + k] = k < nm ? A.w[k] : 1.0;
        s_const[4 * PG_MAXP + k] = k < nm ? A.mom[k] + 2.2 : 0.0;   // the "simulated" moments of the built-in objectives without a simulation
    }
    if (tid == 192) {
        slots[Ng4] = make_uint2(1u, 0u); slots[Ng4 + 1] = make_uint2(2u, 0u);
        *s_minprog = 0; *s_abort = 0u; *s_xmask = 0u; *s_glready = t0 - 1; *s_pub = 0u; *s_read = 0u;
    }
    if (tid >= 256 && tid < 264) s_ts[tid - 256] = 0ull;
    if (wave == 3 && lane < CONE_HDRW) {
        if (A.walk_first) s_hdr[((t0 - 1) & 3) * 16 + lane] = A.cone_hdr[((size_t)(t0 - 1 - A.plan_t0) * tiles + tile) * CONE_HDRW + lane];
        if (t0 < t1 && exch_on(t0)) s_hdr[(t0 & 3) * 16 + lane] = A.cone_hdr[((size_t)(t0 - A.plan_t0) * tiles + tile) * CONE_HDRW + lane];
    }
    PR_BARRIER();
    auto request_lists = [&](const int tx) {
        const size_t tb = (size_t)(tx - A.plan_t0) * tiles + tile;
        const uint32_t hw1 = s_hdr[(tx & 3) * 16];
        const int nsub1 = (int)(hw1 & 0xffffu), ngat1 = (int)(hw1 >> 16);
        const uint32_t b = (uint32_t)(tx & 1);
        if (wave >= 8 && 4 * (wave - 8) < nsub1)
            lds_dma16((const uint4*)(A.cone_pairs + tb * (CONE_LEVELS * 64)) + (tid - 512), pbase + b * (CONE_LEVELS * 64 * 4) + (uint32_t)(wave - 8) * 1024u);
        if (wave == 3 && 8 * lane < ngat1)
            lds_dma16((const uint4*)(A.cone_gather + tb * CONE_GCAP) + lane, gbase + b * (CONE_GCAP * 2));
    };
    auto pad_lists = [&](const int tx) {
        const uint32_t* hd = s_hdr + (tx & 3) * 16;
        const int nsub1 = (int)(hd[0] & 0xffffu);
        const uint32_t unit = 8u >> US;
        const uint32_t dummy = (unit * (uint32_t)Ng4) | ((unit * (uint32_t)(Ng4 + 1)) << 16);
        uint32_t* pw = (uint32_t*)(lds + pbase) + (tx & 1) * (CONE_LEVELS * 64);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int s = 4 * (wave - 8) + q;
            if (s < nsub1) {
                const uint32_t cnt = (hd[1 + (s >> 2)] >> (8 * (s & 3))) & 0xffu;
                if ((uint32_t)lane >= cnt) pw[s * 64 + lane] = dummy;
            }
        }
    };
    if (A.walk_first) {
        if (wave == 3 || wave >= 8) {
            request_lists(t0 - 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (wave >= 8) pad_lists(t0 - 1);
        }
        if (wave == 2 && lane == 0 && A.cone_ok[t0 - 1 - A.plan_t0] == 0u) pr_report(A.err, 3, t0, tile * PG_CT);
        PR_BARRIER();
        if (wave >= 4 && wave < 8) {
            const int ngat = (int)(s_hdr[((t0 - 1) & 3) * 16] >> 16);
            const uint16_t* gl = (const uint16_t*)(lds + gbase) + ((t0 - 1) & 1) * CONE_GCAP;
            for (int e = tid - 256; e < ngat; e += 256) {
                const int g = (int)gl[e];
                slots[g] = make_uint2(order_key32(A.rec_in[(size_t)g * RW]), (uint32_t)g);
            }
        }
    }
    // the randomness of iteration tn into its parity's rows (wave 2): rows [chain][u, z[try][np]] exactly as k_pregen_rng makes them
    auto make_rng = [&](const int tn) {
        double* rows = s_rng + (size_t)(tn & 1) * PG_CT * RNGW;
        if (rng_here) {
            const int Q = (np + 1) / 2, per = 1 + PG_TRIES * Q;
            for (int it = lane; it < PG_CT * per; it += 64) {
                const int cl = it / per, what = it - cl * per, c1 = tile * PG_CT + cl;
                if (c1 >= N) continue;
                double* o = rows + cl * RNGW;
                if (what == 0) o[0] = rng_u(A.seed, (uint32_t)c1, (uint32_t)tn);       // probs_acc[iter], AlgoBGP.jl:85
                else {
                    const int rr = (what - 1) / Q, q = (what - 1) - rr * Q;
                    double z0, z1;
                    rng_prop_normal2(A.seed, (uint32_t)c1, (uint32_t)tn, (uint32_t)rr, (uint32_t)q, z0, z1);   // rand(RAND, d), :404
                    o[1 + rr * np + 2 * q] = z0;
                    if (2 * q + 1 < np) o[1 + rr * np + 2 * q + 1] = z1;
                }
            }
        } else {
            const int nd = 1 + min(A.rb_tries, PG_TRIES) * np;
            for (int it = lane; it < PG_CT * nd; it += 64) {
                const int cl = it / nd, i = it - cl * nd, c1 = tile * PG_CT + cl;
                if (c1 < N) rows[cl * RNGW + i] = A.rb[((size_t)(tn - A.rb_t0) * N + c1) * A.RBW + i];
            }
        }
    };
    auto store_rows = [&](const double* s_row, const int trow, const unsigned mask) {   // history rows out of LDS (wave 3)
        for (int e = lane; e < PG_CT * NPH; e += 64) {
            const int cl = e / NPH, i = e - cl * NPH, c = tile * PG_CT + cl;
            if (c < N && ((mask >> cl) & 1u)) ((double2*)(A.hrec + ((size_t)(trow - 1) * N + c) * HW))[i] = ((const double2*)(s_row + cl * HW))[i];
        }
    };

    if (wave >= 2) {
        // =====================================================================================================================
        // the WORKER waves
        // =====================================================================================================================
        if (wave == 2) make_rng(t0);
        for (int t = t0; t <= t1; ++t) {
            const int rel = t - t0 + 1;
            PR_BARRIER();   // BA: the cone's slots are staged, this iteration's lists and randomness are in LDS
            uint32_t nhdr = 0u;
            const bool want_hdr = wave == 3 && lane < CONE_HDRW && t + 1 < t1 && exch_on(t + 1);
            if (wave == 3) {
                if (t > t0) { store_rows(s_hrow, t - 1, 0xffffffffu); store_rows(s_xrow, t - 2, *s_xmask); }
                if (want_hdr) nhdr = A.cone_hdr[((size_t)(t + 1 - A.plan_t0) * tiles + tile) * CONE_HDRW + lane];
            }
            if (t < t1 && exch_on(t)) request_lists(t);
            PR_BARRIER();   // BW: the walk is done (control wave 1 reads its chains' slots)
            if (wave == 3 || wave >= 8) {
                if (wave == 3 && want_hdr) s_hdr[((t + 1) & 3) * 16 + lane] = nhdr;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (wave >= 8 && t < t1 && exch_on(t)) pad_lists(t);
                if (wave == 3 && lane == 0) __hip_atomic_store(s_glready, t, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (wave == 2) {
                if (t < t1) make_rng(t + 1);
                const int m = pr_min_progress(A.pr_progress, epoch, tiles, lane);
                if (lane == 0) {
                    __hip_atomic_store(s_minprog, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (pr_load4_sys(A.pr_ctl) == epoch) *s_abort = 1u;
                    if (t < t1 && exch_on(t) && A.cone_ok[t - A.plan_t0] == 0u) pr_report(A.err, 3, t + 1, tile * PG_CT);
                }
            }
            if (wave >= 4 && wave < 12 && t < t1 && exch_on(t)) {
                // ---- gather for the NEXT iteration's walk, once both control waves have published (512 lanes: one look per chain of the cone) ----
                while (__hip_atomic_load(s_glready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != t) __builtin_amdgcn_s_sleep(1);
                while (__hip_atomic_load(s_pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 2u * (unsigned)rel) __builtin_amdgcn_s_sleep(1);
                __builtin_amdgcn_s_sleep(PR_GATHER_DELAY);
                const int ngat = (int)(s_hdr[(t & 3) * 16] >> 16);
                const unsigned long long* rs = (const unsigned long long*)A.pr_slot + (size_t)(rel & rmask) * (A.Ng + 4);
                const uint32_t want = pr_tag16(epoch, rel) << 16;
                const uint16_t* gl = (const uint16_t*)(lds + gbase) + (t & 1) * CONE_GCAP;
                for (int e = tid - 256; e < ngat; e += 512) {
                    const int g = (int)gl[e];
                    unsigned long long v = pr_load8_sys(rs + g);
                    if (__builtin_expect((((uint32_t)(v >> 32)) & 0xffff0000u) != want, 0)) v = pr_wait_slot(W, rs + g, want, t + 1, g);
                    slots[g] = make_uint2((uint32_t)v, (uint32_t)(v >> 32) & 0xffffu);
                }
            }
        }
        PR_BARRIER();   // the last epilogue is done
        if (wave == 3) { store_rows(s_hrow, t1, 0xffffffffu); store_rows(s_xrow, t1 - 1, *s_xmask); }
        if (wave == 4 || wave == 5) {   // the result blocks where the next launch (of any form) expects them: out of the chains' lines
            const int l2 = tid - 256, cl = l2 >> 2, r = l2 & 3, c = tile * PG_CT + cl;
            if (c < N) {
                const double2* cs2 = (const double2*)(s_cs + cl * PR_STW);
                const double* rl = s_rec + (((t1 + 1) & 1) * PG_CT + cl) * RW;
                double2* g_cs = (double2*)(A.cs + (size_t)c * CSW);
                for (int i = r; i < 6; i += 4) g_cs[i] = cs2[i];
                for (int i = r; i < RW / 2; i += 4) ((double2*)(A.rec_out + (size_t)c * RW))[i] = ((const double2*)rl)[i];
                if (r == 0) {
                    const double v = rl[0];
                    A.vals_out[c] = v;
                    if (A.slot8_out) { A.slot8_out[c] = make_uint2(order_key32(v), (uint32_t)c); if (v != v) atomicOr(A.walk_flags, 1u); }
                }
            }
        }
        return;
    }

    // =========================================================================================================================
    // the CONTROL waves 0 and 1: 16 chains each, four lanes per chain; lane r takes the component pairs r, r + 4, ...
    // =========================================================================================================================
    for (int t = t0; t <= t1; ++t) {
        const int rel = t - t0 + 1;
        const bool first = t == t0;
        const bool exch = first ? A.walk_first != 0 : exch_on(t - 1);
        PR_BARRIER();   // BA
        unsigned long long ts1 = 0;
        if (A.ts && tid == 0) { ts1 = wall_clock64(); if (!first) s_ts[0] += ts1 - s_ts[7]; }
        const uint32_t lbase = pbase + (uint32_t)((t - 1) & 1) * (CONE_LEVELS * 64 * 4);
        if (exch && wave == 0) {
            // ---- the walk over the cone's sub-levels: wave 0 alone, no barriers ----
            const int nsub = (int)(s_hdr[((t - 1) & 3) * 16] & 0xffffu);
            const PersistWalkValues values{W, (const uint4*)A.pr_rec + (size_t)((rel - 1) & rmask) * A.Ng * RW, first ? A.rec_in : nullptr, pr_tag32(epoch, rel - 1), RW, 0, t};
            if (US == 0) lean_walk_levels<64, 0, false, PersistWalkValues>(nullptr, 1, lbase, (uint32_t)(64 * lane), nsub, lane, 0, 0.0, values);
            else lean_walk_levels<64, 1, false, PersistWalkValues>(nullptr, 1, lbase, (uint32_t)(64 * lane), nsub, lane, 0, 0.0, values);
        }
        unsigned long long ts2 = 0;
        if (A.ts && tid == 0) ts2 = wall_clock64();
        PR_BARRIER();   // BW
        const int l2 = tid;                                   // 0..127
        const int cl = l2 >> 2, r = l2 & 3;
        const int c = tile * PG_CT + cl;
        const bool valid = c < N;
        double* cs = s_cs + cl * PR_STW;
        double* rin = s_rec + ((t & 1) * PG_CT + cl) * RW;          // the record the chain continues from
        double* rout = s_rec + (((t + 1) & 1) * PG_CT + cl) * RW;   // its last accepted record after this iteration
        uint32_t src = (uint32_t)c;
        int partner = 0;
        if (exch && valid) {
            const uint32_t kmeta = slots[c].y;
            src = kmeta & 0xffffu;
            if (kmeta >> 16) partner = US == 0 ? (int)lean_partner<0>(lds, lbase, kmeta, (uint32_t)c) : (int)lean_partner<1>(lds, lbase, kmeta, (uint32_t)c);   // set_exchanged!, :747-748
        }
        // ---- the record the chain continues from: its own (already there) or its donor's (swap_ev_ij!, :734-749) ----
        // (out of the ring by LDS-DMA: lane r of the quad brings the record's 16-byte pieces r, r + 4, ... — all requested at once)
        if (__builtin_expect(__ballot(valid && src != (uint32_t)c) != 0ull, 1)) {
            const bool donor = valid && src != (uint32_t)c;
            if (__builtin_expect(first, 0)) {
                if (donor) for (int i = r; i < RW; i += 4) rin[i] = A.rec_in[(size_t)src * RW + i];
            } else {
                const uint4* g_ll = (const uint4*)A.pr_rec + ((size_t)((rel - 1) & rmask) * A.Ng + src) * RW;
                const uint32_t tag = pr_tag32(epoch, rel - 1);
                const uint32_t land = (uint32_t)((unsigned char*)s_land - lds) + (uint32_t)wave * (uint32_t)NPC * 1024u;
                if (donor) for (int j = 0; j < NPC; ++j) if (r + 4 * j < RW) pr_dma16(g_ll + r + 4 * j, land + (uint32_t)j * 1024u);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (donor) {
                    const uint4* sl = s_land + (size_t)wave * NPC * 64 + lane;
                    for (int j = 0; j < NPC; ++j) {
                        const int i = r + 4 * j;
                        if (i < RW) {
                            uint4 q = sl[j * 64];
                            if (__builtin_expect(!p2p_ll_ok(q, tag), 0)) { const PrLL2 w2 = pr_wait_ll2(W, g_ll + i, g_ll + i, tag, t, c); q = w2.q0; }
                            rin[i] = p2p_ll_double(q);
                        }
                    }
                }
            }
        }
        // (the quad's lanes read each other's pieces below: LDS operations of one wave complete in order)
        asm volatile("" ::: "memory");
        // every read of the ring's last entry is done — by both control waves: say so (the publication of iteration rel + K - 1 waits for it)
        if (lane == 0 && __hip_atomic_fetch_add(s_read, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + 1u == 2u * (unsigned)rel)
            __hip_atomic_store(A.pr_progress + tile, pr_progress_word(epoch, rel), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long ts3 = 0;
        if (A.ts && tid == 0) ts3 = wall_clock64();
        // ---- proposal (mysample, AlgoBGP.jl:400-410; proposal :424-471): the tries in order, each tested by the chain's four lanes ----
        const double sigma = cs[CS_SIGMA];
        const double* rows = s_rng + (size_t)(t & 1) * PG_CT * RNGW + cl * RNGW;
        const int Q = (np + 1) / 2;
        double th[4], m01[4];   // the lane's components: pairs r and r + 4 (np <= 16)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int k = 2 * (r + 4 * j) + e;
                const bool has = k < np;
                const double lbk = s_const[has ? k : 0], ubk = s_const[PG_MAXP + (has ? k : 0)];
                const double old = has ? rin[3 + k] : 0.0;
                th[2 * j + e] = old;
                m01[2 * j + e] = (old - lbk) / (ubk - lbk);   // mapto_01, mprob.jl:248
            }
        }
        {
            const int max_tries = A.user_n ? min(A.rb_tries, A.smpl_iters) : A.smpl_iters;
            const int n_pre = rng_here ? PG_TRIES : min(A.rb_tries, PG_TRIES);
            bool done = !valid;
            for (int rr = 0; rr < max_tries && __ballot(!done) != 0ull; ++rr) {
                double out[4];
                bool okl = true;
                if (!done) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int q = r + 4 * j;
                        if (q < Q) {
                            double z0, z1;
                            if (rr < n_pre) { z0 = rows[1 + rr * np + 2 * q]; z1 = 2 * q + 1 < np ? rows[1 + rr * np + 2 * q + 1] : 0.0; }
                            else if (!rng_here && rr < A.rb_tries) {
                                const double* g_rb = A.rb + ((size_t)(t - A.rb_t0) * N + c) * A.RBW;
                                z0 = g_rb[1 + rr * np + 2 * q]; z1 = 2 * q + 1 < np ? g_rb[1 + rr * np + 2 * q + 1] : 0.0;
                            } else { const double2 zz2 = rng_prop_normal2_outofline(A.seed, (uint32_t)c, (uint32_t)t, (uint32_t)rr, (uint32_t)q); z0 = zz2.x; z1 = zz2.y; }
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                const int k = 2 * q + e;
                                if (k < np) {
                                    const double lbk = s_const[k];
                                    const double span = s_const[PG_MAXP + k] - lbk;
                                    const double step = sigma * (e ? z1 : z0);   // MvNormal(mu01, sigma): x = mu + sigma*z
                                    const double x = m01[2 * j + e] + step;
                                    if (!(x >= 0.0 && x <= 1.0)) okl = false;   // inclusive bounds, :405
                                    const double sc = x * span;
                                    out[2 * j + e] = sc + lbk;   // mapto_ab, mprob.jl:271
                                }
                            }
                        }
                    }
                }
                const unsigned long long m = __ballot(done || okl);
                const unsigned quad = (unsigned)(m >> (lane & ~3)) & 0xfu;
                if (!done && quad == 0xfu) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) th[i] = out[i];
                    done = true;
                }
            }
            if (!done && r == 0) pr_report(A.err, ERRK_NO_DRAW, t, c);   // :409
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int k = 2 * (r + 4 * j) + e;
                if (k < np) s_theta[cl * PG_MAXP + k] = valid ? th[2 * j + e] : 0.0;
            }
        asm volatile("" ::: "memory");
        unsigned long long ts4 = 0, ts5 = 0;
        if (A.ts && tid == 0) ts4 = wall_clock64();
        if (valid) {
            // ---- objective (banana, ObjExamples.jl:251-265, generalised to np dimensions; the terms in order), doAcceptReject! (:324-392) ----
            // (every lane of the quad computes the same value: its lanes read the proposal out of LDS, whose writes completed in order)
            const double* thp = s_theta + cl * PG_MAXP;
            double value = 0.0;
            int status = 1;
#ifdef SMM_GEN_USER
            // evaluateObjective(m, p) (mprob.jl:175-188) -> the user's function, by lane 0 of the chain's quad; the quad's other lanes read
            // its results out of LDS (same wave: the writes complete in order)
            if (r == 0) {
                int st_u = 1;
                double v_u = 0.0;
                smm_user_objective(thp, np, s_const + 2 * PG_MAXP, s_const + 3 * PG_MAXP, nm, A.udata, A.n_udata, s_usm + cl * PG_MAXP, &v_u, &st_u);
                s_uval[cl] = v_u; s_ust[cl] = (double)st_u;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            value = s_uval[cl];
            status = (int)s_ust[cl];
            const bool failed = status < 0;          // the objective's "exception": status -2, rejected with prob 0 (mprob.jl:183-186, AlgoBGP.jl:336-338)
#else
            for (int i = 0; i + 1 < np; ++i) {
                const double a = thp[i], b = thp[i + 1];
                const double t1_ = b - a * a;
                const double t2_ = 1.0 - a;
                const double term = 100.0 * (t1_ * t1_) + t2_ * t2_;
                value = (i == 0) ? term : value + term;
            }
            const bool failed = false;
#endif
            const double atun = cs[CS_ATUN];
            const double uu = rows[0];
            const double old = rin[0];
            double prob;
            bool acc;
            if (failed) { prob = 0.0; acc = false; }                     // :336-338
            else {
                if (!(value >= 0.0) && r == 0) pr_report(A.err, ERRK_NEGATIVE, t, c);   // :341
                const double e = pr_exp(atun * (old - value));
                prob = (e != e) ? e : (e < 1.0 ? e : 1.0);   // minimum([1.0,e]), NaN propagates (:344)
                if (!isfinite(prob)) { prob = 0.0; acc = false; status = -1; }   // :350-353
                else if (!isfinite(old)) { prob = 1.0; acc = true; }             // :355-359
                else { status = 1; acc = prob > uu; }                            // strict >, :362-367
            }
            const double accd = acc ? 1.0 : 0.0;
            const double v = acc ? value : old;
            // ---- the chain's last accepted record (lastAccepted :209-215) = input of the exchange step: by the quad's lanes into LDS ----
            {
#ifdef SMM_GEN_USER
                const double* msim = s_usm + cl * PG_MAXP;
#else
                const double* msim = s_const + 4 * PG_MAXP;
#endif
                for (int k = r; k < np; k += 4) rout[3 + k] = acc ? thp[k] : rin[3 + k];
                for (int k = r; k < nm; k += 4) rout[3 + np + k] = acc ? msim[k] : rin[3 + np + k];
                if (r == 0) {
                    rout[0] = v; rout[1] = acc ? prob : rin[1]; rout[2] = acc ? (double)status : rin[2];
                    if (RW > 3 + np + nm) rout[RW - 1] = 0.0;
                }
            }
            asm volatile("" ::: "memory");
            // ---- publish: the walk slot and the self-validating record of iteration t (write-through stores) ----
            if (t < t1) {
                if (__builtin_expect(rel > rmask && __hip_atomic_load(s_minprog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < rel - rmask, 0))
                    pr_wait_progress(W, A.pr_progress, s_minprog, rel - rmask, tiles, lane, t, c);
#ifdef SMM_TEST_HOOKS
                if (tile == A.slow_tile) { const unsigned long long w0 = wall_clock64(); while (wall_clock64() - w0 < (unsigned long long)A.slow_ticks) __builtin_amdgcn_s_sleep(8); }
#endif
                if (r == 0)
                    pr_store8((uint2*)A.pr_slot + (size_t)(rel & rmask) * (A.Ng + 4) + c,
                              (unsigned long long)order_key32(v) | ((unsigned long long)((uint32_t)c | (pr_tag16(epoch, rel) << 16)) << 32));
                unsigned char* g_ll = (unsigned char*)A.pr_rec + ((size_t)(rel & rmask) * A.Ng + c) * RW * 16;
                const uint32_t tag = pr_tag32(epoch, rel);
                for (int i = 2 * r; i < RW; i += 8)   // lane r: the pairs of doubles r, r + 4, ... (a 32-byte granule each)
                    pr_store_ll(g_ll + (size_t)i * 16, *(const double2*)(rout + i), tag);
            }
            // the gather's clock: this wave has published (the other tiles publish at about the same time; the gather touches other
            // tiles' slots only, so it may run under the bookkeeping below)
            if (lane == 0) __hip_atomic_fetch_add(s_pub, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (N is whole workgroups: lane 0 serves a chain)
            // ================= behind the publication =================
            asm volatile("" ::: "memory");
            if (A.ts && tid == 0) ts5 = wall_clock64();
            const double sig = sigma;
            int nn = (int)cs[CS_NNOEX], na = (int)cs[CS_NACC];
            double bp = cs[CS_BEST], bpid = cs[CS_BESTID];
            if (partner != 0) {   // set_eval!(ci, ej) of swap_ev_ij! as a history record (:231-243)
                const double dv = old;
                if (dv < cs[CS_BESTP]) { bp = dv; bpid = (double)(t - 1); }
                else { bp = cs[CS_BESTP]; bpid = cs[CS_BESTPID]; }
                double* hx = s_xrow + cl * HW;
                if (r == 0) {
                    hx[H_VALUE] = dv; hx[H_PROB] = rin[1]; hx[H_CURR] = dv; hx[H_BEST] = bp; hx[H_BESTID] = bpid;
                    hx[H_EXCH] = (double)partner; hx[H_ACC] = 1.0; hx[H_STATUS] = rin[2];
                    if (HW > H_PARAMS + np + nm) hx[HW - 1] = 0.0;
                }
                for (int k = r; k < np + nm; k += 4) hx[H_PARAMS + k] = rin[3 + k];
            } else { nn += 1; na += (int)cs[CS_LACC]; }   // set_acceptRate!, :253-257
            double nsig = sig;
            const bool upd = (t % A.sigma_update_steps) == 0;
            double rate = 0.0;
            if (upd || t == t1) {
                rate = (double)(na + (acc ? 1 : 0)) / (double)(nn + 1);
                if (upd) nsig = (rate > 0.234) ? sig * (1.0 + A.sigma_adjust_by) : sig * (1.0 - A.sigma_adjust_by);   // :381-390
            }
            double bestv, bestid;
            const double currv = acc ? value : old;
            if (value < bp) { bestv = value; bestid = (double)t; }
            else { bestv = bp; bestid = bpid; }
            // the history row (through LDS: wave 3 stores it)
            double* hv = s_hrow + cl * HW;
            if (r == 0) {
                hv[H_VALUE] = value; hv[H_PROB] = prob; hv[H_CURR] = currv; hv[H_BEST] = bestv; hv[H_BESTID] = bestid;
                hv[H_EXCH] = 0.0; hv[H_ACC] = accd; hv[H_STATUS] = (double)status;
                if (HW > H_PARAMS + np + nm) hv[HW - 1] = 0.0;
            }
            for (int k = r; k < np; k += 4) hv[H_PARAMS + k] = thp[k];
#ifdef SMM_GEN_USER
            for (int k = r; k < nm; k += 4) hv[H_PARAMS + np + k] = s_usm[cl * PG_MAXP + k];
#else
            for (int k = r; k < nm; k += 4) hv[H_PARAMS + np + k] = s_const[4 * PG_MAXP + k];
#endif
            asm volatile("" ::: "memory");
            if (r == 0) {
                slots[c] = make_uint2(order_key32(v), (uint32_t)c);
                if (upd || t == t1) cs[CS_RATE] = rate;
                cs[CS_SIGMA] = nsig; cs[CS_NNOEX] = (double)nn; cs[CS_NACC] = (double)na; cs[CS_LACC] = accd; cs[CS_WASX] = 0.0;
                cs[CS_BEST] = bestv; cs[CS_BESTID] = bestid; cs[CS_BESTP] = bp; cs[CS_BESTPID] = bpid; cs[CS_PARTNER] = (double)partner;
            }
        }
        {
            const unsigned long long xm = __ballot(valid && r == 0 && partner != 0);
            unsigned m = 0u;
#pragma unroll
            for (int q = 0; q < 16; ++q) m |= (unsigned)((xm >> (4 * q)) & 1ull) << q;
            // (two control waves: each owns half of the mask; the workers read it behind the next barrier)
            if (lane == 0) {
                __hip_atomic_fetch_and(s_xmask, wave == 0 ? 0xffff0000u : 0x0000ffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_or(s_xmask, wave == 0 ? m : m << 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the records, rows and slots are written before the wave goes to the barrier)
        }
        if (A.ts && tid == 0) { const unsigned long long ts7 = wall_clock64(); s_ts[1] += ts2 - ts1; s_ts[2] += ts3 - ts2; s_ts[3] += ts4 - ts3; s_ts[4] += ts5 - ts4; s_ts[5] += ts7 - ts5; s_ts[7] = ts7; }
    }
    PR_BARRIER();
    if (A.ts && tid < 7) A.ts[(size_t)tile * 8 + tid] = tid < 6 ? s_ts[tid] : (unsigned long long)(t1 - t0 + 1);
}
