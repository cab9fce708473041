// the PERSISTENT chain kernel of objectives evaluated by a whole tile — objfunc_norm with any number of parameters (the shocks streamed
// from L2) and the dense simulation on the FP64 matrix cores (BASELINE config 5): k_chain_persist_tile — part of libsmmhip (included by
// smmhip.hip inside its anonymous namespace, behind smm_chain_persist_loc.hpp whose window, re-numbering table and progress words it
// shares; gfx950 device code).
#pragma once
// ------------------------------------------------------------------------------------------
// k_chain_persist_loc (np <= 2: a lane's shocks in registers, a control wave of four lanes per chain) and k_chain_persist_gen (np <= 16,
// objectives one thread evaluates) left the general chain kernel — k_chain_iter<1, 8> / <2, 16>: the reference's own larger examples,
// objfunc_norm with 6 and 18 parameters (Examples.jl:210-230, 232-319), and the dense objective with 50 — on one launch per iteration:
// at 50 parameters that is a kernel boundary, 4.7 us of block loads into 143 KB of LDS and 1.7 us of block stores around 12 us of
// products, every iteration.  Here the loop of smm_chain_persist.hpp runs for a tile of 16 chains served by ALL EIGHT waves of a
// 512-lane workgroup (the reduction contract of the objectives: 512 lanes own a tile, smm_params.hpp), 32 lanes of one wave per chain:
//   * chain state, both record parities, constants and the randomness of the iteration stay in LDS for the whole launch; the partial
//     sums of the objective and the two history rows share one region (the rows are stored before the objective starts and written
//     after its partials have been consumed);
//   * the exchange is the locally numbered cone of smm_chain_persist_loc.hpp in its 16-byte form {value, local | stamp << 16} — one
//     threshold >= 0 (or NaN) for all chains, the reference's default included: after its accept step a tile publishes every chain's
//     whole record as self-validating granules (RW x 16 bytes) into a ring of PR_K iterations, the next iteration's walk gathers the
//     VALUE granule of the ~100 chains of its cone, wave 0 walks the sub-levels, and only an exchanged chain fetches its donor's record;
//   * no role specialisation (every wave is needed by the proposal and the objective): what persist_loc's worker waves do under the
//     simulation — the next lists by LDS-DMA, the re-numbering table, the next randomness — is issued before the objective and waited
//     for behind it; the gather runs behind the publication, where the tile would otherwise wait for its peers' stores.
// One proposal batch or several, isotropic proposals (no Cholesky factor), dist_fun = -, a single shard of at most one tile per
// workgroup slot of the device.  Results are bit-identical to k_chain_iter's.  Errors, ring overrun guard, time-outs,
// repair: as in smm_chain_persist.hpp.
// Reference semantics: AlgoBGP.jl:589-640 (computeNextIteration!), :647-716 (exchangeMoves!), :734-749 (swap_ev_ij!).
// ------------------------------------------------------------------------------------------
constexpr int PT_CT = 16;          // chains per tile
constexpr int PT_LPC = WG / PT_CT; // lanes per chain (32: half a wave)
constexpr int PT_NJ = 5;           // 16-byte pieces of a record per lane of its chain (RW <= 160)

struct PersistTileArgs {
    const uint32_t* cone_hdr; const uint32_t* cone_pairs; const uint16_t* cone_gather; const uint32_t* cone_ok;
    unsigned char* self;               // the ring's window (pr_win_layout)
    uint32_t o_ctl, o_progress, o_rec;
    double* cs; const double* rec_in; double* rec_out; double* vals_out; uint2* slot8_out; uint32_t* walk_flags;
    double* hrec; unsigned long long* err; unsigned long long* ts;
    const double *Z, *lb, *ub, *mom, *w, *objp, *dense_Bf, *dense_Af, *dense_A2f;
    const double* rb;                  // randomness blocks of the window (null: drawn in the kernel)
    int N, Ng, np, nm, ns, zstride, RW, HW, RBW, dense_nOt, batch_size, failbox;
    int plan_t0, exch_from, sigma_update_steps, smpl_iters, t0, t1, rb_t0, rb_tries, user_n;
    int ring_k, slow_tile, slow_ticks, walk_first, unit_sh, scout_after, scout_gl;
    uint32_t epoch;
    double sigma_adjust_by, thr;
    uint64_t seed;
    unsigned long long tmo;            // ticks a spin may last
    const double* mi_g;                // min_improve of every chain of the population (the wide walk's per-position thresholds, AlgoBGP.jl:522, :688)
    int u_lanes, n_udata;              // a user objective in its map-reduce form (SMM_TILE_USER below): lanes per evaluation, doubles of its data (objp)
};

// LDS: [slots 16 B x PL_LOCN | pair words | gather list | 4 headers | re-numbering table | flags, stamps] doubles: cs rec[2] theta
// const sm vk rb | region B: the objective's partial sums / the two history rows
struct PtLayout { uint32_t pbase, gbase, hbase, tbase, fbase, thbase, dbase; uint32_t o_cs, o_rec, o_theta, o_const, o_sm, o_vk, o_uv, o_rb, o_B; size_t total; };
// (kind: 1 objfunc_norm, 2 the dense simulation, 3 its spec v2, 4 a user objective in its map-reduce form: user_part doubles of wave totals)
__host__ __device__ inline PtLayout pt_layout(const int np, const int nm, const int RW, const int HW, const int RBW, const int kind, const int nOt, const int user_part = 0) {
    PtLayout L;
    L.pbase = PL_PBASE;
    L.gbase = L.pbase + (uint32_t)CONE_LEVELS * 64 * 4;
    L.hbase = L.gbase + (uint32_t)CONE_GCAP * 2;
    L.tbase = L.hbase + 4u * 16 * 4;
    L.fbase = L.tbase + (uint32_t)PL_HASH * 4;
    L.thbase = L.fbase + 128u;                       // a threshold (double) per local slot
    L.dbase = L.thbase + (uint32_t)PL_LOCN * 8u;
    uint32_t o = 0;   // doubles behind dbase
    L.o_cs = o; o += PT_CT * PR_STW;
    L.o_rec = o; o += 2 * PT_CT * RW;
    L.o_theta = o; o += (PT_CT * np + 1) & ~1;
    L.o_const = o; o += (2 * np + 2 * nm + 1) & ~1;
    L.o_sm = o; o += (PT_CT * nm + 1) & ~1;
    L.o_vk = o; o += (PT_CT * nm + 1) & ~1;
    L.o_uv = o; o += 2 * PT_CT;   // a user objective's value and status per chain
    L.o_rb = o; o += PT_CT * RBW;
    L.o_B = o;
    uint32_t part = kind >= 2 ? (uint32_t)(WG / 64) * (uint32_t)(nOt * 16) * 16u : (uint32_t)(WG / 64) * PT_CT * (uint32_t)nm;
    if (kind == 3 && part < (uint32_t)DENSE_D * 16u) part = (uint32_t)DENSE_D * 16u;
    if (kind == 4) part = (uint32_t)user_part;   // (spec v2 of the dense objective stages its first hidden layer there)
    const uint32_t rows = 2u * PT_CT * (uint32_t)HW;
    o += part > rows ? part : rows;
    L.total = (size_t)L.dbase + (size_t)o * 8;
    return L;
}

struct PtBarrier { __device__ __forceinline__ void operator()() const { PR_BARRIER(); } };

// out of line: a self-validating granule that is not there yet
__device__ __attribute__((noinline)) uint4 pt_wait_ll(const PrWait W, const uint4* p, const uint32_t tag, const int t, const int g) {
    uint4 q = make_uint4(0u, 0u, 0u, 0u);
    if (*W.s_abort) return q;
    const unsigned long long w0 = wall_clock64();
    unsigned spins = 0;
    do {
        __builtin_amdgcn_s_sleep(1);
        q = pr_load16_sys(p);
        if ((++spins & 63u) == 0u && pr_give_up(W, w0)) { pr_abort(W, t, g); break; }
    } while (!p2p_ll_ok(q, tag));
    return q;
}

// A USER OBJECTIVE IN ITS MAP-REDUCE FORM in this loop (round 6; MProb.objfunc, mprob.jl:159,182 — the form a real simulation objective takes: a sum over
// many independent units): the library compiles THIS FILE once more with hiprtc, together with the user's source (smm_register_user_objective_lanes:
// SMM_USER_PARTIAL / SMM_USER_FINISH, SMM_NSUMS sums), as SMM_TILE_USER — the kernel is then smm_user_persist_tile_kernel, KIND 4: the tile's 512 lanes
// evaluate 512 / lanes chains at a time, `lanes` lanes per chain exactly as the stand-alone smm_user_eval_kernel does (lane l of n_lanes, the halving tree
// inside each group of 64, the groups' totals left to right: the numerical contract of include/smmhip.h — results are bit-identical), the chain's lane
// calls the user's finish, and a failing evaluation (status < 0) is the rejection of mprob.jl:183-186 / AlgoBGP.jl:336-338.
#ifdef SMM_TILE_USER
extern "C" __global__ __launch_bounds__(WG) void smm_user_persist_tile_kernel(const PersistTileArgs A) {
    constexpr int KIND = 4;
    constexpr bool PCT = false;   // (one threshold for all chains)
#else
template <int KIND, bool PCT = false>   // PCT: thresholds by chain (a form of its own: see k_chain_persist_loc)
__global__ __launch_bounds__(WG) void k_chain_persist_tile(const PersistTileArgs A) {
    static_assert(KIND == 1 || KIND == 2, "objfunc_norm (shocks streamed) or the dense simulation");
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int CT = PT_CT, LPC = PT_LPC;
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = (int)blockIdx.x, tiles = (int)gridDim.x;
    const int N = A.N, np = A.np, nm = A.nm, RW = A.RW, HW = A.HW, RBW = A.RBW;
    const PtLayout L = pt_layout(np, nm, RW, HW, RBW, KIND, A.dense_nOt);
    uint4* const slots = (uint4*)lds;
    uint32_t* const s_hdr = (uint32_t*)(lds + L.hbase);
    uint32_t* const s_tab = (uint32_t*)(lds + L.tbase);
    unsigned* const s_flags = (unsigned*)(lds + L.fbase);
    int* const s_minprog = (int*)(s_flags + 0);
    unsigned* const s_abort = s_flags + 1;
    unsigned* const s_rngctr = s_flags + 2;   // chunks of 64 draws of the next iteration's randomness handed out so far (fetch_rb_dyn)
    unsigned long long* const s_ts = (unsigned long long*)(lds + L.fbase + 64);   // [8]
    double* const s_thr = (double*)(lds + L.thbase);   // [PL_LOCN]: min_improve of the chain at that local position
    double* const dbl = (double*)(lds + L.dbase);
    double* const s_cs = dbl + L.o_cs;        // [16][PR_STW]
    double* const s_rec = dbl + L.o_rec;      // [2][16][RW]: parity of t = the record the chain continues from; the other = its last accepted record after t
    double* const s_theta = dbl + L.o_theta;  // [16][np]
    double* const s_lb = dbl + L.o_const;
    double* const s_ub = s_lb + np;
    double* const s_mom = s_ub + np;
    double* const s_w = s_mom + nm;
    double* const s_sm = dbl + L.o_sm;        // [16][nm]: simulated moments of the proposals
    double* const s_vk = dbl + L.o_vk;        // [16][nm]: their squared weighted deviations
    double* const s_uv = dbl + L.o_uv;        // [2][16]: a user objective's value / status of the proposals (KIND 4)
    double* const s_rb = dbl + L.o_rb;        // [16][RBW]: u, z[try][np] of the iteration
    double* const s_part = dbl + L.o_B;       // region B: the objective's partial sums ...
    double* const s_hrow = dbl + L.o_B;       // ... / [16][HW] the history rows of iteration t
    double* const s_xrow = s_hrow + CT * HW;  // ... and the rewritten ones of t - 1 (exchanged chains)
    const uint16_t* const gl = (const uint16_t*)(lds + L.gbase);
    const uint32_t epoch = A.epoch;
    const int t0 = A.t0, t1 = A.t1;
    unsigned char* const mine = A.self;
    uint32_t* const pr_ctl = (uint32_t*)(mine + A.o_ctl);
    uint32_t* const pr_progress = (uint32_t*)(mine + A.o_progress);
    const PrWait W{A.err, pr_ctl, s_abort, A.epoch, A.tmo};
    const int rmask = A.ring_k - 1;
    const uint32_t c0g = (uint32_t)(tile * CT);   // the tile's first chain
    const bool exch_any = A.Ng > 1;
    auto exch_on = [&](const int tx) { return exch_any && tx >= A.exch_from; };   // AlgoBGP.jl:637
    const bool rng_here = A.rb == nullptr;
    const int cc = tid / LPC, r2 = tid % LPC;     // chain of the tile, lane of the chain
    const int c = tile * CT + cc;
    const bool valid = c < N;
    const bool chain_lane = valid && r2 == 0;

    if (error_before(*(const volatile unsigned long long*)A.err, t0)) return;   // an EARLIER launch raised a hard error

    // ---- helpers ----
    auto ring_rec = [&](const int rel) { return (uint4*)(mine + A.o_rec) + (size_t)(rel & rmask) * A.Ng * RW; };
    // the randomness of iteration tn: the window's blocks by LDS-DMA (whoever reads them waits for vmcnt(0) and a barrier), or drawn here
    auto fetch_rb = [&](const int tn) {
        if (rng_here) {
            const int Q = (np + 1) / 2, per = 1 + A.rb_tries * Q;
            for (int it = tid; it < CT * per; it += WG) {
                const int cl = it / per, what = it - cl * per, c1 = tile * CT + cl;
                if (c1 >= N) continue;
                double* o = s_rb + cl * RBW;
                if (what == 0) o[0] = rng_u(A.seed, (uint32_t)c1, (uint32_t)tn);       // probs_acc[iter], AlgoBGP.jl:85
                else {
                    const int rr = (what - 1) / Q, q = (what - 1) - rr * Q;
                    const double2 zz2 = rng_prop_normal2_outofline(A.seed, (uint32_t)c1, (uint32_t)tn, (uint32_t)rr, (uint32_t)q);   // rand(RAND, d), :404
                    o[1 + rr * np + 2 * q] = zz2.x;
                    if (2 * q + 1 < np) o[1 + rr * np + 2 * q + 1] = zz2.y;
                }
            }
        } else {
            const int nchain = min(CT, N - tile * CT);
            const int pieces = nchain * RBW / 2;   // 16-byte pieces of the tile's blocks (contiguous: [t][N][RBW])
            const uint4* src = (const uint4*)(A.rb + ((size_t)(tn - A.rb_t0) * N + (size_t)tile * CT) * RBW);
            const uint32_t dst = (uint32_t)((unsigned char*)s_rb - lds);
            for (int p0 = 0; p0 < pieces; p0 += WG)
                if (p0 + wave * 64 < pieces && p0 + tid < pieces) lds_dma16(src + p0 + tid, dst + (uint32_t)(p0 + wave * 64) * 16u);
        }
    };
    // ... drawn here, in chunks of 64 draws that the waves take from an LDS counter as they become free: the waves with entries of the gather list
    // poll for their peers' publications first, the others start drawing at once (the draws in the shadow of the stores' visibility latency)
    auto fetch_rb_dyn = [&](const int tn) {
        const int Q = (np + 1) / 2, per = 1 + A.rb_tries * Q, total = CT * per, nchunk = (total + 63) / 64;
        for (;;) {
            int k = 0;
            if (lane == 0) k = (int)atomicAdd(s_rngctr, 1u);
            k = __builtin_amdgcn_readfirstlane(k);
            if (k >= nchunk) break;
            const int it = 64 * k + lane;
            if (it >= total) continue;
            const int cl = it / per, what = it - cl * per, c1 = tile * CT + cl;
            if (c1 >= N) continue;
            double* o = s_rb + cl * RBW;
            if (what == 0) o[0] = rng_u(A.seed, (uint32_t)c1, (uint32_t)tn);       // probs_acc[iter], AlgoBGP.jl:85
            else {
                const int rr = (what - 1) / Q, q = (what - 1) - rr * Q;
                const double2 zz2 = rng_prop_normal2_outofline(A.seed, (uint32_t)c1, (uint32_t)tn, (uint32_t)rr, (uint32_t)q);   // rand(RAND, d), :404
                o[1 + rr * np + 2 * q] = zz2.x;
                if (2 * q + 1 < np) o[1 + rr * np + 2 * q + 1] = zz2.y;
            }
        }
    };
    // the lists of exchange tx by LDS-DMA (pairs: a KB per wave; gather list: wave 3)
    auto request_lists = [&](const int tx) {
        const size_t tb = (size_t)(tx - A.plan_t0) * tiles + tile;
        const uint32_t hw1 = s_hdr[(tx & 3) * 16];
        const int nsub1 = (int)(hw1 & 0xffffu), ngat1 = (int)(hw1 >> 16);
        if (4 * wave < nsub1) lds_dma16((const uint4*)(A.cone_pairs + tb * (CONE_LEVELS * 64)) + tid, L.pbase + (uint32_t)wave * 1024u);
        if (wave == 3 && 8 * lane < ngat1) lds_dma16((const uint4*)(A.cone_gather + tb * CONE_GCAP) + lane, L.gbase);
    };
    // the re-numbering table (population chain -> local number) of the gather list that has landed: zeroed by everybody (a barrier
    // before the entries go in), the entries by everybody (a barrier before anybody looks one up)
    auto zero_table = [&]() { for (int x = tid; x < PL_HASH / 2; x += WG) ((uint2*)s_tab)[x] = make_uint2(0u, 0u); };
    auto build_table = [&](const int tx) {
        const int ngat = (int)(s_hdr[(tx & 3) * 16] >> 16);
        for (int e = tid; e < ngat; e += WG) pl_hash_put(s_tab, (uint32_t)gl[e], (uint32_t)(CT + e));
    };
    // every wave, once the pair list has landed and the table stands: local slot offsets, the tail of every sub-level padded with
    // dummy pairs that never swap
    auto fix_lists = [&](const int tx, const int t_report) {
        const uint32_t* hd = s_hdr + (tx & 3) * 16;
        const int nsub1 = (int)(hd[0] & 0xffffu);
        const uint32_t nloc = (uint32_t)CT + (hd[0] >> 16);
        const uint32_t dummy = (16u * nloc) * 0x10001u;
        uint32_t* pw = (uint32_t*)(lds + L.pbase);
        bool bad = false;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int s = 4 * wave + q;
            if (s < nsub1) {
                const uint32_t cnt = (hd[1 + (s >> 2)] >> (8 * (s & 3))) & 0xffu;
                uint32_t w = dummy;
                if ((uint32_t)lane < cnt) {
                    const uint32_t pw0 = pw[s * 64 + lane];
                    const uint32_t ci = (pw0 & 0xffffu) >> A.unit_sh, cj = (pw0 >> 16) >> A.unit_sh;
                    uint32_t li = ci - c0g < (uint32_t)CT ? ci - c0g : pl_hash_get(s_tab, ci);
                    uint32_t lj = cj - c0g < (uint32_t)CT ? cj - c0g : pl_hash_get(s_tab, cj);
                    if (li == 0xffffu || lj == 0xffffu) { bad = true; li = nloc; lj = nloc; }
                    w = (16u * li) | ((16u * lj) << 16);
                }
                pw[s * 64 + lane] = w;
            }
        }
        if (__ballot(bad) != 0ull && lane == 0) pr_report(A.err, 3, t_report, (int)c0g);
    };
    // the cone's initial slots out of ring entry `rel`: the value granule of every chain of the gather list, past the caches
    auto gather = [&](const int tx, const int rel, const int t_report) {
        const int ngat = (int)(s_hdr[(tx & 3) * 16] >> 16);
        const uint4* rr = ring_rec(rel);
        const uint32_t tag = pr_tag32(epoch, rel);
        for (int e = tid; e < ngat; e += WG) {
            const int g = (int)gl[e];
            uint4 q = pr_load16_sys(rr + (size_t)g * RW);
            if (__builtin_expect(!p2p_ll_ok(q, tag), 0)) q = pt_wait_ll(W, rr + (size_t)g * RW, tag, t_report, g);
            slots[CT + e] = make_uint4(q.x, q.z, (uint32_t)(CT + e), 0u);
            if constexpr (PCT) s_thr[CT + e] = A.mi_g[g];
        }
        if (tid == WG - 1) { slots[CT + ngat] = make_uint4(0u, 0u, 0u, 0u); if constexpr (PCT) s_thr[CT + ngat] = 0.0; }   // the dummy pair's slot: 0 - 0 > 0 is false
    };
    // a chain's record as iteration `rel` of the launch into the ring: self-validating granules, 32 lanes per chain
    // (lane r2 stores the granules r2, r2 + 32, ...: every store instruction writes 512 contiguous bytes per chain)
    auto publish = [&](const int rel, const double* rec) {
        uint4* g_ll = ring_rec(rel) + (size_t)c * RW;
        const uint32_t tag = pr_tag32(epoch, rel);
        for (int f = r2; f < RW; f += LPC) {
            const unsigned long long a = __builtin_bit_cast(unsigned long long, rec[f]);
            const p2p_u32x4 q = {(unsigned)a, tag, (unsigned)(a >> 32), tag};
            asm volatile("global_store_dwordx4 %0, %1, off " PR_SC "\n\ts_nop 1" :: "v"(g_ll + f), "v"(q) : "memory");
        }
    };

    // ---- once per launch: the tile's chain state and records, the constants, the first randomness and lists ----
    if (valid) {
        const double2* g_cs = (const double2*)(A.cs + (size_t)c * CSW);
        const double2* g_rec = (const double2*)(A.rec_in + (size_t)c * RW);
        for (int i = r2; i < PR_STW / 2; i += LPC) ((double2*)(s_cs + cc * PR_STW))[i] = g_cs[i];
        for (int i = r2; i < RW / 2; i += LPC) ((double2*)(s_rec + ((t0 & 1) * CT + cc) * RW))[i] = g_rec[i];
        if (r2 == 0) {
            const double v0 = A.rec_in[(size_t)c * RW];
            slots[cc] = make_uint4((uint32_t)__double2loint(v0), (uint32_t)__double2hiint(v0), (uint32_t)cc, 0u);
            if constexpr (PCT) s_thr[cc] = A.mi_g[c];
        }
    }
    for (int k = tid; k < np; k += WG) { s_lb[k] = A.lb[k]; s_ub[k] = A.ub[k]; }
    for (int k = tid; k < nm; k += WG) { s_mom[k] = A.mom[k]; s_w[k] = A.w[k]; }
    if (tid == 0) { *s_minprog = 0; *s_abort = 0u; *s_rngctr = 0u; }
    if (tid >= 64 && tid < 72) s_ts[tid - 64] = 0ull;
    if (wave == 3 && lane < CONE_HDRW) {
        if (A.walk_first) s_hdr[((t0 - 1) & 3) * 16 + lane] = A.cone_hdr[((size_t)(t0 - 1 - A.plan_t0) * tiles + tile) * CONE_HDRW + lane];
        if (t0 < t1 && exch_on(t0)) s_hdr[(t0 & 3) * 16 + lane] = A.cone_hdr[((size_t)(t0 - A.plan_t0) * tiles + tile) * CONE_HDRW + lane];
    }
    fetch_rb(t0);
    PR_BARRIER();
    if (A.walk_first) {
        // the previous kernel — of any form — left the exchange of its last iteration to this one: its lists, and the records the launch
        // starts from published as the launch's iteration 0 (the first walk gathers like any other)
        request_lists(t0 - 1);
        if (valid) publish(0, s_rec + ((t0 & 1) * CT + cc) * RW);
        if (wave == 2 && lane == 0 && A.cone_ok[t0 - 1 - A.plan_t0] == 0u) pr_report(A.err, 3, t0, (int)c0g);
        zero_table();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PR_BARRIER();
        build_table(t0 - 1);
        PR_BARRIER();
        fix_lists(t0 - 1, t0);
        gather(t0 - 1, 0, t0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the first randomness has landed
    if (!valid && r2 == 0) for (int k = 0; k < np; ++k) s_theta[cc * np + k] = 0.0;

    ZBuf zb;
    if constexpr (KIND == 1) zb.init_v(A.Z, nm, A.zstride, tid);

    for (int t = t0; t <= t1; ++t) {
        const int rel = t - t0 + 1;
        const bool exch = t == t0 ? A.walk_first != 0 : exch_on(t - 1);
        const bool lists = t < t1 && exch_on(t);
        double* const rin = s_rec + ((t & 1) * CT + cc) * RW;          // the record the chain continues from
        double* const rout = s_rec + (((t + 1) & 1) * CT + cc) * RW;   // its last accepted record after this iteration
        PR_BARRIER();   // B0: the cone's slots are gathered, the lists fixed, the randomness in LDS
        unsigned long long ts0 = 0;
        if (A.ts && tid == 0) { ts0 = wall_clock64(); if (t != t0) s_ts[0] += ts0 - s_ts[7]; }
        // ---- the walk over the cone's sub-levels, on local slots: wave 0 alone, no barriers ----
        if (exch && wave == 0) {
            const int nsub = (int)(s_hdr[((t - 1) & 3) * 16] & 0xffffu);
            lean_walk_levels<64, 0, true, WalkNoGuard, PCT>(nullptr, 1, L.pbase, (uint32_t)(64 * lane), nsub, lane, 0, A.thr, WalkNoGuard(), L.thbase);
        }
        PR_BARRIER();   // B1
        unsigned long long ts1 = 0;
        if (A.ts && tid == 0) ts1 = wall_clock64();
        uint32_t src = (uint32_t)cc;   // local number of the chain whose record this chain continues from
        int partner = 0;               // 1 + the partner's number in the population
        if (exch && valid) {
            const uint32_t kmeta = slots[cc].z;
            src = kmeta & 0xffffu;
            if (kmeta >> 16) {   // set_exchanged!, :747-748
                const uint32_t pl = lean_partner<0, 4>(lds, L.pbase, kmeta, (uint32_t)cc) - 1u;
                partner = 1 + (int)(pl < (uint32_t)CT ? c0g + pl : (uint32_t)gl[pl - CT]);
            }
        }
        // ---- the donor's whole record (swap_ev_ij!, :734-749) out of the ring, by the chain's 32 lanes ----
        const bool donor = valid && src != (uint32_t)cc;
        if (donor) {
            const uint32_t src_g = src < (uint32_t)CT ? c0g + src : (uint32_t)gl[src - CT];
            const uint4* g_ll = ring_rec(rel - 1) + (size_t)src_g * RW;
            const uint32_t tag = pr_tag32(epoch, rel - 1);
            if (RW <= LPC) {
                if (r2 < RW) {
                    uint4 q = pr_load16_sys(g_ll + r2);
                    if (__builtin_expect(!p2p_ll_ok(q, tag), 0)) q = pt_wait_ll(W, g_ll + r2, tag, t, c);
                    rin[r2] = p2p_ll_double(q);
                }
            } else {
                const uint4* p[PT_NJ];
#pragma unroll
                for (int j = 0; j < PT_NJ; ++j) p[j] = g_ll + min(r2 + LPC * j, RW - 1);
                p2p_u32x4 q0, q1, q2, q3, q4;
                asm volatile("global_load_dwordx4 %0, %5, off sc0 sc1\n\tglobal_load_dwordx4 %1, %6, off sc0 sc1\n\tglobal_load_dwordx4 %2, %7, off sc0 sc1\n\t"
                             "global_load_dwordx4 %3, %8, off sc0 sc1\n\tglobal_load_dwordx4 %4, %9, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                             : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&v"(q4) : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]) : "memory");
                const uint4 qq[PT_NJ] = {make_uint4(q0.x, q0.y, q0.z, q0.w), make_uint4(q1.x, q1.y, q1.z, q1.w), make_uint4(q2.x, q2.y, q2.z, q2.w),
                                         make_uint4(q3.x, q3.y, q3.z, q3.w), make_uint4(q4.x, q4.y, q4.z, q4.w)};
#pragma unroll
                for (int j = 0; j < PT_NJ; ++j) {
                    const int f = r2 + LPC * j;
                    if (f < RW) {
                        uint4 q = qq[j];
                        if (__builtin_expect(!p2p_ll_ok(q, tag), 0)) q = pt_wait_ll(W, g_ll + f, tag, t, c);   // (the gather validated the value only)
                        rin[f] = p2p_ll_double(q);
                    }
                }
            }
        }
        if (chain_lane) s_cs[cc * PR_STW + CS_PARTNER] = (double)partner;
        PR_BARRIER();   // B2: every read of the ring's last entry, of the walk's lists and slots is done
        if (tid == 0) {
            __hip_atomic_store(pr_progress + tile, pr_progress_word(epoch, rel), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *s_rngctr = 0u;   // (the last iteration's draws ended before B0; the next ones start behind B5)
        }
        // the next exchange's lists and header; the progress of the slowest tile, the abort word (consumed behind the objective)
        uint32_t nhdr = 0u;
        const bool want_hdr = wave == 3 && lane < CONE_HDRW && t + 1 < t1 && exch_on(t + 1);
        if (want_hdr) nhdr = A.cone_hdr[((size_t)(t + 1 - A.plan_t0) * tiles + tile) * CONE_HDRW + lane];
        if (lists) { request_lists(t); zero_table(); }
        uint32_t pw_[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
        uint32_t ctl_w = 0u, ok_w = 1u;
        if (wave == 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) if (lane + 64 * j < tiles) pw_[j] = pr_load4_sys(pr_progress + lane + 64 * j);
            if (lane == 0) { ctl_w = pr_load4_sys(pr_ctl); if (lists) ok_w = A.cone_ok[t - A.plan_t0]; }
        }
        // ---- settle iteration t - 1 (set_acceptRate!, :253-257; swap_ev_ij!'s set_eval!, :231-243) ----
        const double uu = chain_lane ? s_rb[cc * RBW] : 0.0;   // probs_acc[iter], :85 (kept: the block is the next iteration's from B3 on)
        if (chain_lane) {
            double* csb = s_cs + cc * PR_STW;
            int nn = (int)csb[CS_NNOEX], na = (int)csb[CS_NACC];
            double bp = csb[CS_BEST], bpid = csb[CS_BESTID];
            if (partner != 0) {
                const double dv = rin[0];
                if (dv < csb[CS_BESTP]) { bp = dv; bpid = (double)(t - 1); }
                else { bp = csb[CS_BESTP]; bpid = csb[CS_BESTPID]; }
                double* hx = s_xrow + cc * HW;
                hx[H_VALUE] = dv; hx[H_PROB] = rin[1]; hx[H_CURR] = dv; hx[H_BEST] = bp; hx[H_BESTID] = bpid;
                hx[H_EXCH] = (double)partner; hx[H_ACC] = 1.0; hx[H_STATUS] = rin[2];
                if (HW > H_PARAMS + np + nm) hx[HW - 1] = 0.0;
            } else { nn += 1; na += (int)csb[CS_LACC]; }
            csb[CS_NNOEX] = (double)nn; csb[CS_NACC] = (double)na; csb[CS_BEST] = bp; csb[CS_BESTID] = bpid;
        }
        if (valid && partner != 0) {
            copy_strided(s_xrow + cc * HW + H_PARAMS, rin + 3, np + nm, r2, LPC);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            coop_store_n(A.hrec + ((size_t)(t - 2) * N + c) * HW, s_xrow + cc * HW, HW, r2, LPC);
        }
        double za[ZU];
        if constexpr (KIND == 1) sim_load_chunk_v(zb, A.zstride, false, 0, 0, za);
        unsigned long long ts2 = 0;
        if (A.ts && tid == 0) ts2 = wall_clock64();
        // ---- proposal (mysample, AlgoBGP.jl:400-410; proposal :424-471): every wave, 32 lanes per chain (smm_propose.hpp) ----
        // (made ahead, behind the last publication, from the chains' own records — and again for the exchanged ones — it was no faster:
        // an iteration lasts as long as its slowest tile's work, and that work does not shrink by being done earlier; EXPERIMENTS.md R5.6)
        {
            const CoopProp X{s_rec + (t & 1) * CT * RW, RW, s_rec + ((t + 1) & 1) * CT * RW, RW, s_theta, np, s_hrow, HW, s_rb, RBW, s_cs, PR_STW, s_lb, s_ub,
                             (unsigned long long*)s_hrow + 2, A.err, A.seed, 0, N, A.batch_size, A.rb_tries, A.user_n, A.smpl_iters, A.scout_after, A.scout_gl};
            coop_mysample<CT>(X, t, tile, tid, WG / 64, tid == 0, PtBarrier());
        }
        PR_BARRIER();   // B3: the proposals stand; the randomness block and the rows' region are free
        unsigned long long ts3 = 0;
        if (A.ts && tid == 0) ts3 = wall_clock64();
        if (t < t1 && !rng_here) fetch_rb(t + 1);
        // ---- the objective: all 512 lanes ----
#ifdef SMM_TILE_USER
        {   // evaluateObjective(m, p) (mprob.jl:175-188) -> the user's partial sums: WG / u_lanes chains at a time, u_lanes lanes (whole waves) per chain
            const int UL = A.u_lanes, per = WG / UL, nwv = UL / 64;
            for (int c0r = 0; c0r < CT; c0r += per) {
                const int cr = c0r + tid / UL, ll = tid % UL;
                if (tile * CT + cr < N) {   // (wave-uniform: a chain's lanes are whole waves)
                    double part[SMM_NSUMS];
#pragma unroll
                    for (int i = 0; i < SMM_NSUMS; ++i) part[i] = 0.0;
                    smm_user_partial(s_theta + cr * np, np, A.objp, A.n_udata, ll, UL, part);
#pragma unroll
                    for (int i = 0; i < SMM_NSUMS; ++i) {
                        double a = part[i];
#pragma unroll
                        for (int off = 32; off >= 1; off >>= 1) a = a + __shfl_xor(a, off, 64);
                        if (lane == 0) s_part[(cr * nwv + (ll >> 6)) * SMM_NSUMS + i] = a;
                    }
                }
            }
        }
#else
        if constexpr (KIND == 1) simulate_tile_v<CT>(A.ns, nm, np, A.zstride, false, zb, s_theta, s_part, tid, za);
        else dense_tile_v<CT>(np, A.dense_nOt, A.dense_Bf, A.dense_Af, A.dense_A2f, (uint32_t)((unsigned char*)s_theta - lds), (uint32_t)((unsigned char*)s_part - lds), tid);
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA (lists, randomness) has landed
        if (want_hdr) s_hdr[((t + 1) & 3) * 16 + lane] = nhdr;
        if (wave == 2) {
            uint32_t m = 0xfffu;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (lane + 64 * j < tiles) {
                    const int d = (int)(((pw_[j] >> 12) - epoch) << 12) >> 12;   // (20-bit epochs, wrap-safe: pl_min_progress)
                    m = min(m, d == 0 ? (pw_[j] & 0xfffu) : (d > 0 ? 0xfffu : 0u));
                }
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, off, 64));
            if (lane == 0) {
                __hip_atomic_store(s_minprog, (int)m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (ctl_w == epoch) *s_abort = 1u;
                if (ok_w == 0u) pr_report(A.err, 3, t + 1, (int)c0g);
            }
        }
        PR_BARRIER();   // B4: the partial sums are in LDS; the lists have landed
        unsigned long long ts4 = 0;
        if (A.ts && tid == 0) ts4 = wall_clock64();
        // ---- the moments of a chain (wave totals -> mean -> squared weighted deviation) by its 32 lanes (ObjExamples.jl:79-100) ----
        const bool failed = KIND == 1 && A.failbox && valid && s_theta[cc * np] >= A.objp[0] && s_theta[cc * np] <= A.objp[1];   // mprob.jl:183-186
#ifdef SMM_TILE_USER
        if (chain_lane) {   // the groups' totals left to right (numerical contract), then the user's finish: moments, value, status
            const int nwv = A.u_lanes / 64;
            double tot[SMM_NSUMS];
#pragma unroll
            for (int i = 0; i < SMM_NSUMS; ++i) {
                double a = s_part[(cc * nwv) * SMM_NSUMS + i];
                for (int wv = 1; wv < nwv; ++wv) a = a + s_part[(cc * nwv + wv) * SMM_NSUMS + i];
                tot[i] = a;
            }
            int st_u = 1;
            double v_u = 0.0;
            smm_user_finish(s_theta + cc * np, np, tot, SMM_NSUMS, s_mom, s_w, nm, A.objp, A.n_udata, s_sm + cc * nm, &v_u, &st_u);
            s_uv[cc] = v_u; s_uv[CT + cc] = (double)st_u;
        }
#else
        if constexpr (KIND == 2) {
            // (the dense tile's partial sums lie [wave][moment][chain]: the lanes run over the CHAINS of a moment — consecutive doubles, no
            // bank conflict; the chain's own 32 lanes would stride 128 bytes and meet in one bank)
            const int nmp = A.dense_nOt * 16;
            const int c16 = tid & (CT - 1);
            if (tile * CT + c16 < N) {
                for (int k = tid / CT; k < nm; k += WG / CT) {
                    double tot = s_part[((size_t)0 * nmp + k) * 16 + c16];
#pragma unroll
                    for (int wv = 1; wv < WG / 64; ++wv) tot = tot + s_part[((size_t)wv * nmp + k) * 16 + c16];
                    double d = tot - s_mom[k];
                    const double wk = s_w[k];
                    if (!isnan(wk)) d = d / wk;
                    s_sm[c16 * nm + k] = tot;
                    s_vk[c16 * nm + k] = d * d;
                }
            }
        } else if (valid) {
            for (int k = r2; k < nm; k += LPC) {
                double tot = s_part[(0 * CT + cc) * nm + k];
#pragma unroll
                for (int wv = 1; wv < WG / 64; ++wv) tot = tot + s_part[(wv * CT + cc) * nm + k];
                const double m = tot / (double)A.ns;
                double d = m - s_mom[k];
                const double wk = s_w[k];
                if (!isnan(wk)) d = d / wk;
                s_sm[cc * nm + k] = failed ? NAN : m;
                s_vk[cc * nm + k] = d * d;
            }
        }
#endif
        if (lists) build_table(t);
        PR_BARRIER();   // B5: the partial sums are consumed (the rows' region is free), the re-numbering table stands
        unsigned long long ts4b = 0;
        if (A.ts && tid == 0) ts4b = wall_clock64();
        // ---- objective value, doAcceptReject! (:324-392), set_eval! (:220-245): the chain's lane ----
        if (chain_lane) {
            double* csb = s_cs + cc * PR_STW;
            const double* vk = s_vk + cc * nm;
            double value;
            int status;
#ifdef SMM_TILE_USER
            if (true) { value = s_uv[cc]; status = (int)s_uv[CT + cc]; (void)vk; (void)failed; }   // the user's own (a failing evaluation: status < 0, :336-338 below)
#else
            if (failed) { value = -1.0; status = -2; }   // Eval() default, Eval.jl:84
#endif
            else {
                double vsum = 0.0;
                int k = 0;
                for (; k + 8 <= nm; k += 8) {
                    double v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = vk[k + u];
#pragma unroll
                    for (int u = 0; u < 8; ++u) vsum = (k + u == 0) ? v[u] : vsum + v[u];
                }
                for (; k < nm; ++k) vsum = (k == 0) ? vk[k] : vsum + vk[k];
                value = vsum / (double)nm;
                status = 1;
            }
            const double sig = csb[CS_SIGMA], bp = csb[CS_BEST], bpid = csb[CS_BESTID], atun = csb[CS_ATUN];
            const int nn = (int)csb[CS_NNOEX], na = (int)csb[CS_NACC];
            const double old = rin[0];
            double prob;
            bool acc;
            if (status < 0) {   // :336-338
                prob = 0.0; acc = false;
            } else {
                if (!(value >= 0.0)) pr_report(A.err, ERRK_NEGATIVE, t, c);   // :341
                const double e = pr_exp(atun * (old - value));
                prob = (e != e) ? e : (e < 1.0 ? e : 1.0);   // minimum([1.0,e]), NaN propagates (:344)
                if (!isfinite(prob)) { prob = 0.0; acc = false; status = -1; }   // :350-353
                else if (!isfinite(old)) { prob = 1.0; acc = true; }             // :355-359
                else { status = 1; acc = prob > uu; }                            // strict >, :362-367
            }
            const double rate = (double)(na + (acc ? 1 : 0)) / (double)(nn + 1);   // set_acceptRate!, :253-257
            double nsig = sig;
            if ((t % A.sigma_update_steps) == 0) nsig = (rate > 0.234) ? sig * (1.0 + A.sigma_adjust_by) : sig * (1.0 - A.sigma_adjust_by);   // :381-390
            const double currv = acc ? value : old;
            double bestv, bestid;
            if (value < bp) { bestv = value; bestid = (double)t; }
            else { bestv = bp; bestid = bpid; }
            csb[CS_SIGMA] = nsig; csb[CS_RATE] = rate; csb[CS_LACC] = acc ? 1.0 : 0.0; csb[CS_WASX] = 0.0; csb[CS_BEST] = bestv; csb[CS_BESTID] = bestid;
            csb[CS_BESTP] = bp; csb[CS_BESTPID] = bpid;   // best after t - 1: needed if iteration t gets exchanged
            double* hr = s_hrow + cc * HW;
            hr[H_VALUE] = value; hr[H_PROB] = prob; hr[H_CURR] = currv; hr[H_BEST] = bestv; hr[H_BESTID] = bestid;
            hr[H_EXCH] = 0.0; hr[H_ACC] = acc ? 1.0 : 0.0; hr[H_STATUS] = (double)status;
            if (HW > H_PARAMS + np + nm) hr[HW - 1] = 0.0;
            // the chain's last accepted record (lastAccepted :209-215) = input of the exchange step: its head here, the arrays below
            if (acc) { rout[0] = value; rout[1] = prob; rout[2] = (double)status; }
            else { rout[0] = rin[0]; rout[1] = rin[1]; rout[2] = rin[2]; }
            if (RW > 3 + np + nm) rout[RW - 1] = 0.0;
            const double vnew = acc ? value : old;
            slots[cc] = make_uint4((uint32_t)__double2loint(vnew), (uint32_t)__double2hiint(vnew), (uint32_t)cc, 0u);   // the tile's own slots of the next walk
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();   // (a chain's 32 lanes sit in one wave)
        // ---- the rest of the record (parameters and moments: the proposal's if accepted, else the old ones), and its publication: the
        // self-validating record of iteration t into the ring (write-through stores), granule by granule as it is put together ----
        if (t < t1) {
            if (__builtin_expect(rel > rmask && __hip_atomic_load(s_minprog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < rel - rmask, 0))
                pl_wait_progress(W, pr_progress, s_minprog, rel - rmask, tiles, lane, t, (int)c0g);
#ifdef SMM_TEST_HOOKS
            if (tile == A.slow_tile) { const unsigned long long w0 = wall_clock64(); while (wall_clock64() - w0 < (unsigned long long)A.slow_ticks) __builtin_amdgcn_s_sleep(8); }
#endif
        }
        if (valid) {
            const bool acc = s_hrow[cc * HW + H_ACC] != 0.0;
            uint4* g_ll = ring_rec(rel) + (size_t)c * RW;
            const uint32_t tag = pr_tag32(epoch, rel);
            for (int f = r2; f < RW; f += LPC) {
                double v;
                if (f < 3) v = rout[f];
                else {
                    const int k = f - 3;
                    v = k >= np + nm ? 0.0 : (!acc ? rin[f] : (k < np ? s_theta[cc * np + k] : s_sm[cc * nm + k - np]));
                    rout[f] = v;
                }
                if (t < t1) {
                    const unsigned long long a = __builtin_bit_cast(unsigned long long, v);
                    const p2p_u32x4 q = {(unsigned)a, tag, (unsigned)(a >> 32), tag};
                    asm volatile("global_store_dwordx4 %0, %1, off " PR_SC "\n\ts_nop 1" :: "v"(g_ll + f), "v"(q) : "memory");
                }
            }
        }
        unsigned long long ts5 = 0;
        if (A.ts && tid == 0) ts5 = wall_clock64();
        // ================= behind the publication =================
        // (what nobody waits for, while the other tiles' publications travel: the next randomness, the next walk's pair words on local
        // numbers, the history row's arrays; then the gather for the NEXT iteration's walk, then the row's stores)
        if (lists) fix_lists(t, t + 1);
        if (valid) {
            copy_strided(s_hrow + cc * HW + H_PARAMS, s_theta + cc * np, np, r2, LPC);
            copy_strided(s_hrow + cc * HW + H_PARAMS + np, s_sm + cc * nm, nm, r2, LPC);
        }
        if (lists) gather(t, rel, t + 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (valid) coop_store_n(A.hrec + ((size_t)(t - 1) * N + c) * HW, s_hrow + cc * HW, HW, r2, LPC);
        if (t < t1 && rng_here) fetch_rb_dyn(t + 1);
        if (A.ts && tid == 0) {   // (slot 0: from the publication to the next iteration's B0 — rows, gather, the wait for the tile's other waves)
            s_ts[1] += ts1 - ts0; s_ts[2] += ts2 - ts1; s_ts[3] += ts3 - ts2; s_ts[4] += ts4 - ts3; s_ts[5] += ts4b - ts4; s_ts[6] += ts5 - ts4b; s_ts[7] = ts5;
        }
    }
    PR_BARRIER();   // the last epilogue is done
    // the result blocks where the next launch (of any form) expects them
    if (valid) {
        const double2* cs2 = (const double2*)(s_cs + cc * PR_STW);
        const double* rl = s_rec + (((t1 + 1) & 1) * CT + cc) * RW;
        double2* g_cs = (double2*)(A.cs + (size_t)c * CSW);
        for (int i = r2; i < PR_STW / 2; i += LPC) g_cs[i] = cs2[i];
        for (int i = r2; i < RW / 2; i += LPC) ((double2*)(A.rec_out + (size_t)c * RW))[i] = ((const double2*)rl)[i];
        if (r2 == 0) {
            const double v = rl[0];
            A.vals_out[c] = v;
            if (A.slot8_out) { A.slot8_out[c] = make_uint2(order_key32(v), (uint32_t)c); if (v != v) atomicOr(A.walk_flags, 1u); }
        }
    }
    if (A.ts && tid < 8) A.ts[(size_t)tile * 8 + tid] = tid < 7 ? s_ts[tid] : (unsigned long long)(t1 - t0 + 1);
}
