// constants, kernel parameter block, block layouts, error word — part of libsmmhip (included by smmhip.hip inside its anonymous namespace; gfx950 device code).
#pragma once

constexpr int WG = SMM_REDUCE_LANES;  // 512 lanes own one chain tile (numerical contract)
constexpr int MAX_DIM = 64;           // np, nm <= 64
constexpr int XWG = 1024;             // exchange workgroup
constexpr int XLDS_MAX = 8192;        // largest N_global resolved in LDS (16 B per chain)
constexpr unsigned XSPIN_LIMIT = 1u << 22;
constexpr int P2P_MAXG = 8;           // ranks of the p2p sharded form (one node: 8 GPUs)

// chain state block
constexpr int CSW = 16;
enum : int { CS_SIGMA = 0, CS_RATE, CS_NNOEX, CS_NACC, CS_LACC, CS_WASX, CS_BEST, CS_BESTID, CS_BESTP, CS_BESTPID, CS_ATUN,
             CS_PARTNER /* LDS only */ };
// history record
enum : int { H_VALUE = 0, H_PROB, H_CURR, H_BEST, H_BESTID, H_EXCH, H_ACC, H_STATUS, H_PARAMS };

// error word: min over (iter<<34 | chain<<2 | kind); 1 negative objective, 2 no draw, 3 internal
constexpr unsigned long long ERR_NONE = ~0ull;

enum : int { F_CLOSE_PREV = 1, F_HAS_PENDING = 2, F_WALK_INLINE = 4, F_GLOBAL_REC = 8, F_PROPOSE_ONLY = 16, F_P2P_ARRIVE = 32 };  // F_P2P_ARRIVE: this launch counts the previous launch's pushes in (smm_p2p.hpp)  // F_GLOBAL_REC: rec_in is the all-gathered buffer (global chain ids)

struct KParams {
    // problem
    int np, nm, ns, obj;
    const double *init, *lb, *ub, *mom, *w, *objp;
    const double* Z;  // [nm][zstride]: the shock matrix, every moment padded to whole chunks of ZU rows x 512 lanes
    int zstride;
    // opts
    int N, Ng, offset, T;
    int sigma_update_steps, smpl_iters, batch_size;
    double sigma_adjust_by;
    uint64_t seed;
    const double* min_improve_g;  // [Ng]
    const double* chol_L;         // general Gaussian proposals: lower-triangular factor [np][np] or [Ng][np][np] (chol_per_chain), or null
    int chol_per_chain;
    // user objective (objective_id >= SMM_OBJ_USER_BASE): proposals out, results in, [N][np] / [N][nm] / [N]
    double* u_theta; double* u_simM; double* u_value; int* u_status;
    int mi_uniform;               // all thresholds equal (the usual case): mi_value
    int mi_pct;                   // thresholds DIFFER by chain, every one >= 0 or NaN, dist_fun = `-`: the persistent forms' wide walk reads one per slot position
    int tile_off;                 // doubles in front of the tile's LDS blocks (the inline walk's chain slots)
    double mi_value;
    // dense objective (SMM_OBJ_DENSE): B and A in MFMA fragment order
    const double* dense_Bf;  // [D/16][ceil(np/4)][64]
    const double* dense_Af;  // [nOt][D/16][4][64]
    const double* dense_A2f; // SMM_OBJ_DENSE2 (the 256 x 256 stage; null: spec v1): [8 waves][64 k-steps][64 lanes][2 row tiles of the wave]
    int dense_nOt;           // ceil(nm/16)
    // block widths (doubles, even)
    int RW, HW, RBW;
    int rb_tries;  // proposal tries held in a randomness block
    int user_n;    // normals are injected: tries beyond rb_tries are an error, not the generator's
    const double* rb;  // [W][N][RBW] window of randomness blocks
    int rb_t0;
    // injected tables in the ABI's layout (device copies), consumed by k_pregen_rng / k_exch_plan
    const double* user_utab;  // [T][N]
    const double* user_ntab;  // [T][K][np][N]
    const int32_t* pairtab;   // [T][n_pairs][2]
    int n_pairs_tab;
    // exchange plan window
    const unsigned long long* plan;  // [W][K]: pi | pj<<16 | ri<<32 | rj<<48
    const double* plan_mi;           // [W][K]: min_improve of chain pi
    // the same list grouped by dependency level (pairs of one level touch disjoint chains)
    const uint32_t* lv_pairs;        // [W][K]: pi | pj<<16, level by level
    const double* lv_mi;             // [W][K]: min_improve of chain pi, same order
    const uint32_t* lv_off;          // [W][K+2]: lv_off[l] = first position of level l; entry K+1 = number of levels
    // ... and for the lean walk (smm_walk_lean.hpp; N_global, K <= 8192, min_improve == 0; null otherwise):
    const uint32_t* lv_pairs_p;      // [W][plan_Kp]: level by level, every level padded to a multiple of 64 words with dummy pairs; a
                                     //          word = lean_unit * pi | lean_unit * pj << 16 (lean_unit = 8: the LDS byte offsets of the 8-byte slots)
    const uint32_t* lv_offp;         // [W][LV_OFFP]: entry l = first word of level l (entry nlev: the padded length); [33] = nlev;
                                     //          [34] = 1 when the plan fits this form (at most 31 levels)
    int plan_Kp, lean_unit;          // words per iteration in lv_pairs_p; 8, or 4 when 8 * N_global does not fit 16 bits (smm_walk_lean.hpp)
    int dist_fun;                    // smm_dist_fun_t: the exchange test's distance (AlgoBGP.jl:537,688); the key / lean / rows walks are for 0 (-)
    int gen_lean;                    // k_chain_iter: the inline walk on the lean form (exchange_walk_tile_lean)
    int lean_wide;                   // the lean walk's form for one min_improve > 0 (or NaN) shared by all chains: 16-byte slots, lean_unit 16 or 8
    // ... and for k_exch_resolve_rows (8192 < N_global <= 32768, min_improve == 0; null otherwise):
    const uint32_t* lv_rows;         // [W][rows_cap][1024]: level by level, every level padded to whole rows of 1024 words with dummy pairs; pi | pj << 16
    const uint32_t* lv_rowinfo;      // [W][4]: rows; bit r of words 1 (low) and 2 (high): row r is the last of its level; 1 when the plan fits the form
    int rows_cap;
    int plan_t0, plan_K;
    // ... and the cones of the inline key walk's workgroups (smm_cone.hpp; k_chain_iter's key form, 4096 < N <= 8192; null otherwise):
    const uint32_t* cone_ok;         // [W]: 1 when every workgroup's cone of the iteration fits CONE_LEVELS sub-levels
    const uint32_t* cone_hdr;        // [W][tiles][CONE_HDRW]: word 0 = sub-levels; from word 1: the pairs of sub-level s, one byte each
    const uint32_t* cone_pairs;      // [W][tiles][CONE_LEVELS * 64]: sub-level s = words 64 s .. 64 s + (its count) - 1, as lv_pairs_p words
    int cone_tiles, cone_ct;         // workgroups, chains per workgroup
    uint16_t* cone_gather;           // [W][tiles][CONE_GCAP] (k_chain_persist_norm; null otherwise): the chains of OTHER workgroups whose initial slots the
                                     //          cone needs; their number is in the high half of the header's word 0
    // ... the ring of k_chain_persist_norm (smm_chain_persist.hpp; null otherwise): tagged walk slots and self-validating records of the
    // last PR_K iterations, the tiles' progress words, the launch's abort word
    uint2* pr_slot;                  // [PR_K][Ng + 4]
    uint4* pr_rec;                   // [PR_K][Ng][RW]
    uint32_t* pr_progress;           // [tiles]
    uint32_t* pr_ctl;                // [0]: == pr_epoch when a tile of this launch gave up waiting
    uint32_t pr_epoch;               // launches of the persistent kernel so far (never 0): part of every tag
    int exch_from;                   // first iteration with an exchange (AlgoBGP.jl:637)
    // state
    double* cs;                // [N][CSW]
    unsigned long long* xres;  // [Ng]
    double* vals;              // [N] value of every chain's last accepted record after the accept step of the iteration whose
                               //     exchange is being resolved, contiguous (two buffers by iteration parity: the host points
                               //     vals at the one to read and vals_out at the one this launch's accept step writes — a
                               //     workgroup that starts late must not find its neighbours' new values in its walk's input)
    double* vals_out;
                               //     (what the single-shard exchange resolution reads: 8 B per chain instead of a record)
    uint2* slot8;              // [N] the chain's initial slot of the lean walk: {order_key32(vals[c]), c} (k_chain_iter_norm, or null)
    uint2* slot8_out;          // (like vals_out)
    uint32_t* walk_flags;      // bit 0 (sticky): a NaN value entered vals[]: order keys do not cover it, the 16-byte walk runs
    // single shards of 8192 < N <= 32768 chains: the initial slots of k_exch_resolve_rows / _key, written by the accept step itself
    // (no k_exch_keys pre-pass between chain kernel and resolution): key17 << 15 | chain and the NaN word of the iteration's parity
    // (null: the pre-pass makes them; the 16-bit slots of the rows walk's rare fallback are then made by the fallback itself)
    uint32_t* slots17_out; uint32_t* nan_flags_out;
    // scratch of the any-size exchange kernel
    int32_t *xsrc, *xpartner, *xnext, *xpairs;
    double* xval;
    void* xslot;   // [Ng] 16-byte chain slots {value, src, partner} of k_exch_resolve_lvl_big
    // history
    double* hrec;  // [T][N][HW]
    unsigned long long* err;
    int dbg;                 // SMMHIP_DBG timing experiments (results invalid when != 0)
    int scout_gl;            // ... lanes per scouting group (16; test hook: 8)
    int scout_after;         // k_chain_iter: rounds of one try per lane segment before mysample's remaining tries are scouted by groups of 8 lanes
    unsigned long long* ts;  // SMMHIP_TS=1: per-workgroup phase timestamps of k_chain_iter (tools/)
    int ts_levels;           // SMMHIP_TS=2: and one per level of the inline exchange walk
    // the p2p form of the sharded iteration (smm_p2p.hpp): every rank's window (this rank's own at index p2p_rank), mapped through
    // HIP IPC or, for contexts of one process, plain device pointers
    unsigned char* p2p_win[P2P_MAXG];
    unsigned char* p2p_self;       // == p2p_win[p2p_rank] (a member of its own: no dynamic index into the kernel arguments)
    int p2p_G, p2p_rank;
    uint32_t p2p_epoch;            // publications so far (the same on every rank): part of every tag
    unsigned long long p2p_want;   // arrivals per source rank this launch waits for before it reads its window
    uint32_t p2p_off[12];          // byte offsets inside a window: rec[0], rec[1], val[0], val[1], slot[0], slot[1], llrec[0], llrec[1], llval[0], llval[1], slot4[0], slot4[1] (p2p_layout)
};

// Order keys are computed from the HIGH WORD of the value (sign, exponent, 20 mantissa bits): for finite values >= 0 it is a
// monotone function of the value, and 32-bit integer arithmetic is all the key needs.  The scale is fixed: buckets of 2^10
// high-word steps (a factor 1 + 2^-10 apart, ~3 decimal digits) from 2^-48 upwards, 0xfff0 of them (up to 2^15.9 ~ 6e4);
// smaller values share bucket 0, larger ones the last bucket — undecided among themselves, still ordered against the rest.
// Negative values (the -1.0 of a failed objective, Eval.jl:84) and non-finite ones get the undecidable mark 0xffff.
constexpr uint32_t XKEY_BASE = (1023u - 48u) << 20, XKEY_SHIFT = 10, XKEY_TOP = 0xfff0u;
__host__ __device__ inline uint32_t order_key16(const double v) {
    const uint32_t hw = (uint32_t)(__builtin_bit_cast(unsigned long long, v) >> 32);
    if (hw >= 0x7ff00000u) return 0xffffu;                   // negative, infinite or NaN: always the exact values
    if (hw < XKEY_BASE) return 0u;
    const uint32_t k = 1u + ((hw - XKEY_BASE) >> XKEY_SHIFT);
    return k < XKEY_TOP ? k : XKEY_TOP;
}
// bounds of a bucket: lo inclusive, hi exclusive (bucket 0: [0, 2^-48); the last bucket: up to +inf)
__device__ inline double order_key_lo(const uint32_t k) { return k == 0 ? 0.0 : __hiloint2double((int)(XKEY_BASE + ((k - 1) << XKEY_SHIFT)), 0); }
__device__ inline double order_key_hi(const uint32_t k) {
    return k >= XKEY_TOP ? INFINITY : __hiloint2double((int)(XKEY_BASE + (k << XKEY_SHIFT)), 0);
}

// dist_fun(value_i, value_j) of the exchange test (AlgoBGP.jl:537,688; smm_dist_fun_t in include/smmhip.h): 0 is the reference's
// default `-`; the others are what the header offers in place of an arbitrary Julia function.  (Branches, not selects: the
// default must not pay for the division.)
__host__ __device__ inline double dist_fun_eval(const int kind, const double a, const double b) {
    const double d = a - b;
    if (__builtin_expect(kind == 0, 1)) return d;
    if (kind == 1) return fabs(d);
    return d / fabs(a);
}
constexpr int LV_OFFP = 40, LV_MAXLEV = 31;
constexpr int CONE_LEVELS = 32, CONE_HDRW = 1 + CONE_LEVELS / 4;   // a workgroup's cone: at most 32 sub-levels of 64 pairs (smm_cone.hpp)
constexpr int CONE_GCAP = 512;   // ... and at most this many chains of other workgroups (the persistent kernel's gather list, 1 KB)
// 32-bit order key of a chain value: for any two non-NaN doubles, key(a) > key(b) implies a > b and key(a) < key(b) implies a < b
// (the high word of the double, made monotone across the sign; -0.0 counts as +0.0); equal keys decide nothing.
__host__ __device__ inline uint32_t order_key32(const double v) {
    const uint32_t hw = v == 0.0 ? 0u : (uint32_t)(__builtin_bit_cast(unsigned long long, v) >> 32);
    return (hw & 0x80000000u) ? ~hw : (hw | 0x80000000u);
}

// 17-bit order key of a chain value on a fixed scale (buckets a factor 1 + 2^-11 apart from 2^-48 to 2^16): for non-NaN values a
// larger key means a larger value, equal keys decide nothing.  Everything below 2^-48 — zero, negative values, -Inf — shares bucket
// 0, everything from 2^16 up, +Inf included, the last one.
constexpr uint32_t XKEY17_SHIFT = 9, XKEY17_TOP = 0x1ffffu;
__host__ __device__ inline uint32_t order_key17(const double v) {
    const uint32_t hw = (uint32_t)(__builtin_bit_cast(unsigned long long, v) >> 32);
    if ((hw & 0x80000000u) || hw < XKEY_BASE) return 0u;
    const uint32_t k = 1u + ((hw - XKEY_BASE) >> XKEY17_SHIFT);
    return k < XKEY17_TOP ? k : XKEY17_TOP;
}

#define TS_MARK(i) do { if (P.ts && tid == 0) P.ts[(size_t)tile * 8 + (i)] = wall_clock64(); } while (0)

// has an iteration BEFORE t raised a hard error?  (Not "is the word set": a workgroup of iteration t that starts late may find the
// word set by a faster workgroup of the same launch — the failing iteration itself completes for every chain.)
__device__ inline bool error_before(const unsigned long long err_word, const int t) { return (err_word >> 34) < (unsigned long long)t; }
// the kinds of the sticky error word (its two low bits; atomicMin keeps the smallest key = the first failing iteration, the first failing chain of it,
// and for ONE chain in ONE iteration the error the reference would have raised first: its proposal (mysample, AlgoBGP.jl:409, called at :280) precedes its
// objective's value check (:341, inside doAcceptReject! at :287) — the enumeration of tests/test_gpu_error_enumeration.py found the two the other way
// round: a chain without a draw evaluates whatever its lanes hold, the value may be NaN, and "negative objective" used to win)
constexpr int ERRK_CAPACITY = 0, ERRK_NO_DRAW = 1, ERRK_NEGATIVE = 2, ERRK_FORM = 3;
__device__ inline void report_error(const KParams& P, int kind, int t, int gchain) {
    const unsigned long long key = ((unsigned long long)t << 34) | ((unsigned long long)gchain << 2) | (unsigned)kind;
    atomicMin(P.err, key);
}

__host__ __device__ inline int even_up(int x) { return (x + 1) & ~1; }

// direction of the proposal step for component k of one try, z = the try's np standard normals (LDS):
// z[k] for the reference's isotropic kernel MvNormal(mu01, sigma) (AlgoBGP.jl:442), (L z)_k with a Cholesky factor
// (include/smmhip.h: products rounded, added left to right; the build has -ffp-contract=off)
__device__ inline double prop_direction(const KParams& P, const double* z, const int k, const int gchain) {
    if (!P.chol_L) return z[k];
    const double* __restrict__ Lk = P.chol_L + (P.chol_per_chain ? (size_t)gchain * P.np * P.np : 0) + (size_t)k * P.np;
    double y = Lk[0] * z[0];
    for (int j = 1; j <= k; ++j) {
        const double pr = Lk[j] * z[j];
        y = y + pr;
    }
    return y;
}

// The in-kernel generator behind mysample's rare late tries, out of line: its ~40 live registers
// (Philox rounds, log, sincospi) then weigh only on the path that needs them.
__device__ __attribute__((noinline)) double2 rng_prop_normal2_outofline(uint64_t seed, uint32_t chain, uint32_t iter,
                                                                        uint32_t tr, uint32_t q) {
    double z0, z1;
    rng_prop_normal2(seed, chain, iter, tr, q, z0, z1);
    return make_double2(z0, z1);
}
