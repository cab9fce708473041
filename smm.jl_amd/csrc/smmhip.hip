// smmhip.hip — libsmmhip.so: hand-written HIP (gfx950 / MI355X) implementation of the BGP
// parallel-tempering iteration of floswald/SMM.jl behind the C ABI of include/smmhip.h.
//
// One iteration of computeNextIteration!(algo::MAlgoBGP) (src/mopt/AlgoBGP.jl:589-640) is two
// dependent launches on one HIP stream:
//   k_chain_iter      : next_eval for every chain at once — materialise the previous iteration's
//                       exchange (swap_ev_ij! :734-749), proposal (:424-471), objective
//                       (mprob.jl:175-188 -> ObjExamples.jl:59-116), doAcceptReject! (:324-392),
//                       set_eval! (:220-245).  A 512-lane workgroup owns a tile of CT chains, the
//                       ns simulated draws are spread over the lanes, every shock z is re-used CT
//                       times from a register, moments are reduced by a transposed wave reduction
//                       and combined through LDS.
//   k_exch_resolve_*  : exchangeMoves! (:647-716): ordered swap resolution by ONE workgroup with
//                       exact sequential semantics (data-flow over per-chain tickets in LDS).
// Everything that does not depend on the chains' state is produced ahead of the dependent loop by
// wide, latency-tolerant kernels, one window of iterations at a time:
//   k_pregen_rng      : proposal normals (first tries) and the MH uniforms (probs_acc, :85)
//   k_exch_plan       : the exchange pair list of every iteration and each pair's rank among the
//                       pairs of its two chains (the dependency structure of the sequential walk)
//
// Data layout in HBM (FP64 throughout): everything a chain needs per iteration sits in a few
// 16-byte aligned per-chain blocks (array-of-structures), so that the 64 lanes of a tile's
// control wave move a whole tile's state with a handful of dwordx4 instructions:
//   cs   [N][16]          chain state block (sigma, accept-rate counters, best/bestp, acc_tuner)
//   rec  [2][N][RW]       last accepted record: value, prob, status, params[np], simM[nm]
//   rb   [W][N][RBW]      randomness block of one iteration: u, normals[tries][np]
//   hrec [T][N][HW]       history record: value, prob, curr, best, best_id, exch, acc, status,
//                         params[np], simM[nm]   (transposed to the ABI's SoA on download)
//   xres [Ng]             exchange result: src | partner<<32
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <hip/hiprtc.h>
#include <dlfcn.h>
#include <mutex>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <math.h>
#include <algorithm>
#include <string>
#include <vector>

#include "../../include/smmhip.h"
#include "smm_rng.hpp"

namespace {

using namespace smm;

constexpr int WG = SMM_REDUCE_LANES;  // 512 lanes own one chain tile (numerical contract)
constexpr int MAX_DIM = 64;           // np, nm <= 64
constexpr int XWG = 1024;             // exchange workgroup
constexpr int XLDS_MAX = 8192;        // largest N_global resolved in LDS (16 B per chain)
constexpr unsigned XSPIN_LIMIT = 1u << 22;

// chain state block
constexpr int CSW = 16;
enum : int { CS_SIGMA = 0, CS_RATE, CS_NNOEX, CS_NACC, CS_LACC, CS_WASX, CS_BEST, CS_BESTID, CS_BESTP, CS_BESTPID, CS_ATUN,
             CS_PARTNER /* LDS only */ };
// history record
enum : int { H_VALUE = 0, H_PROB, H_CURR, H_BEST, H_BESTID, H_EXCH, H_ACC, H_STATUS, H_PARAMS };

// error word: min over (iter<<34 | chain<<2 | kind); 1 negative objective, 2 no draw, 3 internal
constexpr unsigned long long ERR_NONE = ~0ull;

enum : int { F_CLOSE_PREV = 1, F_HAS_PENDING = 2, F_WALK_INLINE = 4, F_GLOBAL_REC = 8, F_PROPOSE_ONLY = 16 };  // F_GLOBAL_REC: rec_in is the all-gathered buffer (global chain ids)

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
    // user objective (objective_id >= SMM_OBJ_USER_BASE): proposals out, results in, [N][np] / [N][nm] / [N]
    double* u_theta; double* u_simM; double* u_value; int* u_status;
    int mi_uniform;               // all thresholds equal (the usual case): mi_value
    int tile_off;                 // doubles in front of the tile's LDS blocks (the inline walk's chain slots)
    double mi_value;
    // dense objective (SMM_OBJ_DENSE): B and A in MFMA fragment order
    const double* dense_Bf;  // [D/16][ceil(np/4)][64]
    const double* dense_Af;  // [nOt][D/16][4][64]
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
    int plan_t0, plan_K;
    // state
    double* cs;                // [N][CSW]
    unsigned long long* xres;  // [Ng]
    double* vals;              // [N] value of every chain's last accepted record after the accept step, contiguous
                               //     (what the single-shard exchange resolution reads: 8 B per chain instead of a record)
    // scratch of the any-size exchange kernel
    int32_t *xsrc, *xpartner, *xnext, *xpairs;
    double* xval;
    // history
    double* hrec;  // [T][N][HW]
    unsigned long long* err;
    int dbg;                 // SMMHIP_DBG timing experiments (results invalid when != 0)
    unsigned long long* ts;  // SMMHIP_TS=1: per-workgroup phase timestamps of k_chain_iter (tools/)
};

#define TS_MARK(i) do { if (P.ts && tid == 0) P.ts[(size_t)tile * 8 + (i)] = wall_clock64(); } while (0)

__device__ inline void report_error(const KParams& P, int kind, int t, int gchain) {
    const unsigned long long key = ((unsigned long long)t << 34) | ((unsigned long long)gchain << 2) | (unsigned)kind;
    atomicMin(P.err, key);
}

__host__ __device__ inline int even_up(int x) { return (x + 1) & ~1; }

// The in-kernel generator behind mysample's rare late tries, out of line: its ~40 live registers
// (Philox rounds, log, sincospi) then weigh only on the path that needs them.
__device__ __attribute__((noinline)) double2 rng_prop_normal2_outofline(uint64_t seed, uint32_t chain, uint32_t iter,
                                                                        uint32_t tr, uint32_t q) {
    double z0, z1;
    rng_prop_normal2(seed, chain, iter, tr, q, z0, z1);
    return make_double2(z0, z1);
}

// ------------------------------------------------------------------------------------------
// Transposed wave reduction: every lane holds CT partial sums a[0..CT); on return lane l holds the
// 64-lane total of accumulator acc_index<CT>(l), combined by the canonical halving tree (offsets
// 32,16,8,4,2,1; IEEE addition is commutative so both partners compute the same bits).
// ------------------------------------------------------------------------------------------
template <int CT, int NN, int OFF>
__device__ inline void wave_reduce_step(double (&a)[CT], int lane) {
    if constexpr (NN > 1) {
        const bool upper = (lane & OFF) != 0;
#pragma unroll
        for (int i = 0; i < NN / 2; ++i) {
            const double mine = upper ? a[i + NN / 2] : a[i];
            const double send = upper ? a[i] : a[i + NN / 2];
            const double recv = __shfl_xor(send, OFF, 64);
            a[i] = mine + recv;
        }
        wave_reduce_step<CT, NN / 2, OFF / 2>(a, lane);
    } else if constexpr (OFF >= 1) {
        a[0] = a[0] + __shfl_xor(a[0], OFF, 64);
        wave_reduce_step<CT, 1, OFF / 2>(a, lane);
    }
}
template <int CT>
__device__ inline double wave_reduce_transposed(double (&a)[CT], int lane) {
    wave_reduce_step<CT, CT, 32>(a, lane);
    return a[0];
}
template <int CT>
__device__ inline int acc_index(int lane) {  // the log2(CT) top lane bits
    constexpr int LG = (CT == 1) ? 0 : (CT == 2) ? 1 : (CT == 4) ? 2 : (CT == 8) ? 3 : (CT == 16) ? 4 : (CT == 32) ? 5 : 6;
    return LG == 0 ? 0 : (lane >> (6 - LG));
}
template <int CT>
__device__ inline bool acc_writer(int lane) {
    return (lane & ((64 / CT) - 1)) == 0;
}

// The simulation of objfunc_norm (ObjExamples.jl:76-79) for a tile of CT chains:
// X[k,s] = theta_c[k] + z[k,s]; lane `tid` of the 512 sums its draws tid, tid+512, ... of moment k
// in that order (numerical contract).  Rows are processed in chunks of ZU; the shocks of the next
// chunk (of this or of the next moment) are loaded from the L2-resident matrix while the current
// chunk is added up.  Chunk 0 of moment 0 is loaded by the caller before its serial prologue.
constexpr int ZU = 8;

// The shock matrix is read through a buffer descriptor: row = scalar byte offset (SALU), lane = one constant
// 32-bit vector offset, so a chunk load is ZU buffer_load instructions and no vector address arithmetic.
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
struct ZBuf {
    __amdgpu_buffer_rsrc_t rsrc;
    int lane_off;  // tid * 8
    __device__ inline void init(const KParams& P, int tid) {
        rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)P.Z, 0, (int)((size_t)P.nm * P.zstride * sizeof(double)), 0x00020000);
        lane_off = tid * (int)sizeof(double);
    }
};
// chunk ch of moment k
__device__ inline void sim_load_chunk(const ZBuf& zb, const KParams& P, int k, int ch, double (&z)[ZU]) {
    const int row0 = (k * P.zstride + ((P.dbg & 8) ? 0 : ch) * (ZU * WG)) * (int)sizeof(double);  // dbg 8: timing experiment
#pragma unroll
    for (int u = 0; u < ZU; ++u)
        z[u] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(zb.rsrc, zb.lane_off, row0 + u * WG * (int)sizeof(double), 0));
}

// zc: chunk 0 of moment 0 (already loaded).  s_theta [CT][np], s_part [WG/64][CT][nm] in LDS.
// Moments are reduced in groups of G = 16/CT: one transposed reduction of G*CT accumulators has the
// same number of dependent shuffle steps as one of CT, so grouping halves that latency for CT = 8.
// A moment is nch chunks; the last one may be ragged (rows masked per lane).  Two chunks per trip, the
// two register buffers trade places; the chunk after a moment's last is chunk 0 of the next moment.
template <int CT>
__device__ inline void simulate_tile(const KParams& P, const ZBuf& zb, const double* s_theta, double* s_part, int tid, double (&zc)[ZU]) {
    constexpr int G = (CT >= 16) ? 1 : 16 / CT;
    const int lane = tid & 63, wave = tid >> 6;
    const int ns = P.ns, nm = P.nm;
    const int nch = (ns + ZU * WG - 1) / (ZU * WG);
    const int last_draws = ns - (nch - 1) * (ZU * WG);   // draws of the last chunk, 1 .. ZU*WG
    const bool ragged = last_draws < ZU * WG;
    for (int k0 = 0; k0 < nm; k0 += G) {
        double acc[G * CT];
#pragma unroll
        for (int i = 0; i < G * CT; ++i) acc[i] = 0.0;
#pragma unroll
        for (int kk = 0; kk < G; ++kk) {
            const int k = k0 + kk;
            if (k < nm) {
                double mu[CT];
#pragma unroll
                for (int c = 0; c < CT; ++c) mu[c] = s_theta[c * P.np + k];
                auto add_full = [&](const double (&z)[ZU]) {
#pragma unroll
                    for (int u = 0; u < ZU; ++u) {
#pragma unroll
                        for (int c = 0; c < CT; ++c) {
                            const double x = z[u] + mu[c];
                            acc[kk * CT + c] = acc[kk * CT + c] + x;
                        }
                    }
                };
                auto add_last = [&](const double (&z)[ZU]) {
                    if (!ragged) { add_full(z); return; }
#pragma unroll
                    for (int u = 0; u < ZU; ++u) {
                        if (tid + u * WG < last_draws) {
#pragma unroll
                            for (int c = 0; c < CT; ++c) {
                                const double x = z[u] + mu[c];
                                acc[kk * CT + c] = acc[kk * CT + c] + x;
                            }
                        }
                    }
                };
                const int knext = (k + 1 < nm) ? k + 1 : k;   // last moment: a harmless reload
                double zn[ZU];
                int ch = 0;
                for (; ch + 2 <= nch; ch += 2) {
                    sim_load_chunk(zb, P, k, ch + 1, zn);
                    add_full(zc);
                    const bool last = (ch + 2 == nch);
                    sim_load_chunk(zb, P, last ? knext : k, last ? 0 : ch + 2, zc);
                    if (last) add_last(zn); else add_full(zn);
                }
                if (ch < nch) {  // odd count: the last chunk is in zc; afterwards the buffers are swapped by copy
                    sim_load_chunk(zb, P, knext, 0, zn);
                    add_last(zc);
#pragma unroll
                    for (int u = 0; u < ZU; ++u) zc[u] = zn[u];
                }
            }
        }
        const double tot = wave_reduce_transposed<G * CT>(acc, lane);
        if (acc_writer<G * CT>(lane)) {
            const int a = acc_index<G * CT>(lane);
            const int kk = a / CT, c = a - kk * CT;
            if (k0 + kk < nm) s_part[(wave * CT + c) * nm + k0 + kk] = tot;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Dense objective (SMM_OBJ_DENSE, BASELINE config 5): the simulation is a dense contraction, so it
// runs on the FP64 matrix cores.  A tile is 16 chains = the N dimension of v_mfma_f64_16x16x4; wave w
// of the 8 owns the hidden units d in [32w, 32w+32):
//   x tile [16 d x 16 chains]  = B[16 d x np] * theta[np x 16]        (ceil(np/4) MFMAs)
//   h = tanh(x): the accumulator layout (row = (lane>>4) + 4r, col = lane&15) IS the B-operand layout
//   of the next product (k = 4s + (lane>>4)), so h feeds the second GEMM from registers;
//   y tile [16 k x 16 chains] += A[16 k x 16 d] * h[16 d x 16]        (4 MFMAs per output tile)
// B and A are stored in fragment order (one coalesced 8-byte load per lane per MFMA).  The wave's
// partial y goes to LDS [w][k][chain]; the 8 partials are added left to right by the chain lane.
// ------------------------------------------------------------------------------------------
typedef double d4_t __attribute__((ext_vector_type(4)));
constexpr int DENSE_D = SMM_DENSE_D;

template <int CT>
__device__ inline void dense_tile(const KParams& P, const double* s_theta, double* s_part, int tid) {
    static_assert(CT == 16, "the dense objective tiles 16 chains (MFMA N dimension)");
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int np = P.np, nPs = (np + 3) / 4, nOt = P.dense_nOt, nmp = nOt * 16;
    d4_t yacc[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) yacc[o] = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const int T = 2 * wave + tt;
        d4_t xacc = d4_t{0.0, 0.0, 0.0, 0.0};
        const double* __restrict__ bf = P.dense_Bf + (size_t)T * nPs * 64 + lane;
        for (int s = 0; s < nPs; ++s) {
            const int p = 4 * s + lk;
            const double b = (p < np) ? s_theta[li * np + p] : 0.0;
            xacc = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[(size_t)s * 64], b, xacc, 0, 0, 0);
        }
        double h[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = tanh(xacc[r]);
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            if (o < nOt) {
                const double* __restrict__ af = P.dense_Af + ((size_t)(o * (DENSE_D / 16) + T) * 4) * 64 + lane;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) yacc[o] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[s4 * 64], h[s4], yacc[o], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        if (o < nOt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) s_part[((size_t)wave * nmp + 16 * o + lk + 4 * r) * 16 + li] = yacc[o][r];
        }
    }
}

// value / simulated moments / status for one chain from its reduced sums
// (ObjExamples.jl:79-110; banana :251-265; "exception" -> status -2, mprob.jl:183-186).
// s_mom / s_w: data moments and weights staged in LDS.
template <int CT>
__device__ inline void finish_objective(const KParams& P, const double* theta /*LDS [np]*/, const double* s_part,
                                        const double* s_mom, const double* s_w, int ci, double* simM /*[nm] out, LDS*/,
                                        double& value, int& status, int c_local = 0) {
    if (P.obj == SMM_OBJ_USER) {  // evaluated by the user's kernel between the proposal and the accept launch
        for (int k = 0; k < P.nm; ++k) simM[k] = P.u_simM[(size_t)c_local * P.nm + k];
        value = P.u_value[c_local];
        status = P.u_status[c_local];
        return;
    }
    if (P.obj == SMM_OBJ_BANANA) {
        double v = 0.0;
        for (int i = 0; i + 1 < P.np; ++i) {
            const double a = theta[i], b = theta[i + 1];
            const double t1 = b - a * a;
            const double t2 = 1.0 - a;
            const double term = 100.0 * (t1 * t1) + t2 * t2;
            v = (i == 0) ? term : v + term;
        }
        for (int k = 0; k < P.nm; ++k) simM[k] = s_mom[k] + 2.2;
        value = v;
        status = 1;
        return;
    }
    if (P.obj == SMM_OBJ_NORM_FAILBOX && P.objp && theta[0] >= P.objp[0] && theta[0] <= P.objp[1]) {
        for (int k = 0; k < P.nm; ++k) simM[k] = NAN;
        value = -1.0;  // Eval() default, Eval.jl:84
        status = -2;
        return;
    }
    double vsum = 0.0;
    const bool dense = P.obj == SMM_OBJ_DENSE;
    const int nmp = P.dense_nOt * 16;
    for (int k = 0; k < P.nm; ++k) {
        double tot = dense ? s_part[((size_t)0 * nmp + k) * 16 + ci] : s_part[(0 * CT + ci) * P.nm + k];
#pragma unroll
        for (int wv = 1; wv < WG / 64; ++wv)
            tot = tot + (dense ? s_part[((size_t)wv * nmp + k) * 16 + ci] : s_part[(wv * CT + ci) * P.nm + k]);
        const double m = dense ? tot : tot / (double)P.ns;
        simM[k] = m;
        double d = m - s_mom[k];
        const double wk = s_w[k];
        if (!isnan(wk)) d = d / wk;
        const double v = d * d;
        vsum = (k == 0) ? v : vsum + v;
    }
    value = vsum / (double)P.nm;
    status = 1;
}

// ------------------------------------------------------------------------------------------
// Tile shared memory and wave-cooperative block moves
// ------------------------------------------------------------------------------------------
struct TileSmem {
    double *cs, *rb, *rec, *rout, *h, *hp, *theta, *lb, *ub, *init, *mom, *w, *part;
    unsigned* arrived;
    __device__ inline void carve(double* base, int CT, int np, int nm, int RW, int HW, int RBW, bool sim) {
        cs = base;                  // [CT][CSW]
        rb = cs + CT * CSW;         // [CT][RBW]
        rec = rb + CT * RBW;        // [CT][RW]   record the chain continues from
        rout = rec + CT * RW;       // [CT][RW]   record after this iteration's accept step
        h = rout + CT * RW;         // [CT][HW]   history record of iteration t
        hp = h + CT * HW;           // [CT][HW]   rewritten history record of iteration t-1 (exchanged chains)
        theta = hp + CT * HW;       // [CT][np]
        lb = theta + CT * np;       // [np] ...
        ub = lb + np;
        init = ub + np;
        mom = init + np;            // [nm]
        w = mom + nm;
        part = w + nm;              // [WG/64][CT][nm]
        arrived = (unsigned*)(part + (size_t)(WG / 64) * CT * nm);   // simulation kind: waves whose partial sums are in LDS
        (void)sim;
    }
};
__host__ __device__ inline size_t tile_smem_doubles(int CT, int np, int nm, int RW, int HW, int RBW, int kind) {
    const size_t part = kind == 1 ? (size_t)(WG / 64) * CT * nm : kind == 2 ? (size_t)(WG / 64) * (((nm + 15) / 16) * 16) * 16 : 0;
    return (size_t)CT * (CSW + RBW + 2 * RW + 2 * HW + np) + 3 * np + 2 * nm + part + 2;
}

// lanes (cl, r) of the control wave move chain cl's block of W doubles (W even) in 16-byte pieces
template <int CT>
__device__ inline void coop_load(double* lds_blk, const double* __restrict__ g, int W, int r) {
    constexpr int NR = 64 / CT;
    const double2* __restrict__ gs = (const double2*)g;
    double2* ld = (double2*)lds_blk;
#pragma unroll 2
    for (int i = r; i < W / 2; i += NR) ld[i] = gs[i];
}
// the same in two halves, so that the loads of several blocks are in flight together: coop_fetch requests
// the first NI pieces per lane into registers, coop_put writes them to LDS (and moves what is left of a long block)
template <int CT, int NI>
__device__ inline void coop_fetch(double2 (&v)[NI], const double* __restrict__ g, int W, int r) {
    constexpr int NR = 64 / CT;
    const double2* __restrict__ gs = (const double2*)g;
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int i = r + k * NR;
        v[k] = i < W / 2 ? gs[i] : make_double2(0.0, 0.0);
    }
}
template <int CT, int NI>
__device__ inline void coop_put(double* lds_blk, const double2 (&v)[NI], const double* __restrict__ g, int W, int r) {
    constexpr int NR = 64 / CT;
    const double2* __restrict__ gs = (const double2*)g;
    double2* ld = (double2*)lds_blk;
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int i = r + k * NR;
        if (i < W / 2) ld[i] = v[k];
    }
    for (int i = r + NI * NR; i < W / 2; i += NR) ld[i] = gs[i];
}
template <int CT>
__device__ inline void coop_store(double* __restrict__ g, const double* lds_blk, int W, int r) {
    constexpr int NR = 64 / CT;
    double2* __restrict__ gd = (double2*)g;
    const double2* ld = (const double2*)lds_blk;
#pragma unroll 2
    for (int i = r; i < W / 2; i += NR) gd[i] = ld[i];
}

// set_eval!(ci, ej) of swap_ev_ij! (AlgoBGP.jl:734-749) as a history record: the chain's record of
// the exchanged iteration tp is the donor's last accepted one (accepted = true, the donor's
// prob/status), curr = donor value, best recomputed against iteration tp-1 (:231-243).
__device__ inline void make_swapped_history(const KParams& P, double* hrec /*[HW]*/, const double* donor /*[RW]*/, int tp,
                                            int partner, double bpp, double bppid, double& bestv, double& bestid) {
    const double value = donor[0];
    if (value < bpp) { bestv = value; bestid = (double)tp; }
    else { bestv = bpp; bestid = bppid; }
    hrec[H_VALUE] = value; hrec[H_PROB] = donor[1]; hrec[H_CURR] = value; hrec[H_BEST] = bestv;
    hrec[H_BESTID] = bestid; hrec[H_EXCH] = (double)partner; hrec[H_ACC] = 1.0; hrec[H_STATUS] = donor[2];
    const int nv = P.np + P.nm;
    for (int k = 0; k < nv; ++k) hrec[H_PARAMS + k] = donor[3 + k];
}

// ------------------------------------------------------------------------------------------
// exchangeMoves! inside the chain kernel (single shard, N_global <= XLVL_MAX).
// The level walk of k_exch_resolve_lvl (below) costs ~7 us as a kernel of one workgroup plus a ~2.5 us
// kernel boundary.  Executed redundantly by EVERY tile in the prologue of the next k_chain_iter it costs
// the walk's ~4 us inside a kernel that is latency-structured anyway, and the boundary and the xres round
// trip disappear.  Same plan (k_exch_plan: pairs grouped by dependency level), same arithmetic, same result;
// the working set is 16 bytes per chain + 4 bytes per pair of LDS, the pair list overlaid by the tile's own
// blocks once the walk is over: 80 KB at N = 4096, so two tiles still share a CU.  Thresholds: one scalar when min_improve is uniform, else read from the plan (L2).
// ------------------------------------------------------------------------------------------
constexpr int XLVL_MAX = 4096;
struct __attribute__((aligned(16))) XSlot {  // one chain during the walk: 16 bytes, moved with one ds_read/write_b128
    double val;
    uint32_t src, partner;
};
// LDS of a tile with the inline walk: [XSlot slot[Ng]] [pairs[K] u32, later overlaid by the tile's own blocks]
__host__ __device__ inline size_t walk_slot_bytes(int Ng) { return (size_t)Ng * sizeof(XSlot); }

template <int NT>
__device__ inline void exchange_walk_tile(const KParams& P, const int tx, unsigned char* lds, const int tid) {
    const int Ng = P.Ng, K = P.plan_K;
    const int w = tx - P.plan_t0;
    XSlot* slot = (XSlot*)lds;                              // [Ng]
    uint32_t* pairs = (uint32_t*)(lds + walk_slot_bytes(Ng));   // [K]
    const uint32_t* __restrict__ g_off = P.lv_off + (size_t)w * (K + 2);
    const uint32_t* __restrict__ g_pairs = P.lv_pairs + (size_t)w * K;
    const double* __restrict__ g_mi = P.lv_mi + (size_t)w * K;
    const bool mi_u = P.mi_uniform != 0;
    const double mi_v = P.mi_value;
    // one round trip of global loads
    constexpr int PT = XLVL_MAX / NT;
    const int lane = tid & 63;
    double v_[PT];
    uint32_t pq_[PT];
#pragma unroll
    for (int r = 0; r < PT; ++r) {
        const int g = tid + r * NT;
        v_[r] = g < Ng ? P.vals[g] : 0.0;
    }
#pragma unroll
    for (int r = 0; r < PT; ++r) {
        const int q = tid + r * NT;
        pq_[r] = q < K ? g_pairs[q] : 0u;
    }
    const uint32_t ev = g_off[min(lane, K)];   // lane l: end of level l
    const int nlev = (int)g_off[K + 1];
#pragma unroll
    for (int r = 0; r < PT; ++r) {
        const int g = tid + r * NT;
        if (g < Ng) {
            XSlot s_;
            s_.val = v_[r]; s_.src = (uint32_t)g; s_.partner = 0;
            slot[g] = s_;
        }
    }
#pragma unroll
    for (int r = 0; r < PT; ++r) {
        const int q = tid + r * NT;
        if (q < K) pairs[q] = pq_[r];
    }
    for (int q = tid + PT * NT; q < K; q += NT) pairs[q] = g_pairs[q];   // injected pair lists longer than XLVL_MAX
    auto level_end = [&](int l) -> uint32_t {
        const int lc = min(l, nlev - 1);
        return lc < 64 ? (uint32_t)__builtin_amdgcn_readlane((int)ev, lc) : g_off[lc];
    };
    // The level sizes fall off geometrically.  The narrow tail (every remaining level <= 64 pairs) is walked by
    // wave 0 alone: LDS operations of one wave complete in order, so its levels need no workgroup barrier and
    // cost one LDS round trip each (a barrier level of a tile costs ~0.45 us next to a second walking tile).
    int ltail = nlev;
    if (nlev > 0 && nlev <= 64 && !(P.dbg & 128)) {
        const uint32_t lo = (uint32_t)__shfl_up((int)ev, 1, 64);
        const unsigned long long wide = __ballot(lane < nlev && ev - (lane > 0 ? lo : 0u) > 64u);
        ltail = wide ? 64 - __builtin_clzll(wide) : 0;
    }
    __syncthreads();
    uint32_t b = 0;
    uint32_t e = nlev > 0 ? level_end(0) : 0u;
    uint32_t e2 = nlev > 0 ? level_end(1) : 0u;
    uint32_t pw = (b + tid < e) ? pairs[b + tid] : 0u;
    double m = mi_u ? mi_v : ((b + tid < e) ? g_mi[b + tid] : 0.0);
#pragma clang loop unroll(disable)
    for (int l = 0; l < ltail; ++l) {
        const uint32_t e3 = level_end(l + 2);
        // this thread's first pair of the next level is fetched while this level runs
        const uint32_t pw2 = (e + tid < e2) ? pairs[e + tid] : 0u;
        const double m2 = mi_u ? mi_v : ((e + tid < e2) ? g_mi[e + tid] : 0.0);
        for (uint32_t pos = b + tid; pos < e; pos += NT) {
            if (pos != b + tid) { pw = pairs[pos]; m = mi_u ? mi_v : g_mi[pos]; }
            const uint32_t i = pw & 0xffffu, j = pw >> 16;
            const XSlot si = slot[i], sj = slot[j];
            if (si.val - sj.val > m) {                  // dist_fun = -, AlgoBGP.jl:688
                XSlot ni, nj;                           // swap_ev_ij!, :739-744; set_exchanged!, :747-748
                ni.val = sj.val; ni.src = sj.src; ni.partner = j + 1;
                nj.val = si.val; nj.src = si.src; nj.partner = i + 1;
                slot[i] = ni;
                slot[j] = nj;
            }
        }
        b = e; e = e2; e2 = e3; pw = pw2; m = m2;
        __syncthreads();
    }
    if (ltail < nlev) {
        if (tid < 64) {   // (pw, m) already hold this lane's pair of level ltail, e its end, e2 the next end
            constexpr uint32_t NOPAIR = 0xffffffffu;   // i == j == 0xffff never occurs (chain ids < XLVL_MAX)
            uint32_t cpw = (b + tid < e) ? pw : NOPAIR;
#pragma clang loop unroll(disable)
            for (int l = ltail; l < nlev; ++l) {
                const uint32_t e3 = level_end(l + 2);
                const uint32_t npw = (e + tid < e2) ? pairs[e + tid] : NOPAIR;
                const double m2 = mi_u ? mi_v : ((e + tid < e2) ? g_mi[e + tid] : 0.0);
                if (cpw != NOPAIR) {
                    const uint32_t i = cpw & 0xffffu, j = cpw >> 16;
                    const XSlot si = slot[i], sj = slot[j];
                    if (si.val - sj.val > m) {
                        XSlot ni, nj;
                        ni.val = sj.val; ni.src = sj.src; ni.partner = j + 1;
                        nj.val = si.val; nj.src = si.src; nj.partner = i + 1;
                        slot[i] = ni;
                        slot[j] = nj;
                    }
                }
                __builtin_amdgcn_wave_barrier();
                cpw = npw; m = m2; e = e2; e2 = e3;
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// k_chain_iter: one next_eval (AlgoBGP.jl:272-294) for every local chain, iteration t (1-based).
// rec_in : last accepted records after iteration t-1's accept step   [N][RW]
// rec_out: the same after iteration t's accept step (input of exchangeMoves!)
// Wave 0 is the tile's control wave: lane = r*CT + cl works for chain cl.  It moves the per-chain
// blocks with 16-byte pieces (two dependent levels: state/randomness/exchange result, then the
// record the chain continues from), evaluates the proposal tries side by side, and after the
// simulation lanes r == 0 run the accept step and the wave stores the result blocks.
// ------------------------------------------------------------------------------------------
// TPW tiles per workgroup (TPW = 2 with the inline exchange walk: the two tiles that would share a CU anyway
// become one workgroup of 1024 lanes, so the CU runs ONE walk with twice the lanes instead of two copies
// contending for its LDS; everything else is per tile, on the tile-local lane id).
template <int KIND, int CT, int TPW = 1>
__global__ __launch_bounds__(WG * TPW, 4) void k_chain_iter(const KParams P, const int t, const double* __restrict__ rec_in,
                                                            double* __restrict__ rec_out, const int flags) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int NR = 64 / CT;
    const int np = P.np, nm = P.nm, N = P.N, RW = P.RW, HW = P.HW, RBW = P.RBW;
    const int st = (TPW > 1) ? __builtin_amdgcn_readfirstlane((int)threadIdx.x / WG) : 0;   // tile of this workgroup (wave-uniform: scalar)
    const int tid = (int)threadIdx.x - st * WG;             // lane of the tile
    const int tile = (int)blockIdx.x * TPW + st;
    TileSmem S;
    S.carve(smem + P.tile_off + (size_t)st * ((tile_smem_doubles(CT, np, nm, RW, HW, RBW, KIND) + 1) & ~(size_t)1), CT, np, nm, RW, HW,
            RBW, KIND != 0);
    const int cl = tid % CT, r = (tid % 64) / CT;
    const int c = tile * CT + cl;             // chain served by this lane (control wave only)
    const bool ctl = tid < 64;
    const bool valid = ctl && (c < N);
    const bool chain_lane = valid && r == 0;
    const int gc = P.offset + c;
    TS_MARK(0);

    // ---- global reads, all issued before anything waits ----
    double za[ZU];
    ZBuf zb;
    if constexpr (KIND == 1) { zb.init(P, tid); sim_load_chunk(zb, P, 0, 0, za); }
    int partner = 0;
    // wave 1: problem constants, requested now and written to LDS after the walk
    const bool wave1 = tid >= 64 && tid < 128;
    const int k1 = tid - 64;
    double c_lb = 0.0, c_ub = 0.0, c_init = 0.0, c_mom = 0.0, c_w = 0.0;
    if (wave1) {
        if (k1 < np) { c_lb = P.lb[k1]; c_ub = P.ub[k1]; c_init = P.init[k1]; }
        if (k1 < nm) { c_mom = P.mom[k1]; c_w = P.w[k1]; }
    }
    {
        // level 1: exchange result, chain state block, this iteration's randomness block
        constexpr int NI_MAX = 4;   // pieces per lane held in registers; longer blocks finish with a load-store loop
        constexpr int NI_CS = (CSW / 2 + NR - 1) / NR < NI_MAX ? (CSW / 2 + NR - 1) / NR : NI_MAX;
        constexpr int NI_RB = (12 + NR - 1) / NR < NI_MAX ? (12 + NR - 1) / NR : NI_MAX;
        constexpr int NI_REC = (8 + NR - 1) / NR < NI_MAX ? (8 + NR - 1) / NR : NI_MAX;
        double2 v_cs[NI_CS], v_rb[NI_RB], v_rec[NI_REC];
        const int cc = valid ? c : 0;
        const double* g_cs = P.cs + (size_t)cc * CSW;
        const double* g_rb = P.rb + ((size_t)(t > 1 ? t - P.rb_t0 : 0) * N + cc) * RBW;
        const int rbw = t > 1 ? RBW : 0;
        unsigned long long xr = (unsigned long long)(unsigned)gc;
        if (valid) {
            if ((flags & F_HAS_PENDING) && !(flags & F_WALK_INLINE)) xr = P.xres[gc];
            coop_fetch<CT, NI_CS>(v_cs, g_cs, CSW, r);
            coop_fetch<CT, NI_RB>(v_rb, g_rb, rbw, r);
        }
        if (flags & F_WALK_INLINE) {
            // exchangeMoves! of iteration t-1, by all lanes of the tile, while the level-1 blocks are in flight
            // (the tile's own LDS blocks overlay the walk's pair list: nothing of the tile is written before this returns)
            exchange_walk_tile<WG * TPW>(P, t - 1, (unsigned char*)smem, (int)threadIdx.x);
            if (valid) {
                const XSlot sv = ((const XSlot*)smem)[gc];
                xr = (unsigned long long)sv.src | ((unsigned long long)sv.partner << 32);
            }
        }
        if (KIND == 1 && tid == 64) *S.arrived = 0u;
        if (wave1) {  // problem constants into the tile's LDS (which the walk's pair list occupied until now)
            if (k1 < np) { S.lb[k1] = c_lb; S.ub[k1] = c_ub; S.init[k1] = c_init; }
            if (k1 < nm) { S.mom[k1] = c_mom; S.w[k1] = c_w; }
            for (int k = k1 + 64; k < np; k += 64) { S.lb[k] = P.lb[k]; S.ub[k] = P.ub[k]; S.init[k] = P.init[k]; }
            for (int k = k1 + 64; k < nm; k += 64) { S.mom[k] = P.mom[k]; S.w[k] = P.w[k]; }
        }
        if (valid) {
            // level 2: the record the chain continues from (its own, or the donor's)
            const int s = (int)(unsigned)(xr & 0xffffffffu) - ((flags & F_GLOBAL_REC) ? 0 : P.offset);
            partner = (int)(xr >> 32);
            const double* g_rec = rec_in + (size_t)s * RW;
            coop_fetch<CT, NI_REC>(v_rec, g_rec, RW, r);
            coop_put<CT, NI_CS>(S.cs + cl * CSW, v_cs, g_cs, CSW, r);
            coop_put<CT, NI_RB>(S.rb + cl * RBW, v_rb, g_rb, rbw, r);
            coop_put<CT, NI_REC>(S.rec + cl * RW, v_rec, g_rec, RW, r);
        }
    }
    TS_MARK(1);
    __syncthreads();

    // ---- settle iteration t-1 (chain lanes; registers + LDS only) ----
    // (its results go back to the LDS block and are read again after the simulation: nothing of the serial
    // bookkeeping stays in registers across the register-hungry simulation loop)
    if (chain_lane) {
        double* csb = S.cs + cl * CSW;
        int nn = (int)csb[CS_NNOEX], na = (int)csb[CS_NACC];
        double bp = csb[CS_BEST], bpid = csb[CS_BESTID];
        if (t > 1) {
            bool exch_prev = false;
            if (partner != 0) {  // swap_ev_ij!, :734-749: iteration t-1's record becomes the donor's
                exch_prev = true;
                make_swapped_history(P, S.hp + cl * HW, S.rec + cl * RW, t - 1, partner, csb[CS_BESTP], csb[CS_BESTPID], bp, bpid);
            } else if (csb[CS_WASX] != 0.0) {  // sharded path: k_exch_apply already rewrote record and history
                exch_prev = true;
            }
            if ((flags & F_CLOSE_PREV) && !exch_prev) { nn += 1; na += (int)csb[CS_LACC]; }  // set_acceptRate!, :253-257
        }
        csb[CS_NNOEX] = (double)nn; csb[CS_NACC] = (double)na; csb[CS_BEST] = bp; csb[CS_BESTID] = bpid;
        csb[CS_PARTNER] = (double)partner;
    }
    TS_MARK(5);
    // ---- proposal(c), AlgoBGP.jl:424-471: lane (cl, r) evaluates try r of chain cl ----
    if (ctl) {
        double* th = S.theta + cl * np;
        const double* rc = S.rec + cl * RW;
        if (t == 1 || !valid || (P.dbg & 1)) {
            if (r == 0)
                for (int k = 0; k < np; ++k) th[k] = !valid ? 0.0 : (t == 1 ? S.init[k] : rc[3 + k]);  // :426-427
        } else {
            const int bs = P.batch_size;
            const int max_tries = P.user_n ? min(P.rb_tries, P.smpl_iters) : P.smpl_iters;
            const int npar = min(min(NR, P.rb_tries), max_tries);  // tries evaluated side by side
            const double sg = S.cs[cl * CSW + CS_SIGMA];
            const double* zz = S.rb + cl * RBW + 1;  // [tries][np]
            for (int b0 = 0; b0 < np; b0 += bs) {
                bool ok = r < npar;
                if (ok) {
                    for (int k = b0; k < b0 + bs; ++k) {  // mysample, :400-410, try r
                        const double lbk = S.lb[k];
                        const double mu01 = (rc[3 + k] - lbk) / (S.ub[k] - lbk);  // mapto_01, mprob.jl:248
                        const double step = sg * zz[r * np + k];  // MvNormal(mu01, sigma): x = mu + sigma*z
                        const double x = mu01 + step;
                        if (!(x >= 0.0 && x <= 1.0)) ok = false;  // inclusive bounds, :405
                    }
                }
                const unsigned long long m = __ballot(ok);  // first successful try of every chain
                unsigned long long pat = 0;
#pragma unroll
                for (int rr = 0; rr < NR; ++rr) pat |= ((m >> (rr * CT + cl)) & 1ull) << rr;
                const int rwin = pat ? (__ffsll((long long)pat) - 1) : -1;
                if (rwin == r) {
                    for (int k = b0; k < b0 + bs; ++k) {
                        const double lbk = S.lb[k];
                        const double span = S.ub[k] - lbk;
                        const double mu01 = (rc[3 + k] - lbk) / span;
                        const double step = sg * zz[r * np + k];
                        const double x = mu01 + step;
                        const double sc = x * span;
                        th[k] = sc + lbk;  // mapto_ab, mprob.jl:271
                    }
                } else if (rwin < 0 && r == 0) {  // rare: one try at a time (block, then the in-kernel generator)
                    bool ok2 = false;
                    for (int rr = npar; rr < max_tries && !ok2; ++rr) {
                        ok2 = true;
                        double zc0 = 0.0, zc1 = 0.0;
                        int zq = -1;
                        for (int k = b0; k < b0 + bs; ++k) {
                            const double lbk = S.lb[k];
                            const double mu01 = (rc[3 + k] - lbk) / (S.ub[k] - lbk);
                            double z;
                            if (rr < P.rb_tries) {
                                z = zz[rr * np + k];
                            } else {
                                if ((k >> 1) != zq) {
                                    zq = k >> 1;
                                    const double2 zz2 = rng_prop_normal2_outofline(P.seed, (uint32_t)gc, (uint32_t)t, (uint32_t)rr, (uint32_t)zq);
                                    zc0 = zz2.x; zc1 = zz2.y;
                                }
                                z = (k & 1) ? zc1 : zc0;
                            }
                            const double step = sg * z;
                            const double x = mu01 + step;
                            th[k] = x;
                            if (!(x >= 0.0 && x <= 1.0)) ok2 = false;
                        }
                    }
                    if (!ok2) report_error(P, 2, t, gc);  // :409
                    for (int k = b0; k < b0 + bs; ++k) {
                        const double lbk = S.lb[k];
                        const double span = S.ub[k] - lbk;
                        const double sc = th[k] * span;
                        th[k] = sc + lbk;
                    }
                }
            }
        }
    }
    TS_MARK(6);
    __syncthreads();
    TS_MARK(2);
    if (flags & F_PROPOSE_ONLY) {  // user objective: hand the proposals to the user's kernel; nothing has been stored yet, the
        if (valid)                 // accept launch repeats this (deterministic) prologue
            for (int k = r; k < np; k += NR) P.u_theta[(size_t)c * np + k] = S.theta[cl * np + k];
        return;
    }

    // ---- simulation: all 512 lanes, ns draws x nm moments x CT chains ----
    if constexpr (KIND == 1) {
        if (!(P.dbg & 2)) simulate_tile<CT>(P, zb, S.theta, S.part, tid, za);
        // No workgroup barrier here: only the tile's control wave consumes the partial sums.  Every wave announces
        // its partials with one LDS add and is done; the control wave waits for the tile's 8 announcements.  (With
        // two tiles per workgroup a barrier would also make the faster tile wait for the slower one.)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if ((tid & 63) == 0) __hip_atomic_fetch_add(S.arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (!ctl) return;
        while (__hip_atomic_load(S.arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (unsigned)(WG / 64))
            __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    } else if constexpr (KIND == 2) {
        dense_tile<CT>(P, S.theta, S.part, tid);
        __syncthreads();
    }
    TS_MARK(3);
    if (P.dbg & 4) return;

    // ---- objective value, doAcceptReject! (:324-392), set_eval! (:220-245): chain lanes ----
    if (chain_lane) {
        const double* th = S.theta + cl * np;
        const double* rc = S.rec + cl * RW;
        double* hr = S.h + cl * HW;
        double* ro = S.rout + cl * RW;
        double* csb = S.cs + cl * CSW;
        double* sm = hr + H_PARAMS + np;
        double value;
        int status;
        finish_objective<CT>(P, th, S.part, S.mom, S.w, cl, sm, value, status, c);
        const double sig = csb[CS_SIGMA], bp = csb[CS_BEST], bpid = csb[CS_BESTID], atun = csb[CS_ATUN];
        const int nn = (int)csb[CS_NNOEX], na = (int)csb[CS_NACC];
        const double u = t > 1 ? S.rb[cl * RBW] : 0.0;  // probs_acc[iter], :85

        const double old = rc[0];
        double prob;
        bool acc;
        if (t == 1) {  // :326-332
            prob = 1.0; acc = true; status = 1;
        } else if (status < 0) {  // :336-338
            prob = 0.0; acc = false;
        } else {
            if (!(value >= 0.0)) report_error(P, 1, t, gc);  // :341
            const double e = exp(atun * (old - value));
            prob = (e != e) ? e : (e < 1.0 ? e : 1.0);  // minimum([1.0,e]), NaN propagates (:344)
            if (!isfinite(prob)) { prob = 0.0; acc = false; status = -1; }  // :350-353
            else if (!isfinite(old)) { prob = 1.0; acc = true; }            // :355-359
            else { status = 1; acc = prob > u; }                            // strict >, :362-367
        }
        TS_MARK(7);
        // set_acceptRate!, :253-257 (iteration t has exchanged==0 at this point)
        const double rate = (double)(na + (acc ? 1 : 0)) / (double)(nn + 1);
        double nsig = sig;
        if (t > 1 && (t % P.sigma_update_steps) == 0)  // :381-390
            nsig = (rate > 0.234) ? sig * (1.0 + P.sigma_adjust_by) : sig * (1.0 - P.sigma_adjust_by);
        // set_eval!, :220-245
        double bestv, currv, bestid;
        if (t == 1) { bestv = value; currv = value; bestid = 1.0; }
        else {
            currv = acc ? value : old;  // curr_val[t-1] == value of the last accepted record
            if (value < bp) { bestv = value; bestid = (double)t; }
            else { bestv = bp; bestid = bpid; }
        }
        csb[CS_SIGMA] = nsig; csb[CS_RATE] = rate; csb[CS_NNOEX] = (double)nn; csb[CS_NACC] = (double)na;
        csb[CS_LACC] = acc ? 1.0 : 0.0; csb[CS_WASX] = 0.0; csb[CS_BEST] = bestv; csb[CS_BESTID] = bestid;
        csb[CS_BESTP] = bp; csb[CS_BESTPID] = bpid;  // best after t-1: needed if iteration t gets exchanged
        hr[H_VALUE] = value; hr[H_PROB] = prob; hr[H_CURR] = currv; hr[H_BEST] = bestv; hr[H_BESTID] = bestid;
        hr[H_EXCH] = 0.0; hr[H_ACC] = acc ? 1.0 : 0.0; hr[H_STATUS] = (double)status;
        for (int k = 0; k < np; ++k) hr[H_PARAMS + k] = th[k];
        // the chain's last accepted record (lastAccepted :209-215) = input of the exchange step
        if (acc) {
            ro[0] = value; ro[1] = prob; ro[2] = (double)status;
            for (int k = 0; k < np; ++k) ro[3 + k] = th[k];
            for (int k = 0; k < nm; ++k) ro[3 + np + k] = sm[k];
        } else {
            for (int f = 0; f < RW; ++f) ro[f] = rc[f];
        }
        P.vals[c] = acc ? value : old;
    }
    // ---- the control wave stores the tile's result blocks ----
    if (valid) {
        __builtin_amdgcn_wave_barrier();
        coop_store<CT>(P.cs + (size_t)c * CSW, S.cs + cl * CSW, CSW, r);
        coop_store<CT>(rec_out + (size_t)c * RW, S.rout + cl * RW, RW, r);
        coop_store<CT>(P.hrec + ((size_t)(t - 1) * N + c) * HW, S.h + cl * HW, HW, r);
        if (t > 1 && S.cs[cl * CSW + CS_PARTNER] != 0.0)
            coop_store<CT>(P.hrec + ((size_t)(t - 2) * N + c) * HW, S.hp + cl * HW, HW, r);
    }
    TS_MARK(4);
}

// k_flush: settle the last iteration (pending exchange + accept-rate counters) without starting a
// new one, so that state/history can be read back or saved (save/readMalgo, AlgoAbstract.jl:83-102).
__global__ void k_flush(const KParams P, const int t_next, const double* __restrict__ rec_in, double* __restrict__ rec_out,
                        const int flags) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= P.N) return;
    const int RW = P.RW, HW = P.HW, N = P.N;
    double* csb = P.cs + (size_t)c * CSW;
    const int goff = (flags & F_GLOBAL_REC) ? 0 : P.offset;   // rec_in indexed by global chain id (all-gathered buffer)?
    int s = P.offset + c - goff;
    bool exch = false;
    if (flags & F_HAS_PENDING) {
        const unsigned long long xr = P.xres[P.offset + c];
        const int partner = (int)(xr >> 32);
        if (partner != 0) {
            exch = true;
            s = (int)(unsigned)(xr & 0xffffffffu) - goff;
            const int tp = t_next - 1;
            const double* donor = rec_in + (size_t)s * RW;
            double* hrec = P.hrec + ((size_t)(tp - 1) * N + c) * HW;
            const double value = donor[0];
            double bestv, bestid;
            if (value < csb[CS_BESTP]) { bestv = value; bestid = (double)tp; }
            else { bestv = csb[CS_BESTP]; bestid = csb[CS_BESTPID]; }
            hrec[H_VALUE] = value; hrec[H_PROB] = donor[1]; hrec[H_CURR] = value; hrec[H_BEST] = bestv;
            hrec[H_BESTID] = bestid; hrec[H_EXCH] = (double)partner; hrec[H_ACC] = 1.0; hrec[H_STATUS] = donor[2];
            for (int k = 0; k < P.np + P.nm; ++k) hrec[H_PARAMS + k] = donor[3 + k];
            csb[CS_BEST] = bestv; csb[CS_BESTID] = bestid;
        }
    } else if (csb[CS_WASX] != 0.0) {
        exch = true;
        csb[CS_WASX] = 0.0;
    }
    if ((flags & F_CLOSE_PREV) && !exch) {
        csb[CS_NNOEX] += 1.0;
        csb[CS_NACC] += csb[CS_LACC];
    }
    for (int f = 0; f < RW; ++f) rec_out[(size_t)c * RW + f] = rec_in[(size_t)s * RW + f];
}

// batched evaluateObjective(m,p), mprob.jl:175-188: params [np][M] -> value, simM [nm][M], status
template <int KIND, int CT>
__global__ __launch_bounds__(WG, 4) void k_eval_batch(const KParams P, const double* __restrict__ params, const int M,
                                                      double* __restrict__ value, double* __restrict__ simM,
                                                      int8_t* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    TileSmem S;
    S.carve(smem, CT, P.np, P.nm, P.RW, P.HW, P.RBW, KIND != 0);
    const int tid = threadIdx.x;
    const int i = blockIdx.x * CT + tid;
    const bool chain_lane = (tid < CT) && (i < M);
    double za[ZU];
    ZBuf zb;
    if constexpr (KIND == 1) { zb.init(P, tid); sim_load_chunk(zb, P, 0, 0, za); }
    if (tid >= 64 && tid < 128)
        for (int k = tid - 64; k < P.nm; k += 64) { S.mom[k] = P.mom[k]; S.w[k] = P.w[k]; }
    if (tid < CT)
        for (int k = 0; k < P.np; ++k) S.theta[tid * P.np + k] = chain_lane ? params[(size_t)k * M + i] : 0.0;
    __syncthreads();
    if constexpr (KIND == 1) {
        simulate_tile<CT>(P, zb, S.theta, S.part, tid, za);
        __syncthreads();
    } else if constexpr (KIND == 2) {
        dense_tile<CT>(P, S.theta, S.part, tid);
        __syncthreads();
    }
    if (chain_lane) {
        double v;
        int st;
        double* sm = S.h + tid * P.HW;
        finish_objective<CT>(P, S.theta + tid * P.np, S.part, S.mom, S.w, tid, sm, v, st);
        value[i] = v;
        status[i] = (int8_t)st;
        for (int k = 0; k < P.nm; ++k) simM[(size_t)k * M + i] = sm[k];
    }
}

// ------------------------------------------------------------------------------------------
// k_pregen_rng: the state-independent randomness of iterations t0 .. t0+W-1 as per-chain blocks
//   rb[w][c] = { u, z[try][k] }:  u = the MH uniform (probs_acc = rand(n), AlgoBGP.jl:85),
//   z = standard normals of mysample's first tries (rand(RAND,d), :404) — injected or generated.
// one thread per (iteration, try, parameter pair, chain).
// ------------------------------------------------------------------------------------------
__global__ void k_pregen_rng(const KParams P, const int t0, const int W, double* __restrict__ rb) {
    const int N = P.N, np = P.np, TR = P.rb_tries;
    const int Q = (np + 1) / 2;
    const size_t total = (size_t)W * TR * Q * N;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int q = (int)(i % Q);
    size_t rest = i / Q;
    const int r = (int)(rest % TR); rest /= TR;
    const int c = (int)(rest % N);
    const int w = (int)(rest / N);
    const int t = t0 + w;
    const uint32_t gc = (uint32_t)(P.offset + c);
    double* blk = rb + ((size_t)w * N + c) * P.RBW;
    if (t > 1) {  // iteration 1 proposes the initial value (:426-427)
        double z0, z1 = 0.0;
        if (P.user_ntab) {
            const size_t base = (((size_t)(t - 1) * TR + r) * np) * N + c;
            z0 = P.user_ntab[base + (size_t)(2 * q) * N];
            if (2 * q + 1 < np) z1 = P.user_ntab[base + (size_t)(2 * q + 1) * N];
        } else {
            rng_prop_normal2(P.seed, gc, (uint32_t)t, (uint32_t)r, (uint32_t)q, z0, z1);
        }
        blk[1 + r * np + 2 * q] = z0;
        if (2 * q + 1 < np) blk[1 + r * np + 2 * q + 1] = z1;
    }
    if (r == 0 && q == 0) blk[0] = P.user_utab ? P.user_utab[(size_t)(t - 1) * N + c] : rng_u(P.seed, gc, (uint32_t)t);
}

// ------------------------------------------------------------------------------------------
// k_exch_plan: one workgroup per iteration t = t0 + blockIdx.x.  Samples the exchange pair list
// (sample(props, K, replace=false), AlgoBGP.jl:653-656) and derives the dependency structure of
// the ordered walk (:662-691): for pair q = (i,j), r_i / r_j = number of earlier pairs touching
// chain i / chain j (counting sort of the 2K endpoints by chain: LDS atomics + block scan).
// plan[t-t0][q] = i | j<<16 | r_i<<32 | r_j<<48, plan_mi[t-t0][q] = min_improve[i].
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(XWG) void k_exch_plan(const KParams P, const int t0, unsigned long long* __restrict__ plan,
                                                   double* __restrict__ plan_mi, uint32_t* __restrict__ lv_pairs,
                                                   double* __restrict__ lv_mi, uint32_t* __restrict__ lv_off) {
    extern __shared__ __attribute__((aligned(16))) unsigned char xsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = t0 + blockIdx.x;
    const int Ng = P.Ng, K = P.plan_K;
    uint32_t* cnt = (uint32_t*)xsm;          // [Ng+2]  histogram -> cursor (later: level histogram)
    uint32_t* ep = cnt + Ng + 2;             // [2K]  list positions bucketed by chain (later: levels)
    uint16_t* pi = (uint16_t*)(ep + 2 * K);  // [K]
    uint16_t* pj = pi + K;                   // [K]
    uint32_t* wsum = (uint32_t*)(pj + K);    // [32] (pi,pj: 4K bytes from a 4-byte aligned base)
    unsigned long long* out = plan + (size_t)blockIdx.x * K;
    double* out_mi = plan_mi + (size_t)blockIdx.x * K;

    for (int c = tid; c < Ng; c += XWG) cnt[c] = 0;
    if (P.pairtab) {
        for (int q = tid; q < K; q += XWG) {
            pi[q] = (uint16_t)P.pairtab[((size_t)(t - 1) * K + q) * 2];
            pj[q] = (uint16_t)P.pairtab[((size_t)(t - 1) * K + q) * 2 + 1];
        }
    } else {
        PairPerm pp;
        pp.init(P.seed, (uint32_t)t, (uint64_t)Ng * (uint64_t)(Ng - 1) / 2);
        for (int q = tid; q < K; q += XWG) {
            int32_t i, j;
            pair_unrank(pp.eval((uint64_t)q), i, j);
            pi[q] = (uint16_t)i;
            pj[q] = (uint16_t)j;
        }
    }
    __syncthreads();
    for (int q = tid; q < K; q += XWG) {  // histogram of endpoints
        atomicAdd(&cnt[pi[q]], 1u);
        atomicAdd(&cnt[pj[q]], 1u);
    }
    __syncthreads();
    {   // exclusive scan of cnt[0..Ng) -> bucket start
        constexpr int PER = XLDS_MAX / XWG;
        const int per = (Ng + XWG - 1) / XWG;
        const int c0 = tid * per;
        uint32_t loc[PER];
        uint32_t sum = 0;
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int c = c0 + u;
            const uint32_t v = (u < per && c < Ng) ? cnt[c] : 0u;
            loc[u] = sum;
            sum += v;
        }
        uint32_t incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t base = 0;
        for (int w = 0; w < wave; ++w) base += wsum[w];
        const uint32_t excl = base + incl - sum;
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int c = c0 + u;
            if (u < per && c < Ng) cnt[c] = excl + loc[u];
        }
    }
    __syncthreads();
    for (int q = tid; q < K; q += XWG) {  // scatter (order inside a bucket is arbitrary)
        ep[atomicAdd(&cnt[pi[q]], 1u)] = (uint32_t)q;
        ep[atomicAdd(&cnt[pj[q]], 1u)] = (uint32_t)q;
    }
    __syncthreads();  // now cnt[c] == end of chain c's bucket
    // rank = number of smaller list positions in the bucket; kept in registers for the level pass
    constexpr int MAXPP = XLDS_MAX / XWG;
    uint16_t rri[MAXPP], rrj[MAXPP];
#pragma unroll
    for (int m = 0; m < MAXPP; ++m) {
        const int q = tid + m * XWG;
        rri[m] = 0; rrj[m] = 0;
        if (q < K) {
            const uint32_t i = pi[q], j = pj[q];
            uint32_t b = i ? cnt[i - 1] : 0u, e = cnt[i], ri = 0, rj = 0;
            for (uint32_t x = b; x < e; ++x) ri += (ep[x] < (uint32_t)q) ? 1u : 0u;
            b = j ? cnt[j - 1] : 0u; e = cnt[j];
            for (uint32_t x = b; x < e; ++x) rj += (ep[x] < (uint32_t)q) ? 1u : 0u;
            out[q] = (unsigned long long)i | ((unsigned long long)j << 16) | ((unsigned long long)ri << 32) |
                     ((unsigned long long)rj << 48);
            out_mi[q] = P.min_improve_g[i];  // the threshold of the pair's colder chain, AlgoBGP.jl:688
            rri[m] = (uint16_t)ri; rrj[m] = (uint16_t)rj;
        }
    }
    __syncthreads();
    // ---- dependency levels: level(q) = 1 + max(level of q's predecessor on chain i, on chain j) ----
    // buckets re-written in rank order, so that the predecessor of rank r is the entry of rank r-1
#pragma unroll
    for (int m = 0; m < MAXPP; ++m) {
        const int q = tid + m * XWG;
        if (q < K) {
            const uint32_t i = pi[q], j = pj[q];
            ep[(i ? cnt[i - 1] : 0u) + rri[m]] = (uint32_t)q;
            ep[(j ? cnt[j - 1] : 0u) + rrj[m]] = (uint32_t)q;
        }
    }
    __syncthreads();
    int prei[MAXPP], prej[MAXPP];
#pragma unroll
    for (int m = 0; m < MAXPP; ++m) {
        const int q = tid + m * XWG;
        prei[m] = -1; prej[m] = -1;
        if (q < K) {
            const uint32_t i = pi[q], j = pj[q];
            if (rri[m]) prei[m] = (int)ep[(i ? cnt[i - 1] : 0u) + rri[m] - 1];
            if (rrj[m]) prej[m] = (int)ep[(j ? cnt[j - 1] : 0u) + rrj[m] - 1];
        }
    }
    __syncthreads();  // cnt / ep are free from here on
    uint16_t* lvl = (uint16_t*)ep;          // [K]
    uint32_t* lhist = cnt;                   // [nlev+1] <= Ng+2 entries
    for (int q = tid; q < K; q += XWG) lvl[q] = 0;
    __syncthreads();
    int changed = 1;
    while (changed) {  // Jacobi sweeps: converges after (number of levels) sweeps
        int mine = 0;
        uint16_t nl[MAXPP];
#pragma unroll
        for (int m = 0; m < MAXPP; ++m) {
            const int q = tid + m * XWG;
            nl[m] = 0;
            if (q < K) {
                const uint32_t a = prei[m] >= 0 ? lvl[prei[m]] : 0u, b = prej[m] >= 0 ? lvl[prej[m]] : 0u;
                const bool known = (prei[m] < 0 || a) && (prej[m] < 0 || b);
                nl[m] = known ? (uint16_t)(1u + (a > b ? a : b)) : (uint16_t)0;
            }
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MAXPP; ++m) {
            const int q = tid + m * XWG;
            if (q < K && nl[m] != lvl[q]) { lvl[q] = nl[m]; mine = 1; }
        }
        changed = __syncthreads_or(mine);
    }
    // counting sort of the pairs by level
    for (int c = tid; c < Ng + 2; c += XWG) lhist[c] = 0;
    __syncthreads();
    for (int q = tid; q < K; q += XWG) atomicAdd(&lhist[lvl[q]], 1u);  // lhist[l] = size of level l (1-based), lhist[0] = 0
    __syncthreads();
    uint32_t* wsum2 = wsum + 16;
    __shared__ uint32_t s_nlev;
    if (tid == 0) s_nlev = 0;
    __syncthreads();
    {
        uint32_t mx = 0;
        for (int q = tid; q < K; q += XWG) mx = lvl[q] > mx ? lvl[q] : mx;
        atomicMax(&s_nlev, mx);
    }
    __syncthreads();
    const int nlev = (int)s_nlev;
    {   // exclusive scan of lhist[0..nlev] -> first position of level l (stored at lhist[l-1] after the shift below)
        constexpr int PER = XLDS_MAX / XWG + 1;
        const int n = nlev + 1;
        const int per = (n + XWG - 1) / XWG;
        const int c0 = tid * per;
        uint32_t loc[PER];
        uint32_t sum = 0;
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int c = c0 + u;
            const uint32_t v = (u < per && c < n) ? lhist[c] : 0u;
            loc[u] = sum;
            sum += v;
        }
        uint32_t incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        if (lane == 63) wsum2[wave] = incl;
        __syncthreads();
        uint32_t base = 0;
        for (int w = 0; w < wave; ++w) base += wsum2[w];
        const uint32_t excl = base + incl - sum;
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int c = c0 + u;
            if (u < per && c < n) lhist[c] = excl + loc[u];  // = number of pairs in levels < c  (level c starts here)
        }
    }
    __syncthreads();
    uint32_t* o_off = lv_off + (size_t)blockIdx.x * (K + 2);
    // o_off[l] = end of the l-th level (0-based) = start of 1-based level l+2
    for (int l = tid; l < nlev; l += XWG) o_off[l] = (l + 2 <= nlev) ? lhist[l + 2] : (uint32_t)K;
    if (tid == 0) o_off[K + 1] = (uint32_t)nlev;
    __syncthreads();
    uint32_t* o_pairs = lv_pairs + (size_t)blockIdx.x * K;
    double* o_mi = lv_mi + (size_t)blockIdx.x * K;
    for (int q = tid; q < K; q += XWG) {
        const uint32_t pos = atomicAdd(&lhist[lvl[q]], 1u);
        const uint32_t i = pi[q], j = pj[q];
        o_pairs[pos] = i | (j << 16);
        o_mi[pos] = P.min_improve_g[i];
    }
}

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

#define XTS(i) do { if (P.ts && tid == 0) P.ts[(size_t)8 * 60000 + (i)] = wall_clock64(); } while (0)
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
                if (vi - vj > mi) {                         // dist_fun = -, :688
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
            if (si.val - sj.val > m) {                  // dist_fun = -, AlgoBGP.jl:688
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
__global__ __launch_bounds__(LWG) void k_exch_resolve_lvl_soa(const KParams P, const int t, const double* __restrict__ gathered) {
    extern __shared__ __attribute__((aligned(16))) unsigned char xsm[];
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
            if (vi - vj > m) {                          // dist_fun = -, AlgoBGP.jl:688
                val[i] = vj; val[j] = vi;               // swap_ev_ij!, :739-744; set_exchanged!, :747-748
                sp[i] = (sj & 0xffffu) | ((j + 1) << 16);
                sp[j] = (si & 0xffffu) | ((i + 1) << 16);
            }
        }
        b = e; e = e2; e2 = e3; pw = pw2; m = m2;
        __syncthreads();
    }
    for (int g = tid; g < Ng; g += LWG) {
        const uint32_t s_ = sp[g];
        P.xres[g] = (unsigned long long)(s_ & 0xffffu) | ((unsigned long long)(s_ >> 16) << 32);
    }
}

// ------------------------------------------------------------------------------------------
// Large populations (8192 < N_global <= 65535, e.g. 8 GPUs x 4096 chains): the same level plan and
// level-synchronous walk with their working sets in global memory (the LDS of one CU is too small).
// k_exch_plan_big: one workgroup per iteration, scratch [blockIdx] in global memory; speed is not
// critical (runs ahead of the dependent loop, one window at a time).
// ------------------------------------------------------------------------------------------
struct BigPlanScratch {  // per workgroup
    uint32_t *cnt, *ep, *pi, *pj, *ri, *rj, *prei, *prej, *lvl, *nl;
    __host__ __device__ static size_t words(int Ng, int K) { return (size_t)(Ng + 4) + (size_t)K * 10; }
    __device__ void carve(uint32_t* base, int Ng, int K) {
        cnt = base; ep = cnt + Ng + 4; pi = ep + 2 * K; pj = pi + K; ri = pj + K; rj = ri + K;
        prei = rj + K; prej = prei + K; lvl = prej + K; nl = lvl + K;
    }
};

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

__global__ __launch_bounds__(XWG) void k_exch_plan_big(const KParams P, const int t0, uint32_t* __restrict__ scratch,
                                                       uint32_t* __restrict__ lv_pairs, double* __restrict__ lv_mi,
                                                       uint32_t* __restrict__ lv_off) {
    __shared__ uint32_t wsum[XWG / 64];
    __shared__ uint32_t s_nlev;
    const int tid = threadIdx.x;
    const int t = t0 + blockIdx.x;
    const int Ng = P.Ng, K = P.plan_K;
    BigPlanScratch S;
    S.carve(scratch + (size_t)blockIdx.x * BigPlanScratch::words(Ng, K), Ng, K);
    for (int c = tid; c < Ng + 4; c += XWG) S.cnt[c] = 0;
    if (P.pairtab) {
        for (int q = tid; q < K; q += XWG) {
            S.pi[q] = (uint32_t)P.pairtab[((size_t)(t - 1) * K + q) * 2];
            S.pj[q] = (uint32_t)P.pairtab[((size_t)(t - 1) * K + q) * 2 + 1];
        }
    } else {
        PairPerm pp;
        pp.init(P.seed, (uint32_t)t, (uint64_t)Ng * (uint64_t)(Ng - 1) / 2);
        for (int q = tid; q < K; q += XWG) {
            int32_t i, j;
            pair_unrank(pp.eval((uint64_t)q), i, j);
            S.pi[q] = (uint32_t)i;
            S.pj[q] = (uint32_t)j;
        }
    }
    __syncthreads();
    for (int q = tid; q < K; q += XWG) { atomicAdd(&S.cnt[S.pi[q]], 1u); atomicAdd(&S.cnt[S.pj[q]], 1u); }
    __syncthreads();
    block_excl_scan(S.cnt, Ng, wsum, tid);  // bucket starts
    for (int q = tid; q < K; q += XWG) {   // scatter (arbitrary order inside a bucket)
        S.ep[atomicAdd(&S.cnt[S.pi[q]], 1u)] = (uint32_t)q;
        S.ep[atomicAdd(&S.cnt[S.pj[q]], 1u)] = (uint32_t)q;
    }
    __syncthreads();  // cnt[c] == end of bucket c
    for (int q = tid; q < K; q += XWG) {   // ranks
        const uint32_t i = S.pi[q], j = S.pj[q];
        uint32_t b = i ? S.cnt[i - 1] : 0u, e = S.cnt[i], ri = 0, rj = 0;
        for (uint32_t x = b; x < e; ++x) ri += (S.ep[x] < (uint32_t)q) ? 1u : 0u;
        b = j ? S.cnt[j - 1] : 0u; e = S.cnt[j];
        for (uint32_t x = b; x < e; ++x) rj += (S.ep[x] < (uint32_t)q) ? 1u : 0u;
        S.ri[q] = ri; S.rj[q] = rj;
    }
    __syncthreads();
    for (int q = tid; q < K; q += XWG) {   // buckets in rank order
        const uint32_t i = S.pi[q], j = S.pj[q];
        S.ep[(i ? S.cnt[i - 1] : 0u) + S.ri[q]] = (uint32_t)q;
        S.ep[(j ? S.cnt[j - 1] : 0u) + S.rj[q]] = (uint32_t)q;
    }
    __syncthreads();
    for (int q = tid; q < K; q += XWG) {   // predecessors
        const uint32_t i = S.pi[q], j = S.pj[q];
        S.prei[q] = S.ri[q] ? S.ep[(i ? S.cnt[i - 1] : 0u) + S.ri[q] - 1] : 0xffffffffu;
        S.prej[q] = S.rj[q] ? S.ep[(j ? S.cnt[j - 1] : 0u) + S.rj[q] - 1] : 0xffffffffu;
        S.lvl[q] = 0;
    }
    __syncthreads();
    int changed = 1;
    while (changed) {  // Jacobi sweeps
        for (int q = tid; q < K; q += XWG) {
            const uint32_t pa = S.prei[q], pb = S.prej[q];
            const uint32_t a = pa != 0xffffffffu ? S.lvl[pa] : 0u, b = pb != 0xffffffffu ? S.lvl[pb] : 0u;
            const bool known = (pa == 0xffffffffu || a) && (pb == 0xffffffffu || b);
            S.nl[q] = known ? 1u + (a > b ? a : b) : 0u;
        }
        __syncthreads();
        int mine = 0;
        for (int q = tid; q < K; q += XWG)
            if (S.nl[q] != S.lvl[q]) { S.lvl[q] = S.nl[q]; mine = 1; }
        changed = __syncthreads_or(mine);
    }
    // counting sort by level (cnt is free now)
    for (int c = tid; c < Ng + 4; c += XWG) S.cnt[c] = 0;
    if (tid == 0) s_nlev = 0;
    __syncthreads();
    {
        uint32_t mx = 0;
        for (int q = tid; q < K; q += XWG) { atomicAdd(&S.cnt[S.lvl[q]], 1u); mx = S.lvl[q] > mx ? S.lvl[q] : mx; }
        atomicMax(&s_nlev, mx);
    }
    __syncthreads();
    const int nlev = (int)s_nlev;
    block_excl_scan(S.cnt, nlev + 1, wsum, tid);  // cnt[l] = pairs in levels < l (1-based l)
    uint32_t* o_off = lv_off + (size_t)blockIdx.x * (K + 2);
    for (int l = tid; l < nlev; l += XWG) o_off[l] = (l + 2 <= nlev) ? S.cnt[l + 2] : (uint32_t)K;
    if (tid == 0) o_off[K + 1] = (uint32_t)nlev;
    __syncthreads();
    uint32_t* o_pairs = lv_pairs + (size_t)blockIdx.x * K;
    double* o_mi = lv_mi + (size_t)blockIdx.x * K;
    for (int q = tid; q < K; q += XWG) {
        const uint32_t pos = atomicAdd(&S.cnt[S.lvl[q]], 1u);
        const uint32_t i = S.pi[q], j = S.pj[q];
        o_pairs[pos] = i | (j << 16);
        o_mi[pos] = P.min_improve_g[i];
    }
}

// k_exch_resolve_lvl_big: level-synchronous walk with values / sources / partners in global memory
// (agent-scope relaxed atomics: the lines are shared between the waves of the workgroup through L2).
__global__ __launch_bounds__(XWG) void k_exch_resolve_lvl_big(const KParams P, const int t, const double* __restrict__ gathered) {
    const int tid = threadIdx.x;
    const int Ng = P.Ng, RW = P.RW, K = P.plan_K;
    const int w = t - P.plan_t0;
    const uint32_t* __restrict__ g_off = P.lv_off + (size_t)w * (K + 2);
    const uint32_t* __restrict__ g_pairs = P.lv_pairs + (size_t)w * K;
    const double* __restrict__ g_mi = P.lv_mi + (size_t)w * K;
    double* val = P.xval;
    int32_t* src = P.xsrc;
    int32_t* partner = P.xpartner;
    const int nlev = (int)g_off[K + 1];
    for (int g = tid; g < Ng; g += XWG) {
        __hip_atomic_store(&val[g], gathered[(size_t)g * RW], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&src[g], g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&partner[g], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    uint32_t b = 0;
    constexpr int BATCH = 8;  // pairs of one level are independent: their loads are issued together
    for (int l = 0; l < nlev; ++l) {
        const uint32_t e = g_off[l];
        for (uint32_t p0 = b + tid; p0 < e; p0 += XWG * BATCH) {
            uint32_t pw[BATCH];
            double m[BATCH], vi[BATCH], vj[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const uint32_t pos = p0 + u * XWG;
                pw[u] = pos < e ? g_pairs[pos] : 0u;
                m[u] = pos < e ? g_mi[pos] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const uint32_t pos = p0 + u * XWG;
                if (pos < e) {
                    vi[u] = __hip_atomic_load(&val[pw[u] & 0xffffu], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    vj[u] = __hip_atomic_load(&val[pw[u] >> 16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else { vi[u] = 0.0; vj[u] = 0.0; }
            }
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const uint32_t pos = p0 + u * XWG;
                const uint32_t i = pw[u] & 0xffffu, j = pw[u] >> 16;
                if (pos < e && vi[u] - vj[u] > m[u]) {          // dist_fun = -, AlgoBGP.jl:688
                    __hip_atomic_store(&val[i], vj[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // swap_ev_ij!, :739-744
                    __hip_atomic_store(&val[j], vi[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int si = __hip_atomic_load(&src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int sj = __hip_atomic_load(&src[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&src[i], sj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&src[j], si, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&partner[i], (int)(j + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // :747-748
                    __hip_atomic_store(&partner[j], (int)(i + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        b = e;
        __syncthreads();
    }
    for (int g = tid; g < Ng; g += XWG) {
        const unsigned s_ = (unsigned)__hip_atomic_load(&src[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned p_ = (unsigned)__hip_atomic_load(&partner[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        P.xres[g] = (unsigned long long)s_ | ((unsigned long long)p_ << 32);
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
                if (vi - vj > P.min_improve_g[i]) {  // dist_fun = -, :688
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
// host side
// ------------------------------------------------------------------------------------------
thread_local std::string g_create_err;

// ------------------------------------------------------------------------------------------
// user objectives: compiled with hiprtc (loaded lazily, the library does not link against it)
// ------------------------------------------------------------------------------------------
struct UserObjective { std::vector<char> code; };
std::vector<UserObjective> g_user_objectives;
std::mutex g_user_mutex;

const char* USER_PRELUDE =
    "#define SMM_USER_OBJECTIVE extern \"C\" __device__ void smm_user_objective\n"
    "extern \"C\" __device__ void smm_user_objective(const double* theta, int np, const double* mom, const double* w, int nm,\n"
    "                                              const double* udata, int n_udata, double* sim_moments, double* value, int* status);\n";
const char* USER_KERNEL =
    "\nextern \"C\" __global__ void smm_user_eval_kernel(const double* theta, int N, int np, const double* mom, const double* w, int nm,\n"
    "        const double* udata, int n_udata, double* simM, double* value, int* status) {\n"
    "    const int c = blockIdx.x * blockDim.x + threadIdx.x;\n"
    "    if (c >= N) return;\n"
    "    int st = 1; double v = 0.0;\n"
    "    smm_user_objective(theta + (size_t)c * np, np, mom, w, nm, udata, n_udata, simM + (size_t)c * nm, &v, &st);\n"
    "    value[c] = v; status[c] = st;\n"
    "}\n";

struct Hiprtc {
    void* lib = nullptr;
    decltype(&hiprtcCreateProgram) create = nullptr;
    decltype(&hiprtcCompileProgram) compile = nullptr;
    decltype(&hiprtcGetProgramLogSize) log_size = nullptr;
    decltype(&hiprtcGetProgramLog) log = nullptr;
    decltype(&hiprtcGetCodeSize) code_size = nullptr;
    decltype(&hiprtcGetCode) code = nullptr;
    decltype(&hiprtcDestroyProgram) destroy = nullptr;
    bool load(std::string& err) {
        if (lib) return true;
        lib = dlopen("libhiprtc.so", RTLD_NOW | RTLD_LOCAL);
        if (!lib) lib = dlopen("/opt/rocm/lib/libhiprtc.so", RTLD_NOW | RTLD_LOCAL);
        if (!lib) { err = std::string("cannot load libhiprtc.so: ") + dlerror(); return false; }
#define RTC_SYM(f, name) f = (decltype(f))dlsym(lib, name); if (!f) { err = std::string("libhiprtc.so lacks ") + name; return false; }
        RTC_SYM(create, "hiprtcCreateProgram") RTC_SYM(compile, "hiprtcCompileProgram") RTC_SYM(log_size, "hiprtcGetProgramLogSize")
        RTC_SYM(log, "hiprtcGetProgramLog") RTC_SYM(code_size, "hiprtcGetCodeSize") RTC_SYM(code, "hiprtcGetCode")
        RTC_SYM(destroy, "hiprtcDestroyProgram")
#undef RTC_SYM
        return true;
    }
};
Hiprtc g_rtc;

struct Ctx {
    KParams P{};
    int obj = 0, device = 0, exchange_from = 2;
    int iter = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<void*> allocs;
    std::string err;
    smm_timing_t timing{};
    bool pending_timing = false;
    int profiling = 0;   // 1: event brackets around the kernels; 2: the kernels' own begin/end timestamps (hipExtLaunchKernelGGL)
    hipEvent_t kev0 = nullptr, kev1 = nullptr;   // mode 2: start/stop events of the next launch
    std::vector<char> pev_exch;
    bool force_any_exchange = false;
    std::vector<hipEvent_t> pev;  // profiling events: 4 per iteration (the last two bracket nothing: the event overhead)
    int pev_iters = 0;
    // double-buffered last-accepted records [N][RW]
    double* rec[2] = {nullptr, nullptr};
    int cur = 0;               // rec[cur] holds the records after the last accept step
    bool pending = false;      // exchange of iteration `iter` resolved but not applied
    bool prev_open = false;    // accept-rate counters of iteration `iter` not closed yet
    // look-ahead windows
    int win_cap = 0;           // iterations per window
    double* win_rb = nullptr;
    unsigned long long* win_plan = nullptr;
    double* win_plan_mi = nullptr;
    uint32_t *win_lv_pairs = nullptr, *win_lv_off = nullptr;
    double* win_lv_mi = nullptr;
    bool lvl_exchange = false;
    bool lvl_soa_exchange = false;   // XLVL_MAX < N_global <= XLDS_MAX: level walk on split chain slots
    int tpw = 1;                // tiles per workgroup of k_chain_iter (2 with the inline walk: one walk per CU)
    bool inline_walk = false;   // the exchange walk runs in the prologue of the next k_chain_iter (SMMHIP_INLINE_WALK=0: off)
    const double* ext_rec_in = nullptr;   // sharded_step: donor records come from / results go to the caller's gather buffers
    double* ext_rec_out = nullptr;
    bool pending_ext = false;   // sharded_step: the exchange of iteration `iter` is still to be resolved from the gathered records
    int n_objp = 0;                 // doubles in P.objp
    hipModule_t umod = nullptr;     // user objective: this context's module and kernel
    hipFunction_t ufn = nullptr;
    bool rec_external = false;  // the records after the last accept step were written to the caller's gather buffer (sharded_step)
    bool unresolved = false;    // exchangeMoves! of iteration `iter` is still to be resolved (inline, or by resolve_now)
    bool big_exchange = false;   // 8192 < N_global <= 65535: level plan and walk in global memory
    uint32_t* big_scratch = nullptr;
    int lvl_wg = 1024;
    int rng_t0 = 0, rng_w = 0;    // window currently held: iterations [t0, t0+w)
    int plan_t0 = 0, plan_w = 0;
    bool lds_exchange = false;
    int ct = 8;
};

#define HIPCHK(call)                                                                                  \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) {                                                                       \
            char b_[512];                                                                             \
            snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            throw std::string(b_);                                                                    \
        }                                                                                             \
    } while (0)

template <class T>
T* dalloc(Ctx* c, size_t n) {
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, (n ? n : 1) * sizeof(T)));
    c->allocs.push_back(p);
    return (T*)p;
}
template <class T>
T* dupload(Ctx* c, const T* h, size_t n) {
    T* d = dalloc<T>(c, n);
    if (h && n) HIPCHK(hipMemcpy(d, h, n * sizeof(T), hipMemcpyHostToDevice));
    return d;
}

bool is_sim(int obj) { return obj == SMM_OBJ_NORM || obj == SMM_OBJ_NORM_FAILBOX; }
int obj_kind(int obj) { return is_sim(obj) ? 1 : obj == SMM_OBJ_DENSE ? 2 : 0; }

size_t tile_smem_base(const Ctx* c, int ct) {
    const KParams& P = c->P;
    return tile_smem_doubles(ct, P.np, P.nm, P.RW, P.HW, P.RBW, obj_kind(c->obj)) * sizeof(double);
}
size_t tile_smem(const Ctx* c, int ct, int tpw = 1) {   // dynamic LDS of k_chain_iter: tpw tiles; with the inline exchange
    const size_t base = tile_smem_base(c, ct);           // walk its chain slots in front and its pair list under the tiles
    const size_t tiles = (size_t)tpw * ((base + 15) & ~(size_t)15);
    return c->inline_walk ? walk_slot_bytes(c->P.Ng) + std::max(tiles, (size_t)c->P.plan_K * 4) : tiles;
}
size_t plan_lds_bytes(int Ng, int K) { return (size_t)(Ng + 2) * 4 + (size_t)K * 8 + (size_t)K * 4 + 128 + 16; }
size_t resolve_lds_bytes(int Ng) { return (size_t)Ng * 16 + 16; }
size_t resolve_lvl_soa_bytes(int Ng, int K) { return (size_t)Ng * 12 + (size_t)K * 4 + 16; }
size_t resolve_lvl_bytes(int Ng, int K) { return (size_t)Ng * 16 + (size_t)K * 12 + 64 * 8 + 64; }

int exchange_K(const Ctx* c) { return c->P.pairtab ? c->P.n_pairs_tab : n_exchange_pairs(c->P.Ng); }
bool exchange_active(const Ctx* c, int t) { return t >= c->exchange_from && c->P.Ng > 1; }  // AlgoBGP.jl:637

// make the look-ahead tables cover iteration t (1-based): a new window simply starts at t
void ensure_windows(Ctx* c, int t) {
    KParams& P = c->P;
    if (!(t >= c->rng_t0 && t < c->rng_t0 + c->rng_w)) {
        const int W = std::min(c->win_cap, P.T - t + 1);
        const size_t Q = (size_t)(P.np + 1) / 2;
        const size_t total = (size_t)W * P.rb_tries * Q * P.N;
        hipLaunchKernelGGL(k_pregen_rng, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, P, t, W, c->win_rb);
        c->rng_t0 = t; c->rng_w = W;
        P.rb = c->win_rb; P.rb_t0 = t;
    }
    if (c->big_exchange && !(t >= c->plan_t0 && t < c->plan_t0 + c->plan_w)) {
        const int W = std::min(c->win_cap, P.T - t + 1);
        hipLaunchKernelGGL(k_exch_plan_big, dim3(W), dim3(XWG), 0, c->stream, P, t, c->big_scratch, c->win_lv_pairs, c->win_lv_mi,
                           c->win_lv_off);
        c->plan_t0 = t; c->plan_w = W;
        P.plan_t0 = t;
        P.lv_pairs = c->win_lv_pairs; P.lv_mi = c->win_lv_mi; P.lv_off = c->win_lv_off;
    }
    if (c->lds_exchange && P.Ng > 1 && !(t >= c->plan_t0 && t < c->plan_t0 + c->plan_w)) {
        const int W = std::min(c->win_cap, P.T - t + 1);
        hipLaunchKernelGGL(k_exch_plan, dim3(W), dim3(XWG), plan_lds_bytes(P.Ng, P.plan_K), c->stream, P, t, c->win_plan,
                           c->win_plan_mi, c->win_lv_pairs, c->win_lv_mi, c->win_lv_off);
        c->plan_t0 = t; c->plan_w = W;
        P.plan = c->win_plan; P.plan_mi = c->win_plan_mi; P.plan_t0 = t;
        P.lv_pairs = c->win_lv_pairs; P.lv_mi = c->win_lv_mi; P.lv_off = c->win_lv_off;
    }
}

template <int KIND, int CT, int TPW = 1>
void launch_chain_iter_ct(Ctx* c, int t, int flags) {
    const KParams& P = c->P;
    const int tiles = (P.N + CT - 1) / CT;
    const dim3 grid((tiles + TPW - 1) / TPW), block(WG * TPW);
    const double* rin = c->ext_rec_in ? c->ext_rec_in : (const double*)c->rec[c->cur];
    double* rout = c->ext_rec_out ? c->ext_rec_out : c->rec[c->cur ^ 1];
    if (c->kev0)   // profiling mode 2: begin/end of this dispatch as the command processor stamps them
        hipExtLaunchKernelGGL((k_chain_iter<KIND, CT, TPW>), grid, block, tile_smem(c, CT, TPW), c->stream, c->kev0, c->kev1, 0, P, t,
                              rin, rout, flags);
    else
        hipLaunchKernelGGL((k_chain_iter<KIND, CT, TPW>), grid, block, tile_smem(c, CT, TPW), c->stream, P, t, rin, rout, flags);
}

// one thread per evaluation: theta [n][np] -> simM [n][nm], value [n], status [n]
void launch_user_kernel(Ctx* c, const double* theta, int n, double* simM, double* value, int* status) {
    const KParams& P = c->P;
    int np = P.np, nm = P.nm, nud = c->n_objp;
    const double *mom = P.mom, *w = P.w, *ud = P.objp;
    void* args[] = {(void*)&theta, (void*)&n, (void*)&np, (void*)&mom, (void*)&w, (void*)&nm, (void*)&ud, (void*)&nud,
                    (void*)&simM, (void*)&value, (void*)&status};
    HIPCHK(hipModuleLaunchKernel(c->ufn, (unsigned)((n + 127) / 128), 1, 1, 128, 1, 1, 0, c->stream, args, nullptr));
}

void launch_chain_iter(Ctx* c, int t, int flags) {
    if (c->obj == SMM_OBJ_USER) {
        // proposal launch (stores nothing but the proposals) -> the user's kernel -> accept launch (repeats the
        // deterministic prologue, takes value / moments / status from the user's kernel)
        const KParams& P = c->P;
        const bool big = tile_smem_base(c, 64) <= (size_t)60 * 1024;
        if (big) launch_chain_iter_ct<0, 64>(c, t, flags | F_PROPOSE_ONLY); else launch_chain_iter_ct<0, 8>(c, t, flags | F_PROPOSE_ONLY);
        launch_user_kernel(c, P.u_theta, P.N, P.u_simM, P.u_value, P.u_status);
        if (big) launch_chain_iter_ct<0, 64>(c, t, flags); else launch_chain_iter_ct<0, 8>(c, t, flags);
        if (!c->ext_rec_out) c->cur ^= 1;
        return;
    }
    if (is_sim(c->obj)) {
        if (c->ct == 4) launch_chain_iter_ct<1, 4>(c, t, flags);
        else if (c->ct == 16) launch_chain_iter_ct<1, 16>(c, t, flags);
        else if (c->tpw == 2) launch_chain_iter_ct<1, 8, 2>(c, t, flags);
        else launch_chain_iter_ct<1, 8>(c, t, flags);
    } else if (c->obj == SMM_OBJ_DENSE) {
        launch_chain_iter_ct<2, 16>(c, t, flags);
    } else if (tile_smem_base(c, 64) <= (size_t)60 * 1024 && !c->inline_walk) {
        launch_chain_iter_ct<0, 64>(c, t, flags);
    } else {
        launch_chain_iter_ct<0, 8>(c, t, flags);
    }
    if (!c->ext_rec_out) c->cur ^= 1;
}

void launch_resolve(Ctx* c, int t, const double* gathered) {
    const KParams& P = c->P;
    if (c->lvl_exchange)
        if (c->lvl_wg == 256)
            hipLaunchKernelGGL(k_exch_resolve_lvl<256>, dim3(1), dim3(256), resolve_lvl_bytes(P.Ng, P.plan_K), c->stream, P, t, gathered);
        else if (c->lvl_wg == 512)
            hipLaunchKernelGGL(k_exch_resolve_lvl<512>, dim3(1), dim3(512), resolve_lvl_bytes(P.Ng, P.plan_K), c->stream, P, t, gathered);
        else
            if (c->kev0)
                hipExtLaunchKernelGGL(k_exch_resolve_lvl<1024>, dim3(1), dim3(1024), resolve_lvl_bytes(P.Ng, P.plan_K), c->stream, c->kev0,
                                      c->kev1, 0, P, t, gathered);
            else
                hipLaunchKernelGGL(k_exch_resolve_lvl<1024>, dim3(1), dim3(1024), resolve_lvl_bytes(P.Ng, P.plan_K), c->stream, P, t, gathered);
    else if (c->lvl_soa_exchange)
        hipLaunchKernelGGL(k_exch_resolve_lvl_soa<1024>, dim3(1), dim3(1024), resolve_lvl_soa_bytes(P.Ng, P.plan_K), c->stream, P, t, gathered);
    else if (c->lds_exchange)
        hipLaunchKernelGGL(k_exch_resolve_lds, dim3(1), dim3(XWG), resolve_lds_bytes(P.Ng), c->stream, P, t, gathered);
    else if (c->big_exchange)
        hipLaunchKernelGGL(k_exch_resolve_lvl_big, dim3(1), dim3(XWG), 0, c->stream, P, t, gathered);
    else
        hipLaunchKernelGGL(k_exch_resolve_any, dim3(1), dim3(XWG), 0, c->stream, P, t, gathered);
}

// settle the open end of the last iteration (no-op when nothing is open)
// the exchange of iteration c->iter has been left to the next chain kernel, but something else needs it now
void resolve_now(Ctx* c) {
    if (!c->unresolved) return;
    launch_resolve(c, c->iter, nullptr);
    c->unresolved = false;
}

void flush(Ctx* c) {
    if (c->rec_external) throw std::string("records are in the gather buffer: call smm_bgp_sharded_finish first");
    if (!c->pending && !c->prev_open) return;
    resolve_now(c);
    const KParams& P = c->P;
    const int flags = (c->prev_open ? F_CLOSE_PREV : 0) | (c->pending ? F_HAS_PENDING : 0);
    hipLaunchKernelGGL(k_flush, dim3((P.N + 255) / 256), dim3(256), 0, c->stream, P, c->iter + 1, (const double*)c->rec[c->cur],
                       c->rec[c->cur ^ 1], flags);
    c->cur ^= 1;
    c->pending = false;
    c->prev_open = false;
}

int check_device_error(Ctx* c) {
    unsigned long long e = ERR_NONE;
    HIPCHK(hipMemcpy(&e, c->P.err, sizeof e, hipMemcpyDeviceToHost));
    if (e == ERR_NONE) return SMM_OK;
    const int kind = (int)(e & 3), chain = (int)((e >> 2) & 0xffffffffu), it = (int)(e >> 34);
    char b[256];
    if (kind == 3) {
        snprintf(b, sizeof b, "internal error: exchange resolution did not converge (iteration %d)", it);
        c->err = b;
        return SMM_ERR_HIP;
    }
    if (kind == 1) {
        snprintf(b, sizeof b, "AlgoBGP assumes that your objective function returns a non-negative number "
                 "(chain %d, iteration %d)", chain + 1, it);
        c->err = b;
        return SMM_ERR_NEGATIVE_OBJECTIVE;
    }
    snprintf(b, sizeof b, "no draw in support after %d trials (chain %d, iteration %d): increase smpl_iters",
             c->P.user_n ? std::min(c->P.rb_tries, c->P.smpl_iters) : c->P.smpl_iters, chain + 1, it);
    c->err = b;
    return SMM_ERR_NO_DRAW_IN_SUPPORT;
}

int fail(Ctx* c, int code, const std::string& m) {
    if (c) c->err = m;
    else g_create_err = m;
    return code;
}

}  // namespace

extern "C" {

int smm_abi_version(void) { return SMMHIP_ABI_VERSION; }

int smm_register_user_objective(const char* hip_source, int32_t* objective_id_out) {
    if (!hip_source || !objective_id_out) { g_create_err = "smm_register_user_objective: null argument"; return SMM_ERR_INVALID_ARG; }
    std::lock_guard<std::mutex> lock(g_user_mutex);
    std::string err;
    if (!g_rtc.load(err)) { g_create_err = err; return SMM_ERR_HIP; }
    const std::string src = std::string(USER_PRELUDE) + hip_source + USER_KERNEL;
    hiprtcProgram prog = nullptr;
    if (g_rtc.create(&prog, src.c_str(), "smm_user_objective.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
        g_create_err = "hiprtcCreateProgram failed";
        return SMM_ERR_HIP;
    }
    const char* opts[] = {"--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17"};
    const hiprtcResult rc = g_rtc.compile(prog, 4, opts);
    if (rc != HIPRTC_SUCCESS) {
        size_t n = 0;
        g_rtc.log_size(prog, &n);
        std::string log(n, ' ');
        if (n) g_rtc.log(prog, &log[0]);
        g_create_err = "user objective does not compile:\n" + log;
        g_rtc.destroy(&prog);
        return SMM_ERR_INVALID_ARG;
    }
    size_t cs = 0;
    g_rtc.code_size(prog, &cs);
    UserObjective u;
    u.code.resize(cs);
    g_rtc.code(prog, u.code.data());
    g_rtc.destroy(&prog);
    g_user_objectives.push_back(std::move(u));
    *objective_id_out = SMM_OBJ_USER_BASE + (int32_t)g_user_objectives.size() - 1;
    return SMM_OK;
}

int smm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* smm_last_error(void* ctx) { return ctx ? ((Ctx*)ctx)->err.c_str() : g_create_err.c_str(); }

void smm_ctx_destroy(void* ctx) {
    Ctx* c = (Ctx*)ctx;
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (void* p : c->allocs) (void)hipFree(p);
    if (c->umod) (void)hipModuleUnload(c->umod);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    for (hipEvent_t e : c->pev) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int smm_ctx_create(const smm_problem_t* prob, const smm_bgp_opts_t* opts, const smm_tables_t* tab, void** out) {
    if (!prob || !opts || !out) return fail(nullptr, SMM_ERR_INVALID_ARG, "null argument");
    const int np = prob->np, nm = prob->nm, ns = prob->ns, N = opts->N, T = opts->maxiter, Ng = opts->N_global;
    if (np < 1 || nm < 1 || ns < 1 || np > MAX_DIM || nm > MAX_DIM)
        return fail(nullptr, SMM_ERR_INVALID_ARG, "need 1 <= np,nm <= 64 and ns >= 1");
    if (N < 1 || T < 1 || Ng < N || opts->chain_offset < 0 || opts->chain_offset + N > Ng || (Ng % N) != 0)
        return fail(nullptr, SMM_ERR_INVALID_ARG, "bad N / N_global / chain_offset / maxiter");
    const bool user_obj = prob->objective_id >= SMM_OBJ_USER_BASE;
    if (user_obj) {
        std::lock_guard<std::mutex> lock(g_user_mutex);
        if (prob->objective_id - SMM_OBJ_USER_BASE >= (int)g_user_objectives.size())
            return fail(nullptr, SMM_ERR_INVALID_ARG, "unknown user objective handle");
    } else if (prob->objective_id < 0 || prob->objective_id > SMM_OBJ_DENSE)
        return fail(nullptr, SMM_ERR_INVALID_ARG, "unknown objective_id");
    if (is_sim(prob->objective_id) && np != nm)
        return fail(nullptr, SMM_ERR_INVALID_ARG, "objfunc_norm needs one moment per parameter (ObjExamples.jl:66-78)");
    if (opts->batch_size < 1 || opts->batch_size > np || (np % opts->batch_size) != 0)
        return fail(nullptr, SMM_ERR_BAD_BATCH, "batch_size must divide the number of parameters (AlgoBGP.jl:95-103)");
    if (opts->sigma_update_steps < 1) return fail(nullptr, SMM_ERR_INVALID_ARG, "sigma_update_steps < 1");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(nullptr, SMM_ERR_NO_DEVICE, "no HIP device available: libsmmhip has no CPU fallback");
    if (opts->device < 0 || opts->device >= ndev) return fail(nullptr, SMM_ERR_INVALID_ARG, "bad device ordinal");
    Ctx* c = new Ctx();
    try {
        c->device = opts->device;
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        HIPCHK(hipEventCreate(&c->ev0));
        HIPCHK(hipEventCreate(&c->ev1));
        KParams& P = c->P;
        c->obj = user_obj ? SMM_OBJ_USER : prob->objective_id;
        c->exchange_from = opts->exchange_from_iter;
        {
            const char* e = getenv("SMMHIP_ANY_EXCHANGE");  // test hook: force the any-size resolution kernel
            c->force_any_exchange = e && e[0] == '1';
            const char* d = getenv("SMMHIP_DBG");
            P.dbg = d ? atoi(d) : 0;
            const char* tsv = getenv("SMMHIP_TS");
            if (tsv && tsv[0] == '1') P.ts = dalloc<unsigned long long>(c, (size_t)8 * 65536);
            const char* ct = getenv("SMMHIP_CT");  // tuning hook: chains per tile (4, 8, 16); numerics unaffected
            c->ct = ct ? atoi(ct) : 8;
        }
        P.np = np; P.nm = nm; P.ns = ns; P.obj = c->obj;
        P.init = dupload(c, prob->init, np); P.lb = dupload(c, prob->lb, np); P.ub = dupload(c, prob->ub, np);
        P.mom = dupload(c, prob->mom, nm); P.w = dupload(c, prob->w, nm);
        P.objp = prob->n_obj_params > 0 ? dupload(c, prob->obj_params, prob->n_obj_params) : nullptr;
        c->n_objp = prob->n_obj_params > 0 ? prob->n_obj_params : 0;
        if (user_obj) {
            {
                std::lock_guard<std::mutex> lock(g_user_mutex);
                const UserObjective& u = g_user_objectives[prob->objective_id - SMM_OBJ_USER_BASE];
                HIPCHK(hipModuleLoadData(&c->umod, u.code.data()));
            }
            HIPCHK(hipModuleGetFunction(&c->ufn, c->umod, "smm_user_eval_kernel"));
            P.u_theta = dalloc<double>(c, (size_t)N * np);
            P.u_simM = dalloc<double>(c, (size_t)N * nm);
            P.u_value = dalloc<double>(c, N);
            P.u_status = dalloc<int>(c, N);
        }
        if (prob->objective_id == SMM_OBJ_DENSE) {
            const size_t nB = (size_t)DENSE_D * np, nA = (size_t)nm * DENSE_D;
            if (prob->n_obj_params != 0 && (size_t)prob->n_obj_params != nB + nA)
                throw std::string("SMM_OBJ_DENSE: obj_params must hold B (256 x np) and A (nm x 256), or be empty");
            std::vector<double> M(nB + nA);
            if (prob->n_obj_params) memcpy(M.data(), prob->obj_params, M.size() * 8);
            else {  // N(0,1)/sqrt(fan-in) from the counter RNG, stream 5
                for (size_t i = 0; i < nB + nA; i += 2) {
                    double z0, z1;
                    box_muller(philox_stream(opts->seed, 5, (uint32_t)(i >> 1), (uint32_t)((i >> 1) >> 32), 0, 0), z0, z1);
                    M[i] = z0 / sqrt(i < nB ? (double)np : (double)DENSE_D);
                    if (i + 1 < nB + nA) M[i + 1] = z1 / sqrt(i + 1 < nB ? (double)np : (double)DENSE_D);
                }
            }
            const int nPs = (np + 3) / 4, nOt = (nm + 15) / 16;
            std::vector<double> Bf((size_t)(DENSE_D / 16) * nPs * 64, 0.0), Af((size_t)nOt * (DENSE_D / 16) * 4 * 64, 0.0);
            for (int T = 0; T < DENSE_D / 16; ++T)
                for (int s = 0; s < nPs; ++s)
                    for (int l = 0; l < 64; ++l) {
                        const int d = 16 * T + (l & 15), p = 4 * s + (l >> 4);
                        if (p < np) Bf[((size_t)T * nPs + s) * 64 + l] = M[(size_t)d * np + p];
                    }
            for (int o = 0; o < nOt; ++o)
                for (int T = 0; T < DENSE_D / 16; ++T)
                    for (int s = 0; s < 4; ++s)
                        for (int l = 0; l < 64; ++l) {
                            const int k = 16 * o + (l & 15), d = 16 * T + 4 * s + (l >> 4);
                            if (k < nm) Af[(((size_t)o * (DENSE_D / 16) + T) * 4 + s) * 64 + l] = M[nB + (size_t)k * DENSE_D + d];
                        }
            P.dense_Bf = dupload(c, Bf.data(), Bf.size());
            P.dense_Af = dupload(c, Af.data(), Af.size());
            P.dense_nOt = nOt;
        }
        {
            const int rows = (ns + WG - 1) / WG;
            P.zstride = ((rows + ZU) / ZU) * ZU * WG;  // at least one chunk beyond the last full one
            std::vector<double> Z((size_t)nm * P.zstride, 0.0);
            for (int k = 0; k < nm; ++k)
                for (int s = 0; s < ns; ++s)
                    Z[(size_t)k * P.zstride + s] = (tab && tab->Z) ? tab->Z[(size_t)k * ns + s] : rng_Z(opts->seed, (uint32_t)k, (uint32_t)s);
            P.Z = dupload(c, Z.data(), Z.size());
        }
        P.N = N; P.Ng = Ng; P.offset = opts->chain_offset; P.T = T;
        P.sigma_update_steps = opts->sigma_update_steps; P.smpl_iters = opts->smpl_iters;
        P.batch_size = opts->batch_size; P.sigma_adjust_by = opts->sigma_adjust_by; P.seed = opts->seed;
        P.min_improve_g = dupload(c, opts->min_improve, Ng);
        P.mi_uniform = 1; P.mi_value = opts->min_improve[0];
        for (int i = 1; i < Ng; ++i)
            if (!(opts->min_improve[i] == P.mi_value)) P.mi_uniform = 0;
        const size_t TN = (size_t)T * N;
        if (tab && tab->probs_acc) P.user_utab = dupload(c, tab->probs_acc, TN);
        if (tab && tab->prop_normals && tab->prop_tries > 0) {
            P.rb_tries = tab->prop_tries; P.user_n = 1;
            P.user_ntab = dupload(c, tab->prop_normals, TN * (size_t)tab->prop_tries * np);
        } else {
            P.rb_tries = np <= 8 ? 8 : (np <= 32 ? 4 : 2);
        }
        if (tab && tab->pairs && tab->n_pairs > 0) {
            P.n_pairs_tab = tab->n_pairs;
            P.pairtab = dupload(c, tab->pairs, (size_t)T * tab->n_pairs * 2);
        }
        P.RW = even_up(3 + np + nm);
        P.HW = even_up(H_PARAMS + np + nm);
        P.RBW = even_up(1 + P.rb_tries * np);
        const int K = exchange_K(c);
        P.plan_K = K;
        c->lds_exchange = Ng > 1 && Ng <= XLDS_MAX && K >= 1 && K <= Ng && !c->force_any_exchange;
        {
            const char* e = getenv("SMMHIP_DATAFLOW_EXCHANGE");  // test hook: force the ticket (data-flow) resolution kernel
            c->lvl_exchange = c->lds_exchange && Ng <= XLVL_MAX && !(e && e[0] == '1');
            c->lvl_soa_exchange = c->lds_exchange && !c->lvl_exchange && !(e && e[0] == '1');
            const char* lw = getenv("SMMHIP_LVL_WG");  // tuning hook
            if (lw) c->lvl_wg = atoi(lw);
            const char* be = getenv("SMMHIP_BIG_EXCHANGE");  // test hook: force the global-memory level kernels
            const bool force_big = be && be[0] == '1';
            c->big_exchange = Ng > 1 && Ng <= 65535 && K >= 1 && K <= Ng && !c->force_any_exchange && (force_big || !c->lds_exchange);
            if (c->big_exchange) { c->lds_exchange = false; c->lvl_exchange = false; c->lvl_soa_exchange = false; }
            // inline exchange walk: single shard, level plan available, and two tiles must still share a CU's 160 KB LDS
            const char* iw = getenv("SMMHIP_INLINE_WALK");
            const int tile_ct = is_sim(c->obj) ? c->ct : (c->obj == SMM_OBJ_DENSE ? 16 : 8);
            const size_t tile_b = (tile_smem_base(c, tile_ct) + 15) & ~(size_t)15;
            c->inline_walk = !(iw && iw[0] == '0') && c->lvl_exchange && N == Ng && c->obj != SMM_OBJ_USER &&
                             walk_slot_bytes(Ng) + std::max(tile_b, (size_t)K * 4) <= (size_t)80 * 1024;
            P.tile_off = c->inline_walk ? (int)(walk_slot_bytes(Ng) / sizeof(double)) : 0;
            // two tiles per workgroup share one walk (the 2p/2m-style simulation tile of 8 chains only)
            const char* tp = getenv("SMMHIP_TPW");
            c->tpw = (c->inline_walk && is_sim(c->obj) && c->ct == 8 && N > 8 && !(tp && tp[0] == '1') &&
                      walk_slot_bytes(Ng) + std::max(2 * tile_b, (size_t)K * 4) <= (size_t)160 * 1024) ? 2 : 1;
        }
        {   // look-ahead window: as many iterations as ~192 MiB of tables allow, at most 256
            const size_t per_iter = (size_t)P.RBW * N * 8 + (size_t)K * 36 + (c->big_exchange ? BigPlanScratch::words(Ng, K) * 4 : 0);
            c->win_cap = (int)std::max<size_t>(1, std::min<size_t>(256, ((size_t)192 << 20) / per_iter));
            c->win_cap = std::min(c->win_cap, T);
            c->win_rb = dalloc<double>(c, (size_t)c->win_cap * N * P.RBW);
            HIPCHK(hipMemset(c->win_rb, 0, (size_t)c->win_cap * N * P.RBW * 8));
            if (c->big_exchange) {
                c->win_lv_pairs = dalloc<uint32_t>(c, (size_t)c->win_cap * K);
                c->win_lv_mi = dalloc<double>(c, (size_t)c->win_cap * K);
                c->win_lv_off = dalloc<uint32_t>(c, (size_t)c->win_cap * (K + 2));
                c->big_scratch = dalloc<uint32_t>(c, (size_t)c->win_cap * BigPlanScratch::words(Ng, K));
            }
            if (c->lds_exchange) {
                c->win_plan = dalloc<unsigned long long>(c, (size_t)c->win_cap * K);
                c->win_plan_mi = dalloc<double>(c, (size_t)c->win_cap * K);
                c->win_lv_pairs = dalloc<uint32_t>(c, (size_t)c->win_cap * K);
                c->win_lv_mi = dalloc<double>(c, (size_t)c->win_cap * K);
                c->win_lv_off = dalloc<uint32_t>(c, (size_t)c->win_cap * (K + 2));
            }
        }
        {   // chain state blocks and records (BGPChain ctor, AlgoBGP.jl:78-109: best = Inf, best_id = -1, ...)
            std::vector<double> cs((size_t)N * CSW, 0.0);
            for (int i = 0; i < N; ++i) {
                double* b = cs.data() + (size_t)i * CSW;
                b[CS_SIGMA] = opts->sigma[opts->chain_offset + i];
                b[CS_BEST] = INFINITY; b[CS_BESTID] = -1.0; b[CS_BESTP] = INFINITY; b[CS_BESTPID] = -1.0;
                b[CS_ATUN] = opts->acc_tuner[opts->chain_offset + i];
            }
            P.cs = dupload(c, cs.data(), cs.size());
            std::vector<double> rec((size_t)N * P.RW, 0.0);
            for (int i = 0; i < N; ++i) rec[(size_t)i * P.RW] = INFINITY;  // value: Inf until the first accept
            for (int b = 0; b < 2; ++b) c->rec[b] = dupload(c, rec.data(), rec.size());
        }
        P.xres = dalloc<unsigned long long>(c, Ng);
        P.vals = dalloc<double>(c, N);
        if (!c->lds_exchange) {
            const int Kmax = std::max(K, 1);
            P.xval = dalloc<double>(c, Ng); P.xnext = dalloc<int32_t>(c, Ng); P.xpairs = dalloc<int32_t>(c, (size_t)Kmax * 2);
            P.xsrc = dalloc<int32_t>(c, Ng); P.xpartner = dalloc<int32_t>(c, Ng);
        }
        {   // history: NaN values, curr/best = Inf, best_id = -1, exchanged = accepted = status = 0
            std::vector<double> row((size_t)N * P.HW, NAN);
            for (int i = 0; i < N; ++i) {
                double* h = row.data() + (size_t)i * P.HW;
                h[H_CURR] = INFINITY; h[H_BEST] = INFINITY; h[H_BESTID] = -1.0; h[H_EXCH] = 0.0; h[H_ACC] = 0.0; h[H_STATUS] = 0.0;
            }
            P.hrec = dalloc<double>(c, TN * P.HW);
            for (int t = 0; t < T; ++t)
                HIPCHK(hipMemcpy(P.hrec + (size_t)t * N * P.HW, row.data(), row.size() * 8, hipMemcpyHostToDevice));
        }
        P.err = dalloc<unsigned long long>(c, 1);
        {
            const unsigned long long e = ERR_NONE;
            HIPCHK(hipMemcpy(P.err, &e, 8, hipMemcpyHostToDevice));
        }
        if (c->lds_exchange) {
            HIPCHK(hipFuncSetAttribute((const void*)k_exch_resolve_lds, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)resolve_lds_bytes(XLDS_MAX)));
            HIPCHK(hipFuncSetAttribute((const void*)k_exch_plan, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)plan_lds_bytes(XLDS_MAX, XLDS_MAX)));
            HIPCHK(hipFuncSetAttribute((const void*)k_exch_resolve_lvl<256>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)resolve_lvl_bytes(XLVL_MAX, XLVL_MAX)));
            HIPCHK(hipFuncSetAttribute((const void*)k_exch_resolve_lvl<512>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)resolve_lvl_bytes(XLVL_MAX, XLVL_MAX)));
            HIPCHK(hipFuncSetAttribute((const void*)k_exch_resolve_lvl_soa<1024>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)resolve_lvl_soa_bytes(XLDS_MAX, XLDS_MAX)));
            HIPCHK(hipFuncSetAttribute((const void*)k_exch_resolve_lvl<1024>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)resolve_lvl_bytes(XLVL_MAX, XLVL_MAX)));
        }
        {   // tiles of problems with many parameters need more than the default 64 KiB of dynamic LDS
            const int lim = 160 * 1024;
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter<1, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter<1, 8, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter<1, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter<2, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter<0, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_eval_batch<1, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_eval_batch<2, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_eval_batch<0, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            if (tile_smem(c, is_sim(c->obj) ? c->ct : (c->obj == SMM_OBJ_DENSE ? 16 : 8)) > (size_t)lim)
                throw std::string("tile does not fit the 160 KiB LDS");
        }
        HIPCHK(hipDeviceSynchronize());
    } catch (const std::string& m) {
        g_create_err = m;
        smm_ctx_destroy(c);
        return SMM_ERR_HIP;
    }
    *out = c;
    return SMM_OK;
}

void* smm_stream(void* ctx) { return ctx ? (void*)((Ctx*)ctx)->stream : nullptr; }

int smm_sync(void* ctx) {
    Ctx* c = (Ctx*)ctx;
    if (!c) return SMM_ERR_INVALID_ARG;
    try {
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (c->pending_timing) {
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
            c->timing.step_ms = ms;
            c->pending_timing = false;
            c->timing.iter_kernel_ms = 0.0;
            c->timing.exch_kernel_ms = 0.0;
            c->timing.null_bracket_ms = 0.0;
            for (int i = 0; i < c->pev_iters; ++i) {
                float a = 0.f, b = 0.f, n = 0.f;
                HIPCHK(hipEventElapsedTime(&a, c->pev[4 * i], c->pev[4 * i + 1]));
                if (c->profiling == 2) {
                    if (c->pev_exch[i]) HIPCHK(hipEventElapsedTime(&b, c->pev[4 * i + 2], c->pev[4 * i + 3]));
                } else {
                    HIPCHK(hipEventElapsedTime(&b, c->pev[4 * i + 1], c->pev[4 * i + 2]));
                    HIPCHK(hipEventElapsedTime(&n, c->pev[4 * i + 2], c->pev[4 * i + 3]));
                }
                c->timing.iter_kernel_ms += a;
                c->timing.exch_kernel_ms += b;
                c->timing.null_bracket_ms += n;
            }
            c->pev_iters = 0;
        }
        return check_device_error(c);
    } catch (const std::string& m) {
        return fail(c, SMM_ERR_HIP, m);
    }
}

int smm_bgp_step_async(void* ctx, int32_t n_iters) {
    Ctx* c = (Ctx*)ctx;
    if (!c || n_iters < 0) return SMM_ERR_INVALID_ARG;
    if (c->P.N != c->P.Ng) return fail(c, SMM_ERR_STATE, "smm_bgp_step needs a single shard (N == N_global); use the sharded calls");
    if (c->iter + n_iters > c->P.T) return fail(c, SMM_ERR_MAXITER, "step beyond maxiter (history capacity)");
    try {
        HIPCHK(hipSetDevice(c->device));
        if (c->profiling) {
            while ((int)c->pev.size() < 4 * n_iters) {
                hipEvent_t e;
                HIPCHK(hipEventCreate(&e));
                c->pev.push_back(e);
            }
            c->pev_exch.assign((size_t)n_iters, 0);
        }
        c->pev_iters = 0;
        HIPCHK(hipEventRecord(c->ev0, c->stream));
        for (int it = 0; it < n_iters; ++it) {
            const int t = c->iter + 1;
            // an exchange left to this chain kernel needs its plan: resolve it now if the plan window is about to move on
            if (c->unresolved && !(t >= c->plan_t0 && t < c->plan_t0 + c->plan_w)) resolve_now(c);
            ensure_windows(c, t);
            const int flags = (c->prev_open ? F_CLOSE_PREV : 0) | (c->pending ? F_HAS_PENDING : 0) | (c->unresolved ? F_WALK_INLINE : 0);
            const bool kscoped = c->profiling == 2 && c->lvl_exchange && c->lvl_wg == 1024;
            if (c->profiling && !kscoped) HIPCHK(hipEventRecord(c->pev[4 * it], c->stream));
            if (kscoped) { c->kev0 = c->pev[4 * it]; c->kev1 = c->pev[4 * it + 1]; }
            launch_chain_iter(c, t, flags);
            c->kev0 = c->kev1 = nullptr;
            if (c->profiling && !kscoped) HIPCHK(hipEventRecord(c->pev[4 * it + 1], c->stream));
            c->prev_open = true;
            c->pending = false;
            if (c->profiling) c->pev_exch[it] = 0;
            c->unresolved = false;
            if (exchange_active(c, t)) {
                if (c->inline_walk) {
                    c->unresolved = true;   // resolved in the prologue of the next chain kernel (or by resolve_now)
                } else {
                    if (kscoped) { c->kev0 = c->pev[4 * it + 2]; c->kev1 = c->pev[4 * it + 3]; c->pev_exch[it] = 1; }
                    launch_resolve(c, t, (c->lvl_exchange || c->lds_exchange) ? nullptr : c->rec[c->cur]);
                    c->kev0 = c->kev1 = nullptr;
                }
                c->pending = true;
            }
            if (c->profiling && !kscoped) { HIPCHK(hipEventRecord(c->pev[4 * it + 2], c->stream)); HIPCHK(hipEventRecord(c->pev[4 * it + 3], c->stream)); }
            c->iter = t;
        }
        HIPCHK(hipEventRecord(c->ev1, c->stream));
        HIPCHK(hipGetLastError());
        if (c->profiling) c->pev_iters = n_iters;
        c->pending_timing = true;
        c->timing.iters = n_iters;
        c->timing.chain_evals = (int64_t)n_iters * c->P.N;
    } catch (const std::string& m) {
        return fail(c, SMM_ERR_HIP, m);
    }
    return SMM_OK;
}

int smm_bgp_step(void* ctx, int32_t n_iters) {
    const int rc = smm_bgp_step_async(ctx, n_iters);
    if (rc != SMM_OK) return rc;
    return smm_sync(ctx);
}

int smm_bgp_local_step(void* ctx) {
    Ctx* c = (Ctx*)ctx;
    if (!c) return SMM_ERR_INVALID_ARG;
    if (c->rec_external) return fail(c, SMM_ERR_STATE, "records are in the gather buffer: call smm_bgp_sharded_finish first");
    if (c->iter + 1 > c->P.T) return fail(c, SMM_ERR_MAXITER, "step beyond maxiter (history capacity)");
    try {
        HIPCHK(hipSetDevice(c->device));
        const int t = c->iter + 1;
        ensure_windows(c, t);
        const int flags = (c->prev_open ? F_CLOSE_PREV : 0) | (c->pending ? F_HAS_PENDING : 0);
        launch_chain_iter(c, t, flags);
        HIPCHK(hipGetLastError());
        c->prev_open = true;
        c->pending = false;
        c->iter = t;
    } catch (const std::string& m) {
        return fail(c, SMM_ERR_HIP, m);
    }
    return SMM_OK;
}

// Sharded iteration in two enqueues (instead of local_step / export / exchange = five): the exchange of the previous
// iteration is resolved from gathered_prev, the chain kernel takes every chain's continuation record (its own or the
// donor's) straight from gathered_prev and writes the new records into this shard's slice of gathered_next.
int smm_bgp_sharded_step(void* ctx, const void* gathered_prev_dev, void* gathered_next_dev) {
    Ctx* c = (Ctx*)ctx;
    if (!c || !gathered_next_dev) return SMM_ERR_INVALID_ARG;
    if (c->iter + 1 > c->P.T) return fail(c, SMM_ERR_MAXITER, "step beyond maxiter (history capacity)");
    if (c->rec_external && !gathered_prev_dev) return fail(c, SMM_ERR_INVALID_ARG, "gathered_prev required: the last records live there");
    if (c->unresolved) return fail(c, SMM_ERR_STATE, "mixing smm_bgp_step and smm_bgp_sharded_step without a flush");
    try {
        HIPCHK(hipSetDevice(c->device));
        const int t = c->iter + 1;
        const KParams& P = c->P;
        int flags = (c->prev_open ? F_CLOSE_PREV : 0);
        if (c->rec_external) {
            if (c->pending_ext) {   // exchangeMoves! of iteration t-1 over the gathered records (before its plan window can move on)
                launch_resolve(c, t - 1, (const double*)gathered_prev_dev);
                flags |= F_HAS_PENDING;
            }
            flags |= F_GLOBAL_REC;
            c->ext_rec_in = (const double*)gathered_prev_dev;
        } else if (c->pending) {
            flags |= F_HAS_PENDING;   // resolved earlier through the three-phase calls
        }
        ensure_windows(c, t);
        c->ext_rec_out = (double*)gathered_next_dev + (size_t)P.offset * P.RW;
        launch_chain_iter(c, t, flags);
        c->ext_rec_in = nullptr; c->ext_rec_out = nullptr;
        HIPCHK(hipGetLastError());
        c->prev_open = true;
        c->pending = false;
        c->rec_external = true;
        c->pending_ext = exchange_active(c, t);
        c->iter = t;
    } catch (const std::string& m) {
        return fail(c, SMM_ERR_HIP, m);
    }
    return SMM_OK;
}

// settle the last sharded_step: resolve its exchange from the gathered records and bring records, history and
// counters into the context (afterwards history/state can be read, or stepping continues in either form)
int smm_bgp_sharded_finish(void* ctx, const void* gathered_dev) {
    Ctx* c = (Ctx*)ctx;
    if (!c) return SMM_ERR_INVALID_ARG;
    if (!c->rec_external) return SMM_OK;
    if (!gathered_dev) return SMM_ERR_INVALID_ARG;
    try {
        HIPCHK(hipSetDevice(c->device));
        const KParams& P = c->P;
        int flags = (c->prev_open ? F_CLOSE_PREV : 0) | F_GLOBAL_REC;
        if (c->pending_ext) {
            launch_resolve(c, c->iter, (const double*)gathered_dev);
            flags |= F_HAS_PENDING;
        }
        hipLaunchKernelGGL(k_flush, dim3((P.N + 255) / 256), dim3(256), 0, c->stream, P, c->iter + 1, (const double*)gathered_dev,
                           c->rec[c->cur ^ 1], flags);
        HIPCHK(hipGetLastError());
        c->cur ^= 1;
        c->pending = false; c->prev_open = false; c->rec_external = false; c->pending_ext = false;
    } catch (const std::string& m) {
        return fail(c, SMM_ERR_HIP, m);
    }
    return SMM_OK;
}

int smm_bgp_record_doubles(void* ctx) {
    Ctx* c = (Ctx*)ctx;
    return c ? c->P.RW : SMM_ERR_INVALID_ARG;
}

int smm_bgp_export_records_dev(void* ctx, void* rec_dev) {
    Ctx* c = (Ctx*)ctx;
    if (!c || !rec_dev) return SMM_ERR_INVALID_ARG;
    try {
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(hipMemcpyAsync(rec_dev, c->rec[c->cur], (size_t)c->P.RW * c->P.N * sizeof(double), hipMemcpyDeviceToDevice,
                              c->stream));
    } catch (const std::string& m) {
        return fail(c, SMM_ERR_HIP, m);
    }
    return SMM_OK;
}

int smm_bgp_exchange_dev(void* ctx, const void* gathered_dev) {
    Ctx* c = (Ctx*)ctx;
    if (!c || !gathered_dev) return SMM_ERR_INVALID_ARG;
    if (c->iter < 1) return fail(c, SMM_ERR_STATE, "exchange before the first local step");
    if (c->pending) return fail(c, SMM_ERR_STATE, "exchange already resolved for this iteration");
    try {
        HIPCHK(hipSetDevice(c->device));
        if (exchange_active(c, c->iter)) {
            const KParams& P = c->P;
            launch_resolve(c, c->iter, (const double*)gathered_dev);
            hipLaunchKernelGGL(k_exch_apply, dim3((P.N + 255) / 256), dim3(256), 0, c->stream, P, c->iter,
                               (const double*)gathered_dev, c->rec[c->cur]);
        }
        HIPCHK(hipGetLastError());
    } catch (const std::string& m) {
        return fail(c, SMM_ERR_HIP, m);
    }
    return SMM_OK;
}

int smm_eval_batch(void* ctx, const double* params, int32_t M, double* value, double* sim_moments, int8_t* status) {
    Ctx* c = (Ctx*)ctx;
    if (!c || !params || M < 0 || !value || !sim_moments || !status) return SMM_ERR_INVALID_ARG;
    if (M == 0) return SMM_OK;
    try {
        HIPCHK(hipSetDevice(c->device));
        const KParams& P = c->P;
        double *dp = nullptr, *dv = nullptr, *dm = nullptr;
        int8_t* ds = nullptr;
        HIPCHK(hipMalloc((void**)&dp, (size_t)P.np * M * 8));
        HIPCHK(hipMalloc((void**)&dv, (size_t)M * 8));
        HIPCHK(hipMalloc((void**)&dm, (size_t)P.nm * M * 8));
        HIPCHK(hipMalloc((void**)&ds, (size_t)M));
        if (c->obj == SMM_OBJ_USER) {   // the user's kernel wants [M][np] / [M][nm]: transpose on the host
            std::vector<double> tp((size_t)M * P.np), tm((size_t)M * P.nm);
            std::vector<int> ts((size_t)M);
            for (int i = 0; i < M; ++i)
                for (int k = 0; k < P.np; ++k) tp[(size_t)i * P.np + k] = params[(size_t)k * M + i];
            int* dsi = nullptr;
            HIPCHK(hipMalloc((void**)&dsi, (size_t)M * sizeof(int)));
            HIPCHK(hipMemcpyAsync(dp, tp.data(), tp.size() * 8, hipMemcpyHostToDevice, c->stream));
            launch_user_kernel(c, dp, M, dm, dv, dsi);
            HIPCHK(hipMemcpyAsync(value, dv, (size_t)M * 8, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipMemcpyAsync(tm.data(), dm, tm.size() * 8, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipMemcpyAsync(ts.data(), dsi, ts.size() * sizeof(int), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            for (int i = 0; i < M; ++i) {
                status[i] = (int8_t)ts[i];
                for (int k = 0; k < P.nm; ++k) sim_moments[(size_t)k * M + i] = tm[(size_t)i * P.nm + k];
            }
            (void)hipFree(dp); (void)hipFree(dv); (void)hipFree(dm); (void)hipFree(ds); (void)hipFree(dsi);
            return SMM_OK;
        }
        HIPCHK(hipMemcpyAsync(dp, params, (size_t)P.np * M * 8, hipMemcpyHostToDevice, c->stream));
        constexpr int CT = 8;
        if (is_sim(c->obj))
            hipLaunchKernelGGL((k_eval_batch<1, CT>), dim3((M + CT - 1) / CT), dim3(WG), tile_smem(c, CT), c->stream, P, dp, M, dv, dm, ds);
        else if (c->obj == SMM_OBJ_DENSE)
            hipLaunchKernelGGL((k_eval_batch<2, 16>), dim3((M + 15) / 16), dim3(WG), tile_smem(c, 16), c->stream, P, dp, M, dv, dm, ds);
        else
            hipLaunchKernelGGL((k_eval_batch<0, CT>), dim3((M + CT - 1) / CT), dim3(WG), tile_smem(c, CT), c->stream, P, dp, M, dv, dm, ds);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(value, dv, (size_t)M * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(sim_moments, dm, (size_t)P.nm * M * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(status, ds, (size_t)M, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        (void)hipFree(dp); (void)hipFree(dv); (void)hipFree(dm); (void)hipFree(ds);
    } catch (const std::string& m) {
        return fail(c, SMM_ERR_HIP, m);
    }
    return SMM_OK;
}

// history(c) (AlgoBGP.jl:138-160): download iterations t0..t1-1 and transpose the per-chain history
// records into the ABI's structure-of-arrays buffers.
int smm_get_history(void* ctx, int32_t t0, int32_t t1, smm_history_t* out) {
    Ctx* c = (Ctx*)ctx;
    if (!c || !out || t0 < 0 || t1 < t0 || t1 > c->P.T) return SMM_ERR_INVALID_ARG;
    try {
        HIPCHK(hipSetDevice(c->device));
        flush(c);
        HIPCHK(hipStreamSynchronize(c->stream));
        const KParams& P = c->P;
        const size_t N = P.N, HW = P.HW, np = P.np, nm = P.nm;
        std::vector<double> row(N * HW);
        for (int t = t0; t < t1; ++t) {
            HIPCHK(hipMemcpy(row.data(), P.hrec + (size_t)t * N * HW, row.size() * 8, hipMemcpyDeviceToHost));
            const size_t o = (size_t)(t - t0) * N;
            for (size_t i = 0; i < N; ++i) {
                const double* h = row.data() + i * HW;
                if (out->value) out->value[o + i] = h[H_VALUE];
                if (out->prob) out->prob[o + i] = h[H_PROB];
                if (out->curr_val) out->curr_val[o + i] = h[H_CURR];
                if (out->best_val) out->best_val[o + i] = h[H_BEST];
                if (out->best_id) out->best_id[o + i] = (int32_t)h[H_BESTID];
                if (out->exchanged) out->exchanged[o + i] = (int32_t)h[H_EXCH];
                if (out->accepted) out->accepted[o + i] = (uint8_t)h[H_ACC];
                if (out->status) out->status[o + i] = (int8_t)h[H_STATUS];
                if (out->params)
                    for (size_t k = 0; k < np; ++k) out->params[((size_t)(t - t0) * np + k) * N + i] = h[H_PARAMS + k];
                if (out->sim_moments)
                    for (size_t k = 0; k < nm; ++k) out->sim_moments[((size_t)(t - t0) * nm + k) * N + i] = h[H_PARAMS + np + k];
            }
        }
    } catch (const std::string& m) {
        return fail(c, SMM_ERR_HIP, m);
    }
    return SMM_OK;
}

int smm_get_state(void* ctx, smm_state_t* s) {
    Ctx* c = (Ctx*)ctx;
    if (!c || !s) return SMM_ERR_INVALID_ARG;
    try {
        HIPCHK(hipSetDevice(c->device));
        flush(c);
        HIPCHK(hipStreamSynchronize(c->stream));
        const KParams& P = c->P;
        const size_t N = P.N, RW = P.RW, np = P.np, nm = P.nm;
        std::vector<double> cs(N * CSW), rec(N * RW);
        HIPCHK(hipMemcpy(cs.data(), P.cs, cs.size() * 8, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(rec.data(), c->rec[c->cur], rec.size() * 8, hipMemcpyDeviceToHost));
        s->iter = c->iter;
        for (size_t i = 0; i < N; ++i) {
            const double* b = cs.data() + i * CSW;
            const double* r = rec.data() + i * RW;
            if (s->sigma) s->sigma[i] = b[CS_SIGMA];
            if (s->accept_rate) s->accept_rate[i] = b[CS_RATE];
            if (s->n_noex) s->n_noex[i] = (int32_t)b[CS_NNOEX];
            if (s->n_acc_noex) s->n_acc_noex[i] = (int32_t)b[CS_NACC];
            if (s->best_val) s->best_val[i] = b[CS_BEST];
            if (s->best_id) s->best_id[i] = (int32_t)b[CS_BESTID];
            if (s->la_value) s->la_value[i] = r[0];
            if (s->la_prob) s->la_prob[i] = r[1];
            if (s->la_status) s->la_status[i] = (int8_t)r[2];
            if (s->la_params)
                for (size_t k = 0; k < np; ++k) s->la_params[k * N + i] = r[3 + k];
            if (s->la_sim_moments)
                for (size_t k = 0; k < nm; ++k) s->la_sim_moments[k * N + i] = r[3 + np + k];
        }
    } catch (const std::string& m) {
        return fail(c, SMM_ERR_HIP, m);
    }
    return SMM_OK;
}

// restart! (AlgoBGP.jl:804-884) with clean resume semantics: continue at iteration iter+1
int smm_set_state(void* ctx, const smm_state_t* s, const smm_history_t* h) {
    Ctx* c = (Ctx*)ctx;
    if (!c || !s || s->iter < 0 || s->iter > c->P.T) return SMM_ERR_INVALID_ARG;
    if (s->iter > 0 && !h) return fail(c, SMM_ERR_INVALID_ARG, "history of iterations 0..iter-1 required");
    if (!s->sigma || !s->accept_rate || !s->n_noex || !s->n_acc_noex || !s->best_val || !s->best_id || !s->la_value ||
        !s->la_prob || !s->la_status || !s->la_params || !s->la_sim_moments)
        return fail(c, SMM_ERR_INVALID_ARG, "smm_set_state needs every field of smm_state_t");
    if (s->iter > 0 && (!h->value || !h->prob || !h->curr_val || !h->best_val || !h->params || !h->sim_moments ||
                        !h->best_id || !h->exchanged || !h->accepted || !h->status))
        return fail(c, SMM_ERR_INVALID_ARG, "smm_set_state needs every field of smm_history_t");
    try {
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(hipStreamSynchronize(c->stream));
        KParams& P = c->P;
        const size_t N = P.N, RW = P.RW, HW = P.HW, np = P.np, nm = P.nm;
        std::vector<double> cs(N * CSW), rec(N * RW, 0.0);
        HIPCHK(hipMemcpy(cs.data(), P.cs, cs.size() * 8, hipMemcpyDeviceToHost));  // keeps acc_tuner
        for (size_t i = 0; i < N; ++i) {
            double* b = cs.data() + i * CSW;
            double* r = rec.data() + i * RW;
            b[CS_SIGMA] = s->sigma[i]; b[CS_RATE] = s->accept_rate[i];
            b[CS_NNOEX] = (double)s->n_noex[i]; b[CS_NACC] = (double)s->n_acc_noex[i];
            b[CS_LACC] = 0.0; b[CS_WASX] = 0.0;
            b[CS_BEST] = s->best_val[i]; b[CS_BESTID] = (double)s->best_id[i];
            b[CS_BESTP] = s->best_val[i]; b[CS_BESTPID] = (double)s->best_id[i];
            r[0] = s->la_value[i]; r[1] = s->la_prob[i]; r[2] = (double)s->la_status[i];
            for (size_t k = 0; k < np; ++k) r[3 + k] = s->la_params[k * N + i];
            for (size_t k = 0; k < nm; ++k) r[3 + np + k] = s->la_sim_moments[k * N + i];
        }
        HIPCHK(hipMemcpy(P.cs, cs.data(), cs.size() * 8, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(c->rec[c->cur], rec.data(), rec.size() * 8, hipMemcpyHostToDevice));
        std::vector<double> row(N * HW, 0.0);
        for (int t = 0; t < s->iter; ++t) {
            const size_t o = (size_t)t * N;
            for (size_t i = 0; i < N; ++i) {
                double* hr = row.data() + i * HW;
                hr[H_VALUE] = h->value[o + i]; hr[H_PROB] = h->prob[o + i]; hr[H_CURR] = h->curr_val[o + i];
                hr[H_BEST] = h->best_val[o + i]; hr[H_BESTID] = (double)h->best_id[o + i];
                hr[H_EXCH] = (double)h->exchanged[o + i]; hr[H_ACC] = (double)h->accepted[o + i];
                hr[H_STATUS] = (double)h->status[o + i];
                for (size_t k = 0; k < np; ++k) hr[H_PARAMS + k] = h->params[((size_t)t * np + k) * N + i];
                for (size_t k = 0; k < nm; ++k) hr[H_PARAMS + np + k] = h->sim_moments[((size_t)t * nm + k) * N + i];
            }
            HIPCHK(hipMemcpy(P.hrec + (size_t)t * N * HW, row.data(), row.size() * 8, hipMemcpyHostToDevice));
        }
        c->iter = s->iter;
        c->unresolved = false;
        c->pending = false;
        c->prev_open = false;
    } catch (const std::string& m) {
        return fail(c, SMM_ERR_HIP, m);
    }
    return SMM_OK;
}

int smm_get_timing(void* ctx, smm_timing_t* out) {
    Ctx* c = (Ctx*)ctx;
    if (!c || !out) return SMM_ERR_INVALID_ARG;
    *out = c->timing;
    return SMM_OK;
}

int smm_set_profiling(void* ctx, int32_t on) {
    Ctx* c = (Ctx*)ctx;
    if (!c) return SMM_ERR_INVALID_ARG;
    c->profiling = on < 0 ? 0 : (on > 2 ? 2 : on);
    return SMM_OK;
}

// debug (not part of the public header): per-workgroup wall_clock64 stamps of the last k_chain_iter
int smm_debug_ts(void* ctx, unsigned long long* out, int n_wg) {
    Ctx* c = (Ctx*)ctx;
    if (!c || !c->P.ts) return SMM_ERR_INVALID_ARG;
    if (n_wg < 0) {  // the exchange kernel's stamps
        if (hipMemcpy(out, c->P.ts + (size_t)8 * 60000, 8 * 100, hipMemcpyDeviceToHost) != hipSuccess) return SMM_ERR_HIP;
        return SMM_OK;
    }
    if (hipMemcpy(out, c->P.ts, (size_t)n_wg * 8 * 8, hipMemcpyDeviceToHost) != hipSuccess) return SMM_ERR_HIP;
    return SMM_OK;
}

int smm_get_Z(void* ctx, double* Z) {
    Ctx* c = (Ctx*)ctx;
    if (!c || !Z) return SMM_ERR_INVALID_ARG;
    try {
        HIPCHK(hipSetDevice(c->device));
        for (int k = 0; k < c->P.nm; ++k)
            HIPCHK(hipMemcpy(Z + (size_t)k * c->P.ns, c->P.Z + (size_t)k * c->P.zstride, (size_t)c->P.ns * sizeof(double),
                             hipMemcpyDeviceToHost));
    } catch (const std::string& m) {
        return fail(c, SMM_ERR_HIP, m);
    }
    return SMM_OK;
}

}  // extern "C"
