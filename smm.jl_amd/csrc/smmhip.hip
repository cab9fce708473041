// smmhip.hip — libsmmhip.so: hand-written HIP (gfx950 / MI355X) implementation of the BGP
// parallel-tempering iteration of floswald/SMM.jl behind the C ABI of include/smmhip.h.
//
// One iteration of computeNextIteration!(algo::MAlgoBGP) (src/mopt/AlgoBGP.jl:589-640) is
//   k_chain_iter   : next_eval for every chain at once (proposal :424-471, objective
//                    mprob.jl:175-188 -> ObjExamples.jl:59-116, doAcceptReject! :324-392,
//                    set_eval! :220-245); a 256-lane workgroup owns a tile of CT chains,
//                    the ns simulated draws are spread over the lanes, the shock matrix Z is
//                    re-used CT times from registers, moments are reduced by a transposed
//                    wave reduction + LDS.
//   k_exch_resolve : exchangeMoves! (:647-716): pair sampling + ordered swap resolution by
//                    ONE workgroup, exact sequential semantics via dependency rounds.
//   k_exch_apply   : swap_ev_ij! (:734-749) + set_exchanged! for the local chains.
// Data layout: structure-of-arrays, chain index fastest (coalesced), FP64 throughout.
#include <hip/hip_runtime.h>
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

constexpr int WG = SMM_REDUCE_LANES;  // 256 lanes own one chain tile (numerical contract)
constexpr int MAX_DIM = 64;           // np, nm <= 64
constexpr int XWG = 1024;             // exchange-resolution workgroup

// error word: min over (iter<<34 | chain<<2 | kind); kind 1 = negative objective, 2 = no draw
constexpr unsigned long long ERR_NONE = ~0ull;

struct KParams {
    // problem
    int np, nm, ns, obj;
    const double *init, *lb, *ub, *mom, *w, *objp;
    const double* Z;  // [nm][ns]
    // opts
    int N, Ng, offset, T;
    int sigma_update_steps, smpl_iters, batch_size;
    double sigma_adjust_by;
    uint64_t seed;
    const double *acc_tuner_g, *min_improve_g;  // [Ng]
    // tables
    const double* utab;   // [T][N] or null
    const double* ntab;   // [T][K][np][N] or null
    int ntries;
    const int32_t* pairtab;  // [T][n_pairs][2] or null
    int n_pairs_tab;
    // chain state [N]
    double *sigma, *accept_rate;
    double *la_value, *la_prob, *la_params, *la_simM;
    int8_t* la_status;
    int32_t *n_noex, *n_acc;
    double* best_val;
    int32_t* best_id;
    double* rec;  // [(3+np+nm)][N] last accepted records after the accept step
    // exchange scratch [Ng]
    double* xval;
    int32_t *xsrc, *xpartner, *xnext, *xpairs;
    // history [T][..][N]
    double *h_value, *h_prob, *h_curr, *h_best, *h_params, *h_simM;
    int32_t *h_best_id, *h_exch;
    uint8_t* h_acc;
    int8_t* h_status;
    unsigned long long* err;
};

__device__ inline void report_error(const KParams& P, int kind, int t, int gchain) {
    const unsigned long long key = ((unsigned long long)t << 34) | ((unsigned long long)gchain << 2) | (unsigned)kind;
    atomicMin(P.err, key);
}

// Transposed wave reduction: every lane holds CT partial sums a[0..CT); on return lane l holds
// the 64-lane total of accumulator acc_index<CT>(l), combined by the canonical halving tree
// (offsets 32,16,8,4,2,1; IEEE addition is commutative so both partners get the same bits).
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
__device__ inline int acc_index(int lane) {
    // the log2(CT) top lane bits, most significant first
    constexpr int LG = (CT == 1) ? 0 : (CT == 2) ? 1 : (CT == 4) ? 2 : (CT == 8) ? 3 : (CT == 16) ? 4 : (CT == 32) ? 5 : 6;
    return LG == 0 ? 0 : (lane >> (6 - LG));
}
template <int CT>
__device__ inline bool acc_writer(int lane) {
    return (lane & ((64 / CT) - 1)) == 0;
}

// The simulation of objfunc_norm (ObjExamples.jl:76-79) for a tile of CT chains:
// X[k,s] = theta_c[k] + z[k,s]; partial sums over the draws of each lane; reduction.
// s_theta [CT][np] (LDS), s_part [4][CT][nm] (LDS).
template <int CT>
__device__ inline void simulate_tile(const KParams& P, const double* s_theta, double* s_part, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    for (int k = 0; k < P.nm; ++k) {
        double mu[CT], acc[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            mu[c] = s_theta[c * P.np + k];
            acc[c] = 0.0;
        }
        const double* __restrict__ Zk = P.Z + (size_t)k * P.ns;
        // lane `tid` sums its draws tid, tid+256, ... in that order (numerical contract).  U rows
        // are loaded ahead of use so that U independent L2 reads are in flight per lane.
        constexpr int U = 8;
        int s = tid;
        double zc[U];
        bool have = (s + (U - 1) * WG) < P.ns;
        if (have) {
#pragma unroll
            for (int u = 0; u < U; ++u) zc[u] = Zk[s + u * WG];
        }
        while (have) {
            const int sn = s + U * WG;
            const bool have_next = (sn + (U - 1) * WG) < P.ns;
            double zn[U];
            if (have_next) {
#pragma unroll
                for (int u = 0; u < U; ++u) zn[u] = Zk[sn + u * WG];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    const double x = zc[u] + mu[c];
                    acc[c] = acc[c] + x;
                }
            }
            if (have_next) {
#pragma unroll
                for (int u = 0; u < U; ++u) zc[u] = zn[u];
            }
            s = sn;
            have = have_next;
        }
        {   // remaining (< U full rows + the ragged last row): masked
            double zt[U];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                ok[u] = (s + u * WG) < P.ns;
                zt[u] = ok[u] ? Zk[s + u * WG] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (ok[u]) {
#pragma unroll
                    for (int c = 0; c < CT; ++c) {
                        const double x = zt[u] + mu[c];
                        acc[c] = acc[c] + x;
                    }
                }
            }
        }
        const double tot = wave_reduce_transposed<CT>(acc, lane);
        if (acc_writer<CT>(lane)) s_part[(wave * CT + acc_index<CT>(lane)) * P.nm + k] = tot;
    }
}

// value / simulated moments / status for one chain from its reduced sums
// (ObjExamples.jl:79-110; banana :251-265; "exception" -> status -2, mprob.jl:183-186).
template <int CT>
__device__ inline void finish_objective(const KParams& P, const double* theta /*LDS [np]*/, const double* s_part,
                                        int ci, double* simM /*[nm] out, LDS*/, double& value, int& status) {
    if (P.obj == SMM_OBJ_BANANA) {
        double v = 0.0;
        for (int i = 0; i + 1 < P.np; ++i) {
            const double a = theta[i], b = theta[i + 1];
            const double t1 = b - a * a;
            const double t2 = 1.0 - a;
            const double term = 100.0 * (t1 * t1) + t2 * t2;
            v = (i == 0) ? term : v + term;
        }
        for (int k = 0; k < P.nm; ++k) simM[k] = P.mom[k] + 2.2;
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
    for (int k = 0; k < P.nm; ++k) {
        double tot = s_part[(0 * CT + ci) * P.nm + k];
#pragma unroll
        for (int wv = 1; wv < WG / 64; ++wv) tot = tot + s_part[(wv * CT + ci) * P.nm + k];
        const double m = tot / (double)P.ns;
        simM[k] = m;
        double d = m - P.mom[k];
        const double wk = P.w[k];
        if (!isnan(wk)) d = d / wk;
        const double v = d * d;
        vsum = (k == 0) ? v : vsum + v;
    }
    value = vsum / (double)P.nm;
    status = 1;
}

// ------------------------------------------------------------------------------------------
// k_chain_iter: one next_eval (AlgoBGP.jl:272-294) for every local chain, iteration t (1-based)
// ------------------------------------------------------------------------------------------
template <bool SIM, int CT>
__global__ __launch_bounds__(WG) void k_chain_iter(const KParams P, const int t, const int close_iter) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* s_theta = smem;                         // [CT][np]
    double* s_simM = s_theta + CT * P.np;           // [CT][nm]
    double* s_part = s_simM + CT * P.nm;            // [4][CT][nm]
    const int tid = threadIdx.x;
    const int c = blockIdx.x * CT + tid;
    const bool chain_lane = (tid < CT) && (c < P.N);
    const int gc = P.offset + c;
    const int N = P.N, np = P.np, nm = P.nm;

    // ---- proposal(c), AlgoBGP.jl:424-471 ----
    if (tid < CT) {
        double* th = s_theta + tid * np;
        if (!chain_lane) {
            for (int k = 0; k < np; ++k) th[k] = 0.0;
        } else if (t == 1) {
            for (int k = 0; k < np; ++k) th[k] = P.init[k];  // :426-427
        } else {
            const double sig = P.sigma[c];
            const int bs = P.batch_size;
            const int max_tries = P.ntab ? min(P.ntries, P.smpl_iters) : P.smpl_iters;
            for (int b0 = 0; b0 < np; b0 += bs) {
                bool ok = false;
                for (int r = 0; r < max_tries && !ok; ++r) {  // mysample, :400-410
                    ok = true;
                    double zc0 = 0.0, zc1 = 0.0;
                    int zq = -1;
                    for (int k = b0; k < b0 + bs; ++k) {
                        const double lbk = P.lb[k], ubk = P.ub[k];
                        const double mu01 = (P.la_params[(size_t)k * N + c] - lbk) / (ubk - lbk);  // mprob.jl:248
                        double z;
                        if (P.ntab) {
                            z = P.ntab[((((size_t)(t - 1) * P.ntries + r) * np + k) * N) + c];
                        } else {
                            if ((k >> 1) != zq) {
                                zq = k >> 1;
                                rng_prop_normal2(P.seed, (uint32_t)gc, (uint32_t)t, (uint32_t)r, (uint32_t)zq, zc0, zc1);
                            }
                            z = (k & 1) ? zc1 : zc0;
                        }
                        const double step = sig * z;  // MvNormal(mu01, sigma): x = mu + sigma*z
                        const double x = mu01 + step;
                        th[k] = x;
                        if (!(x >= 0.0 && x <= 1.0)) ok = false;  // inclusive bounds, :405
                    }
                }
                if (!ok) report_error(P, 2, t, gc);  // :409
            }
            for (int k = 0; k < np; ++k) {
                const double lbk = P.lb[k];
                const double span = P.ub[k] - lbk;
                const double sc = th[k] * span;
                th[k] = sc + lbk;  // mapto_ab, mprob.jl:271
            }
        }
    }
    __syncthreads();

    // ---- simulation: all 256 lanes, ns draws x nm moments x CT chains ----
    if constexpr (SIM) {
        simulate_tile<CT>(P, s_theta, s_part, tid);
        __syncthreads();
    }

    // ---- objective value, doAcceptReject! (:324-392), set_eval! (:220-245) ----
    if (chain_lane) {
        const double* th = s_theta + tid * np;
        double* sm = s_simM + tid * nm;
        double value;
        int status;
        finish_objective<CT>(P, th, s_part, tid, sm, value, status);

        double prob;
        bool acc;
        const double old = P.la_value[c];
        if (t == 1) {  // :326-332
            prob = 1.0; acc = true; status = 1;
        } else if (status < 0) {  // :336-338
            prob = 0.0; acc = false;
        } else {
            if (!(value >= 0.0)) report_error(P, 1, t, gc);  // :341
            const double e = exp(P.acc_tuner_g[gc] * (old - value));
            prob = (e != e) ? e : (e < 1.0 ? e : 1.0);  // minimum([1.0,e]), NaN propagates (:344)
            if (!isfinite(prob)) { prob = 0.0; acc = false; status = -1; }  // :350-353
            else if (!isfinite(old)) { prob = 1.0; acc = true; }            // :355-359
            else {
                status = 1;
                const double u = P.utab ? P.utab[(size_t)(t - 1) * N + c] : rng_u(P.seed, (uint32_t)gc, (uint32_t)t);
                acc = prob > u;  // strict, :362-367
            }
        }
        // set_acceptRate!, :253-257 (iteration t has exchanged==0 at this point)
        const int nn = P.n_noex[c], na = P.n_acc[c];
        const double rate = (double)(na + (acc ? 1 : 0)) / (double)(nn + 1);
        P.accept_rate[c] = rate;
        if (t > 1 && (t % P.sigma_update_steps) == 0) {  // :381-390
            const double s0 = P.sigma[c];
            P.sigma[c] = (rate > 0.234) ? s0 * (1.0 + P.sigma_adjust_by) : s0 * (1.0 - P.sigma_adjust_by);
        }
        // set_eval!, :220-245
        const size_t row = (size_t)(t - 1) * N + c;
        double bestv, currv;
        int bestid;
        if (t == 1) { bestv = value; currv = value; bestid = 1; }
        else {
            currv = acc ? value : old;  // curr_val[t-1] == value of the last accepted record
            const double bp = P.best_val[c];
            if (value < bp) { bestv = value; bestid = t; }
            else { bestv = bp; bestid = P.best_id[c]; }
        }
        P.best_val[c] = bestv; P.best_id[c] = bestid;
        P.h_value[row] = value; P.h_prob[row] = prob; P.h_curr[row] = currv; P.h_best[row] = bestv;
        P.h_best_id[row] = bestid; P.h_exch[row] = 0; P.h_acc[row] = acc ? 1 : 0; P.h_status[row] = (int8_t)status;
        for (int k = 0; k < np; ++k) P.h_params[((size_t)(t - 1) * np + k) * N + c] = th[k];
        for (int k = 0; k < nm; ++k) P.h_simM[((size_t)(t - 1) * nm + k) * N + c] = sm[k];
        // last accepted record (lastAccepted :209-215) + its export row for the exchange step
        double rv, rp; int rs;
        if (acc) {
            rv = value; rp = prob; rs = status;
            P.la_value[c] = value; P.la_prob[c] = prob; P.la_status[c] = (int8_t)status;
            for (int k = 0; k < np; ++k) P.la_params[(size_t)k * N + c] = th[k];
            for (int k = 0; k < nm; ++k) P.la_simM[(size_t)k * N + c] = sm[k];
        } else {
            rv = old; rp = P.la_prob[c]; rs = P.la_status[c];
        }
        P.rec[c] = rv; P.rec[(size_t)N + c] = rp; P.rec[(size_t)2 * N + c] = (double)rs;
        for (int k = 0; k < np; ++k) P.rec[(size_t)(3 + k) * N + c] = acc ? th[k] : P.la_params[(size_t)k * N + c];
        for (int k = 0; k < nm; ++k) P.rec[(size_t)(3 + np + k) * N + c] = acc ? sm[k] : P.la_simM[(size_t)k * N + c];
        if (close_iter) {  // no exchange phase follows: this iteration counts towards accept_rate
            P.n_noex[c] = nn + 1;
            P.n_acc[c] = na + (acc ? 1 : 0);
        }
    }
}

// batched evaluateObjective(m,p), mprob.jl:175-188: params [np][M] -> value, simM [nm][M], status
template <bool SIM, int CT>
__global__ __launch_bounds__(WG) void k_eval_batch(const KParams P, const double* __restrict__ params, const int M,
                                                   double* __restrict__ value, double* __restrict__ simM,
                                                   int8_t* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* s_theta = smem;
    double* s_simM = s_theta + CT * P.np;
    double* s_part = s_simM + CT * P.nm;
    const int tid = threadIdx.x;
    const int i = blockIdx.x * CT + tid;
    const bool chain_lane = (tid < CT) && (i < M);
    if (tid < CT)
        for (int k = 0; k < P.np; ++k) s_theta[tid * P.np + k] = chain_lane ? params[(size_t)k * M + i] : 0.0;
    __syncthreads();
    if constexpr (SIM) {
        simulate_tile<CT>(P, s_theta, s_part, tid);
        __syncthreads();
    }
    if (chain_lane) {
        double v;
        int st;
        double* sm = s_simM + tid * P.nm;
        finish_objective<CT>(P, s_theta + tid * P.np, s_part, tid, sm, v, st);
        value[i] = v;
        status[i] = (int8_t)st;
        for (int k = 0; k < P.nm; ++k) simM[(size_t)k * M + i] = sm[k];
    }
}

// ------------------------------------------------------------------------------------------
// k_exch_resolve: exchangeMoves! (AlgoBGP.jl:647-716) over all Ng chains, one workgroup.
// The reference walks the K sampled pairs in order and swaps the two chains' last accepted
// records when value_i - value_j > min_improve_i (:688).  Pairs that share no chain commute,
// so the list is executed in dependency rounds: in each round every not-yet-executed pair
// bids (atomicMin of its list position) on both of its chains; a pair that wins both bids has
// no unexecuted predecessor touching either chain and is executed.  Result == sequential walk.
// Outputs: xsrc[g] = chain whose post-accept record chain g ends up with, xpartner[g] = last
// exchange partner (1-based) or 0.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(XWG) void k_exch_resolve(const KParams P, const int t, const double* __restrict__ gathered) {
    const int tid = threadIdx.x;
    const int Ng = P.Ng, N = P.N, R = 3 + P.np + P.nm;
    const int K = P.pairtab ? P.n_pairs_tab : n_exchange_pairs(Ng);
    for (int g = tid; g < Ng; g += XWG) {
        const int shard = g / N, l = g - shard * N;
        P.xval[g] = gathered[(size_t)shard * R * N + l];
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
}

// k_exch_resolve_lds: same result as k_exch_resolve, for K <= Ng <= XLDS_MAX, everything in LDS.
// Instead of barrier-separated rounds the pair list is executed as a data-flow graph:
//   1. per chain c, the list positions of the pairs touching c are bucketed (counting sort:
//      LDS atomics + block scan) and every pair learns its rank r_i, r_j among the pairs of
//      its two chains;
//   2. ticket[c] counts the executed pairs of chain c; pair q may run exactly when
//      ticket[i]==r_i && ticket[j]==r_j, i.e. when all its predecessors on both chains ran
//      and none of its successors did: the sequential order of AlgoBGP.jl:662-691 per chain;
//   3. after running it publishes ticket+1 on both chains (release/acquire, workgroup scope).
// The critical path is the longest dependency chain (~ log N) times one LDS round trip.
constexpr int XLDS_MAX = 4096;
constexpr unsigned XSPIN_LIMIT = 1u << 22;

__global__ __launch_bounds__(XWG) void k_exch_resolve_lds(const KParams P, const int t, const double* __restrict__ gathered) {
    extern __shared__ __attribute__((aligned(16))) unsigned char xsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Ng = P.Ng, N = P.N, R = 3 + P.np + P.nm;
    const int K = P.pairtab ? P.n_pairs_tab : n_exchange_pairs(Ng);
    double* val = (double*)xsm;                       // [Ng]   (aliases ep[2K] during the build)
    uint32_t* ep = (uint32_t*)xsm;                    // [2K]
    uint32_t* cnt = (uint32_t*)(val + Ng);            // [Ng]   histogram -> cursor -> ticket
    uint16_t* src = (uint16_t*)(cnt + Ng);            // [Ng]
    uint16_t* partner = src + Ng;                     // [Ng]
    uint16_t* pi = partner + Ng;                      // [K]
    uint16_t* pj = pi + K;                            // [K]
    uint16_t* ri = pj + K;                            // [K]
    uint16_t* rj = ri + K;                            // [K]
    uint32_t* wsum = (uint32_t*)(rj + K);             // [16] (4 u16 arrays of K = 8K bytes: 4-byte aligned)

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
    // 1a. histogram of endpoints
    for (int q = tid; q < K; q += XWG) {
        atomicAdd(&cnt[pi[q]], 1u);
        atomicAdd(&cnt[pj[q]], 1u);
    }
    __syncthreads();
    // 1b. exclusive scan of cnt[0..Ng) -> cursor (segment start)
    {
        const int per = (Ng + XWG - 1) / XWG;  // consecutive entries per thread (<= 4)
        const int c0 = tid * per;
        uint32_t loc[XLDS_MAX / XWG];
        uint32_t sum = 0;
#pragma unroll
        for (int u = 0; u < XLDS_MAX / XWG; ++u) {
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
        for (int u = 0; u < XLDS_MAX / XWG; ++u) {
            const int c = c0 + u;
            if (u < per && c < Ng) cnt[c] = excl + loc[u];
        }
    }
    __syncthreads();
    // 1c. scatter list positions into the chain buckets (order inside a bucket is arbitrary)
    for (int q = tid; q < K; q += XWG) {
        ep[atomicAdd(&cnt[pi[q]], 1u)] = (uint32_t)q;
        ep[atomicAdd(&cnt[pj[q]], 1u)] = (uint32_t)q;
    }
    __syncthreads();  // now cnt[c] == end of chain c's bucket
    // 1d. rank of each pair inside the buckets of its two chains = number of smaller positions
    for (int q = tid; q < K; q += XWG) {
        const int i = pi[q], j = pj[q];
        uint32_t b = i ? cnt[i - 1] : 0u, e = cnt[i], r = 0;
        for (uint32_t x = b; x < e; ++x) r += (ep[x] < (uint32_t)q) ? 1u : 0u;
        ri[q] = (uint16_t)r;
        b = j ? cnt[j - 1] : 0u; e = cnt[j]; r = 0;
        for (uint32_t x = b; x < e; ++x) r += (ep[x] < (uint32_t)q) ? 1u : 0u;
        rj[q] = (uint16_t)r;
    }
    __syncthreads();
    // 2. state: values of the last accepted records, identity permutation, zero tickets
    for (int g = tid; g < Ng; g += XWG) {
        const int shard = g / N, l = g - shard * N;
        val[g] = gathered[(size_t)shard * R * N + l];
        src[g] = (uint16_t)g;
        partner[g] = 0;
        cnt[g] = 0;
    }
    __syncthreads();
    // 3. data-flow execution
    {
        int q = tid;
        double mi = (q < K) ? P.min_improve_g[pi[q]] : 0.0;
        unsigned spins = 0;
        while (true) {
            bool progressed = false;
            if (q < K) {
                const int i = pi[q], j = pj[q];
                const uint32_t ti = __hip_atomic_load(&cnt[i], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                const uint32_t tj = __hip_atomic_load(&cnt[j], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (ti == ri[q] && tj == rj[q]) {
                    const double vi = val[i], vj = val[j];
                    if (vi - vj > mi) {                         // dist_fun = -, AlgoBGP.jl:688
                        val[i] = vj; val[j] = vi;               // swap_ev_ij!, :739-744
                        const uint16_t si = src[i];
                        src[i] = src[j]; src[j] = si;
                        partner[i] = (uint16_t)(j + 1); partner[j] = (uint16_t)(i + 1);  // :747-748
                    }
                    __hip_atomic_store(&cnt[i], ti + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_store(&cnt[j], tj + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    q += XWG;
                    if (q < K) mi = P.min_improve_g[pi[q]];
                    progressed = true;
                }
            }
            if (__all(q >= K)) break;
            if (!__any(progressed)) {
                if (++spins > XSPIN_LIMIT) {  // cannot happen (the smallest pending position is always runnable)
                    if (lane == 0) report_error(P, 3, t, 0);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
    }
    __syncthreads();
    for (int g = tid; g < Ng; g += XWG) {
        P.xsrc[g] = src[g];
        P.xpartner[g] = partner[g];
    }
}

// k_exch_apply: for local chains, set_eval!(ci, ej) (+ set_exchanged!) of swap_ev_ij!
// (AlgoBGP.jl:734-749): the chain's record of iteration t is overwritten by the donor's last
// accepted record; curr/best are recomputed against iteration t-1 (:231-243).  Also closes the
// iteration's acceptance-rate counters (set_acceptRate!, :253-257).
__global__ void k_exch_apply(const KParams P, const int t, const double* __restrict__ gathered) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= P.N) return;
    const int N = P.N, np = P.np, nm = P.nm, R = 3 + np + nm;
    const int g = P.offset + c;
    const int partner = P.xpartner[g];
    const size_t row = (size_t)(t - 1) * N + c;
    if (partner == 0) {
        P.n_noex[c] += 1;
        P.n_acc[c] += P.h_acc[row];
        return;
    }
    const int s = P.xsrc[g];
    const int shard = s / N, l = s - shard * N;
    const double* rec = gathered + (size_t)shard * R * N;
    const double value = rec[l], prob = rec[(size_t)N + l];
    const int8_t status = (int8_t)rec[(size_t)2 * N + l];
    const size_t prow = (size_t)(t - 2) * N + c;
    const double bp = P.h_best[prow];
    double bestv; int bestid;
    if (value < bp) { bestv = value; bestid = t; }
    else { bestv = bp; bestid = P.h_best_id[prow]; }
    P.best_val[c] = bestv; P.best_id[c] = bestid;
    P.h_value[row] = value; P.h_prob[row] = prob; P.h_curr[row] = value; P.h_best[row] = bestv;
    P.h_best_id[row] = bestid; P.h_exch[row] = partner; P.h_acc[row] = 1; P.h_status[row] = status;
    P.la_value[c] = value; P.la_prob[c] = prob; P.la_status[c] = status;
    for (int k = 0; k < np; ++k) {
        const double v = rec[(size_t)(3 + k) * N + l];
        P.h_params[((size_t)(t - 1) * np + k) * N + c] = v;
        P.la_params[(size_t)k * N + c] = v;
    }
    for (int k = 0; k < nm; ++k) {
        const double v = rec[(size_t)(3 + np + k) * N + l];
        P.h_simM[((size_t)(t - 1) * nm + k) * N + c] = v;
        P.la_simM[(size_t)k * N + c] = v;
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
thread_local std::string g_create_err;

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
    bool profiling = false;
    bool force_generic_exchange = false;
    std::vector<hipEvent_t> pev;  // profiling events: 3 per iteration (before iter, after iter, after exchange)
    int pev_iters = 0;
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
template <class T>
void dfill(Ctx* c, T* d, size_t n, T v) {
    std::vector<T> h(n, v);
    if (n) HIPCHK(hipMemcpy(d, h.data(), n * sizeof(T), hipMemcpyHostToDevice));
}

bool is_sim(int obj) { return obj == SMM_OBJ_NORM || obj == SMM_OBJ_NORM_FAILBOX; }

size_t tile_smem(const Ctx* c, int ct) {
    return (size_t)(ct * c->P.np + ct * c->P.nm + (is_sim(c->obj) ? (WG / 64) * ct * c->P.nm : 0)) * sizeof(double);
}

void launch_chain_iter(Ctx* c, int t, int close_iter) {
    const KParams& P = c->P;
    if (is_sim(c->obj)) {
        constexpr int CT = 8;
        const int grid = (P.N + CT - 1) / CT;
        hipLaunchKernelGGL((k_chain_iter<true, CT>), dim3(grid), dim3(WG), tile_smem(c, CT), c->stream, P, t, close_iter);
    } else {
        constexpr int CT = 64;
        const int grid = (P.N + CT - 1) / CT;
        hipLaunchKernelGGL((k_chain_iter<false, CT>), dim3(grid), dim3(WG), tile_smem(c, CT), c->stream, P, t, close_iter);
    }
}

size_t xlds_bytes(int Ng, int K) {
    return (size_t)Ng * (8 + 4 + 2 + 2) + (size_t)(K + (K & 1)) * 2 * 4 + 16 * 4 + 16;
}

void launch_exchange(Ctx* c, int t, const double* gathered) {
    const KParams& P = c->P;
    const int K = P.pairtab ? P.n_pairs_tab : n_exchange_pairs(P.Ng);
    if (P.Ng <= XLDS_MAX && K <= P.Ng && !c->force_generic_exchange)
        hipLaunchKernelGGL(k_exch_resolve_lds, dim3(1), dim3(XWG), xlds_bytes(P.Ng, K), c->stream, P, t, gathered);
    else
        hipLaunchKernelGGL(k_exch_resolve, dim3(1), dim3(XWG), 0, c->stream, P, t, gathered);
    hipLaunchKernelGGL(k_exch_apply, dim3((P.N + 255) / 256), dim3(256), 0, c->stream, P, t, gathered);
}

bool exchange_active(const Ctx* c, int t) { return t >= c->exchange_from && c->P.Ng > 1; }  // AlgoBGP.jl:637

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
             c->P.ntab ? (c->P.ntries < c->P.smpl_iters ? c->P.ntries : c->P.smpl_iters) : c->P.smpl_iters, chain + 1, it);
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
    if (is_sim(prob->objective_id) && np != nm)
        return fail(nullptr, SMM_ERR_INVALID_ARG, "objfunc_norm needs one moment per parameter (ObjExamples.jl:66-78)");
    if (prob->objective_id < 0 || prob->objective_id > SMM_OBJ_NORM_FAILBOX)
        return fail(nullptr, SMM_ERR_INVALID_ARG, "unknown objective_id");
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
        c->obj = prob->objective_id;
        c->exchange_from = opts->exchange_from_iter;
        {
            const char* e = getenv("SMMHIP_GENERIC_EXCHANGE");  // test hook: force the any-size resolution kernel
            c->force_generic_exchange = e && e[0] == '1';
        }
        P.np = np; P.nm = nm; P.ns = ns; P.obj = prob->objective_id;
        P.init = dupload(c, prob->init, np); P.lb = dupload(c, prob->lb, np); P.ub = dupload(c, prob->ub, np);
        P.mom = dupload(c, prob->mom, nm); P.w = dupload(c, prob->w, nm);
        P.objp = prob->n_obj_params > 0 ? dupload(c, prob->obj_params, prob->n_obj_params) : nullptr;
        {
            std::vector<double> Z((size_t)nm * ns);
            if (tab && tab->Z) memcpy(Z.data(), tab->Z, Z.size() * sizeof(double));
            else
                for (int k = 0; k < nm; ++k)
                    for (int s = 0; s < ns; ++s) Z[(size_t)k * ns + s] = rng_Z(opts->seed, (uint32_t)k, (uint32_t)s);
            P.Z = dupload(c, Z.data(), Z.size());
        }
        P.N = N; P.Ng = Ng; P.offset = opts->chain_offset; P.T = T;
        P.sigma_update_steps = opts->sigma_update_steps; P.smpl_iters = opts->smpl_iters;
        P.batch_size = opts->batch_size; P.sigma_adjust_by = opts->sigma_adjust_by; P.seed = opts->seed;
        P.acc_tuner_g = dupload(c, opts->acc_tuner, Ng); P.min_improve_g = dupload(c, opts->min_improve, Ng);
        const size_t TN = (size_t)T * N;
        P.utab = (tab && tab->probs_acc) ? dupload(c, tab->probs_acc, TN) : nullptr;
        if (tab && tab->prop_normals && tab->prop_tries > 0) {
            P.ntries = tab->prop_tries;
            P.ntab = dupload(c, tab->prop_normals, TN * (size_t)tab->prop_tries * np);
        }
        if (tab && tab->pairs && tab->n_pairs > 0) {
            P.n_pairs_tab = tab->n_pairs;
            P.pairtab = dupload(c, tab->pairs, (size_t)T * tab->n_pairs * 2);
        }
        P.sigma = dupload(c, opts->sigma + opts->chain_offset, N);
        P.accept_rate = dalloc<double>(c, N); dfill(c, P.accept_rate, N, 0.0);
        P.la_value = dalloc<double>(c, N); dfill(c, P.la_value, N, (double)INFINITY);
        P.la_prob = dalloc<double>(c, N); dfill(c, P.la_prob, N, 0.0);
        P.la_params = dalloc<double>(c, (size_t)np * N); dfill(c, P.la_params, (size_t)np * N, 0.0);
        P.la_simM = dalloc<double>(c, (size_t)nm * N); dfill(c, P.la_simM, (size_t)nm * N, 0.0);
        P.la_status = dalloc<int8_t>(c, N); dfill(c, P.la_status, N, (int8_t)0);
        P.n_noex = dalloc<int32_t>(c, N); dfill(c, P.n_noex, N, 0);
        P.n_acc = dalloc<int32_t>(c, N); dfill(c, P.n_acc, N, 0);
        P.best_val = dalloc<double>(c, N); dfill(c, P.best_val, N, (double)INFINITY);
        P.best_id = dalloc<int32_t>(c, N); dfill(c, P.best_id, N, -1);
        P.rec = dalloc<double>(c, (size_t)(3 + np + nm) * N);
        const int Kmax = std::max(std::max(n_exchange_pairs(Ng), P.n_pairs_tab), 1);
        P.xval = dalloc<double>(c, Ng); P.xsrc = dalloc<int32_t>(c, Ng); P.xpartner = dalloc<int32_t>(c, Ng);
        P.xnext = dalloc<int32_t>(c, Ng); P.xpairs = dalloc<int32_t>(c, (size_t)Kmax * 2);
        P.h_value = dalloc<double>(c, TN); dfill(c, P.h_value, TN, (double)NAN);
        P.h_prob = dalloc<double>(c, TN); dfill(c, P.h_prob, TN, (double)NAN);
        P.h_curr = dalloc<double>(c, TN); dfill(c, P.h_curr, TN, (double)INFINITY);
        P.h_best = dalloc<double>(c, TN); dfill(c, P.h_best, TN, (double)INFINITY);
        P.h_params = dalloc<double>(c, TN * np); dfill(c, P.h_params, TN * np, (double)NAN);
        P.h_simM = dalloc<double>(c, TN * nm); dfill(c, P.h_simM, TN * nm, (double)NAN);
        P.h_best_id = dalloc<int32_t>(c, TN); dfill(c, P.h_best_id, TN, -1);
        P.h_exch = dalloc<int32_t>(c, TN); dfill(c, P.h_exch, TN, 0);
        P.h_acc = dalloc<uint8_t>(c, TN); dfill(c, P.h_acc, TN, (uint8_t)0);
        P.h_status = dalloc<int8_t>(c, TN); dfill(c, P.h_status, TN, (int8_t)0);
        P.err = dalloc<unsigned long long>(c, 1); dfill(c, P.err, 1, ERR_NONE);
        HIPCHK(hipFuncSetAttribute((const void*)k_exch_resolve_lds, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)xlds_bytes(XLDS_MAX, XLDS_MAX)));
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
            for (int i = 0; i < c->pev_iters; ++i) {
                float a = 0.f, b = 0.f;
                HIPCHK(hipEventElapsedTime(&a, c->pev[3 * i], c->pev[3 * i + 1]));
                HIPCHK(hipEventElapsedTime(&b, c->pev[3 * i + 1], c->pev[3 * i + 2]));
                c->timing.iter_kernel_ms += a;
                c->timing.exch_kernel_ms += b;
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
            while ((int)c->pev.size() < 3 * n_iters) {
                hipEvent_t e;
                HIPCHK(hipEventCreate(&e));
                c->pev.push_back(e);
            }
        }
        c->pev_iters = 0;
        HIPCHK(hipEventRecord(c->ev0, c->stream));
        for (int it = 0; it < n_iters; ++it) {
            const int t = c->iter + 1;
            const bool ex = exchange_active(c, t);
            if (c->profiling) HIPCHK(hipEventRecord(c->pev[3 * it], c->stream));
            launch_chain_iter(c, t, ex ? 0 : 1);
            if (c->profiling) HIPCHK(hipEventRecord(c->pev[3 * it + 1], c->stream));
            if (ex) launch_exchange(c, t, c->P.rec);
            if (c->profiling) HIPCHK(hipEventRecord(c->pev[3 * it + 2], c->stream));
            c->iter = t;
        }
        if (c->profiling) c->pev_iters = n_iters;
        HIPCHK(hipEventRecord(c->ev1, c->stream));
        HIPCHK(hipGetLastError());
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
    if (c->iter + 1 > c->P.T) return fail(c, SMM_ERR_MAXITER, "step beyond maxiter (history capacity)");
    try {
        HIPCHK(hipSetDevice(c->device));
        const int t = c->iter + 1;
        launch_chain_iter(c, t, exchange_active(c, t) ? 0 : 1);
        HIPCHK(hipGetLastError());
        c->iter = t;
    } catch (const std::string& m) {
        return fail(c, SMM_ERR_HIP, m);
    }
    return SMM_OK;
}

int smm_bgp_record_doubles(void* ctx) {
    Ctx* c = (Ctx*)ctx;
    return c ? 3 + c->P.np + c->P.nm : SMM_ERR_INVALID_ARG;
}

int smm_bgp_export_records_dev(void* ctx, void* rec_dev) {
    Ctx* c = (Ctx*)ctx;
    if (!c || !rec_dev) return SMM_ERR_INVALID_ARG;
    try {
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(hipMemcpyAsync(rec_dev, c->P.rec, (size_t)(3 + c->P.np + c->P.nm) * c->P.N * sizeof(double),
                              hipMemcpyDeviceToDevice, c->stream));
    } catch (const std::string& m) {
        return fail(c, SMM_ERR_HIP, m);
    }
    return SMM_OK;
}

int smm_bgp_exchange_dev(void* ctx, const void* gathered_dev) {
    Ctx* c = (Ctx*)ctx;
    if (!c || !gathered_dev) return SMM_ERR_INVALID_ARG;
    if (c->iter < 1) return fail(c, SMM_ERR_STATE, "exchange before the first local step");
    try {
        HIPCHK(hipSetDevice(c->device));
        if (exchange_active(c, c->iter)) launch_exchange(c, c->iter, (const double*)gathered_dev);
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
        HIPCHK(hipMemcpyAsync(dp, params, (size_t)P.np * M * 8, hipMemcpyHostToDevice, c->stream));
        if (is_sim(c->obj)) {
            constexpr int CT = 8;
            hipLaunchKernelGGL((k_eval_batch<true, CT>), dim3((M + CT - 1) / CT), dim3(WG), tile_smem(c, CT), c->stream, P, dp, M, dv, dm, ds);
        } else {
            constexpr int CT = 64;
            hipLaunchKernelGGL((k_eval_batch<false, CT>), dim3((M + CT - 1) / CT), dim3(WG), tile_smem(c, CT), c->stream, P, dp, M, dv, dm, ds);
        }
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

#define D2H(dst, src, n, sz) do { if (dst) HIPCHK(hipMemcpy(dst, src, (size_t)(n) * (sz), hipMemcpyDeviceToHost)); } while (0)
#define H2D(dst, src, n, sz) do { if (src) HIPCHK(hipMemcpy(dst, src, (size_t)(n) * (sz), hipMemcpyHostToDevice)); } while (0)

int smm_get_history(void* ctx, int32_t t0, int32_t t1, smm_history_t* out) {
    Ctx* c = (Ctx*)ctx;
    if (!c || !out || t0 < 0 || t1 < t0 || t1 > c->P.T) return SMM_ERR_INVALID_ARG;
    try {
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(hipStreamSynchronize(c->stream));
        const KParams& P = c->P;
        const size_t N = P.N, nt = (size_t)(t1 - t0), off = (size_t)t0 * N;
        D2H(out->value, P.h_value + off, nt * N, 8); D2H(out->prob, P.h_prob + off, nt * N, 8);
        D2H(out->curr_val, P.h_curr + off, nt * N, 8); D2H(out->best_val, P.h_best + off, nt * N, 8);
        D2H(out->params, P.h_params + off * P.np, nt * N * P.np, 8);
        D2H(out->sim_moments, P.h_simM + off * P.nm, nt * N * P.nm, 8);
        D2H(out->best_id, P.h_best_id + off, nt * N, 4); D2H(out->exchanged, P.h_exch + off, nt * N, 4);
        D2H(out->accepted, P.h_acc + off, nt * N, 1); D2H(out->status, P.h_status + off, nt * N, 1);
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
        HIPCHK(hipStreamSynchronize(c->stream));
        const KParams& P = c->P;
        const size_t N = P.N;
        s->iter = c->iter;
        D2H(s->sigma, P.sigma, N, 8); D2H(s->accept_rate, P.accept_rate, N, 8);
        D2H(s->la_value, P.la_value, N, 8); D2H(s->la_prob, P.la_prob, N, 8);
        D2H(s->la_params, P.la_params, N * P.np, 8); D2H(s->la_sim_moments, P.la_simM, N * P.nm, 8);
        D2H(s->la_status, P.la_status, N, 1); D2H(s->n_noex, P.n_noex, N, 4); D2H(s->n_acc_noex, P.n_acc, N, 4);
        D2H(s->best_val, P.best_val, N, 8); D2H(s->best_id, P.best_id, N, 4);
    } catch (const std::string& m) {
        return fail(c, SMM_ERR_HIP, m);
    }
    return SMM_OK;
}

int smm_set_state(void* ctx, const smm_state_t* s, const smm_history_t* h) {
    Ctx* c = (Ctx*)ctx;
    if (!c || !s || s->iter < 0 || s->iter > c->P.T) return SMM_ERR_INVALID_ARG;
    if (s->iter > 0 && !h) return fail(c, SMM_ERR_INVALID_ARG, "history of iterations 0..iter-1 required");
    try {
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(hipStreamSynchronize(c->stream));
        KParams& P = c->P;
        const size_t N = P.N, nt = (size_t)s->iter;
        H2D(P.sigma, s->sigma, N, 8); H2D(P.accept_rate, s->accept_rate, N, 8);
        H2D(P.la_value, s->la_value, N, 8); H2D(P.la_prob, s->la_prob, N, 8);
        H2D(P.la_params, s->la_params, N * P.np, 8); H2D(P.la_simM, s->la_sim_moments, N * P.nm, 8);
        H2D(P.la_status, s->la_status, N, 1); H2D(P.n_noex, s->n_noex, N, 4); H2D(P.n_acc, s->n_acc_noex, N, 4);
        H2D(P.best_val, s->best_val, N, 8); H2D(P.best_id, s->best_id, N, 4);
        if (h && nt) {
            H2D(P.h_value, h->value, nt * N, 8); H2D(P.h_prob, h->prob, nt * N, 8);
            H2D(P.h_curr, h->curr_val, nt * N, 8); H2D(P.h_best, h->best_val, nt * N, 8);
            H2D(P.h_params, h->params, nt * N * P.np, 8); H2D(P.h_simM, h->sim_moments, nt * N * P.nm, 8);
            H2D(P.h_best_id, h->best_id, nt * N, 4); H2D(P.h_exch, h->exchanged, nt * N, 4);
            H2D(P.h_acc, h->accepted, nt * N, 1); H2D(P.h_status, h->status, nt * N, 1);
        }
        c->iter = s->iter;
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
    c->profiling = on != 0;
    return SMM_OK;
}

int smm_get_Z(void* ctx, double* Z) {
    Ctx* c = (Ctx*)ctx;
    if (!c || !Z) return SMM_ERR_INVALID_ARG;
    try {
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(hipMemcpy(Z, c->P.Z, (size_t)c->P.nm * c->P.ns * sizeof(double), hipMemcpyDeviceToHost));
    } catch (const std::string& m) {
        return fail(c, SMM_ERR_HIP, m);
    }
    return SMM_OK;
}

}  // extern "C"
