/*
 * smm_oracle.c — CPU restatement (plain C) of the BGP parallel-tempering hot path of
 * floswald/SMM.jl.  TEST INFRASTRUCTURE ONLY: nothing in the product (smm.jl_amd/,
 * libsmmhip.so) may link, import or call this file.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() use it, as the checker.
 *
 * PARITY STATUS: "parity unpinned" against the reference's own numbers.  The reference is
 * Julia (no julia binary in the build container), its proposals draw from a non-seedable
 * RandomDevice (src/SMM.jl:60, src/mopt/AlgoBGP.jl:404) and its tests hold no known-answer
 * vector for this path.  What IS pinned here:
 *   - the behavioural properties P1..P7 of test/test_BGPchain.jl, test/test_objfunc.jl,
 *     test/test_algoBGP.jl (lifted into tests/test_oracle_properties.py),
 *   - the analytic anchor of ObjExamples.jl:90-101 (Z == 0 => value = mean(((mu-mom)/w)^2)),
 *   - Philox4x32-10 known-answer vectors (Random123 kat_vectors).
 * The pin that would lift this status exists as a script nobody here could run: julia/reference_golden.jl drives the
 * reference's OWN doAcceptReject! / set_eval! / exchangeMoves! / objfunc_norm with injected randomness and writes
 * tests/golden/ref_bgp.json + ref_Z.bin; tests/test_golden.py::test_oracle_matches_reference_vectors replays them through
 * this file (skipped until a maintainer with a julia binary commits those two files).
 * Every function cites the reference lines it restates (paths relative to /root/reference).
 *
 * All randomness is either INJECTED (tables) or drawn from a counter-based generator
 * (Philox4x32-10 + Box-Muller) whose definition is restated here independently of the
 * HIP library.  Third-party arithmetic that the reference takes from Distributions.jl /
 * Random (MvNormal rand, sample(..;replace=false), randn) cannot be reproduced bit-for-bit
 * (unpinned dependency versions, Project.toml:28-29); only its distributional meaning is.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_OK 0
#define ORC_ERR_INVALID_ARG (-1)
#define ORC_ERR_NEGATIVE_OBJECTIVE (-3)
#define ORC_ERR_NO_DRAW_IN_SUPPORT (-4)
#define ORC_ERR_BAD_BATCH (-5)
#define ORC_ERR_MAXITER (-6)

#define ORC_OBJ_NORM 0
#define ORC_OBJ_BANANA 1
#define ORC_OBJ_NORM_FAILBOX 2
#define ORC_OBJ_DENSE 3
#define ORC_OBJ_DENSE2 5   /* spec v2: with the 256 x 256 stage (include/smmhip.h) */
#define ORC_OBJ_USER_BASE 1000
#define ORC_MAX_USER 64
typedef void (*orc_user_fn)(const double* theta, int np, const double* mom, const double* w, int nm, const double* udata,
                            int n_udata, double* sim_moments, double* value, int* status);
static orc_user_fn g_user_fn[ORC_MAX_USER];
/* map-reduce form (SMM_USER_PARTIAL / SMM_USER_FINISH, include/smmhip.h) */
typedef void (*orc_user_partial_fn)(const double* theta, int np, const double* udata, int n_udata, int lane, int n_lanes,
                                    double* partial);
typedef void (*orc_user_finish_fn)(const double* theta, int np, const double* totals, int n_sums, const double* mom,
                                   const double* w, int nm, const double* udata, int n_udata, double* sim_moments,
                                   double* value, int* status);
static orc_user_partial_fn g_user_partial[ORC_MAX_USER];
static orc_user_finish_fn g_user_finish[ORC_MAX_USER];
static int g_user_nsums[ORC_MAX_USER], g_user_lanes[ORC_MAX_USER];
#define ORC_DENSE_D 256

#define ORC_REDUCE_LANES 512 /* numerical contract, see include/smmhip.h */

/* same memory layout as smm_problem_t / smm_bgp_opts_t / smm_tables_t / smm_history_t /
 * smm_state_t of include/smmhip.h so that one set of ctypes classes drives both. */
typedef struct {
    int32_t np, nm, ns, objective_id;
    const double *init, *lb, *ub, *mom, *w, *obj_params;
    int32_t n_obj_params, reserved;
} orc_problem_t;

typedef struct {
    int32_t N, maxiter;
    const double *sigma, *acc_tuner, *min_improve; /* [N_global] each */
    int32_t sigma_update_steps, smpl_iters;
    double sigma_adjust_by;
    int32_t batch_size, exchange_from_iter;
    uint64_t seed;
    int32_t chain_offset, N_global, device, chol_per_chain;
    const double* chol_L; /* general Gaussian proposals, include/smmhip.h: NULL = the reference's MvNormal(mu01, sigma) */
    int32_t dist_fun, reserved;   /* smm_dist_fun_t (include/smmhip.h): opts["dist_fun"], AlgoBGP.jl:537 */
} orc_opts_t;

/* algo.dist_fun(evi.value, evj.value), AlgoBGP.jl:688: the default `-` (:537) and the header's two other menu entries */
static double orc_dist_fun(int kind, double a, double b) {
    const double d = a - b;
    if (kind == 0) return d;
    if (kind == 1) return fabs(d);
    return d / fabs(a);
}

typedef struct {
    const double* probs_acc;
    const double* prop_normals;
    int32_t prop_tries, n_pairs;
    const int32_t* pairs;
    const double* Z;
} orc_tables_t;

typedef struct {
    double *value, *prob, *curr_val, *best_val, *params, *sim_moments;
    int32_t *best_id, *exchanged;
    uint8_t* accepted;
    int8_t* status;
} orc_history_t;

typedef struct {
    int32_t iter, reserved;
    double *sigma, *accept_rate, *la_value, *la_prob, *la_params, *la_sim_moments;
    int8_t* la_status;
    int32_t *n_noex, *n_acc_noex;
    double* best_val;
    int32_t* best_id;
} orc_state_t;

/* ------------------------------------------------------------------------------------ */
/* Counter-based RNG: Philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11).                 */
/* ------------------------------------------------------------------------------------ */
static void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void orc_philox4x32_10(const uint32_t* ctr, const uint32_t* key, uint32_t* out) { philox4x32_10(ctr, key, out); }

enum { STREAM_U = 1, STREAM_PROP = 2, STREAM_Z = 3, STREAM_PAIRS = 4 };

static void stream_key(uint64_t seed, uint32_t stream, uint32_t key[2]) {
    key[0] = (uint32_t)seed;
    key[1] = (uint32_t)(seed >> 32) ^ (stream * 0x9E3779B9u);
}

/* uniform in [0,1): top 53 bits of a 64-bit word (the range of Julia's rand(), AlgoBGP.jl:85) */
static double u53(uint32_t hi, uint32_t lo) {
    uint64_t w = ((uint64_t)hi << 32) | lo;
    return (double)(w >> 11) * 0x1.0p-53;
}
/* uniform in (0,1] for the log of Box-Muller */
static double u53_open0(uint32_t hi, uint32_t lo) {
    uint64_t w = ((uint64_t)hi << 32) | lo;
    return (double)((w >> 11) + 1) * 0x1.0p-53;
}
/* The elementary functions of the path are part of its numerical contract (include/smmhip.h; the device's copies: smm_rng.hpp): plain
 * sequences of correctly rounded operations, compiled without contraction, each within 1 ulp of the true value (sine / cosine: 2^-53
 * absolute) — tests/test_oracle_properties.py compares them with libm's.  After fdlibm's e_log.c, k_sin.c, k_cos.c, e_exp.c. */
static double smm_log(const double x) {   /* a positive normal double */
    uint64_t b;
    memcpy(&b, &x, 8);
    int e = (int)((b >> 52) & 0x7ffu) - 1023;
    const uint64_t mb = (b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m;
    memcpy(&m, &mb, 8);
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s, w = z * z;
    const double t1 = w * (3.999999999940941908e-01 + w * (2.222219843214978396e-01 + w * 1.531383769920937332e-01));
    const double t2 = z * (6.666666666666735130e-01 + w * (2.857142874366239149e-01 + w * (1.818357216161805012e-01 + w * 1.479819860511658591e-01)));
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)e;
    return dk * 6.93147180369123816490e-01 - ((hfsq - (s * (hfsq + R) + dk * 1.90821492927058770002e-10)) - f);
}
static void smm_sincos2pi(const double u, double* sn, double* cs) {   /* u in [0, 1) */
    const double t = 4.0 * u;
    const double q = rint(t);
    const double r = t - q;
    const double x = r * 1.57079632679489655800e+00 + r * 6.12323399573676603587e-17;
    const double z = x * x;
    const double v = z * x;
    const double rs = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
    const double s = x + v * (-1.66666666666666324348e-01 + z * rs);
    const double rc = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 + z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    const double c = w + (((1.0 - w) - hz) + z * rc);
    const int qi = (int)q & 3;
    *sn = qi == 0 ? s : (qi == 1 ? c : (qi == 2 ? -s : -c));
    *cs = qi == 0 ? c : (qi == 1 ? -s : (qi == 2 ? -c : s));
}
static double smm_exp(const double x) {
    if (x != x) return x;
    if (x > 709.782712893383973096) return INFINITY;
    if (x < -745.13321910194110842) return 0.0;
    const double k = rint(x * 1.44269504088896338700e+00);
    double r = fma(-k, 6.93147180369123816490e-01, x);
    r = fma(-k, 1.90821492927058770002e-10, r);
    /* exp(r) - 1 = r + r^2 q(r), q = the Taylor coefficients 1/2! .. 1/13! (|r| <= ln2 / 2: 4e-18), Estrin's scheme: four levels of fma */
    const double r2 = r * r, r4 = r2 * r2, r8 = r4 * r4;
    const double a0 = fma(1.0 / 6.0, r, 0.5), a1 = fma(1.0 / 120.0, r, 1.0 / 24.0), a2 = fma(1.0 / 5040.0, r, 1.0 / 720.0);
    const double a3 = fma(1.0 / 362880.0, r, 1.0 / 40320.0), a4 = fma(1.0 / 39916800.0, r, 1.0 / 3628800.0), a5 = fma(1.0 / 6227020800.0, r, 1.0 / 479001600.0);
    const double b0 = fma(a1, r2, a0), b1 = fma(a3, r2, a2), b2 = fma(a5, r2, a4);
    const double q = fma(b2, r8, fma(b1, r4, b0));
    const double p = fma(r2, q, r);
    return ldexp(1.0 + p, (int)k);
}
/* (exported for the tests) what: 0 log, 1 exp, 2 sin(2 pi x), 3 cos(2 pi x) */
void orc_math(int what, const double* x, double* y, int n) {
    for (int i = 0; i < n; ++i) {
        double s, c;
        if (what == 0) y[i] = smm_log(x[i]);
        else if (what == 1) y[i] = smm_exp(x[i]);
        else { smm_sincos2pi(x[i], &s, &c); y[i] = what == 2 ? s : c; }
    }
}

/* Box-Muller: two standard normals from one Philox block */
static void box_muller(const uint32_t x[4], double z[2]) {
    double u1 = u53_open0(x[0], x[1]);
    double u2 = u53(x[2], x[3]);
    double r = sqrt(-2.0 * smm_log(u1));
    double s, c;
    smm_sincos2pi(u2, &s, &c);
    z[0] = r * c;
    z[1] = r * s;
}

/* MH uniform of (global chain c, iteration t>=1): one entry of probs_acc = rand(n), AlgoBGP.jl:85 */
static double rng_u(uint64_t seed, uint32_t c, uint32_t t) {
    uint32_t key[2], ctr[4] = {c, t, 0, 0}, x[4];
    stream_key(seed, STREAM_U, key);
    philox4x32_10(ctr, key, x);
    return u53(x[0], x[1]);
}
/* standard normal for (chain c, iteration t, try r, parameter k): rand(RAND,d), AlgoBGP.jl:404 */
static double rng_prop_normal(uint64_t seed, uint32_t c, uint32_t t, uint32_t r, uint32_t k) {
    uint32_t key[2], ctr[4] = {c, t, r, k >> 1}, x[4];
    double z[2];
    stream_key(seed, STREAM_PROP, key);
    philox4x32_10(ctr, key, x);
    box_muller(x, z);
    return z[k & 1];
}
/* shock z[k][s] of the objective: the seed-1234 draw matrix of ObjExamples.jl:74-79 */
static double rng_Z(uint64_t seed, uint32_t k, uint32_t s) {
    uint32_t key[2], ctr[4] = {s, k >> 1, 0, 0}, x[4];
    double z[2];
    stream_key(seed, STREAM_Z, key);
    philox4x32_10(ctr, key, x);
    box_muller(x, z);
    return z[k & 1];
}
void orc_gen_Z(uint64_t seed, int nm, int ns, double* Z) {
    for (int k = 0; k < nm; ++k)
        for (int s = 0; s < ns; ++s) Z[(size_t)k * ns + s] = rng_Z(seed, (uint32_t)k, (uint32_t)s);
}

/* ------------------------------------------------------------------------------------ */
/* Exchange-pair sampling: K distinct pairs, uniformly, in random order, out of           */
/* props = [(i,j) for i in 1:N, j in 1:N if i<j]; sample(props,K,replace=false)           */
/* (AlgoBGP.jl:653-656).  Restated as a keyed bijection (6-round Feistel network with     */
/* cycle walking) of the linear pair index evaluated at 0..K-1, so that no O(N^2) array   */
/* is ever materialised.  Linear index m <-> (i,j), i<j (0-based): m = j(j-1)/2 + i,       */
/* i.e. i runs fastest, as in the reference's comprehension order.                        */
/* ------------------------------------------------------------------------------------ */
static uint32_t fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
typedef struct { uint32_t k[6]; uint32_t half_bits; uint64_t M; } pair_perm_t;

static void pair_perm_init(pair_perm_t* p, uint64_t seed, uint32_t t, uint64_t M) {
    uint32_t key[2], x[4];
    stream_key(seed, STREAM_PAIRS, key);
    uint32_t c0[4] = {t, 0, 0, 0}, c1[4] = {t, 1, 0, 0};
    philox4x32_10(c0, key, x);
    p->k[0] = x[0]; p->k[1] = x[1]; p->k[2] = x[2]; p->k[3] = x[3];
    philox4x32_10(c1, key, x);
    p->k[4] = x[0]; p->k[5] = x[1];
    uint32_t bits = 0;
    while (bits < 62 && ((uint64_t)1 << bits) < M) ++bits; /* 2^bits >= M */
    p->half_bits = (bits + 1) / 2;
    if (p->half_bits == 0) p->half_bits = 1;
    p->M = M;
}
static uint64_t pair_perm_eval(const pair_perm_t* p, uint64_t x) {
    const uint32_t h = p->half_bits;
    const uint32_t mask = (h >= 32) ? 0xFFFFFFFFu : ((1u << h) - 1u);
    do {
        uint32_t L = (uint32_t)(x >> h) & mask, R = (uint32_t)x & mask;
        for (int r = 0; r < 6; ++r) {
            uint32_t F = fmix32(R + p->k[r]) & mask;
            uint32_t nL = R;
            R = L ^ F;
            L = nL;
        }
        x = ((uint64_t)L << h) | R;
    } while (x >= p->M);
    return x;
}
static void pair_unrank(uint64_t m, int32_t* i, int32_t* j) {
    uint64_t jj = (uint64_t)((1.0 + sqrt(1.0 + 8.0 * (double)m)) * 0.5);
    while (jj * (jj - 1) / 2 > m) --jj;
    while ((jj + 1) * jj / 2 <= m) ++jj;
    *j = (int32_t)jj;
    *i = (int32_t)(m - jj * (jj - 1) / 2);
}
/* number of exchange proposals per iteration: AlgoBGP.jl:655 */
static int n_exchange_pairs(int N) { return N < 3 ? N - 1 : N; }

void orc_gen_pairs(uint64_t seed, int32_t t /*1-based iteration*/, int32_t Ng, int32_t* pairs /*[K][2]*/) {
    int K = n_exchange_pairs(Ng);
    if (K <= 0) return;
    pair_perm_t p;
    pair_perm_init(&p, seed, (uint32_t)t, (uint64_t)Ng * (uint64_t)(Ng - 1) / 2);
    for (int q = 0; q < K; ++q) {
        uint64_t m = pair_perm_eval(&p, (uint64_t)q);
        pair_unrank(m, &pairs[2 * q], &pairs[2 * q + 1]);
    }
}

/* ------------------------------------------------------------------------------------ */
/* Objectives                                                                             */
/* ------------------------------------------------------------------------------------ */

/* canonical reduction of SMM_REDUCE_LANES partial sums (see include/smmhip.h) */
static double reduce_partials(double* p /*[512], clobbered*/) {
    double tot = 0.0;
    for (int g = 0; g < ORC_REDUCE_LANES / 64; ++g) {
        double* q = p + 64 * g;
        for (int off = 32; off >= 1; off >>= 1)
            for (int i = 0; i < off; ++i) q[i] = q[i] + q[i + off];
        tot = (g == 0) ? q[0] : tot + q[0];
    }
    return tot;
}

/* objfunc_norm, ObjExamples.jl:59-116.
 *   mu = params (:66); X[k,s] = mu_k + 1.0*z[k,s] (MvNormal(mu,PDiagMat(ones)), :76-78;
 *   multiplying by sqrt(1.0) is exact so x = z + mu in one rounding);
 *   simM = mean(X,dims=2) (:79); v_k = ((simM_k - mom_k)/w_k)^2, or without the division
 *   when the moment has no weight (:90-100); value = mean(v) (:101); status = 1 (:110).
 *   Z == the draws after Random.seed!(1234) (:74): identical for every evaluation.
 *   regen != 0: regenerate every z from the counter RNG inside the evaluation (what the
 *   reference's CPU path pays for: 20000 randn per call), bit-identical to a cached Z. */
static void objfunc_norm(int nm, int ns, const double* theta, const double* Z, int regen, uint64_t seed,
                         const double* mom, const double* w, double* simM, double* value, int8_t* status) {
    double part[ORC_REDUCE_LANES];
    double vsum = 0.0;
    double* zbuf = NULL;
    if (regen) {
        /* the reference draws its ns x nm normals inside every evaluation (ObjExamples.jl:74-79).  Same draws as the cached
         * matrix: one Philox block + Box-Muller yields the shocks of moments 2q and 2q+1 of draw s (both outputs are used). */
        zbuf = (double*)malloc((size_t)nm * ns * sizeof(double));
        uint32_t key[2];
        stream_key(seed, STREAM_Z, key);
        for (int q = 0; 2 * q < nm; ++q)
            for (int s = 0; s < ns; ++s) {
                uint32_t ctr[4] = {(uint32_t)s, (uint32_t)q, 0, 0}, x[4];
                double z[2];
                philox4x32_10(ctr, key, x);
                box_muller(x, z);
                zbuf[(size_t)(2 * q) * ns + s] = z[0];
                if (2 * q + 1 < nm) zbuf[(size_t)(2 * q + 1) * ns + s] = z[1];
            }
        Z = zbuf;
    }
    for (int k = 0; k < nm; ++k) {
        const double mu = theta[k];
        const double* zk = Z + (size_t)k * ns;
        /* lane l sums its draws l, l+512, ... in that order (numerical contract); walking the draws row by row keeps the 512
         * running sums contiguous, so that the compiler vectorises the inner loop (independent accumulators: same bits) */
        for (int l = 0; l < ORC_REDUCE_LANES; ++l) part[l] = 0.0;
        for (int base = 0; base < ns; base += ORC_REDUCE_LANES) {
            const int n = ns - base < ORC_REDUCE_LANES ? ns - base : ORC_REDUCE_LANES;
            for (int l = 0; l < n; ++l) {
                const double x = zk[base + l] + mu;
                part[l] = part[l] + x;
            }
        }
        double tot = reduce_partials(part);
        simM[k] = tot / (double)ns;
        double d = simM[k] - mom[k];
        if (!isnan(w[k])) d = d / w[k];
        double v = d * d;
        vsum = (k == 0) ? v : vsum + v;
    }
    free(zbuf);
    *value = vsum / (double)nm;
    *status = 1;
}

/* banana, ObjExamples.jl:251-265: value = 100*(b-a^2)^2 + (1-a)^2 (:255), simMoments_k =
 * dataMoment_k + 2.2 (:259-261).  The reference function cannot run (String keys on a Symbol
 * dict, never sets status); SURVEY.md §8d C4 defines the generalisation to np dimensions:
 * value = sum_{i<np-1} 100*(x_{i+1}-x_i^2)^2 + (1-x_i)^2, status=1.  PARITY UNPINNED. */
static void objfunc_banana(int np, int nm, const double* theta, const double* mom, double* simM, double* value,
                           int8_t* status) {
    double v = 0.0;
    for (int i = 0; i + 1 < np; ++i) {
        double a = theta[i], b = theta[i + 1];
        double t1 = b - a * a;
        double t2 = 1.0 - a;
        double term = 100.0 * (t1 * t1) + t2 * t2;
        v = (i == 0) ? term : v + term;
    }
    for (int k = 0; k < nm; ++k) simM[k] = mom[k] + 2.2;
    *value = v;
    *status = 1;
}

/* the hidden layer's tanh of the dense simulation, as include/smmhip.h freezes it: E = exp(2|x|) = 2^n (1 + p), p = expm1(r) on
 * |r| <= ln2 / 2 by its Taylor series to r^13, tanh = (E - 1) / (E + 1) with E -+ 1 = fma(2^n, p, 2^n -+ 1); at most 3 ulp from the
 * true value (tests/test_oracle_properties.py compares it with libm's).  Only correctly rounded operations: the device evaluates the
 * same expression (smm_chain.hpp: smm_tanh) to the same bits. */
static double smm_tanh(const double x) {
    const double ax = fabs(x);
    const double z = ax + ax;
    const double zc = z > 40.0 ? 40.0 : z;
    const double n = rint(zc * 1.44269504088896338700e+00);
    double r = fma(-n, 6.93147180369123816490e-01, zc);
    r = fma(-n, 1.90821492927058770002e-10, r);
    double q = 1.0 / 6227020800.0;
    q = fma(q, r, 1.0 / 479001600.0);
    q = fma(q, r, 1.0 / 39916800.0);
    q = fma(q, r, 1.0 / 3628800.0);
    q = fma(q, r, 1.0 / 362880.0);
    q = fma(q, r, 1.0 / 40320.0);
    q = fma(q, r, 1.0 / 5040.0);
    q = fma(q, r, 1.0 / 720.0);
    q = fma(q, r, 1.0 / 120.0);
    q = fma(q, r, 1.0 / 24.0);
    q = fma(q, r, 1.0 / 6.0);
    q = fma(q, r, 0.5);
    const double p = fma(r * r, q, r);
    const int ni = (n == n) ? (int)n : 0;
    const double s = ldexp(1.0, ni);
    const double em1 = fma(s, p, s - 1.0), ep1 = fma(s, p, s + 1.0);
    const double t = ax >= 19.0625 ? 1.0 : em1 / ep1;
    return copysign(t, x);
}
/* (exported for the tests) */
void orc_tanh(const double* x, double* y, int n) {
    for (int i = 0; i < n; ++i) y[i] = smm_tanh(x[i]);
}

/* synthetic dense simulation (BASELINE config 5; no reference counterpart, PARITY UNPINNED; the
 * spec is frozen in include/smmhip.h): x = B*theta, h = tanh(x) (smm_tanh above), y = A*h, simM = y,
 * value = mean(((simM-mom)/w)^2) as in objfunc_norm (ObjExamples.jl:90-101). */
static void objfunc_dense(int np, int nm, const double* theta, const double* Bm /*[D][np]*/, const double* Am /*[nm][D]*/,
                          const double* mom, const double* w, double* simM, double* value, int8_t* status) {
    double h[ORC_DENSE_D];
    for (int d = 0; d < ORC_DENSE_D; ++d) {
        double acc = 0.0;
        for (int p = 0; p < np; ++p) acc = fma(Bm[(size_t)d * np + p], theta[p], acc);
        h[d] = smm_tanh(acc);
    }
    double vsum = 0.0;
    for (int k = 0; k < nm; ++k) {
        double tot = 0.0;
        for (int wv = 0; wv < 8; ++wv) {
            double acc = 0.0;
            for (int d = 32 * wv; d < 32 * wv + 32; ++d) acc = fma(Am[(size_t)k * ORC_DENSE_D + d], h[d], acc);
            tot = (wv == 0) ? acc : tot + acc;
        }
        simM[k] = tot;
        double dd = tot - mom[k];
        if (!isnan(w[k])) dd = dd / w[k];
        double v = dd * dd;
        vsum = (k == 0) ? v : vsum + v;
    }
    *value = vsum / (double)nm;
    *status = 1;
}

/* spec v2 (SMM_OBJ_DENSE2, include/smmhip.h; BASELINE config 5 as worded — a 256 x 256 matvec per evaluation; no reference counterpart,
 * PARITY UNPINNED; the plugin seam: MProb.objfunc, mprob.jl:159,182): x = B*theta, h1 = tanh(x), g = A2*h1, h2 = tanh(g), y = A*h2.
 * Summation order: g_j is ONE fma chain over d = 0..255 (a row tile's accumulator through its 64 matrix instructions); x and y as above. */
static void objfunc_dense2(int np, int nm, const double* theta, const double* Bm /*[D][np]*/, const double* A2 /*[D][D]*/,
                           const double* Am /*[nm][D]*/, const double* mom, const double* w, double* simM, double* value, int8_t* status) {
    double h1[ORC_DENSE_D], h2[ORC_DENSE_D];
    for (int d = 0; d < ORC_DENSE_D; ++d) {
        double acc = 0.0;
        for (int p = 0; p < np; ++p) acc = fma(Bm[(size_t)d * np + p], theta[p], acc);
        h1[d] = smm_tanh(acc);
    }
    for (int j = 0; j < ORC_DENSE_D; ++j) {
        double acc = 0.0;
        for (int d = 0; d < ORC_DENSE_D; ++d) acc = fma(A2[(size_t)j * ORC_DENSE_D + d], h1[d], acc);
        h2[j] = smm_tanh(acc);
    }
    double vsum = 0.0;
    for (int k = 0; k < nm; ++k) {
        double tot = 0.0;
        for (int wv = 0; wv < 8; ++wv) {
            double acc = 0.0;
            for (int d = 32 * wv; d < 32 * wv + 32; ++d) acc = fma(Am[(size_t)k * ORC_DENSE_D + d], h2[d], acc);
            tot = (wv == 0) ? acc : tot + acc;
        }
        simM[k] = tot;
        double dd = tot - mom[k];
        if (!isnan(w[k])) dd = dd / w[k];
        double v = dd * dd;
        vsum = (k == 0) ? v : vsum + v;
    }
    *value = vsum / (double)nm;
    *status = 1;
}

/* default matrices of the dense objective: N(0,1)/sqrt(fan-in) from the counter RNG (stream 5) */
static void gen_dense_n(uint64_t seed, int np, size_t nB, size_t nA, double* out);
void orc_gen_dense(uint64_t seed, int np, int nm, double* out /* [D*np + nm*D] */) {
    gen_dense_n(seed, np, (size_t)ORC_DENSE_D * np, (size_t)nm * ORC_DENSE_D, out);
}
/* spec v2: [B, A2, A]: every entry behind B has fan-in D */
void orc_gen_dense2(uint64_t seed, int np, int nm, double* out /* [D*np + D*D + nm*D] */) {
    gen_dense_n(seed, np, (size_t)ORC_DENSE_D * np, (size_t)ORC_DENSE_D * ORC_DENSE_D + (size_t)nm * ORC_DENSE_D, out);
}
static void gen_dense_n(uint64_t seed, int np, size_t nB, size_t nA, double* out) {
    uint32_t key[2];
    stream_key(seed, 5, key);
    for (size_t i = 0; i < nB + nA; i += 2) {
        uint32_t ctr[4] = {(uint32_t)(i >> 1), (uint32_t)((i >> 1) >> 32), 0, 0}, x[4];
        double z[2];
        philox4x32_10(ctr, key, x);
        box_muller(x, z);
        for (int e = 0; e < 2 && i + e < nB + nA; ++e) {
            const size_t idx = i + e;
            out[idx] = z[e] / sqrt(idx < nB ? (double)np : (double)ORC_DENSE_D);
        }
    }
}

typedef struct {
    orc_problem_t prob;
    orc_opts_t opts;
    int have_u, have_norm, have_pairs;
    double *init, *lb, *ub, *mom, *w, *obj_params;
    double *acc_tuner, *min_improve; /* [Ng] */
    double *u_tab, *norm_tab, *Z;
    int32_t* pair_tab;
    int32_t prop_tries, n_pairs;
    /* chain state [N] */
    int32_t iter;
    double *sigma, *accept_rate;
    double *la_value, *la_prob, *la_params /*[np][N]*/, *la_simM /*[nm][N]*/;
    int8_t* la_status;
    int32_t *n_noex, *n_acc_noex;
    double* best_val; int32_t* best_id;
    /* history [T][...] */
    orc_history_t h;
    char err[256];
    int regen_z, threads;
    double* chol_L;   /* owned copy of opts.chol_L */
} orc_t;

/* evaluateObjective(m,p), mprob.jl:175-188: run the objective; an exception => status=-2
 * with the Eval left at its constructor defaults value=-1.0 (Eval.jl:84), no simMoments. */
static void evaluate_objective(const orc_t* o, const double* theta, double* simM, double* value, int8_t* status) {
    const orc_problem_t* p = &o->prob;
    switch (p->objective_id) {
    case ORC_OBJ_NORM:
        objfunc_norm(p->nm, p->ns, theta, o->Z, o->regen_z, o->opts.seed, o->mom, o->w, simM, value, status);
        break;
    case ORC_OBJ_BANANA:
        objfunc_banana(p->np, p->nm, theta, o->mom, simM, value, status);
        break;
    case ORC_OBJ_DENSE:
        objfunc_dense(p->np, p->nm, theta, o->obj_params, o->obj_params + (size_t)ORC_DENSE_D * p->np, o->mom, o->w, simM,
                      value, status);
        break;
    case ORC_OBJ_DENSE2:
        objfunc_dense2(p->np, p->nm, theta, o->obj_params, o->obj_params + (size_t)ORC_DENSE_D * p->np,
                       o->obj_params + (size_t)ORC_DENSE_D * p->np + (size_t)ORC_DENSE_D * ORC_DENSE_D, o->mom, o->w, simM, value, status);
        break;
    case ORC_OBJ_NORM_FAILBOX:
        if (o->obj_params && theta[0] >= o->obj_params[0] && theta[0] <= o->obj_params[1]) {
            for (int k = 0; k < p->nm; ++k) simM[k] = NAN;
            *value = -1.0;
            *status = -2;
        } else {
            objfunc_norm(p->nm, p->ns, theta, o->Z, o->regen_z, o->opts.seed, o->mom, o->w, simM, value, status);
        }
        break;
    default:
        if (p->objective_id >= ORC_OBJ_USER_BASE && p->objective_id - ORC_OBJ_USER_BASE < ORC_MAX_USER &&
            g_user_partial[p->objective_id - ORC_OBJ_USER_BASE]) {
            /* map-reduce form: every lane's partial sums, then the library's reduction order — inside each group of 64
             * lanes the halving tree (offsets 32..1; lane l takes l + off), the group totals left to right */
            const int u = p->objective_id - ORC_OBJ_USER_BASE, ns = g_user_nsums[u], nl = g_user_lanes[u];
            double* parts = (double*)calloc((size_t)nl * ns, sizeof(double));
            double* tot = (double*)calloc((size_t)ns, sizeof(double));
            for (int l = 0; l < nl; ++l) g_user_partial[u](theta, p->np, o->obj_params, p->n_obj_params, l, nl, parts + (size_t)l * ns);
            for (int i = 0; i < ns; ++i) {
                for (int g = 0; g < nl / 64; ++g) {
                    double a[64];
                    for (int l = 0; l < 64; ++l) a[l] = parts[(size_t)(g * 64 + l) * ns + i];
                    for (int off = 32; off >= 1; off >>= 1)
                        for (int l = 0; l < off; ++l) a[l] = a[l] + a[l + off];
                    tot[i] = (g == 0) ? a[0] : tot[i] + a[0];
                }
            }
            int st = 1;
            double v = 0.0;
            g_user_finish[u](theta, p->np, tot, ns, o->mom, o->w, p->nm, o->obj_params, p->n_obj_params, simM, &v, &st);
            *value = v;
            *status = (int8_t)st;
            free(parts); free(tot);
            break;
        }
        if (p->objective_id >= ORC_OBJ_USER_BASE && p->objective_id - ORC_OBJ_USER_BASE < ORC_MAX_USER &&
            g_user_fn[p->objective_id - ORC_OBJ_USER_BASE]) {
            /* user objective (include/smmhip.h, SMM_USER_OBJECTIVE): the same source text the device compiles,
             * built for the host by the test and registered under the same handle */
            int st = 1;
            double v = 0.0;
            g_user_fn[p->objective_id - ORC_OBJ_USER_BASE](theta, p->np, o->mom, o->w, p->nm, o->obj_params, p->n_obj_params, simM,
                                                           &v, &st);
            *value = v;
            *status = (int8_t)st;
            break;
        }
        for (int k = 0; k < p->nm; ++k) simM[k] = NAN;
        *value = -1.0;
        *status = -2;
    }
}

void orc_set_user_objective_lanes(int objective_id, orc_user_partial_fn pf, orc_user_finish_fn ff, int n_sums, int lanes) {
    if (objective_id >= ORC_OBJ_USER_BASE && objective_id - ORC_OBJ_USER_BASE < ORC_MAX_USER) {
        const int u = objective_id - ORC_OBJ_USER_BASE;
        g_user_partial[u] = pf; g_user_finish[u] = ff; g_user_nsums[u] = n_sums; g_user_lanes[u] = lanes;
    }
}

void orc_set_user_objective(int objective_id, orc_user_fn fn) {
    if (objective_id >= ORC_OBJ_USER_BASE && objective_id - ORC_OBJ_USER_BASE < ORC_MAX_USER)
        g_user_fn[objective_id - ORC_OBJ_USER_BASE] = fn;
}

static double* dupd(const double* s, size_t n) {
    double* d = (double*)malloc((n ? n : 1) * sizeof(double));
    if (s && n) memcpy(d, s, n * sizeof(double));
    return d;
}

void orc_ctx_destroy(void* v) {
    orc_t* o = (orc_t*)v;
    if (!o) return;
    free(o->init); free(o->lb); free(o->ub); free(o->mom); free(o->w); free(o->obj_params);
    free(o->acc_tuner); free(o->min_improve); free(o->chol_L); free(o->u_tab); free(o->norm_tab); free(o->Z); free(o->pair_tab);
    free(o->sigma); free(o->accept_rate); free(o->la_value); free(o->la_prob); free(o->la_params); free(o->la_simM);
    free(o->la_status); free(o->n_noex); free(o->n_acc_noex); free(o->best_val); free(o->best_id);
    free(o->h.value); free(o->h.prob); free(o->h.curr_val); free(o->h.best_val); free(o->h.params);
    free(o->h.sim_moments); free(o->h.best_id); free(o->h.exchanged); free(o->h.accepted); free(o->h.status);
    free(o);
}

/* MAlgoBGP(m,opts) + BGPChain(id,n;...) constructors: AlgoBGP.jl:505-537, :78-109.
 * best_val = curr_val = Inf, best_id = -1, accepted = false, exchanged = 0, accept_rate = 0,
 * iter = 0 (:82-92).  probs_acc = rand(n) (:85) is either injected or generated on demand. */
int orc_ctx_create(const orc_problem_t* prob, const orc_opts_t* opts, const orc_tables_t* tab, void** out) {
    if (!prob || !opts || !out) return ORC_ERR_INVALID_ARG;
    const int np = prob->np, nm = prob->nm, ns = prob->ns, N = opts->N, T = opts->maxiter, Ng = opts->N_global;
    if (np < 1 || nm < 1 || ns < 1 || N < 1 || T < 1 || Ng < N || opts->chain_offset < 0 ||
        opts->chain_offset + N > Ng)
        return ORC_ERR_INVALID_ARG;
    if ((prob->objective_id == ORC_OBJ_NORM || prob->objective_id == ORC_OBJ_NORM_FAILBOX) && np != nm)
        return ORC_ERR_INVALID_ARG; /* objfunc_norm pairs param k with moment k, ObjExamples.jl:66-78 */
    if (opts->batch_size < 1 || opts->batch_size > np || np % opts->batch_size != 0) return ORC_ERR_BAD_BATCH;
    orc_t* o = (orc_t*)calloc(1, sizeof(orc_t));
    o->prob = *prob; o->opts = *opts;
    o->init = dupd(prob->init, np); o->lb = dupd(prob->lb, np); o->ub = dupd(prob->ub, np);
    o->mom = dupd(prob->mom, nm); o->w = dupd(prob->w, nm);
    o->obj_params = prob->n_obj_params > 0 ? dupd(prob->obj_params, prob->n_obj_params) : NULL;
    if (prob->objective_id == ORC_OBJ_DENSE && !o->obj_params) {
        o->obj_params = (double*)malloc(((size_t)ORC_DENSE_D * np + (size_t)nm * ORC_DENSE_D) * sizeof(double));
        orc_gen_dense(opts->seed, np, nm, o->obj_params);
    }
    if (prob->objective_id == ORC_OBJ_DENSE2 && !o->obj_params) {
        o->obj_params = (double*)malloc(((size_t)ORC_DENSE_D * np + (size_t)ORC_DENSE_D * ORC_DENSE_D + (size_t)nm * ORC_DENSE_D) * sizeof(double));
        orc_gen_dense2(opts->seed, np, nm, o->obj_params);
    }
    o->acc_tuner = dupd(opts->acc_tuner, Ng); o->min_improve = dupd(opts->min_improve, Ng);
    if (opts->chol_L) {
        if (opts->batch_size != np) { free(o); return ORC_ERR_BAD_BATCH; }   /* one proposal batch, include/smmhip.h */
        o->chol_L = dupd(opts->chol_L, (opts->chol_per_chain ? (size_t)Ng : 1) * np * np);
    }
    o->sigma = dupd(opts->sigma + opts->chain_offset, N);
    size_t TN = (size_t)T * N;
    if (tab && tab->probs_acc) { o->u_tab = dupd(tab->probs_acc, TN); o->have_u = 1; }
    if (tab && tab->prop_normals && tab->prop_tries > 0) {
        o->prop_tries = tab->prop_tries;
        o->norm_tab = dupd(tab->prop_normals, TN * (size_t)tab->prop_tries * np);
        o->have_norm = 1;
    }
    if (tab && tab->pairs && tab->n_pairs > 0) {
        o->n_pairs = tab->n_pairs;
        size_t n = (size_t)T * tab->n_pairs * 2;
        o->pair_tab = (int32_t*)malloc(n * sizeof(int32_t));
        memcpy(o->pair_tab, tab->pairs, n * sizeof(int32_t));
        o->have_pairs = 1;
    }
    o->Z = (double*)malloc((size_t)nm * ns * sizeof(double));
    if (tab && tab->Z) memcpy(o->Z, tab->Z, (size_t)nm * ns * sizeof(double));
    else orc_gen_Z(opts->seed, nm, ns, o->Z);
    o->accept_rate = (double*)calloc(N, sizeof(double));
    o->la_value = (double*)calloc(N, sizeof(double)); o->la_prob = (double*)calloc(N, sizeof(double));
    o->la_params = (double*)calloc((size_t)np * N, sizeof(double));
    o->la_simM = (double*)calloc((size_t)nm * N, sizeof(double));
    o->la_status = (int8_t*)calloc(N, 1);
    o->n_noex = (int32_t*)calloc(N, sizeof(int32_t)); o->n_acc_noex = (int32_t*)calloc(N, sizeof(int32_t));
    o->best_val = (double*)malloc(N * sizeof(double)); o->best_id = (int32_t*)malloc(N * sizeof(int32_t));
    for (int c = 0; c < N; ++c) { o->best_val[c] = INFINITY; o->best_id[c] = -1; o->la_value[c] = INFINITY; }
    o->h.value = (double*)malloc(TN * 8); o->h.prob = (double*)malloc(TN * 8);
    o->h.curr_val = (double*)malloc(TN * 8); o->h.best_val = (double*)malloc(TN * 8);
    o->h.params = (double*)malloc(TN * np * 8); o->h.sim_moments = (double*)malloc(TN * nm * 8);
    o->h.best_id = (int32_t*)malloc(TN * 4); o->h.exchanged = (int32_t*)calloc(TN, 4);
    o->h.accepted = (uint8_t*)calloc(TN, 1); o->h.status = (int8_t*)calloc(TN, 1);
    for (size_t i = 0; i < TN; ++i) { o->h.best_val[i] = INFINITY; o->h.curr_val[i] = INFINITY; o->h.best_id[i] = -1;
        o->h.value[i] = NAN; o->h.prob[i] = NAN; }
    for (size_t i = 0; i < TN * np; ++i) o->h.params[i] = NAN;
    for (size_t i = 0; i < TN * nm; ++i) o->h.sim_moments[i] = NAN;
    o->threads = 1;
    *out = o;
    return ORC_OK;
}

const char* orc_last_error(void* v) { return v ? ((orc_t*)v)->err : "null ctx"; }
void orc_set_mode(void* v, int threads, int regen_z) {
    orc_t* o = (orc_t*)v;
    o->threads = threads < 1 ? 1 : threads;
    o->regen_z = regen_z;
}

/* set_eval!(c,ev), AlgoBGP.jl:220-245, for chain c at iteration t (1-based), given the
 * best_val/best_id of iteration t-1.  Writes history row t and the running best. */
static void store_record(orc_t* o, int c, int t, const double* params, const double* simM, double value, double prob,
                         int accepted, int8_t status, double best_prev, int32_t best_id_prev, double curr_prev) {
    const int N = o->opts.N, np = o->prob.np, nm = o->prob.nm;
    const size_t r = (size_t)(t - 1) * N + c;
    o->h.value[r] = value; o->h.prob[r] = prob; o->h.accepted[r] = (uint8_t)accepted; o->h.status[r] = status;
    for (int k = 0; k < np; ++k) o->h.params[((size_t)(t - 1) * np + k) * N + c] = params[k];
    for (int k = 0; k < nm; ++k) o->h.sim_moments[((size_t)(t - 1) * nm + k) * N + c] = simM[k];
    if (t == 1) {                                  /* :225-228 */
        o->h.best_val[r] = value; o->h.curr_val[r] = value; o->h.best_id[r] = 1;
    } else {
        o->h.curr_val[r] = accepted ? value : curr_prev;      /* :231-235 */
        if (value < best_prev) { o->h.best_val[r] = value; o->h.best_id[r] = t; }   /* :236-238 */
        else { o->h.best_val[r] = best_prev; o->h.best_id[r] = best_id_prev; }      /* :239-243 */
    }
}

/* next_eval(c), AlgoBGP.jl:272-294, for local chain c at iteration t:
 * proposal (:424-471, mysample :400-410, mapto_01/ab mprob.jl:246-272) ->
 * evaluateObjective (mprob.jl:175-188) -> doAcceptReject! (:324-392) -> set_eval! (:220-245).
 * Returns 0 or an error code. */
static int next_eval(orc_t* o, int c, int t, double* theta, double* simM, double* x01) {
    const int N = o->opts.N, np = o->prob.np, nm = o->prob.nm, T = o->opts.maxiter;
    const uint32_t gc = (uint32_t)(o->opts.chain_offset + c);
    const uint64_t seed = o->opts.seed;
    (void)T;
    /* ---- proposal ---- */
    if (t == 1) {
        for (int k = 0; k < np; ++k) theta[k] = o->init[k];        /* :426-427 */
    } else {
        const double sig = o->sigma[c];
        const int bs = o->opts.batch_size;
        const int max_tries = o->have_norm ? (o->prop_tries < o->opts.smpl_iters ? o->prop_tries : o->opts.smpl_iters)
                                           : o->opts.smpl_iters;
        for (int b0 = 0; b0 < np; b0 += bs) {          /* one batch when bs==np (:441-442), else per batch (:444-453) */
            int ok = 0;
            for (int r = 0; r < max_tries && !ok; ++r) { /* mysample :403-408 */
                ok = 1;
                for (int k = b0; k < b0 + bs; ++k) {
                    double mu01 = (o->la_params[(size_t)k * N + c] - o->lb[k]) / (o->ub[k] - o->lb[k]); /* mprob.jl:248 */
                    double z;
                    if (o->chol_L) {
                        /* general Gaussian kernel (north star "Cholesky apply"; vector sigma hinted at AlgoBGP.jl:218):
                         * direction (L z)_k = sum_{j<=k} L[k][j]*z[j], products rounded, added left to right */
                        const double* Lk = o->chol_L + (o->opts.chol_per_chain ? (size_t)gc * np * np : 0) + (size_t)k * np;
                        z = 0.0;
                        for (int j = 0; j <= k; ++j) {
                            double zj = o->have_norm
                                            ? o->norm_tab[((((size_t)(t - 1) * o->prop_tries + r) * np + j) * N) + c]
                                            : rng_prop_normal(seed, gc, (uint32_t)t, (uint32_t)r, (uint32_t)j);
                            double pr = Lk[j] * zj;
                            z = (j == 0) ? pr : z + pr;
                        }
                    } else {
                        z = o->have_norm ? o->norm_tab[((((size_t)(t - 1) * o->prop_tries + r) * np + k) * N) + c]
                                         : rng_prop_normal(seed, gc, (uint32_t)t, (uint32_t)r, (uint32_t)k);
                    }
                    double step = sig * z;              /* MvNormal(mu01, sigma::Float64): x = mu + sigma*z */
                    double x = mu01 + step;
                    x01[k] = x;
                    if (!(x >= 0.0 && x <= 1.0)) ok = 0; /* inclusive bounds :405 */
                }
            }
            if (!ok) {                                  /* :409 (batch mode: error raised, not swallowed) */
                snprintf(o->err, sizeof o->err, "no draw in support after %d trials: chain %u iter %d", max_tries,
                         gc + 1, t);
                return ORC_ERR_NO_DRAW_IN_SUPPORT;
            }
        }
        for (int k = 0; k < np; ++k) {
            double span = o->ub[k] - o->lb[k];
            double sc = x01[k] * span;
            theta[k] = sc + o->lb[k];                   /* mapto_ab, mprob.jl:271 */
        }
    }
    /* ---- objective ---- */
    double value; int8_t status;
    evaluate_objective(o, theta, simM, &value, &status);
    /* ---- doAcceptReject! ---- */
    double prob; int acc;
    if (t == 1) {                                       /* :326-332 */
        prob = 1.0; acc = 1; status = 1;
    } else {
        const double old = o->la_value[c];
        if (status < 0) {                               /* :336-338 */
            prob = 0.0; acc = 0;
        } else {
            if (!(value >= 0.0)) {                      /* :341 (NaN >= 0 is false => error as well) */
                snprintf(o->err, sizeof o->err, "objective returned a negative or NaN value %g: chain %u iter %d",
                         value, gc + 1, t);
                return ORC_ERR_NEGATIVE_OBJECTIVE;
            }
            double e = smm_exp(o->acc_tuner[gc] * (old - value));
            prob = (e != e) ? e : (e < 1.0 ? e : 1.0);  /* minimum([1.0, e]) propagates NaN, :344 */
            if (!isfinite(prob)) { prob = 0.0; acc = 0; status = -1; }            /* :350-353 */
            else if (!isfinite(old)) { prob = 1.0; acc = 1; }                     /* :355-359 */
            else {
                status = 1;
                double u = o->have_u ? o->u_tab[(size_t)(t - 1) * N + c] : rng_u(seed, gc, (uint32_t)t);
                acc = (prob > u) ? 1 : 0;                                          /* strict >, :362-367 */
            }
        }
    }
    /* set_acceptRate!, :253-257: mean(accepted[1:iter][exchanged[1:iter].==0]); at this point
     * exchanged[iter]==0, earlier iterations have their final exchanged status. */
    o->accept_rate[c] = (double)(o->n_acc_noex[c] + acc) / (double)(o->n_noex[c] + 1);
    if (t > 1 && (t % o->opts.sigma_update_steps) == 0) {                          /* :381-390 */
        if (o->accept_rate[c] > 0.234) o->sigma[c] = o->sigma[c] * (1.0 + o->opts.sigma_adjust_by);
        else o->sigma[c] = o->sigma[c] * (1.0 - o->opts.sigma_adjust_by);
    }
    /* ---- set_eval! ---- */
    store_record(o, c, t, theta, simM, value, prob, acc, status, o->best_val[c], o->best_id[c], o->la_value[c]);
    const size_t r = (size_t)(t - 1) * N + c;
    o->best_val[c] = o->h.best_val[r]; o->best_id[c] = o->h.best_id[r];
    if (acc) { /* the record becomes the chain's last accepted one (lastAccepted :209-215) */
        o->la_value[c] = value; o->la_prob[c] = prob; o->la_status[c] = status;
        for (int k = 0; k < np; ++k) o->la_params[(size_t)k * N + c] = theta[k];
        for (int k = 0; k < nm; ++k) o->la_simM[(size_t)k * N + c] = simM[k];
    }
    return ORC_OK;
}

/* all local chains: map(next_eval, chains), AlgoBGP.jl:614 (or the pmap form :596-605) */
int orc_bgp_local_step(void* v) {
    orc_t* o = (orc_t*)v;
    const int N = o->opts.N, np = o->prob.np, nm = o->prob.nm;
    if (o->iter >= o->opts.maxiter) { snprintf(o->err, sizeof o->err, "maxiter reached"); return ORC_ERR_MAXITER; }
    const int t = o->iter + 1;
    int rc = ORC_OK;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(o->threads) if (o->threads > 1)
#endif
    for (int c = 0; c < N; ++c) {
        double* buf = (double*)malloc((size_t)(2 * np + nm) * sizeof(double));
        int e = next_eval(o, c, t, buf, buf + np, buf + np + nm);
        if (e != ORC_OK) {
#ifdef _OPENMP
#pragma omp critical
#endif
            { if (rc == ORC_OK) rc = e; }
        }
        free(buf);
    }
    if (rc != ORC_OK) return rc;
    o->iter = t;
    return ORC_OK;
}

/* doubles per exported record: value, prob, status, params[np], simM[nm], padded to an even count */
int orc_bgp_record_doubles(void* v) { orc_t* o = (orc_t*)v; return (3 + o->prob.np + o->prob.nm + 1) & ~1; }

/* last-accepted records of the local chains, [N][RW] */
void orc_bgp_export_records(void* v, double* rec) {
    orc_t* o = (orc_t*)v;
    const int N = o->opts.N, np = o->prob.np, nm = o->prob.nm, RW = orc_bgp_record_doubles(v);
    for (int c = 0; c < N; ++c) {
        double* r = rec + (size_t)c * RW;
        for (int f = 0; f < RW; ++f) r[f] = 0.0;
        r[0] = o->la_value[c]; r[1] = o->la_prob[c]; r[2] = (double)o->la_status[c];
        for (int k = 0; k < np; ++k) r[3 + k] = o->la_params[(size_t)k * N + c];
        for (int k = 0; k < nm; ++k) r[3 + np + k] = o->la_simM[(size_t)k * N + c];
    }
}

/* after the exchange phase of iteration t: iterations with exchanged==0 count towards the
 * acceptance rate (set_acceptRate!, :253-257) */
static void close_iteration(orc_t* o) {
    const int N = o->opts.N, t = o->iter;
    for (int c = 0; c < N; ++c) {
        const size_t r = (size_t)(t - 1) * N + c;
        if (o->h.exchanged[r] == 0) { o->n_noex[c] += 1; o->n_acc_noex[c] += o->h.accepted[r]; }
    }
}

/* exchangeMoves!(algo), AlgoBGP.jl:647-716 + swap_ev_ij! :734-749, over ALL N_global chains.
 * gathered = records of every chain in global id order, [N_global][RW] (shards concatenated);
 * must be called after orc_bgp_local_step for the same iteration; applies swaps to local chains. */
int orc_bgp_exchange(void* v, const double* gathered) {
    orc_t* o = (orc_t*)v;
    const int N = o->opts.N, Ng = o->opts.N_global, np = o->prob.np, nm = o->prob.nm, t = o->iter;
    const int RW = orc_bgp_record_doubles(v);
    if (!(t >= o->opts.exchange_from_iter && Ng > 1)) { close_iteration(o); return ORC_OK; }   /* :637 */
    const int K = o->have_pairs ? o->n_pairs : n_exchange_pairs(Ng);
    int32_t* pairs = (int32_t*)malloc((size_t)(K > 0 ? K : 1) * 2 * sizeof(int32_t));
    if (o->have_pairs) memcpy(pairs, o->pair_tab + (size_t)(t - 1) * K * 2, (size_t)K * 2 * sizeof(int32_t));
    else orc_gen_pairs(o->opts.seed, t, Ng, pairs);
    double* val = (double*)malloc((size_t)Ng * sizeof(double));
    int32_t* src = (int32_t*)malloc((size_t)Ng * sizeof(int32_t));
    int32_t* partner = (int32_t*)calloc((size_t)Ng, sizeof(int32_t));
    for (int g = 0; g < Ng; ++g) { val[g] = gathered[(size_t)g * RW]; src[g] = g; }
    for (int q = 0; q < K; ++q) {                      /* sequential, order dependent: :662-691 */
        int i = pairs[2 * q], j = pairs[2 * q + 1];
        if (orc_dist_fun(o->opts.dist_fun, val[i], val[j]) > o->min_improve[i]) {     /* :688 */
            double tv = val[i]; val[i] = val[j]; val[j] = tv;       /* swap_ev_ij! :739-744 */
            int32_t ts = src[i]; src[i] = src[j]; src[j] = ts;
            partner[i] = j + 1; partner[j] = i + 1;                  /* set_exchanged! :747-748 */
        }
    }
    for (int c = 0; c < N; ++c) {
        int g = o->opts.chain_offset + c;
        if (partner[g] == 0) continue;
        const double* rec = gathered + (size_t)src[g] * RW;
        double value = rec[0], prob = rec[1];
        int8_t status = (int8_t)rec[2];
        const double* params = rec + 3;
        const double* simM = rec + 3 + np;
        /* set_eval!(ci, ej): overwrites the chain's record of iteration t; best/curr are
         * recomputed against iteration t-1 (:231-243); ej.accepted is true. */
        const size_t rp = (size_t)(t - 2) * N + c;
        store_record(o, c, t, params, simM, value, prob, 1, status, o->h.best_val[rp], o->h.best_id[rp],
                     o->h.curr_val[rp]);
        const size_t r = (size_t)(t - 1) * N + c;
        o->h.exchanged[r] = partner[g];
        o->best_val[c] = o->h.best_val[r]; o->best_id[c] = o->h.best_id[r];
        o->la_value[c] = value; o->la_prob[c] = prob; o->la_status[c] = status;
        for (int k = 0; k < np; ++k) o->la_params[(size_t)k * N + c] = params[k];
        for (int k = 0; k < nm; ++k) o->la_simM[(size_t)k * N + c] = simM[k];
    }
    free(pairs); free(val); free(src); free(partner);
    close_iteration(o);
    return ORC_OK;
}

/* the ordered walk of exchangeMoves! (:662-691) alone, from every chain's value: src_out[g] = whose record chain g continues
 * from, partner_out[g] = 1 + its last exchange partner (0: none) — what the values form of the sharded exchange
 * (include/smmhip.h) decides its traffic by */
int orc_bgp_resolve_values(void* v, const double* vals_all, int32_t* src_out, int32_t* partner_out) {
    orc_t* o = (orc_t*)v;
    const int Ng = o->opts.N_global, t = o->iter;
    for (int g = 0; g < Ng; ++g) { src_out[g] = g; partner_out[g] = 0; }
    if (!(t >= o->opts.exchange_from_iter && Ng > 1)) return ORC_OK;   /* :637 */
    const int K = o->have_pairs ? o->n_pairs : n_exchange_pairs(Ng);
    int32_t* pairs = (int32_t*)malloc((size_t)(K > 0 ? K : 1) * 2 * sizeof(int32_t));
    if (o->have_pairs) memcpy(pairs, o->pair_tab + (size_t)(t - 1) * K * 2, (size_t)K * 2 * sizeof(int32_t));
    else orc_gen_pairs(o->opts.seed, t, Ng, pairs);
    double* val = (double*)malloc((size_t)Ng * sizeof(double));
    memcpy(val, vals_all, (size_t)Ng * sizeof(double));
    for (int q = 0; q < K; ++q) {
        int i = pairs[2 * q], j = pairs[2 * q + 1];
        if (orc_dist_fun(o->opts.dist_fun, val[i], val[j]) > o->min_improve[i]) {
            double tv = val[i]; val[i] = val[j]; val[j] = tv;
            int32_t ts = src_out[i]; src_out[i] = src_out[j]; src_out[j] = ts;
            partner_out[i] = j + 1; partner_out[j] = i + 1;
        }
    }
    free(pairs); free(val);
    return ORC_OK;
}

/* computeNextIteration!(algo), AlgoBGP.jl:589-640, n_iters times (run!, AlgoAbstract.jl:38-45);
 * single shard only. */
int orc_bgp_step(void* v, int n_iters) {
    orc_t* o = (orc_t*)v;
    if (o->opts.N != o->opts.N_global) { snprintf(o->err, sizeof o->err, "orc_bgp_step needs a single shard"); return ORC_ERR_INVALID_ARG; }
    const int R = orc_bgp_record_doubles(v);
    double* rec = (double*)malloc((size_t)R * o->opts.N * sizeof(double));
    int rc = ORC_OK;
    for (int it = 0; it < n_iters && rc == ORC_OK; ++it) {
        rc = orc_bgp_local_step(v);
        if (rc != ORC_OK) break;
        orc_bgp_export_records(v, rec);
        rc = orc_bgp_exchange(v, rec);
    }
    free(rec);
    return rc;
}

/* objfunc_norm with options[:noseed] = true (ObjExamples.jl:71-75): every evaluation draws its own shocks — here the
 * counter generator keyed by base_seed + i — as getSigma's repetitions do (econometrics.jl:125-145). */
int orc_eval_batch_noseed(void* v, const double* params, int M, uint64_t base_seed, double* value, double* simM, int8_t* status) {
    orc_t* o = (orc_t*)v;
    const int np = o->prob.np, nm = o->prob.nm;
    if (o->prob.objective_id != ORC_OBJ_NORM && o->prob.objective_id != ORC_OBJ_NORM_FAILBOX) return ORC_ERR_INVALID_ARG;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(o->threads) if (o->threads > 1)
#endif
    for (int i = 0; i < M; ++i) {
        double* th = (double*)malloc((size_t)(np + nm) * sizeof(double));
        double* sm = th + np;
        for (int k = 0; k < np; ++k) th[k] = params[(size_t)k * M + i];
        objfunc_norm(nm, o->prob.ns, th, NULL, 1, base_seed + (uint64_t)i, o->mom, o->w, sm, &value[i], &status[i]);
        for (int k = 0; k < nm; ++k) simM[(size_t)k * M + i] = sm[k];
        free(th);
    }
    return ORC_OK;
}

/* batched evaluateObjective: params [np][M] -> value[M], simM[nm][M], status[M] */
int orc_eval_batch(void* v, const double* params, int M, double* value, double* simM, int8_t* status) {
    orc_t* o = (orc_t*)v;
    const int np = o->prob.np, nm = o->prob.nm;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(o->threads) if (o->threads > 1)
#endif
    for (int i = 0; i < M; ++i) {
        double* th = (double*)malloc((size_t)(np + nm) * sizeof(double));
        double* sm = th + np;
        for (int k = 0; k < np; ++k) th[k] = params[(size_t)k * M + i];
        evaluate_objective(o, th, sm, &value[i], &status[i]);
        for (int k = 0; k < nm; ++k) simM[(size_t)k * M + i] = sm[k];
        free(th);
    }
    return ORC_OK;
}

#define CPY(dst, src, n, sz) do { if (dst) memcpy(dst, src, (size_t)(n) * (sz)); } while (0)

int orc_get_history(void* v, int t0, int t1, orc_history_t* out) {
    orc_t* o = (orc_t*)v;
    const size_t N = o->opts.N, np = o->prob.np, nm = o->prob.nm;
    if (t0 < 0 || t1 < t0 || t1 > o->opts.maxiter) return ORC_ERR_INVALID_ARG;
    const size_t nt = (size_t)(t1 - t0), off = (size_t)t0 * N;
    CPY(out->value, o->h.value + off, nt * N, 8); CPY(out->prob, o->h.prob + off, nt * N, 8);
    CPY(out->curr_val, o->h.curr_val + off, nt * N, 8); CPY(out->best_val, o->h.best_val + off, nt * N, 8);
    CPY(out->params, o->h.params + off * np, nt * N * np, 8);
    CPY(out->sim_moments, o->h.sim_moments + off * nm, nt * N * nm, 8);
    CPY(out->best_id, o->h.best_id + off, nt * N, 4); CPY(out->exchanged, o->h.exchanged + off, nt * N, 4);
    CPY(out->accepted, o->h.accepted + off, nt * N, 1); CPY(out->status, o->h.status + off, nt * N, 1);
    return ORC_OK;
}

int orc_get_state(void* v, orc_state_t* s) {
    orc_t* o = (orc_t*)v;
    const size_t N = o->opts.N, np = o->prob.np, nm = o->prob.nm;
    s->iter = o->iter;
    CPY(s->sigma, o->sigma, N, 8); CPY(s->accept_rate, o->accept_rate, N, 8);
    CPY(s->la_value, o->la_value, N, 8); CPY(s->la_prob, o->la_prob, N, 8);
    CPY(s->la_params, o->la_params, N * np, 8); CPY(s->la_sim_moments, o->la_simM, N * nm, 8);
    CPY(s->la_status, o->la_status, N, 1); CPY(s->n_noex, o->n_noex, N, 4); CPY(s->n_acc_noex, o->n_acc_noex, N, 4);
    CPY(s->best_val, o->best_val, N, 8); CPY(s->best_id, o->best_id, N, 4);
    return ORC_OK;
}

/* the twin of smm_set_state (include/smmhip.h): restart! (AlgoBGP.jl:804-884) — the chains' state and the history of the
 * iterations before s->iter come from the caller */
int orc_set_state(void* v, const orc_state_t* s, const orc_history_t* hin) {
    orc_t* o = (orc_t*)v;
    const size_t N = o->opts.N, np = o->prob.np, nm = o->prob.nm;
    if (!s || s->iter < 0 || s->iter > o->opts.maxiter || (s->iter > 0 && !hin)) return ORC_ERR_INVALID_ARG;
    const size_t nt = (size_t)s->iter;
    memcpy(o->sigma, s->sigma, N * 8); memcpy(o->accept_rate, s->accept_rate, N * 8);
    memcpy(o->la_value, s->la_value, N * 8); memcpy(o->la_prob, s->la_prob, N * 8);
    memcpy(o->la_params, s->la_params, N * np * 8); memcpy(o->la_simM, s->la_sim_moments, N * nm * 8);
    memcpy(o->la_status, s->la_status, N); memcpy(o->n_noex, s->n_noex, N * 4); memcpy(o->n_acc_noex, s->n_acc_noex, N * 4);
    memcpy(o->best_val, s->best_val, N * 8); memcpy(o->best_id, s->best_id, N * 4);
    if (nt) {
        memcpy(o->h.value, hin->value, nt * N * 8); memcpy(o->h.prob, hin->prob, nt * N * 8);
        memcpy(o->h.curr_val, hin->curr_val, nt * N * 8); memcpy(o->h.best_val, hin->best_val, nt * N * 8);
        memcpy(o->h.params, hin->params, nt * N * np * 8); memcpy(o->h.sim_moments, hin->sim_moments, nt * N * nm * 8);
        memcpy(o->h.best_id, hin->best_id, nt * N * 4); memcpy(o->h.exchanged, hin->exchanged, nt * N * 4);
        memcpy(o->h.accepted, hin->accepted, nt * N); memcpy(o->h.status, hin->status, nt * N);
    }
    o->iter = s->iter;
    return ORC_OK;
}

int orc_get_Z(void* v, double* Z) {
    orc_t* o = (orc_t*)v;
    memcpy(Z, o->Z, (size_t)o->prob.nm * o->prob.ns * sizeof(double));
    return ORC_OK;
}

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
