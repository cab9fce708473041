/*
 * smmhip.h — C ABI of libsmmhip.so: the MI355X (gfx950) backend for the BGP
 * parallel-tempering hot path of floswald/SMM.jl.
 *
 * The reference has no FFI; its two seams are Julia dynamic dispatch:
 *   (1) computeNextIteration!(algo::MAlgoBGP)      src/mopt/AlgoBGP.jl:589-640
 *       (called once per iteration from run!       src/mopt/AlgoAbstract.jl:38-45)
 *   (2) the objective contract f(ev::Eval)::Eval   src/mopt/mprob.jl:175-205
 * A Julia maintainer binds the functions below with `ccall`;
 * the Python host layer in smm.jl_amd/ binds them with ctypes.
 *
 * Conventions
 *   - every entry point returns 0 on success, <0 = smm_status_t error code;
 *     smm_last_error(ctx) returns a human readable message (ctx may be NULL for
 *     errors raised by smm_ctx_create).
 *   - all pointers are HOST pointers unless the name ends in `_dev`.
 *   - host buffers are borrowed for the duration of the call only.
 *   - chain ids and iteration numbers in *downloaded* data are 1-based exactly
 *     as in the reference (BGPChain.id, BGPChain.exchanged, BGPChain.best_id;
 *     AlgoBGP.jl:42-110): exchanged==0 means "no exchange", best_id==-1 "unset".
 *   - all floating point data is IEEE double (the reference is Float64 throughout).
 *   - a ctx is single-threaded (one caller thread), like the reference's master task.
 */
#ifndef SMMHIP_H
#define SMMHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMMHIP_ABI_VERSION 3

/* Numerical contract shared with the oracle (oracle/smm_oracle.c):
 * the ns simulated draws of one moment are summed as SMM_REDUCE_LANES lane-strided
 * sequential partial sums (lane l takes draws l, l+512, l+1024, ...), each group of 64
 * partials is combined by a halving tree (offsets 32,16,8,4,2,1) and the 8 group
 * totals are added left to right.  Replaces mean(X,dims=2), ObjExamples.jl:79.
 * The elementary functions of the path are part of the contract too: the logarithm and the sine / cosine of the generator's
 * Box-Muller transform and the exponential of the acceptance probability (AlgoBGP.jl:344) are fixed sequences of correctly
 * rounded operations (smm.jl_amd/csrc/smm_rng.hpp: smm_log, smm_sincos2pi — after fdlibm —, smm_exp; each within 1 ulp), the
 * dense objective's tanh likewise (below).  Consequence: a run is reproduced BIT FOR BIT by any implementation of the contract —
 * every floating-point field of the history, not only the bookkeeping. */
#define SMM_REDUCE_LANES 512

typedef enum {
    SMM_OK = 0,
    SMM_ERR_INVALID_ARG = -1,
    SMM_ERR_NO_DEVICE = -2,          /* no HIP device / HIP runtime failure          */
    SMM_ERR_NEGATIVE_OBJECTIVE = -3, /* objective value <0 or NaN: AlgoBGP.jl:341     */
    SMM_ERR_NO_DRAW_IN_SUPPORT = -4, /* mysample exhausted smpl_iters: AlgoBGP.jl:409 */
    SMM_ERR_BAD_BATCH = -5,          /* batch_size not a divisor of np: AlgoBGP.jl:95-103 (quirk not replicated) */
    SMM_ERR_MAXITER = -6,            /* step beyond opts.maxiter (history capacity)   */
    SMM_ERR_HIP = -7,
    SMM_ERR_STATE = -8,
    SMM_ERR_EXCHANGE_CAPACITY = -9   /* values form of the sharded exchange: a (source, destination) block overflowed */
} smm_status_t;

/* objective_id: device objectives replacing MProb.objfunc (mprob.jl:159,182) */
typedef enum {
    SMM_OBJ_NORM = 0,   /* objfunc_norm, ObjExamples.jl:59-116 (requires np==nm)          */
    SMM_OBJ_BANANA = 1, /* banana, ObjExamples.jl:251-265, generalised to np dims        */
    SMM_OBJ_NORM_FAILBOX = 2, /* objfunc_norm that "throws" (status=-2, mprob.jl:183-186)
                                when obj_params[0] <= theta_0 <= obj_params[1]; the role of
                                Testobj_fails, ObjExamples.jl:27-32 */
    SMM_OBJ_DENSE = 3,  /* synthetic dense simulation (BASELINE config 5, no reference counterpart):
                           x = B*theta (B: SMM_DENSE_D x np), h = tanh(x), y = A*h (A: nm x SMM_DENSE_D),
                           simM = y, value = mean(((simM-mom)/w)^2).  obj_params = [B row-major, A row-major]
                           (SMM_DENSE_D*np + nm*SMM_DENSE_D doubles) or empty = generated from the seed.
                           Summation order (numerical contract): x_d = fma chain over p; y_k = 8 fma chains
                           over d in [32w, 32w+32), added left to right.  FP64 MFMA on the device.
                           tanh (numerical contract, at most 3 ulp from the true value): with z = 2|x|, n = rint(z log2 e),
                           r = z - n ln2 (two fma), p = expm1(r) by its Taylor series to r^13 (Horner, fma):
                           tanh|x| = fma(2^n, p, 2^n - 1) / fma(2^n, p, 2^n + 1), 1 from |x| = 19.0625 on. */
    /* (4 is SMM_OBJ_USER, the internal kind of every user objective) */
    SMM_OBJ_DENSE2 = 5  /* BASELINE config 5 AS WORDED — "256x256 matvec per eval" (spec v2 of the synthetic dense simulation; SMM_OBJ_DENSE
                           keeps its id and its goldens):
                               x = B*theta (B: 256 x np), h1 = tanh(x), g = A2*h1 (A2: 256 x 256), h2 = tanh(g), y = A*h2 (A: nm x 256),
                           simM = y, value = mean(((simM-mom)/w)^2): 2*256*np + 2*256*256 + 2*nm*256 flop per evaluation (1.8e5 at
                           np = nm = 50).  obj_params = [B row-major, A2 row-major, A row-major] (256*np + 65536 + nm*256 doubles) or
                           empty = generated from the seed (N(0,1)/sqrt(fan-in), counter stream 5, in that order).
                           Summation order (numerical contract): x_d = fma chain over p; g_j = ONE fma chain over d = 0..255 (the
                           accumulator of a row tile's 64 v_mfma_f64_16x16x4); y_k = 8 fma chains over d in [32w, 32w+32), added
                           left to right; the tanh above.  The plugin seam it exercises: MProb.objfunc, mprob.jl:159,182. */
} smm_objective_t;
#define SMM_DENSE_D 256

/* User objectives — the reference's "bring your own objfunc" (MProb.objfunc, mprob.jl:159,182) on the device.
 * objective_id >= SMM_OBJ_USER_BASE is a handle returned by smm_register_user_objective.  The source is HIP/C++
 * text defining ONE function with this exact signature (SMM_USER_OBJECTIVE expands to the required linkage):
 *
 *   SMM_USER_OBJECTIVE(const double* theta, int np, const double* mom, const double* w, int nm,
 *                      const double* udata, int n_udata, double* sim_moments, double* value, int* status)
 *
 * It must fill sim_moments[0..nm), *value (>= 0, AlgoBGP.jl:341) and *status (1 = ok; < 0 = failed, the record is
 * rejected like an objective that threw, mprob.jl:183-186).  It must be a deterministic function of its inputs
 * (a simulation seeds its own generator, as objfunc_norm does with Random.seed!(1234)); udata = obj_params.
 * It is compiled at registration (hiprtc, -ffp-contract=off) and evaluated for all chains of an iteration by one
 * kernel launch, one thread per chain, between the proposal and the accept step.
 *
 * Map-reduce form for simulations that are sums over many independent units (agents, draws, paths) —
 * smm_register_user_objective_lanes(source, n_sums, lanes, &id): `lanes` threads (a multiple of 64, <= 1024)
 * evaluate one chain.  The source defines TWO functions:
 *
 *   SMM_USER_PARTIAL(const double* theta, int np, const double* udata, int n_udata, int lane, int n_lanes,
 *                    double* partial)          lane's partial sums, partial[0..n_sums) (zero on entry);
 *                                              by convention lane l works on units l, l + n_lanes, ...
 *   SMM_USER_FINISH (const double* theta, int np, const double* totals, int n_sums, const double* mom,
 *                    const double* w, int nm, const double* udata, int n_udata, double* sim_moments,
 *                    double* value, int* status)   from the totals to moments, objective value, status.
 *
 * The library reduces the partials in a fixed order (numerical contract): inside each group of 64 lanes the
 * halving tree (offsets 32,16,..,1), then the group totals left to right. */
#define SMM_OBJ_USER_BASE 1000
#define SMM_OBJ_USER 4   /* internal kind of every user objective */

/* MProb (mprob.jl:29-53) flattened: parameters to sample with bounds and start
 * values (addSampledParam!, mprob.jl:81-98), data moments and weights
 * (addMoment!, mprob.jl:123-155). */
typedef struct {
    int32_t np;              /* number of sampled parameters                              */
    int32_t nm;              /* number of moments                                         */
    int32_t ns;              /* simulated draws per moment (10000 in ObjExamples.jl:76)   */
    int32_t objective_id;    /* smm_objective_t                                           */
    const double* init;      /* [np] MProb.initial_value                                  */
    const double* lb;        /* [np]                                                      */
    const double* ub;        /* [np]                                                      */
    const double* mom;       /* [nm] data moments                                         */
    const double* w;         /* [nm] weights; NaN = no weight (ObjExamples.jl:96-97)      */
    const double* obj_params;/* objective specific blob, may be NULL                      */
    int32_t n_obj_params;
    int32_t reserved;
} smm_problem_t;

/* opts["dist_fun"] (AlgoBGP.jl:494,537): the reference takes any Julia function of two objective values; the device offers a
 * menu.  value_i belongs to the colder chain i < j of the pair; the pair swaps when the distance exceeds min_improve_i. */
typedef enum {
    SMM_DIST_MINUS   = 0,    /* value_i - value_j            (the default `-`: j is better by more than min_improve_i)      */
    SMM_DIST_ABSDIFF = 1,    /* |value_i - value_j|          (swap whenever the two differ by more than min_improve_i)      */
    SMM_DIST_RELDIFF = 2     /* (value_i - value_j)/|value_i| (j is better by more than the fraction min_improve_i)         */
} smm_dist_fun_t;

/* opts Dict of MAlgoBGP (AlgoBGP.jl:505-537) flattened; per-chain vectors are
 * supplied already expanded (sigma[i] = opts["sigma"]*temps[i], AlgoBGP.jl:508,518) and
 * GLOBAL (length N_global); the context uses entries [chain_offset, chain_offset+N). */
typedef struct {
    int32_t N;               /* chains owned by THIS context (local shard)                */
    int32_t maxiter;         /* history capacity T (BGPChain(n), AlgoBGP.jl:78)           */
    const double* sigma;     /* [N_global] initial proposal std-dev in [0,1]-space        */
    const double* acc_tuner; /* [N_global] AlgoBGP.jl:523                                 */
    const double* min_improve;/* [N_global] AlgoBGP.jl:522 (the exchange test of pair (i,j)
                                 reads chain i's threshold on every shard, :688)          */
    int32_t sigma_update_steps; /* AlgoBGP.jl:519                                         */
    int32_t smpl_iters;         /* AlgoBGP.jl:521                                         */
    double  sigma_adjust_by;    /* AlgoBGP.jl:520                                         */
    int32_t batch_size;         /* AlgoBGP.jl:524; must divide np                         */
    int32_t exchange_from_iter; /* 2 in the reference (AlgoBGP.jl:637)                    */
    uint64_t seed;
    int32_t chain_offset;    /* global id (0-based) of local chain 0                      */
    int32_t N_global;        /* total chains over all shards (== N on one GPU)            */
    int32_t device;          /* HIP device ordinal                                        */
    int32_t chol_per_chain;  /* 0: chol_L is one [np][np] factor shared by all chains; 1: [N_global][np][np] */
    const double* chol_L;    /* General Gaussian proposals ("Cholesky apply"): NULL = the reference's isotropic kernel
                                MvNormal(mu01, sigma) (AlgoBGP.jl:442).  Otherwise a lower-triangular factor L (row-major,
                                entries above the diagonal ignored) of the proposal's shape in [0,1]-space:
                                    x = mu01 + sigma_c * (L z),   z ~ N(0, I)      i.e. covariance sigma_c^2 L L'
                                with the chain's adaptive scalar sigma_c (AlgoBGP.jl:381-390) as the scale; L = diag(s)
                                is the per-parameter sigma vector hinted at AlgoBGP.jl:218.  Requires one proposal batch
                                (batch_size == np).  Numerical contract: (L z)_k = sum_{j<=k} L[k][j]*z[j], products
                                rounded, added left to right (no fma). */
    int32_t dist_fun;        /* smm_dist_fun_t: opts["dist_fun"], AlgoBGP.jl:537 — the exchange test of pair (i, j) is
                                dist_fun(value_i, value_j) > min_improve_i (:688).  0 = the reference's default `-`.           */
    int32_t reserved;
} smm_bgp_opts_t;

/* Injected randomness ("parity mode").  Any pointer may be NULL = use the built-in
 * counter-based generator (Philox4x32-10 + Box-Muller, documented in DESIGN.md).
 * Tables cover the LOCAL shard's chains, except `pairs` which is global. */
typedef struct {
    const double* probs_acc;   /* [T][N]  the MH uniforms, BGPChain.probs_acc AlgoBGP.jl:85    */
    const double* prop_normals;/* [T][K][np][N] standard normals for try k of mysample         */
    int32_t prop_tries;        /* K; tries beyond K raise SMM_ERR_NO_DRAW_IN_SUPPORT           */
    int32_t n_pairs;           /* pairs per iteration in `pairs` (N_global, or N_global-1 <3)  */
    const int32_t* pairs;      /* [T][n_pairs][2] 0-based global chain ids i<j, AlgoBGP.jl:656 */
    const double* Z;           /* [nm][ns] shock matrix of objfunc_norm (seed-1234 draws)      */
} smm_tables_t;

/* Caller-allocated SoA download buffers for iterations t0..t1-1 (0-based t = iter-1).
 * nt = t1-t0.  Any pointer may be NULL (skipped).  Mirrors history(c) AlgoBGP.jl:138-160
 * plus the Eval fields read by params()/allAccepted() (:117-131). */
typedef struct {
    double* value;      /* [nt][N]      evals[t].value                         */
    double* prob;       /* [nt][N]      evals[t].prob                          */
    double* curr_val;   /* [nt][N]                                              */
    double* best_val;   /* [nt][N]                                              */
    double* params;     /* [nt][np][N]                                          */
    double* sim_moments;/* [nt][nm][N]                                          */
    int32_t* best_id;   /* [nt][N]      1-based iteration                      */
    int32_t* exchanged; /* [nt][N]      1-based partner id, 0 none             */
    uint8_t* accepted;  /* [nt][N]                                              */
    int8_t*  status;    /* [nt][N]      evals[t].status                        */
} smm_history_t;

/* Per-chain scalar state (what save/readMalgo/restart! need besides history;
 * AlgoAbstract.jl:83-102, AlgoBGP.jl:759-884). Caller-allocated, any may be NULL. */
typedef struct {
    int32_t iter;        /* out/in: completed iterations                        */
    int32_t reserved;
    double* sigma;       /* [N]                                                  */
    double* accept_rate; /* [N]                                                  */
    double* la_value;    /* [N] last accepted record (getLastAccepted :217)      */
    double* la_prob;     /* [N]                                                  */
    double* la_params;   /* [np][N]                                              */
    double* la_sim_moments; /* [nm][N]                                           */
    int8_t* la_status;   /* [N]                                                  */
    int32_t* n_noex;     /* [N] iterations with exchanged==0 (set_acceptRate! :253-257) */
    int32_t* n_acc_noex; /* [N] accepted among those                             */
    double* best_val;    /* [N]                                                  */
    int32_t* best_id;    /* [N]                                                  */
} smm_state_t;

typedef struct {
    double step_ms;       /* device time of the last smm_bgp_step (hipEvent)           */
    double iter_kernel_ms;/* summed device time of the per-iteration chain kernel       */
    double exch_kernel_ms;/* summed device time of the exchange kernels                 */
    int64_t chain_evals;  /* chain evaluations performed by the last smm_bgp_step      */
    int32_t iters;
    int32_t reserved;
    double null_bracket_ms;/* summed device time of event pairs that bracket nothing: the per-bracket
                              overhead contained in iter_kernel_ms / exch_kernel_ms            */
} smm_timing_t;

int  smm_abi_version(void);
/* compile a user objective; errors (with the compiler log) through smm_last_error(NULL) */
int  smm_register_user_objective(const char* hip_source, int32_t* objective_id_out);
int  smm_register_user_objective_lanes(const char* hip_source, int32_t n_sums, int32_t lanes, int32_t* objective_id_out);
int  smm_device_count(void);

/* MAlgoBGP(m,opts) constructor, AlgoBGP.jl:505-537 + BGPChain ctor :78-109 */
int  smm_ctx_create(const smm_problem_t* prob, const smm_bgp_opts_t* opts,
                    const smm_tables_t* tables /* may be NULL */, void** ctx_out);
void smm_ctx_destroy(void* ctx);
const char* smm_last_error(void* ctx);

/* n_iters x computeNextIteration! (AlgoBGP.jl:589-640) incl. exchangeMoves!
 * (:647-716); single shard only (N_global == N). Blocks until the device is done. */
int  smm_bgp_step(void* ctx, int32_t n_iters);
/* same, but returns after enqueueing; smm_sync waits and reports device errors. */
int  smm_bgp_step_async(void* ctx, int32_t n_iters);
int  smm_sync(void* ctx);

/* Sharded form (one ctx per GPU), the three phases of one iteration:
 *   smm_bgp_local_step : next_eval for the local chains (AlgoBGP.jl:272-294)
 *   smm_bgp_export_records_dev : copy the last-accepted records of the local chains into
 *        rec_dev [N][RW] doubles, RW = smm_bgp_record_doubles(ctx) (value, prob, status,
 *        params[np], simM[nm], zero padded to an even count) — the RCCL all-gather payload
 *   smm_bgp_exchange_dev : exchangeMoves! over all N_global chains given the gathered
 *        records [N_global][RW] in global chain order (identical on every rank), applied
 *        to the local chains */
int  smm_bgp_local_step(void* ctx);
int  smm_bgp_record_doubles(void* ctx);
int  smm_bgp_export_records_dev(void* ctx, void* rec_dev);
int  smm_bgp_exchange_dev(void* ctx, const void* gathered_dev);
/* The values form of the exchange phase, for long records: instead of every record, only every chain's
 * VALUE goes to every rank, and the record a chain continues from goes to that chain's owner alone.  Ranks own equal
 * blocks of N chains (G = N_global / N ranks).  After smm_bgp_local_step:
 *   smm_bgp_export_values_dev(ctx, vals [N])              the local chains' last accepted values -> all-gather to [N_global]
 *   smm_bgp_a2a_pack_dev(ctx, vals_all [N_global], send)  resolves exchangeMoves! from the values and fills this rank's
 *        send buffer [G][cap][RW]: block b holds, in the order of the receiving chains, the records that chains of rank b
 *        continue from (cap = smm_bgp_a2a_capacity(ctx) records per block; the block to itself included)
 *   -> one all-to-all of equal blocks (RCCL: ncclAllToAll / all_to_all_single of cap * RW doubles per pair)
 *   smm_bgp_a2a_apply_dev(ctx, recv [G][cap][RW])         applies the swaps to the local chains
 * Same result as smm_bgp_export_records_dev + all-gather + smm_bgp_exchange_dev.  A block that would need more than cap
 * records raises SMM_ERR_EXCHANGE_CAPACITY at the next smm_sync (cap = min(N, 2 N / G + 64): about four times the expected
 * count for the reference's pair sampling). */
int  smm_bgp_a2a_capacity(void* ctx);
int  smm_bgp_export_values_dev(void* ctx, void* vals_dev);
int  smm_bgp_a2a_pack_dev(void* ctx, const void* vals_all_dev, void* send_dev);
int  smm_bgp_a2a_apply_dev(void* ctx, const void* recv_dev);
/* The same iteration in two enqueues instead of five.  Both buffers are [N_global][RW] in global chain order:
 *   smm_bgp_sharded_step(ctx, gathered_prev, gathered_next): resolves exchangeMoves! of the previous iteration
 *        from gathered_prev (the all-gathered records after that iteration's accept step; may be NULL before
 *        the first sharded step), runs next_eval for the local chains — every chain continues from its own or
 *        its donor's record taken straight from gathered_prev — and writes the new last-accepted records into
 *        THIS shard's slice of gathered_next (rows chain_offset .. chain_offset+N).  The caller then all-gathers
 *        gathered_next in place (RCCL: ncclAllGather with sendbuff = recvbuff + rank*N*RW) and passes it as
 *        gathered_prev of the next call; two buffers alternate.
 *   smm_bgp_sharded_finish(ctx, gathered): settles the last sharded step (its exchange, history, counters) into
 *        the context; required before smm_get_history / smm_get_state / the three-phase calls. */
int  smm_bgp_sharded_step(void* ctx, const void* gathered_prev_dev, void* gathered_next_dev);
int  smm_bgp_sharded_finish(void* ctx, const void* gathered_dev);
/* The p2p form of the sharded iteration: NO collective call at all.  The xGMI fabric of an MI355X node is point-to-point, so
 * the all-gather of the last-accepted records is done by the chains' accept step itself: every rank owns a WINDOW of device
 * memory that all other ranks map (HIP IPC between processes; plain device pointers between contexts of one process), a
 * chain's accept step stores its record, value and walk slot into every rank's window, and the next iteration's kernel reads
 * its own window.  Where the single shard needs one launch per iteration so does a shard (objfunc_norm, np == nm <= 4,
 * min_improve == 0, N_global <= 8192): every word in a window carries the iteration it belongs to, a reader that finds an older
 * one looks again (nobody waits for an acknowledgement, nothing is counted); the same objectives at 8192 < N_global <= 32768 (four
 * and eight shards of 4096): two launches, the exchange resolution reading the tagged words of its window itself and the chain kernel
 * pushing from its accept step; everywhere else: chain kernel + push kernel (stores, then one arrival count per 16 chains and rank)
 * + wait + resolve kernel.  The host enqueues nothing else.  Same results as every other form (bit-identical to the single shard).
 * smm.jl_amd/csrc/smm_p2p.hpp has the protocol.
 *   smm_bgp_p2p_init(ctx, handle_out, window_out): allocates this rank's window (rank = chain_offset / N, equal shards, at most
 *        8 ranks); handle_out (SMM_P2P_HANDLE_BYTES bytes, may be NULL) receives its hipIpcMemHandle_t for the other
 *        PROCESSES, window_out (may be NULL) its device pointer for other contexts of THIS process.
 *   smm_bgp_p2p_attach(ctx, rank, handle, window): rank's window, by IPC handle or by device pointer (exactly one non-NULL).
 *   smm_bgp_p2p_step(ctx, n): enqueues n iterations and returns (smm_sync waits).  All ranks call it with the same n, in the
 *        same order relative to each other's p2p calls (publications and pushes are counted).  A rank whose peers' stores never
 *        arrive gives up after ~4 s and reports SMM_ERR_HIP at the next smm_sync.  Where the context qualifies (smm_set_persistent,
 *        below) n >= 2 iterations behind a completed one are ONE launch per look-ahead window and rank; smm_sync of such steps is a
 *        rendezvous of the ranks (their error words travel through the windows): every rank must reach it.
 *   smm_bgp_p2p_finish(ctx): settles the last iteration into the context (required before smm_get_history / smm_get_state /
 *        the other stepping forms).  Callers must not destroy a context while a peer may still be stepping.
 *        A BARRIER ACROSS THE RANKS belongs between smm_bgp_p2p_finish (+ smm_sync) and the next smm_bgp_p2p_step: that step's
 *        first publication rewrites the windows with a new epoch, and a rank still in its finish would find the words of the
 *        last iteration replaced (a time-out in the tagged forms, other records without notice in the generic one).
 *        smm.jl_amd/dist.py::ShardedBGP.sync does it.
 *   Which of the forms a context steps in is decided from what every rank knows (population, objective, thresholds),
 *   never from a shard's own values.  A NaN value in an uploaded state (smm_set_state) reaches every window with the first
 *   publication: the rows form resolves such iterations on the exact values, the one-launch form (N_global <= 8192) has no
 *   second walk and reports SMM_ERR_HIP on every rank in the same iteration — step such a state once with smm_bgp_sharded_step
 *   (or as a single shard) first.  The persistent form reports it from inside its launch; the ranks agree on that at their
 *   rendezvous and replay the step on the forms above.
 *   What a shard in the persistent form sends per chain, iteration and PEER: the parameters and the value of its last accepted
 *   record as self-validating granules — (np + 1) x 16 bytes, 48 at two parameters; the rest of a record (prob, status, simulated
 *   moments) stays in the owner's window and is fetched by the one chain that continues from it (swap_ev_ij!, AlgoBGP.jl:734-749). */
#define SMM_P2P_HANDLE_BYTES 64
int  smm_bgp_p2p_init(void* ctx, void* ipc_handle_out, void** window_dev_out);
int  smm_bgp_p2p_attach(void* ctx, int32_t rank, const void* ipc_handle, void* window_dev);
int  smm_bgp_p2p_step(void* ctx, int32_t n_iters);
int  smm_bgp_p2p_finish(void* ctx);
/* the HIP stream all of the ctx's work is enqueued on (hipStream_t as void*).  (Large single shards, 8192 < N <= 32768, also own a
 * private second stream on which the NEXT window's exchange plan is computed ahead; the work on smm_stream waits for it through
 * events, so everything a caller can observe is ordered by smm_stream alone; smm_ctx_destroy drains both.) */
void* smm_stream(void* ctx);

/* batched evaluateObjective(m,p) (mprob.jl:175-188) for M parameter vectors
 * params [np][M] -> value[M], sim_moments[nm][M], status[M].  Used by tests and by
 * the other callers of evaluateObjective (slices.jl:153, econometrics.jl:42). */
int  smm_eval_batch(void* ctx, const double* params, int32_t M,
                    double* value, double* sim_moments, int8_t* status);
/* the same for objfunc_norm with options[:noseed] = true (ObjExamples.jl:71-75): evaluation i draws its own
 * shocks (generator keyed by base_seed + i) instead of the fixed seed-1234 matrix — the repetitions of
 * getSigma (econometrics.jl:125-145). */
int  smm_eval_batch_noseed(void* ctx, const double* params, int32_t M, uint64_t base_seed,
                           double* value, double* sim_moments, int8_t* status);

int  smm_get_history(void* ctx, int32_t t0, int32_t t1, smm_history_t* out);
int  smm_get_state(void* ctx, smm_state_t* out);
/* smm_set_state is also the recovery from a hard error (AlgoBGP.jl:341,409): the context steps again from the uploaded state.  A failure that
 * no entry point has handed to the caller yet — raised on the device by asynchronous steps nobody synchronised; the state readers do not
 * raise — is returned by THIS call, once, and nothing is uploaded: an error never disappears into a recovery the caller did not know it was
 * making.  The next smm_set_state goes through. */
int  smm_set_state(void* ctx, const smm_state_t* in, const smm_history_t* hist /* iterations 0..iter-1 */);
int  smm_get_timing(void* ctx, smm_timing_t* out);
/* on = 1: bracket every kernel of smm_bgp_step with hipEvents on the ctx stream so that
 * smm_get_timing reports iter_kernel_ms / exch_kernel_ms (sums over the last step; each bracket
 * contains the event overhead reported as null_bracket_ms).
 * on = 2: the kernels carry their own start/stop events (hipExtLaunchKernelGGL): the sums are the
 * dispatch-begin to dispatch-end durations the command processor stamps, i.e. what rocprofv3
 * --kernel-trace reports; null_bracket_ms = 0.   on = 0: off. */
int  smm_set_profiling(void* ctx, int32_t on);
/* The persistent form of smm_bgp_step and smm_bgp_p2p_step (smm.jl_amd/csrc/smm_chain_persist_loc.hpp, smm_chain_persist_gen.hpp, smm_chain_persist_tile.hpp;
 * replaces the loop of run!, AlgoAbstract.jl:38-45, over computeNextIteration!, AlgoBGP.jl:589-640): where a context qualifies a step of
 * n >= 2 iterations is ONE kernel launch per look-ahead window (<= 256 iterations) instead of one per iteration.  Results are
 * bit-identical.  A context qualifies with
 *   - objfunc_norm with at most two parameters / moments and ns <= 10240, one proposal batch, isotropic proposals, dist_fun = `-`, ONE
 *     min_improve >= 0 (or NaN) for all chains — 0, or the reference's default 0.5 (AlgoBGP.jl:522); a single shard also one PER chain,
 *     below —, at most one 16-chain tile per
 *     compute unit: a single shard of up to 4096 chains (smm_bgp_step), or a SHARD of a sharded run (smm_bgp_p2p_step: N a multiple
 *     of 16, N <= 4096 per rank, N_global <= 32768; the ring of tagged words then lives in every rank's window, a hard error inside
 *     such a launch is agreed upon by the ranks at their next smm_sync / smm_bgp_p2p_finish and replayed by every rank up to the
 *     failing iteration);
 *   - the banana objective, or a USER objective in the one-thread-per-evaluation form (smm_register_user_objective: the library compiles
 *     the persistent kernel once more with the user's source inside, through hiprtc, when the first such context is created: ~1.5 s),
 *     with at most 16 parameters / moments (one proposal batch, isotropic, min_improve == 0) on a single shard of up to 8192 chains in
 *     whole groups of 32;
 *   - objfunc_norm with MORE than two parameters (the reference's own larger examples have 6 and 18, Examples.jl:210-230, 232-319), the
 *     dense objectives (SMM_OBJ_DENSE, SMM_OBJ_DENSE2), or a USER objective in its MAP-REDUCE form (smm_register_user_objective_lanes with
 *     64, 128, 256 or 512 lanes per evaluation: the library compiles the persistent tile kernel once more with the user's source inside,
 *     through hiprtc, when the first such context is created; a tile's 512 lanes then evaluate 512 / lanes chains at a time with the
 *     stand-alone kernel's reduction order — the same bits; it pays while an evaluation is short beside the ~40 us of three launches per
 *     iteration: a long simulation fills the device better from its own launches, smm_set_persistent(ctx, 0)):
 *     one proposal batch or several, isotropic proposals, dist_fun = `-`, ONE min_improve >= 0 (or NaN) for all chains, a single shard of
 *     at most two 16-chain tiles per compute unit whose blocks fit the LDS (np = nm = 50: yes; 64 + 64: no).
 *   - (round 6) min_improve BY CHAIN, what the reference's API takes (a vector, AlgoBGP.jl:522; the pair (i, j) is tested against chain i's,
 *     :688): single shards of the objfunc_norm and tile forms above walk one threshold per chain, every one >= 0 or NaN; a context's single
 *     iterations keep the per-iteration kernels' walk on any thresholds.
 * A negative threshold, other dist_fun, Cholesky proposals, user objectives with 1024 lanes, shards of anything but objfunc_norm with at
 * most two parameters and one threshold: the per-iteration kernels.
 * on = 0 keeps the one-launch-per-iteration kernels (default: on).  A hard error of the algorithm inside such a launch is found at
 * the next call that checks (smm_sync, smm_bgp_step, the state readers): the library then repeats those iterations from the state
 * it saved on the one-launch-per-iteration path, so that the context stands at the failing iteration exactly as documented above.
 * All tiles of such a launch must be resident together; on a device that shows the process fewer compute units than it reports (a CU
 * mask, a partition) the first launch gives up after 0.4 s, the step is replayed on the per-iteration kernels, and after the second
 * such time-out the form is off for the context (smm_get_persistent says so).
 * smm_get_persistent: whether the next step would take this form; launches of it so far; repairs so far. */
int  smm_set_persistent(void* ctx, int32_t on);
int  smm_get_persistent(void* ctx, int32_t* available, int32_t* launches, int32_t* repairs);
/* one line naming the forms this context was given at creation — per-iteration chain kernel, where the exchange is walked, the stand-alone
 * resolution, the persistent form, the look-ahead plan and its window: "chain=iter_norm walk=inline_lean exchange=lean persistent=loc
 * plan=lds window=256" (diagnostic; tests/test_gpu_forms.py holds the table of what is chosen when) */
int  smm_describe(void* ctx, char* out, int32_t cap);
/* copy of the shock matrix actually used, [nm][ns] */
int  smm_get_Z(void* ctx, double* Z);

#ifdef __cplusplus
}
#endif
#endif /* SMMHIP_H */
