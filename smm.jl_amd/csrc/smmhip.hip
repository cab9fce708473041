// smmhip.hip — libsmmhip.so: hand-written HIP (gfx950 / MI355X) implementation of the BGP
// parallel-tempering iteration of floswald/SMM.jl behind the C ABI of include/smmhip.h.
//
// One iteration of computeNextIteration!(algo::MAlgoBGP) (src/mopt/AlgoBGP.jl:589-640) is ONE launch on a single
// shard (two when sharded or above 4096 chains):
//   k_chain_iter      : exchangeMoves! of the previous iteration (:647-716; the level walk of
//                       exchange_walk_tile, executed by every workgroup) and next_eval for every chain at once —
//                       materialise that exchange (swap_ev_ij! :734-749), proposal (:424-471), objective
//                       (mprob.jl:175-188 -> ObjExamples.jl:59-116), doAcceptReject! (:324-392),
//                       set_eval! (:220-245).  A 512-lane tile owns CT chains, the ns simulated draws are
//                       spread over the lanes, every shock z is re-used CT times from a register, moments
//                       are reduced by a transposed wave reduction and combined through LDS.
//   k_exch_resolve_*  : the same exchange resolution as a kernel of one workgroup (sharded path, larger
//                       populations, and whenever the result is needed before the next chain kernel).
// Files: smm_params.hpp (parameter block, layouts), smm_chain.hpp (chain kernel and its parts),
// smm_lookahead.hpp (k_pregen_rng, k_exch_plan), smm_exchange.hpp (stand-alone exchange kernels), this file (host).
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
#include <chrono>
#include <mutex>
#include <thread>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <math.h>
#include <algorithm>
#include <string>
#include <vector>
#include <type_traits>

#include "../../include/smmhip.h"
#include "smm_rng.hpp"

namespace {

using namespace smm;

#include "smm_params.hpp"
#include "smm_walk_lean.hpp"
#include "smm_propose.hpp"
#include "smm_chain.hpp"
#include "smm_p2p.hpp"
#include "smm_chain_norm.hpp"
#include "smm_chain_persist.hpp"
#include "smm_chain_persist_gen.hpp"
#include "smm_chain_persist_loc.hpp"
#include "smm_chain_persist_tile.hpp"
#include "smm_lookahead.hpp"
#include "smm_exchange.hpp"
#include "smm_cone_big.hpp"

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
thread_local std::string g_create_err;

// Test seams.  The shipped library (libsmmhip.so) reads two environment variables, both diagnostics: SMMHIP_TS (phase stamps for
// tools/) and SMMHIP_DBG (timing experiments; results invalid).  Everything that changes which kernel runs or how much it may
// hold — forcing an exchange kernel, switching a fast path off, a tiny capacity, poisoned allocations — exists only in the build
// the tests load next to it (libsmmhip_hooks.so, -DSMM_TEST_HOOKS): a stray variable cannot change what production runs.
#ifdef SMM_TEST_HOOKS
#define SMM_HOOK(name) getenv(name)
#else
#define SMM_HOOK(name) ((const char*)nullptr)
#endif

// ------------------------------------------------------------------------------------------
// user objectives: compiled with hiprtc (loaded lazily, the library does not link against it)
// ------------------------------------------------------------------------------------------
struct UserObjective {
    std::vector<char> code; int lanes = 0; /* 0: one thread per chain; else lanes per chain (map-reduce form) */
    std::string source;                    // the user's text (one thread per chain form): compiled once more INTO the persistent kernel on demand
    std::vector<char> persist_code;        // k_chain_persist_gen with the user's objective inside (smm_chain_persist_gen.hpp, SMM_GEN_USER)
    int persist_state = 0;                 // 0: not tried yet, 1: persist_code stands, -1: did not compile (persist_log)
    std::string persist_log;
    int n_sums = 1;                        // map-reduce form: partial sums per lane
    std::vector<char> tile_code;           // k_chain_persist_tile with the user's map-reduce objective inside (smm_chain_persist_tile.hpp, SMM_TILE_USER)
    int tile_state = 0;                    // 0: not tried yet, 1: tile_code stands, -1: did not compile (tile_log)
    std::string tile_log;
};
// the device headers the persistent kernel is made of, as text: hiprtc compiles them together with the user's source
struct EmbeddedSource { const char* name; const char* text; };
const EmbeddedSource g_embedded[] = {
#include "smm_embedded.inc"
};
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

const char* USER_PRELUDE_LANES =
    "#define SMM_USER_PARTIAL extern \"C\" __device__ void smm_user_partial\n"
    "#define SMM_USER_FINISH extern \"C\" __device__ void smm_user_finish\n"
    "extern \"C\" __device__ void smm_user_partial(const double* theta, int np, const double* udata, int n_udata, int lane, int n_lanes,\n"
    "                                            double* partial);\n"
    "extern \"C\" __device__ void smm_user_finish(const double* theta, int np, const double* totals, int n_sums, const double* mom,\n"
    "                                           const double* w, int nm, const double* udata, int n_udata, double* sim_moments,\n"
    "                                           double* value, int* status);\n";
// one workgroup of n_lanes threads per evaluation.  Numerical contract of the reduction: inside a
// wave the 64 partials are combined by the halving tree (offsets 32,16,...,1), the wave totals are added left to right.
const char* USER_KERNEL_LANES =
    "\nextern \"C\" __global__ void smm_user_eval_kernel(const double* theta, int N, int np, const double* mom, const double* w, int nm,\n"
    "        const double* udata, int n_udata, double* simM, double* value, int* status) {\n"
    "    __shared__ double wsum[16][SMM_NSUMS];\n"
    "    const int c = blockIdx.x, lane = threadIdx.x, nl = blockDim.x;\n"
    "    double part[SMM_NSUMS];\n"
    "    for (int i = 0; i < SMM_NSUMS; ++i) part[i] = 0.0;\n"
    "    smm_user_partial(theta + (size_t)c * np, np, udata, n_udata, lane, nl, part);\n"
    "    for (int i = 0; i < SMM_NSUMS; ++i) {\n"
    "        double a = part[i];\n"
    "        for (int off = 32; off >= 1; off >>= 1) a = a + __shfl_xor(a, off, 64);\n"
    "        if ((lane & 63) == 0) wsum[lane >> 6][i] = a;\n"
    "    }\n"
    "    __syncthreads();\n"
    "    if (lane == 0) {\n"
    "        double tot[SMM_NSUMS];\n"
    "        for (int i = 0; i < SMM_NSUMS; ++i) { double a = wsum[0][i]; for (int wv = 1; wv < nl / 64; ++wv) a = a + wsum[wv][i]; tot[i] = a; }\n"
    "        int st = 1; double v = 0.0;\n"
    "        smm_user_finish(theta + (size_t)c * np, np, tot, SMM_NSUMS, mom, w, nm, udata, n_udata, simM + (size_t)c * nm, &v, &st);\n"
    "        value[c] = v; status[c] = st;\n"
    "    }\n"
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

// k_chain_persist_gen with a user objective inside: the library's own device headers (embedded as text) + the user's source through
// hiprtc, once per registered objective and on demand (the first context that qualifies for the persistent form: ~1.5 s).  false: the
// form is not available for this objective (u.persist_log says why); the per-iteration launches serve it as before.
bool user_persist_compile(UserObjective& u) {   // (g_user_mutex held)
    if (u.persist_state != 0) return u.persist_state > 0;
    u.persist_state = -1;
    if (u.source.empty() || u.lanes != 0) { u.persist_log = "not the one-thread-per-chain form"; return false; }
    std::string err;
    if (!g_rtc.load(err)) { u.persist_log = err; return false; }
    std::string tu =
        "#include <type_traits>\n#include <stdint.h>\n#include <math.h>\n"
        "#define SMM_USER_OBJECTIVE extern \"C\" __device__ void smm_user_objective\n"
        "extern \"C\" __device__ void smm_user_objective(const double* theta, int np, const double* mom, const double* w, int nm,\n"
        "        const double* udata, int n_udata, double* sim_moments, double* value, int* status);\n";
    tu += u.source;
    tu += "\n#define SMM_GEN_USER 1\n#include \"smmhip.h\"\n#include \"smm_rng.hpp\"\nusing namespace smm;\n#include \"smm_params.hpp\"\n"
          "#include \"smm_walk_lean.hpp\"\n#include \"smm_propose.hpp\"\n#include \"smm_chain.hpp\"\n#include \"smm_p2p.hpp\"\n#include \"smm_chain_norm.hpp\"\n"
          "#include \"smm_chain_persist.hpp\"\n#include \"smm_chain_persist_gen.hpp\"\n";
    std::vector<const char*> names, texts;
    for (const EmbeddedSource& e : g_embedded) { names.push_back(e.name); texts.push_back(e.text); }
    // (what the headers ask the host's toolchain for: hiprtc brings its own runtime header and has no system headers to lean on)
    static const char* stub_stdint = "typedef unsigned int uint32_t; typedef unsigned long uint64_t; typedef int int32_t; typedef long int64_t;\n"
                                     "typedef unsigned short uint16_t; typedef unsigned char uint8_t; typedef signed char int8_t; typedef short int16_t;\n";
    static const char* stub_math = "#ifndef INFINITY\n#define INFINITY __builtin_huge_val()\n#endif\n#ifndef NAN\n#define NAN __builtin_nan(\"\")\n#endif\n";
    names.push_back("hip/hip_runtime.h"); texts.push_back("\n");
    names.push_back("stdint.h"); texts.push_back(stub_stdint);
    names.push_back("math.h"); texts.push_back(stub_math);
    hiprtcProgram prog = nullptr;
    if (g_rtc.create(&prog, tu.c_str(), "smm_user_persist.hip", (int)names.size(), texts.data(), names.data()) != HIPRTC_SUCCESS) {
        u.persist_log = "hiprtcCreateProgram failed";
        return false;
    }
    const char* opts[] = {"--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-fast-math", "-std=c++17"};
    const hiprtcResult rc = g_rtc.compile(prog, 5, opts);
    if (rc != HIPRTC_SUCCESS) {
        size_t n = 0;
        g_rtc.log_size(prog, &n);
        std::string log(n, ' ');
        if (n) g_rtc.log(prog, &log[0]);
        u.persist_log = log;
        g_rtc.destroy(&prog);
        return false;
    }
    size_t cs = 0;
    g_rtc.code_size(prog, &cs);
    u.persist_code.resize(cs);
    g_rtc.code(prog, u.persist_code.data());
    g_rtc.destroy(&prog);
    u.persist_state = 1;
    return true;
}

// k_chain_persist_tile with a user objective in its MAP-REDUCE form inside (smm_register_user_objective_lanes; SMM_TILE_USER in
// smm_chain_persist_tile.hpp): the same recipe, on demand, once per registered objective
bool user_tile_compile(UserObjective& u) {   // (g_user_mutex held)
    if (u.tile_state != 0) return u.tile_state > 0;
    u.tile_state = -1;
    if (u.source.empty() || u.lanes == 0) { u.tile_log = "not the map-reduce form"; return false; }
    std::string err;
    if (!g_rtc.load(err)) { u.tile_log = err; return false; }
    std::string tu = "#include <type_traits>\n#include <stdint.h>\n#include <math.h>\n";
    tu += USER_PRELUDE_LANES;
    tu += u.source;
    tu += "\n#define SMM_TILE_USER 1\n#include \"smmhip.h\"\n#include \"smm_rng.hpp\"\nusing namespace smm;\n#include \"smm_params.hpp\"\n"
          "#include \"smm_walk_lean.hpp\"\n#include \"smm_propose.hpp\"\n#include \"smm_chain.hpp\"\n#include \"smm_p2p.hpp\"\n#include \"smm_chain_norm.hpp\"\n"
          "#include \"smm_chain_persist.hpp\"\n#include \"smm_chain_persist_loc.hpp\"\n#include \"smm_chain_persist_tile.hpp\"\n";
    std::vector<const char*> names, texts;
    for (const EmbeddedSource& e : g_embedded) { names.push_back(e.name); texts.push_back(e.text); }
    static const char* stub_stdint = "typedef unsigned int uint32_t; typedef unsigned long uint64_t; typedef int int32_t; typedef long int64_t;\n"
                                     "typedef unsigned short uint16_t; typedef unsigned char uint8_t; typedef signed char int8_t; typedef short int16_t;\n";
    static const char* stub_math = "#ifndef INFINITY\n#define INFINITY __builtin_huge_val()\n#endif\n#ifndef NAN\n#define NAN __builtin_nan(\"\")\n#endif\n";
    names.push_back("hip/hip_runtime.h"); texts.push_back("\n");
    names.push_back("stdint.h"); texts.push_back(stub_stdint);
    names.push_back("math.h"); texts.push_back(stub_math);
    hiprtcProgram prog = nullptr;
    if (g_rtc.create(&prog, tu.c_str(), "smm_user_persist_tile.hip", (int)names.size(), texts.data(), names.data()) != HIPRTC_SUCCESS) {
        u.tile_log = "hiprtcCreateProgram failed";
        return false;
    }
    const std::string nsd = "-DSMM_NSUMS=" + std::to_string(u.n_sums);
    const char* opts[] = {"--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-fast-math", "-std=c++17", nsd.c_str()};
    const hiprtcResult rc = g_rtc.compile(prog, 6, opts);
    if (rc != HIPRTC_SUCCESS) {
        size_t n = 0;
        g_rtc.log_size(prog, &n);
        std::string log(n, ' ');
        if (n) g_rtc.log(prog, &log[0]);
        u.tile_log = log;
        g_rtc.destroy(&prog);
        return false;
    }
    size_t cs = 0;
    g_rtc.code_size(prog, &cs);
    u.tile_code.resize(cs);
    g_rtc.code(prog, u.tile_code.data());
    g_rtc.destroy(&prog);
    u.tile_state = 1;
    return true;
}

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
    int win_cap = 0;           // iterations per window of pre-generated randomness
    int plan_cap = 0;          // iterations per window of exchange plans
    double* win_rb = nullptr;
    unsigned long long* win_plan = nullptr;
    double* win_plan_mi = nullptr;
    uint32_t *win_lv_pairs = nullptr, *win_lv_off = nullptr, *win_lv_pairs_p = nullptr, *win_lv_offp = nullptr;
    bool rows_exchange = false;  // min_improve == 0, 8192 < N_global <= 32768: k_exch_resolve_rows (falls back to k_exch_resolve_key's walk)
    uint32_t *win_lv_rows = nullptr, *win_lv_rowinfo = nullptr, *slots17 = nullptr, *nan_flags = nullptr;
    int32_t *a2a_send_idx = nullptr, *a2a_send_cnt = nullptr, *a2a_rowidx = nullptr;   // the values form of the sharded exchange
    int a2a_cap = 0, a2a_G = 0;
    bool a2a_open = false;
    int slots_iter = -1;        // single shard, rows exchange: the accept step of this iteration wrote the resolution's initial slots (no pre-pass)
    int xk = 0;                 // ExchKernel: the stand-alone exchange resolution of this context (choose_exchange)
    bool exch_done = false;     // the three-phase / values forms: exchangeMoves! of iteration `iter` has been applied (cleared by the next local step)
    double* vals_buf[2] = {nullptr, nullptr};   // KParams::vals / vals_out, by iteration parity (point_values)
    uint2* slot8_buf[2] = {nullptr, nullptr};
    bool deep_plan = false;      // an injected pair list has an iteration of more than LV_MAXLEV dependency levels
    bool nan_values = false;     // the uploaded state holds NaN values (smm_set_state)
    bool gen_lean = false;       // k_chain_iter walks inline on the lean form (16-byte slots) when the plan fits it
    bool gen_keys = false;       // ... on the lean KEY form (8-byte slots): single shards of 4096 < N <= 8192 chains without a simulation (two 16-chain tiles per workgroup)
    bool cone_big = false;       // large single shards of objfunc_norm (8192 < N <= 32768): the narrow chain kernel's tiles walk their own, locally numbered cones (smm_cone_big.hpp)
    uint32_t* cb_scratch = nullptr;
    std::vector<uint32_t> cone_big_ok;   // per iteration of the plan window: its cones fit their caps
    // ... their windows are planned AHEAD: the plan of a window depends on (seed, iteration) only, so while the chain kernels of one
    // window run, the three plan kernels of the next run beside them on a second stream into the other of two sets of tables
    struct PlanSet {
        uint32_t *lv_pairs = nullptr, *lv_off = nullptr, *lv_rows = nullptr, *lv_rowinfo = nullptr, *cone_ok = nullptr, *cone_hdr = nullptr, *cone_pairs = nullptr;
        double* lv_mi = nullptr;
        uint16_t* cone_gather = nullptr;
        uint32_t* ok_host = nullptr;   // (pinned) cone_ok as the host reads it
        hipEvent_t done = nullptr;
        int t0 = 0, w = 0;             // iterations [t0, t0 + w) are (being) planned into this set
    } ps[2];
    int ps_act = 0;                      // the set the chain kernels read
    bool plan_ahead = false;
    hipStream_t pstream = nullptr;
    hipEvent_t ev_free = nullptr;        // main stream: every launch that reads the set about to be planned into has been enqueued before it (plan_window_into)
    bool dense_keys = false;     // ... and the dense objective's tiles (one 16-chain tile per workgroup, N <= 4096): the walk's slots and lists UNDER the tile's blocks
    bool lean_resolve = false;   // one min_improve >= 0 for all chains, N_global <= 8192 (~7400 when > 0): k_exch_resolve_lean is the stand-alone resolve kernel
    double* win_lv_mi = nullptr;
    bool lvl_exchange = false;
    bool lvl_soa_exchange = false;   // XLVL_MAX < N_global <= XLDS_MAX: level walk on split chain slots
    int tpw = 1;                // tiles per workgroup of k_chain_iter (2 with the inline walk: one walk per CU)
    bool inline_walk = false;   // the exchange walk runs in the prologue of the next k_chain_iter (SMMHIP_INLINE_WALK=0: off)
    const double* ext_rec_in = nullptr;   // sharded_step: donor records come from / results go to the caller's gather buffers
    double* ext_rec_out = nullptr;
    bool pending_ext = false;   // sharded_step: the exchange of iteration `iter` is still to be resolved from the gathered records
    int u_lanes = 0;                // user objective: lanes per evaluation (0 = one thread per chain)
    int n_objp = 0;                 // doubles in P.objp
    hipModule_t umod = nullptr;     // user objective: this context's module and kernel
    hipFunction_t ufn = nullptr;
    hipModule_t upmod = nullptr;    // ... and the persistent kernel compiled with it inside (smm_chain_persist_gen.hpp, SMM_GEN_USER)
    hipFunction_t upfn = nullptr;
    hipModule_t utmod = nullptr;    // ... and the persistent TILE kernel with its map-reduce form inside (smm_chain_persist_tile.hpp, SMM_TILE_USER)
    hipFunction_t utfn = nullptr;
    int u_nsums = 1;                // ... its partial sums per lane
    bool persist_user = false;      // persist_gen launches upfn
    bool defer_resolve = false;     // the exchange of an iteration is left unresolved until somebody needs it (the next launch may be the persistent kernel's)
    bool rec_external = false;  // the records after the last accept step were written to the caller's gather buffer (sharded_step)
    bool unresolved = false;    // exchangeMoves! of iteration `iter` is still to be resolved (inline, or by resolve_now)
    bool big_exchange = false;   // 8192 < N_global <= 65535: level plan and walk in global memory
    bool key_exchange = false;   // ... up to 32768: the walk in LDS on 4-byte slots (src + 16-bit order key), k_exch_resolve_key
    uint32_t* big_scratch = nullptr;
    int lvl_wg = 1024;
    int rng_t0 = 0, rng_w = 0;    // window currently held: iterations [t0, t0+w)
    int plan_t0 = 0, plan_w = 0;
    bool lds_exchange = false;
    int ct = 8;
    bool norm_fast = false;     // objfunc_norm with np == nm <= 4 and one proposal batch: k_chain_iter_norm (16-chain tiles)
    int failed = 0;             // a hard device error (AlgoBGP.jl:341,409) stopped the run at iteration `iter`: sticky until smm_set_state
    // the p2p form of the sharded iteration (smm_p2p.hpp)
    unsigned char* p2p_mine = nullptr;         // this rank's window (null: smm_bgp_p2p_init not called)
    void* p2p_opened[P2P_MAXG] = {};           // peers' windows opened through HIP IPC (closed with the context)
    unsigned p2p_attached = 0;                 // bit r: rank r's window is known
    unsigned long long p2p_seq = 0;            // pushes so far (every rank counts the same)
    bool p2p_current = false;                  // the windows hold the records after iteration `iter`
    bool p2p_inline = false;                   // the inline form is available: k_chain_iter_norm_p2p walks inline and pushes from its epilogue
    bool norm_narrow = false;                  // k_chain_iter_norm_narrow instead of k_chain_iter_norm<., false>: shards of more than one round of tiles
    bool cone = false;                         // the key form of k_chain_iter walks its workgroups' cones (smm_cone.hpp)
    bool p2p_rows = false;                     // the same kernel without the walk + k_exch_resolve_rows<., true> on the window's slots
    bool p2p_mode_inline = false;              // ... and is what the windows currently hold (decided at every publication)
    bool p2p_unwaited = false;                 // nobody has waited for the arrivals of the last push yet
    double* ext_vals_out = nullptr;            // p2p generic form: the accept step's values go into the window
    // the persistent chain kernel (smm_chain_persist.hpp): one launch for a run of iterations
    bool persist = false;                      // this context can run it (objfunc_norm np <= 2, single shard of at most one tile per CU, key walk)
    bool persist_gen = false;                  // ... its form for objectives without a simulation (smm_chain_persist_gen.hpp: banana, 4096 < N <= 8192)
    bool persist_loc = false;                  // ... on locally numbered cones (smm_chain_persist_loc.hpp): thresholds (min_improve > 0), shards
    bool persist_tile = false;                 // ... its form for objectives a whole tile evaluates (smm_chain_persist_tile.hpp): objfunc_norm of any size, the dense simulation
    bool persist_wide = false;                 // ... its 16-byte slots: one min_improve > 0 (or NaN) for all chains
    unsigned char* prw = nullptr;              // persist_loc: the ring's window (pr_win_layout) — a single shard's own allocation, a shard's: inside its p2p window
    bool persist_sh = false;                   // ... as a SHARD of a sharded run: the ring lives in every rank's p2p window (smm_bgp_p2p_step)
    bool persist_sh_big = false;               // ... of a population past 8192 chains: k_exch_plan_big + k_cone_chains + k_cone_tiles list its tiles' cones (locally numbered)
    size_t prw_off = 0;                        // persist_sh: offset of the ring's window inside the p2p window
    int p2p_ranks_here = 1;                    // ranks whose windows live on THIS device (this one included): their tiles must all be resident together
    int persist_max_tiles = 0;                 // tiles of the persistent kernel this device holds at once (occupancy x compute units)
    bool persist_proven = false;               // a launch of the persistent form has come through: its spins may last P2P_TIMEOUT_TICKS from now on
    int persist_strikes = 0;                   // time-outs so far (two: the form is off for the context)
    int persist_on = 1;                        // smm_set_persistent
    bool persist_broken = false;               // a launch gave up waiting (tiles not resident together?): the form is off for this context
    uint32_t pr_epoch = 0;                     // launches so far
    int persist_launches = 0, persist_repairs = 0;
    int pregen_seen_launches = 0;              // persist_launches when the last window of randomness blocks was made (ensure_windows)
    int pr_ring_k = PR_K, pr_slow_tile = -1, pr_slow_ticks = 0;   // (test build: SMMHIP_PR_RING, SMMHIP_PR_SLOW_TILE, SMMHIP_PR_SLOW_US)
    bool failed_told = false;   // `failed` has been returned to the caller by some entry point
    bool in_repair = false;
    // ... and what persist_repair restores when a launch of it ends with the error word set: the state before the FIRST such launch
    // since the last check of the error word
    bool snap_valid = false;
    int snap_iter = 0, snap_cur = 0, snap_slots_iter = -1;
    bool snap_pending = false, snap_prev_open = false, snap_unresolved = false, snap_exch_done = false;
    double *snap_cs = nullptr, *snap_rec = nullptr, *snap_vals[2] = {nullptr, nullptr}, *hist_fill = nullptr;
    uint2* snap_slot8[2] = {nullptr, nullptr};
    unsigned long long* snap_xres = nullptr;
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
    if (const char* f = SMM_HOOK("SMMHIP_FILL"))   // test hook: every allocation starts as this byte (reads of memory nobody wrote show up)
        HIPCHK(hipMemset(p, atoi(f), (n ? n : 1) * sizeof(T)));
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
// ... as the LDS layouts see it: 3 = the dense objective's spec v2 (SMM_OBJ_DENSE2: internally SMM_OBJ_DENSE with the 256 x 256 stage's
// operand set; its first hidden layer is staged in the partial sums' region)
int lay_kind(const Ctx* c);

// the two value arrays (and slot arrays) alternate by iteration: reads of iteration t_read's values, writes of iteration t_write's
void point_values(const Ctx* c, KParams& P, int t_read, int t_write) {
    P.vals = c->vals_buf[t_read & 1]; P.vals_out = c->vals_buf[t_write & 1];
    P.slot8 = c->slot8_buf[t_read & 1]; P.slot8_out = c->slot8_buf[t_write & 1];
}

size_t tile_smem_base(const Ctx* c, int ct) {
    const KParams& P = c->P;
    return tile_smem_doubles(ct, P.np, P.nm, P.RW, P.HW, P.RBW, lay_kind(c)) * sizeof(double);
}
int lay_kind(const Ctx* c) { const int k = obj_kind(c->obj); return k == 2 && c->P.dense_A2f ? 3 : (c->obj == SMM_OBJ_USER && c->u_lanes > 0) ? 4 : k; }
// dynamic LDS of k_chain_persist_tile for this context (a user objective's wave totals: 16 chains x lanes / 64 groups x its sums)
size_t persist_tile_smem(const Ctx* c) {
    const KParams& P = c->P;
    return pt_layout(P.np, P.nm, P.RW, P.HW, P.RBW, lay_kind(c), P.dense_nOt, PT_CT * (c->u_lanes / 64) * c->u_nsums).total;
}
size_t norm_smem(const Ctx* c) {   // k_chain_iter_norm: [walk: chain slots | pair list] theta, partial sums, parked state
    const size_t b = (size_t)c->P.tile_off * sizeof(double) + norm_tile_doubles(c->P.np) * sizeof(double);
    return c->cone_big ? std::max(b, cone_local_lds_bytes()) : b;   // (the local cone walk lies UNDER the tile's blocks)
}
size_t tile_smem(const Ctx* c, int ct, int tpw = 1) {   // dynamic LDS of k_chain_iter: tpw tiles; with the inline exchange
    if (c->norm_fast) return norm_smem(c);
    const size_t base = tile_smem_base(c, ct);           // walk its chain slots in front and its pair list under the tiles
    const size_t tiles = (size_t)tpw * ((base + 15) & ~(size_t)15);
    if (!c->inline_walk) return tiles;
    if (c->dense_keys) return std::max(tiles, (size_t)(((c->P.Ng + 3) & ~3) + 4) * 8 + std::max((size_t)CONE_LEVELS * 64 * 4, (size_t)lean_walk_Kp(c->P.plan_K) * 4));
    if (c->gen_keys) return (size_t)(((c->P.Ng + 3) & ~3) + 4) * 8 + std::max(tiles, (size_t)lean_walk_Kp(c->P.plan_K) * 4);
    return c->gen_lean ? tile_lean_slot_bytes(c->P.Ng) + std::max(tiles, (size_t)lean_walk_Kp(c->P.plan_K) * 4)
                       : walk_slot_bytes(c->P.Ng) + std::max(tiles, (size_t)c->P.plan_K * 4);
}
size_t plan_lds_bytes(int Ng, int K) { return (size_t)(Ng + 2) * 4 + (size_t)K * 8 + (size_t)K * 4 + 128 + 16; }
#ifdef SMM_TEST_HOOKS
size_t resolve_lds_bytes(int Ng) { return (size_t)Ng * 16 + 16; }
#endif
size_t resolve_lvl_soa_bytes(int Ng, int K) { return (size_t)Ng * 12 + (size_t)K * 4 + 16; }
size_t resolve_lvl_bytes(int Ng, int K) { return (size_t)Ng * 16 + (size_t)K * 12 + 64 * 8 + 64; }

int exchange_K(const Ctx* c) { return c->P.pairtab ? c->P.n_pairs_tab : n_exchange_pairs(c->P.Ng); }
bool exchange_active(const Ctx* c, int t) { return t >= c->exchange_from && c->P.Ng > 1; }  // AlgoBGP.jl:637

void launch_cone_big(Ctx* c, const KParams& Pw, int W, const uint32_t* lv_pairs, const uint32_t* lv_off, hipStream_t st) {
    hipLaunchKernelGGL(k_cone_chains, dim3(W), dim3(XWG), (size_t)Pw.Ng * 4, st, Pw, lv_pairs, lv_off, c->cb_scratch);
    hipLaunchKernelGGL(k_cone_tiles, dim3((unsigned)(((Pw.cone_tiles + CONEB_WAVES - 1) / CONEB_WAVES) * ((W + 7) & ~7))), dim3(64 * CONEB_WAVES), cone_tiles_lds_bytes(Pw.plan_K),
                       st, Pw, W, (const uint32_t*)c->cb_scratch);
    HIPCHK(hipGetLastError());
}
// the plan window starting at iteration t into set k, on the plan stream (the kernels' scratch is theirs alone: one window at a time
// there) — behind every launch enqueued on the main stream so far: the ones that read set k are all among them (k is not the active set)
void plan_window_into(Ctx* c, int k, int t) {
    Ctx::PlanSet& S = c->ps[k];
    KParams Pw = c->P;
    HIPCHK(hipEventRecord(c->ev_free, c->stream));
    HIPCHK(hipStreamWaitEvent(c->pstream, c->ev_free, 0));
    Pw.cone_ok = S.cone_ok; Pw.cone_hdr = S.cone_hdr; Pw.cone_pairs = S.cone_pairs; Pw.cone_gather = S.cone_gather;
    const int W = std::min(c->plan_cap, Pw.T - t + 1);
    hipLaunchKernelGGL(k_exch_plan_big, dim3(W), dim3(XWG), plan_big_lds_bytes(Pw.Ng), c->pstream, Pw, t, c->big_scratch, S.lv_pairs, S.lv_mi, S.lv_off, S.lv_rows, S.lv_rowinfo);
    launch_cone_big(c, Pw, W, S.lv_pairs, S.lv_off, c->pstream);
    HIPCHK(hipMemcpyAsync(S.ok_host, S.cone_ok, (size_t)W * 4, hipMemcpyDeviceToHost, c->pstream));
    HIPCHK(hipEventRecord(S.done, c->pstream));
    S.t0 = t; S.w = W;
}

// iteration t lies behind the plan window, the exchange of t - 1 is still to be walked: does the window planned ahead hold both?
bool plan_ahead_covers(const Ctx* c, int t) {
    const Ctx::PlanSet& S = c->ps[c->ps_act ^ 1];
    return c->plan_ahead && S.w >= 2 && S.t0 == t - 1;
}

// make the look-ahead tables cover iteration t (1-based): a new window simply starts at t
void ensure_windows(Ctx* c, int t, bool rng = true) {
    KParams& P = c->P;
    // (k_chain_iter_norm generates its randomness itself unless tables are injected: no randomness blocks)
    const bool pregen = rng && !(c->norm_fast && !P.user_ntab && !P.user_utab);
    if (pregen && !(t >= c->rng_t0 && t < c->rng_t0 + c->rng_w)) {
        int W = std::min(c->win_cap, P.T - t + 1);
        // (k_chain_persist_gen draws in the kernel too: blocks only for the iterations between its launches)
        // — and only where such launches are really being made: a caller that steps one iteration at a time (the reference's run! loop
        // over computeNextIteration!, AlgoAbstract.jl:38-45) never takes the persistent form and keeps its full windows
        if ((c->persist_gen || c->persist_tile) && c->persist && c->persist_on && !c->persist_broken && !c->in_repair && !P.user_ntab && !P.user_utab &&
            c->persist_launches != c->pregen_seen_launches) W = std::min(W, 2);
        c->pregen_seen_launches = c->persist_launches;
        const size_t Q = (size_t)(P.np + 1) / 2;
        const size_t per_iter = (size_t)P.rb_tries * Q * P.N;   // (< 2^31: checked at creation)
        hipLaunchKernelGGL(k_pregen_rng, dim3((unsigned)((per_iter + 255) / 256), (unsigned)W), dim3(256), 0, c->stream, P, t, W, c->win_rb);
        c->rng_t0 = t; c->rng_w = W;
        P.rb = c->win_rb; P.rb_t0 = t;
    }
    if (c->plan_ahead && !(t >= c->plan_t0 && t < c->plan_t0 + c->plan_w)) {
        const int nx = c->ps_act ^ 1;
        Ctx::PlanSet& S = c->ps[nx];
        // (the window planned ahead starts with the LAST iteration of the one before it, so that an exchange of that iteration still
        // to be walked by this launch's tiles finds its cones in the new window: plan_ahead_covers)
        if (!(S.w > 0 && (S.t0 == t || (S.t0 == t - 1 && S.w >= 2)))) plan_window_into(c, nx, t);   // (the first window, a jump: not the window planned ahead)
        HIPCHK(hipEventSynchronize(S.done));                         // the host reads the window's flags (long there when planned ahead)
        HIPCHK(hipStreamWaitEvent(c->stream, S.done, 0));
        c->ps_act = nx; c->plan_t0 = S.t0; c->plan_w = S.w;
        P.plan_t0 = S.t0;
        P.lv_rows = S.lv_rows; P.lv_rowinfo = S.lv_rowinfo; P.lv_pairs = S.lv_pairs; P.lv_mi = S.lv_mi; P.lv_off = S.lv_off;
        P.cone_ok = S.cone_ok; P.cone_hdr = S.cone_hdr; P.cone_pairs = S.cone_pairs; P.cone_gather = S.cone_gather;
        c->cone_big_ok.assign(S.ok_host, S.ok_host + S.w);
        // the next window, into the set the launches enqueued so far read
        c->ps[nx ^ 1].w = 0;
        if (S.t0 + S.w <= P.T) plan_window_into(c, nx ^ 1, S.t0 + S.w - (S.w >= 2 && c->plan_cap >= 2 ? 1 : 0));
    }
    if (c->big_exchange && !c->plan_ahead && !(t >= c->plan_t0 && t < c->plan_t0 + c->plan_w)) {
        const int W = std::min(c->plan_cap, P.T - t + 1);
        hipLaunchKernelGGL(k_exch_plan_big, dim3(W), dim3(XWG), plan_big_lds_bytes(P.Ng), c->stream, P, t, c->big_scratch, c->win_lv_pairs, c->win_lv_mi,
                           c->win_lv_off, c->win_lv_rows, c->win_lv_rowinfo);
        c->plan_t0 = t; c->plan_w = W;
        P.plan_t0 = t;
        if (c->cone_big) {   // the tiles' locally numbered cones, from the plan's scratch (pairs in list order, their levels)
            launch_cone_big(c, P, W, c->win_lv_pairs, c->win_lv_off, c->stream);
            // a cone that does not fit its caps (a pair list of very deep dependency chains: the user's, or an unlucky sample) sends its
            // iteration to the stand-alone resolution: the host looks at the window's flags once (one synchronisation per window)
            c->cone_big_ok.resize((size_t)W);
            HIPCHK(hipMemcpyAsync(c->cone_big_ok.data(), P.cone_ok, (size_t)W * 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
        } else if (c->persist_sh_big) {   // (the persistent kernel looks at the window's flags itself: no synchronisation)
            launch_cone_big(c, P, W, c->win_lv_pairs, c->win_lv_off, c->stream);
        }
        P.lv_rows = c->win_lv_rows; P.lv_rowinfo = c->win_lv_rowinfo;
        P.lv_pairs = c->win_lv_pairs; P.lv_mi = c->win_lv_mi; P.lv_off = c->win_lv_off;
    }
    if (c->lds_exchange && P.Ng > 1 && !(t >= c->plan_t0 && t < c->plan_t0 + c->plan_w)) {
        const int W = std::min(c->plan_cap, P.T - t + 1);
        hipLaunchKernelGGL(k_exch_plan, dim3(W), dim3(XWG), (c->cone || c->persist) ? std::max(plan_lds_bytes(P.Ng, P.plan_K), plan_cone_bytes()) : plan_lds_bytes(P.Ng, P.plan_K), c->stream, P, t, c->win_plan,
                           c->win_plan_mi, c->win_lv_pairs, c->win_lv_mi, c->win_lv_off, c->win_lv_pairs_p, c->win_lv_offp);
        c->plan_t0 = t; c->plan_w = W;
        P.plan = c->win_plan; P.plan_mi = c->win_plan_mi; P.plan_t0 = t;
        P.lv_pairs = c->win_lv_pairs; P.lv_mi = c->win_lv_mi; P.lv_off = c->win_lv_off;
        P.lv_pairs_p = c->win_lv_pairs_p; P.lv_offp = c->win_lv_offp;
    }
}

template <int KIND, int CT, int TPW = 1>
void launch_chain_iter_ct(Ctx* c, int t, int flags) {
    const KParams& P = c->P;
    const int tiles = (P.N + CT - 1) / CT;
    // an objective without a simulation (banana, user objectives) has work for the tile's control wave only: unless the exchange
    // walk runs inline (all lanes stage its inputs) the tile is launched as that one wave, so that every tile of a large
    // population is resident at once instead of queueing behind 448 idle lanes each
    const bool slim = KIND == 0 && TPW == 1 && !(flags & F_WALK_INLINE) && !c->inline_walk;
    const dim3 grid((tiles + TPW - 1) / TPW), block(slim ? 64 : WG * TPW);
    const double* rin = c->ext_rec_in ? c->ext_rec_in : (const double*)c->rec[c->cur];
    double* rout = c->ext_rec_out ? c->ext_rec_out : c->rec[c->cur ^ 1];
    // (contexts that never walk inline run the kernel compiled without the walk)
    auto kern = c->inline_walk ? k_chain_iter<KIND, CT, TPW, true> : k_chain_iter<KIND, CT, TPW, false>;
    if (c->kev0)   // profiling mode 2: begin/end of this dispatch as the command processor stamps them
        hipExtLaunchKernelGGL(kern, grid, block, tile_smem(c, CT, TPW), c->stream, c->kev0, c->kev1, 0, P, t, rin, rout, flags);
    else
        hipLaunchKernelGGL(kern, grid, block, tile_smem(c, CT, TPW), c->stream, P, t, rin, rout, flags);
}

template <int NP, bool WALK>
void launch_chain_iter_norm_t(Ctx* c, int t, int flags) {
    const KParams& P = c->P;
    const dim3 grid((P.N + NORM_CT - 1) / NORM_CT), block(NORM_WG);
    const double* rin = c->ext_rec_in ? c->ext_rec_in : (const double*)c->rec[c->cur];
    double* rout = c->ext_rec_out ? c->ext_rec_out : c->rec[c->cur ^ 1];
    if (c->kev0)
        hipExtLaunchKernelGGL((k_chain_iter_norm<NP, WALK>), grid, block, norm_smem(c), c->stream, c->kev0, c->kev1, 0, P, t, rin, rout, flags);
    else
        hipLaunchKernelGGL((k_chain_iter_norm<NP, WALK>), grid, block, norm_smem(c), c->stream, P, t, rin, rout, flags);
}
template <int NP>
void launch_chain_iter_norm_narrow_t(Ctx* c, int t, int flags) {
    const KParams& P = c->P;
    const dim3 grid((P.N + NORM_CT - 1) / NORM_CT), block(NORM_WG / 2);
    const double* rin = c->ext_rec_in ? c->ext_rec_in : (const double*)c->rec[c->cur];
    double* rout = c->ext_rec_out ? c->ext_rec_out : c->rec[c->cur ^ 1];
    if (c->kev0)
        hipExtLaunchKernelGGL((k_chain_iter_norm_narrow<NP>), grid, block, norm_smem(c), c->stream, c->kev0, c->kev1, 0, P, t, rin, rout, flags);
    else
        hipLaunchKernelGGL((k_chain_iter_norm_narrow<NP>), grid, block, norm_smem(c), c->stream, P, t, rin, rout, flags);
}
template <int NP>
void launch_chain_iter_norm_wide_t(Ctx* c, int t, int flags) {
    const KParams& P = c->P;
    const dim3 grid((P.N + NORM_CT - 1) / NORM_CT), block(NORM_WG);
    const double* rin = c->ext_rec_in ? c->ext_rec_in : (const double*)c->rec[c->cur];
    double* rout = c->ext_rec_out ? c->ext_rec_out : c->rec[c->cur ^ 1];
    if (c->kev0)
        hipExtLaunchKernelGGL((k_chain_iter_norm_wide<NP>), grid, block, norm_smem(c), c->stream, c->kev0, c->kev1, 0, P, t, rin, rout, flags);
    else
        hipLaunchKernelGGL((k_chain_iter_norm_wide<NP>), grid, block, norm_smem(c), c->stream, P, t, rin, rout, flags);
}
template <int NP>
void launch_chain_iter_norm_any_t(Ctx* c, int t, int flags) {
    const KParams& P = c->P;
    const dim3 grid((P.N + NORM_CT - 1) / NORM_CT), block(NORM_WG);
    const double* rin = c->ext_rec_in ? c->ext_rec_in : (const double*)c->rec[c->cur];
    double* rout = c->ext_rec_out ? c->ext_rec_out : c->rec[c->cur ^ 1];
    if (c->kev0)
        hipExtLaunchKernelGGL((k_chain_iter_norm_any<NP>), grid, block, norm_smem(c), c->stream, c->kev0, c->kev1, 0, P, t, rin, rout, flags);
    else
        hipLaunchKernelGGL((k_chain_iter_norm_any<NP>), grid, block, norm_smem(c), c->stream, P, t, rin, rout, flags);
}
template <int NP>
void launch_chain_iter_norm_narrow_cone_t(Ctx* c, int t, int flags) {
    const KParams& P = c->P;
    const dim3 grid((P.N + NORM_CT - 1) / NORM_CT), block(NORM_WG / 2);
    const double* rin = (const double*)c->rec[c->cur];
    double* rout = c->rec[c->cur ^ 1];
    if (c->kev0)
        hipExtLaunchKernelGGL((k_chain_iter_norm_narrow_cone<NP>), grid, block, norm_smem(c), c->stream, c->kev0, c->kev1, 0, P, t, rin, rout, flags);
    else
        hipLaunchKernelGGL((k_chain_iter_norm_narrow_cone<NP>), grid, block, norm_smem(c), c->stream, P, t, rin, rout, flags);
}
void launch_chain_iter_norm(Ctx* c, int t, int flags) {
    const bool walk = (flags & F_WALK_INLINE) != 0;
    if (walk && c->cone_big) {   // large single shards: every tile walks its own, locally numbered cone (smm_cone_big.hpp)
        switch (c->P.np) {
            case 1: launch_chain_iter_norm_narrow_cone_t<1>(c, t, flags); break;
            case 2: launch_chain_iter_norm_narrow_cone_t<2>(c, t, flags); break;
            case 3: launch_chain_iter_norm_narrow_cone_t<3>(c, t, flags); break;
            default: launch_chain_iter_norm_narrow_cone_t<4>(c, t, flags); break;
        }
        return;
    }
    // the lean walks need a padded plan of at most 31 levels and values without NaN: where that is not given — per-chain or
    // negative thresholds (no padded plan), an injected pair list that goes deeper, an uploaded state with NaN values — the
    // kernel with the walk on 16-byte slots {value, src, partner} runs
    if (walk && (!c->P.lv_pairs_p || c->deep_plan || c->nan_values || c->P.mi_pct)) {   // (mi_pct: the lean plan stands for the persistent launches only)
        switch (c->P.np) {
            case 1: launch_chain_iter_norm_any_t<1>(c, t, flags); break;
            case 2: launch_chain_iter_norm_any_t<2>(c, t, flags); break;
            case 3: launch_chain_iter_norm_any_t<3>(c, t, flags); break;
            default: launch_chain_iter_norm_any_t<4>(c, t, flags); break;
        }
        return;
    }
    if (walk && c->P.lean_wide) {   // one min_improve > 0 for all chains: the lean walk on 16-byte slots
        switch (c->P.np) {
            case 1: launch_chain_iter_norm_wide_t<1>(c, t, flags); break;
            case 2: launch_chain_iter_norm_wide_t<2>(c, t, flags); break;
            case 3: launch_chain_iter_norm_wide_t<3>(c, t, flags); break;
            default: launch_chain_iter_norm_wide_t<4>(c, t, flags); break;
        }
        return;
    }
    if (!walk && c->norm_narrow) {   // more than one round of tiles: two half-size workgroups per CU (k_chain_iter_norm_narrow)
        switch (c->P.np) {
            case 1: launch_chain_iter_norm_narrow_t<1>(c, t, flags); break;
            case 2: launch_chain_iter_norm_narrow_t<2>(c, t, flags); break;
            case 3: launch_chain_iter_norm_narrow_t<3>(c, t, flags); break;
            default: launch_chain_iter_norm_narrow_t<4>(c, t, flags); break;
        }
        return;
    }
    switch (c->P.np * 2 + (walk ? 1 : 0)) {
        case 2: launch_chain_iter_norm_t<1, false>(c, t, flags); break;
        case 3: launch_chain_iter_norm_t<1, true>(c, t, flags); break;
        case 4: launch_chain_iter_norm_t<2, false>(c, t, flags); break;
        case 5: launch_chain_iter_norm_t<2, true>(c, t, flags); break;
        case 6: launch_chain_iter_norm_t<3, false>(c, t, flags); break;
        case 7: launch_chain_iter_norm_t<3, true>(c, t, flags); break;
        case 8: launch_chain_iter_norm_t<4, false>(c, t, flags); break;
        default: launch_chain_iter_norm_t<4, true>(c, t, flags); break;
    }
}

// one thread per evaluation: theta [n][np] -> simM [n][nm], value [n], status [n]
void launch_user_kernel(Ctx* c, const double* theta, int n, double* simM, double* value, int* status) {
    const KParams& P = c->P;
    int np = P.np, nm = P.nm, nud = c->n_objp;
    const double *mom = P.mom, *w = P.w, *ud = P.objp;
    void* args[] = {(void*)&theta, (void*)&n, (void*)&np, (void*)&mom, (void*)&w, (void*)&nm, (void*)&ud, (void*)&nud,
                    (void*)&simM, (void*)&value, (void*)&status};
    if (c->u_lanes > 0)   // map-reduce form: one workgroup of u_lanes threads per evaluation
        HIPCHK(hipModuleLaunchKernel(c->ufn, (unsigned)n, 1, 1, (unsigned)c->u_lanes, 1, 1, 0, c->stream, args, nullptr));
    else
        HIPCHK(hipModuleLaunchKernel(c->ufn, (unsigned)((n + 127) / 128), 1, 1, 128, 1, 1, 0, c->stream, args, nullptr));
}

void launch_chain_iter(Ctx* c, int t, int flags) {
    point_values(c, c->P, t - 1, t);   // the inline walk reads what the accept step of t-1 wrote; this accept step writes the other array
    if (c->ext_vals_out) { c->P.vals_out = c->ext_vals_out; c->P.slot8_out = nullptr; }
    // large single shards (k_exch_resolve_rows): the accept step writes the resolution's initial slots itself
    const bool own_slots = c->rows_exchange && c->P.N == c->P.Ng && !c->ext_rec_out && !c->ext_vals_out && c->obj != SMM_OBJ_USER;
    c->P.slots17_out = own_slots ? c->slots17 : nullptr;
    c->P.nan_flags_out = own_slots ? c->nan_flags + (t & 1) : nullptr;
    if (own_slots) c->slots_iter = t;
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
    if (c->norm_fast) {
        launch_chain_iter_norm(c, t, flags);
    } else if (is_sim(c->obj)) {
        if (c->tpw == 2) launch_chain_iter_ct<1, 8, 2>(c, t, flags);
        else launch_chain_iter_ct<1, 8>(c, t, flags);
    } else if (c->obj == SMM_OBJ_DENSE) {
        launch_chain_iter_ct<2, 16>(c, t, flags);
    } else if (c->gen_keys) {
        launch_chain_iter_ct<0, 16, 2>(c, t, flags);
    } else if (tile_smem_base(c, 64) <= (size_t)60 * 1024 && !c->inline_walk) {
        launch_chain_iter_ct<0, 64>(c, t, flags);
    } else if (c->tpw == 2) {
        launch_chain_iter_ct<0, 8, 2>(c, t, flags);
    } else {
        launch_chain_iter_ct<0, 8>(c, t, flags);
    }
    if (!c->ext_rec_out) c->cur ^= 1;
}

size_t resolve_lean_bytes(int Ng, int K, bool wide) { return std::max(wide ? lean_wide_bytes(Ng, K) : lean_walk_bytes(Ng, K), resolve_lvl_soa_bytes(Ng, K)); }

void launch_resolve_p(Ctx* c, const KParams& P, int t, const double* gathered);
void launch_resolve(Ctx* c, int t, const double* gathered) { launch_resolve_p(c, c->P, t, gathered); }
// (P: the context's parameters, or a copy whose RW is the stride of the value column in `gathered`)
// Which stand-alone kernel resolves exchangeMoves! (AlgoBGP.jl:647-716) — ONE decision, taken once per context (choose_exchange,
// at the end of smm_ctx_create), from the population, the thresholds and dist_fun:
//
//   N_global        min_improve                      dist_fun   kernel                       slots / where
//   <= 8192 (~7400) one value >= 0 for all chains    -          k_exch_resolve_lean          8-byte keys (0) or 16-byte values (> 0), LDS
//   <= 4096         anything else                    any        k_exch_resolve_lvl<1024>     16-byte slots, LDS
//   <= 8192         anything else                    any        k_exch_resolve_lvl_soa       split slots, LDS
//   <= 32768        0 for all chains                 -          k_exch_keys + _rows          4-byte slots (17-bit keys), rows of 1024 pairs, LDS
//   <= 32768        anything else                    -          k_exch_keys + _key           4-byte slots (16-bit keys, intervals), LDS
//   <= 65535        anything else / other dist_fun   any        k_exch_resolve_lvl_big       16-byte slots, global memory
//   above, or K > N_global                           any        k_exch_resolve_any           barrier rounds, global atomics
//
// (the p2p form of a sharded run launches k_exch_resolve_rows itself where this table says _rows — launch_resolve_rows_window:
//  <., true> reads the tagged slots of its window instead of a key pre-pass, <false, true, true> keeps the partners of the rank's
//  own chains only; every other p2p population resolves from the window's plain values through this table.)
// (single shards of objfunc_norm up to 4096 chains do not get here in steady state: their chain kernel walks inline.)  The test
// build can force an entry (SMM_TEST_HOOKS: SMMHIP_*_EXCHANGE, SMMHIP_KEY_WALK) and adds the ticket kernel k_exch_resolve_lds.
enum ExchKernel { XK_LEAN, XK_LVL, XK_LVL_SOA, XK_TICKETS, XK_ROWS, XK_KEY, XK_LVL_BIG, XK_ANY };
ExchKernel choose_exchange(const Ctx* c) {
    if (c->lean_resolve) return XK_LEAN;
    if (c->lvl_exchange) return XK_LVL;
    if (c->lvl_soa_exchange) return XK_LVL_SOA;
    if (c->lds_exchange) return XK_TICKETS;   // (test build only: SMMHIP_DATAFLOW_EXCHANGE)
    if (c->key_exchange) return c->rows_exchange ? XK_ROWS : XK_KEY;
    if (c->big_exchange) return XK_LVL_BIG;
    return XK_ANY;
}
void launch_resolve_p(Ctx* c, const KParams& P_in, int t, const double* gathered) {
    KParams P = P_in;
    point_values(c, P, t, t);   // (single shard: the values the accept step of iteration t wrote)
    switch ((ExchKernel)c->xk) {
    case XK_LEAN:
        if (c->kev0)
            hipExtLaunchKernelGGL(k_exch_resolve_lean, dim3(1), dim3(XWG), resolve_lean_bytes(P.Ng, P.plan_K, P.lean_wide != 0), c->stream, c->kev0, c->kev1, 0, P, t,
                                  gathered);
        else
            hipLaunchKernelGGL(k_exch_resolve_lean, dim3(1), dim3(XWG), resolve_lean_bytes(P.Ng, P.plan_K, P.lean_wide != 0), c->stream, P, t, gathered);
        break;
    case XK_LVL:
#ifdef SMM_TEST_HOOKS
        if (c->lvl_wg == 256) { hipLaunchKernelGGL(k_exch_resolve_lvl<256>, dim3(1), dim3(256), resolve_lvl_bytes(P.Ng, P.plan_K), c->stream, P, t, gathered); break; }
        if (c->lvl_wg == 512) { hipLaunchKernelGGL(k_exch_resolve_lvl<512>, dim3(1), dim3(512), resolve_lvl_bytes(P.Ng, P.plan_K), c->stream, P, t, gathered); break; }
#endif
        if (c->kev0)
            hipExtLaunchKernelGGL(k_exch_resolve_lvl<1024>, dim3(1), dim3(1024), resolve_lvl_bytes(P.Ng, P.plan_K), c->stream, c->kev0, c->kev1, 0, P, t, gathered);
        else
            hipLaunchKernelGGL(k_exch_resolve_lvl<1024>, dim3(1), dim3(1024), resolve_lvl_bytes(P.Ng, P.plan_K), c->stream, P, t, gathered);
        break;
    case XK_LVL_SOA:
        hipLaunchKernelGGL(k_exch_resolve_lvl_soa<1024>, dim3(1), dim3(1024), resolve_lvl_soa_bytes(P.Ng, P.plan_K), c->stream, P, t, gathered);
        break;
    case XK_TICKETS:
#ifdef SMM_TEST_HOOKS
        hipLaunchKernelGGL(k_exch_resolve_lds, dim3(1), dim3(XWG), resolve_lds_bytes(P.Ng), c->stream, P, t, gathered);
#endif
        break;
    case XK_ROWS:
    case XK_KEY: {
        // values and initial slots by the whole chip (gathered records: their value column; single shard: the compact array)
        const double* src = gathered ? gathered : (const double*)P.vals;
        double* vals = gathered ? P.xval : P.vals;
        if (gathered || c->slots_iter != t)   // (a single shard's accept step of iteration t has made the slots already)
            hipLaunchKernelGGL(k_exch_keys, dim3((P.Ng + 255) / 256), dim3(256), 0, c->stream, src, gathered ? P.RW : 1, P.Ng, vals,
                               (uint32_t*)P.xsrc, c->rows_exchange ? c->slots17 : (uint32_t*)nullptr, c->nan_flags, t);
        const bool plds = P.Ng <= XKEY_PARTNER_MAX;   // partners in LDS, else the ballot replay
        const uint32_t* slots16 = (gathered || c->slots_iter != t) ? (const uint32_t*)P.xsrc : (const uint32_t*)nullptr;
        if (c->xk == XK_ROWS && plds)
            hipLaunchKernelGGL(k_exch_resolve_rows<true>, dim3(1), dim3(XWG), resolve_rows_bytes(P.Ng, P.plan_K, P.rows_cap), c->stream, P, t,
                               (const double*)vals, slots16, (const uint32_t*)c->slots17, c->nan_flags);
        else if (c->xk == XK_ROWS)
            hipLaunchKernelGGL(k_exch_resolve_rows<false>, dim3(1), dim3(XWG), resolve_rows_bytes(P.Ng, P.plan_K, P.rows_cap), c->stream, P, t,
                               (const double*)vals, slots16, (const uint32_t*)c->slots17, c->nan_flags);
        else if (plds)
            hipLaunchKernelGGL(k_exch_resolve_key<true>, dim3(1), dim3(XWG), resolve_key_bytes(P.Ng, P.plan_K), c->stream, P, t,
                               (const double*)vals, (const uint32_t*)P.xsrc);
        else
            hipLaunchKernelGGL(k_exch_resolve_key<false>, dim3(1), dim3(XWG), resolve_key_bytes(P.Ng, P.plan_K), c->stream, P, t,
                               (const double*)vals, (const uint32_t*)P.xsrc);
        break;
    }
    case XK_LVL_BIG:
        hipLaunchKernelGGL(k_exch_resolve_lvl_big, dim3(1), dim3(XWG), 0, c->stream, P, t, gathered);
        break;
    case XK_ANY:
        hipLaunchKernelGGL(k_exch_resolve_any, dim3(1), dim3(XWG), 0, c->stream, P, t, gathered);
        break;
    }
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
    if (c->unresolved && c->P.N < c->P.Ng) throw std::string("the exchange of the last iteration needs every shard's records: call smm_bgp_p2p_finish first");
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

void persist_repair(Ctx* c, int n_replay = -1);
void p2p_enqueue(Ctx* c, int n_iters);
// The ranks of a sharded run agree on what their launches of the persistent form ended with: every rank writes its error word into
// every rank's window — {word, number of its last launch} — and reads its own window until all ranks' words of that launch are there
// (the host side of a rendezvous every rank reaches: smm_sync / smm_bgp_p2p_finish behind the same steps).  The smallest word wins,
// as it does among the chains of one device: the first failing iteration, the first failing chain of the population.
unsigned long long p2p_agree_error(Ctx* c, unsigned long long e_local) {
    const KParams& P = c->P;
    const int G = P.p2p_G;
    const PrWin WL = pr_win_layout(P.Ng, P.RW, G, (P.N + NORM_CT - 1) / NORM_CT);
    const uint32_t seq = c->pr_epoch;
    for (int r = 0; r < G; ++r)
        HIPCHK(hipMemcpy(P.p2p_win[r] + c->prw_off + WL.fin + 128 * (size_t)P.p2p_rank, &e_local, 8, hipMemcpyHostToDevice));
    for (int r = 0; r < G; ++r)   // (the word is there before the number that says so)
        HIPCHK(hipMemcpy(P.p2p_win[r] + c->prw_off + WL.fin + 128 * (size_t)P.p2p_rank + 8, &seq, 4, hipMemcpyHostToDevice));
    std::vector<unsigned char> buf((size_t)128 * G);
    const auto t0 = std::chrono::steady_clock::now();
    unsigned long long e = e_local;
    for (;;) {
        HIPCHK(hipMemcpy(buf.data(), c->p2p_mine + c->prw_off + WL.fin, buf.size(), hipMemcpyDeviceToHost));
        bool all = true;
        e = e_local;
        for (int r = 0; r < G; ++r) {
            uint32_t s_r; unsigned long long e_r;
            memcpy(&e_r, buf.data() + 128 * (size_t)r, 8); memcpy(&s_r, buf.data() + 128 * (size_t)r + 8, 4);
            if (s_r != seq) { all = false; break; }
            e = std::min(e, e_r);
        }
        if (all) {
            // (word and number arrive by two copies and are read by one unfenced DMA: a number seen with the word of the launch before is
            // possible in theory — ADVICE r5 —: every number is there now, so one more look holds every word that was written before its number)
            HIPCHK(hipMemcpy(buf.data(), c->p2p_mine + c->prw_off + WL.fin, buf.size(), hipMemcpyDeviceToHost));
            e = e_local;
            for (int r = 0; r < G; ++r) { unsigned long long e_r; memcpy(&e_r, buf.data() + 128 * (size_t)r, 8); e = std::min(e, e_r); }
            return e;
        }
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30)) throw std::string("sharded run: a rank did not report the end of its step within 30 s (is every rank calling smm_sync / smm_bgp_p2p_finish?)");
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}
// the second launch of the persistent form that gave up (its tiles not resident together: a masked or shared device; a cone that did not fit;
// a state the form cannot walk): the form is off for this context from here on — said ONCE on stderr, since nothing else changes for the
// caller but the speed (smm_get_persistent / smm_describe report it too)
void persist_give_up(Ctx* c) {
    if (!c->persist_broken)
        fprintf(stderr, "libsmmhip: context %p (device %d, chains %d of %d): the persistent form gave up twice (iteration %d) and is OFF for this context; "
                        "the per-iteration kernels run instead (same results, more launches). smm_get_persistent reports it.\n",
                (void*)c, c->device, c->P.N, c->P.Ng, c->iter);
    c->persist_broken = true;
}
// (an entry point hands the sticky failure to its caller)
int told(Ctx* c) { c->failed_told = true; return c->failed; }
int told(Ctx* c, int rc) { if (rc != SMM_OK && rc == c->failed) c->failed_told = true; return rc; }
int check_device_error(Ctx* c) {
    if (c->failed) return c->failed;   // err holds the message of the first failure
    unsigned long long e = ERR_NONE;
    HIPCHK(hipMemcpy(&e, c->P.err, sizeof e, hipMemcpyDeviceToHost));
    if (c->persist_sh && c->snap_valid && !c->in_repair && c->p2p_mine) {
        // launches of the persistent form ran on every rank: what one of them ended with concerns all (its tiles did not stop either)
        const unsigned long long eg = p2p_agree_error(c, e);
        if (eg != ERR_NONE) {
            const int kind = (int)(eg & 3), it = (int)(eg >> 34);
            if (kind == ERRK_FORM && ++c->persist_strikes >= 2) persist_give_up(c);
            // a hard error (AlgoBGP.jl:341,409): every rank replays up to and including the failing iteration — which completes for all
            // chains, as everywhere — and stands there; a time-out or a cone that did not fit: the whole step again, on the other forms
            // (a hard error raised BEFORE the first of these launches — by a one-iteration launch ahead of them on the stream —: they saw the word at
            // their entry and stored nothing; n = 0 puts the host's bookkeeping back to the snapshot and replays nothing)
            persist_repair(c, (kind == ERRK_NO_DRAW || kind == ERRK_NEGATIVE) ? std::max(0, it - c->snap_iter) : -1);
            HIPCHK(hipMemcpy(&e, c->P.err, sizeof e, hipMemcpyDeviceToHost));
            if (kind == ERRK_NO_DRAW || kind == ERRK_NEGATIVE) e = std::min(e, eg);   // (the first failing chain of the POPULATION — maybe another rank's: every rank reports the same)
        }
        c->snap_valid = false;
        if (eg == ERR_NONE) c->persist_proven = true;
    } else if (e != ERR_NONE && c->snap_valid && !c->in_repair) {
        // launches of the persistent kernel ran since the last check: their tiles do not stop at the failing iteration.  Back to the
        // state before the first of them, and the same iterations again on the one-launch-per-iteration path, which does.
        // (a tile gave up waiting, or a cone did not fit.  Once may be somebody else's doing — another context or process held compute
        // units while the tiles wanted to be resident together —: the form is tried again; the second time it is off for the context)
        if ((e & 3) != 3 && (int)(e >> 34) <= c->snap_iter) {
            // ... unless the word was raised BEFORE the first of them, by a one-iteration launch ahead of them on the stream that nobody had
            // looked at yet: they saw the word at their entry and stored nothing.  Nothing to replay — rolling back, clearing the word and
            // running on LOST the error (tools/fuzz_errors.py, 2 of 600 cases) —: the host's bookkeeping back to the snapshot, the word stays
            const unsigned long long keep = e;
            persist_repair(c, 0);
            HIPCHK(hipMemcpy(c->P.err, &keep, sizeof keep, hipMemcpyHostToDevice));
            e = keep;
        } else {
            if ((e & 3) == 3 && ++c->persist_strikes >= 2) persist_give_up(c);
            persist_repair(c);
            HIPCHK(hipMemcpy(&e, c->P.err, sizeof e, hipMemcpyDeviceToHost));
        }
    }
    if (!c->in_repair) {
        if (e == ERR_NONE && c->snap_valid) c->persist_proven = true;   // launches of the persistent form came through: its tiles ARE resident together
        c->snap_valid = false;
    }
    if (e == ERR_NONE) return SMM_OK;
    const int kind = (int)(e & 3), chain = (int)((e >> 2) & 0xffffffffu), it = (int)(e >> 34);
    char b[256];
    int rc;
    if (kind == ERRK_FORM) {
        snprintf(b, sizeof b, "internal error: the exchange of iteration %d could not be resolved in the form chosen for it (chain %d)", it, chain + 1);
        rc = SMM_ERR_HIP;
    } else if (kind == ERRK_CAPACITY) {
        snprintf(b, sizeof b, "values form of the sharded exchange: more than %d records between one pair of ranks (chain %d, iteration %d): "
                 "use the record all-gather (smm_bgp_exchange_dev / smm_bgp_sharded_step)", c->a2a_cap, chain + 1, it);
        rc = SMM_ERR_EXCHANGE_CAPACITY;
    } else if (kind == ERRK_NEGATIVE) {
        snprintf(b, sizeof b, "AlgoBGP assumes that your objective function returns a non-negative number "
                 "(chain %d, iteration %d)", chain + 1, it);
        rc = SMM_ERR_NEGATIVE_OBJECTIVE;
    } else {
        snprintf(b, sizeof b, "no draw in support after %d trials (chain %d, iteration %d): increase smpl_iters",
                 c->P.user_n ? std::min(c->P.rb_tries, c->P.smpl_iters) : c->P.smpl_iters, chain + 1, it);
        rc = SMM_ERR_NO_DRAW_IN_SUPPORT;
    }
    c->err = b;
    // The reference aborts inside the failing iteration (AlgoBGP.jl:341,409).  Here that iteration completes for all chains
    // and every later launch of the step sees the error word and stores nothing: the run stands at the failing iteration,
    // its exchange is never applied, and the context refuses to go on until smm_set_state.
    if (kind != ERRK_FORM && it >= 1 && it <= c->iter) {   // (kind 0, a block of the values form overflowed: the exchange of iteration `it` was not applied either)
        c->iter = it;
        c->pending = false; c->prev_open = false; c->unresolved = false; c->pending_ext = false; c->rec_external = false;
        c->a2a_open = false; c->p2p_current = false;
    }
    c->failed = rc;
    return rc;
}

int fail(Ctx* c, int code, const std::string& m) {
    if (c) c->err = m;
    else g_create_err = m;
    return code;
}

// temporary device buffer of one call (freed on every exit path)
template <class T>
struct DevBuf {
    T* p = nullptr;
    explicit DevBuf(size_t n) { HIPCHK(hipMalloc((void**)&p, (n ? n : 1) * sizeof(T))); }
    ~DevBuf() { if (p) (void)hipFree(p); }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
};


// ---- the p2p form of the sharded iteration (smm_p2p.hpp, include/smmhip.h) ----
template <int NP, bool BIG>
void launch_chain_iter_norm_p2p_tb(Ctx* c, const KParams& P, int t, int flags, size_t smem) {
    const dim3 grid((P.N + NORM_CT - 1) / NORM_CT), block(NORM_WG);
    if (c->kev0)
        hipExtLaunchKernelGGL((k_chain_iter_norm_p2p<NP, BIG>), grid, block, smem, c->stream, c->kev0, c->kev1, 0, P, t, (const double*)nullptr, (double*)nullptr, flags);
    else
        hipLaunchKernelGGL((k_chain_iter_norm_p2p<NP, BIG>), grid, block, smem, c->stream, P, t, (const double*)nullptr, (double*)nullptr, flags);
}
template <int NP>
void launch_chain_iter_norm_p2p_t(Ctx* c, const KParams& P, int t, int flags, size_t smem) {
    // (BIG: staging loops for up to 8192 chains — two shards of 4096; the small form serves populations up to 4096)
    if (P.Ng > XLVL_MAX || P.plan_K > XLVL_MAX) launch_chain_iter_norm_p2p_tb<NP, true>(c, P, t, flags, smem);
    else launch_chain_iter_norm_p2p_tb<NP, false>(c, P, t, flags, smem);
}
size_t p2p_walk_bytes(const Ctx* c) { return (lean_walk_bytes(c->P.Ng, c->P.plan_K) + 15) & ~(size_t)15; }
template <int NP>
void launch_chain_iter_norm_p2p_rows_t(Ctx* c, const KParams& P, int t, int flags, size_t smem) {
    const dim3 grid((P.N + NORM_CT - 1) / NORM_CT), block(c->norm_narrow ? NORM_WG / 2 : NORM_WG);
    if (c->norm_narrow) {   // a shard of more than one round of tiles
        if (c->kev0)
            hipExtLaunchKernelGGL((k_chain_iter_norm_p2p_rows_narrow<NP>), grid, block, smem, c->stream, c->kev0, c->kev1, 0, P, t, (const double*)nullptr, (double*)nullptr, flags);
        else
            hipLaunchKernelGGL((k_chain_iter_norm_p2p_rows_narrow<NP>), grid, block, smem, c->stream, P, t, (const double*)nullptr, (double*)nullptr, flags);
    } else if (c->kev0)
        hipExtLaunchKernelGGL((k_chain_iter_norm_p2p_rows<NP>), grid, block, smem, c->stream, c->kev0, c->kev1, 0, P, t, (const double*)nullptr, (double*)nullptr, flags);
    else
        hipLaunchKernelGGL((k_chain_iter_norm_p2p_rows<NP>), grid, block, smem, c->stream, P, t, (const double*)nullptr, (double*)nullptr, flags);
}
void launch_chain_iter_norm_p2p(Ctx* c, int t, int flags) {
    KParams P = c->P;
    point_values(c, P, t - 1, t);
    const bool walk = (flags & F_WALK_INLINE) != 0;
    P.tile_off = walk ? (int)(p2p_walk_bytes(c) / sizeof(double)) : 0;
    const size_t smem = (size_t)P.tile_off * sizeof(double) + norm_tile_doubles(P.np) * sizeof(double);
    if (c->p2p_rows) {   // the kernel without a walk; its accept step stores the 4-byte slots of k_exch_resolve_rows<., true>
        switch (P.np) {
            case 1: launch_chain_iter_norm_p2p_rows_t<1>(c, P, t, flags, smem); break;
            case 2: launch_chain_iter_norm_p2p_rows_t<2>(c, P, t, flags, smem); break;
            case 3: launch_chain_iter_norm_p2p_rows_t<3>(c, P, t, flags, smem); break;
            default: launch_chain_iter_norm_p2p_rows_t<4>(c, P, t, flags, smem); break;
        }
        return;
    }
    switch (P.np) {
        case 1: launch_chain_iter_norm_p2p_t<1>(c, P, t, flags, smem); break;
        case 2: launch_chain_iter_norm_p2p_t<2>(c, P, t, flags, smem); break;
        case 3: launch_chain_iter_norm_p2p_t<3>(c, P, t, flags, smem); break;
        default: launch_chain_iter_norm_p2p_t<4>(c, P, t, flags, smem); break;
    }
}
// this rank's slice after iteration t -> every rank's window (parity t & 1); rec_src: out of the context's own record array (a
// publication), else out of its own window (generic form, after a chain kernel); ll: the self-validating form of the inline kernels
void launch_p2p_push(Ctx* c, int t, const double* rec_src, bool ll) {
    KParams P = c->P;
    const dim3 grid(p2p_units(P.N)), block(256);
    if (ll) hipLaunchKernelGGL((k_p2p_push<true, true>), grid, block, 0, c->stream, P, t, rec_src);
    else if (rec_src) hipLaunchKernelGGL((k_p2p_push<true, false>), grid, block, 0, c->stream, P, t, rec_src);
    else hipLaunchKernelGGL((k_p2p_push<false, false>), grid, block, 0, c->stream, P, t, (const double*)nullptr);
    if (!ll) { c->p2p_seq += 1; c->p2p_unwaited = true; }
}
void launch_p2p_wait(Ctx* c) {
    KParams P = c->P;
    P.p2p_want = (unsigned long long)p2p_units(P.N) * c->p2p_seq;
    hipLaunchKernelGGL(k_p2p_wait, dim3(1), dim3(64), 0, c->stream, P, c->iter);
    c->p2p_unwaited = false;
}
// inline form: everybody's records and values after iteration t, out of their self-validating form into this rank's plain arrays
void launch_p2p_unpack(Ctx* c, int t) {
    hipLaunchKernelGGL(k_p2p_unpack, dim3((c->P.Ng + 255) / 256), dim3(256), 0, c->stream, c->P, t);
}
// exchangeMoves! of iteration t from the values in this rank's window (complete: somebody has waited for the arrivals)
void launch_resolve_window(Ctx* c, int t) {
    const P2PLayout L = p2p_layout(c->P.Ng, c->P.RW);
    KParams P1 = c->P;
    P1.RW = 1;   // (the resolve kernels read value s at gathered[s * RW])
    launch_resolve_p(c, P1, t, (const double*)(c->p2p_mine + L.val[t & 1]));
}
// the same straight from the tagged slots and self-validating values of the window (p2p form of large norm populations): no wait,
// no unpacking, no key pre-pass in front of k_exch_resolve_rows
void launch_resolve_rows_window(Ctx* c, int t) {
    const KParams& P = c->P;
    // a shard of a population past XKEY_PARTNER_MAX: partners of its own chains only, written by the swaps (no second pass over the rows)
    const bool own = P.Ng > XKEY_PARTNER_MAX && P.N < P.Ng && resolve_rows_bytes(P.Ng, P.plan_K, P.rows_cap, P.N) <= (size_t)158 * 1024;
    if (own) {
        const size_t sm = resolve_rows_bytes(P.Ng, P.plan_K, P.rows_cap, P.N);
        if (c->kev0) hipExtLaunchKernelGGL((k_exch_resolve_rows<false, true, true>), dim3(1), dim3(XWG), sm, c->stream, c->kev0, c->kev1, 0, P, t, (const double*)nullptr, (const uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr);
        else hipLaunchKernelGGL((k_exch_resolve_rows<false, true, true>), dim3(1), dim3(XWG), sm, c->stream, P, t, (const double*)nullptr, (const uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr);
        return;
    }
    const size_t smem = resolve_rows_bytes(P.Ng, P.plan_K, P.rows_cap);
    if (P.Ng <= XKEY_PARTNER_MAX) {
        if (c->kev0) hipExtLaunchKernelGGL((k_exch_resolve_rows<true, true>), dim3(1), dim3(XWG), smem, c->stream, c->kev0, c->kev1, 0, P, t, (const double*)nullptr, (const uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr);
        else hipLaunchKernelGGL((k_exch_resolve_rows<true, true>), dim3(1), dim3(XWG), smem, c->stream, P, t, (const double*)nullptr, (const uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr);
    } else {
        if (c->kev0) hipExtLaunchKernelGGL((k_exch_resolve_rows<false, true>), dim3(1), dim3(XWG), smem, c->stream, c->kev0, c->kev1, 0, P, t, (const double*)nullptr, (const uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr);
        else hipLaunchKernelGGL((k_exch_resolve_rows<false, true>), dim3(1), dim3(XWG), smem, c->stream, P, t, (const double*)nullptr, (const uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr);
    }
}
// ---- the persistent chain kernels (smm_chain_persist.hpp, smm_chain_persist_loc.hpp) ----
const void* persist_loc_fn(int np, bool wide, bool sh, bool pct = false) {
    if (pct) return np == 1 ? (const void*)k_chain_persist_loc<1, true, false, true> : (const void*)k_chain_persist_loc<2, true, false, true>;
    if (np == 1) return wide ? (sh ? (const void*)k_chain_persist_loc<1, true, true> : (const void*)k_chain_persist_loc<1, true, false>)
                             : (sh ? (const void*)k_chain_persist_loc<1, false, true> : (const void*)k_chain_persist_loc<1, false, false>);
    return wide ? (sh ? (const void*)k_chain_persist_loc<2, true, true> : (const void*)k_chain_persist_loc<2, true, false>)
                : (sh ? (const void*)k_chain_persist_loc<2, false, true> : (const void*)k_chain_persist_loc<2, false, false>);
}
// can the iterations from c->iter + 1 on run as one launch of it?  At least two (a single iteration is the ordinary kernel's), behind
// an iteration some chain kernel has completed (the launch continues from the plain state blocks: no first iteration, no uploaded
// state, no exchange applied by the three-phase calls), and an exchange left to "the next chain kernel" must have its plan in the
// current window together with this iteration's.
bool persist_usable(const Ctx* c, int n_left) {
    if (!(c->persist && c->persist_on && !c->persist_broken && !c->in_repair && !c->nan_values && n_left >= 2 && !c->ext_rec_in && !c->ext_rec_out &&
          !c->ext_vals_out && !c->rec_external && c->P.N == c->P.Ng))
        return false;
    if (c->iter < 1 || !c->prev_open || c->exch_done || (c->pending && !c->unresolved)) return false;
    const int t0 = c->iter + 1;
    // (the locally numbered form starts a new plan window at the pending exchange's iteration instead: launch_chain_persist)
    if (!c->persist_loc && !c->persist_tile && c->unresolved && !(t0 - 1 >= c->plan_t0 && t0 + 1 < c->plan_t0 + c->plan_w)) return false;
    return true;
}
// ... as a shard (smm_bgp_p2p_step): from the p2p state — the records after iteration `iter` in the windows, its exchange not resolved
// yet — or behind a launch of its own; the first iteration of a run, the iteration behind a settled state are the per-iteration forms'
bool persist_sh_usable(const Ctx* c, int n_left) {
    // (NOT c->nan_values: that flag is this shard's own — smm_set_state saw a NaN among ITS values — and the form must be chosen from what every
    // rank knows, or one rank would take the per-iteration kernels while its peers wait at the launches' start barrier.  The launch reports such a
    // state itself — kind 3 at its first iteration, smm_chain_persist_loc.hpp —, the ranks agree on the word and replay the step on the other forms)
    if (!(c->persist_sh && c->persist_on && !c->persist_broken && !c->in_repair && n_left >= 2 && c->p2p_mine)) return false;
    if (c->p2p_ranks_here * ((c->P.N + NORM_CT - 1) / NORM_CT) > c->persist_max_tiles) return false;   // (ranks sharing this device: not resident together)
    if (c->iter < 1 || !c->prev_open || c->exch_done || c->a2a_open) return false;
    if (c->p2p_current) return c->rec_external && (c->pending_ext || !exchange_active(c, c->iter));
    return !c->rec_external && (c->unresolved || !c->pending);
}
// the state a failed launch of the persistent kernel is rolled back to (device copies on the stream, before anything of the step runs)
void persist_snapshot(Ctx* c) {
    const KParams& P = c->P;
    const size_t N = P.N;
    HIPCHK(hipMemcpyAsync(c->snap_cs, P.cs, N * CSW * 8, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->snap_rec, c->rec[c->cur], N * P.RW * 8, hipMemcpyDeviceToDevice, c->stream));
    for (int b = 0; b < 2; ++b) {
        HIPCHK(hipMemcpyAsync(c->snap_vals[b], c->vals_buf[b], (N + 4) * 8, hipMemcpyDeviceToDevice, c->stream));
        if (c->slot8_buf[b]) HIPCHK(hipMemcpyAsync(c->snap_slot8[b], c->slot8_buf[b], (N + 4 + 128) * 8, hipMemcpyDeviceToDevice, c->stream));
    }
    HIPCHK(hipMemcpyAsync(c->snap_xres, P.xres, (size_t)P.Ng * 8, hipMemcpyDeviceToDevice, c->stream));
    c->snap_iter = c->iter; c->snap_cur = c->cur; c->snap_slots_iter = c->slots_iter;
    c->snap_pending = c->pending; c->snap_prev_open = c->prev_open; c->snap_unresolved = c->unresolved; c->snap_exch_done = c->exch_done;
    c->snap_valid = true;
}
// iterations c->iter + 1 .. as ONE launch, as far as the look-ahead windows reach; returns how many it covers (0: not this time)
int launch_chain_persist(Ctx* c, int n_left) {
    const int t0 = c->iter + 1;
    // (k_chain_persist_norm and _gen draw in the kernel unless tables are injected)
    // (k_chain_persist_tile too, since round 5: its 512 lanes draw the next iteration's randomness behind the publication, where the tile waits for
    // its peers' stores anyway — k_pregen_rng was 2.2 us per iteration of C5 on the main stream)
    const bool pregen = (c->persist_gen || c->persist_tile) ? (c->P.user_ntab || c->P.user_utab) : !(c->norm_fast && !c->P.user_ntab && !c->P.user_utab);
    if ((c->persist_loc || c->persist_tile) && c->unresolved && !(t0 - 1 >= c->plan_t0 && t0 + 1 < c->plan_t0 + c->plan_w)) {
        // the pending exchange's plan is not in a window that also reaches past this iteration: a new window from ITS iteration on (one
        // iteration planned twice per window; no per-iteration launch, no stand-alone resolution at the windows' ends)
        c->plan_w = 0;
        ensure_windows(c, t0 - 1, pregen);
    } else if (!c->unresolved) ensure_windows(c, t0, pregen);   // (with an exchange pending the window holds its plan and this iteration's: persist_usable)
    if (pregen && !(t0 >= c->rng_t0 && t0 < c->rng_t0 + c->rng_w)) ensure_windows(c, t0);
    int t1 = std::min(c->iter + n_left, c->plan_t0 + c->plan_w - 1);
    if (pregen) t1 = std::min(t1, c->rng_t0 + c->rng_w - 1);
    t1 = std::min(t1, t0 + PR_MAX_ITERS - 1);
    if (t1 - t0 + 1 < 2) return 0;
    if (!c->snap_valid) persist_snapshot(c);
    ++c->pr_epoch;
    if ((c->pr_epoch & 0x7fu) == 0u) {   // the slot tags' epoch bits start over: nothing older may look current
        if (c->persist_loc || c->persist_tile) {   // (a shard zeroes its own window's ring: its peers store into it only behind the launch's start barrier)
            const PrWin WL = pr_win_layout(c->P.Ng, c->P.RW, c->persist_sh ? c->P.p2p_G : 1, (c->P.N + NORM_CT - 1) / NORM_CT);   // (PT_CT == NORM_CT)
            unsigned char* base = c->persist_sh ? c->p2p_mine + c->prw_off : c->prw;
            HIPCHK(hipMemsetAsync(base + WL.slot, 0, WL.total - WL.slot, c->stream));
        } else {
            HIPCHK(hipMemsetAsync(c->P.pr_slot, 0, persist_ring_slot_bytes(c->P.Ng), c->stream));
            HIPCHK(hipMemsetAsync(c->P.pr_rec, 0, persist_ring_rec_bytes(c->P.Ng, c->P.RW), c->stream));
        }
    }
    KParams& P = c->P;
    P.pr_epoch = c->pr_epoch;
    point_values(c, P, t0 - 1, t1);   // (nothing is read from the value arrays: the last iteration writes them)
    // (a spin of the form may last 4 s — a peer is gone, not late — once a launch of this context has come through; until then a
    // tenth of that: tiles that are not resident together, a masked or partitioned device, must not look like a hang)
    // (a shard waits for its PEERS' launches at the start barrier: processes that start a second apart are late, not gone — 4 s from the start)
    const unsigned long long tmo = (c->persist_proven || c->persist_sh) ? P2P_TIMEOUT_TICKS : PERSIST_TMO_FIRST;
    if (c->persist_loc) {
        PersistLocArgs A{};
        const int G = c->persist_sh ? P.p2p_G : 1;
        const int tiles = (P.N + NORM_CT - 1) / NORM_CT;
        const PrWin WL = pr_win_layout(P.Ng, P.RW, G, tiles);
        A.cone_hdr = P.cone_hdr; A.cone_pairs = P.cone_pairs; A.cone_gather = P.cone_gather; A.cone_ok = P.cone_ok;
        for (int r = 0; r < P2P_MAXG; ++r) A.win[r] = nullptr;
        if (c->persist_sh) {
            for (int r = 0; r < G; ++r) A.win[r] = P.p2p_win[r] + c->prw_off;
            A.self = c->p2p_mine + c->prw_off;
        } else { A.win[0] = c->prw; A.self = c->prw; }
        A.o_ctl = WL.ctl; A.o_arrive = WL.arrive; A.o_progress = WL.progress; A.o_slot = WL.slot; A.o_rec = WL.rec;
        A.cs = P.cs; A.rec_in = c->rec[c->cur]; A.rec_out = c->rec[c->cur ^ 1]; A.vals_out = P.vals_out; A.slot8_out = P.slot8_out; A.walk_flags = P.walk_flags;
        A.hrec = P.hrec; A.err = P.err; A.ts = P.ts;
        A.Z = P.Z; A.lb = P.lb; A.ub = P.ub; A.mom = P.mom; A.w = P.w; A.objp = P.objp;
        A.rb = pregen ? P.rb : nullptr;
        A.N = P.N; A.Ng = P.Ng; A.offset = P.offset; A.G = G; A.rank = c->persist_sh ? P.p2p_rank : 0; A.ns = P.ns; A.zstride = P.zstride; A.plan_t0 = P.plan_t0;
        A.exch_from = c->exchange_from;
        A.sigma_update_steps = P.sigma_update_steps; A.smpl_iters = P.smpl_iters; A.t0 = t0; A.t1 = t1;
        A.rb_t0 = P.rb_t0; A.RBW = P.RBW; A.rb_tries = P.rb_tries; A.user_n = P.user_n;
        A.failbox = (P.obj == SMM_OBJ_NORM_FAILBOX && P.objp) ? 1 : 0;
        A.walk_first = c->unresolved ? 1 : 0;
        A.ring_k = c->pr_ring_k; A.slow_tile = c->pr_slow_tile; A.slow_ticks = c->pr_slow_ticks;
        A.tables_local = c->persist_sh_big ? 1 : 0; A.unit_sh = P.lean_unit == 16 ? 4 : (P.lean_unit == 8 ? 3 : 2);
        A.epoch = c->pr_epoch; A.sigma_adjust_by = P.sigma_adjust_by; A.thr = P.mi_value; A.seed = P.seed; A.tmo = tmo; A.mi_g = P.min_improve_g;
        const dim3 grid(tiles), block(NORM_WG);
        const size_t smem = persist_loc_smem_bytes(P.np);
        auto go = [&](auto kern) {
            if (c->kev0) hipExtLaunchKernelGGL(kern, grid, block, smem, c->stream, c->kev0, c->kev1, 0, A);
            else hipLaunchKernelGGL(kern, grid, block, smem, c->stream, A);
        };
        const bool wd = c->persist_wide, sh = c->persist_sh;
        if (P.mi_pct) { if (P.np == 1) go(k_chain_persist_loc<1, true, false, true>); else go(k_chain_persist_loc<2, true, false, true>); }
        else if (P.np == 1) {
            if (wd) { if (sh) go(k_chain_persist_loc<1, true, true>); else go(k_chain_persist_loc<1, true, false>); }
            else { if (sh) go(k_chain_persist_loc<1, false, true>); else go(k_chain_persist_loc<1, false, false>); }
        } else {
            if (wd) { if (sh) go(k_chain_persist_loc<2, true, true>); else go(k_chain_persist_loc<2, true, false>); }
            else { if (sh) go(k_chain_persist_loc<2, false, true>); else go(k_chain_persist_loc<2, false, false>); }
        }
    } else if (c->persist_tile) {
        PersistTileArgs A{};
        const int tiles = (P.N + PT_CT - 1) / PT_CT;
        const PrWin WL = pr_win_layout(P.Ng, P.RW, 1, tiles);
        const int kind = obj_kind(c->obj);
        A.cone_hdr = P.cone_hdr; A.cone_pairs = P.cone_pairs; A.cone_gather = P.cone_gather; A.cone_ok = P.cone_ok;
        A.self = c->prw; A.o_ctl = WL.ctl; A.o_progress = WL.progress; A.o_rec = WL.rec;
        A.cs = P.cs; A.rec_in = c->rec[c->cur]; A.rec_out = c->rec[c->cur ^ 1]; A.vals_out = P.vals_out; A.slot8_out = P.slot8_out; A.walk_flags = P.walk_flags;
        A.hrec = P.hrec; A.err = P.err; A.ts = P.ts;
        A.Z = P.Z; A.lb = P.lb; A.ub = P.ub; A.mom = P.mom; A.w = P.w; A.objp = P.objp; A.dense_Bf = P.dense_Bf; A.dense_Af = P.dense_Af; A.dense_A2f = P.dense_A2f;
        A.rb = pregen ? P.rb : nullptr;
        A.N = P.N; A.Ng = P.Ng; A.np = P.np; A.nm = P.nm; A.ns = P.ns; A.zstride = P.zstride; A.RW = P.RW; A.HW = P.HW; A.RBW = P.RBW; A.dense_nOt = P.dense_nOt;
        A.batch_size = P.batch_size; A.failbox = (P.obj == SMM_OBJ_NORM_FAILBOX && P.objp) ? 1 : 0;
        A.plan_t0 = P.plan_t0; A.exch_from = c->exchange_from; A.sigma_update_steps = P.sigma_update_steps; A.smpl_iters = P.smpl_iters; A.t0 = t0; A.t1 = t1;
        A.rb_t0 = P.rb_t0; A.rb_tries = P.rb_tries; A.user_n = P.user_n;
        A.ring_k = c->pr_ring_k; A.slow_tile = c->pr_slow_tile; A.slow_ticks = c->pr_slow_ticks; A.walk_first = c->unresolved ? 1 : 0;
        A.unit_sh = P.lean_unit == 16 ? 4 : (P.lean_unit == 8 ? 3 : 2); A.scout_after = P.scout_after; A.scout_gl = P.scout_gl;
        A.epoch = c->pr_epoch; A.sigma_adjust_by = P.sigma_adjust_by; A.thr = P.mi_value; A.seed = P.seed; A.tmo = tmo; A.mi_g = P.min_improve_g;
        const dim3 grid(tiles), block(WG);
        A.u_lanes = c->u_lanes; A.n_udata = c->n_objp;
        const size_t smem = persist_tile_smem(c);
        auto go = [&](auto kern) {
            if (c->kev0) hipExtLaunchKernelGGL(kern, grid, block, smem, c->stream, c->kev0, c->kev1, 0, A);
            else hipLaunchKernelGGL(kern, grid, block, smem, c->stream, A);
        };
        if (c->utfn) {   // the same kernel, compiled with the user's map-reduce objective inside (user_tile_compile)
            void* args[] = {(void*)&A};
            if (c->kev0) HIPCHK(hipExtModuleLaunchKernel(c->utfn, grid.x * (unsigned)WG, 1, 1, WG, 1, 1, smem, c->stream, args, nullptr, c->kev0, c->kev1, 0));
            else HIPCHK(hipModuleLaunchKernel(c->utfn, grid.x, 1, 1, WG, 1, 1, (unsigned)smem, c->stream, args, nullptr));
        } else if (P.mi_pct) { if (kind == 2) go(k_chain_persist_tile<2, true>); else go(k_chain_persist_tile<1, true>); }
        else if (kind == 2) go(k_chain_persist_tile<2>); else go(k_chain_persist_tile<1>);
    } else if (c->persist_gen) {
        PersistGenArgs A{};
        A.cone_hdr = P.cone_hdr; A.cone_pairs = P.cone_pairs; A.cone_gather = P.cone_gather; A.cone_ok = P.cone_ok;
        A.pr_slot = P.pr_slot; A.pr_rec = P.pr_rec; A.pr_progress = P.pr_progress; A.pr_ctl = P.pr_ctl;
        A.cs = P.cs; A.rec_in = c->rec[c->cur]; A.rec_out = c->rec[c->cur ^ 1]; A.vals_out = P.vals_out; A.slot8_out = P.slot8_out; A.walk_flags = P.walk_flags;
        A.hrec = P.hrec; A.err = P.err; A.ts = P.ts;
        A.lb = P.lb; A.ub = P.ub; A.mom = P.mom; A.w = P.w;
        A.rb = pregen ? P.rb : nullptr;
        A.N = P.N; A.Ng = P.Ng; A.np = P.np; A.nm = P.nm; A.RW = P.RW; A.HW = P.HW; A.plan_t0 = P.plan_t0; A.exch_from = c->exchange_from;
        A.sigma_update_steps = P.sigma_update_steps; A.smpl_iters = P.smpl_iters; A.t0 = t0; A.t1 = t1;
        A.rb_t0 = P.rb_t0; A.RBW = P.RBW; A.rb_tries = P.rb_tries; A.user_n = P.user_n;
        A.walk_first = c->unresolved ? 1 : 0;
        A.ring_k = c->pr_ring_k; A.slow_tile = c->pr_slow_tile; A.slow_ticks = c->pr_slow_ticks;
        A.epoch = c->pr_epoch; A.sigma_adjust_by = P.sigma_adjust_by; A.seed = P.seed; A.tmo = tmo;
        A.udata = P.objp; A.n_udata = c->n_objp;
        const dim3 grid(P.N / PG_CT), block(1024);
        const size_t smem = persist_gen_smem_bytes(P.Ng, P.np, P.RW, P.HW) + (c->persist_user ? persist_gen_user_bytes() : 0);
        if (c->persist_user) {   // the same kernel, compiled with the user's objective inside (user_persist_compile)
            void* args[] = {(void*)&A};
            if (c->kev0) HIPCHK(hipExtModuleLaunchKernel(c->upfn, grid.x * 1024u, 1, 1, 1024, 1, 1, smem, c->stream, args, nullptr, c->kev0, c->kev1, 0));
            else HIPCHK(hipModuleLaunchKernel(c->upfn, grid.x, 1, 1, 1024, 1, 1, (unsigned)smem, c->stream, args, nullptr));
        } else if (c->kev0) hipExtLaunchKernelGGL(k_chain_persist_gen, grid, block, smem, c->stream, c->kev0, c->kev1, 0, A);
        else hipLaunchKernelGGL(k_chain_persist_gen, grid, block, smem, c->stream, A);
    }
    c->cur ^= 1;
    ++c->persist_launches;
    return t1 - t0 + 1;
}

// n_iters iterations onto the stream (smm_bgp_step_async, and persist_repair with the persistent kernel off)
void enqueue_iterations(Ctx* c, int n_iters) {
    int slot = 0;   // profiling: events of this launch
    for (int it = 0; it < n_iters;) {
        const int t = c->iter + 1;
        int n = 0;
        if (persist_usable(c, n_iters - it)) {
            if (c->profiling == 1) HIPCHK(hipEventRecord(c->pev[4 * slot], c->stream));
            if (c->profiling == 2) { c->kev0 = c->pev[4 * slot]; c->kev1 = c->pev[4 * slot + 1]; }
            n = launch_chain_persist(c, n_iters - it);
            c->kev0 = c->kev1 = nullptr;
        }
        if (n > 0) {
            if (c->profiling == 1) { for (int e = 1; e < 4; ++e) HIPCHK(hipEventRecord(c->pev[4 * slot + e], c->stream)); }
            if (c->profiling) c->pev_exch[slot] = 0;
            ++slot;
            const int t1 = c->iter + n;
            c->prev_open = true;
            c->pending = false; c->unresolved = false;
            if (exchange_active(c, t1)) { c->unresolved = true; c->pending = true; }   // resolved in the prologue of the next chain kernel (or by resolve_now)
            c->iter = t1; c->exch_done = false;
            it += n;
            continue;
        }
        // an exchange left to this chain kernel needs its plan: resolve it now if the plan window is about to move on
        if (c->unresolved && !(t >= c->plan_t0 && t < c->plan_t0 + c->plan_w) && !plan_ahead_covers(c, t)) resolve_now(c);
        // ... or if this chain kernel cannot walk (a user objective's launches): the resolution was only put off in case the persistent
        // kernel came next (defer_resolve)
        if (c->unresolved && c->defer_resolve && !c->inline_walk) resolve_now(c);
        ensure_windows(c, t);
        const int flags = (c->prev_open ? F_CLOSE_PREV : 0) | (c->pending ? F_HAS_PENDING : 0) | (c->unresolved ? F_WALK_INLINE : 0);
        const bool kscoped = c->profiling == 2 && c->lvl_exchange && c->lvl_wg == 1024;
        if (c->profiling && !kscoped) HIPCHK(hipEventRecord(c->pev[4 * slot], c->stream));
        if (kscoped) { c->kev0 = c->pev[4 * slot]; c->kev1 = c->pev[4 * slot + 1]; }
        launch_chain_iter(c, t, flags);
        c->kev0 = c->kev1 = nullptr;
        if (c->profiling && !kscoped) HIPCHK(hipEventRecord(c->pev[4 * slot + 1], c->stream));
        c->prev_open = true;
        c->pending = false;
        if (c->profiling) c->pev_exch[slot] = 0;
        c->unresolved = false;
        if (exchange_active(c, t)) {
            const bool cone_next = c->cone_big && !c->nan_values && !c->ext_rec_in && !c->ext_rec_out && !c->ext_vals_out && !c->rec_external &&
                                   t >= c->plan_t0 && t < c->plan_t0 + c->plan_w && c->cone_big_ok[(size_t)(t - c->plan_t0)] != 0u;
            if (cone_next || (c->inline_walk && !(c->gen_keys && (c->deep_plan || c->nan_values))) ||
                (c->defer_resolve && c->persist_on && !c->persist_broken && !c->in_repair && !c->nan_values)) {   // (the key form has no second walk to fall back to)
                c->unresolved = true;   // resolved in the prologue of the next chain kernel (or by resolve_now)
            } else {
                if (kscoped) { c->kev0 = c->pev[4 * slot + 2]; c->kev1 = c->pev[4 * slot + 3]; c->pev_exch[slot] = 1; }
                launch_resolve(c, t, (c->lvl_exchange || c->lds_exchange || c->key_exchange) ? nullptr : c->rec[c->cur]);
                c->kev0 = c->kev1 = nullptr;
            }
            c->pending = true;
        }
        if (c->profiling && !kscoped) { HIPCHK(hipEventRecord(c->pev[4 * slot + 2], c->stream)); HIPCHK(hipEventRecord(c->pev[4 * slot + 3], c->stream)); }
        c->iter = t; c->exch_done = false;
        ++slot; ++it;
    }
    if (c->profiling) c->pev_iters = slot;
}

// A launch of the persistent kernel ended with the error word set (a hard error of the algorithm, AlgoBGP.jl:341,409 — or a tile gave
// up waiting).  Its tiles do not stop at the failing iteration, so: the state of before (persist_snapshot), the history rows of the
// iterations since filled as the constructor fills them, the error word cleared, and the same iterations again, one launch each —
// that path stops at the failing iteration with the documented state (include/smmhip.h), or runs through if the failure was the form's.
void persist_repair(Ctx* c, int n_replay) {
    KParams& P = c->P;
    const size_t N = P.N;
    const int n = n_replay >= 0 ? std::min(n_replay, c->iter - c->snap_iter) : c->iter - c->snap_iter;
    c->in_repair = true;
    ++c->persist_repairs;
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(P.cs, c->snap_cs, N * CSW * 8, hipMemcpyDeviceToDevice));
    HIPCHK(hipMemcpy(c->rec[c->snap_cur], c->snap_rec, N * P.RW * 8, hipMemcpyDeviceToDevice));
    for (int b = 0; b < 2; ++b) {
        HIPCHK(hipMemcpy(c->vals_buf[b], c->snap_vals[b], (N + 4) * 8, hipMemcpyDeviceToDevice));
        if (c->slot8_buf[b]) HIPCHK(hipMemcpy(c->slot8_buf[b], c->snap_slot8[b], (N + 4 + 128) * 8, hipMemcpyDeviceToDevice));
    }
    HIPCHK(hipMemcpy(P.xres, c->snap_xres, (size_t)P.Ng * 8, hipMemcpyDeviceToDevice));
    // (an exchanged chain's row of iteration snap_iter may have been rewritten by the first launch's prologue: the replay rewrites it identically)
    for (int t = c->snap_iter; t < c->iter; ++t)
        HIPCHK(hipMemcpy(P.hrec + (size_t)t * N * P.HW, c->hist_fill, N * P.HW * 8, hipMemcpyDeviceToDevice));
    const unsigned long long e = ERR_NONE;
    HIPCHK(hipMemcpy(P.err, &e, 8, hipMemcpyHostToDevice));
    c->iter = c->snap_iter; c->cur = c->snap_cur; c->slots_iter = c->snap_slots_iter;
    c->pending = c->snap_pending; c->prev_open = c->snap_prev_open; c->unresolved = c->snap_unresolved; c->exch_done = c->snap_exch_done;
    c->plan_w = 0; c->rng_w = 0; c->ps[0].w = c->ps[1].w = 0;   // (the windows are rebuilt: cheap, and nothing assumes where the failed run left them)
    // ... with the exchange of the snapshot's iteration still to be applied, its plan must be in the window the replay starts with: the
    // failed launches may have moved the window on (a step across a window boundary), and what enqueue_iterations does for an
    // iteration outside the window — resolve the pending exchange from "the window that just ended" — would read another window's plan
    if (c->unresolved) ensure_windows(c, c->iter);
    const int prof = c->profiling;
    c->profiling = 0;
    if (P.N < P.Ng) {   // a shard: the same iterations through the windows, one (or two) launches each — every rank does (smm_sync agrees on the error first)
        c->rec_external = false; c->pending_ext = false; c->p2p_current = false; c->p2p_unwaited = false;
        p2p_enqueue(c, n);
    } else enqueue_iterations(c, n);
    c->profiling = prof;
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    c->in_repair = false;
    c->snap_valid = false;
}
// calls that read or change the run's state other than by stepping: first make sure that what stands there is final (see persist_repair)
void settle_persist(Ctx* c) {
    if (!c->snap_valid) return;
    HIPCHK(hipStreamSynchronize(c->stream));
    (void)check_device_error(c);
}

// ---- a shard between its two states (smm_bgp_p2p_step) ----
// "p2p": the records after iteration `iter` are in the ranks' windows (rec_external), the exchange of `iter` not resolved yet (pending_ext);
// "plain": they are in the context's own array, as the persistent form reads and leaves them (unresolved: the exchange still pending).
// plain -> p2p: a publication.  With the exchange of `iter` still pending it travels as it is — a shard cannot settle it alone.
void p2p_publish(Ctx* c) {
    if (c->p2p_current) return;
    const bool carry = c->unresolved && c->P.N < c->P.Ng;
    if (!carry) flush(c);
    // the form is decided from what EVERY rank knows (population, objective, thresholds): a rank that looked at its own shard's
    // values here (an uploaded state with a NaN) could choose differently from its peers, and each side would wait for words the
    // other never sends.  A NaN in any shard reaches every window with the publication (in the slot words themselves, and the NaN word
    // tagged with the epoch): the rows form resolves such an iteration on the exact values inside its launch, the inline form
    // (N_global <= 8192) reports it — on every rank, in the same iteration (include/smmhip.h).
    c->p2p_mode_inline = c->p2p_inline || c->p2p_rows;
    c->P.p2p_epoch += 1;   // (a new generation of tags: words of an earlier publication are nobody's any more)
    launch_p2p_push(c, c->iter, c->rec[c->cur], c->p2p_mode_inline);
    c->p2p_current = true;
    c->pending_ext = false;
    if (carry) { c->pending_ext = true; c->rec_external = true; c->pending = false; c->unresolved = false; }
}
// p2p -> plain: this rank's own records after iteration `iter` out of its window into the context's array (its own stores: complete)
void p2p_to_plain(Ctx* c) {
    if (!c->p2p_current) return;
    const KParams& P = c->P;
    const P2PLayout L = p2p_layout(P.Ng, P.RW);
    if (c->p2p_mode_inline)
        hipLaunchKernelGGL(k_p2p_own_records, dim3((unsigned)((P.N * P.RW + 255) / 256)), dim3(256), 0, c->stream, P, c->iter, c->rec[c->cur]);
    else
        HIPCHK(hipMemcpyAsync(c->rec[c->cur], (const double*)(c->p2p_mine + L.rec[c->iter & 1]) + (size_t)P.offset * P.RW, (size_t)P.N * P.RW * 8,
                              hipMemcpyDeviceToDevice, c->stream));
    c->unresolved = c->pending_ext; c->pending = c->pending_ext;
    c->pending_ext = false; c->rec_external = false; c->p2p_current = false; c->p2p_unwaited = false;
}

// n iterations of a shard onto the stream (smm_bgp_p2p_step; persist_repair with the persistent form off)
void p2p_enqueue(Ctx* c, int n_iters) {
    KParams& P = c->P;
    const P2PLayout L = p2p_layout(P.Ng, P.RW);
    int slot = 0;   // profiling: events of this launch
    for (int it = 0; it < n_iters; ++it) {
        if (persist_sh_usable(c, n_iters - it)) {   // as many of the remaining iterations as the look-ahead windows hold, in ONE launch
            p2p_to_plain(c);
            if (c->profiling == 2) { c->kev0 = c->pev[4 * slot]; c->kev1 = c->pev[4 * slot + 1]; }
            const int n = launch_chain_persist(c, n_iters - it);
            c->kev0 = c->kev1 = nullptr;
            if (n > 0) {
                const int t1 = c->iter + n;
                c->prev_open = true; c->pending = false; c->unresolved = false;
                if (exchange_active(c, t1)) { c->unresolved = true; c->pending = true; }
                c->iter = t1; c->exch_done = false;
                it += n - 1; ++slot;
                continue;
            }
        }
        p2p_publish(c);
        const int t = c->iter + 1;
        const bool prof = c->profiling == 2;
        int flags = (c->prev_open ? F_CLOSE_PREV : 0) | F_GLOBAL_REC;
        if (c->p2p_mode_inline) {
            if (c->pending_ext) {
                flags |= F_HAS_PENDING;
                // the walk of iteration t-1 needs that iteration's plan: where the plan window is about to move on, the
                // exchange is resolved by the stand-alone kernel first (once per window of 256 iterations)
                if (c->p2p_rows) {
                    if (prof) { c->kev0 = c->pev[4 * slot + 2]; c->kev1 = c->pev[4 * slot + 3]; c->pev_exch[slot] = 1; }
                    launch_resolve_rows_window(c, t - 1);
                    c->kev0 = c->kev1 = nullptr;
#ifdef SMM_TEST_HOOKS
                    if (SMM_HOOK("SMMHIP_ROWS_WIN_CHECK")) {   // the same exchange through the unpacked values and the plain kernels
                        std::vector<unsigned long long> a((size_t)P.Ng), b((size_t)P.Ng);
                        HIPCHK(hipStreamSynchronize(c->stream));
                        HIPCHK(hipMemcpy(a.data(), P.xres, a.size() * 8, hipMemcpyDeviceToHost));
                        launch_p2p_unpack(c, t - 1);
                        launch_resolve_window(c, t - 1);
                        HIPCHK(hipStreamSynchronize(c->stream));
                        HIPCHK(hipMemcpy(b.data(), P.xres, b.size() * 8, hipMemcpyDeviceToHost));
                        int bad = 0;
                        for (int g = 0; g < P.Ng; ++g)
                            if (a[g] != b[g] && bad++ < 8) fprintf(stderr, "rows window check: iteration %d chain %d: %llx != %llx\n", t - 1, g, a[g], b[g]);
                        fprintf(stderr, "rows window check: iteration %d: %d of %d differ\n", t - 1, bad, P.Ng);
                    }
#endif
                } else if (t >= c->plan_t0 && t < c->plan_t0 + c->plan_w) flags |= F_WALK_INLINE;
                else {
                    launch_p2p_unpack(c, t - 1);
                    launch_resolve_window(c, t - 1);
                }
            }
            ensure_windows(c, t);
            if (prof) { c->kev0 = c->pev[4 * slot]; c->kev1 = c->pev[4 * slot + 1]; }
            launch_chain_iter_norm_p2p(c, t, flags);
            c->kev0 = c->kev1 = nullptr;
        } else {
            if (c->pending_ext) {   // exchangeMoves! of iteration t-1 (before its plan window can move on)
                if (c->p2p_unwaited) launch_p2p_wait(c);
                if (prof && c->lean_resolve) { c->kev0 = c->pev[4 * slot + 2]; c->kev1 = c->pev[4 * slot + 3]; c->pev_exch[slot] = 1; }
                launch_resolve_window(c, t - 1);
                c->kev0 = c->kev1 = nullptr;
                flags |= F_HAS_PENDING;
            }
            ensure_windows(c, t);
            c->ext_rec_in = (const double*)(c->p2p_mine + L.rec[(t - 1) & 1]);
            c->ext_rec_out = (double*)(c->p2p_mine + L.rec[t & 1]) + (size_t)P.offset * P.RW;
            c->ext_vals_out = (double*)(c->p2p_mine + L.val[t & 1]) + P.offset;
            if (prof) { c->kev0 = c->pev[4 * slot]; c->kev1 = c->pev[4 * slot + 1]; }
            launch_chain_iter(c, t, flags);
            c->kev0 = c->kev1 = nullptr;
            c->ext_rec_in = nullptr; c->ext_rec_out = nullptr; c->ext_vals_out = nullptr;
            launch_p2p_push(c, t, nullptr, false);
        }
        c->prev_open = true;
        c->pending = false;
        c->rec_external = true;
        c->pending_ext = exchange_active(c, t);
        c->iter = t; c->exch_done = false;
        ++slot;
    }
    HIPCHK(hipEventRecord(c->ev1, c->stream));
    HIPCHK(hipGetLastError());
    if (c->profiling == 2) c->pev_iters = slot;
}

}  // namespace

extern "C" {

int smm_abi_version(void) { return SMMHIP_ABI_VERSION; }

// 1 in the build the tests load for their seams (libsmmhip_hooks.so), 0 in the shipped library (not part of the public header)
int smm_debug_has_test_hooks(void) {
#ifdef SMM_TEST_HOOKS
    return 1;
#else
    return 0;
#endif
}

static int register_user_source(const std::string& src, int n_sums, int lanes, int32_t* objective_id_out, const char* user_text = nullptr) {
    std::lock_guard<std::mutex> lock(g_user_mutex);
    std::string err;
    if (!g_rtc.load(err)) { g_create_err = err; return SMM_ERR_HIP; }
    hiprtcProgram prog = nullptr;
    if (g_rtc.create(&prog, src.c_str(), "smm_user_objective.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
        g_create_err = "hiprtcCreateProgram failed";
        return SMM_ERR_HIP;
    }
    const std::string nsd = "-DSMM_NSUMS=" + std::to_string(n_sums > 0 ? n_sums : 1);
    const char* opts[] = {"--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", nsd.c_str()};
    const hiprtcResult rc = g_rtc.compile(prog, 5, opts);
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
    u.lanes = lanes;
    u.n_sums = n_sums > 0 ? n_sums : 1;
    if (user_text) u.source = user_text;
    g_rtc.code(prog, u.code.data());
    g_rtc.destroy(&prog);
    g_user_objectives.push_back(std::move(u));
    *objective_id_out = SMM_OBJ_USER_BASE + (int32_t)g_user_objectives.size() - 1;
    return SMM_OK;
}

int smm_register_user_objective(const char* hip_source, int32_t* objective_id_out) {
    if (!hip_source || !objective_id_out) { g_create_err = "smm_register_user_objective: null argument"; return SMM_ERR_INVALID_ARG; }
    return register_user_source(std::string(USER_PRELUDE) + hip_source + USER_KERNEL, 1, 0, objective_id_out, hip_source);
}

int smm_register_user_objective_lanes(const char* hip_source, int32_t n_sums, int32_t lanes, int32_t* objective_id_out) {
    if (!hip_source || !objective_id_out) { g_create_err = "smm_register_user_objective_lanes: null argument"; return SMM_ERR_INVALID_ARG; }
    if (n_sums < 1 || n_sums > 64 || lanes < 64 || lanes > 1024 || lanes % 64 != 0) {
        g_create_err = "smm_register_user_objective_lanes: need 1 <= n_sums <= 64 and lanes a multiple of 64 in [64, 1024]";
        return SMM_ERR_INVALID_ARG;
    }
    return register_user_source(std::string(USER_PRELUDE_LANES) + hip_source + USER_KERNEL_LANES, n_sums, lanes, objective_id_out, hip_source);
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
    if (c->pstream) { (void)hipStreamSynchronize(c->pstream); (void)hipStreamDestroy(c->pstream); }
    if (c->ev_free) (void)hipEventDestroy(c->ev_free);
    for (Ctx::PlanSet& S : c->ps) { if (S.done) (void)hipEventDestroy(S.done); if (S.ok_host) (void)hipHostFree(S.ok_host); }
    for (void* p : c->allocs) (void)hipFree(p);
    for (void* w : c->p2p_opened) if (w) (void)hipIpcCloseMemHandle(w);
    if (c->p2p_mine) (void)hipFree(c->p2p_mine);
    if (c->umod) (void)hipModuleUnload(c->umod);
    if (c->upmod) (void)hipModuleUnload(c->upmod);
    if (c->utmod) (void)hipModuleUnload(c->utmod);
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
    } else if (prob->objective_id < 0 || (prob->objective_id > SMM_OBJ_DENSE && prob->objective_id != SMM_OBJ_DENSE2))
        return fail(nullptr, SMM_ERR_INVALID_ARG, "unknown objective_id");
    if (is_sim(prob->objective_id) && np != nm)
        return fail(nullptr, SMM_ERR_INVALID_ARG, "objfunc_norm needs one moment per parameter (ObjExamples.jl:66-78)");
    if (opts->batch_size < 1 || opts->batch_size > np || (np % opts->batch_size) != 0)
        return fail(nullptr, SMM_ERR_BAD_BATCH, "batch_size must divide the number of parameters (AlgoBGP.jl:95-103)");
    if (opts->sigma_update_steps < 1) return fail(nullptr, SMM_ERR_INVALID_ARG, "sigma_update_steps < 1");
    if (opts->smpl_iters < 1) return fail(nullptr, SMM_ERR_INVALID_ARG, "smpl_iters < 1 (AlgoBGP.jl:521: at least one proposal try)");
    // the exchange of iteration t reads the history of iteration t-1: the reference starts at algo.i >= 2 (AlgoBGP.jl:637)
    if (opts->exchange_from_iter < 2) return fail(nullptr, SMM_ERR_INVALID_ARG, "exchange_from_iter < 2 (AlgoBGP.jl:637)");
    if (!prob->init || !prob->lb || !prob->ub || !prob->mom || !prob->w)
        return fail(nullptr, SMM_ERR_INVALID_ARG, "smm_problem_t: init / lb / ub / mom / w must not be NULL");
    if (!opts->sigma || !opts->acc_tuner || !opts->min_improve)
        return fail(nullptr, SMM_ERR_INVALID_ARG, "smm_bgp_opts_t: sigma / acc_tuner / min_improve must not be NULL (length N_global)");
    if (prob->n_obj_params > 0 && !prob->obj_params) return fail(nullptr, SMM_ERR_INVALID_ARG, "n_obj_params > 0 but obj_params is NULL");
    if (tab && tab->pairs && tab->n_pairs > 0) {   // injected pair lists: 0 <= i < j < N_global (16-bit packing in the plans)
        for (size_t q = 0; q < (size_t)T * tab->n_pairs; ++q) {
            const int32_t i = tab->pairs[2 * q], j = tab->pairs[2 * q + 1];
            if (i < 0 || j <= i || j >= Ng) return fail(nullptr, SMM_ERR_INVALID_ARG, "smm_tables_t.pairs: need 0 <= i < j < N_global");
        }
    }
    if (tab && tab->prop_normals && tab->prop_tries < 1) return fail(nullptr, SMM_ERR_INVALID_ARG, "prop_normals given but prop_tries < 1");
    if (opts->dist_fun < SMM_DIST_MINUS || opts->dist_fun > SMM_DIST_RELDIFF)
        return fail(nullptr, SMM_ERR_INVALID_ARG, "smm_bgp_opts_t.dist_fun: SMM_DIST_MINUS, SMM_DIST_ABSDIFF or SMM_DIST_RELDIFF");
    if (opts->chol_L && opts->batch_size != np)
        return fail(nullptr, SMM_ERR_BAD_BATCH, "Cholesky proposals (chol_L) draw all parameters in one batch: batch_size must equal np");
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
        c->obj = user_obj ? SMM_OBJ_USER : prob->objective_id == SMM_OBJ_DENSE2 ? SMM_OBJ_DENSE : prob->objective_id;   // (spec v2: the dense kind with P.dense_A2f set)
        c->exchange_from = opts->exchange_from_iter;
        P.exch_from = opts->exchange_from_iter;
        {
            const char* e = SMM_HOOK("SMMHIP_ANY_EXCHANGE");  // test hook: force the any-size resolution kernel
            c->force_any_exchange = e && e[0] == '1';
            const char* d = getenv("SMMHIP_DBG");
            P.dbg = d ? atoi(d) : 0;
            const char* sa = SMM_HOOK("SMMHIP_SCOUT_AFTER");   // test hook: rounds before the scouting form of mysample's late tries
            P.scout_after = sa ? atoi(sa) : SMM_SCOUT_AFTER;
            const char* sg = SMM_HOOK("SMMHIP_SCOUT_GL");
            P.scout_gl = (sg && atoi(sg) == 8) ? 8 : 16;
            const char* tsv = getenv("SMMHIP_TS");
            if (tsv && (tsv[0] == '1' || tsv[0] == '2')) P.ts = dalloc<unsigned long long>(c, (size_t)8 * 65536);
            P.ts_levels = tsv && tsv[0] == '2';   // also a stamp per level of the inline walk (the stamps stretch the levels: not with '1')
            c->ct = 8;   // chains per simulation tile (4 and 16 were measured and rejected: more shock traffic / spills)
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
                c->u_lanes = u.lanes; c->u_nsums = u.n_sums;
            }
            HIPCHK(hipModuleGetFunction(&c->ufn, c->umod, "smm_user_eval_kernel"));
            P.u_theta = dalloc<double>(c, (size_t)N * np);
            P.u_simM = dalloc<double>(c, (size_t)N * nm);
            P.u_value = dalloc<double>(c, N);
            P.u_status = dalloc<int>(c, N);
        }
        if (prob->objective_id == SMM_OBJ_DENSE || prob->objective_id == SMM_OBJ_DENSE2) {
            const bool v2 = prob->objective_id == SMM_OBJ_DENSE2;   // [B, A2, A]: with the 256 x 256 stage
            const size_t nB = (size_t)DENSE_D * np, n2 = v2 ? (size_t)DENSE_D * DENSE_D : 0, nA = (size_t)nm * DENSE_D;
            if (prob->n_obj_params != 0 && (size_t)prob->n_obj_params != nB + n2 + nA)
                throw std::string(v2 ? "SMM_OBJ_DENSE2: obj_params must hold B (256 x np), A2 (256 x 256) and A (nm x 256), or be empty"
                                     : "SMM_OBJ_DENSE: obj_params must hold B (256 x np) and A (nm x 256), or be empty");
            std::vector<double> M(nB + n2 + nA);
            if (prob->n_obj_params) memcpy(M.data(), prob->obj_params, M.size() * 8);
            else {  // N(0,1)/sqrt(fan-in) from the counter RNG, stream 5
                for (size_t i = 0; i < M.size(); i += 2) {
                    double z0, z1;
                    box_muller(philox_stream(opts->seed, 5, (uint32_t)(i >> 1), (uint32_t)((i >> 1) >> 32), 0, 0), z0, z1);
                    M[i] = z0 / sqrt(i < nB ? (double)np : (double)DENSE_D);
                    if (i + 1 < M.size()) M[i + 1] = z1 / sqrt(i + 1 < nB ? (double)np : (double)DENSE_D);
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
                            if (k < nm) Af[(((size_t)o * (DENSE_D / 16) + T) * 4 + s) * 64 + l] = M[nB + n2 + (size_t)k * DENSE_D + d];
                        }
            P.dense_Bf = dupload(c, Bf.data(), Bf.size());
            P.dense_Af = dupload(c, Af.data(), Af.size());
            P.dense_nOt = nOt;
            if (v2) {   // A2 in fragment order [wave][k-step][lane][the wave's two row tiles] (smm_chain.hpp: dense2_tile_n)
                std::vector<double> A2f((size_t)DENSE_D * DENSE_D);
                for (int wv = 0; wv < 8; ++wv)
                    for (int s = 0; s < DENSE_D / 4; ++s)
                        for (int l = 0; l < 64; ++l)
                            for (int tt = 0; tt < 2; ++tt) {
                                const int j = 16 * (2 * wv + tt) + (l & 15), d = 4 * s + (l >> 4);
                                A2f[(((size_t)wv * (DENSE_D / 4) + s) * 64 + l) * 2 + tt] = M[nB + (size_t)j * DENSE_D + d];
                            }
                P.dense_A2f = dupload(c, A2f.data(), A2f.size());
            }
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
        if (opts->chol_L) {
            P.chol_per_chain = opts->chol_per_chain ? 1 : 0;
            P.chol_L = dupload(c, opts->chol_L, (size_t)(P.chol_per_chain ? Ng : 1) * np * np);
        }
        P.dist_fun = opts->dist_fun;
        P.mi_uniform = 1; P.mi_value = opts->min_improve[0];
        for (int i = 1; i < Ng; ++i)
            if (!(opts->min_improve[i] == P.mi_value || (opts->min_improve[i] != opts->min_improve[i] && P.mi_value != P.mi_value))) P.mi_uniform = 0;   // (NaN everywhere is one threshold too: nothing ever swaps)
        // per-chain thresholds (what the reference's API takes: opts["min_improve"] is a vector, AlgoBGP.jl:522): the persistent forms walk them too
        // (a threshold per slot position, smm_walk_lean.hpp PCT) while this context's single iterations keep the forms they had (the level walk on any
        // thresholds): every threshold >= 0 or NaN (the dummy pair's 0 - 0 must not exceed it), dist_fun = `-`
        P.mi_pct = 0;
        if (!P.mi_uniform && opts->dist_fun == SMM_DIST_MINUS) {
            P.mi_pct = 1;
            for (int i = 0; i < Ng; ++i) if (opts->min_improve[i] < 0.0) P.mi_pct = 0;
        }
        const size_t TN = (size_t)T * N;
        if (tab && tab->probs_acc) P.user_utab = dupload(c, tab->probs_acc, TN);
        if (tab && tab->prop_normals && tab->prop_tries > 0) {
            P.rb_tries = tab->prop_tries; P.user_n = 1;
            P.user_ntab = dupload(c, tab->prop_normals, TN * (size_t)tab->prop_tries * np);
        } else {
            // tries of mysample whose normals are made ahead of time (k_pregen_rng); later ones come from the generator inside the chain
            // kernel, ~1.4 us per try and tile.  Past 8 parameters: two (C4, 10 parameters: four cost 9 % — 398 -> 433 M chain-evals/s
            // with two, 447 with one; a first try outside the support is rare, a second one rarer)
            P.rb_tries = np <= 8 ? 8 : 2;
        }
        if ((size_t)P.rb_tries * (size_t)((np + 1) / 2) * (size_t)N >= ((size_t)1 << 31))   // (k_pregen_rng indexes one iteration's pieces in 32 bits)
            throw std::string("injected proposal normals: tries x parameters x chains of one iteration must stay below 2^31 pieces");
        if (tab && tab->pairs && tab->n_pairs > 0) {
            P.n_pairs_tab = tab->n_pairs;
            P.pairtab = dupload(c, tab->pairs, (size_t)T * tab->n_pairs * 2);
            {   // dependency depth of the injected lists (pairs sharing a chain keep their order): the lean walks hold LV_MAXLEV levels
                std::vector<int> last((size_t)Ng);
                for (int it = 0; it < T && !c->deep_plan; ++it) {
                    std::fill(last.begin(), last.end(), 0);
                    for (int q = 0; q < tab->n_pairs; ++q) {
                        const int32_t i = tab->pairs[2 * ((size_t)it * tab->n_pairs + q)], j = tab->pairs[2 * ((size_t)it * tab->n_pairs + q) + 1];
                        const int lv = std::max(last[i], last[j]) + 1;
                        last[i] = last[j] = lv;
                        if (lv > LV_MAXLEV) { c->deep_plan = true; break; }
                    }
                }
            }
        }
        P.RW = even_up(3 + np + nm);
        P.HW = even_up(H_PARAMS + np + nm);
        P.RBW = even_up(1 + P.rb_tries * np);
        const int K = exchange_K(c);
        P.plan_K = K;
        c->lds_exchange = Ng > 1 && Ng <= XLDS_MAX && K >= 1 && K <= Ng && !c->force_any_exchange;
        {
            const char* e = SMM_HOOK("SMMHIP_DATAFLOW_EXCHANGE");  // test hook: force the ticket (data-flow) resolution kernel
            c->lvl_exchange = c->lds_exchange && Ng <= XLVL_MAX && !(e && e[0] == '1');
            c->lvl_soa_exchange = c->lds_exchange && !c->lvl_exchange && !(e && e[0] == '1');
            const char* lw = SMM_HOOK("SMMHIP_LVL_WG");  // tuning hook
            if (lw) c->lvl_wg = atoi(lw);
            const char* be = SMM_HOOK("SMMHIP_BIG_EXCHANGE");  // test hook: force the global-memory level kernels
            // (... and a SHARD whose population is past what the lean plan's 16-byte walk holds in LDS — one min_improve > 0 for all chains of
            // 7400 < N_global <= 8192, e.g. 2 x 4096 with the reference's default threshold —: the global-memory plan lists its tiles' cones,
            // locally numbered, so that the shard can still take the persistent form)
            const bool shard_wide_big = N < Ng && c->lds_exchange && P.mi_uniform && P.mi_value != 0.0 && !(P.mi_value < 0.0) && opts->dist_fun == SMM_DIST_MINUS &&
                                        resolve_lean_bytes(Ng, K, true) > (size_t)160 * 1024;
            const bool force_big = (be && be[0] == '1') || shard_wide_big;
            c->big_exchange = Ng > 1 && Ng <= 65535 && K >= 1 && K <= Ng && !c->force_any_exchange && (force_big || !c->lds_exchange);
            if (c->big_exchange) { c->lds_exchange = false; c->lvl_exchange = false; c->lvl_soa_exchange = false; }
            const char* ke = SMM_HOOK("SMMHIP_KEY_EXCHANGE");   // test hook: "0" keeps the global-memory walk
            c->key_exchange = c->big_exchange && Ng <= XKEY_MAX && K <= XKEY_MAX && !(ke && ke[0] == '0') && opts->dist_fun == SMM_DIST_MINUS;   // (the keys order value_i - value_j)
            // inline exchange walk: single shard, level plan available, and two tiles must still share a CU's 160 KB LDS
            const char* iw = SMM_HOOK("SMMHIP_INLINE_WALK");
            const int tile_ct = is_sim(c->obj) ? c->ct : (c->obj == SMM_OBJ_DENSE ? 16 : 8);
            const size_t tile_b = (tile_smem_base(c, tile_ct) + 15) & ~(size_t)15;
            const char* nf = SMM_HOOK("SMMHIP_NORM_FAST");   // test hook: "0" keeps the general kernel for objfunc_norm
            c->norm_fast = is_sim(c->obj) && np == nm && np <= 4 && opts->batch_size == np && P.dbg == 0 && !opts->chol_L && !(nf && nf[0] == '0');
            {   // more tiles than CUs: the walk-free kernel on half-size workgroups, two to a CU (k_chain_iter_norm_narrow)
                int cus = 256;
                (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device);
                const char* nn = SMM_HOOK("SMMHIP_NORM_NARROW");   // test hook: "0" never, "1" always
                c->norm_narrow = c->norm_fast && ((N + NORM_CT - 1) / NORM_CT > cus || (nn && nn[0] == '1')) && !(nn && nn[0] == '0');
            }
            // k_chain_iter_norm: pair list NOT overlaid; room for either walk (16-byte slots, 4-byte slots + value table)
            const bool wide_form = P.mi_uniform && P.mi_value != 0.0 && !(P.mi_value < 0.0);   // (the lean walk on 16-byte slots, below)
            const size_t walk_b = std::max(walk_slot_bytes(Ng) + (((size_t)K * 4 + 15) & ~(size_t)15),
                                           ((wide_form ? lean_wide_bytes(Ng, K) : lean_walk_bytes(Ng, K)) + 15) & ~(size_t)15);
            c->inline_walk = !(iw && iw[0] == '0') && c->lvl_exchange && N == Ng && c->obj != SMM_OBJ_USER &&
                             (!c->norm_fast || opts->dist_fun == SMM_DIST_MINUS) &&   // (k_chain_iter_norm's walks are for `-`)
                             (c->norm_fast ? walk_b + norm_tile_doubles(np) * 8 <= (size_t)160 * 1024
                                           : walk_slot_bytes(Ng) + std::max(tile_b, (size_t)K * 4) <= (size_t)80 * 1024);
            P.tile_off = c->inline_walk ? (int)((c->norm_fast ? walk_b : walk_slot_bytes(Ng)) / sizeof(double)) : 0;
            // (the lean plan: the same conditions as further down, where its tables are allocated)
            const char* kw0 = SMM_HOOK("SMMHIP_KEY_WALK");
            const bool lean_plan = P.mi_uniform && !(P.mi_value < 0.0) && opts->dist_fun == SMM_DIST_MINUS && K <= XLDS_MAX && !(kw0 && kw0[0] == '0') &&
                                   (P.mi_value == 0.0 || resolve_lean_bytes(Ng, K, true) <= (size_t)160 * 1024);
            // two tiles per workgroup share one walk (the 2p/2m-style simulation tile of 8 chains only)
            const char* tp = SMM_HOOK("SMMHIP_TPW");
            int n_cu = 256;
            (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, c->device);
            // ... worth it only when two tiles would share a CU anyway (more tiles than CUs)
            const bool force2 = tp && tp[0] == '2';   // test hook: two tiles per workgroup at any size
            c->tpw = (c->inline_walk && !c->norm_fast && (is_sim(c->obj) ? c->ct == 8 : c->obj != SMM_OBJ_DENSE) && ((N + 7) / 8 > n_cu || force2) &&
                      !(tp && tp[0] == '1') &&
                      walk_slot_bytes(Ng) + std::max(2 * tile_b, (size_t)K * 4) <= (size_t)160 * 1024) ? 2 : 1;
            // k_chain_iter on the lean walk (16-byte slots, padded pair list): where its somewhat larger LDS keeps the same budget
            c->gen_lean = c->inline_walk && !c->norm_fast && lean_plan &&
                          tile_lean_slot_bytes(Ng) + std::max((size_t)c->tpw * tile_b, (size_t)lean_walk_Kp(K) * 4) <= (size_t)(c->tpw == 2 ? 160 : 80) * 1024;
            if (c->gen_lean) { P.tile_off = (int)(tile_lean_slot_bytes(Ng) / sizeof(double)); P.gen_lean = 1; }
            // single shards of 4096 < N <= 8192 chains without a simulation (banana, BASELINE config 4): the key walk inline, two
            // 16-chain tiles per workgroup of 1024 lanes (256 workgroups at 8192 chains: one per CU, one launch per iteration)
            {
                const size_t tile16 = (tile_smem_base(c, 16) + 15) & ~(size_t)15;
                const size_t slots = (size_t)(((Ng + 3) & ~3) + 4) * 8;
                c->gen_keys = !c->inline_walk && !(iw && iw[0] == '0') && obj_kind(c->obj) == 0 && c->obj != SMM_OBJ_USER && N == Ng && Ng > XLVL_MAX &&
                              Ng <= XLDS_MAX && K <= XLDS_MAX && c->lds_exchange && P.mi_uniform && P.mi_value == 0.0 && opts->dist_fun == SMM_DIST_MINUS &&
                              !(kw0 && kw0[0] == '0') && slots + std::max(2 * tile16, (size_t)lean_walk_Kp(K) * 4) <= (size_t)160 * 1024;
                if (c->gen_keys) { c->inline_walk = true; c->tpw = 2; P.gen_lean = 2; P.tile_off = (int)(slots / sizeof(double)); }
                // the dense objective (BASELINE config 5): its tile fills a CU's LDS (143 KB at 50 parameters / 50 moments), so the key
                // walk's slots and lists lie UNDER the tile's blocks — the walk is over before anything of the tile is written
                const char* dk = SMM_HOOK("SMMHIP_DENSE_KEYS");   // test hook: "0" keeps the stand-alone resolution
                c->dense_keys = !c->inline_walk && !c->gen_keys && !(iw && iw[0] == '0') && !(dk && dk[0] == '0') && c->obj == SMM_OBJ_DENSE && N == Ng &&
                                Ng >= 2 && Ng <= XLVL_MAX && K <= XLVL_MAX && c->lds_exchange && P.mi_uniform && P.mi_value == 0.0 &&
                                opts->dist_fun == SMM_DIST_MINUS && !(kw0 && kw0[0] == '0') && lean_walk_unit(Ng) == 8 &&
                                std::max(tile16, slots + std::max((size_t)CONE_LEVELS * 64 * 4, (size_t)lean_walk_Kp(K) * 4)) <= (size_t)160 * 1024;
                if (c->dense_keys) { c->gen_keys = true; c->inline_walk = true; c->tpw = 1; P.gen_lean = 2; P.tile_off = 0; }
                // (its slots are written by the accept step, like the headline kernel's: allocated further down, with the lean plan)
            }
        }
        {   // look-ahead window: as many iterations as ~192 MiB of tables allow, at most 256
            const size_t per_iter = (size_t)P.RBW * N * 8 + (size_t)K * 36 + (c->big_exchange ? BigPlanScratch::words(Ng, K) * 4 : 0);
            (void)per_iter;
            // Two windows with budgets of their own (288 GB of HBM: a few hundred MiB of look-ahead tables are nothing), at most 256
            // iterations each.  The plan kernels are launched with one workgroup per iteration of the window: a short window leaves
            // the chip idle while they run — at 32768 chains k_exch_plan_big needs 1.7 ms per launch, which with the 26 iterations the
            // former common budget of 192 MiB allowed was 63 us per iteration, more than the exchange itself (round 3, rocprofv3).
            const bool pregen = !(c->norm_fast && !(tab && tab->prop_normals) && !(tab && tab->probs_acc));   // (k_chain_iter_norm draws in the kernel)
            const size_t rb_iter = (size_t)P.RBW * N * 8;
            // the workgroups' cones of the inline key walk (smm_cone.hpp): where k_chain_iter walks 8192 chains' keys in every workgroup
            const char* nc = SMM_HOOK("SMMHIP_NO_CONE");   // test hook: every workgroup walks the whole list
            const bool want_cone = c->gen_keys && !(nc && nc[0] == '1') &&
                                   (c->dense_keys ? (N % 16 == 0 && N / 16 <= 256) : (c->tpw == 2 && N % 32 == 0 && N / 32 <= 256));
            const int cone_ct = c->dense_keys ? 16 : 32;
            // the persistent chain kernel (smm_chain_persist.hpp): objfunc_norm with at most two moments (the lane's shocks of ONE moment
            // stay in registers), a single shard of at most one 16-chain tile per CU, the key walk's conditions (one threshold 0, `-`),
            // a pair list the lean plan holds
            const char* pe = SMM_HOOK("SMMHIP_PERSIST");   // test hook: "0" never
            int n_cus = 256;
            (void)hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, c->device);
            const char* ploc = SMM_HOOK("SMMHIP_PERSIST_LOC");   // test hook: "1" the locally numbered form wherever it applies, "0" never
            // ... and its form for objectives without a simulation (smm_chain_persist_gen.hpp): where k_chain_iter walks its workgroups'
            // cones inline (4096 < N <= 8192 in whole workgroups of 32 chains, one per CU), one proposal batch, isotropic proposals
            const bool want_persist_gen = want_cone && !c->dense_keys && c->obj == SMM_OBJ_BANANA && np <= PG_MAXP && nm <= PG_MAXP && opts->batch_size == np && !opts->chol_L &&
                                          N == Ng && N / PG_CT <= n_cus && !c->deep_plan && P.dbg == 0 &&
                                          persist_gen_smem_bytes(Ng, np, P.RW, P.HW) <= (size_t)160 * 1024 && !(pe && pe[0] == '0');
            // ... and for the smaller populations of the same objective (up to 4096 chains in whole groups of 32: VERDICT r4 "banana at 2048
            // chains takes the per-iteration path"): the same kernel — the cones are listed behind the lean plan whether or not the
            // per-iteration kernel walks them
            const bool want_persist_gen_small = !want_cone && c->obj == SMM_OBJ_BANANA && np <= PG_MAXP && nm <= PG_MAXP && opts->batch_size == np && !opts->chol_L &&
                                                N == Ng && N % PG_CT == 0 && N / PG_CT >= 1 && N / PG_CT <= n_cus && c->inline_walk && c->lds_exchange && P.mi_uniform &&
                                                P.mi_value == 0.0 && opts->dist_fun == SMM_DIST_MINUS && K <= XLDS_MAX && !c->deep_plan && P.dbg == 0 &&
                                                persist_gen_smem_bytes(Ng, np, P.RW, P.HW) <= (size_t)160 * 1024 && !(pe && pe[0] == '0');
            // ... and a USER objective (one thread per evaluation) in the same loop: the kernel is compiled with the user's source inside
            // (user_persist_compile), on demand, further down
            const bool want_persist_user = user_obj && c->u_lanes == 0 && np <= PG_MAXP && nm <= PG_MAXP && opts->batch_size == np && !opts->chol_L &&
                                           N == Ng && N % PG_CT == 0 && N / PG_CT >= 1 && N / PG_CT <= n_cus && c->lds_exchange && P.mi_uniform &&
                                           P.mi_value == 0.0 && opts->dist_fun == SMM_DIST_MINUS && K <= XLDS_MAX && !c->deep_plan && P.dbg == 0 &&
                                           persist_gen_smem_bytes(Ng, np, P.RW, P.HW) + persist_gen_user_bytes() <= (size_t)160 * 1024 && !(pe && pe[0] == '0');
            // ... and on LOCALLY NUMBERED cones (smm_chain_persist_loc.hpp): the same objective with one threshold >= 0 (or NaN: nothing
            // ever swaps) for all chains — min_improve > 0 is the reference's default (AlgoBGP.jl:522) —, whatever the population's size
            // does to the tile's LDS
            const bool mi_ok = (P.mi_uniform && !(P.mi_value < 0.0)) || P.mi_pct;   // one threshold >= 0 (or NaN) for all chains, or one per chain, each >= 0 (or NaN)
            const bool want_persist_loc = c->norm_fast && np <= 2 && ns <= WG * PR_ZR && N == Ng && Ng >= 2 && c->inline_walk && mi_ok &&
                                          opts->dist_fun == SMM_DIST_MINUS && K <= XLDS_MAX && Ng <= XLDS_MAX && !c->deep_plan && (N + NORM_CT - 1) / NORM_CT <= n_cus &&
                                          P.dbg == 0 && !(pe && pe[0] == '0') && !(ploc && ploc[0] == '0');
            // ... and as a shard of a sharded run (one process per GPU: smm_bgp_p2p_step): the same kernel, the ring in the ranks' windows
            const bool want_persist_sh = N < Ng && N > 0 && Ng % N == 0 && opts->chain_offset % N == 0 && Ng / N <= P2P_MAXG && N % NORM_CT == 0 && c->norm_fast && np <= 2 &&
                                         ns <= WG * PR_ZR && P.mi_uniform && !(P.mi_value < 0.0) && opts->dist_fun == SMM_DIST_MINUS && !c->deep_plan && N / NORM_CT <= n_cus &&
                                         P.dbg == 0 && !(pe && pe[0] == '0') && !(ploc && ploc[0] == '0') &&
                                         (c->lds_exchange ? K <= XLDS_MAX : (c->big_exchange && Ng <= 32768 && K <= 65535 && (size_t)Ng * 4 <= (size_t)160 * 1024));
            // ... and for the objectives a whole tile evaluates (smm_chain_persist_tile.hpp): objfunc_norm with any number of parameters — the
            // reference's larger examples have 6 and 18, Examples.jl:210-230, 232-319 — and the dense simulation (BASELINE config 5); one
            // threshold >= 0 (or NaN) for all chains, isotropic proposals, one 16-chain tile per workgroup, all of them resident
            const char* ptile = SMM_HOOK("SMMHIP_PERSIST_TILE");   // test hook: "0" never
            const int tile_kind = obj_kind(c->obj);
            // ... and a USER objective in its map-reduce form (smm_register_user_objective_lanes) whose lanes are a whole share of the tile's 512
            const bool user_tile = user_obj && c->u_lanes > 0 && c->u_lanes <= WG && WG % c->u_lanes == 0 && PT_CT % (WG / c->u_lanes) == 0 && !P.mi_pct;   // (compiled for ONE threshold)
            const bool want_persist_tile = (tile_kind == 1 || tile_kind == 2 || user_tile) && !(c->norm_fast && np <= 2 && ns <= WG * PR_ZR) && N == Ng && Ng >= 2 && c->lds_exchange &&
                                           mi_ok && opts->dist_fun == SMM_DIST_MINUS && K <= XLDS_MAX && Ng <= XLDS_MAX && !c->deep_plan &&
                                           !opts->chol_L && P.dbg == 0 && !(pe && pe[0] == '0') && !(ptile && ptile[0] == '0') && P.RW <= PT_LPC * PT_NJ &&
                                           (tile_kind != 2 || N % PT_CT == 0) && (N + PT_CT - 1) / PT_CT <= 2 * n_cus &&
                                           persist_tile_smem(c) <= (size_t)160 * 1024;
            const size_t persist_tiles = (want_persist_gen || want_persist_gen_small || want_persist_user) ? (size_t)N / PG_CT : (size_t)(N + NORM_CT - 1) / NORM_CT;
            // large single shards of objfunc_norm (C3 on one GPU): the narrow chain kernel's tiles walk their own, locally numbered cones
            // (smm_cone_big.hpp) instead of waiting for the one-workgroup resolution between two launches
            const char* cbh = SMM_HOOK("SMMHIP_CONE_BIG");   // test hook: "0" keeps k_exch_resolve_rows between the launches
            const char* kwb = SMM_HOOK("SMMHIP_KEY_WALK");
            const bool want_cone_big = c->big_exchange && c->key_exchange && P.mi_uniform && P.mi_value == 0.0 && !(kwb && kwb[0] == '0') && c->norm_fast && c->norm_narrow &&
                                       N == Ng && N % NORM_CT == 0 && Ng <= 32768 && K <= 65535 && P.dbg == 0 && !(cbh && cbh[0] == '0') &&
                                       (size_t)Ng * 4 <= (size_t)160 * 1024;
            const size_t plan_iter = (want_cone_big ? (size_t)(N / NORM_CT) * ((CONE_LEVELS * 64 + CONE_HDRW) * 4 + CONE_GCAP * 2) + cone_big_scratch_words(Ng, K) * 4 : 0) + (size_t)K * 36 + (c->big_exchange ? BigPlanScratch::words(Ng, K) * 4 + (size_t)(XROWS_MAX + 1) * XWG * 4 : 0) +
                                     (size_t)lean_walk_Kp(K) * 4 + 1024 + (want_cone ? (size_t)(N / cone_ct) * (CONE_LEVELS * 64 + CONE_HDRW) * 4 + 4 : 0) +
                                     ((want_persist_loc || want_persist_sh || want_persist_tile) ? persist_tiles * ((CONE_LEVELS * 64 + CONE_HDRW) * 4 + CONE_GCAP * 2) + 4 : 0) +
                                     ((want_persist_sh && c->big_exchange) ? cone_big_scratch_words(Ng, K) * 4 : 0) +
                                     (want_persist_gen ? persist_tiles * (CONE_GCAP * 2) : 0) +
                                     ((want_persist_gen_small || want_persist_user) ? persist_tiles * ((CONE_LEVELS * 64 + CONE_HDRW) * 4 + CONE_GCAP * 2) + 4 : 0);
            c->win_cap = pregen ? (int)std::max<size_t>(1, std::min<size_t>(256, ((size_t)768 << 20) / rb_iter)) : 1;
            c->win_cap = std::min(c->win_cap, T);
            c->plan_cap = (int)std::max<size_t>(1, std::min<size_t>(256, ((size_t)1536 << 20) / plan_iter));
            c->plan_cap = std::min(c->plan_cap, T);
            if (const char* pc = SMM_HOOK("SMMHIP_PLAN_CAP")) c->plan_cap = std::max(1, std::min(c->plan_cap, atoi(pc)));   // test hook: short plan windows
            c->win_rb = dalloc<double>(c, (size_t)c->win_cap * N * P.RBW);
            HIPCHK(hipMemset(c->win_rb, 0, (size_t)c->win_cap * N * P.RBW * 8));
            if (c->big_exchange) {
                c->win_lv_pairs = dalloc<uint32_t>(c, (size_t)c->plan_cap * K);
                c->win_lv_mi = dalloc<double>(c, (size_t)c->plan_cap * K);
                c->win_lv_off = dalloc<uint32_t>(c, (size_t)c->plan_cap * (K + 2));
                c->big_scratch = dalloc<uint32_t>(c, (size_t)c->plan_cap * BigPlanScratch::words(Ng, K));
                const char* kw = SMM_HOOK("SMMHIP_KEY_WALK");   // test hook: "0" keeps k_exch_resolve_key
                if (c->key_exchange && P.mi_uniform && P.mi_value == 0.0 && !(kw && kw[0] == '0')) {
                    c->rows_exchange = true;
                    P.rows_cap = std::min(XROWS_MAX, (K + XWG - 1) / XWG + LV_MAXLEV);
                    c->win_lv_rows = dalloc<uint32_t>(c, (size_t)c->plan_cap * P.rows_cap * XWG);
                    c->win_lv_rowinfo = dalloc<uint32_t>(c, (size_t)c->plan_cap * 4);
                    c->slots17 = dalloc<uint32_t>(c, (size_t)Ng + 4);
                    c->nan_flags = dalloc<uint32_t>(c, 4);
                    HIPCHK(hipMemset(c->nan_flags, 0, 16));
                    if (want_cone_big) {
                        const size_t tiles = (size_t)N / NORM_CT;
                        c->cone_big = true;
                        P.cone_tiles = (int)tiles; P.cone_ct = NORM_CT;
                        P.cone_ok = dalloc<uint32_t>(c, (size_t)c->plan_cap);
                        P.cone_hdr = dalloc<uint32_t>(c, (size_t)c->plan_cap * tiles * CONE_HDRW);
                        P.cone_pairs = dalloc<uint32_t>(c, (size_t)c->plan_cap * tiles * (CONE_LEVELS * 64));
                        P.cone_gather = dalloc<uint16_t>(c, (size_t)c->plan_cap * tiles * CONE_GCAP);
                        HIPCHK(hipMemset((void*)P.cone_ok, 0, (size_t)c->plan_cap * 4));
                        c->cb_scratch = dalloc<uint32_t>(c, (size_t)c->plan_cap * cone_big_scratch_words(Ng, K));
                        const char* pah = SMM_HOOK("SMMHIP_PLAN_AHEAD");   // test hook: "0" plans each window on the main stream when it is entered
                        if (!(pah && pah[0] == '0')) {
                            c->plan_ahead = true;
                            HIPCHK(hipStreamCreateWithFlags(&c->pstream, hipStreamNonBlocking));
                            HIPCHK(hipEventCreateWithFlags(&c->ev_free, hipEventDisableTiming));
                            for (int b = 0; b < 2; ++b) {
                                Ctx::PlanSet& S = c->ps[b];
                                S.lv_pairs = b ? dalloc<uint32_t>(c, (size_t)c->plan_cap * K) : c->win_lv_pairs;
                                S.lv_mi = b ? dalloc<double>(c, (size_t)c->plan_cap * K) : c->win_lv_mi;
                                S.lv_off = b ? dalloc<uint32_t>(c, (size_t)c->plan_cap * (K + 2)) : c->win_lv_off;
                                S.lv_rows = b ? dalloc<uint32_t>(c, (size_t)c->plan_cap * P.rows_cap * XWG) : c->win_lv_rows;
                                S.lv_rowinfo = b ? dalloc<uint32_t>(c, (size_t)c->plan_cap * 4) : c->win_lv_rowinfo;
                                S.cone_ok = b ? dalloc<uint32_t>(c, (size_t)c->plan_cap) : (uint32_t*)P.cone_ok;
                                S.cone_hdr = b ? dalloc<uint32_t>(c, (size_t)c->plan_cap * tiles * CONE_HDRW) : (uint32_t*)P.cone_hdr;
                                S.cone_pairs = b ? dalloc<uint32_t>(c, (size_t)c->plan_cap * tiles * (CONE_LEVELS * 64)) : (uint32_t*)P.cone_pairs;
                                S.cone_gather = b ? dalloc<uint16_t>(c, (size_t)c->plan_cap * tiles * CONE_GCAP) : P.cone_gather;
                                HIPCHK(hipHostMalloc((void**)&S.ok_host, (size_t)c->plan_cap * 4, hipHostMallocDefault));
                                HIPCHK(hipEventCreateWithFlags(&S.done, hipEventDisableTiming));
                            }
                        }
                        for (int b = 0; b < 2; ++b) c->slot8_buf[b] = dalloc<uint2>(c, (size_t)N + 4 + 128);
                        P.slot8 = c->slot8_buf[0];
                        P.walk_flags = dalloc<uint32_t>(c, 4);
                        HIPCHK(hipMemset(P.walk_flags, 0, 16));
                    }
                }
            }
            if (c->big_exchange && want_persist_sh) {   // a shard of a large population: its own tiles' cones, locally numbered (smm_cone_big.hpp)
                const size_t tiles = persist_tiles;
                P.cone_tiles = (int)tiles; P.cone_ct = NORM_CT;
                P.cone_ok = dalloc<uint32_t>(c, (size_t)c->plan_cap);
                P.cone_hdr = dalloc<uint32_t>(c, (size_t)c->plan_cap * tiles * CONE_HDRW);
                P.cone_pairs = dalloc<uint32_t>(c, (size_t)c->plan_cap * tiles * (CONE_LEVELS * 64) + 1024);
                P.cone_gather = dalloc<uint16_t>(c, (size_t)c->plan_cap * tiles * CONE_GCAP + 512);
                HIPCHK(hipMemset((void*)P.cone_ok, 0, (size_t)c->plan_cap * 4));
                c->cb_scratch = dalloc<uint32_t>(c, (size_t)c->plan_cap * cone_big_scratch_words(Ng, K));
                c->persist = true; c->persist_loc = true; c->persist_wide = P.mi_value != 0.0; c->persist_sh = true; c->persist_sh_big = true;
                if (const char* rk = SMM_HOOK("SMMHIP_PR_RING")) { const int k = atoi(rk); if (k == 2 || k == 4) c->pr_ring_k = k; }
                if (const char* st = SMM_HOOK("SMMHIP_PR_SLOW_TILE")) c->pr_slow_tile = atoi(st);
                if (const char* su = SMM_HOOK("SMMHIP_PR_SLOW_US")) c->pr_slow_ticks = 100 * atoi(su);
            }
            if (c->lds_exchange) {
                c->win_plan = dalloc<unsigned long long>(c, (size_t)c->plan_cap * K);
                c->win_plan_mi = dalloc<double>(c, (size_t)c->plan_cap * K);
                c->win_lv_pairs = dalloc<uint32_t>(c, (size_t)c->plan_cap * K);
                c->win_lv_mi = dalloc<double>(c, (size_t)c->plan_cap * K);
                c->win_lv_off = dalloc<uint32_t>(c, (size_t)c->plan_cap * (K + 2));
                // the lean walk (smm_walk_lean.hpp): one min_improve for every chain — 0: 8-byte slots of order keys; > 0 (or NaN:
                // nothing ever swaps): 16-byte slots of values, as far as the 160 KB of LDS reach (~7400 chains)
                const char* kw = SMM_HOOK("SMMHIP_KEY_WALK");   // test hook: "0" keeps the walks on 16-byte / split slots
                const bool keys = P.mi_uniform && P.mi_value == 0.0;
                const bool wide = P.mi_uniform && !keys && !(P.mi_value < 0.0) && resolve_lean_bytes(Ng, K, true) <= (size_t)160 * 1024;
                // (thresholds by chain: the lean PLAN — padded levels on 16-byte units, the tiles' cones behind it — is made for the persistent launches; the
                // stand-alone lean resolution and the inline lean walks, which test ONE threshold, stay off: lean_resolve below, launch_chain_iter_norm)
                const bool pct = P.mi_pct != 0 && (want_persist_loc || want_persist_tile) && !c->persist;
                if ((keys || wide || pct) && K <= XLDS_MAX && !(kw && kw[0] == '0') && opts->dist_fun == SMM_DIST_MINUS) {
                    P.lean_wide = (wide || pct) ? 1 : 0;
                    P.plan_Kp = lean_walk_Kp(K);
                    P.lean_unit = (wide || pct) ? lean_wide_unit(Ng) : lean_walk_unit(Ng);
                    c->win_lv_pairs_p = dalloc<uint32_t>(c, (size_t)c->plan_cap * P.plan_Kp + 512);   // (+512: whole 1 KB pieces may be read past the last iteration's words)
                    c->win_lv_offp = dalloc<uint32_t>(c, (size_t)c->plan_cap * LV_OFFP);
                    c->lean_resolve = !pct;
                    if (keys && ((c->norm_fast && c->inline_walk && Ng <= XLVL_MAX && K <= XLVL_MAX) || c->gen_keys)) {   // ... in the prologue of k_chain_iter_norm, or of k_chain_iter (key form)
                        for (int b = 0; b < 2; ++b) c->slot8_buf[b] = dalloc<uint2>(c, (size_t)N + 4 + 128);
                        P.slot8 = c->slot8_buf[0];
                        P.walk_flags = dalloc<uint32_t>(c, 4);
                        HIPCHK(hipMemset(P.walk_flags, 0, 16));
                        if (want_cone) {
                            const size_t tiles = (size_t)N / cone_ct;
                            c->cone = true;
                            P.cone_tiles = (int)tiles; P.cone_ct = cone_ct;
                            P.cone_ok = dalloc<uint32_t>(c, (size_t)c->plan_cap);
                            P.cone_hdr = dalloc<uint32_t>(c, (size_t)c->plan_cap * tiles * CONE_HDRW);
                            P.cone_pairs = dalloc<uint32_t>(c, (size_t)c->plan_cap * tiles * (CONE_LEVELS * 64) + 1024);   // (+: whole 1 KB pieces are fetched)
                            HIPCHK(hipMemset((void*)P.cone_ok, 0, (size_t)c->plan_cap * 4));
                            if (want_persist_gen) {
                                P.cone_gather = dalloc<uint16_t>(c, (size_t)c->plan_cap * tiles * CONE_GCAP + 512);
                                P.pr_slot = (uint2*)dalloc<unsigned char>(c, persist_ring_slot_bytes(Ng));
                                P.pr_rec = (uint4*)dalloc<unsigned char>(c, persist_ring_rec_bytes(Ng, P.RW));
                                P.pr_progress = dalloc<uint32_t>(c, tiles);
                                P.pr_ctl = dalloc<uint32_t>(c, 4);
                                HIPCHK(hipMemset(P.pr_slot, 0, persist_ring_slot_bytes(Ng)));
                                HIPCHK(hipMemset(P.pr_rec, 0, persist_ring_rec_bytes(Ng, P.RW)));
                                HIPCHK(hipMemset(P.pr_progress, 0, tiles * 4));
                                HIPCHK(hipMemset(P.pr_ctl, 0, 16));
                                c->persist = true; c->persist_gen = true;
                                if (const char* rk = SMM_HOOK("SMMHIP_PR_RING")) { const int k = atoi(rk); if (k == 2 || k == 4) c->pr_ring_k = k; }
                                if (const char* st = SMM_HOOK("SMMHIP_PR_SLOW_TILE")) c->pr_slow_tile = atoi(st);
                                if (const char* su = SMM_HOOK("SMMHIP_PR_SLOW_US")) c->pr_slow_ticks = 100 * atoi(su);
                            }
                        }
                    }
                    bool user_ok = false;
                    if (want_persist_user && keys && !c->persist) {   // the kernel with the user's objective inside: compiled now (once per registered objective)
                        std::lock_guard<std::mutex> lock(g_user_mutex);
                        UserObjective& u = g_user_objectives[prob->objective_id - SMM_OBJ_USER_BASE];
                        if (user_persist_compile(u)) {
                            HIPCHK(hipModuleLoadData(&c->upmod, u.persist_code.data()));
                            HIPCHK(hipModuleGetFunction(&c->upfn, c->upmod, "smm_user_persist_kernel"));
                            user_ok = true;
                        } else if (getenv("SMMHIP_VERBOSE")) fprintf(stderr, "libsmmhip: the persistent form of this user objective is not available:\n%s\n", u.persist_log.c_str());
                    }
                    if ((want_persist_gen_small || user_ok) && keys && !c->persist) {
                        const size_t tiles = persist_tiles;
                        P.cone_tiles = (int)tiles; P.cone_ct = PG_CT;
                        P.cone_ok = dalloc<uint32_t>(c, (size_t)c->plan_cap);
                        P.cone_hdr = dalloc<uint32_t>(c, (size_t)c->plan_cap * tiles * CONE_HDRW);
                        P.cone_pairs = dalloc<uint32_t>(c, (size_t)c->plan_cap * tiles * (CONE_LEVELS * 64) + 1024);
                        P.cone_gather = dalloc<uint16_t>(c, (size_t)c->plan_cap * tiles * CONE_GCAP + 512);
                        HIPCHK(hipMemset((void*)P.cone_ok, 0, (size_t)c->plan_cap * 4));
                        P.pr_slot = (uint2*)dalloc<unsigned char>(c, persist_ring_slot_bytes(Ng));
                        P.pr_rec = (uint4*)dalloc<unsigned char>(c, persist_ring_rec_bytes(Ng, P.RW));
                        P.pr_progress = dalloc<uint32_t>(c, tiles);
                        P.pr_ctl = dalloc<uint32_t>(c, 4);
                        HIPCHK(hipMemset(P.pr_slot, 0, persist_ring_slot_bytes(Ng)));
                        HIPCHK(hipMemset(P.pr_rec, 0, persist_ring_rec_bytes(Ng, P.RW)));
                        HIPCHK(hipMemset(P.pr_progress, 0, tiles * 4));
                        HIPCHK(hipMemset(P.pr_ctl, 0, 16));
                        c->persist = true; c->persist_gen = true; c->persist_user = user_ok; c->defer_resolve = user_ok;
                        if (const char* rk = SMM_HOOK("SMMHIP_PR_RING")) { const int k = atoi(rk); if (k == 2 || k == 4) c->pr_ring_k = k; }
                        if (const char* st = SMM_HOOK("SMMHIP_PR_SLOW_TILE")) c->pr_slow_tile = atoi(st);
                        if (const char* su = SMM_HOOK("SMMHIP_PR_SLOW_US")) c->pr_slow_ticks = 100 * atoi(su);
                    }
                    if ((want_persist_loc || want_persist_sh) && !c->persist && c->norm_fast) {   // (the lean plan stands: k_exch_plan lists the tiles' cones behind it)
                        const size_t tiles = persist_tiles;
                        P.cone_tiles = (int)tiles; P.cone_ct = NORM_CT;
                        P.cone_ok = dalloc<uint32_t>(c, (size_t)c->plan_cap);
                        P.cone_hdr = dalloc<uint32_t>(c, (size_t)c->plan_cap * tiles * CONE_HDRW);
                        P.cone_pairs = dalloc<uint32_t>(c, (size_t)c->plan_cap * tiles * (CONE_LEVELS * 64) + 1024);   // (+: whole 1 KB pieces are fetched)
                        P.cone_gather = dalloc<uint16_t>(c, (size_t)c->plan_cap * tiles * CONE_GCAP + 512);
                        HIPCHK(hipMemset((void*)P.cone_ok, 0, (size_t)c->plan_cap * 4));
                        if (!want_persist_sh) {   // (a shard's ring lives in its p2p window: smm_bgp_p2p_init)
                            const PrWin WL = pr_win_layout(Ng, P.RW, 1, (int)tiles);
                            c->prw = dalloc<unsigned char>(c, WL.total);
                            HIPCHK(hipMemset(c->prw, 0, WL.total));
                        }
                        c->persist = true; c->persist_loc = true; c->persist_wide = wide || pct; c->persist_sh = want_persist_sh;
                        if (const char* rk = SMM_HOOK("SMMHIP_PR_RING")) { const int k = atoi(rk); if (k == 2 || k == 4) c->pr_ring_k = k; }
                        if (const char* st = SMM_HOOK("SMMHIP_PR_SLOW_TILE")) c->pr_slow_tile = atoi(st);
                        if (const char* su = SMM_HOOK("SMMHIP_PR_SLOW_US")) c->pr_slow_ticks = 100 * atoi(su);
                    }
                    bool tile_ok = want_persist_tile && !c->persist;
                    if (tile_ok && user_tile) {   // the tile kernel with the user's map-reduce objective inside: compiled now (once per registered objective)
                        std::lock_guard<std::mutex> lock(g_user_mutex);
                        UserObjective& u = g_user_objectives[prob->objective_id - SMM_OBJ_USER_BASE];
                        if (user_tile_compile(u)) {
                            HIPCHK(hipModuleLoadData(&c->utmod, u.tile_code.data()));
                            HIPCHK(hipModuleGetFunction(&c->utfn, c->utmod, "smm_user_persist_tile_kernel"));
                        } else {
                            tile_ok = false;
                            if (getenv("SMMHIP_VERBOSE")) fprintf(stderr, "libsmmhip: the persistent form of this user objective is not available:\n%s\n", u.tile_log.c_str());
                        }
                    }
                    if (tile_ok) {   // (the lean plan stands: k_exch_plan lists the tiles' cones and gather lists behind it)
                        const size_t tiles = (size_t)(N + PT_CT - 1) / PT_CT;
                        if (!P.cone_ok) {   // (the dense tiles of the per-iteration kernel walk the same cones: want_cone above)
                            P.cone_tiles = (int)tiles; P.cone_ct = PT_CT;
                            P.cone_ok = dalloc<uint32_t>(c, (size_t)c->plan_cap);
                            P.cone_hdr = dalloc<uint32_t>(c, (size_t)c->plan_cap * tiles * CONE_HDRW);
                            P.cone_pairs = dalloc<uint32_t>(c, (size_t)c->plan_cap * tiles * (CONE_LEVELS * 64) + 1024);   // (+: whole 1 KB pieces are fetched)
                            HIPCHK(hipMemset((void*)P.cone_ok, 0, (size_t)c->plan_cap * 4));
                        }
                        if (P.cone_ct == PT_CT && (size_t)P.cone_tiles == tiles) {
                            P.cone_gather = dalloc<uint16_t>(c, (size_t)c->plan_cap * tiles * CONE_GCAP + 512);
                            const PrWin WL = pr_win_layout(Ng, P.RW, 1, (int)tiles);
                            c->prw = dalloc<unsigned char>(c, WL.total);
                            HIPCHK(hipMemset(c->prw, 0, WL.total));
                            c->persist = true; c->persist_tile = true; c->persist_wide = true;
                            if (!c->inline_walk) c->defer_resolve = true;   // (the exchange of an iteration is left to the next launch: it may be this kernel's)
                            if (const char* rk = SMM_HOOK("SMMHIP_PR_RING")) { const int k = atoi(rk); if (k == 2 || k == 4) c->pr_ring_k = k; }
                            if (const char* st = SMM_HOOK("SMMHIP_PR_SLOW_TILE")) c->pr_slow_tile = atoi(st);
                            if (const char* su = SMM_HOOK("SMMHIP_PR_SLOW_US")) c->pr_slow_ticks = 100 * atoi(su);
                        }
                    }
                }
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
        if (N > 0 && Ng % N == 0 && opts->chain_offset % N == 0) {   // equal shards: the values form of the sharded exchange is available
            c->a2a_G = Ng / N;
            const char* ce = SMM_HOOK("SMMHIP_A2A_CAP");   // test hook: a small capacity
            c->a2a_cap = ce ? std::max(1, atoi(ce)) : a2a_capacity(N, c->a2a_G);
            c->a2a_send_idx = dalloc<int32_t>(c, (size_t)c->a2a_G * c->a2a_cap);
            c->a2a_send_cnt = dalloc<int32_t>(c, (size_t)c->a2a_G);
            c->a2a_rowidx = dalloc<int32_t>(c, (size_t)N);
        }
        for (int b = 0; b < 2; ++b) c->vals_buf[b] = dalloc<double>(c, (size_t)N + 4);   // (+4: read as 16-byte pieces)
        P.vals = c->vals_buf[0];
        if (!c->lds_exchange) {
            const int Kmax = std::max(K, 1);
            P.xval = dalloc<double>(c, Ng); P.xnext = dalloc<int32_t>(c, Ng); P.xpairs = dalloc<int32_t>(c, (size_t)Kmax * 2);
            P.xsrc = dalloc<int32_t>(c, Ng); P.xpartner = dalloc<int32_t>(c, Ng);
            P.xslot = dalloc<double>(c, (size_t)Ng * 2);
        }
        {   // history: NaN values, curr/best = Inf, best_id = -1, exchanged = accepted = status = 0
            std::vector<double> row((size_t)N * P.HW, NAN);
            for (int i = 0; i < N; ++i) {
                double* h = row.data() + (size_t)i * P.HW;
                h[H_CURR] = INFINITY; h[H_BEST] = INFINITY; h[H_BESTID] = -1.0; h[H_EXCH] = 0.0; h[H_ACC] = 0.0; h[H_STATUS] = 0.0;
            }
            P.hrec = dalloc<double>(c, TN * P.HW);
            // (one row from the host, then doubling copies on the device: a long history — bench.py's repetitions hold 50 000 iterations — is filled
            // at HBM speed instead of row by row over PCIe)
            HIPCHK(hipMemcpy(P.hrec, row.data(), row.size() * 8, hipMemcpyHostToDevice));
            for (size_t have = 1; have < (size_t)T; have *= 2) {
                const size_t n = std::min(have, (size_t)T - have);
                HIPCHK(hipMemcpy(P.hrec + have * N * P.HW, P.hrec, n * N * P.HW * 8, hipMemcpyDeviceToDevice));
            }
        }
        P.err = dalloc<unsigned long long>(c, 1);
        {
            const unsigned long long e = ERR_NONE;
            HIPCHK(hipMemcpy(P.err, &e, 8, hipMemcpyHostToDevice));
        }
        if (c->persist_user) {
            const size_t smem = persist_gen_smem_bytes(Ng, np, P.RW, P.HW) + persist_gen_user_bytes();
            (void)hipFuncSetAttribute((const void*)c->upfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);   // (a module's function: where the runtime takes it)
            (void)hipGetLastError();
            int per_cu = 0, cus = 0;
            HIPCHK(hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, c->upfn, 1024, smem));
            HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
            if (N / PG_CT > per_cu * cus) { c->persist = false; c->persist_gen = false; c->persist_user = false; c->defer_resolve = false; }
        } else if (c->persist_gen) {
            const size_t smem = persist_gen_smem_bytes(Ng, np, P.RW, P.HW);
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_persist_gen, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            int per_cu = 0, cus = 0;
            HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_chain_persist_gen, 1024, smem));
            HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
            if (N / PG_CT > per_cu * cus) { c->persist = false; c->persist_gen = false; }
        } else if (c->persist_loc) {
            const size_t smem = persist_loc_smem_bytes(np);
            const void* fn = persist_loc_fn(np, c->persist_wide, c->persist_sh, P.mi_pct != 0);
            HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            int per_cu = 0, cus = 0;
            HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, NORM_WG, smem));
            HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
            c->persist_max_tiles = per_cu * cus;
            if ((N + NORM_CT - 1) / NORM_CT > per_cu * cus) { c->persist = false; c->persist_loc = false; c->persist_sh = false; }
        }
        if (c->persist_tile) {
            const int kind = obj_kind(c->obj);
            const size_t smem = persist_tile_smem(c);
            int per_cu = 0, cus = 0;
            if (c->utfn) {
                (void)hipFuncSetAttribute((const void*)c->utfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);   // (a module's function: not every runtime takes it this way; the launch asks for what it needs)
                (void)hipGetLastError();
                HIPCHK(hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, c->utfn, WG, smem));
            } else {
                const void* fn = P.mi_pct ? (kind == 2 ? (const void*)k_chain_persist_tile<2, true> : (const void*)k_chain_persist_tile<1, true>)
                                          : (kind == 2 ? (const void*)k_chain_persist_tile<2> : (const void*)k_chain_persist_tile<1>);
                HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
                HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, WG, smem));
            }
            HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
            c->persist_max_tiles = per_cu * cus;
            if ((N + PT_CT - 1) / PT_CT > per_cu * cus) { c->persist = false; c->persist_tile = false; c->defer_resolve = false; }
        }
        if (c->persist) {
            c->snap_cs = dalloc<double>(c, (size_t)N * CSW);
            c->snap_rec = dalloc<double>(c, (size_t)N * P.RW);
            for (int b = 0; b < 2; ++b) { c->snap_vals[b] = dalloc<double>(c, (size_t)N + 4); c->snap_slot8[b] = dalloc<uint2>(c, (size_t)N + 4 + 128); }
            c->snap_xres = dalloc<unsigned long long>(c, Ng);
            c->hist_fill = dalloc<double>(c, (size_t)N * P.HW);
            HIPCHK(hipMemcpy(c->hist_fill, P.hrec, (size_t)N * P.HW * 8, hipMemcpyDeviceToDevice));   // (a row of the constructor's fill)
        }
        if (c->big_exchange)
            HIPCHK(hipFuncSetAttribute((const void*)k_exch_plan_big, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan_big_lds_bytes(65535)));
        if (c->key_exchange)
        {
            HIPCHK(hipFuncSetAttribute((const void*)k_exch_resolve_key<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)resolve_key_bytes(XKEY_MAX, XKEY_MAX)));
            HIPCHK(hipFuncSetAttribute((const void*)k_exch_resolve_key<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)resolve_key_bytes(XKEY_PARTNER_MAX, XKEY_PARTNER_MAX)));
            HIPCHK(hipFuncSetAttribute((const void*)k_exch_resolve_rows<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)resolve_rows_bytes(XKEY_MAX, XKEY_MAX, XROWS_MAX)));
            HIPCHK(hipFuncSetAttribute((const void*)k_exch_resolve_rows<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)resolve_rows_bytes(XKEY_PARTNER_MAX, XKEY_PARTNER_MAX, XROWS_MAX)));
            HIPCHK(hipFuncSetAttribute((const void*)k_exch_resolve_rows<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)resolve_rows_bytes(XKEY_MAX, XKEY_MAX, XROWS_MAX)));
            HIPCHK(hipFuncSetAttribute((const void*)k_exch_resolve_rows<false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
            HIPCHK(hipFuncSetAttribute((const void*)k_exch_resolve_rows<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)resolve_rows_bytes(XKEY_PARTNER_MAX, XKEY_PARTNER_MAX, XROWS_MAX)));
        }
        if (c->lds_exchange) {
#ifdef SMM_TEST_HOOKS
            HIPCHK(hipFuncSetAttribute((const void*)k_exch_resolve_lds, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)resolve_lds_bytes(XLDS_MAX)));
#endif
            HIPCHK(hipFuncSetAttribute((const void*)k_exch_plan, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)std::max(plan_lds_bytes(XLDS_MAX, XLDS_MAX), plan_cone_bytes())));
#ifdef SMM_TEST_HOOKS
            HIPCHK(hipFuncSetAttribute((const void*)k_exch_resolve_lvl<256>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)resolve_lvl_bytes(XLVL_MAX, XLVL_MAX)));
            HIPCHK(hipFuncSetAttribute((const void*)k_exch_resolve_lvl<512>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)resolve_lvl_bytes(XLVL_MAX, XLVL_MAX)));
#endif
            HIPCHK(hipFuncSetAttribute((const void*)k_exch_resolve_lvl_soa<1024>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)resolve_lvl_soa_bytes(XLDS_MAX, XLDS_MAX)));
            HIPCHK(hipFuncSetAttribute((const void*)k_exch_resolve_lvl<1024>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)resolve_lvl_bytes(XLVL_MAX, XLVL_MAX)));
            HIPCHK(hipFuncSetAttribute((const void*)k_exch_resolve_lean, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024));
        }
        {   // tiles of problems with many parameters need more than the default 64 KiB of dynamic LDS
            const int lim = 160 * 1024;
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter<1, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter<1, 8, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter<2, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter<0, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter<0, 8, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter<0, 16, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter<1, 8, 1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter<2, 16, 1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter<0, 8, 1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter_norm<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter_norm<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter_norm<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter_norm<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter_norm<3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter_norm<3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter_norm<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter_norm<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter_norm_narrow<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter_norm_narrow<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter_norm_narrow<3>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter_norm_narrow<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter_norm_narrow_cone<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter_norm_narrow_cone<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter_norm_narrow_cone<3>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter_norm_narrow_cone<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            if (c->cone_big) HIPCHK(hipFuncSetAttribute((const void*)k_cone_chains, hipFuncAttributeMaxDynamicSharedMemorySize, Ng * 4));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter_norm_wide<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter_norm_wide<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter_norm_wide<3>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter_norm_wide<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter_norm_any<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter_norm_any<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter_norm_any<3>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter_norm_any<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter_norm_p2p_rows_narrow<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter_norm_p2p_rows_narrow<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter_norm_p2p_rows_narrow<3>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter_norm_p2p_rows_narrow<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter_norm_p2p_rows<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter_norm_p2p_rows<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter_norm_p2p_rows<3>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter_norm_p2p_rows<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter_norm_p2p<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter_norm_p2p<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter_norm_p2p<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter_norm_p2p<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter_norm_p2p<3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter_norm_p2p<3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter_norm_p2p<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_chain_iter_norm_p2p<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_eval_batch<1, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_eval_batch<2, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute((const void*)k_eval_batch<0, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            if (tile_smem(c, is_sim(c->obj) ? c->ct : (c->obj == SMM_OBJ_DENSE ? 16 : 8)) > (size_t)lim)
                throw std::string("tile does not fit the 160 KiB LDS");
        }
        c->xk = (int)choose_exchange(c);
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
        return told(c, check_device_error(c));
    } catch (const std::string& m) {
        return fail(c, SMM_ERR_HIP, m);
    }
}

int smm_bgp_step_async(void* ctx, int32_t n_iters) {
    Ctx* c = (Ctx*)ctx;
    if (!c || n_iters < 0) return SMM_ERR_INVALID_ARG;
    if (c->failed) return told(c);
    if (c->P.N != c->P.Ng) return fail(c, SMM_ERR_STATE, "smm_bgp_step needs a single shard (N == N_global); use the sharded calls");
    if (c->iter + n_iters > c->P.T) return fail(c, SMM_ERR_MAXITER, "step beyond maxiter (history capacity)");
    if (c->rec_external) return fail(c, SMM_ERR_STATE, "records are in the gather buffer: call smm_bgp_sharded_finish first");
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
        enqueue_iterations(c, n_iters);
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
    if (c->failed) return told(c);
    if (c->rec_external) return fail(c, SMM_ERR_STATE, "records are in the gather buffer: call smm_bgp_sharded_finish first");
    if (c->a2a_open) return fail(c, SMM_ERR_STATE, "smm_bgp_a2a_pack_dev without smm_bgp_a2a_apply_dev: the exchange of this iteration would be dropped");
    if (c->iter + 1 > c->P.T) return fail(c, SMM_ERR_MAXITER, "step beyond maxiter (history capacity)");
    try {
        HIPCHK(hipSetDevice(c->device));
        settle_persist(c);
        if (c->failed) return told(c);
        const int t = c->iter + 1;
        // smm_bgp_step leaves the exchange of its last iteration to the next chain kernel (inline walk): the three-phase
        // form reads P.xres, so resolve it now (local_step after step read an unresolved xres)
        resolve_now(c);
        ensure_windows(c, t);
        const int flags = (c->prev_open ? F_CLOSE_PREV : 0) | (c->pending ? F_HAS_PENDING : 0);
        launch_chain_iter(c, t, flags);
        HIPCHK(hipGetLastError());
        c->prev_open = true;
        c->pending = false;
        c->iter = t; c->exch_done = false;
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
    if (c->failed) return told(c);
    if (c->iter + 1 > c->P.T) return fail(c, SMM_ERR_MAXITER, "step beyond maxiter (history capacity)");
    if (c->rec_external && !gathered_prev_dev) return fail(c, SMM_ERR_INVALID_ARG, "gathered_prev required: the last records live there");
    if (c->unresolved) return fail(c, SMM_ERR_STATE, "mixing smm_bgp_step and smm_bgp_sharded_step without a flush");
    if (c && c->p2p_current) return fail(c, SMM_ERR_STATE, "the records of the last iteration are in the p2p windows: call smm_bgp_p2p_finish first");
    try {
        HIPCHK(hipSetDevice(c->device));
        settle_persist(c);
        if (c->failed) return told(c);
        const int t = c->iter + 1;
        const KParams& P = c->P;
        int flags = (c->prev_open ? F_CLOSE_PREV : 0);
        // profiling mode 2: this call is one more iteration of the "step" smm_sync sums up (the kernels' own begin/end stamps)
        const bool prof = c->profiling == 2;
        const int it = c->pev_iters;
        if (prof) {
            while ((int)c->pev.size() < 4 * (it + 1)) {
                hipEvent_t e;
                HIPCHK(hipEventCreate(&e));
                c->pev.push_back(e);
            }
            c->pev_exch.resize((size_t)it + 1, 0);
            c->pev_exch[it] = 0;
            if (it == 0) HIPCHK(hipEventRecord(c->ev0, c->stream));
        }
        if (c->rec_external) {
            if (c->pending_ext) {   // exchangeMoves! of iteration t-1 over the gathered records (before its plan window can move on)
                if (prof && c->lean_resolve) { c->kev0 = c->pev[4 * it + 2]; c->kev1 = c->pev[4 * it + 3]; c->pev_exch[it] = 1; }
                launch_resolve(c, t - 1, (const double*)gathered_prev_dev);
                c->kev0 = c->kev1 = nullptr;
                flags |= F_HAS_PENDING;
            }
            flags |= F_GLOBAL_REC;
            c->ext_rec_in = (const double*)gathered_prev_dev;
        } else if (c->pending) {
            flags |= F_HAS_PENDING;   // resolved earlier through the three-phase calls
        }
        ensure_windows(c, t);
        c->ext_rec_out = (double*)gathered_next_dev + (size_t)P.offset * P.RW;
        if (prof) { c->kev0 = c->pev[4 * it]; c->kev1 = c->pev[4 * it + 1]; }
        launch_chain_iter(c, t, flags);
        c->kev0 = c->kev1 = nullptr;
        c->ext_rec_in = nullptr; c->ext_rec_out = nullptr;
        if (prof) {
            HIPCHK(hipEventRecord(c->ev1, c->stream));
            c->pev_iters = it + 1;
            c->pending_timing = true;
            c->timing.iters = it + 1;
            c->timing.chain_evals = (int64_t)(it + 1) * P.N;
        }
        HIPCHK(hipGetLastError());
        c->prev_open = true;
        c->pending = false;
        c->rec_external = true;
        c->pending_ext = exchange_active(c, t);
        c->iter = t; c->exch_done = false;
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
    if (c && c->p2p_current) return fail(c, SMM_ERR_STATE, "the records of the last iteration are in the p2p windows: call smm_bgp_p2p_finish first");
    try {
        HIPCHK(hipSetDevice(c->device));
        settle_persist(c);
        if (c->failed) return told(c);
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

int smm_bgp_p2p_init(void* ctx, void* ipc_handle_out, void** window_dev_out) {
    Ctx* c = (Ctx*)ctx;
    if (!c) return SMM_ERR_INVALID_ARG;
    if (c->a2a_G < 1) return fail(c, SMM_ERR_STATE, "the p2p form needs equal shards (N_global a multiple of N, chain_offset a multiple of N)");
    if (c->a2a_G > P2P_MAXG) return fail(c, SMM_ERR_INVALID_ARG, "the p2p form serves up to 8 ranks (one node)");
    try {
        HIPCHK(hipSetDevice(c->device));
        KParams& P = c->P;
        const P2PLayout L = p2p_layout(P.Ng, P.RW);
        if (!c->p2p_mine) {
            // Plain device memory — what the receive buffers of the collective libraries are.  (An UNCACHED allocation,
            // hipDeviceMallocUncached, looked like the natural choice and was the first one: it passed every test of its own file and
            // failed in the full suite, in both forms, whenever the process had run other contexts before — later kernels were served
            // lines that an earlier allocation at the same address had left behind; the cache maintenance at kernel boundaries, which
            // plain memory gets, does not seem to cover it.  Within a launch nothing relies on the caches: stores into a window are
            // system-scope stores, self-validating words are re-read past the caches until their tag is the wanted one.)
            // (a shard that can run the persistent form keeps its ring behind the p2p window proper: one allocation, one IPC handle)
            size_t total = L.total;
            if (c->persist_sh) {
                c->prw_off = (L.total + 255) & ~(size_t)255;
                total = c->prw_off + pr_win_layout(P.Ng, P.RW, c->a2a_G, (P.N + NORM_CT - 1) / NORM_CT).total;
            }
            void* w = nullptr;
            HIPCHK(hipMalloc(&w, total));
            c->p2p_mine = (unsigned char*)w;
            HIPCHK(hipMemset(w, 0, total));
            HIPCHK(hipDeviceSynchronize());
            P.p2p_G = c->a2a_G;
            P.p2p_rank = P.offset / P.N;
            for (int r = 0; r < P2P_MAXG; ++r) P.p2p_win[r] = nullptr;
            P.p2p_win[P.p2p_rank] = c->p2p_mine;
            P.p2p_self = c->p2p_mine;
            for (int b = 0; b < 2; ++b) {
                P.p2p_off[b] = (uint32_t)L.rec[b]; P.p2p_off[2 + b] = (uint32_t)L.val[b]; P.p2p_off[4 + b] = (uint32_t)L.slot[b];
                P.p2p_off[6 + b] = (uint32_t)L.llrec[b]; P.p2p_off[8 + b] = (uint32_t)L.llval[b];
                P.p2p_off[10 + b] = (uint32_t)L.slot4[b];
            }
            if (L.total >= ((size_t)1 << 32)) throw std::string("p2p window larger than 4 GiB");
            c->p2p_attached = 1u << P.p2p_rank;
            c->p2p_seq = 0;
            c->p2p_current = false;
            // one launch per iteration (the lean key walk in the chain kernel's prologue, the push in its epilogue) where the
            // single shard has it too: objfunc_norm with np == nm <= 4, min_improve == 0, N_global <= 8192
            c->p2p_inline = c->norm_fast && c->win_lv_pairs_p && !P.lean_wide && P.Ng <= XLDS_MAX && P.plan_K <= XLDS_MAX && !c->deep_plan &&
                            P.dist_fun == SMM_DIST_MINUS && p2p_walk_bytes(c) + norm_tile_doubles(P.np) * 8 <= (size_t)160 * 1024;
            // two launches per iteration for the large norm populations (8192 < N_global <= 32768, min_improve == 0: BASELINE
            // configs[2], 4 and 8 shards of 4096): k_exch_resolve_rows reads the window's tagged slots itself, the chain kernel
            // pushes from its epilogue like the inline form's
            c->p2p_rows = c->norm_fast && !c->p2p_inline && c->xk == XK_ROWS;
        }
        if (ipc_handle_out) {
            hipIpcMemHandle_t h;
            HIPCHK(hipIpcGetMemHandle(&h, c->p2p_mine));
            static_assert(sizeof(hipIpcMemHandle_t) == SMM_P2P_HANDLE_BYTES, "hipIpcMemHandle_t is 64 bytes");
            memcpy(ipc_handle_out, &h, sizeof h);
        }
        if (window_dev_out) *window_dev_out = c->p2p_mine;
    } catch (const std::string& m) {
        return fail(c, SMM_ERR_HIP, m);
    }
    return SMM_OK;
}

int smm_bgp_p2p_attach(void* ctx, int32_t rank, const void* ipc_handle, void* window_dev) {
    Ctx* c = (Ctx*)ctx;
    if (!c || (!ipc_handle) == (!window_dev)) return SMM_ERR_INVALID_ARG;
    if (!c->p2p_mine) return fail(c, SMM_ERR_STATE, "smm_bgp_p2p_init comes first");
    if (rank < 0 || rank >= c->P.p2p_G || rank == c->P.p2p_rank) return fail(c, SMM_ERR_INVALID_ARG, "smm_bgp_p2p_attach: rank of ANOTHER shard, 0 <= rank < N_global / N");
    if (c->p2p_attached & (1u << rank)) return fail(c, SMM_ERR_STATE, "smm_bgp_p2p_attach: this rank's window is attached already");
    try {
        HIPCHK(hipSetDevice(c->device));
        void* w = window_dev;
        if (ipc_handle) {
            hipIpcMemHandle_t h;
            memcpy(&h, ipc_handle, sizeof h);
            HIPCHK(hipIpcOpenMemHandle(&w, h, hipIpcMemLazyEnablePeerAccess));
            c->p2p_opened[rank] = w;
            // (a peer PROCESS on this very device — several ranks on one GPU, the tests' way: its tiles compete with this rank's for
            // the compute units, and the persistent form needs all of them resident at once)
            hipPointerAttribute_t at;
            if (hipPointerGetAttributes(&at, w) == hipSuccess && at.device == c->device) c->p2p_ranks_here += 1;
            (void)hipGetLastError();
        } else {   // a context of this process: on another device the two must see each other
            hipPointerAttribute_t at;
            HIPCHK(hipPointerGetAttributes(&at, w));
            if (at.device == c->device) c->p2p_ranks_here += 1;
            if (at.device != c->device) {
                const hipError_t e = hipDeviceEnablePeerAccess(at.device, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) HIPCHK(e);
                (void)hipGetLastError();
            }
        }
        c->P.p2p_win[rank] = (unsigned char*)w;
        c->p2p_attached |= 1u << rank;
    } catch (const std::string& m) {
        return fail(c, SMM_ERR_HIP, m);
    }
    return SMM_OK;
}

int smm_bgp_p2p_step(void* ctx, int32_t n_iters) {
    Ctx* c = (Ctx*)ctx;
    if (!c || n_iters < 0) return SMM_ERR_INVALID_ARG;
    if (c->failed) return told(c);
    if (!c->p2p_mine) return fail(c, SMM_ERR_STATE, "smm_bgp_p2p_init comes first");
    if (c->p2p_attached != (1u << c->P.p2p_G) - 1u) return fail(c, SMM_ERR_STATE, "smm_bgp_p2p_step: not every rank's window is attached");
    if (c->iter + n_iters > c->P.T) return fail(c, SMM_ERR_MAXITER, "step beyond maxiter (history capacity)");
    if (c->rec_external && !c->p2p_current) return fail(c, SMM_ERR_STATE, "records are in the gather buffer: call smm_bgp_sharded_finish first");
    if (c->a2a_open) return fail(c, SMM_ERR_STATE, "smm_bgp_a2a_apply_dev comes first");
    try {
        HIPCHK(hipSetDevice(c->device));
        // (launches of the persistent form that this step simply continues are looked at — and agreed upon by the ranks — when the run is
        // synchronised, not between two steps: the host stays out of the way)
        if (!(!c->p2p_current && persist_sh_usable(c, n_iters))) settle_persist(c);
        if (c->failed) return told(c);
        KParams& P = c->P;
        if (c->profiling == 2) {
            while ((int)c->pev.size() < 4 * n_iters) {
                hipEvent_t e;
                HIPCHK(hipEventCreate(&e));
                c->pev.push_back(e);
            }
            c->pev_exch.assign((size_t)n_iters, 0);
        }
        c->pev_iters = 0;
        HIPCHK(hipEventRecord(c->ev0, c->stream));
        p2p_enqueue(c, n_iters);
        c->pending_timing = true;
        c->timing.iters = n_iters;
        c->timing.chain_evals = (int64_t)n_iters * P.N;
    } catch (const std::string& m) {
        c->kev0 = c->kev1 = nullptr; c->ext_rec_in = nullptr; c->ext_rec_out = nullptr; c->ext_vals_out = nullptr;
        return fail(c, SMM_ERR_HIP, m);
    }
    return SMM_OK;
}

// settle the last p2p step (its exchange, history, counters) into the context, like smm_bgp_sharded_finish
int smm_bgp_p2p_finish(void* ctx) {
    Ctx* c = (Ctx*)ctx;
    if (!c) return SMM_ERR_INVALID_ARG;
    if (c->p2p_current && !c->rec_external) return SMM_OK;
    if (!c->p2p_current && !(c->p2p_mine && c->unresolved && c->P.N < c->P.Ng)) return SMM_OK;
    try {
        HIPCHK(hipSetDevice(c->device));
        settle_persist(c);
        if (c->failed) return told(c);
        p2p_publish(c);   // (behind a launch of the persistent form: the records after `iter` into the windows, their exchange still to be resolved)
        const KParams& P = c->P;
        const P2PLayout L = p2p_layout(P.Ng, P.RW);
        int flags = (c->prev_open ? F_CLOSE_PREV : 0) | F_GLOBAL_REC;
        // the donors' records of the last iteration must have landed, in plain form
        if (c->p2p_mode_inline) launch_p2p_unpack(c, c->iter);
        else if (c->p2p_unwaited) launch_p2p_wait(c);
        if (c->pending_ext) {
            launch_resolve_window(c, c->iter);
            flags |= F_HAS_PENDING;
        }
        hipLaunchKernelGGL(k_flush, dim3((P.N + 255) / 256), dim3(256), 0, c->stream, P, c->iter + 1,
                           (const double*)(c->p2p_mine + L.rec[c->iter & 1]), c->rec[c->cur ^ 1], flags);
        HIPCHK(hipGetLastError());
        c->cur ^= 1;
        c->pending = false; c->prev_open = false; c->rec_external = false; c->pending_ext = false; c->p2p_current = false;
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
    if (c->failed) return told(c);
    if (c->rec_external) return fail(c, SMM_ERR_STATE, "records are in the gather buffer: call smm_bgp_sharded_finish first");
    try {
        HIPCHK(hipSetDevice(c->device));
        settle_persist(c);
        if (c->failed) return told(c);
        // after smm_bgp_step the exchange of the last iteration is still pending (resolved or not): the exported records must
        // be the ones AFTER that exchange, as the three-phase protocol defines them
        if (c->pending || c->unresolved) flush(c);
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
    if (c->failed) return told(c);
    if (c->iter < 1) return fail(c, SMM_ERR_STATE, "exchange before the first local step");
    if (c->pending || c->exch_done || c->a2a_open) return fail(c, SMM_ERR_STATE, "exchange already resolved for this iteration");
    try {
        HIPCHK(hipSetDevice(c->device));
        settle_persist(c);
        if (c->failed) return told(c);
        c->exch_done = true;
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

// ---- the values form of the sharded exchange (include/smmhip.h) ----
int smm_bgp_a2a_capacity(void* ctx) {
    Ctx* c = (Ctx*)ctx;
    if (!c) return SMM_ERR_INVALID_ARG;
    return c->a2a_cap;
}

int smm_bgp_export_values_dev(void* ctx, void* vals_dev) {
    Ctx* c = (Ctx*)ctx;
    if (!c || !vals_dev) return SMM_ERR_INVALID_ARG;
    if (c->failed) return told(c);
    if (c->rec_external) return fail(c, SMM_ERR_STATE, "records are in the gather buffer: call smm_bgp_sharded_finish first");
    if (c->iter < 1) return fail(c, SMM_ERR_STATE, "no iteration yet");
    try {
        HIPCHK(hipSetDevice(c->device));
        settle_persist(c);
        if (c->failed) return told(c);
        if (c->pending || c->unresolved) flush(c);
        HIPCHK(hipMemcpyAsync(vals_dev, c->vals_buf[c->iter & 1], (size_t)c->P.N * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    } catch (const std::string& m) {
        return fail(c, SMM_ERR_HIP, m);
    }
    return SMM_OK;
}

int smm_bgp_a2a_pack_dev(void* ctx, const void* vals_all_dev, void* send_dev) {
    Ctx* c = (Ctx*)ctx;
    if (!c || !vals_all_dev || !send_dev) return SMM_ERR_INVALID_ARG;
    if (c->failed) return told(c);
    if (c->a2a_cap <= 0) return fail(c, SMM_ERR_STATE, "the values form needs equal shards (N_global a multiple of N, chain_offset a multiple of N)");
    if (c->iter < 1) return fail(c, SMM_ERR_STATE, "exchange before the first local step");
    if (c->pending || c->a2a_open || c->exch_done) return fail(c, SMM_ERR_STATE, "exchange already resolved for this iteration");
    try {
        HIPCHK(hipSetDevice(c->device));
        settle_persist(c);
        if (c->failed) return told(c);
        if (exchange_active(c, c->iter)) {
            KParams P1 = c->P;
            P1.RW = 1;   // (the resolve kernels read value s at gathered[s * RW])
            launch_resolve_p(c, P1, c->iter, (const double*)vals_all_dev);
            const KParams& P = c->P;
            hipLaunchKernelGGL(k_a2a_index, dim3(c->a2a_G), dim3(XWG), 0, c->stream, P, c->iter, c->a2a_G, c->a2a_cap, c->a2a_send_idx,
                               c->a2a_send_cnt, c->a2a_rowidx);
            const size_t ne = (size_t)c->a2a_G * c->a2a_cap * P.RW;
            hipLaunchKernelGGL(k_a2a_pack, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, c->stream, P, c->a2a_G, c->a2a_cap,
                               (const int32_t*)c->a2a_send_idx, (const int32_t*)c->a2a_send_cnt, (const double*)c->rec[c->cur], (double*)send_dev);
        }
        HIPCHK(hipGetLastError());
        c->a2a_open = true;
    } catch (const std::string& m) {
        return fail(c, SMM_ERR_HIP, m);
    }
    return SMM_OK;
}

int smm_bgp_a2a_apply_dev(void* ctx, const void* recv_dev) {
    Ctx* c = (Ctx*)ctx;
    if (!c || !recv_dev) return SMM_ERR_INVALID_ARG;
    if (c->failed) return told(c);
    if (!c->a2a_open) return fail(c, SMM_ERR_STATE, "smm_bgp_a2a_pack_dev comes first");
    try {
        HIPCHK(hipSetDevice(c->device));
        settle_persist(c);
        if (c->failed) return told(c);
        if (exchange_active(c, c->iter)) {
            const KParams& P = c->P;
            hipLaunchKernelGGL(k_a2a_apply, dim3((P.N + 255) / 256), dim3(256), 0, c->stream, P, c->iter, (const double*)recv_dev,
                               (const int32_t*)c->a2a_rowidx, c->rec[c->cur]);
        }
        HIPCHK(hipGetLastError());
        c->a2a_open = false;
        c->exch_done = true;
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
        DevBuf<double> dp((size_t)P.np * M), dv((size_t)M), dm((size_t)P.nm * M);
        if (c->obj == SMM_OBJ_USER) {   // the user's kernel wants [M][np] / [M][nm]: transpose on the host
            std::vector<double> tp((size_t)M * P.np), tm((size_t)M * P.nm);
            std::vector<int> ts((size_t)M);
            for (int i = 0; i < M; ++i)
                for (int k = 0; k < P.np; ++k) tp[(size_t)i * P.np + k] = params[(size_t)k * M + i];
            DevBuf<int> dsi((size_t)M);
            HIPCHK(hipMemcpyAsync(dp.p, tp.data(), tp.size() * 8, hipMemcpyHostToDevice, c->stream));
            launch_user_kernel(c, dp.p, M, dm.p, dv.p, dsi.p);
            HIPCHK(hipMemcpyAsync(value, dv.p, (size_t)M * 8, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipMemcpyAsync(tm.data(), dm.p, tm.size() * 8, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipMemcpyAsync(ts.data(), dsi.p, ts.size() * sizeof(int), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            for (int i = 0; i < M; ++i) {
                status[i] = (int8_t)ts[i];
                for (int k = 0; k < P.nm; ++k) sim_moments[(size_t)k * M + i] = tm[(size_t)i * P.nm + k];
            }
            return SMM_OK;
        }
        DevBuf<int8_t> ds((size_t)M);
        HIPCHK(hipMemcpyAsync(dp.p, params, (size_t)P.np * M * 8, hipMemcpyHostToDevice, c->stream));
        constexpr int CT = 8;
        if (is_sim(c->obj))
            hipLaunchKernelGGL((k_eval_batch<1, CT>), dim3((M + CT - 1) / CT), dim3(WG), tile_smem_base(c, CT), c->stream, P, dp.p, M, dv.p,
                               dm.p, ds.p);
        else if (c->obj == SMM_OBJ_DENSE)
            hipLaunchKernelGGL((k_eval_batch<2, 16>), dim3((M + 15) / 16), dim3(WG), tile_smem_base(c, 16), c->stream, P, dp.p, M, dv.p, dm.p,
                               ds.p);
        else
            hipLaunchKernelGGL((k_eval_batch<0, CT>), dim3((M + CT - 1) / CT), dim3(WG), tile_smem_base(c, CT), c->stream, P, dp.p, M, dv.p,
                               dm.p, ds.p);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(value, dv.p, (size_t)M * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(sim_moments, dm.p, (size_t)P.nm * M * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(status, ds.p, (size_t)M, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    } catch (const std::string& m) {
        return fail(c, SMM_ERR_HIP, m);
    }
    return SMM_OK;
}

int smm_eval_batch_noseed(void* ctx, const double* params, int32_t M, uint64_t base_seed, double* value, double* sim_moments,
                          int8_t* status) {
    Ctx* c = (Ctx*)ctx;
    if (!c || !params || M < 0 || !value || !sim_moments || !status) return SMM_ERR_INVALID_ARG;
    if (!is_sim(c->obj)) return fail(c, SMM_ERR_INVALID_ARG, "noseed evaluations exist for objfunc_norm only");
    if (M == 0) return SMM_OK;
    try {
        HIPCHK(hipSetDevice(c->device));
        const KParams& P = c->P;
        DevBuf<double> dp((size_t)P.np * M), dv((size_t)M), dm((size_t)P.nm * M);
        DevBuf<int8_t> ds((size_t)M);
        HIPCHK(hipMemcpyAsync(dp.p, params, (size_t)P.np * M * 8, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_eval_batch_noseed, dim3(M), dim3(WG), (size_t)(WG / 64) * P.nm * 8, c->stream, P, (const double*)dp.p, M,
                           (uint64_t)base_seed, dv.p, dm.p, ds.p);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(value, dv.p, (size_t)M * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(sim_moments, dm.p, (size_t)P.nm * M * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(status, ds.p, (size_t)M, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
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
        settle_persist(c);
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
        settle_persist(c);
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
        settle_persist(c);
        HIPCHK(hipStreamSynchronize(c->stream));
        // a hard error nobody has been TOLD of yet (raised on the device by steps this caller enqueued and never synchronised; the state readers
        // do not raise): this call is the recovery from a failure, it must not be the place where one disappears — it reports it, once, and uploads
        // nothing; the caller's next smm_set_state goes through (tests/test_gpu_error_enumeration.py: every `... step ss` sequence lost its error here)
        (void)check_device_error(c);
        if (c->failed && !c->failed_told) { c->failed_told = true; return c->failed; }
        KParams& P = c->P;
        const size_t N = P.N, RW = P.RW, HW = P.HW, np = P.np, nm = P.nm;
        std::vector<double> cs(N * CSW), rec(N * RW, 0.0);
        HIPCHK(hipMemcpy(cs.data(), P.cs, cs.size() * 8, hipMemcpyDeviceToHost));  // keeps acc_tuner
        bool nan_seen = false;   // (NaN orders under no key: the lean walks are not launched from such a state, launch_chain_iter_norm)
        for (size_t i = 0; i < N; ++i) {
            double* b = cs.data() + i * CSW;
            double* r = rec.data() + i * RW;
            b[CS_SIGMA] = s->sigma[i]; b[CS_RATE] = s->accept_rate[i];
            b[CS_NNOEX] = (double)s->n_noex[i]; b[CS_NACC] = (double)s->n_acc_noex[i];
            b[CS_LACC] = 0.0; b[CS_WASX] = 0.0;
            b[CS_BEST] = s->best_val[i]; b[CS_BESTID] = (double)s->best_id[i];
            b[CS_BESTP] = s->best_val[i]; b[CS_BESTPID] = (double)s->best_id[i];
            r[0] = s->la_value[i]; r[1] = s->la_prob[i]; r[2] = (double)s->la_status[i];
            if (s->la_value[i] != s->la_value[i]) nan_seen = true;
            for (size_t k = 0; k < np; ++k) r[3 + k] = s->la_params[k * N + i];
            for (size_t k = 0; k < nm; ++k) r[3 + np + k] = s->la_sim_moments[k * N + i];
        }
        c->nan_values = nan_seen;
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
        if (c->failed) {   // a state from before the failure: the run may go on
            const unsigned long long e = ERR_NONE;
            HIPCHK(hipMemcpy(P.err, &e, 8, hipMemcpyHostToDevice));
            c->failed = 0; c->failed_told = false;
        }
        c->rec_external = false; c->pending_ext = false; c->unresolved = false;
        c->pending = false;
        c->prev_open = false;
        c->a2a_open = false;
        c->p2p_current = false;   // (the next smm_bgp_p2p_step publishes the uploaded state)
        c->exch_done = false;
        c->slots_iter = -1;
        if (P.walk_flags) HIPCHK(hipMemset(P.walk_flags, 0, 16));   // (the values the next exchange sees are written by the next accept step)
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

int smm_set_persistent(void* ctx, int32_t on) {
    Ctx* c = (Ctx*)ctx;
    if (!c) return SMM_ERR_INVALID_ARG;
    c->persist_on = on ? 1 : 0;
    return SMM_OK;
}

int smm_get_persistent(void* ctx, int32_t* available, int32_t* launches, int32_t* repairs) {
    Ctx* c = (Ctx*)ctx;
    if (!c) return SMM_ERR_INVALID_ARG;
    if (available) *available = (c->persist && c->persist_on && !c->persist_broken) ? 1 : 0;
    if (launches) *launches = c->persist_launches;
    if (repairs) *repairs = c->persist_repairs;
    return SMM_OK;
}

// which forms this context was given at creation (one line; tests/test_gpu_forms.py pins the table)
int smm_describe(void* ctx, char* out, int32_t cap) {
    Ctx* c = (Ctx*)ctx;
    if (!c || !out || cap < 1) return SMM_ERR_INVALID_ARG;
    const KParams& P = c->P;
    const char* chain;
    if (c->obj == SMM_OBJ_USER) chain = c->u_lanes ? "user_lanes_3launches" : "user_3launches";
    else if (c->norm_fast)
        chain = c->cone_big ? "iter_norm_narrow_cone" : (!c->win_lv_pairs_p || c->deep_plan || c->nan_values || P.mi_pct) && c->inline_walk ? "iter_norm_any"
              : (P.lean_wide && c->inline_walk) ? "iter_norm_wide" : c->norm_narrow ? "iter_norm_narrow" : "iter_norm";
    else chain = c->obj == SMM_OBJ_DENSE ? (P.dense_A2f ? "iter<dense2,16>" : "iter<dense,16>") : is_sim(c->obj) ? (c->tpw == 2 ? "iter<sim,8,2>" : "iter<sim,8>")
               : (c->gen_keys ? "iter<gen,16,2>" : c->tpw == 2 ? "iter<gen,8,2>" : "iter<gen,8>");
    static const char* xk[] = {"lean", "lvl", "lvl_soa", "tickets", "rows", "key", "lvl_big", "any"};
    const char* walk = c->cone_big ? "cone_local" : !c->inline_walk ? "standalone" : c->dense_keys ? "inline_keys_under_tile" : c->gen_keys ? (c->cone ? "inline_keys_cone" : "inline_keys")
                     : c->norm_fast ? ((P.lean_wide && !P.mi_pct) ? "inline_lean_wide" : (c->win_lv_pairs_p && !P.mi_pct) ? "inline_lean" : "inline_slots") : c->gen_lean ? "inline_lean16" : "inline_slots";
    const char* pers = !c->persist ? "none" : c->persist_loc ? (c->persist_sh ? (c->persist_sh_big ? (c->persist_wide ? "loc_wide_shard_bigplan" : "loc_shard_bigplan")
                                                                                                    : (c->persist_wide ? "loc_wide_shard" : "loc_shard"))
                                                                              : (c->persist_wide ? "loc_wide" : "loc"))
                     : c->persist_tile ? (c->utfn ? "tile_user" : c->obj == SMM_OBJ_DENSE ? (P.dense_A2f ? "tile_dense2" : "tile_dense") : "tile_sim") : c->persist_user ? "gen_user" : "gen";
    snprintf(out, (size_t)cap, "chain=%s walk=%s exchange=%s persistent=%s plan=%s window=%d", chain, walk, xk[c->xk], pers,
             c->big_exchange ? (c->plan_ahead ? "big_ahead" : "big") : c->lds_exchange ? "lds" : "none", c->plan_cap);
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

#ifdef SMM_TEST_HOOKS
// debug (test build only, not part of the public header): the look-ahead window starting at iteration t, planned now; milliseconds of its
// kernels on the stream (tools/exp/shard_plan_time.py: what a shard of 8 x 4096 spends on its plan per window)
int smm_debug_plan_window(void* ctx, int t, double* ms_out, int* window_out) {
    Ctx* c = (Ctx*)ctx;
    if (!c || !ms_out) return SMM_ERR_INVALID_ARG;
    try {
        HIPCHK(hipSetDevice(c->device));
        hipEvent_t e0, e1;
        HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
        c->plan_w = 0;
        HIPCHK(hipEventRecord(e0, c->stream));
        ensure_windows(c, t, false);
        HIPCHK(hipEventRecord(e1, c->stream));
        HIPCHK(hipEventSynchronize(e1));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        *ms_out = ms;
        if (window_out) *window_out = c->plan_w;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    } catch (const std::string& m) {
        return fail(c, SMM_ERR_HIP, m);
    }
    return SMM_OK;
}
// debug (test build only, not part of the public header): the cone of one tile in iteration plan_t0 + w of the current plan window
int smm_debug_cone(void* ctx, int w, int tile, uint32_t* hdr, uint32_t* pairs, uint16_t* gather, int32_t* info) {
    Ctx* c = (Ctx*)ctx;
    if (!c || !c->P.cone_hdr) return SMM_ERR_INVALID_ARG;
    const KParams& P = c->P;
    if (hipStreamSynchronize(c->stream) != hipSuccess) return SMM_ERR_HIP;
    if (hipMemcpy(hdr, P.cone_hdr + ((size_t)w * P.cone_tiles + tile) * CONE_HDRW, CONE_HDRW * 4, hipMemcpyDeviceToHost) != hipSuccess) return SMM_ERR_HIP;
    if (hipMemcpy(pairs, P.cone_pairs + ((size_t)w * P.cone_tiles + tile) * (CONE_LEVELS * 64), CONE_LEVELS * 64 * 4, hipMemcpyDeviceToHost) != hipSuccess) return SMM_ERR_HIP;
    if (P.cone_gather && hipMemcpy(gather, P.cone_gather + ((size_t)w * P.cone_tiles + tile) * CONE_GCAP, CONE_GCAP * 2, hipMemcpyDeviceToHost) != hipSuccess) return SMM_ERR_HIP;
    uint32_t ok = 0;
    if (hipMemcpy(&ok, P.cone_ok + w, 4, hipMemcpyDeviceToHost) != hipSuccess) return SMM_ERR_HIP;
    info[0] = c->plan_t0; info[1] = c->plan_w; info[2] = (int32_t)ok; info[3] = P.cone_ct;
    return SMM_OK;
}
#endif

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
