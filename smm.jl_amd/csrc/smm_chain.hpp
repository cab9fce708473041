// the chain kernel: tile reduction, simulation, objectives, block moves, inline exchange walk, k_chain_iter, k_flush, k_eval_batch — part of libsmmhip (included by smmhip.hip inside its anonymous namespace; gfx950 device code).
#pragma once
// ------------------------------------------------------------------------------------------
// Transposed wave reduction: every lane holds CT partial sums a[0..CT); on return lane l holds the
// 64-lane total of accumulator acc_index<CT>(l), combined by the canonical halving tree (offsets
// 32,16,8,4,2,1; IEEE addition is commutative so both partners compute the same bits).
// ------------------------------------------------------------------------------------------
// One step of the tree for two accumulators at once with gfx950's lane swaps: v_permlane32_swap exchanges the upper half of
// its first operand with the lower half of its second, v_permlane16_swap the odd rows of the first with the even rows of the
// second.  Afterwards, in the lanes whose bit OFF is clear, x = own x and y = the partner lane's x; in the other lanes
// x = the partner lane's y and y = own y: x + y is "mine + received" for the accumulator the lane keeps (3 instructions per
// pair of 64-bit accumulators instead of 4 selects, 2 LDS permutes and the add).
template <int OFF>
__device__ inline double swap_add(double x, double y) {
    static_assert(OFF == 32 || OFF == 16, "lane swaps exist for offsets 32 and 16");
    const unsigned long long ux = __builtin_bit_cast(unsigned long long, x), uy = __builtin_bit_cast(unsigned long long, y);
    unsigned xl, xh, yl, yh;
    if constexpr (OFF == 32) {
        const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)ux, (unsigned)uy, false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(ux >> 32), (unsigned)(uy >> 32), false, false);
        xl = lo[0]; yl = lo[1]; xh = hi[0]; yh = hi[1];
    } else {
        const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)ux, (unsigned)uy, false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(ux >> 32), (unsigned)(uy >> 32), false, false);
        xl = lo[0]; yl = lo[1]; xh = hi[0]; yh = hi[1];
    }
    const double x2 = __builtin_bit_cast(double, ((unsigned long long)xh << 32) | xl);
    const double y2 = __builtin_bit_cast(double, ((unsigned long long)yh << 32) | yl);
    return x2 + y2;
}
template <int CT, int NN, int OFF>
__device__ inline void wave_reduce_step(double (&a)[CT], int lane) {
    if constexpr (NN > 1) {
        if constexpr (OFF >= 16) {
#pragma unroll
            for (int i = 0; i < NN / 2; ++i) a[i] = swap_add<OFF>(a[i], a[i + NN / 2]);
        } else {
            const bool upper = (lane & OFF) != 0;
#pragma unroll
            for (int i = 0; i < NN / 2; ++i) {
                const double mine = upper ? a[i + NN / 2] : a[i];
                const double send = upper ? a[i] : a[i + NN / 2];
                const double recv = __shfl_xor(send, OFF, 64);
                a[i] = mine + recv;
            }
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
constexpr int SMM_SCOUT_AFTER = 2;   // rounds of one try per lane segment before the remaining tries of mysample are scouted (k_chain_iter)

// The shock matrix is read through a buffer descriptor: row = scalar byte offset (SALU), lane = one constant
// 32-bit vector offset, so a chunk load is ZU buffer_load instructions and no vector address arithmetic.
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
struct ZBuf {
    __amdgpu_buffer_rsrc_t rsrc;
    int lane_off;  // tid * 8
    __device__ inline void init_v(const double* Z, int nm, int zstride, int tid) {
        rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Z, 0, (int)((size_t)nm * zstride * sizeof(double)), 0x00020000);
        lane_off = tid * (int)sizeof(double);
    }
    __device__ inline void init(const KParams& P, int tid) { init_v(P.Z, P.nm, P.zstride, tid); }
};
// chunk ch of moment k
__device__ inline void sim_load_chunk_v(const ZBuf& zb, const int zstride, const bool dbg8, int k, int ch, double (&z)[ZU]) {
    const int row0 = (k * zstride + (dbg8 ? 0 : ch) * (ZU * WG)) * (int)sizeof(double);  // dbg 8: timing experiment
#pragma unroll
    for (int u = 0; u < ZU; ++u)
        z[u] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(zb.rsrc, zb.lane_off, row0 + u * WG * (int)sizeof(double), 0));
}
__device__ inline void sim_load_chunk(const ZBuf& zb, const KParams& P, int k, int ch, double (&z)[ZU]) { sim_load_chunk_v(zb, P.zstride, (P.dbg & 8) != 0, k, ch, z); }

// zc: chunk 0 of moment 0 (already loaded).  s_theta [CT][np], s_part [WG/64][CT][nm] in LDS.
// Moments are reduced in groups of G = 16/CT: one transposed reduction of G*CT accumulators has the
// same number of dependent shuffle steps as one of CT, so grouping halves that latency for CT = 8.
// A moment is nch chunks; the last one may be ragged (rows masked per lane).  Two chunks per trip, the
// two register buffers trade places; the chunk after a moment's last is chunk 0 of the next moment.
// (the problem's sizes by value: the persistent tile kernel has no KParams)
template <int CT>
__device__ inline void simulate_tile_v(const int ns, const int nm, const int np, const int zstride, const bool dbg8, const ZBuf& zb, const double* s_theta, double* s_part,
                                       int tid, double (&zc)[ZU]) {
    constexpr int G = (CT >= 16) ? 1 : 16 / CT;
    const int lane = tid & 63, wave = tid >> 6;
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
                for (int c = 0; c < CT; ++c) mu[c] = s_theta[c * np + k];
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
                    sim_load_chunk_v(zb, zstride, dbg8, k, ch + 1, zn);
                    add_full(zc);
                    const bool last = (ch + 2 == nch);
                    sim_load_chunk_v(zb, zstride, dbg8, last ? knext : k, last ? 0 : ch + 2, zc);
                    if (last) add_last(zn); else add_full(zn);
                }
                if (ch < nch) {  // odd count: the last chunk is in zc; afterwards the buffers are swapped by copy
                    sim_load_chunk_v(zb, zstride, dbg8, knext, 0, zn);
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

template <int CT>
__device__ inline void simulate_tile(const KParams& P, const ZBuf& zb, const double* s_theta, double* s_part, int tid, double (&zc)[ZU]) {
    simulate_tile_v<CT>(P.ns, P.nm, P.np, P.zstride, (P.dbg & 8) != 0, zb, s_theta, s_part, tid, zc);
}

// ------------------------------------------------------------------------------------------
// Dense objective (SMM_OBJ_DENSE, BASELINE config 5): the simulation is a dense contraction, so it
// runs on the FP64 matrix cores.  A tile is 16 chains = the N dimension of v_mfma_f64_16x16x4; wave w
// of the 8 owns the hidden units d in [32w, 32w+32):
//   x tile [16 d x 16 chains]  = B[16 d x np] * theta[np x 16]        (ceil(np/4) MFMAs)
//   h = tanh(x) (smm_tanh below): the accumulator layout (row = (lane>>4) + 4r, col = lane&15) IS the B-operand layout
//   of the next product (k = 4s + (lane>>4)), so h feeds the second GEMM from registers;
//   y tile [16 k x 16 chains] += A[16 k x 16 d] * h[16 d x 16]        (4 MFMAs per output tile)
// B and A are stored in fragment order (one coalesced 8-byte load per lane per MFMA).  The wave's
// partial y goes to LDS [w][k][chain]; the 8 partials are added left to right by the chain lane.
// ------------------------------------------------------------------------------------------
typedef double d4_t __attribute__((ext_vector_type(4)));
constexpr int DENSE_D = SMM_DENSE_D;

// the hidden layer's tanh (include/smmhip.h, SMM_OBJ_DENSE): ONE exponential and ONE division — E = exp(2|x|) = 2^n (1 + p) with p = expm1(r)
// on |r| <= ln2 / 2 (Taylor to r^13: 4e-18), tanh = (E - 1) / (E + 1) with E -+ 1 = fma(2^n, p, 2^n -+ 1) (2^n -+ 1 is exact) — about 45
// instructions against ocml's ~165 (tools/dense_bench.hip: 4.4 -> 1.4 us of a tile's evaluation); at most 3 ulp from the true value.  Only
// correctly rounded operations (fma, rint, ldexp, IEEE division), so a plain C restatement of the same expression is BIT-IDENTICAL (the tests hold one).
__device__ __forceinline__ double smm_tanh(const double x) {
    const double ax = __builtin_fabs(x);
    const double z = ax + ax;
    const double zc = z > 40.0 ? 40.0 : z;   // (tanh is 1 from 19.0625 on; a NaN stays one)
    const double n = __builtin_rint(zc * 1.44269504088896338700e+00);
    double r = __builtin_fma(-n, 6.93147180369123816490e-01, zc);
    r = __builtin_fma(-n, 1.90821492927058770002e-10, r);
    double q = 1.0 / 6227020800.0;
    q = __builtin_fma(q, r, 1.0 / 479001600.0);
    q = __builtin_fma(q, r, 1.0 / 39916800.0);
    q = __builtin_fma(q, r, 1.0 / 3628800.0);
    q = __builtin_fma(q, r, 1.0 / 362880.0);
    q = __builtin_fma(q, r, 1.0 / 40320.0);
    q = __builtin_fma(q, r, 1.0 / 5040.0);
    q = __builtin_fma(q, r, 1.0 / 720.0);
    q = __builtin_fma(q, r, 1.0 / 120.0);
    q = __builtin_fma(q, r, 1.0 / 24.0);
    q = __builtin_fma(q, r, 1.0 / 6.0);
    q = __builtin_fma(q, r, 0.5);
    const double p = __builtin_fma(r * r, q, r);
    const int ni = (n == n) ? (int)n : 0;
    const double s = __builtin_ldexp(1.0, ni);
    const double em1 = __builtin_fma(s, p, s - 1.0), ep1 = __builtin_fma(s, p, s + 1.0);
    // (1 from 19.0625 on: the select sits on the NUMERATOR — ep1 / ep1 is exactly 1 — so that the division stays unconditional: selecting between
    // 1.0 and the quotient made the compiler branch around the division: four branches per lane and tile in the middle of the matrix instructions)
    const double t = (ax >= 19.0625 ? ep1 : em1) / ep1;
    return __builtin_copysign(t, x);
}

// (inlined into its kernels — out of line the operands came through generic pointers, one dependent load per MFMA —; the operand
// fragments of a product are requested TOGETHER, ahead of the MFMAs that consume them: the B fragments of both hidden-unit tiles of the
// wave before the first product, the A fragments of a tile before its tanh — an MFMA never waits for a load of its own)
// OUT OF LINE, one copy for every kernel (its ~170 live registers — operand fragments of a whole tile, accumulators, tanh — are then
// its own, not the callers'); the tile's proposals and partial sums are named by their byte offsets in the dynamic LDS (a generic pointer
// would make every access a flat one).  NPS4: the products of the first GEMM in groups of four (np <= 16 NPS4) — straight-line code:
// fragments past the last parameter are read again from the last one (any finite value) and meet a zero of the proposal's side.
template <int CT, int NPS4>
__device__ __attribute__((noinline)) void dense_tile_n(const int np_, const int nOt_, const double* dense_Bf_, const double* dense_Af_,
                                                       const uint32_t theta_off_, const uint32_t part_off_, const int tid) {
    static_assert(CT == 16, "the dense objective tiles 16 chains (MFMA N dimension)");
    extern __shared__ __attribute__((aligned(16))) unsigned char dense_lds[];
    // (the arguments of an out-of-line function arrive in vector registers: what is uniform goes back into scalar ones, and the
    // matrices are GLOBAL memory, not "somewhere")
    typedef const __attribute__((address_space(1))) double* gptr_t;
    const int np = __builtin_amdgcn_readfirstlane(np_), nOt = __builtin_amdgcn_readfirstlane(nOt_);
    const uint32_t theta_off = (uint32_t)__builtin_amdgcn_readfirstlane((int)theta_off_), part_off = (uint32_t)__builtin_amdgcn_readfirstlane((int)part_off_);
    auto uniform_ptr = [](const double* p) {
        const unsigned long long u = (unsigned long long)p;
        return (gptr_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)u));
    };
    const gptr_t dense_Bf = uniform_ptr(dense_Bf_), dense_Af = uniform_ptr(dense_Af_);
    const double* s_theta = (const double*)(dense_lds + theta_off);
    double* s_part = (double*)(dense_lds + part_off);
    constexpr int PS = 4 * NPS4;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int nPs = (np + 3) / 4, nmp = nOt * 16;
    d4_t yacc[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) yacc[o] = d4_t{0.0, 0.0, 0.0, 0.0};
    double bfr[2][PS];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const gptr_t bf = dense_Bf + (size_t)(2 * wave + tt) * nPs * 64 + lane;
#pragma unroll
        for (int s = 0; s < PS; ++s) bfr[tt][s] = bf[(size_t)min(s, nPs - 1) * 64];
    }
    double th[PS];   // the proposal's components k = 4 s + lk of chain li (zero past the last parameter)
#pragma unroll
    for (int s = 0; s < PS; ++s) {
        const int p = 4 * s + lk;
        const double v = s_theta[li * np + min(p, np - 1)];
        th[s] = p < np ? v : 0.0;
    }
    // both first products ahead of the first tanh: the matrix pipe works on the second tile's product (and later on the first tile's second
    // product) while the wave's vector instructions evaluate the tanh; the A fragments of a tile are requested a product ahead of their use
    double afr[2][4][4];
    auto load_a = [&](const int tt) {
        const int T = 2 * wave + tt;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const gptr_t af = dense_Af + ((size_t)(min(o, nOt - 1) * (DENSE_D / 16) + T) * 4) * 64 + lane;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) afr[tt][o][s4] = af[s4 * 64];
        }
    };
    load_a(0);
    d4_t xacc[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        xacc[tt] = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < PS; ++s) xacc[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(bfr[tt][s], th[s], xacc[tt], 0, 0, 0);
        if (tt == 0) load_a(1);
    }
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        double h[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = smm_tanh(xacc[tt][r]);
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            if (o < nOt) {
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) yacc[o] = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[tt][o][s4], h[s4], yacc[o], 0, 0, 0);
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
// ------------------------------------------------------------------------------------------
// SMM_OBJ_DENSE2 (spec v2, BASELINE config 5 AS WORDED: a 256 x 256 matvec per evaluation): x = B theta, h1 = tanh x, g = A2 h1, h2 = tanh g,
// y = A h2.  The first and the last product are spec v1's; between them wave w owns the rows [32w, 32w + 32) of g — two row tiles, each ONE
// accumulator through 64 v_mfma_f64_16x16x4 (the contract's single fma chain over d = 0..255):
//   * h1 crosses the waves through LDS: [d][16 chains] = 32 KB in the region of the partial sums (free until the last product is done) —
//     the accumulator layout (row lk + 4r of tile T, chain li) lands at 256 T + 64 r + lane, and the B operand of k-step s (d = 4s + lk) is
//     read back at 64 s + lane: both conflict-free, one 8-byte LDS access per lane;
//   * A2 (512 KB, L2-resident; every tile streams all of it per evaluation: 134 MB of L2 reads per iteration at 256 tiles) lies in fragment
//     order [wave][k-step][lane][the wave's two row tiles]: ONE global_load_dwordx4 per lane feeds both MFMAs of a k-step; the loads run
//     D2_DEPTH k-steps ahead of the MFMAs that consume them (a register ring; the loop is straight-line code);
//   * two workgroup barriers: h1 complete before the first k-step; every wave done reading h1 before the partial sums overwrite it.
// 192 MFMAs per wave (32 + 128 + 32), 1536 per tile of 16 chains: 10.2 us of the matrix pipe at two waves per SIMD.
// ------------------------------------------------------------------------------------------
typedef double d2v_t __attribute__((ext_vector_type(2)));
constexpr int D2_DEPTH = 12;
template <int CT, int NPS4>
__device__ __attribute__((noinline)) void dense2_tile_n(const int np_, const int nOt_, const double* dense_Bf_, const double* dense_A2f_, const double* dense_Af_,
                                                        const uint32_t theta_off_, const uint32_t part_off_, const int tid) {
    static_assert(CT == 16, "the dense objective tiles 16 chains (MFMA N dimension)");
    extern __shared__ __attribute__((aligned(16))) unsigned char dense_lds[];
    typedef const __attribute__((address_space(1))) double* gptr_t;
    typedef const __attribute__((address_space(1))) d2v_t* g2ptr_t;
    const int np = __builtin_amdgcn_readfirstlane(np_), nOt = __builtin_amdgcn_readfirstlane(nOt_);
    const uint32_t theta_off = (uint32_t)__builtin_amdgcn_readfirstlane((int)theta_off_), part_off = (uint32_t)__builtin_amdgcn_readfirstlane((int)part_off_);
    auto uniform_ptr = [](const double* p) {
        const unsigned long long u = (unsigned long long)p;
        return (gptr_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)u));
    };
    const gptr_t dense_Bf = uniform_ptr(dense_Bf_), dense_Af = uniform_ptr(dense_Af_);
    const double* s_theta = (const double*)(dense_lds + theta_off);
    double* s_part = (double*)(dense_lds + part_off);
    double* s_h1 = s_part;   // [256 hidden units][16 chains]
    constexpr int PS = 4 * NPS4;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int nPs = (np + 3) / 4, nmp = nOt * 16;
    const g2ptr_t a2 = (g2ptr_t)uniform_ptr(dense_A2f_) + (size_t)wave * 64 * 64 + lane;
    // (the first product's operands are asked for first: the second product's ring is not looked at before the first barrier)
    double bfr[2][PS];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const gptr_t bf = dense_Bf + (size_t)(2 * wave + tt) * nPs * 64 + lane;
#pragma unroll
        for (int s = 0; s < PS; ++s) bfr[tt][s] = bf[(size_t)min(s, nPs - 1) * 64];
    }
    d2v_t ring[D2_DEPTH];
#pragma unroll
    for (int s = 0; s < D2_DEPTH; ++s) ring[s] = a2[s * 64];
    double th[PS];
#pragma unroll
    for (int s = 0; s < PS; ++s) {
        const int p = 4 * s + lk;
        const double v = s_theta[li * np + min(p, np - 1)];
        th[s] = p < np ? v : 0.0;
    }
    d4_t xacc[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        xacc[tt] = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < PS; ++s) xacc[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(bfr[tt][s], th[s], xacc[tt], 0, 0, 0);
    }
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) s_h1[256 * (2 * wave + tt) + 64 * r + lane] = smm_tanh(xacc[tt][r]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // h1 stands
    d4_t gacc[2] = {d4_t{0.0, 0.0, 0.0, 0.0}, d4_t{0.0, 0.0, 0.0, 0.0}};
    // the last product's operands (A's fragments of this wave's two column tiles, out of L2) are requested UNDER the second product, as its ring drains:
    // asked for where they are used, every group of four matrix instructions waited a round trip to the L2 of its own (eight per evaluation)
    double afr[2][4][4];
    auto load_afr = [&](const int tt) {
        const int T = 2 * wave + tt;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const gptr_t af = dense_Af + ((size_t)(min(o, nOt - 1) * (DENSE_D / 16) + T) * 4) * 64 + lane;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) afr[tt][o][s4] = af[s4 * 64];
        }
    };
#pragma unroll
    for (int s = 0; s < 64; ++s) {
        const d2v_t a = ring[s % D2_DEPTH];
        if (s + D2_DEPTH < 64) ring[s % D2_DEPTH] = a2[(s + D2_DEPTH) * 64];
        if (s == 64 - D2_DEPTH) load_afr(0);
        if (s == 64 - D2_DEPTH / 2) load_afr(1);
        const double hb = s_h1[64 * s + lane];
        gacc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, hb, gacc[0], 0, 0, 0);
        gacc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, hb, gacc[1], 0, 0, 0);
    }
    d4_t yacc[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) yacc[o] = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        double h[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = smm_tanh(gacc[tt][r]);
        // (an accumulator's chain in the contract's order — column tile by column tile, k-step by k-step —, the four moments' chains side by side)
        if (nOt == 4) {   // (uniform; straight-line code for the full four tiles of moments)
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
                for (int o = 0; o < 4; ++o) yacc[o] = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[tt][o][s4], h[s4], yacc[o], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                if (o < nOt) {
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) yacc[o] = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[tt][o][s4], h[s4], yacc[o], 0, 0, 0);
                }
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        if (o < nOt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) s_part[((size_t)wave * nmp + 16 * o + lk + 4 * r) * 16 + li] = yacc[o][r];
        }
    }
}

template <int CT>
__device__ __forceinline__ void dense_tile_v(const int np, const int nOt, const double* dense_Bf, const double* dense_Af, const double* dense_A2f, const uint32_t theta_off, const uint32_t part_off, const int tid) {
    const int g = (np + 15) / 16;   // (uniform)
    if (dense_A2f) {   // (uniform: spec v2)
        if (g <= 1) dense2_tile_n<CT, 1>(np, nOt, dense_Bf, dense_A2f, dense_Af, theta_off, part_off, tid);
        else if (g == 2) dense2_tile_n<CT, 2>(np, nOt, dense_Bf, dense_A2f, dense_Af, theta_off, part_off, tid);
        else if (g == 3) dense2_tile_n<CT, 3>(np, nOt, dense_Bf, dense_A2f, dense_Af, theta_off, part_off, tid);
        else dense2_tile_n<CT, 4>(np, nOt, dense_Bf, dense_A2f, dense_Af, theta_off, part_off, tid);
        return;
    }
    if (g <= 1) dense_tile_n<CT, 1>(np, nOt, dense_Bf, dense_Af, theta_off, part_off, tid);
    else if (g == 2) dense_tile_n<CT, 2>(np, nOt, dense_Bf, dense_Af, theta_off, part_off, tid);
    else if (g == 3) dense_tile_n<CT, 3>(np, nOt, dense_Bf, dense_Af, theta_off, part_off, tid);
    else dense_tile_n<CT, 4>(np, nOt, dense_Bf, dense_Af, theta_off, part_off, tid);
}

template <int CT>
__device__ __forceinline__ void dense_tile(const KParams& P, const double* s_theta, double* s_part, int tid) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dense_lds_base[];
    dense_tile_v<CT>(P.np, P.dense_nOt, P.dense_Bf, P.dense_Af, P.dense_A2f, (uint32_t)((const unsigned char*)s_theta - dense_lds_base), (uint32_t)((const unsigned char*)s_part - dense_lds_base), tid);
}

// value / simulated moments / status for one chain from its reduced sums
// (ObjExamples.jl:79-110; banana :251-265; "exception" -> status -2, mprob.jl:183-186).
// s_mom / s_w: data moments and weights staged in LDS.
// moment k of chain ci from the reduced sums: the 8 wave totals left to right, mean, deviation over the weight, square
// (ObjExamples.jl:79-100).  Independent across k: the lanes that serve a chain share the moments.
template <int CT>
__device__ inline void moment_term(const KParams& P, const double* s_part, const double* s_mom, const double* s_w, const int ci,
                                   const int k, double& m_out, double& v_out) {
    const bool dense = P.obj == SMM_OBJ_DENSE;
    const int nmp = P.dense_nOt * 16;
    double tot = dense ? s_part[((size_t)0 * nmp + k) * 16 + ci] : s_part[(0 * CT + ci) * P.nm + k];
#pragma unroll
    for (int wv = 1; wv < WG / 64; ++wv)
        tot = tot + (dense ? s_part[((size_t)wv * nmp + k) * 16 + ci] : s_part[(wv * CT + ci) * P.nm + k]);
    const double m = dense ? tot : tot / (double)P.ns;
    m_out = m;
    double d = m - s_mom[k];
    const double wk = s_w[k];
    if (!isnan(wk)) d = d / wk;
    v_out = d * d;
}

template <int CT>
__device__ inline void finish_objective(const KParams& P, const double* theta /*LDS [np]*/, const double* s_part,
                                        const double* s_mom, const double* s_w, int ci, double* simM /*[nm] out, LDS*/,
                                        double& value, int& status, int c_local = 0, const double* vk = nullptr) {
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
    int k = 0;
    if (vk) {   // mean and squared deviation already there (moment_term by the chain's lanes): eight at a time from LDS, added in order
        for (; k + 8 <= P.nm; k += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = vk[k + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) vsum = (k + u == 0) ? v[u] : vsum + v[u];
        }
    }
    for (; k < P.nm; ++k) {
        double v;
        if (vk) v = vk[k];
        else moment_term<CT>(P, s_part, s_mom, s_w, ci, k, simM[k], v);
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
    // (kind 3: the dense objective's spec v2 — its partial sums' region also stages the first hidden layer, [256][16])
    size_t part = kind == 1 ? (size_t)(WG / 64) * CT * nm : kind >= 2 ? (size_t)(WG / 64) * (((nm + 15) / 16) * 16) * 16 : 0;
    if (kind == 3 && part < (size_t)DENSE_D * 16) part = (size_t)DENSE_D * 16;
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
// ... with nr lanes per block
template <int NI>
__device__ inline void coop_fetch_n(double2 (&v)[NI], const double* __restrict__ g, const int W, const int r, const int nr) {
    const double2* __restrict__ gs = (const double2*)g;
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int i = r + k * nr;
        v[k] = i < W / 2 ? gs[i] : make_double2(0.0, 0.0);
    }
}
template <int NI>
__device__ inline void coop_put_n(double* lds_blk, const double2 (&v)[NI], const double* __restrict__ g, const int W, const int r, const int nr) {
    const double2* __restrict__ gs = (const double2*)g;
    double2* ld = (double2*)lds_blk;
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int i = r + k * nr;
        if (i < W / 2) ld[i] = v[k];
    }
    for (int i = r + NI * nr; i < W / 2; i += nr) ld[i] = gs[i];
}
template <int CT>
__device__ inline void coop_store(double* __restrict__ g, const double* lds_blk, int W, int r) {
    constexpr int NR = 64 / CT;
    double2* __restrict__ gd = (double2*)g;
    const double2* ld = (const double2*)lds_blk;
#pragma unroll 2
    for (int i = r; i < W / 2; i += NR) gd[i] = ld[i];
}

// the same with nr lanes per block
__device__ inline void coop_store_n(double* __restrict__ g, const double* lds_blk, const int W, const int r, const int nr) {
    double2* __restrict__ gd = (double2*)g;
    const double2* ld = (const double2*)lds_blk;
    for (int i = r; i < W / 2; i += nr) gd[i] = ld[i];
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
    // (the parameters and moments of the donor's record are copied by all lanes of the chain: copy_strided below)
}
// dst[k] = src[k] for k = r, r + nr, ... < n: a copy shared by the nr lanes that serve one chain
__device__ inline void copy_strided(double* dst, const double* src, const int n, const int r, const int nr) {
    for (int k = r; k < n; k += nr) dst[k] = src[k];
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
__device__ inline void exchange_walk_tile(const KParams& P, const int tx, unsigned char* lds, const int tid, const int ts_tile = 0) {
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
    if (P.ts && tid == 0) P.ts[(size_t)ts_tile * 8 + 5] = wall_clock64();   // staged
    uint32_t b = 0;
    uint32_t e = nlev > 0 ? level_end(0) : 0u;
    uint32_t e2 = nlev > 0 ? level_end(1) : 0u;
    uint32_t pw = (b + tid < e) ? pairs[b + tid] : 0u;
    double m = mi_u ? mi_v : ((b + tid < e) ? g_mi[b + tid] : 0.0);
    // (the loops twice: the default dist_fun `-` pays nothing for the menu)
    auto levels = [&](auto gen) {
        const int dk = decltype(gen)::value ? P.dist_fun : 0;
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
                if (dist_fun_eval(dk, si.val, sj.val) > m) {   // dist_fun (default -), AlgoBGP.jl:688
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
                        if (dist_fun_eval(dk, si.val, sj.val) > m) {
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
    };
    if (P.dist_fun != 0) levels(std::true_type{}); else levels(std::false_type{});
}

// The lean walk of smm_walk_lean.hpp in k_chain_iter (one min_improve >= 0 for all chains, dist_fun = -; the plan in its padded
// form): 16-byte slots {value, src | stamp << 16} built from the chains' values — with min_improve == 0 too, whose plan (made for
// the 8-byte key slots: offsets in units of 8, a dummy pair of two slots) is read with doubled offsets and finds two zero-valued
// slots behind the chains'.  The control wave of every tile of the workgroup takes its chains' result (src | partner << 32, as
// k_exch_resolve_* write it) into xr before the tile's blocks may overwrite the pair list.  false (nothing done, no barrier
// passed): this iteration's plan has more than 31 levels — the caller walks with exchange_walk_tile.
template <int NT>
__device__ inline bool exchange_walk_tile_lean(const KParams& P, const int tx, unsigned char* lds, const int tid, const bool valid, const int gc,
                                               unsigned long long& xr, const int ts_tile) {
    const int Ng = P.Ng;
    const int w = tx - P.plan_t0;
    const uint32_t* __restrict__ g_offp = P.lv_offp + (size_t)w * LV_OFFP;
    const uint4* __restrict__ g_pairs = (const uint4*)(P.lv_pairs_p + (size_t)w * P.plan_Kp);
    const int lane = tid & 63;
    const uint32_t Ng4 = (uint32_t)((Ng + 3) & ~3);
    const uint32_t pbase = 16u * (Ng4 + 2u);              // LDS offset of the pair words
    constexpr int PT = XLVL_MAX / NT;                     // chains per lane
    constexpr int PR = (XLVL_MAX + 64 * LV_MAXLEV + 4 * NT - 1) / (4 * NT);   // rounds of 16-byte loads for the pair words
    const uint32_t ov = g_offp[min(lane, LV_OFFP - 1)];   // lane l: first word of level l; lane 33: levels; lane 34: the plan fits
    double v_[PT];
#pragma unroll
    for (int r = 0; r < PT; ++r) {
        const int g = tid + r * NT;
        v_[r] = g < Ng ? P.vals[g] : 0.0;
    }
    uint4 p_[PR];
#pragma unroll
    for (int r = 0; r < PR; ++r) {
        const int q4 = tid + r * NT;
        p_[r] = 4 * q4 < P.plan_Kp ? g_pairs[q4] : make_uint4(0u, 0u, 0u, 0u);
    }
    const int nlev = __builtin_amdgcn_readlane((int)ov, 33);
    if (__builtin_amdgcn_readlane((int)ov, 34) == 0 || (uint32_t)(size_t)lds != 0u) return false;
#pragma unroll
    for (int r = 0; r < PT; ++r) {
        const int g = tid + r * NT;
        if (g < Ng) ((uint4*)lds)[g] = make_uint4((uint32_t)__double2loint(v_[r]), (uint32_t)__double2hiint(v_[r]), (uint32_t)g, 0u);
    }
#pragma unroll
    for (int r = 0; r < PR; ++r) {
        const int q4 = tid + r * NT;
        if (4 * q4 < P.plan_Kp) ((uint4*)(lds + pbase))[q4] = p_[r];
    }
    if (tid < 2) ((uint4*)lds)[Ng4 + tid] = make_uint4(0u, 0u, 0u, 0u);   // the dummy pair's slots: 0 - 0 > min_improve is false
    const int ltail = lean_walk_tail(ov, nlev, lane);
    __syncthreads();
    if (P.ts && tid == 0) P.ts[(size_t)ts_tile * 8 + 5] = wall_clock64();   // staged
    const bool bytes16 = P.lean_wide && P.lean_unit == 16;   // the pair words hold byte offsets of 16-byte slots; else halves of them
    if (bytes16) lean_walk_levels<NT, 0, true>(nullptr, 0, pbase, ov, nlev, tid, ltail, P.mi_value);
    else lean_walk_levels<NT, 1, true>(nullptr, 0, pbase, ov, nlev, tid, ltail, P.mi_value);
    __syncthreads();   // (the narrow tail was wave 0's alone; the control waves of other tiles read now)
    if (valid) {
        const uint32_t meta = ((const uint4*)lds)[gc].z;
        const uint32_t partner = bytes16 ? lean_partner<0, 4>(lds, pbase, meta, (uint32_t)gc) : lean_partner<1, 4>(lds, pbase, meta, (uint32_t)gc);
        xr = (unsigned long long)(meta & 0xffffu) | ((unsigned long long)partner << 32);
    }
    __syncthreads();   // (from here on the tile's blocks may overwrite the pair list)
    return true;
}

#include "smm_cone.hpp"

// The lean KEY walk in k_chain_iter, for single shards of 4096 < N <= 8192 chains (BASELINE config 4: banana, 8192 chains — one
// launch per iteration instead of chain kernel + stand-alone resolution): 8-byte slots {order_key32(value), src | stamp << 16}
// built from the chains' values exactly as k_exch_resolve_lean builds them, min_improve == 0, dist_fun = -.  The control wave of
// every tile takes its chains' result (src | partner << 32) into xr.  false (uniform; nothing done): the plan has more than 31
// levels or a value is NaN — the host does not launch this form where it knows of either; loud where it happens anyway.
template <int NT>
__device__ inline bool exchange_walk_tile_keys(const KParams& P, const int tx, unsigned char* lds, const int tid, const bool valid, const int gc,
                                               unsigned long long& xr, const int ts_tile, const int gc2 = -1, uint32_t* src2 = nullptr) {
    const int Ng = P.Ng;
    const int w = tx - P.plan_t0;
    const uint32_t* __restrict__ g_offp = P.lv_offp + (size_t)w * LV_OFFP;
    const uint4* __restrict__ g_pairs = (const uint4*)(P.lv_pairs_p + (size_t)w * P.plan_Kp);
    const int lane = tid & 63;
    const uint32_t Ng4 = (uint32_t)((Ng + 3) & ~3);
    const uint32_t pbase = 8u * (Ng4 + 4u);               // LDS offset of the pair words
    constexpr int PR = (XLDS_MAX + 64 * LV_MAXLEV + 4 * NT - 1) / (4 * NT);   // rounds of 16-byte loads for the pair words
    const uint32_t ov = g_offp[min(lane, LV_OFFP - 1)];   // lane l: first word of level l; lane 33: levels; lane 34: the plan fits
    // the chains' slots as the accept step wrote them ({order_key32(value), chain}; a NaN value set the sticky flag): two per 16 bytes
    constexpr int SR = XLDS_MAX / (2 * NT);               // 16-byte pieces per lane
    const uint32_t wflags = P.walk_flags[tid & 3];        // (word 0 is looked at; a per-lane address keeps it a vector load among the others)
    uint4 s_[SR];
#pragma unroll
    for (int r = 0; r < SR; ++r) {
        const int q = tid + r * NT;
        s_[r] = 2 * q < Ng ? ((const uint4*)P.slot8)[q] : make_uint4(0u, 0u, 0u, 0u);
    }
    uint4 p_[PR];
#pragma unroll
    for (int r = 0; r < PR; ++r) {
        const int q4 = tid + r * NT;
        p_[r] = 4 * q4 < P.plan_Kp ? g_pairs[q4] : make_uint4(0u, 0u, 0u, 0u);
    }
    const int nlev = __builtin_amdgcn_readlane((int)ov, 33);
    if (__builtin_amdgcn_readlane((int)ov, 34) == 0 || __builtin_amdgcn_readlane((int)wflags, 0) != 0 || (uint32_t)(size_t)lds != 0u) return false;
    uint2* slot = (uint2*)lds;
#pragma unroll
    for (int r = 0; r < SR; ++r) {
        const int q = tid + r * NT;
        if (2 * q < Ng) ((uint4*)lds)[q] = s_[r];
    }
#pragma unroll
    for (int r = 0; r < PR; ++r) {
        const int q4 = tid + r * NT;
        if (4 * q4 < P.plan_Kp) ((uint4*)(lds + pbase))[q4] = p_[r];
    }
    if (tid == 0) { slot[Ng4] = make_uint2(1u, 0u); slot[Ng4 + 1] = make_uint2(2u, 0u); }   // the dummy pair's slots: keys 1 < 2, "no swap"
    const int ltail = lean_walk_tail(ov, nlev, lane);
    __syncthreads();
    if (P.ts && tid == 0) P.ts[(size_t)ts_tile * 8 + 5] = wall_clock64();   // staged
    if (P.lean_unit == 8) lean_walk_levels<NT, 0>(P.vals, 1, pbase, ov, nlev, tid, ltail);
    else lean_walk_levels<NT, 1>(P.vals, 1, pbase, ov, nlev, tid, ltail);
    __syncthreads();   // (the narrow tail was wave 0's alone; the control waves of other tiles read now)
    if (valid) {
        const uint32_t meta = slot[gc].y;
        const uint32_t partner = P.lean_unit == 8 ? lean_partner<0>(lds, pbase, meta, (uint32_t)gc) : lean_partner<1>(lds, pbase, meta, (uint32_t)gc);
        xr = (unsigned long long)(meta & 0xffffu) | ((unsigned long long)partner << 32);
    }
    if (gc2 >= 0) *src2 = slot[gc2].y & 0xffffu;   // (the dense kind: the record source of the chain this lane loads)
    __syncthreads();   // (from here on the tile's blocks may overwrite the pair list)
    return true;
}

// The same walk with the level loop written for latency: a level is ONE LDS round trip (both 16-byte slots of a pair with a
// ds_read_b128 each), the compare, the two 16-byte writes of a swap and the barrier; the next level's pair word is fetched
// ahead; nothing else is in the loop (thresholds: a scalar when min_improve is uniform — the loop is instantiated twice).
// The level walk is latency, not bandwidth: a level with a handful of pairs costs almost what a level with a thousand does
// (measured per level of the first form: 1784 cycles for 1000 pairs, ~850 for 50), so what counts is the dependent chain
// per level.  FINAL_BARRIER = false: only the walking wave 0 reads the result (its own LDS operations complete in order).
constexpr uint32_t XNOPAIR = 0xffffffffu;   // i == j == 0xffff never occurs (chain ids < XLVL_MAX)
// (inline asm: left to itself the compiler reads only the 8 value bytes first and fetches src with a second, dependent LDS
// read inside the swap branch — two round trips per level instead of one.  The asm's own LDS operations are invisible to the
// compiler's counters, hence the explicit waits: after the reads, and walk_wait_lds() before every barrier.)
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ inline void walk_pair(const uint32_t slot_base, const uint32_t pw, const double m) {
    const uint32_t i = pw & 0xffffu, j = pw >> 16;
    const uint32_t ai = slot_base + i * 16u, aj = slot_base + j * 16u;
    u32x4_t si, sj;
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(si), "=&v"(sj) : "v"(ai), "v"(aj) : "memory");
    const double vi = __builtin_bit_cast(double, ((unsigned long long)si.y << 32) | si.x);
    const double vj = __builtin_bit_cast(double, ((unsigned long long)sj.y << 32) | sj.x);
    if (vi - vj > m) {                                   // dist_fun = -, AlgoBGP.jl:688
        u32x4_t ni = sj, nj = si;                        // swap_ev_ij!, :739-744; set_exchanged!, :747-748
        ni.w = j + 1; nj.w = i + 1;
        asm volatile("ds_write_b128 %0, %2\n\tds_write_b128 %1, %3" :: "v"(ai), "v"(aj), "v"(ni), "v"(nj) : "memory");
    }
}
__device__ inline void walk_wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
template <int NT, bool MI_U, bool FINAL_BARRIER>
__device__ inline void walk_levels(const KParams& P, const uint32_t slot, const uint32_t* pairs, const double* __restrict__ g_mi, const uint32_t ev,
                                   const uint32_t* __restrict__ g_off, const int nlev, const int ltail, const int tid) {
    const double mi_v = P.mi_value;
    auto level_end = [&](int l) -> uint32_t {
        const int lc = min(l, nlev - 1);
        return lc < 64 ? (uint32_t)__builtin_amdgcn_readlane((int)ev, lc) : g_off[lc];
    };
    // (level ends and this wave's first position are scalars: a wave without a pair in a level runs no vector instruction for it)
    uint32_t b = 0;
    uint32_t e = nlev > 0 ? level_end(0) : 0u;
    uint32_t e2 = nlev > 0 ? level_end(1) : 0u;
    uint32_t pw = (b + tid < e) ? pairs[b + tid] : XNOPAIR;
    double m = MI_U ? mi_v : ((b + tid < e) ? g_mi[b + tid] : 0.0);
#pragma clang loop unroll(disable)
    for (int l = 0; l < ltail; ++l) {
        const uint32_t e3 = level_end(l + 2);
        uint32_t pw2 = XNOPAIR;
        double m2 = mi_v;
        {   // this thread's first pair of the next level
            pw2 = (e + tid < e2) ? pairs[e + tid] : XNOPAIR;
            if constexpr (!MI_U) m2 = (e + tid < e2) ? g_mi[e + tid] : 0.0;
        }
        {
            if (pw != XNOPAIR) walk_pair(slot, pw, m);
            for (uint32_t pos = b + tid + NT; pos < e; pos += NT) walk_pair(slot, pairs[pos], MI_U ? mi_v : g_mi[pos]);   // levels wider than the workgroup
        }
        b = e; e = e2; e2 = e3; pw = pw2; m = m2;
        walk_wait_lds();
        __syncthreads();
        if (P.ts_levels && tid == 0 && blockIdx.x == 0 && l < 30) P.ts[(size_t)8 * 60000 + 16 + l] = clock64();
    }
    if (P.ts && tid == 0 && blockIdx.x == 0) { P.ts[(size_t)8 * 60000 + 14] = (unsigned long long)ltail; P.ts[(size_t)8 * 60000 + 7] = (unsigned long long)nlev; }
    if (ltail < nlev) {
        if (tid < 64) {   // the narrow tail: wave 0 alone, no barriers (pw, m hold this lane's pair of level ltail, e its end, e2 the next end)
#pragma clang loop unroll(disable)
            for (int l = ltail; l < nlev; ++l) {
                const uint32_t e3 = level_end(l + 2);
                const uint32_t pw2 = (e + tid < e2) ? pairs[e + tid] : XNOPAIR;
                double m2 = mi_v;
                if constexpr (!MI_U) m2 = (e + tid < e2) ? g_mi[e + tid] : 0.0;
                if (pw != XNOPAIR) walk_pair(slot, pw, m);
                __builtin_amdgcn_wave_barrier();
                pw = pw2; m = m2; e = e2; e2 = e3;
                if (P.ts_levels && tid == 0 && blockIdx.x == 0 && l < 30) { walk_wait_lds(); P.ts[(size_t)8 * 60000 + 16 + l] = clock64(); }
            }
            walk_wait_lds();
        }
        if constexpr (FINAL_BARRIER) __syncthreads();
    }
}

struct WalkNoWork { __device__ inline void operator()() const {} };
// while_loading: work of the caller that needs no memory (it runs between the request of the walk's inputs and their arrival)
template <int NT, bool FINAL_BARRIER, class F = WalkNoWork>
__device__ inline void exchange_walk_fast(const KParams& P, const int tx, unsigned char* lds, const int tid, const int ts_tile,
                                          F while_loading = F()) {
    const int Ng = P.Ng, K = P.plan_K;
    const int w = tx - P.plan_t0;
    uint4* slot = (uint4*)lds;                                  // [Ng] {value lo, value hi, src, partner}
    uint32_t* pairs = (uint32_t*)(lds + walk_slot_bytes(Ng));   // [K]
    const uint32_t* __restrict__ g_off = P.lv_off + (size_t)w * (K + 2);
    const uint32_t* __restrict__ g_pairs = P.lv_pairs + (size_t)w * K;
    const double* __restrict__ g_mi = P.lv_mi + (size_t)w * K;
    constexpr int PT = XLVL_MAX / NT;
    const int lane = tid & 63;
    double v_[PT];
    uint32_t pq_[PT];
#pragma unroll
    for (int r = 0; r < PT; ++r) {   // one round trip of global loads
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
    while_loading();
#pragma unroll
    for (int r = 0; r < PT; ++r) {
        const int g = tid + r * NT;
        const unsigned long long uv = __builtin_bit_cast(unsigned long long, v_[r]);
        if (g < Ng) slot[g] = make_uint4((uint32_t)uv, (uint32_t)(uv >> 32), (uint32_t)g, 0u);
    }
#pragma unroll
    for (int r = 0; r < PT; ++r) {
        const int q = tid + r * NT;
        if (q < K) pairs[q] = pq_[r];
    }
    for (int q = tid + PT * NT; q < K; q += NT) pairs[q] = g_pairs[q];   // injected pair lists longer than XLVL_MAX
    // the narrow tail (every remaining level <= 64 pairs) is walked by wave 0 alone
    int ltail = nlev;
    if (nlev > 0 && nlev <= 64) {
        const uint32_t lo = (uint32_t)__shfl_up((int)ev, 1, 64);
        const unsigned long long wide = __ballot(lane < nlev && ev - (lane > 0 ? lo : 0u) > 64u);
        ltail = wide ? 64 - __builtin_clzll(wide) : 0;
    }
    __syncthreads();
    if (P.ts && tid == 0) P.ts[(size_t)ts_tile * 8 + 5] = wall_clock64();   // staged
    const uint32_t slot_base = (uint32_t)(size_t)lds;   // LDS byte address of the chain slots
    // (dist_fun = - only: k_chain_iter_norm does not walk inline for the other entries of the menu — the host sees to it)
    if (P.mi_uniform) walk_levels<NT, true, FINAL_BARRIER>(P, slot_base, pairs, g_mi, ev, g_off, nlev, ltail, tid);
    else walk_levels<NT, false, FINAL_BARRIER>(P, slot_base, pairs, g_mi, ev, g_off, nlev, ltail, tid);
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
// proposal batches of at least this many components are drawn a lane per component pair (below: a lane per try walks them)
#ifndef SMM_COOP_MIN_BATCH
#define SMM_COOP_MIN_BATCH 6
#endif
// (IW: the inline exchange walk is compiled in — contexts that never walk inline, larger populations and shards, run the kernel
// without it: the walk's code costs the latency-bound prologue registers and scalar spills even when it is never entered)
template <int KIND, int CT, int TPW = 1, bool IW = true>
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
    // a hard error raised by an earlier iteration (AlgoBGP.jl:341,409 abort the run): later launches store nothing
    unsigned long long err_word = ERR_NONE;
    if (ctl || KIND == 2 || (KIND == 0 && blockDim.x != 64)) err_word = *(const volatile unsigned long long*)P.err;   // (every wave that takes part in the shared epilogue: they meet at barriers behind the exit below)
    // wave 1: problem constants, requested now and written to LDS after the walk
    // (objectives without a simulation are launched with the control wave only, 64 lanes per tile: it loads the constants itself)
    const bool slim = KIND == 0 && blockDim.x == 64;
    const bool wave1 = slim ? ctl : (tid >= 64 && tid < 128);
    const int k1 = slim ? tid : tid - 64;
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
        // the dense kind (long blocks: 832-byte records, 808 bytes of randomness per chain at 50 parameters) without an inline walk:
        // every wave of the tile moves the blocks, 32 lanes per chain — two round trips of two or three 16-byte loads per lane
        // (the control wave alone, 4 lanes per chain: 13 + 13 pieces per lane, most of them load-store round trips: 6 us)
        // (with the key walk in the prologue — P.gen_lean == 2, N_global <= 4096 — as well: its slots and lists lie UNDER the tile's blocks)
        const bool wide_load = KIND == 2 && (!(flags & F_WALK_INLINE) || P.gen_lean == 2);
        if (wide_load) {
            constexpr int L2 = WG / CT;
            const int ccw = tid / L2, rw = tid % L2;
            const int cw = tile * CT + ccw;
            const bool vw = cw < N;
            const int cwc = vw ? cw : 0;
            const int gcw = P.offset + cwc;
            double2 w_cs[1], w_rb[2], w_rec[2];
            const double* g_csw = P.cs + (size_t)cwc * CSW;
            const double* g_rbw = P.rb + ((size_t)(t > 1 ? t - P.rb_t0 : 0) * N + cwc) * RBW;
            const int rbww = t > 1 ? RBW : 0;
            unsigned long long xrw = (unsigned long long)(unsigned)gcw;
            const bool walk_here = IW && (flags & F_WALK_INLINE);
            if (vw) {
                if ((flags & F_HAS_PENDING) && !walk_here) xrw = P.xres[gcw];
                coop_fetch_n<1>(w_cs, g_csw, CSW, rw, L2);
                coop_fetch_n<2>(w_rb, g_rbw, rbww, rw, L2);
            }
            unsigned long long xr_ctl = (unsigned long long)(unsigned)gc;
            if constexpr (IW && KIND == 2 && TPW == 1) {
                if (walk_here) {
                    // exchangeMoves! of iteration t-1 by the tile's own walk over its cone (smm_cone.hpp; where the plan kernel made none: over
                    // the whole list), while the blocks above are in flight; nothing of the tile has been written to LDS yet
                    uint32_t srcw = (uint32_t)gcw;
                    bool done = false;
                    if (P.cone_ok) done = exchange_walk_tile_cone<WG, 0>(P, t - 1, (unsigned char*)smem, (int)threadIdx.x, valid, gc, xr_ctl, (int)blockIdx.x, tile, vw ? gcw : -1, &srcw);
                    if (!done && !exchange_walk_tile_keys<WG>(P, t - 1, (unsigned char*)smem, (int)threadIdx.x, valid, gc, xr_ctl, tile, vw ? gcw : -1, &srcw) && threadIdx.x == 0)
                        report_error(P, 3, t, gc);
                    xrw = (unsigned long long)srcw;
                }
            }
            if (wave1) {  // problem constants into the tile's LDS
                if (k1 < np) { S.lb[k1] = c_lb; S.ub[k1] = c_ub; S.init[k1] = c_init; }
                if (k1 < nm) { S.mom[k1] = c_mom; S.w[k1] = c_w; }
                for (int k = k1 + 64; k < np; k += 64) { S.lb[k] = P.lb[k]; S.ub[k] = P.ub[k]; S.init[k] = P.init[k]; }
                for (int k = k1 + 64; k < nm; k += 64) { S.mom[k] = P.mom[k]; S.w[k] = P.w[k]; }
            }
            if (vw) {
                const int sw = (int)(unsigned)(xrw & 0xffffffffu) - ((flags & F_GLOBAL_REC) ? 0 : P.offset);
                const double* g_recw = rec_in + (size_t)sw * RW;
                coop_fetch_n<2>(w_rec, g_recw, RW, rw, L2);
                coop_put_n<1>(S.cs + ccw * CSW, w_cs, g_csw, CSW, rw, L2);
                coop_put_n<2>(S.rb + ccw * RBW, w_rb, g_rbw, rbww, rw, L2);
                coop_put_n<2>(S.rec + ccw * RW, w_rec, g_recw, RW, rw, L2);
            }
            if (walk_here) { if (valid) partner = (int)(xr_ctl >> 32); }
            else if (valid && (flags & F_HAS_PENDING)) partner = (int)(P.xres[gc] >> 32);   // (the control wave's lanes serve other chains than they loaded)
        } else {
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
        if (IW && (flags & F_WALK_INLINE)) {
            // exchangeMoves! of iteration t-1, by all lanes of the tile, while the level-1 blocks are in flight
            // (the tile's own LDS blocks overlay the walk's pair list: nothing of the tile is written before this returns)
            if (P.gen_lean == 2) {   // 4096 < N <= 8192: the key walk (no other form fits the LDS at this size)
                bool done = false;
                if constexpr (WG * TPW == 1024) {   // the workgroup's cone, where the plan kernel made one (smm_cone.hpp)
                    if (P.cone_ok)
                        done = P.lean_unit == 8 ? exchange_walk_tile_cone<1024, 0>(P, t - 1, (unsigned char*)smem, (int)threadIdx.x, valid, gc, xr, (int)blockIdx.x, tile)
                                                : exchange_walk_tile_cone<1024, 1>(P, t - 1, (unsigned char*)smem, (int)threadIdx.x, valid, gc, xr, (int)blockIdx.x, tile);
                }
                if (!done && !exchange_walk_tile_keys<WG * TPW>(P, t - 1, (unsigned char*)smem, (int)threadIdx.x, valid, gc, xr, tile) && threadIdx.x == 0)
                    report_error(P, 3, t, gc);
            } else if (!(P.gen_lean && exchange_walk_tile_lean<WG * TPW>(P, t - 1, (unsigned char*)smem, (int)threadIdx.x, valid, gc, xr, tile))) {
                exchange_walk_tile<WG * TPW>(P, t - 1, (unsigned char*)smem, (int)threadIdx.x, tile);
                if (valid) {
                    const XSlot sv = ((const XSlot*)smem)[gc];
                    xr = (unsigned long long)sv.src | ((unsigned long long)sv.partner << 32);
                }
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
    if (valid && t > 1 && partner != 0)   // parameters and moments of the donor's record into the rewritten history row, NR lanes per chain
        copy_strided(S.hp + cl * HW + H_PARAMS, S.rec + cl * RW + 3, np + nm, r, NR);
    TS_MARK(5);
    // ---- proposal(c), AlgoBGP.jl:424-471: lane (cl, r) evaluates try r of chain cl ----
    {
        double* th = S.theta + cl * np;
        const double* rc = S.rec + cl * RW;
        const bool draws = (t > 1) && !(P.dbg & 1);   // uniform: iteration 1 proposes the initial value (:426-427)
        if (ctl && (!draws || !valid)) {
            if (r == 0)
                for (int k = 0; k < np; ++k) th[k] = !valid ? 0.0 : (t == 1 ? S.init[k] : rc[3 + k]);
        }
        if (draws) {   // the control wave's 64 lanes walk the batches together; lanes of absent chains take no part in the tries
            const int bs = P.batch_size;
            const int max_tries = P.user_n ? min(P.rb_tries, P.smpl_iters) : P.smpl_iters;
            const int npar = min(min(NR, P.rb_tries), max_tries);  // tries evaluated side by side
            const double sg = S.cs[cl * CSW + CS_SIGMA];
            const double* zz = S.rb + cl * RBW + 1;  // [tries][np]
            // mapto_01 (mprob.jl:248) once per chain and parameter — one division each, shared by all tries —, computed by the
            // NR lanes of the chain and kept in the (still unused) output record block
            double* m01 = S.rout + cl * RW;
            // Many components (the whole proposal was 10.5 us per iteration at 50 parameters: one lane per chain and try walking
            // all the components, then the redraw loop of mysample): every wave of the tile works, a chain is served by
            // 64 * waves / CT lanes of ONE wave, a lane by the component pairs q = sl, sl + LPC, ... (one generator call per pair
            // and try).  The tries are taken in order, each one tested by all the chain's lanes at once (a segment of the
            // wave's ballot); the first one inside the unit box wins: same tries, same order, same winner as the serial form.
            const bool coop = bs >= SMM_COOP_MIN_BATCH && !P.chol_L;   // (uniform)
            if (coop) {
                __syncthreads();   // the blocks the control wave staged (records, state, randomness) are every wave's now
                // (smm_propose.hpp: every wave of the tile works, 64 * waves / CT lanes per chain)
                const CoopProp X{S.rec, RW, S.rout, RW, S.theta, np, S.h, HW, S.rb, RBW, S.cs, CSW, S.lb, S.ub,
                                 (unsigned long long*)(S.h - (size_t)st * ((tile_smem_doubles(CT, np, nm, RW, HW, RBW, KIND) + 1) & ~(size_t)1)) + 2, P.err,
                                 P.seed, P.offset, N, bs, P.rb_tries, P.user_n, P.smpl_iters, P.scout_after, P.scout_gl};
                coop_mysample<CT>(X, t, tile, tid, (int)blockDim.x / (64 * TPW), threadIdx.x == 0, CoopSyncThreads());
            } else if (ctl) {
            if (valid)
                for (int k = r; k < np; k += NR) {
                    const double lbk = S.lb[k];
                    m01[k] = (rc[3 + k] - lbk) / (S.ub[k] - lbk);
                }
            __builtin_amdgcn_wave_barrier();
            for (int b0 = 0; b0 < np; b0 += bs) {
                bool ok = valid && r < npar;
                if (ok) {
                    for (int k = b0; k < b0 + bs; ++k) {  // mysample, :400-410, try r
                        const double mu01 = m01[k];
                        const double step = sg * prop_direction(P, zz + r * np, k, gc);  // MvNormal(mu01, sigma): x = mu + sigma*z
                        const double x = mu01 + step;
                        if (!(x >= 0.0 && x <= 1.0)) ok = false;  // inclusive bounds, :405
                    }
                }
                const unsigned long long m = __ballot(ok);  // first successful try of every chain
                unsigned long long pat = 0;
#pragma unroll
                for (int rr = 0; rr < NR; ++rr) pat |= ((m >> (rr * CT + cl)) & 1ull) << rr;
                const int rwin = pat ? (__ffsll((long long)pat) - 1) : -1;
                if (valid && rwin == r) {
                    for (int k = b0; k < b0 + bs; ++k) {
                        const double lbk = S.lb[k];
                        const double span = S.ub[k] - lbk;
                        const double mu01 = m01[k];
                        const double step = sg * prop_direction(P, zz + r * np, k, gc);
                        const double x = mu01 + step;
                        const double sc = x * span;
                        th[k] = sc + lbk;  // mapto_ab, mprob.jl:271
                    }
                }
                if (valid && r == 0 && rwin < 0) {   // every failing chain's own lane redraws one try at a time (chains in parallel)
                    bool ok2 = false;
                    for (int rr = npar; rr < max_tries && !ok2; ++rr) {
                        ok2 = true;
                        double zc0 = 0.0, zc1 = 0.0;
                        int zq = -1;
                        const double* zv = zz + rr * np;   // Cholesky kernel: the whole try's normals
                        if (P.chol_L && rr >= P.rb_tries) {   // ... generated into the (still unused) history row of the chain
                            double* sc = S.h + cl * HW;
                            for (int q = 0; 2 * q < np; ++q) {
                                const double2 zz2 = rng_prop_normal2_outofline(P.seed, (uint32_t)gc, (uint32_t)t, (uint32_t)rr, (uint32_t)q);
                                sc[2 * q] = zz2.x;
                                if (2 * q + 1 < np) sc[2 * q + 1] = zz2.y;
                            }
                            zv = sc;
                        }
                        for (int k = b0; k < b0 + bs; ++k) {
                            const double lbk = S.lb[k];
                            const double span = S.ub[k] - lbk;
                            const double mu01 = m01[k];
                            double z;
                            if (P.chol_L) {
                                z = prop_direction(P, zv, k, gc);
                            } else if (rr < P.rb_tries) {
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
                            if (!(x >= 0.0 && x <= 1.0)) ok2 = false;
                            const double sc = x * span;
                            th[k] = sc + lbk;
                        }
                    }
                    if (!ok2) report_error(P, ERRK_NO_DRAW, t, gc);  // :409
                }
            }
            }
        }
    }
    TS_MARK(6);
    __syncthreads();
    TS_MARK(2);
    if (flags & F_PROPOSE_ONLY) {  // user objective: hand the proposals to the user's kernel; nothing has been stored yet, the
        if (valid && !error_before(err_word, t))   // accept launch repeats this (deterministic) prologue
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
    if (error_before(err_word, t)) return;

    // ---- objective value, doAcceptReject! (:324-392), set_eval! (:220-245) ----
    // the moments of a chain (wave totals -> mean -> squared weighted deviation) are independent: its NR lanes share them
    const bool sumsq = KIND != 0 && P.obj != SMM_OBJ_USER && P.obj != SMM_OBJ_BANANA;
    // (the dense kind has all its waves here: the terms, and below the copies and the stores of the long blocks, are shared by the
    // tile's 512 lanes, 32 per chain — the control wave alone: 7.3 + 3.5 us per iteration at 50 parameters and 50 moments)
    // (so do objectives without a simulation wherever the tile was launched with all its waves — the key form of C4: its other waves
    // have nothing to do after the walk: 403 -> 410 M chain-evals/s)
    constexpr int LPC2 = WG / CT;   // lanes per chain when every wave of the tile takes part
    const int cc2 = tid / LPC2, r2 = tid % LPC2;
    const bool coop_epi = KIND == 2 || (KIND == 0 && !slim);
    const bool valid2 = coop_epi && tile * CT + cc2 < N;
    if (coop_epi) {
        if (valid2 && sumsq) {
            double* smk = S.h + cc2 * HW + H_PARAMS + np;
            double* vkk = S.rout + cc2 * RW;          // (the proposal's scratch: free again)
            for (int k = r2; k < nm; k += LPC2) moment_term<CT>(P, S.part, S.mom, S.w, cc2, k, smk[k], vkk[k]);
        }
        __syncthreads();
    } else if (valid && sumsq) {
        double* smk = S.h + cl * HW + H_PARAMS + np;
        double* vkk = S.rout + cl * RW;          // (the proposal's scratch: free again)
        for (int k = r; k < nm; k += NR) moment_term<CT>(P, S.part, S.mom, S.w, cl, k, smk[k], vkk[k]);
    }
    __builtin_amdgcn_wave_barrier();
    if (chain_lane) {
        const double* th = S.theta + cl * np;
        const double* rc = S.rec + cl * RW;
        double* hr = S.h + cl * HW;
        double* ro = S.rout + cl * RW;
        double* csb = S.cs + cl * CSW;
        double* sm = hr + H_PARAMS + np;
        double value;
        int status;
        finish_objective<CT>(P, th, S.part, S.mom, S.w, cl, sm, value, status, c, sumsq ? S.rout + cl * RW : nullptr);
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
            if (!(value >= 0.0)) report_error(P, ERRK_NEGATIVE, t, gc);  // :341
            const double e = smm_exp(atun * (old - value));   // (the contract exponential, smm_rng.hpp)
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
        // the chain's last accepted record (lastAccepted :209-215) = input of the exchange step: its head here, the
        // parameter and moment arrays (and the history row's parameters) by all lanes of the chain below
        if (acc) { ro[0] = value; ro[1] = prob; ro[2] = (double)status; }
        else { ro[0] = rc[0]; ro[1] = rc[1]; ro[2] = rc[2]; }
        const double vnew = acc ? value : old;
        P.vals_out[c] = vnew;
        if (P.slots17_out) {   // the chain's initial slots of k_exch_resolve_rows / _key (what k_exch_keys would make of vals[c])
            P.slots17_out[c] = (uint32_t)gc | (order_key17(vnew) << 15);
            if (vnew != vnew) atomicOr(P.nan_flags_out, 1u);
        }
        if (P.slot8_out) {   // the chain's slot at the start of the next inline key walk (exchange_walk_tile_keys)
            P.slot8_out[c] = make_uint2(order_key32(vnew), (uint32_t)gc);
            if (vnew != vnew) atomicOr(P.walk_flags, 1u);
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (coop_epi) {
        __syncthreads();   // the accept step's results (history head, record head, state block) are every wave's now
        if (valid2) {      // (a chain's 32 lanes sit in one wave: its copies are in place before its stores read them)
            const int c2 = tile * CT + cc2;
            const bool acc = S.h[cc2 * HW + H_ACC] != 0.0;
            copy_strided(S.h + cc2 * HW + H_PARAMS, S.theta + cc2 * np, np, r2, LPC2);
            if (acc) {
                copy_strided(S.rout + cc2 * RW + 3, S.theta + cc2 * np, np, r2, LPC2);
                copy_strided(S.rout + cc2 * RW + 3 + np, S.h + cc2 * HW + H_PARAMS + np, nm, r2, LPC2);
            } else {
                copy_strided(S.rout + cc2 * RW + 3, S.rec + cc2 * RW + 3, RW - 3, r2, LPC2);
            }
            __builtin_amdgcn_wave_barrier();
            coop_store_n(P.cs + (size_t)c2 * CSW, S.cs + cc2 * CSW, CSW, r2, LPC2);
            coop_store_n(rec_out + (size_t)c2 * RW, S.rout + cc2 * RW, RW, r2, LPC2);
            coop_store_n(P.hrec + ((size_t)(t - 1) * N + c2) * HW, S.h + cc2 * HW, HW, r2, LPC2);
            if (t > 1 && S.cs[cc2 * CSW + CS_PARTNER] != 0.0)
                coop_store_n(P.hrec + ((size_t)(t - 2) * N + c2) * HW, S.hp + cc2 * HW, HW, r2, LPC2);
        }
        TS_MARK(4);
        return;
    }
    if (valid) {
        const bool acc = S.h[cl * HW + H_ACC] != 0.0;
        copy_strided(S.h + cl * HW + H_PARAMS, S.theta + cl * np, np, r, NR);
        if (acc) {
            copy_strided(S.rout + cl * RW + 3, S.theta + cl * np, np, r, NR);
            copy_strided(S.rout + cl * RW + 3 + np, S.h + cl * HW + H_PARAMS + np, nm, r, NR);
        } else {
            copy_strided(S.rout + cl * RW + 3, S.rec + cl * RW + 3, RW - 3, r, NR);
        }
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
    if (*(const volatile unsigned long long*)P.err != ERR_NONE) return;   // the run stopped at the failing iteration
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

// objfunc_norm with options[:noseed] = true (ObjExamples.jl:71-75): every evaluation draws its own shock matrix (the
// counter generator keyed by base_seed + i) — getSigma's repetitions (econometrics.jl:125-145).  One workgroup per
// evaluation; lane l generates and sums its draws l, l + 512, ... (the numerical contract of the seeded form).
__global__ __launch_bounds__(WG) void k_eval_batch_noseed(const KParams P, const double* __restrict__ params, const int M,
                                                          const uint64_t base_seed, double* __restrict__ value,
                                                          double* __restrict__ simM, int8_t* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) double smem[];   // [WG/64][nm] wave totals
    const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nm = P.nm, ns = P.ns;
    const uint64_t seed = base_seed + (uint64_t)i;
    for (int q = 0; 2 * q < nm; ++q) {   // one Philox block serves moments 2q and 2q+1
        const int k0 = 2 * q, k1 = 2 * q + 1;
        const double mu0 = params[(size_t)k0 * M + i], mu1 = k1 < nm ? params[(size_t)k1 * M + i] : 0.0;
        double a0 = 0.0, a1 = 0.0;
        for (int s = tid; s < ns; s += WG) {
            double z0, z1;
            box_muller(philox_stream(seed, STREAM_Z, (uint32_t)s, (uint32_t)q, 0, 0), z0, z1);
            const double x0 = z0 + mu0, x1 = z1 + mu1;
            a0 = a0 + x0;
            a1 = a1 + x1;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { a0 = a0 + __shfl_xor(a0, off, 64); a1 = a1 + __shfl_xor(a1, off, 64); }
        if (lane == 0) { smem[wave * nm + k0] = a0; if (k1 < nm) smem[wave * nm + k1] = a1; }
    }
    __syncthreads();
    if (tid == 0) {
        double vsum = 0.0;
        for (int k = 0; k < nm; ++k) {
            double tot = smem[k];
            for (int wv = 1; wv < WG / 64; ++wv) tot = tot + smem[wv * nm + k];
            const double m = tot / (double)ns;
            simM[(size_t)k * M + i] = m;
            double d = m - P.mom[k];
            const double wk = P.w[k];
            if (!isnan(wk)) d = d / wk;
            const double v = d * d;
            vsum = (k == 0) ? v : vsum + v;
        }
        value[i] = vsum / (double)nm;
        status[i] = 1;
    }
}
