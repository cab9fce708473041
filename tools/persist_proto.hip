// tools/persist_proto.hip — the kill-criterion prototype of VERDICT r3 "Next #1": ONE persistent launch for a whole step of the
// headline configuration (4096 chains = 256 tiles of 16, one 1024-lane workgroup per CU, objfunc_norm 2p/2m, ns = 10000) whose
// workgroups are coupled by tagged slots instead of a kernel boundary.  Everything that costs time in the real iteration is here
// (cone gather past the caches with tag polling, the lone-wave walk over the cone's sub-levels, the donor's self-validating record,
// proposal, 640 FP64 adds per lane with the shocks resident in registers, accept step, publication, history row); what is NOT here
// is the library's bookkeeping detail (accept-rate counters, best/curr, errors).  The run is a real Markov chain, so the exchange
// pattern (who swaps with whom) is the real workload's.  Determinism check: two runs must end in bit-identical values — a stale
// or torn read of another workgroup's slot would change them.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/persist_proto tools/persist_proto.hip && tools/persist_proto [iters] [mode]
//   mode bit 0: no exchange at all (the floor: proposal + simulation + accept per iteration, no coupling)
//   mode bit 1: every workgroup gathers ALL 4096 slots instead of its cone's (what the inline p2p kernel did)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <algorithm>
#include <random>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int CT = 16, NWG = 1024, NSUB = 32, K = 8, GCAP = 512, HDRW = 16, HW = 12, ZR = 20;
constexpr unsigned long long TIMEOUT = 200000000ull;   // 2 s of the 100 MHz wall clock

struct PArgs {
    const double* Z;              // [2][zstride]
    const uint32_t* cone_pairs;   // [W][tiles][NSUB * 64]: (8 i) | (8 j) << 16, sub-level by sub-level, dummy-padded
    const uint16_t* cone_gather;  // [W][tiles][GCAP]: the chains whose initial slots the cone needs (the tile's own excluded)
    unsigned long long* ring_slot;   // [K][N + 4]: {key32, chain | tag << 16}
    uint4* ring_rec;              // [K][N][8]: per double {lo, tag, hi, tag}
    uint32_t* progress;           // [tiles]: last iteration whose prologue reads are complete
    double* hist;                 // [T][N][HW]
    double* out;                  // [N][4]: value, theta0, theta1, exchanged count
    unsigned long long* ts;       // [tiles][8]: accumulated phase times (wall clock ticks)
    uint32_t* err;
    int N, ns, zstride, T, W, mode;
    uint32_t seed;
};

__device__ inline unsigned long long wall_clock() { return wall_clock64(); }
__host__ __device__ inline uint32_t order_key32(const double v) {
    unsigned long long u;
    memcpy(&u, &v, 8);
    uint32_t h = (uint32_t)(u >> 32);
    if (u == 0x8000000000000000ull) h = 0u;
    return (h & 0x80000000u) ? ~h : (h | 0x80000000u);
}
__host__ __device__ inline uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ inline double u01(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    const uint32_t h1 = mix32(a * 0x9e3779b9u + mix32(b * 0x85ebca6bu + mix32(c + 0xc2b2ae35u * d)));
    const uint32_t h2 = mix32(h1 ^ 0x27d4eb2fu);
    return ((double)(((unsigned long long)h1 << 21) ^ (unsigned long long)h2 >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}
__device__ inline uint32_t tag_of(int t) { return 0x8000u | ((uint32_t)t & 0x7fffu); }

__device__ inline unsigned long long load8_sys(const void* p) {
    unsigned long long v;
    asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ inline void load16x2_sys(const void* p0, const void* p1, uint4& a, uint4& b) {
    u32x4 q0, q1;
    asm volatile("global_load_dwordx4 %0, %2, off sc0 sc1\n\tglobal_load_dwordx4 %1, %3, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(q0), "=&v"(q1) : "v"(p0), "v"(p1) : "memory");
    a = make_uint4(q0.x, q0.y, q0.z, q0.w); b = make_uint4(q1.x, q1.y, q1.z, q1.w);
}
__device__ inline void store16_sys(void* p, const u32x4 q) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p), "v"(q) : "memory"); }
__device__ inline void store8_sys(void* p, const unsigned long long v) { __hip_atomic_store((unsigned long long*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ inline void store4_sys(void* p, const uint32_t v) { __hip_atomic_store((uint32_t*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ inline uint32_t load4_sys(const void* p) { return __hip_atomic_load((const uint32_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// one LDS-DMA instruction: 16 bytes per lane from gsrc (per lane) to LDS byte address lds_dst (wave-uniform) + 16 * lane
__device__ inline void lds_dma16(const void* gsrc, const uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int CTN, int NN, int OFF>
__device__ inline void wave_reduce_step(double (&a)[CTN], int lane) {
    if constexpr (NN > 1) {
        const bool upper = (lane & OFF) != 0;
#pragma unroll
        for (int i = 0; i < NN / 2; ++i) {
            const double mine = upper ? a[i + NN / 2] : a[i];
            const double send = upper ? a[i] : a[i + NN / 2];
            const double recv = __shfl_xor(send, OFF, 64);
            a[i] = mine + recv;
        }
        wave_reduce_step<CTN, NN / 2, OFF / 2>(a, lane);
    } else if constexpr (OFF >= 1) {
        a[0] = a[0] + __shfl_xor(a[0], OFF, 64);
        wave_reduce_step<CTN, 1, OFF / 2>(a, lane);
    }
}

__global__ __launch_bounds__(NWG) void k_proto(const PArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = wave >> 3, wih = wave & 7;
    const int tile = (int)blockIdx.x, tiles = (int)gridDim.x;
    const int N = A.N, N4 = (N + 3) & ~3;
    const uint32_t pbase = 8u * (uint32_t)(N4 + 4);
    uint2* slots = (uint2*)lds;
    uint32_t* s_pairs = (uint32_t*)(lds + pbase);                    // [2][NSUB * 64]
    uint16_t* s_gl = (uint16_t*)(s_pairs + 2 * NSUB * 64);            // [2][GCAP]: nsub, ngather, -, ..; the list from entry 8
    double* s_theta = (double*)(s_gl + 2 * GCAP);                     // [CT][2]
    double* s_part = s_theta + CT * 2;                                // [2][8][CT]
    double* s_rng = s_part + 2 * 8 * CT;                              // [2][64][3], by iteration parity
    double* s_state = s_rng + 2 * 64 * 3;                             // [CT][8]: value, th0, th1, nexch (parked across the simulation)
    unsigned long long* s_ts = (unsigned long long*)(s_state + CT * 8);   // [8]
    unsigned* s_arrived = (unsigned*)(s_ts + 8);
    unsigned* s_minprog = s_arrived + 1;
    const bool noex = (A.mode & 1) != 0, gall = (A.mode & 2) != 0;

    // ---- the lane's shocks, once: lane l of half h sums the draws l, l + 512, ... of moment h (numerical contract) ----
    double z[ZR];
    bool has_last;
    {
        const int l = wih * 64 + lane;
#pragma unroll
        for (int u = 0; u < ZR; ++u) z[u] = (l + u * 512 < A.ns) ? A.Z[(size_t)h * A.zstride + l + u * 512] : 0.0;
        has_last = l + (ZR - 1) * 512 < A.ns;
    }
    const bool ctl = tid < 64;
    if (ctl) {   // control wave: four lanes per chain with identical state
        const int cl = lane >> 2, c = tile * CT + cl;
        if ((lane & 3) == 0) {
            double* st = s_state + cl * 8;
            st[0] = 1e30;
            st[1] = 0.5 + 0.3 * u01(A.seed, (uint32_t)c, 0u, 7u);
            st[2] = -0.5 - 0.3 * u01(A.seed, (uint32_t)c, 0u, 8u);
            st[3] = 0.0;
        }
    }
    if (tid == 0) { slots[N4] = make_uint2(1u, 0u); slots[N4 + 1] = make_uint2(2u, 0u); *s_arrived = 0u; *s_minprog = 0u; }
    if (tid < 8) s_ts[tid] = 0ull;

    for (int t = 1; t <= A.T; ++t) {
        const int buf = t & 1;
        const bool exch = !noex && t > 1;
        {   // the list of the NEXT iteration's exchange (state independent) by LDS-DMA: no register is held across the simulation
            const int wn = (t + 1) % A.W, nb = (t + 1) & 1;
            const size_t tb = (size_t)wn * tiles + tile;
            if (wave >= 4 && wave < 12) lds_dma16((const uint4*)(A.cone_pairs + tb * (NSUB * 64)) + (tid - 256), pbase + (uint32_t)nb * (NSUB * 64 * 4) + (uint32_t)(wave - 4) * 1024u);
            if (wave == 12) lds_dma16((const uint4*)(A.cone_gather + tb * GCAP) + lane, pbase + 2u * NSUB * 64 * 4 + (uint32_t)nb * (GCAP * 2));
        }
        __syncthreads();   // B0: this iteration's lists staged (requested an iteration ago), rng made, last epilogue done
        unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
        if (tid == 0) t0 = wall_clock();
        const int nsub = exch ? (int)s_gl[buf * GCAP + 0] : 0;
        const int ngat = exch ? (int)s_gl[buf * GCAP + 1] : 0;
        // ---- gather: the cone's initial slots, past the caches, every word says which iteration it is from ----
        if (exch) {
            const unsigned long long* rs = A.ring_slot + (size_t)((t - 1) % K) * (N + 4);
            const uint32_t want = tag_of(t - 1) << 16;
            const int n = gall ? N : ngat;
            for (int e = tid; e < n; e += NWG) {
                const int g = gall ? e : (int)s_gl[buf * GCAP + 8 + e];
                unsigned long long v = load8_sys(rs + g);
                if ((((uint32_t)(v >> 32)) & 0xffff0000u) != want) {
                    const unsigned long long ts0 = wall_clock();
                    do {
                        __builtin_amdgcn_s_sleep(1);
                        v = load8_sys(rs + g);
                        if (wall_clock() - ts0 > TIMEOUT) { atomicOr(A.err, 1u); break; }
                    } while ((((uint32_t)(v >> 32)) & 0xffff0000u) != want);
                }
                slots[g] = make_uint2((uint32_t)v, (uint32_t)(v >> 32) & 0xffffu);
            }
            if (!gall && ctl && (lane & 3) == 0) {
                const int cl = lane >> 2, c = tile * CT + cl;
                slots[c] = make_uint2(order_key32(s_state[cl * 8]), (uint32_t)c);
            }
        }
        __syncthreads();   // B1
        if (ctl) {
            const int cl = lane >> 2, r = lane & 3, c = tile * CT + cl;
            if (tid == 0) t1 = wall_clock();
            // ---- the walk over the cone's sub-levels: wave 0 alone, no barriers (its LDS operations complete in order) ----
            const uint32_t* pw_ = s_pairs + buf * (NSUB * 64);
            for (int s = 0; s < nsub; ++s) {
                const uint32_t pw = pw_[s * 64 + lane];
                const uint32_t ai = pw & 0xffffu, aj = pw >> 16;
                const uint2 si = *(const uint2*)(lds + ai), sj = *(const uint2*)(lds + aj);
                if (si.x > sj.x) {
                    const uint32_t stamp = (uint32_t)(s * 64 + lane + 1) << 16;
                    *(uint2*)(lds + ai) = make_uint2(sj.x, (sj.y & 0xffffu) | stamp);
                    *(uint2*)(lds + aj) = make_uint2(si.x, (si.y & 0xffffu) | stamp);
                }
            }
            if (tid == 0) t2 = wall_clock();
            double* st = s_state + cl * 8;
            double value = st[0], th0 = st[1], th1 = st[2];
            // ---- the record the chain continues from: its own or its donor's (self-validating, out of the ring) ----
            if (exch) {
                const uint2 me = slots[c];
                const int src = (int)(me.y & 0xffffu);
                if (src != c) {
                    const uint4* g_ll = A.ring_rec + ((size_t)((t - 1) % K) * N + src) * 8;
                    const uint32_t tag = tag_of(t - 1);
                    uint4 q0, q1;
                    load16x2_sys(g_ll + 2 * r, g_ll + 2 * r + 1, q0, q1);
                    if (!(q0.y == tag && q0.w == tag && q1.y == tag && q1.w == tag)) {
                        const unsigned long long ts0 = wall_clock();
                        do {
                            __builtin_amdgcn_s_sleep(1);
                            load16x2_sys(g_ll + 2 * r, g_ll + 2 * r + 1, q0, q1);
                            if (wall_clock() - ts0 > TIMEOUT) { atomicOr(A.err, 2u); break; }
                        } while (!(q0.y == tag && q0.w == tag && q1.y == tag && q1.w == tag));
                    }
                    const double d0 = __hiloint2double((int)q0.z, (int)q0.x), d1 = __hiloint2double((int)q1.z, (int)q1.x);
                    // lane r holds doubles 2r, 2r+1 of the record {value, prob, status, th0, th1, sm0, sm1, -}
                    value = __shfl(d0, (lane & ~3) + 0, 64);
                    th0 = __shfl(d1, (lane & ~3) + 1, 64);
                    th1 = __shfl(d0, (lane & ~3) + 2, 64);
                    // the exchanged chain's history row of iteration t-1 is rewritten (set_eval! of swap_ev_ij!)
                    double* hr = A.hist + ((size_t)(t - 2) * N + c) * HW;
                    hr[2 * r] = d0; hr[2 * r + 1] = d1;
                    if (r == 0) { st[0] = value; st[1] = th0; st[2] = th1; st[3] += 1.0; }
                }
            }
            if (lane == 0) store4_sys(A.progress + tile, (uint32_t)t);
            if (tid == 0) t3 = wall_clock();
            // ---- proposal: four tries side by side, first inside the box wins ----
            double p0 = th0, p1 = th1;
            if (t > 1) {
                const double sigma = 0.015 * (1.0 + 99.0 * (double)c / (double)(N - 1));
                const double* o = s_rng + (buf * 64 + lane) * 3;
                const double lb0 = -3.0, ub0 = 3.0;
                const double m0 = (th0 - lb0) / (ub0 - lb0), m1 = (th1 - lb0) / (ub0 - lb0);
                const double x0 = m0 + sigma * o[1], x1 = m1 + sigma * o[2];
                const bool ok = x0 >= 0.0 && x0 <= 1.0 && x1 >= 0.0 && x1 <= 1.0;
                const unsigned long long m = __ballot(ok);
                const unsigned quad = (unsigned)(m >> (lane & ~3)) & 0xfu;
                const int rwin = quad ? __builtin_ctz(quad) : 0;
                const double q0 = x0 * (ub0 - lb0) + lb0, q1 = x1 * (ub0 - lb0) + lb0;
                p0 = quad ? __shfl(q0, (lane & ~3) + rwin, 64) : th0;
                p1 = quad ? __shfl(q1, (lane & ~3) + rwin, 64) : th1;
            }
            if (r == 0) { s_theta[cl * 2] = p0; s_theta[cl * 2 + 1] = p1; }
            if (tid == 0) { t4 = wall_clock(); s_ts[0] += t1 - t0; s_ts[1] += t2 - t1; s_ts[2] += t3 - t2; s_ts[3] += t4 - t3; s_ts[6] = t4; }
        }
        __syncthreads();   // B2
        // ---- simulation: every lane, its 20 resident shocks x 16 chains (two passes of 8), the means from scalar registers ----
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            double acc[8], mu[8];
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) {
                const unsigned long long um = __builtin_bit_cast(unsigned long long, s_theta[(pass * 8 + cc) * 2 + h]);
                const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)um), hi = __builtin_amdgcn_readfirstlane((unsigned)(um >> 32));
                mu[cc] = __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
                acc[cc] = 0.0;
            }
#pragma unroll
            for (int u = 0; u < ZR - 1; ++u) {
#pragma unroll
                for (int cc = 0; cc < 8; ++cc) { const double x = z[u] + mu[cc]; acc[cc] = acc[cc] + x; }
            }
            if (has_last) {
#pragma unroll
                for (int cc = 0; cc < 8; ++cc) { const double x = z[ZR - 1] + mu[cc]; acc[cc] = acc[cc] + x; }
            }
            wave_reduce_step<8, 8, 32>(acc, lane);
            if ((lane & 7) == 0) s_part[(h * 8 + wih) * CT + pass * 8 + (lane >> 3)] = acc[0];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_fetch_add(s_arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        // ---- behind the simulation: the next iteration's randomness, the slowest tile's progress ----
        if (wave == 1) {
            const int c1 = tile * CT + (lane >> 2);
            double* o = s_rng + ((buf ^ 1) * 64 + lane) * 3;
            const double u1 = u01(A.seed, (uint32_t)c1, (uint32_t)(t + 1), 1u + 4u * (uint32_t)(lane & 3));
            const double u2 = u01(A.seed, (uint32_t)c1, (uint32_t)(t + 1), 2u + 4u * (uint32_t)(lane & 3));
            const double rr = sqrt(-2.0 * log(u1));
            o[0] = u01(A.seed, (uint32_t)c1, (uint32_t)(t + 1), 0u);
            o[1] = rr * cos(6.283185307179586 * u2);
            o[2] = rr * sin(6.283185307179586 * u2);
        }
        if (wave == 2) {
            uint32_t m = 0xffffffffu;
            for (int b = lane; b < tiles; b += 64) m = min(m, load4_sys(A.progress + b));
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, off, 64));
            if (lane == 0) __hip_atomic_store(s_minprog, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (ctl) {
            const int cl = lane >> 2, r = lane & 3, c = tile * CT + cl;
            while (__hip_atomic_load(s_arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 16u * (unsigned)t) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            unsigned long long t5 = 0;
            if (tid == 0) t5 = wall_clock();
            // ---- objective, accept, publish ----
            double mk = 0.0, vk = 0.0;
            if (r < 2) {
                double tot = s_part[(r * 8 + 0) * CT + cl];
#pragma unroll
                for (int wv = 1; wv < 8; ++wv) tot = tot + s_part[(r * 8 + wv) * CT + cl];
                mk = tot / (double)A.ns;
                const double d = mk - (r == 0 ? 1.0 : -1.0);
                vk = d * d;
            }
            const double sm0 = __shfl(mk, lane & ~3, 64), sm1 = __shfl(mk, (lane & ~3) + 1, 64);
            const double nv = (__shfl(vk, lane & ~3, 64) + __shfl(vk, (lane & ~3) + 1, 64)) / 2.0;
            double* st = s_state + cl * 8;
            double value = st[0], th0 = st[1], th1 = st[2];
            const double p0 = s_theta[cl * 2], p1 = s_theta[cl * 2 + 1];
            const double uu = s_rng[(buf * 64 + lane) * 3];
            const double atun = 2.0 / (1.0 + 99.0 * (double)c / (double)(N - 1));   // hotter chains accept more
            double prob = 1.0;
            bool accd = true;
            if (t > 1) {
                const double e = exp(atun * (value - nv));
                prob = e < 1.0 ? e : 1.0;
                accd = prob > uu;
            }
            if (accd) { value = nv; th0 = p0; th1 = p1; }
            // the ring entry of iteration t replaces iteration t-K's: every tile must have finished the prologue reads of t-K+1
            if (t >= K && !noex) {
                const unsigned long long ts0 = wall_clock();
                while ((int)__hip_atomic_load(s_minprog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < t - K + 1) {
                    uint32_t m = 0xffffffffu;
                    for (int b = lane; b < tiles; b += 64) m = min(m, load4_sys(A.progress + b));
#pragma unroll
                    for (int off = 32; off >= 1; off >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, off, 64));
                    if (lane == 0) __hip_atomic_store(s_minprog, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (wall_clock() - ts0 > TIMEOUT) { atomicOr(A.err, 4u); break; }
                }
            }
            if (!noex) {
                const uint32_t tag = tag_of(t);
                if (r == 0) store8_sys(A.ring_slot + (size_t)(t % K) * (N + 4) + c,
                                       (unsigned long long)order_key32(value) | ((unsigned long long)((uint32_t)c | (tag << 16)) << 32));
                // record {value, prob, status, th0, th1, sm0, sm1, 0}: lane r stores doubles 2r, 2r+1
                const double e0 = r == 0 ? value : r == 1 ? 1.0 : r == 2 ? th1 : sm1;
                const double e1 = r == 0 ? prob : r == 1 ? th0 : r == 2 ? sm0 : 0.0;
                const unsigned long long a = __builtin_bit_cast(unsigned long long, e0), b = __builtin_bit_cast(unsigned long long, e1);
                uint4* g_ll = A.ring_rec + ((size_t)(t % K) * N + c) * 8;
                const u32x4 q0 = {(unsigned)a, tag, (unsigned)(a >> 32), tag}, q1 = {(unsigned)b, tag, (unsigned)(b >> 32), tag};
                store16_sys(g_ll + 2 * r, q0);
                store16_sys(g_ll + 2 * r + 1, q1);
            }
            if (r == 0) { st[0] = value; st[1] = th0; st[2] = th1; }
            {   // history row (plain stores, nobody reads them in this launch)
                double* hr = A.hist + ((size_t)(t - 1) * N + c) * HW;
                double2 v;
                v.x = r == 0 ? nv : r == 1 ? value : r == 2 ? p0 : sm0;
                v.y = r == 0 ? prob : r == 1 ? (accd ? 1.0 : 0.0) : r == 2 ? p1 : sm1;
                ((double2*)hr)[r] = v;
                if (r < 2) ((double2*)hr)[4 + r] = make_double2(atun, (double)t);
            }
            if (tid == 0) { const unsigned long long t6 = wall_clock(); s_ts[4] += t5 - s_ts[6]; s_ts[5] += t6 - t5; }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the LDS-DMA of the next lists has landed before anybody passes B0)
    }
    if (ctl && (lane & 3) == 0) {
        const int cl = lane >> 2, c = tile * CT + cl;
        const double* st = s_state + cl * 8;
        A.out[(size_t)c * 4] = st[0]; A.out[(size_t)c * 4 + 1] = st[1]; A.out[(size_t)c * 4 + 2] = st[2]; A.out[(size_t)c * 4 + 3] = st[3];
    }
    if (tid < 6) A.ts[(size_t)tile * 8 + tid] = s_ts[tid];
}

int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 200, mode = argc > 2 ? atoi(argv[2]) : 0;
    const int N = 4096, ns = 10000, W = 16, tiles = N / CT, zstride = 10240;
    const int N4 = (N + 3) & ~3;
    // ---- plans: W iterations of N random pairs, their cones per tile ----
    std::mt19937_64 rng(1234);
    std::vector<uint32_t> pairs((size_t)W * tiles * NSUB * 64);
    std::vector<uint16_t> gl((size_t)W * tiles * GCAP, 0);
    const uint32_t dummy = (8u * N4) | ((8u * (N4 + 1)) << 16);
    double sum_pairs = 0, sum_sub = 0, sum_g = 0; int max_sub = 0, max_g = 0, max_pairs = 0;
    for (int w = 0; w < W; ++w) {
        const int Kp = N;
        std::vector<int> pi(Kp), pj(Kp), lvl(Kp), prei(Kp, -1), prej(Kp, -1);
        std::vector<int> last(N, -1), llev(N, 0), first(N, -1);
        for (int q = 0; q < Kp; ++q) {
            int i = (int)(rng() % N), j = (int)(rng() % N);
            while (j == i) j = (int)(rng() % N);
            if (i > j) std::swap(i, j);
            pi[q] = i; pj[q] = j;
            prei[q] = last[i]; prej[q] = last[j];
            lvl[q] = 1 + std::max(llev[i], llev[j]);
            llev[i] = llev[j] = lvl[q];
            last[i] = last[j] = q;
            if (first[i] < 0) first[i] = q;
            if (first[j] < 0) first[j] = q;
        }
        std::vector<std::vector<uint64_t>> need(Kp, std::vector<uint64_t>(tiles / 64, 0));
        for (int cc = 0; cc < N; ++cc) if (last[cc] >= 0) need[last[cc]][(cc / CT) >> 6] |= 1ull << ((cc / CT) & 63);
        for (int q = Kp - 1; q >= 0; --q)
            for (int x = 0; x < tiles / 64; ++x) {
                if (prei[q] >= 0) need[prei[q]][x] |= need[q][x];
                if (prej[q] >= 0) need[prej[q]][x] |= need[q][x];
            }
        for (int b = 0; b < tiles; ++b) {
            std::vector<std::pair<int, int>> mine;   // (level, q)
            for (int q = 0; q < Kp; ++q) if (need[q][b >> 6] >> (b & 63) & 1) mine.push_back({lvl[q], q});
            std::sort(mine.begin(), mine.end());
            uint32_t* op = &pairs[((size_t)w * tiles + b) * NSUB * 64];
            for (int x = 0; x < NSUB * 64; ++x) op[x] = dummy;
            int sub = 0, pos = 0, curl = -1;
            std::vector<char> seen(N, 0);
            int ng = 0;
            uint16_t* og = &gl[((size_t)w * tiles + b) * GCAP];
            for (auto& m : mine) {
                if (m.first != curl) { if (pos) { ++sub; pos = 0; } curl = m.first; }
                if (pos == 64) { ++sub; pos = 0; }
                if (sub >= NSUB) { printf("cone too deep\n"); return 1; }
                const int q = m.second;
                op[sub * 64 + pos++] = (8u * pi[q]) | ((8u * pj[q]) << 16);
                for (int cc : {pi[q], pj[q]})
                    if (!seen[cc]) { seen[cc] = 1; if (cc / CT != b) { if (ng + 8 >= GCAP) { printf("gather list too long\n"); return 1; } og[8 + ng++] = (uint16_t)cc; } }
            }
            if (pos) ++sub;
            og[0] = (uint16_t)sub; og[1] = (uint16_t)ng;
            sum_pairs += mine.size(); sum_sub += sub; sum_g += ng;
            max_sub = std::max(max_sub, sub); max_g = std::max(max_g, ng); max_pairs = std::max(max_pairs, (int)mine.size());
        }
    }
    printf("cones: pairs mean %.1f max %d; sub-levels mean %.1f max %d; gathered chains mean %.1f max %d\n", sum_pairs / (W * tiles), max_pairs,
           sum_sub / (W * tiles), max_sub, sum_g / (W * tiles), max_g);
    std::vector<double> Z((size_t)2 * zstride);
    { std::normal_distribution<double> nd; for (auto& v : Z) v = nd(rng); }
    PArgs A{};
    double* dZ; uint32_t *dp, *dprog, *derr; uint16_t* dg; unsigned long long *dslot, *dts; uint4* drec; double *dhist, *dout;
    CHK(hipMalloc(&dZ, Z.size() * 8)); CHK(hipMemcpy(dZ, Z.data(), Z.size() * 8, hipMemcpyHostToDevice));
    CHK(hipMalloc(&dp, pairs.size() * 4)); CHK(hipMemcpy(dp, pairs.data(), pairs.size() * 4, hipMemcpyHostToDevice));
    CHK(hipMalloc(&dg, gl.size() * 2)); CHK(hipMemcpy(dg, gl.data(), gl.size() * 2, hipMemcpyHostToDevice));
    CHK(hipMalloc(&dslot, (size_t)K * (N + 4) * 8)); CHK(hipMalloc(&drec, (size_t)K * N * 8 * 16));
    CHK(hipMalloc(&dprog, tiles * 4)); CHK(hipMalloc(&derr, 4)); CHK(hipMalloc(&dts, (size_t)tiles * 8 * 8));
    CHK(hipMalloc(&dhist, (size_t)T * N * HW * 8)); CHK(hipMalloc(&dout, (size_t)N * 4 * 8));
    A.Z = dZ; A.cone_pairs = dp; A.cone_gather = dg; A.ring_slot = dslot; A.ring_rec = drec; A.progress = dprog;
    A.hist = dhist; A.out = dout; A.ts = dts; A.err = derr; A.N = N; A.ns = ns; A.zstride = zstride; A.T = T; A.W = W; A.mode = mode; A.seed = 99u;
    const size_t smem = 8 * (size_t)(N4 + 4) + 2 * NSUB * 64 * 4 + 2 * GCAP * 2 + (CT * 2 + 2 * 8 * CT + 2 * 64 * 3 + CT * 8 + 8) * 8 + 16;
    CHK(hipFuncSetAttribute((const void*)k_proto, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    int occ = 0;
    CHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_proto, NWG, smem));
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    printf("LDS %zu B, occupancy %d block(s) per CU, %d CUs, grid %d, mode %d, T %d\n", smem, occ, prop.multiProcessorCount, tiles, mode, T);
    if (occ * prop.multiProcessorCount < tiles) { printf("grid not co-resident\n"); return 1; }
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    std::vector<double> out0((size_t)N * 4), out1((size_t)N * 4);
    for (int rep = 0; rep < 4; ++rep) {
        CHK(hipMemset(dslot, 0, (size_t)K * (N + 4) * 8)); CHK(hipMemset(drec, 0, (size_t)K * N * 8 * 16));
        CHK(hipMemset(dprog, 0, tiles * 4)); CHK(hipMemset(derr, 0, 4));
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_proto, dim3(tiles), dim3(NWG), smem, 0, A);
        CHK(hipEventRecord(e1));
        CHK(hipDeviceSynchronize());
        float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
        uint32_t err = 0; CHK(hipMemcpy(&err, derr, 4, hipMemcpyDeviceToHost));
        std::vector<double>& o = rep == 0 ? out0 : out1;
        CHK(hipMemcpy(o.data(), dout, o.size() * 8, hipMemcpyDeviceToHost));
        std::vector<unsigned long long> ts((size_t)tiles * 8);
        CHK(hipMemcpy(ts.data(), dts, ts.size() * 8, hipMemcpyDeviceToHost));
        double ph[6] = {0, 0, 0, 0, 0, 0};
        for (int b = 0; b < tiles; ++b) for (int i = 0; i < 6; ++i) ph[i] += (double)ts[(size_t)b * 8 + i];
        double mv = 0, ex = 0; for (int cc = 0; cc < N; ++cc) { mv += o[(size_t)cc * 4]; ex += o[(size_t)cc * 4 + 3]; }
        printf("run %d: %.3f ms = %.2f us per iteration = %.1f M chain-evals/s; err %u; mean value %.3e; exchanged fraction %.3f; identical to run 0: %s\n",
               rep, ms, 1e3 * ms / T, 1e-3 * N * (double)T / ms, err, mv / N, ex / ((double)N * T),
               rep == 0 ? "-" : (memcmp(out0.data(), o.data(), o.size() * 8) == 0 ? "yes" : "NO"));
        printf("       phases (us per iteration, mean over tiles): gather %.2f  walk %.2f  donor %.2f  proposal %.2f  simulation %.2f  accept+publish %.2f\n",
               ph[0] / tiles / T / 100.0, ph[1] / tiles / T / 100.0, ph[2] / tiles / T / 100.0, ph[3] / tiles / T / 100.0, ph[4] / tiles / T / 100.0, ph[5] / tiles / T / 100.0);
    }
    return 0;
}
