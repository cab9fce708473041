// smm_rng.hpp — counter-based randomness of libsmmhip (host + gfx950 device code).
//
// Replaces the three random sources of the reference path:
//   probs_acc = rand(n)                      src/mopt/AlgoBGP.jl:85   -> rng_u
//   rand(RAND, MvNormal(mu01, sigma))        src/mopt/AlgoBGP.jl:404  -> rng_prop_normal
//   Random.seed!(1234); rand(MvNormal..,ns)  src/mopt/ObjExamples.jl:74-79 -> rng_Z
//   sample(props, N, replace=false)          src/mopt/AlgoBGP.jl:653-656 -> PairPerm
// Philox4x32-10 (Salmon et al., SC'11) keyed by (seed, stream); Box-Muller for normals.
// Stateless: value = f(seed, chain, iteration, try, index), so chains, shards and restarts
// never share or carry generator state.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

namespace smm {

enum : uint32_t { STREAM_U = 1, STREAM_PROP = 2, STREAM_Z = 3, STREAM_PAIRS = 4 };

struct U4 { uint32_t x, y, z, w; };

__host__ __device__ inline uint32_t mulhi32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}

__host__ __device__ inline U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = mulhi32(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const uint32_t hi1 = mulhi32(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        U4 n;
        n.x = hi1 ^ c.y ^ k0;
        n.y = lo1;
        n.z = hi0 ^ c.w ^ k1;
        n.w = lo0;
        c = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

__host__ __device__ inline U4 philox_stream(uint64_t seed, uint32_t stream, uint32_t c0, uint32_t c1, uint32_t c2,
                                            uint32_t c3) {
    U4 c{c0, c1, c2, c3};
    return philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32) ^ (stream * 0x9E3779B9u));
}

// [0,1): top 53 bits
__host__ __device__ inline double u53(uint32_t hi, uint32_t lo) {
    const uint64_t w = ((uint64_t)hi << 32) | lo;
    return (double)(w >> 11) * 0x1.0p-53;
}
// (0,1]
__host__ __device__ inline double u53_open0(uint32_t hi, uint32_t lo) {
    const uint64_t w = ((uint64_t)hi << 32) | lo;
    return (double)((w >> 11) + 1) * 0x1.0p-53;
}

__host__ __device__ inline void box_muller(const U4& x, double& z0, double& z1) {
    const double u1 = u53_open0(x.x, x.y);
    const double u2 = u53(x.z, x.w);
    const double r = sqrt(-2.0 * log(u1));
    double s, c;
#if defined(__HIP_DEVICE_COMPILE__)
    // angle 2*pi*u2 in units of pi: exact argument reduction, no large-argument (Payne-Hanek) code in the kernel
    sincospi(2.0 * u2, &s, &c);
#else
    const double a = 6.283185307179586476925286766559 * u2;
    s = sin(a);
    c = cos(a);
#endif
    z0 = r * c;
    z1 = r * s;
}

__host__ __device__ inline double rng_u(uint64_t seed, uint32_t chain, uint32_t iter) {
    const U4 x = philox_stream(seed, STREAM_U, chain, iter, 0, 0);
    return u53(x.x, x.y);
}

// both normals of the Philox block that serves parameters 2q and 2q+1
__host__ __device__ inline void rng_prop_normal2(uint64_t seed, uint32_t chain, uint32_t iter, uint32_t tr, uint32_t q,
                                                 double& z0, double& z1) {
    box_muller(philox_stream(seed, STREAM_PROP, chain, iter, tr, q), z0, z1);
}
__host__ __device__ inline double rng_prop_normal(uint64_t seed, uint32_t chain, uint32_t iter, uint32_t tr,
                                                  uint32_t k) {
    double z0, z1;
    rng_prop_normal2(seed, chain, iter, tr, k >> 1, z0, z1);
    return (k & 1) ? z1 : z0;
}

__host__ __device__ inline double rng_Z(uint64_t seed, uint32_t k, uint32_t s) {
    double z0, z1;
    box_muller(philox_stream(seed, STREAM_Z, s, k >> 1, 0, 0), z0, z1);
    return (k & 1) ? z1 : z0;
}

// ---- exchange pairs: keyed bijection of the linear pair index (6-round Feistel, cycle walking) ----
__host__ __device__ inline uint32_t fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}

struct PairPerm {
    uint32_t k[6];
    uint32_t half_bits;
    uint64_t M;

    __host__ __device__ void init(uint64_t seed, uint32_t iter, uint64_t M_) {
        const U4 a = philox_stream(seed, STREAM_PAIRS, iter, 0, 0, 0);
        const U4 b = philox_stream(seed, STREAM_PAIRS, iter, 1, 0, 0);
        k[0] = a.x; k[1] = a.y; k[2] = a.z; k[3] = a.w; k[4] = b.x; k[5] = b.y;
        uint32_t bits = 0;
        while (bits < 62 && ((uint64_t)1 << bits) < M_) ++bits;
        half_bits = (bits + 1) / 2;
        if (half_bits == 0) half_bits = 1;
        M = M_;
    }
    __host__ __device__ uint64_t eval(uint64_t x) const {
        const uint32_t h = half_bits;
        const uint32_t mask = (h >= 32) ? 0xFFFFFFFFu : ((1u << h) - 1u);
        do {
            uint32_t L = (uint32_t)(x >> h) & mask, R = (uint32_t)x & mask;
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                const uint32_t F = fmix32(R + k[r]) & mask;
                const uint32_t nL = R;
                R = L ^ F;
                L = nL;
            }
            x = ((uint64_t)L << h) | R;
        } while (x >= M);
        return x;
    }
};

// linear index m = j(j-1)/2 + i  ->  (i,j), 0 <= i < j
__host__ __device__ inline void pair_unrank(uint64_t m, int32_t& i, int32_t& j) {
    uint64_t jj = (uint64_t)((1.0 + sqrt(1.0 + 8.0 * (double)m)) * 0.5);
    while (jj * (jj - 1) / 2 > m) --jj;
    while ((jj + 1) * jj / 2 <= m) ++jj;
    j = (int32_t)jj;
    i = (int32_t)(m - jj * (jj - 1) / 2);
}

__host__ __device__ inline int n_exchange_pairs(int N) { return N < 3 ? N - 1 : N; }  // AlgoBGP.jl:655

}  // namespace smm
