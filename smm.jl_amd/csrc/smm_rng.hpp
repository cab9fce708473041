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

// ------------------------------------------------------------------------------------------
// The elementary functions of the path are part of its NUMERICAL CONTRACT (include/smmhip.h): the logarithm and the sine / cosine of
// Box-Muller, the exponential of the acceptance probability (AlgoBGP.jl:344).  Plain sequences of correctly rounded operations (+ - * /,
// rint, ldexp; compiled without contraction), so that every implementation of the contract — device, host, a C restatement — produces
// the same bits; each within 1 ulp of the true value (sine and cosine: within 2^-53 absolute).  After fdlibm's e_log.c, k_sin.c, k_cos.c
// (Sun Microsystems 1993, freely distributable); the exponential: range reduction + Taylor by fma: the argument reductions are exact here because of what the arguments are.
// ------------------------------------------------------------------------------------------
// log of a positive normal double
__host__ __device__ inline double smm_log(const double x) {
    const uint64_t b = __builtin_bit_cast(uint64_t, x);
    int e = (int)((b >> 52) & 0x7ffu) - 1023;
    double m = __builtin_bit_cast(double, (b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);
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
// sine and cosine of 2 pi u, u in [0, 1): the quadrant is exact (t = 4 u, q = rint(t), r = t - q with |r| <= 1/2)
__host__ __device__ inline void smm_sincos2pi(const double u, double& sn, double& cs) {
    const double t = 4.0 * u;
    const double q = __builtin_rint(t);
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
    // quadrant qi: (sin, cos) = (s, c), (c, -s), (-s, -c), (-c, s) — one swap, two sign bits (no chain of selects)
    const uint32_t qi = (uint32_t)(int)q & 3u;
    const bool odd = (qi & 1u) != 0u;
    const uint64_t sb = __builtin_bit_cast(uint64_t, odd ? c : s) ^ ((uint64_t)(qi & 2u) << 62);
    const uint64_t cb = __builtin_bit_cast(uint64_t, odd ? s : c) ^ ((uint64_t)((qi + 1u) & 2u) << 62);
    sn = __builtin_bit_cast(double, sb);
    cs = __builtin_bit_cast(double, cb);
}
// exp of any double: a NaN stays one, overflow to +inf, the subnormal results through ldexp's rounding; no division (it sits on the
// accept step's critical path: fdlibm's form with its quotient cost banana at 8192 chains 2 %)
__host__ __device__ inline double smm_exp(const double x) {
    if (x != x) return x;
    if (x > 709.782712893383973096) return __builtin_huge_val();
    if (x < -745.13321910194110842) return 0.0;
    const double k = __builtin_rint(x * 1.44269504088896338700e+00);
    double r = __builtin_fma(-k, 6.93147180369123816490e-01, x);
    r = __builtin_fma(-k, 1.90821492927058770002e-10, r);
    // exp(r) - 1 = r + r^2 q(r), q = the Taylor coefficients 1/2! .. 1/13! (|r| <= ln2 / 2: 4e-18), Estrin's scheme: four levels of fma
    const double r2 = r * r, r4 = r2 * r2, r8 = r4 * r4;
    const double a0 = __builtin_fma(1.0 / 6.0, r, 0.5), a1 = __builtin_fma(1.0 / 120.0, r, 1.0 / 24.0), a2 = __builtin_fma(1.0 / 5040.0, r, 1.0 / 720.0);
    const double a3 = __builtin_fma(1.0 / 362880.0, r, 1.0 / 40320.0), a4 = __builtin_fma(1.0 / 39916800.0, r, 1.0 / 3628800.0), a5 = __builtin_fma(1.0 / 6227020800.0, r, 1.0 / 479001600.0);
    const double b0 = __builtin_fma(a1, r2, a0), b1 = __builtin_fma(a3, r2, a2), b2 = __builtin_fma(a5, r2, a4);
    const double q = __builtin_fma(b2, r8, __builtin_fma(b1, r4, b0));
    const double p = __builtin_fma(r2, q, r);
    return __builtin_ldexp(1.0 + p, (int)k);
}

// the transform itself, OUT OF LINE in device code: inlined into the persistent kernels (k_chain_persist_loc<2, true, true>), the logarithm next to
// the sine / cosine makes this compiler (ROCm 7.2) emit an instruction its own verifier refuses ("Operand has incorrect register class")
#if defined(__HIP_DEVICE_COMPILE__)
#define SMM_NOINLINE_DEV __attribute__((noinline))
#else
#define SMM_NOINLINE_DEV
#endif
struct BM2 { double z0, z1; };   // (returned by value: in registers — results by reference would travel through scratch memory)
__host__ __device__ SMM_NOINLINE_DEV inline BM2 box_muller_u(const double u1, const double u2) {
    const double r = __builtin_sqrt(-2.0 * smm_log(u1));   // (IEEE square root: correctly rounded everywhere)
    double s, c;
    smm_sincos2pi(u2, s, c);
    return BM2{r * c, r * s};
}
__host__ __device__ inline void box_muller(const U4& x, double& z0, double& z1) {
    const BM2 z = box_muller_u(u53_open0(x.x, x.y), u53(x.z, x.w));
    z0 = z.z0; z1 = z.z1;
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
