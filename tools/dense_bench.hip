// tools/dense_bench.hip — the two products of the dense objective (BASELINE config 5: x = B theta, h = tanh x, y = A h on
// v_mfma_f64_16x16x4) in isolation, as a persistent tile would run them: 256 workgroups of 512 lanes, 16 chains each, ITER evaluations
// per launch.  What does an evaluation cost when the operand fragments come out of L2 every time (what smm_chain.hpp's dense_tile_n
// does), when they stay in the wave's registers for the whole launch, and how much of either is the tanh?
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/dense_bench tools/dense_bench.hip && tools/dense_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int WG = 512, DENSE_D = 256, NP = 50, NM = 50, PS = 16, NPS = 13, NOT = 4;
typedef double d4_t __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) double* gptr_t;

struct Args { const double* Bf; const double* Af; const double* theta; double* out; int iters; };

// tanh by one exponential and one division: E = exp(2|x|) = 2^n (1 + p), p = expm1(r) on |r| <= ln2 / 2 (Taylor to r^13), tanh = (E - 1) / (E + 1)
// with E -+ 1 = fma(2^n, p, 2^n -+ 1) (2^n -+ 1 exact): only correctly rounded operations, so a C restatement is bit-identical
__device__ __forceinline__ double tanh_fast(const double x) {
    const double ax = __builtin_fabs(x);
    const double z = ax + ax;
    const double n = __builtin_rint(z * 1.44269504088896338700e+00);
    double r = __builtin_fma(-n, 6.93147180369123816490e-01, z);
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
    const double s = __builtin_ldexp(1.0, (int)n);
    const double em1 = __builtin_fma(s, p, s - 1.0), ep1 = __builtin_fma(s, p, s + 1.0);
    const double t = ax >= 19.0625 ? 1.0 : em1 / ep1;
    return __builtin_copysign(t, x);
}

// MODE bit 0: operands resident (loaded once); bit 1: no tanh; bit 2: tanh_fast and both first products ahead of the first tanh
template <int MODE>
__global__ __launch_bounds__(WG) void k_dense(const Args A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    double* s_theta = (double*)lds;                 // [16][NP]
    double* s_part = s_theta + 16 * NP;             // [8][64][16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    for (int i = tid; i < 16 * NP; i += WG) s_theta[i] = A.theta[(size_t)blockIdx.x * 16 * NP + i];
    __syncthreads();
    const gptr_t dense_Bf = (gptr_t)A.Bf, dense_Af = (gptr_t)A.Af;
    constexpr bool RES = (MODE & 1) != 0, NOTANH = (MODE & 2) != 0, FAST = (MODE & 4) != 0;
    double bfr[2][PS], afr[2][4][4];
    auto load_ops = [&]() {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const gptr_t bf = dense_Bf + (size_t)(2 * wave + tt) * NPS * 64 + lane;
#pragma unroll
            for (int s = 0; s < PS; ++s) bfr[tt][s] = bf[(size_t)(s < NPS - 1 ? s : NPS - 1) * 64];
        }
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int T = 2 * wave + tt;
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const gptr_t af = dense_Af + ((size_t)(o * (DENSE_D / 16) + T) * 4) * 64 + lane;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) afr[tt][o][s4] = af[s4 * 64];
            }
        }
    };
    if (RES) load_ops();
    double keep = 0.0;
    for (int it = 0; it < A.iters; ++it) {
        if (!RES) { asm volatile("" ::: "memory"); load_ops(); }
        d4_t yacc[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) yacc[o] = d4_t{0.0, 0.0, 0.0, 0.0};
        double th[PS];
#pragma unroll
        for (int s = 0; s < PS; ++s) {
            const int p = 4 * s + lk;
            const double v = s_theta[li * NP + (p < NP ? p : NP - 1)];
            th[s] = p < NP ? v : 0.0;
        }
        if (FAST) {
            d4_t xacc[2];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                xacc[tt] = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s = 0; s < PS; ++s) xacc[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(bfr[tt][s], th[s], xacc[tt], 0, 0, 0);
            }
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                double h[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) h[r] = tanh_fast(xacc[tt][r]);
#pragma unroll
                for (int o = 0; o < 4; ++o)
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) yacc[o] = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[tt][o][s4], h[s4], yacc[o], 0, 0, 0);
            }
        } else
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            d4_t xacc = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < PS; ++s) xacc = __builtin_amdgcn_mfma_f64_16x16x4f64(bfr[tt][s], th[s], xacc, 0, 0, 0);
            double h[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) h[r] = NOTANH ? xacc[r] * 0.001 : tanh(xacc[r]);
#pragma unroll
            for (int o = 0; o < 4; ++o)
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) yacc[o] = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[tt][o][s4], h[s4], yacc[o], 0, 0, 0);
        }
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int r = 0; r < 4; ++r) s_part[((size_t)wave * 64 + 16 * o + lk + 4 * r) * 16 + li] = yacc[o][r];
        __syncthreads();
        // the 8 wave partials of (k, chain) added left to right; the proposal of the next evaluation depends on them
        if (tid < 16 * NP) {
            const int c = tid / NP, k = tid - c * NP;
            double y = 0.0;
            for (int w = 0; w < 8; ++w) y += s_part[((size_t)w * 64 + k) * 16 + c];
            keep += y;
            s_theta[tid] = s_theta[tid] + 1e-9 * y;
        }
        __syncthreads();
    }
    if (tid < 16 * NP) A.out[(size_t)blockIdx.x * 16 * NP + tid] = keep;
}

int main() {
    const int tiles = 256, iters = 400;
    std::vector<double> B((size_t)16 * NPS * 64), Af((size_t)NOT * 16 * 4 * 64), th((size_t)tiles * 16 * NP);
    srand(1);
    for (auto& v : B) v = (rand() / (double)RAND_MAX - 0.5) * 0.3;
    for (auto& v : Af) v = (rand() / (double)RAND_MAX - 0.5) * 0.3;
    for (auto& v : th) v = rand() / (double)RAND_MAX;
    Args A;
    double *dB, *dA, *dth, *dout;
    CHK(hipMalloc(&dB, B.size() * 8)); CHK(hipMalloc(&dA, Af.size() * 8)); CHK(hipMalloc(&dth, th.size() * 8)); CHK(hipMalloc(&dout, th.size() * 8));
    CHK(hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice)); CHK(hipMemcpy(dA, Af.data(), Af.size() * 8, hipMemcpyHostToDevice));
    CHK(hipMemcpy(dth, th.data(), th.size() * 8, hipMemcpyHostToDevice));
    A.Bf = dB; A.Af = dA; A.theta = dth; A.out = dout; A.iters = iters;
    const size_t smem = (size_t)(16 * NP + 8 * 64 * 16) * 8;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const char* names[6] = {"operands out of L2 every evaluation", "operands resident in registers", "out of L2, no tanh", "resident, no tanh", "out of L2, tanh_fast, products ahead", "resident, tanh_fast, products ahead"};
    std::vector<double> ref(th.size()), got(th.size());
    for (int mode = 0; mode < 6; ++mode) {
        void (*k)(const Args) = mode == 0 ? k_dense<0> : mode == 1 ? k_dense<1> : mode == 2 ? k_dense<2> : mode == 3 ? k_dense<3> : mode == 4 ? k_dense<4> : k_dense<5>;
        CHK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            CHK(hipEventRecord(e0));
            hipLaunchKernelGGL(k, dim3(tiles), dim3(WG), smem, 0, A);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        CHK(hipMemcpy(got.data(), dout, got.size() * 8, hipMemcpyDeviceToHost));
        if (mode == 0) ref = got;
        size_t diff = 0;
        if (mode == 1) for (size_t i = 0; i < got.size(); ++i) diff += got[i] != ref[i];
        double rel = 0.0;
        if (mode >= 4) for (size_t i = 0; i < got.size(); ++i) { const double e = fabs(got[i] - ref[i]) / (fabs(ref[i]) + 1e-300); if (e > rel) rel = e; }
        if (mode >= 4) printf("   (largest relative difference of the accumulated outputs from mode 0: %.2e)\n", rel);
        const double us = best * 1e3 / iters;
        printf("%-40s %7.2f us per evaluation of 256 tiles  (%.1f TFLOP/s executed MFMA)%s\n", names[mode], us, 256.0 * 464 * 2048 / us * 1e-6,
               mode == 1 ? (diff ? "  DIFFERS from mode 0" : "  bit-identical to mode 0") : "");
    }
    return 0;
}
