// tools/dense2_bench.hip — the dense objective's tile function (smm_chain.hpp: dense_tile_v, spec v1 and v2 = SMM_OBJ_DENSE2 with its 256 x 256
// stage) in isolation, THE LIBRARY'S OWN CODE (the headers are included as smmhip.hip includes them): 256 workgroups of 512 lanes, 16 chains each,
// ITER evaluations per launch, operands out of L2 every time — what one evaluation costs inside the persistent tile kernel, without the rest of
// the iteration.  The matrix pipe's floor: 1536 (v2) / 512 (v1) v_mfma_f64_16x16x4 per tile at 64 cycles each, two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -o tools/dense2_bench tools/dense2_bench.hip && tools/dense2_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <algorithm>
#include <vector>

#include "../include/smmhip.h"
#include "../smm.jl_amd/csrc/smm_rng.hpp"
namespace {
using namespace smm;
#include "../smm.jl_amd/csrc/smm_params.hpp"
#include "../smm.jl_amd/csrc/smm_walk_lean.hpp"
#include "../smm.jl_amd/csrc/smm_propose.hpp"
#include "../smm.jl_amd/csrc/smm_chain.hpp"
}

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Args { const double *Bf, *Af, *A2f, *theta; double* out; int np, nOt, iters; };

__global__ __launch_bounds__(WG) void k_dense2(const Args A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    double* s_theta = (double*)lds;                               // [16][np]
    double* s_part = s_theta + ((16 * A.np + 1) & ~1);            // [8][nOt*16][16] (>= 4096 doubles)
    const int tid = threadIdx.x;
    for (int i = tid; i < 16 * A.np; i += WG) s_theta[i] = A.theta[(size_t)blockIdx.x * 16 * A.np + i];
    __syncthreads();
    double acc = 0.0;
    for (int it = 0; it < A.iters; ++it) {
        dense_tile_v<16>(A.np, A.nOt, A.Bf, A.Af, A.A2f, (uint32_t)((unsigned char*)s_theta - lds), (uint32_t)((unsigned char*)s_part - lds), tid);
        __syncthreads();
        acc += s_part[tid];       // (consume: keeps every evaluation alive)
        __syncthreads();
        if (tid < 16) s_theta[tid * A.np] += 1e-9;
        __syncthreads();
    }
    A.out[(size_t)blockIdx.x * WG + tid] = acc;
}

// the matrix pipe alone: NACC accumulator chains per wave, operands in registers, nothing else in the loop
template <int NACC>
__global__ __launch_bounds__(WG) void k_pure_mfma(double* out, const int n, const double a0, const double b0) {
    typedef double d4 __attribute__((ext_vector_type(4)));
    d4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
    const double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double r = 0.0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int main(int argc, char** argv) {
    const int np = 50, nm = 50, nOt = (nm + 15) / 16, nPs = (np + 3) / 4, tiles = 256, iters = argc > 1 ? atoi(argv[1]) : 200;
    std::vector<double> Bf((size_t)16 * nPs * 64), Af((size_t)nOt * 16 * 4 * 64), A2f((size_t)256 * 256), th((size_t)tiles * 16 * np);
    srand(1);
    auto rnd = [] { return (rand() / (double)RAND_MAX - 0.5) * 0.25; };
    for (auto& v : Bf) v = rnd();
    for (auto& v : Af) v = rnd();
    for (auto& v : A2f) v = rnd();
    for (auto& v : th) v = rnd();
    double *dB, *dA, *dA2, *dth, *dout;
    CHK(hipMalloc(&dB, Bf.size() * 8)); CHK(hipMalloc(&dA, Af.size() * 8)); CHK(hipMalloc(&dA2, A2f.size() * 8)); CHK(hipMalloc(&dth, th.size() * 8));
    CHK(hipMalloc(&dout, (size_t)tiles * WG * 8));
    CHK(hipMemcpy(dB, Bf.data(), Bf.size() * 8, hipMemcpyHostToDevice)); CHK(hipMemcpy(dA, Af.data(), Af.size() * 8, hipMemcpyHostToDevice));
    CHK(hipMemcpy(dA2, A2f.data(), A2f.size() * 8, hipMemcpyHostToDevice)); CHK(hipMemcpy(dth, th.data(), th.size() * 8, hipMemcpyHostToDevice));
    const size_t smem = (size_t)(((16 * np + 1) & ~1) + 8 * nOt * 16 * 16) * 8;
    CHK(hipFuncSetAttribute((const void*)k_dense2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    {   // calibration: what the FP64 matrix pipe sustains on this device (clock included): 256 workgroups, 1 / 2 waves per SIMD, 1 / 2 / 4 chains per wave
        const int n = 4000;
        auto run = [&](auto kern, int nacc, int threads) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                CHK(hipEventRecord(e0));
                hipLaunchKernelGGL(kern, dim3(tiles), dim3(threads), 0, 0, dout, n, 0.5, 0.25);
                CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
                float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
            }
            const double mf = (double)tiles * (threads / 64) * n * 8 * nacc;
            printf("pure v_mfma_f64_16x16x4: %d waves per SIMD, %d chains per wave: %.1f TFLOP/s = %.3f of 78.6 (%.1f cycles per MFMA and SIMD at 2.4 GHz)\n", threads / 256, nacc,
                   mf * 2048 / best / 1e9, mf * 2048 / best / 1e9 / 78.6, best * 1e-3 * 2.4e9 / (mf / tiles / 4));
        };
        run(k_pure_mfma<1>, 1, 256); run(k_pure_mfma<2>, 2, 256); run(k_pure_mfma<4>, 4, 256);
        run(k_pure_mfma<1>, 1, 512); run(k_pure_mfma<2>, 2, 512);
    }
    for (int v2 = 0; v2 < 2; ++v2) {
        Args A{dB, dA, v2 ? dA2 : nullptr, dth, dout, np, nOt, iters};
        for (int rep = 0; rep < 3; ++rep) {
            CHK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_dense2, dim3(tiles), dim3(WG), smem, 0, A);
            CHK(hipEventRecord(e1));
            CHK(hipEventSynchronize(e1));
            float ms = 0;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / iters, mfma = v2 ? 1536.0 : 512.0;
            if (rep == 2)
                printf("spec v%d: %.2f us per evaluation of 256 tiles (%d evaluations); matrix pipe floor %.2f us at 2.4 GHz; executed %.1f TFLOP/s = %.3f of 78.6\n", v2 + 1, us, iters,
                       mfma / 8 * 2 * 64 / 2400.0, mfma * 2048 * tiles / us / 1e6, mfma * 2048 * tiles / us / 1e6 / 78.6);
        }
    }
    return 0;
}
