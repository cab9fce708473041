// tools/barrier_bench.hip — what does one level of the exchange walk cost at least?  s_barrier alone, and with one dependent
// LDS read-compare-write in front of it, by workgroup size.   hipcc --offload-arch=gfx950 -O3 -o tools/barrier_bench tools/barrier_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void k(unsigned long long* out, const unsigned* idx, int iters, int active) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x;
    u32x4_t* slot = (u32x4_t*)lds;
    for (int g = tid; g < 4096; g += blockDim.x) slot[g] = u32x4_t{(unsigned)g * 2654435761u, 0x3ff00000u | (g & 0xffff), (unsigned)g, 0u};
    __syncthreads();
    unsigned i = idx[tid] & 4095, j = idx[tid + 1024] & 4095;
    const unsigned long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE >= 1 && MODE <= 3 && tid < active) {
            const unsigned ai = i * 16, aj = j * 16;
            u32x4_t si, sj;
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(si), "=&v"(sj) : "v"(ai), "v"(aj) : "memory");
            const double vi = __builtin_bit_cast(double, ((unsigned long long)si.y << 32) | si.x);
            const double vj = __builtin_bit_cast(double, ((unsigned long long)sj.y << 32) | sj.x);
            if (MODE >= 2 && vi - vj > 0.0) {
                u32x4_t ni = sj, nj = si;
                ni.w = j + 1; nj.w = i + 1;
                asm volatile("ds_write_b128 %0, %2\n\tds_write_b128 %1, %3" :: "v"(ai), "v"(aj), "v"(ni), "v"(nj) : "memory");
            }
            i = (i + 37 + si.z) & 4095; j = (j + 101 + sj.z) & 4095;
        }
        if (MODE >= 4 && MODE <= 6 && tid < active) {   // 4-byte slots {src | key << 16}, 2-byte partner array behind them
            const unsigned ai = i * 4, aj = j * 4;
            unsigned si, sj;
            asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(si), "=&v"(sj) : "v"(ai), "v"(aj) : "memory");
            if ((si >> 16) > (sj >> 16)) {
                asm volatile("ds_write_b32 %0, %2\n\tds_write_b32 %1, %3" :: "v"(ai), "v"(aj), "v"(sj), "v"(si) : "memory");
                if (MODE == 4) asm volatile("ds_write_b16 %0, %2 offset:16384\n\tds_write_b16 %1, %3 offset:16384" :: "v"(i * 2), "v"(j * 2), "v"(j + 1), "v"(i + 1) : "memory");
            }
            i = (i + 37 + (si & 0xfff)) & 4095; j = (j + 101 + (sj & 0xfff)) & 4095;
        }
        if (MODE == 7 && tid < active) {   // 8-byte slots {key32, src | partner << 16}
            const unsigned ai = i * 8, aj = j * 8;
            typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
            u32x2_t si, sj;
            asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(si), "=&v"(sj) : "v"(ai), "v"(aj) : "memory");
            if (si.x > sj.x) {
                u32x2_t ni = sj, nj = si;
                ni.y = (ni.y & 0xffffu) | ((j + 1) << 16); nj.y = (nj.y & 0xffffu) | ((i + 1) << 16);
                asm volatile("ds_write_b64 %0, %2\n\tds_write_b64 %1, %3" :: "v"(ai), "v"(aj), "v"(ni), "v"(nj) : "memory");
            }
            i = (i + 37 + (si.y & 0xfff)) & 4095; j = (j + 101 + (sj.y & 0xfff)) & 4095;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (MODE != 3 && MODE != 6) __syncthreads();
    }
    const unsigned long long c1 = clock64();
    if (tid == 0) out[blockIdx.x] = c1 - c0;
    if (tid == 1) out[1024 + blockIdx.x] = i + j;
}

template <int MODE>
void run(const char* what, int block, int active, unsigned long long* d, unsigned* didx) {
    const int iters = 200;
    k<MODE><<<256, block, 65536>>>(d, didx, iters, active);
    k<MODE><<<256, block, 65536>>>(d, didx, iters, active);
    CHK(hipDeviceSynchronize());
    unsigned long long h[256];
    CHK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
    double s = 0;
    for (int i = 0; i < 256; ++i) s += (double)h[i];
    printf("%-52s block %4d active %4d: %7.1f cycles per level\n", what, block, active, s / 256 / iters);
}

int main() {
    unsigned long long* d; unsigned* didx;
    CHK(hipMalloc(&d, 2048 * 8)); CHK(hipMalloc(&didx, 2048 * 4));
    unsigned h[2048];
    srand(3);
    for (auto& x : h) x = rand();
    CHK(hipMemcpy(didx, h, sizeof h, hipMemcpyHostToDevice));
    CHK(hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CHK(hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CHK(hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CHK(hipFuncSetAttribute((const void*)k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CHK(hipFuncSetAttribute((const void*)k<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CHK(hipFuncSetAttribute((const void*)k<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CHK(hipFuncSetAttribute((const void*)k<7>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CHK(hipFuncSetAttribute((const void*)k<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    for (int block : {1024, 512, 256, 64}) {
        run<0>("barrier only", block, block, d, didx);
        run<1>("2 x ds_read_b128 + wait + barrier", block, block, d, didx);
        run<2>("read, compare, 2 x ds_write_b128 (half swap), barrier", block, block, d, didx);
        if (block >= 128) run<2>("the same, one wave's worth of pairs", block, 64, d, didx);
    }
    run<3>("one wave, no barrier: read, compare, write, wait", 64, 64, d, didx);
    for (int active : {1024, 512, 256, 128, 64}) {
        run<2>("16-byte slots", 1024, active, d, didx);
        run<4>("4-byte slots + 2-byte partner writes", 1024, active, d, didx);
        run<5>("4-byte slots, no partner writes", 1024, active, d, didx);
        run<7>("8-byte slots", 1024, active, d, didx);
    }
    run<6>("4-byte slots, one wave, no barrier", 64, 64, d, didx);
    return 0;
}
