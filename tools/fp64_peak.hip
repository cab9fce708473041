// FP64 VALU add-rate calibration for gfx950 (the in-container guide lists no FP64
// vector peak).  Every lane runs ILP independent dependent-add chains: x = x + a.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int ILP, bool FMA>
__global__ __launch_bounds__(256) void k(double* out, double a, int iters) {
    double x[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) x[i] = threadIdx.x * 1e-9 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) x[i] = FMA ? __builtin_fma(x[i], a, a) : (x[i] + a);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ILP, bool FMA>
void run(int blocks_per_cu, int iters) {
    double* d;
    const int grid = 256 * blocks_per_cu;
    (void)hipMalloc(&d, (size_t)grid * 256 * 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<ILP, FMA><<<grid, 256>>>(d, 1.0000001, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<ILP, FMA><<<grid, 256>>>(d, 1.0000001, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)grid * 256 * ILP * iters;
    printf("%s ILP=%2d waves/SIMD=%d  iters=%d : %.3f ms  %.2f T %s/s  (%.2f TFLOP/s)\n", FMA ? "fma" : "add", ILP,
           blocks_per_cu, iters, ms, ops / ms / 1e9, FMA ? "fma" : "add", ops * (FMA ? 2 : 1) / ms / 1e9);
    (void)hipFree(d);
}

int main() {
    for (int bpc : {1, 2, 4}) {
        run<8, false>(bpc, 20000);
        run<16, false>(bpc, 20000);
        run<16, true>(bpc, 20000);
    }
    // short kernel comparable to one chain iteration: 1250 adds per lane, 2 waves/SIMD
    run<16, false>(2, 78);
    return 0;
}
