// tools/launch_floor.hip — what a dependent launch of the chain kernel's geometry costs before it computes anything:
// 256 workgroups x 1024 lanes, 86 KB of dynamic LDS; (a) empty, (b) each workgroup reads the 48 KB (values + pair list) that
// the previous launch wrote (one round trip), (c) the same plus a second dependent 64-byte read per lane group.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(1024) void k(double* vals, const unsigned* pairs, double* recs, int iter, unsigned long long* ts) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x;
    const unsigned long long w0 = wall_clock64();
    if (MODE == 0) { while (wall_clock64() - w0 < 600) __builtin_amdgcn_s_sleep(2); }
    if (MODE >= 1) {
        double v[4]; unsigned p[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[r] = vals[tid + r * 1024]; p[r] = pairs[tid + r * 1024]; }
#pragma unroll
        for (int r = 0; r < 4; ++r) { ((double*)lds)[2 * (tid + r * 1024)] = v[r]; ((unsigned*)(lds + 65536))[tid + r * 1024] = p[r]; }
        __syncthreads();
        double acc = ((double*)lds)[2 * ((tid * 37) & 4095)];
        if (MODE >= 2 && tid < 64) {
            const int s = (int)(((unsigned*)(lds + 65536))[(blockIdx.x * 16 + (tid >> 2)) & 4095] & 4095u);
            acc += recs[(size_t)s * 8 + (tid & 3) * 2];
        }
        while (wall_clock64() - w0 < 600) __builtin_amdgcn_s_sleep(2);   // ~6 us of "work" from kernel entry: the device, not the host, paces the loop
        if (tid < 16) vals[blockIdx.x * 16 + tid] = acc * 0.5 + iter;   // next launch's values
        if (MODE >= 2 && tid < 64) recs[(size_t)(blockIdx.x * 16 + (tid >> 2)) * 8 + (tid & 3) * 2] = acc;
        if (ts && tid == 0) { ts[blockIdx.x * 2] = w0; ts[blockIdx.x * 2 + 1] = wall_clock64(); }
    }
}

template <int MODE>
void run(const char* what, double* vals, unsigned* pairs, double* recs) {
    CHK(hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 88064));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int i = 0; i < 100; ++i) k<MODE><<<256, 1024, 88064>>>(vals, pairs, recs, i, nullptr);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    for (int i = 0; i < 2000; ++i) k<MODE><<<256, 1024, 88064>>>(vals, pairs, recs, i, nullptr);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-70s %6.2f us per dependent launch\n", what, ms * 1e3 / 2000);
    // the same 2000 launches as ONE graph of kernel nodes (captured from the stream), launched once
    hipStream_t st;
    CHK(hipStreamCreate(&st));
    hipGraph_t g; hipGraphExec_t ge;
    CHK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 2000; ++i) k<MODE><<<256, 1024, 88064, st>>>(vals, pairs, recs, i, nullptr);
    CHK(hipStreamEndCapture(st, &g));
    CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CHK(hipGraphLaunch(ge, st));
    CHK(hipStreamSynchronize(st));
    CHK(hipEventRecord(e0, st));
    CHK(hipGraphLaunch(ge, st));
    CHK(hipEventRecord(e1, st));
    CHK(hipEventSynchronize(e1));
    CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-70s %6.2f us per launch as a node of one graph\n", "", ms * 1e3 / 2000);
    CHK(hipGraphExecDestroy(ge)); CHK(hipGraphDestroy(g)); CHK(hipStreamDestroy(st));
}

int main() {
    double *vals, *recs; unsigned* pairs;
    CHK(hipMalloc(&vals, 4096 * 8)); CHK(hipMalloc(&recs, 4096 * 64)); CHK(hipMalloc(&pairs, 4096 * 4));
    CHK(hipMemset(vals, 0, 4096 * 8)); CHK(hipMemset(recs, 0, 4096 * 64));
    unsigned h[4096];
    srand(1);
    for (auto& x : h) x = rand();
    CHK(hipMemcpy(pairs, h, sizeof h, hipMemcpyHostToDevice));
    run<0>("empty kernel", vals, pairs, recs);
    run<1>("+ every workgroup stages 32 KB values (just written) + 16 KB pairs", vals, pairs, recs);
    run<2>("+ one dependent 64-byte record read per chain after that", vals, pairs, recs);
    return 0;
}
