// tools/sim_bench.hip — the simulation phase of objfunc_norm in isolation (no proposal, no accept step, no exchange):
// which arrangement of 16 chains x 2 moments x 10000 draws on one CU issues the 640 FP64 adds per lane fastest?
// Every variant computes the same lane-strided partial sums (numerical contract) and writes the wave totals.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/sim_bench tools/sim_bench.hip && tools/sim_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int WG = 512;
constexpr int NS = 10000, NM = 2;

struct Args {
    const double* Z;      // [NM][zstride]
    const double* theta;  // [nchains][NM]
    double* out;          // [nchains][NM][8 waves]
    unsigned long long* ts;  // [grid][4]: realtime start, realtime end, memtime start, memtime end
    int zstride;
    int mode;             // 1: every chunk re-reads chunk 0 (L1 resident); 2: no loads at all
    int chmask;           // mode 1: 0, else ~0
};

// transposed halving tree (the library's wave_reduce_transposed)
template <int CT, int NN, int OFF>
__device__ inline void wave_reduce_step(double (&a)[CT], int lane) {
    if constexpr (NN > 1) {
        const bool upper = (lane & OFF) != 0;
#pragma unroll
        for (int i = 0; i < NN / 2; ++i) {
            const double mine = upper ? a[i + NN / 2] : a[i];
            const double send = upper ? a[i] : a[i + NN / 2];
            const double recv = __shfl_xor(send, OFF, 64);
            a[i] = mine + recv;
        }
        wave_reduce_step<CT, NN / 2, OFF / 2>(a, lane);
    } else if constexpr (OFF >= 1) {
        a[0] = a[0] + __shfl_xor(a[0], OFF, 64);
        wave_reduce_step<CT, 1, OFF / 2>(a, lane);
    }
}
template <int NACC>
__device__ inline double wave_reduce_t(double (&a)[NACC], int lane) {
    wave_reduce_step<NACC, NACC, 32>(a, lane);
    return a[0];
}

// shocks through a buffer descriptor: row = scalar byte offset, lane = one constant 32-bit vector offset (as the library does)
struct ZBuf {
    __amdgpu_buffer_rsrc_t rsrc;
    int lane_off;
    __device__ inline void init(const Args& A, int l) {
        rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)A.Z, 0, (int)((size_t)NM * A.zstride * sizeof(double)), 0x00020000);
        lane_off = l * 8;
    }
};
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
template <int MODE, int ZU>
__device__ inline void load_chunk(const Args& A, const ZBuf& zb, int k, int ch, int l, double (&z)[ZU]) {
    const int row0 = (k * A.zstride + (ch & A.chmask) * (ZU * WG)) * 8;
#pragma unroll
    for (int u = 0; u < ZU; ++u)
        z[u] = (MODE == 2) ? (double)(u + l + ch) : __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(zb.rsrc, zb.lane_off, row0 + u * WG * 8, 0));
}

// one moment k for CT chains by the 512 lanes (l = lane id 0..511) of a group
template <int CT, int MODE, int ZU>
__device__ inline void sim_moment(const Args& A, const ZBuf& zb, int k, int knext, int l, const double (&mu)[CT], double (&acc)[CT], double (&zc)[ZU]) {
    const int nch = (NS + ZU * WG - 1) / (ZU * WG);
    const int last_draws = NS - (nch - 1) * (ZU * WG);
    double zn[ZU];
    auto add_full = [&](const double (&z)[ZU]) {
#pragma unroll
        for (int u = 0; u < ZU; ++u)
#pragma unroll
            for (int c = 0; c < CT; ++c) { const double x = z[u] + mu[c]; acc[c] = acc[c] + x; }
    };
    auto add_last = [&](const double (&z)[ZU]) {
#pragma unroll
        for (int u = 0; u < ZU; ++u)
            if (l + u * WG < last_draws) {
#pragma unroll
                for (int c = 0; c < CT; ++c) { const double x = z[u] + mu[c]; acc[c] = acc[c] + x; }
            }
    };
    int ch = 0;
#pragma clang loop unroll(disable)
    for (; ch + 2 <= nch; ch += 2) {
        load_chunk<MODE, ZU>(A, zb, k, ch + 1, l, zn);
        add_full(zc);
        const bool last = (ch + 2 == nch);
        load_chunk<MODE, ZU>(A, zb, last ? knext : k, last ? 0 : ch + 2, l, zc);
        if (last) add_last(zn); else add_full(zn);
    }
    if (ch < nch) {
        load_chunk<MODE, ZU>(A, zb, knext, 0, l, zn);
        add_last(zc);
#pragma unroll
        for (int u = 0; u < ZU; ++u) zc[u] = zn[u];
    }
}

#define STAMP0() unsigned long long rt0 = wall_clock64(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); unsigned long long rt1 = wall_clock64(), mt0 = clock64(), rt2 = 0
#define STAMPMID() rt2 = wall_clock64()
#define STAMP1() do { const unsigned long long rt3 = wall_clock64(), mt1 = clock64(); if ((threadIdx.x & 63) == 0) { unsigned long long* t = A.ts + ((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 6; t[0] = rt0; t[1] = rt1; t[2] = rt2; t[3] = rt3; t[4] = mt0; t[5] = mt1; } } while (0)

// A: two tiles of 8 chains in a 1024-thread workgroup, every lane does both moments (the round-1 arrangement)
template <int MODE, int ZU>
__global__ __launch_bounds__(1024, 4) void k_tiles2x8(const Args A) {
    const int st = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 9), l = threadIdx.x & 511, lane = l & 63, wave = l >> 6;
    const int c0 = (blockIdx.x * 2 + st) * 8;
    double zc[ZU];
    ZBuf zb; zb.init(A, l);
    load_chunk<MODE, ZU>(A, zb, 0, 0, l, zc);
    STAMP0();
#pragma clang loop unroll(disable)
    for (int k = 0; k < NM; ++k) {
        double mu[8], acc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) { mu[c] = A.theta[(c0 + c) * NM + k]; acc[c] = 0.0; }
        sim_moment<8, MODE, ZU>(A, zb, k, k + 1 < NM ? k + 1 : k, l, mu, acc, zc);
        STAMPMID();
        const double tot = wave_reduce_t<8>(acc, lane);
        if ((lane & 7) == 0) A.out[((size_t)(c0 + (lane >> 3)) * NM + k) * 8 + wave] = tot;
    }
    STAMP1();
}

// B: one tile of 16 chains in a 1024-thread workgroup; half h = tid >> 9 takes the moments k = h, h+2, ...
template <int MODE, int ZU>
__global__ __launch_bounds__(1024, 4) void k_tile16_split(const Args A) {
    const int h = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 9), l = threadIdx.x & 511, lane = l & 63, wave = l >> 6;
    const int c0 = blockIdx.x * 16;
    double zc[ZU];
    ZBuf zb; zb.init(A, l);
    load_chunk<MODE, ZU>(A, zb, h, 0, l, zc);
    STAMP0();
#pragma clang loop unroll(disable)
    for (int k = h; k < NM; k += 2) {
        double mu[16], acc[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) { mu[c] = A.theta[(c0 + c) * NM + k]; acc[c] = 0.0; }
        sim_moment<16, MODE, ZU>(A, zb, k, k + 2 < NM ? k + 2 : k, l, mu, acc, zc);
        STAMPMID();
        const double tot = wave_reduce_t<16>(acc, lane);
        if ((lane & 3) == 0) A.out[((size_t)(c0 + (lane >> 2)) * NM + k) * 8 + wave] = tot;
    }
    STAMP1();
}

// C: 512-thread workgroup, 16 chains, every lane does both moments (2 waves per SIMD)
template <int MODE, int ZU>
__global__ __launch_bounds__(512, 2) void k_tile16_512(const Args A) {
    const int l = threadIdx.x, lane = l & 63, wave = l >> 6;
    const int c0 = blockIdx.x * 16;
    double zc[ZU];
    ZBuf zb; zb.init(A, l);
    load_chunk<MODE, ZU>(A, zb, 0, 0, l, zc);
    STAMP0();
#pragma clang loop unroll(disable)
    for (int k = 0; k < NM; ++k) {
        double mu[16], acc[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) { mu[c] = A.theta[(c0 + c) * NM + k]; acc[c] = 0.0; }
        sim_moment<16, MODE, ZU>(A, zb, k, k + 1 < NM ? k + 1 : k, l, mu, acc, zc);
        STAMPMID();
        const double tot = wave_reduce_t<16>(acc, lane);
        if ((lane & 3) == 0) A.out[((size_t)(c0 + (lane >> 2)) * NM + k) * 8 + wave] = tot;
    }
    STAMP1();
}

// D: 512-thread workgroups of 8 chains, two per CU (the round-1 unfused arrangement, TPW = 1)
template <int MODE, int ZU>
__global__ __launch_bounds__(512, 4) void k_tile8_512(const Args A) {
    const int l = threadIdx.x, lane = l & 63, wave = l >> 6;
    const int c0 = blockIdx.x * 8;
    double zc[ZU];
    ZBuf zb; zb.init(A, l);
    load_chunk<MODE, ZU>(A, zb, 0, 0, l, zc);
    STAMP0();
#pragma clang loop unroll(disable)
    for (int k = 0; k < NM; ++k) {
        double mu[8], acc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) { mu[c] = A.theta[(c0 + c) * NM + k]; acc[c] = 0.0; }
        sim_moment<8, MODE, ZU>(A, zb, k, k + 1 < NM ? k + 1 : k, l, mu, acc, zc);
        STAMPMID();
        const double tot = wave_reduce_t<8>(acc, lane);
        if ((lane & 7) == 0) A.out[((size_t)(c0 + (lane >> 3)) * NM + k) * 8 + wave] = tot;
    }
    STAMP1();
}


// E: as B, the 16 means in SGPRs (uniform per wave): frees 32 VGPRs
template <int MODE, int ZU>
__global__ __launch_bounds__(1024, 4) void k_tile16_split_smu(const Args A) {
    const int h = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 9), l = threadIdx.x & 511, lane = l & 63, wave = l >> 6;
    const int c0 = blockIdx.x * 16;
    double zc[ZU];
    ZBuf zb; zb.init(A, l);
    load_chunk<MODE, ZU>(A, zb, h, 0, l, zc);
    STAMP0();
#pragma clang loop unroll(disable)
    for (int k = h; k < NM; k += 2) {
        double mu[16], acc[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const double m = A.theta[(c0 + c) * NM + k];
            const unsigned long long um = __builtin_bit_cast(unsigned long long, m);
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)um), hi = __builtin_amdgcn_readfirstlane((unsigned)(um >> 32));
            mu[c] = __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
            acc[c] = 0.0;
        }
        sim_moment<16, MODE, ZU>(A, zb, k, k + 2 < NM ? k + 2 : k, l, mu, acc, zc);
        STAMPMID();
        const double tot = wave_reduce_t<16>(acc, lane);
        if ((lane & 3) == 0) A.out[((size_t)(c0 + (lane >> 2)) * NM + k) * 8 + wave] = tot;
    }
    STAMP1();
}

template <class K>
void run(const char* name, K kern, int grid, int block, Args A, int mode, int reps = 200) {
    A.mode = mode; A.chmask = mode == 1 ? 0 : ~0;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, A);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, A);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    const int wpb = block / 64;
    std::vector<unsigned long long> ts((size_t)grid * 16 * 6);
    CHK(hipMemcpy(ts.data(), A.ts, ts.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long r0 = ~0ull, r3 = 0;
    double skew = 0, sim_mean = 0, sim_max = 0, red = 0, clk = 0, lastadd = 0;
    int nw = 0;
    for (int g = 0; g < grid; ++g) {
        unsigned long long wg_end_adds = 0, wg_bar = 0;
        for (int w = 0; w < wpb; ++w) {
            const unsigned long long* t = &ts[((size_t)g * 16 + w) * 6];
            r0 = std::min(r0, t[0]); r3 = std::max(r3, t[3]);
            skew += (double)(t[1] - t[0]) / 100.0;
            const double d = (double)(t[2] - t[1]) / 100.0;
            sim_mean += d; sim_max = std::max(sim_max, d);
            red += (double)(t[3] - t[2]) / 100.0;
            clk += (double)(t[5] - t[4]) / ((double)(t[3] - t[1]) / 100.0);
            wg_end_adds = std::max(wg_end_adds, t[2]); wg_bar = t[1];
            ++nw;
        }
        lastadd += (double)(wg_end_adds - wg_bar) / 100.0;
    }
    const double adds = 4096.0 * NM * NS * 2;
    printf("%-18s mode %d: %6.2f us/launch (events)  span %6.2f | entry->barrier %5.2f | adds per wave mean %5.2f max %5.2f, last wave of WG %5.2f | reduce %4.2f us | %4.0f MHz | %.1f T add/s on last-wave time\n",
           name, mode, ms * 1e3 / reps, (double)(r3 - r0) / 100.0, skew / nw, sim_mean / nw, sim_max, lastadd / grid, red / nw, clk / nw, adds / (lastadd / grid) / 1e6);
    CHK(hipEventDestroy(e0)); CHK(hipEventDestroy(e1));
}

int main() {
    Args A{};
    const int rows = (NS + WG - 1) / WG;
    A.zstride = ((rows + 8) / 8) * 8 * WG + 8 * WG;
    std::vector<double> Z((size_t)NM * A.zstride, 0.0), th(4096 * NM);
    srand(1);
    for (auto& z : Z) z = (rand() / (double)RAND_MAX - 0.5) * 3.0;
    for (auto& t : th) t = (rand() / (double)RAND_MAX - 0.5) * 6.0;
    double *dZ, *dth, *dout;
    unsigned long long* dts;
    CHK(hipMalloc(&dZ, Z.size() * 8)); CHK(hipMalloc(&dth, th.size() * 8)); CHK(hipMalloc(&dout, 4096 * NM * 8 * 8));
    CHK(hipMalloc(&dts, 1024 * 16 * 6 * 8));
    CHK(hipMemcpy(dZ, Z.data(), Z.size() * 8, hipMemcpyHostToDevice));
    CHK(hipMemcpy(dth, th.data(), th.size() * 8, hipMemcpyHostToDevice));
    A.Z = dZ; A.theta = dth; A.out = dout; A.ts = dts;
    printf("4096 chains x %d moments x %d draws: %.3g FP64 adds per launch; ideal %.2f us at 39.3 T/s, %.2f us at 34.2 T/s\n", NM, NS,
           4096.0 * NM * NS * 2, 4096.0 * NM * NS * 2 / 39.3e6, 4096.0 * NM * NS * 2 / 34.2e6);
    std::vector<double> ref(4096 * NM * 8), got(4096 * NM * 8);
#define RUN1(label, kern, grid, block, mode) \
    run(label, kern, grid, block, A, mode); \
    CHK(hipMemcpy(got.data(), dout, got.size() * 8, hipMemcpyDeviceToHost)); \
    if (mode == 0 && !have_ref) { ref = got; have_ref = true; } \
    else if (mode == 0) printf("   %s vs first variant: %s\n", label, ref == got ? "bit-identical" : "DIFFERENT");
#define RUNALL(mode, M) \
    RUN1("tiles2x8/1024 zu8", (k_tiles2x8<M, 8>), 256, 1024, mode) \
    RUN1("tiles2x8/1024 zu4", (k_tiles2x8<M, 4>), 256, 1024, mode) \
    RUN1("tile16split zu8", (k_tile16_split<M, 8>), 256, 1024, mode) \
    RUN1("tile16split zu4", (k_tile16_split<M, 4>), 256, 1024, mode) \
    RUN1("tile16split zu2", (k_tile16_split<M, 2>), 256, 1024, mode) \
    RUN1("t16split smu zu8", (k_tile16_split_smu<M, 8>), 256, 1024, mode) \
    RUN1("t16split smu zu4", (k_tile16_split_smu<M, 4>), 256, 1024, mode) \
    RUN1("tile16/512 zu8", (k_tile16_512<M, 8>), 256, 512, mode) \
    RUN1("tile8/512x2 zu8", (k_tile8_512<M, 8>), 512, 512, mode)
    bool have_ref = false;
    RUNALL(0, 0) RUNALL(1, 0) RUNALL(2, 2)
    return 0;
}
