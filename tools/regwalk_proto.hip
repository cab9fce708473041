// tools/regwalk_proto.hip — VERDICT r5 "Next #3c": the exchange walk of a tile's cone IN REGISTERS instead of LDS, built instead of judged on paper
// (EXPERIMENTS.md R5.3).  A cone of the headline configuration is ~100 pairs over ~116 chains (16 own + ~100 gathered) in ~8.4 dependency sub-levels.
//   lds : what k_chain_persist_loc does (smm_walk_lean.hpp, 8-byte slots {key32, src | stamp}): a lone wave, one LDS round trip per sub-level, a lane per pair
//   reg : the slots in TWO vector registers (lane = local chain, <= 64 chains: already the optimistic case), the pairs one after the other in list order —
//         v_readlane x 4, a scalar compare, v_writelane x 4 under a scalar branch; the pair words through scalar loads, several ahead
// Same pair lists (random pairs, greedy levels), same result required.   hipcc --offload-arch=gfx950 -O3 -o tools/regwalk_proto tools/regwalk_proto.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
#include <algorithm>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NS = 64;      // slots (chains of the cone)
// v_writelane_b32 with the value in a scalar register and the lane in m0 (one scalar operand besides m0 on gfx9)
__device__ inline uint32_t writelane(const uint32_t val, const int lane, uint32_t v) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(v) : "s"(__builtin_amdgcn_readfirstlane((int)val)), "s"(__builtin_amdgcn_readfirstlane(lane)) : "m0");
    return v;
}
struct Args { const uint32_t* pairs; const uint32_t* lvl_off; int npairs, nlev, reps; const uint32_t* keys0; uint32_t* out; unsigned long long* ticks; };

// serial walk in registers: wave 0 of each workgroup; every lane holds slot `lane`
__global__ __launch_bounds__(64) void k_reg(const Args A) {
    const int lane = threadIdx.x;
    uint32_t key = A.keys0[lane], meta = (uint32_t)lane;
    const unsigned long long t0 = wall_clock64();
    for (int r = 0; r < A.reps; ++r) {
        key = A.keys0[lane] + (uint32_t)r; meta = (uint32_t)lane;
        for (int p = 0; p < A.npairs; ++p) {
            const uint32_t pw = A.pairs[p];                        // (uniform: a scalar load)
            const int i = __builtin_amdgcn_readfirstlane((int)(pw & 0xffffu)), j = __builtin_amdgcn_readfirstlane((int)(pw >> 16));
            const uint32_t ki = __builtin_amdgcn_readlane(key, i), kj = __builtin_amdgcn_readlane(key, j);
            if (ki > kj) {                                         // dist_fun = -, min_improve = 0 on order keys
                const uint32_t mi = __builtin_amdgcn_readlane(meta, i), mj = __builtin_amdgcn_readlane(meta, j);
                const uint32_t stamp = (uint32_t)(p + 1) << 16;
                key = writelane(kj, i, key); key = writelane(ki, j, key);
                meta = writelane((mj & 0xffffu) | stamp, i, meta); meta = writelane((mi & 0xffffu) | stamp, j, meta);
            }
        }
    }
    const unsigned long long t1 = wall_clock64();
    A.out[(size_t)blockIdx.x * 128 + lane] = key; A.out[(size_t)blockIdx.x * 128 + 64 + lane] = meta;
    if (lane == 0) A.ticks[blockIdx.x] = t1 - t0;
}

// level-parallel walk on LDS slots: a lone wave, one pair per lane and sub-level (the shape of lean_walk_levels' narrow tail)
__global__ __launch_bounds__(64) void k_lds(const Args A) {
    __shared__ uint2 slots[NS];
    const int lane = threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    for (int r = 0; r < A.reps; ++r) {
        slots[lane] = make_uint2(A.keys0[lane] + (uint32_t)r, (uint32_t)lane);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        for (int l = 0; l < A.nlev; ++l) {
            const uint32_t o0 = A.lvl_off[l], o1 = A.lvl_off[l + 1];
            if ((uint32_t)lane < o1 - o0) {
                const uint32_t pw = A.pairs[o0 + lane];
                const int i = (int)(pw & 0xffffu), j = (int)(pw >> 16);
                const uint2 si = slots[i], sj = slots[j];
                if (si.x > sj.x) {
                    const uint32_t stamp = (o0 + lane + 1) << 16;
                    slots[i] = make_uint2(sj.x, (sj.y & 0xffffu) | stamp); slots[j] = make_uint2(si.x, (si.y & 0xffffu) | stamp);
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);    // lgkmcnt(0): the wave's own LDS operations complete in order
        }
    }
    const unsigned long long t1 = wall_clock64();
    A.out[(size_t)blockIdx.x * 128 + lane] = slots[lane].x; A.out[(size_t)blockIdx.x * 128 + 64 + lane] = slots[lane].y;
    if (lane == 0) A.ticks[blockIdx.x] = t1 - t0;
}

int main() {
    const int npairs = 100, reps = 2000, wgs = 256;
    srand(7);
    // random pairs i < j over NS slots, then greedy dependency levels (a pair goes one level behind the last pair that touched either chain), list order kept inside a level
    std::vector<std::pair<int, int>> pr;
    while ((int)pr.size() < npairs) { int i = rand() % NS, j = rand() % NS; if (i == j) continue; if (i > j) std::swap(i, j); pr.push_back({i, j}); }
    std::vector<int> last(NS, -1), lvl(npairs);
    int nlev = 0;
    for (int p = 0; p < npairs; ++p) { lvl[p] = std::max(last[pr[p].first], last[pr[p].second]) + 1; last[pr[p].first] = last[pr[p].second] = lvl[p]; nlev = std::max(nlev, lvl[p] + 1); }
    std::vector<uint32_t> byl, off(nlev + 1, 0), serial;
    for (int l = 0; l < nlev; ++l) { off[l] = (uint32_t)byl.size(); for (int p = 0; p < npairs; ++p) if (lvl[p] == l) byl.push_back((uint32_t)pr[p].first | ((uint32_t)pr[p].second << 16)); }
    off[nlev] = (uint32_t)byl.size();
    // (the serial walk takes the pairs in LEVEL order too: the same sequence of swaps, so that the two results can be compared word for word)
    serial = byl;
    std::vector<uint32_t> keys(NS);
    for (auto& k : keys) k = (uint32_t)rand();
    uint32_t *d_ser, *d_byl, *d_off, *d_keys, *d_out; unsigned long long* d_t;
    CHK(hipMalloc(&d_ser, npairs * 4)); CHK(hipMalloc(&d_byl, npairs * 4)); CHK(hipMalloc(&d_off, (nlev + 1) * 4)); CHK(hipMalloc(&d_keys, NS * 4));
    CHK(hipMalloc(&d_out, (size_t)wgs * 128 * 4)); CHK(hipMalloc(&d_t, wgs * 8));
    CHK(hipMemcpy(d_ser, serial.data(), npairs * 4, hipMemcpyHostToDevice)); CHK(hipMemcpy(d_byl, byl.data(), npairs * 4, hipMemcpyHostToDevice));
    CHK(hipMemcpy(d_off, off.data(), (nlev + 1) * 4, hipMemcpyHostToDevice)); CHK(hipMemcpy(d_keys, keys.data(), NS * 4, hipMemcpyHostToDevice));
    std::vector<uint32_t> o1((size_t)wgs * 128), o2((size_t)wgs * 128);
    std::vector<unsigned long long> t(wgs);
    for (int which = 0; which < 2; ++which) {
        Args A{which ? d_byl : d_ser, d_off, npairs, nlev, reps, d_keys, d_out, d_t};
        for (int rep = 0; rep < 2; ++rep) {
            if (which) hipLaunchKernelGGL(k_lds, dim3(wgs), dim3(64), 0, 0, A); else hipLaunchKernelGGL(k_reg, dim3(wgs), dim3(64), 0, 0, A);
            CHK(hipDeviceSynchronize());
        }
        CHK(hipMemcpy((which ? o2 : o1).data(), d_out, o1.size() * 4, hipMemcpyDeviceToHost)); CHK(hipMemcpy(t.data(), d_t, wgs * 8, hipMemcpyDeviceToHost));
        double m = 0; for (auto x : t) m += (double)x; m /= wgs;
        printf("%s walk: %d pairs over %d slots in %d sub-levels, a lone wave per CU (256 workgroups): %.3f us per walk (%.1f ns per pair, %.0f ns per sub-level)\n",
               which ? "LDS level-parallel" : "register serial  ", npairs, NS, nlev, m * 10.0 / reps / 1000.0, m * 10.0 / reps / npairs, m * 10.0 / reps / nlev);
    }
    printf("results %s\n", o1 == o2 ? "identical" : "DIFFER");
    return 0;
}
