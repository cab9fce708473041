// tools/f64_issue.hip — issue rate of v_add_f64 on gfx950 by operand kind (inline asm, fixed registers):
// the calibration of tools/fp64_peak.hip adds an SGPR constant; the simulation adds two VGPR pairs.  Does the
// VGPR bank of the two 64-bit sources matter?   hipcc --offload-arch=gfx950 -O3 -o tools/f64_issue tools/f64_issue.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// 16 independent accumulators v[32:63]; sources vary.  One trip = 16 adds.
#define ACC16(SRC) \
    "v_add_f64 v[32:33], v[32:33], " SRC(0) "\n" "v_add_f64 v[34:35], v[34:35], " SRC(1) "\n" \
    "v_add_f64 v[36:37], v[36:37], " SRC(2) "\n" "v_add_f64 v[38:39], v[38:39], " SRC(3) "\n" \
    "v_add_f64 v[40:41], v[40:41], " SRC(4) "\n" "v_add_f64 v[42:43], v[42:43], " SRC(5) "\n" \
    "v_add_f64 v[44:45], v[44:45], " SRC(6) "\n" "v_add_f64 v[46:47], v[46:47], " SRC(7) "\n" \
    "v_add_f64 v[48:49], v[48:49], " SRC(8) "\n" "v_add_f64 v[50:51], v[50:51], " SRC(9) "\n" \
    "v_add_f64 v[52:53], v[52:53], " SRC(10) "\n" "v_add_f64 v[54:55], v[54:55], " SRC(11) "\n" \
    "v_add_f64 v[56:57], v[56:57], " SRC(12) "\n" "v_add_f64 v[58:59], v[58:59], " SRC(13) "\n" \
    "v_add_f64 v[60:61], v[60:61], " SRC(14) "\n" "v_add_f64 v[62:63], v[62:63], " SRC(15) "\n"

#define S_SGPR(i) "s[8:9]"
#define S_V64(i) "v[64:65]"        /* one shared VGPR source, bank 0/1 like the even accumulators */
#define S_V66(i) "v[66:67]"        /* bank 2/3 */
// distinct sources v[64+2i : 65+2i]: the banks alternate 0/1, 2/3 like the accumulators' (same bank class pairwise)
#define S_VSAME(i) "v[%c[b0] + 2*" #i " : %c[b0] + 2*" #i " + 1]"

#define CLOB "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","s8","s9"

#define INIT() asm volatile( \
    "v_mov_b32 v32, 0\nv_mov_b32 v33, 0\nv_mov_b32 v34, 0\nv_mov_b32 v35, 0\nv_mov_b32 v36, 0\nv_mov_b32 v37, 0\nv_mov_b32 v38, 0\nv_mov_b32 v39, 0\n" \
    "v_mov_b32 v40, 0\nv_mov_b32 v41, 0\nv_mov_b32 v42, 0\nv_mov_b32 v43, 0\nv_mov_b32 v44, 0\nv_mov_b32 v45, 0\nv_mov_b32 v46, 0\nv_mov_b32 v47, 0\n" \
    "v_mov_b32 v48, 0\nv_mov_b32 v49, 0\nv_mov_b32 v50, 0\nv_mov_b32 v51, 0\nv_mov_b32 v52, 0\nv_mov_b32 v53, 0\nv_mov_b32 v54, 0\nv_mov_b32 v55, 0\n" \
    "v_mov_b32 v56, 0\nv_mov_b32 v57, 0\nv_mov_b32 v58, 0\nv_mov_b32 v59, 0\nv_mov_b32 v60, 0\nv_mov_b32 v61, 0\nv_mov_b32 v62, 0\nv_mov_b32 v63, 0\n" \
    "v_mov_b32 v64, 0\nv_mov_b32 v65, 0x3ff00000\nv_mov_b32 v66, 0\nv_mov_b32 v67, 0x3ff00000\n" \
    "v_mov_b32 v68, 0\nv_mov_b32 v69, 0x3ff00000\nv_mov_b32 v70, 0\nv_mov_b32 v71, 0x3ff00000\nv_mov_b32 v72, 0\nv_mov_b32 v73, 0x3ff00000\nv_mov_b32 v74, 0\nv_mov_b32 v75, 0x3ff00000\n" \
    "v_mov_b32 v76, 0\nv_mov_b32 v77, 0x3ff00000\nv_mov_b32 v78, 0\nv_mov_b32 v79, 0x3ff00000\nv_mov_b32 v80, 0\nv_mov_b32 v81, 0x3ff00000\nv_mov_b32 v82, 0\nv_mov_b32 v83, 0x3ff00000\n" \
    "v_mov_b32 v84, 0\nv_mov_b32 v85, 0x3ff00000\nv_mov_b32 v86, 0\nv_mov_b32 v87, 0x3ff00000\nv_mov_b32 v88, 0\nv_mov_b32 v89, 0x3ff00000\nv_mov_b32 v90, 0\nv_mov_b32 v91, 0x3ff00000\n" \
    "v_mov_b32 v92, 0\nv_mov_b32 v93, 0x3ff00000\nv_mov_b32 v94, 0\nv_mov_b32 v95, 0x3ff00000\nv_mov_b32 v96, 0\nv_mov_b32 v97, 0x3ff00000\n" \
    "s_mov_b32 s8, 0\ns_mov_b32 s9, 0x3ff00000\n" ::: CLOB)

#define FINI(out) do { double r_; asm volatile("v_add_f64 %0, v[32:33], v[62:63]" : "=v"(r_) :: CLOB); (out)[blockIdx.x * blockDim.x + threadIdx.x] = r_; } while (0)

// kind 0: acc += SGPR; 1: acc += one VGPR pair of the same bank class as half the accumulators; 2: acc_i += v[64+2i] (same bank class as acc_i);
// 3: acc_i += v[66+2i] (the other bank class); 4: x = z + mu (v[64+2i] = v[98] ... no: dst v[32+2i] = v[64+2i] + v[66..]) independent, 2 VGPR sources
template <int KIND>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
    INIT();
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == 0) asm volatile(ACC16(S_SGPR) ::: CLOB);
        if constexpr (KIND == 1) asm volatile(ACC16(S_V64) ::: CLOB);
        if constexpr (KIND == 2) asm volatile(
            "v_add_f64 v[32:33], v[32:33], v[64:65]\nv_add_f64 v[34:35], v[34:35], v[66:67]\nv_add_f64 v[36:37], v[36:37], v[68:69]\nv_add_f64 v[38:39], v[38:39], v[70:71]\n"
            "v_add_f64 v[40:41], v[40:41], v[72:73]\nv_add_f64 v[42:43], v[42:43], v[74:75]\nv_add_f64 v[44:45], v[44:45], v[76:77]\nv_add_f64 v[46:47], v[46:47], v[78:79]\n"
            "v_add_f64 v[48:49], v[48:49], v[80:81]\nv_add_f64 v[50:51], v[50:51], v[82:83]\nv_add_f64 v[52:53], v[52:53], v[84:85]\nv_add_f64 v[54:55], v[54:55], v[86:87]\n"
            "v_add_f64 v[56:57], v[56:57], v[88:89]\nv_add_f64 v[58:59], v[58:59], v[90:91]\nv_add_f64 v[60:61], v[60:61], v[92:93]\nv_add_f64 v[62:63], v[62:63], v[94:95]\n" ::: CLOB);
        if constexpr (KIND == 3) asm volatile(
            "v_add_f64 v[32:33], v[32:33], v[66:67]\nv_add_f64 v[34:35], v[34:35], v[68:69]\nv_add_f64 v[36:37], v[36:37], v[70:71]\nv_add_f64 v[38:39], v[38:39], v[72:73]\n"
            "v_add_f64 v[40:41], v[40:41], v[74:75]\nv_add_f64 v[42:43], v[42:43], v[76:77]\nv_add_f64 v[44:45], v[44:45], v[78:79]\nv_add_f64 v[46:47], v[46:47], v[80:81]\n"
            "v_add_f64 v[48:49], v[48:49], v[82:83]\nv_add_f64 v[50:51], v[50:51], v[84:85]\nv_add_f64 v[52:53], v[52:53], v[86:87]\nv_add_f64 v[54:55], v[54:55], v[88:89]\n"
            "v_add_f64 v[56:57], v[56:57], v[90:91]\nv_add_f64 v[58:59], v[58:59], v[92:93]\nv_add_f64 v[60:61], v[60:61], v[94:95]\nv_add_f64 v[62:63], v[62:63], v[96:97]\n" ::: CLOB);
        if constexpr (KIND == 4) asm volatile(   // the simulation's pattern: x = z + mu_c (two VGPR pairs, new destination), acc_c = acc_c + x
            "v_add_f64 v[80:81], v[64:65], v[66:67]\nv_add_f64 v[32:33], v[32:33], v[80:81]\nv_add_f64 v[82:83], v[64:65], v[68:69]\nv_add_f64 v[34:35], v[34:35], v[82:83]\n"
            "v_add_f64 v[84:85], v[64:65], v[70:71]\nv_add_f64 v[36:37], v[36:37], v[84:85]\nv_add_f64 v[86:87], v[64:65], v[72:73]\nv_add_f64 v[38:39], v[38:39], v[86:87]\n"
            "v_add_f64 v[88:89], v[64:65], v[74:75]\nv_add_f64 v[40:41], v[40:41], v[88:89]\nv_add_f64 v[90:91], v[64:65], v[76:77]\nv_add_f64 v[42:43], v[42:43], v[90:91]\n"
            "v_add_f64 v[92:93], v[64:65], v[78:79]\nv_add_f64 v[44:45], v[44:45], v[92:93]\nv_add_f64 v[94:95], v[64:65], v[66:67]\nv_add_f64 v[46:47], v[46:47], v[94:95]\n" ::: CLOB);
        if constexpr (KIND == 5) asm volatile(   // the same with mu_c in SGPRs: x = z + s_mu, acc = acc + x
            "v_add_f64 v[80:81], v[64:65], s[8:9]\nv_add_f64 v[32:33], v[32:33], v[80:81]\nv_add_f64 v[82:83], v[64:65], s[8:9]\nv_add_f64 v[34:35], v[34:35], v[82:83]\n"
            "v_add_f64 v[84:85], v[64:65], s[8:9]\nv_add_f64 v[36:37], v[36:37], v[84:85]\nv_add_f64 v[86:87], v[64:65], s[8:9]\nv_add_f64 v[38:39], v[38:39], v[86:87]\n"
            "v_add_f64 v[88:89], v[64:65], s[8:9]\nv_add_f64 v[40:41], v[40:41], v[88:89]\nv_add_f64 v[90:91], v[64:65], s[8:9]\nv_add_f64 v[42:43], v[42:43], v[90:91]\n"
            "v_add_f64 v[92:93], v[64:65], s[8:9]\nv_add_f64 v[44:45], v[44:45], v[92:93]\nv_add_f64 v[94:95], v[64:65], s[8:9]\nv_add_f64 v[46:47], v[46:47], v[94:95]\n" ::: CLOB);
        if constexpr (KIND == 6) asm volatile(   // f32 adds with two VGPR sources, for comparison (16 per trip)
            "v_add_f32 v32, v32, v64\nv_add_f32 v33, v33, v65\nv_add_f32 v34, v34, v66\nv_add_f32 v35, v35, v67\nv_add_f32 v36, v36, v68\nv_add_f32 v37, v37, v69\nv_add_f32 v38, v38, v70\nv_add_f32 v39, v39, v71\n"
            "v_add_f32 v40, v40, v72\nv_add_f32 v41, v41, v73\nv_add_f32 v42, v42, v74\nv_add_f32 v43, v43, v75\nv_add_f32 v44, v44, v76\nv_add_f32 v45, v45, v77\nv_add_f32 v46, v46, v78\nv_add_f32 v47, v47, v79\n" ::: CLOB);
        if constexpr (KIND == 7) asm volatile(   // v_fma_f64 acc = acc*1 + z ... three VGPR pairs
            "v_fma_f64 v[32:33], v[64:65], v[66:67], v[32:33]\nv_fma_f64 v[34:35], v[64:65], v[66:67], v[34:35]\nv_fma_f64 v[36:37], v[64:65], v[66:67], v[36:37]\nv_fma_f64 v[38:39], v[64:65], v[66:67], v[38:39]\n"
            "v_fma_f64 v[40:41], v[64:65], v[66:67], v[40:41]\nv_fma_f64 v[42:43], v[64:65], v[66:67], v[42:43]\nv_fma_f64 v[44:45], v[64:65], v[66:67], v[44:45]\nv_fma_f64 v[46:47], v[64:65], v[66:67], v[46:47]\n"
            "v_fma_f64 v[48:49], v[64:65], v[66:67], v[48:49]\nv_fma_f64 v[50:51], v[64:65], v[66:67], v[50:51]\nv_fma_f64 v[52:53], v[64:65], v[66:67], v[52:53]\nv_fma_f64 v[54:55], v[64:65], v[66:67], v[54:55]\n"
            "v_fma_f64 v[56:57], v[64:65], v[66:67], v[56:57]\nv_fma_f64 v[58:59], v[64:65], v[66:67], v[58:59]\nv_fma_f64 v[60:61], v[64:65], v[66:67], v[60:61]\nv_fma_f64 v[62:63], v[64:65], v[66:67], v[62:63]\n" ::: CLOB);
    }
    FINI(out);
}

template <int KIND>
void run(const char* what, int blocks_per_cu, int iters) {
    double* d;
    const int grid = 256 * blocks_per_cu;
    CHK(hipMalloc(&d, (size_t)grid * 256 * 8));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    k<KIND><<<grid, 256>>>(d, iters);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    k<KIND><<<grid, 256>>>(d, iters);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    const double ops = (double)grid * 256 * 16 * iters;
    // cycles per wave-instruction at 2.4 GHz if the SIMD were fully busy: waves/SIMD * 16 * iters instr per SIMD
    printf("%-52s waves/SIMD=%d : %.3f ms  %6.2f T op/s  (%.2f cycles per wave-instruction per SIMD at 2.4 GHz)\n", what, blocks_per_cu, ms, ops / ms / 1e9,
           ms * 1e-3 * 2.4e9 / ((double)blocks_per_cu * 16 * iters));
    CHK(hipFree(d));
}


// ---- short kernels (640 adds per lane like one simulation), 1024-thread workgroups, per-wave in-kernel stamps ----
#define SIM16 "v_add_f64 v[80:81], v[64:65], v[66:67]\nv_add_f64 v[32:33], v[32:33], v[80:81]\nv_add_f64 v[82:83], v[64:65], v[68:69]\nv_add_f64 v[34:35], v[34:35], v[82:83]\n" \
            "v_add_f64 v[84:85], v[64:65], v[70:71]\nv_add_f64 v[36:37], v[36:37], v[84:85]\nv_add_f64 v[86:87], v[64:65], v[72:73]\nv_add_f64 v[38:39], v[38:39], v[86:87]\n" \
            "v_add_f64 v[88:89], v[64:65], v[74:75]\nv_add_f64 v[40:41], v[40:41], v[88:89]\nv_add_f64 v[90:91], v[64:65], v[76:77]\nv_add_f64 v[42:43], v[42:43], v[90:91]\n" \
            "v_add_f64 v[92:93], v[64:65], v[78:79]\nv_add_f64 v[44:45], v[44:45], v[92:93]\nv_add_f64 v[94:95], v[64:65], v[66:67]\nv_add_f64 v[46:47], v[46:47], v[94:95]\n"
#define SIM160 SIM16 SIM16 SIM16 SIM16 SIM16 SIM16 SIM16 SIM16 SIM16 SIM16
template <int STRAIGHT, int BLOCK>
__global__ __launch_bounds__(BLOCK) void kshort(double* out, unsigned long long* ts, int iters) {
    const unsigned long long t0 = wall_clock64();
    __syncthreads();
    const unsigned long long t1 = wall_clock64(), c1 = clock64();
    INIT();
    if constexpr (STRAIGHT) {
        for (int it = 0; it < iters / 40; ++it) { asm volatile(SIM160 SIM160 SIM160 SIM160 ::: CLOB); }
    } else {
        for (int it = 0; it < iters; ++it) asm volatile(SIM16 ::: CLOB);
    }
    const unsigned long long t2 = wall_clock64(), c2 = clock64();
    FINI(out);
    if ((threadIdx.x & 63) == 0) { unsigned long long* t = ts + ((size_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6)) * 5; t[0] = t0; t[1] = t1; t[2] = t2; t[3] = c1; t[4] = c2; }
}
template <int STRAIGHT, int BLOCK>
void run_short(const char* what, int iters) {
    const int grid = 256 * (1024 / BLOCK), wpb = BLOCK / 64;
    double* d; unsigned long long* ts;
    CHK(hipMalloc(&d, (size_t)grid * BLOCK * 8)); CHK(hipMalloc(&ts, (size_t)grid * wpb * 5 * 8));
    for (int i = 0; i < 50; ++i) kshort<STRAIGHT, BLOCK><<<grid, BLOCK>>>(d, ts, iters);
    CHK(hipDeviceSynchronize());
    unsigned long long* h = (unsigned long long*)malloc((size_t)grid * wpb * 5 * 8);
    CHK(hipMemcpy(h, ts, (size_t)grid * wpb * 5 * 8, hipMemcpyDeviceToHost));
    double mean = 0, mx = 0, clk = 0, cu_last = 0;
    unsigned long long r0 = ~0ull, r1 = 0;
    for (int g = 0; g < grid; ++g) {
        double last = 0;
        for (int w = 0; w < wpb; ++w) {
            const unsigned long long* t = h + ((size_t)g * wpb + w) * 5;
            const double dd = (double)(t[2] - t[1]) / 100.0;
            mean += dd; mx = dd > mx ? dd : mx; last = dd > last ? dd : last;
            clk += (double)(t[4] - t[3]) / dd;
            r0 = t[0] < r0 ? t[0] : r0; r1 = t[2] > r1 ? t[2] : r1;
        }
        cu_last += last;
    }
    printf("%-44s %d adds/lane, %4d-thread WGs: per wave mean %5.2f max %5.2f us, last wave of WG %5.2f us, span %5.2f us, %4.0f MHz; ideal %.2f us at 4.65 cyc\n", what, iters * 16, BLOCK,
           mean / (grid * wpb), mx, cu_last / grid, (double)(r1 - r0) / 100.0, clk / (grid * wpb), 4.0 * iters * 16 * 4.65 / 2400.0);
    free(h); CHK(hipFree(d)); CHK(hipFree(ts));
}

int main() {
    run_short<0, 1024>("rolled loop of 16 adds", 40);
    run_short<1, 1024>("straight-line 640 adds", 40);
    run_short<0, 256>("rolled loop of 16 adds", 40);
    run_short<1, 256>("straight-line 640 adds", 40);
    run_short<0, 1024>("rolled loop of 16 adds", 400);
    run_short<1, 1024>("straight-line 640 adds x10", 400);

    const int it = 20000;
    for (int bpc : {1, 2, 4}) {
        run<0>("acc += SGPR", bpc, it);
        run<1>("acc += one VGPR pair (v[64:65])", bpc, it);
        run<2>("acc_i += v[64+2i] (same bank class as acc_i)", bpc, it);
        run<3>("acc_i += v[66+2i] (other bank class)", bpc, it);
        run<4>("x = z + mu_c (VGPR); acc_c += x   [the simulation]", bpc, it);
        run<5>("x = z + mu_c (SGPR); acc_c += x", bpc, it);
        run<6>("f32: acc += VGPR", bpc, it);
        run<7>("fma f64, three VGPR pairs", bpc, it);
    }
    return 0;
}
