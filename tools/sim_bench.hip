// Standalone timing of the simulation phase of k_chain_iter (tools only, not part of the library).
#include "../smm.jl_amd/csrc/smmhip.hip"

namespace {

template <int CT, int VAR>
__device__ inline void sim_variant(const KParams& P, const double* s_theta, double* s_part, int tid, double (&zc)[ZU]) {
    const int lane = tid & 63, wave = tid >> 6;
    const int ns = P.ns, nm = P.nm;
    const int nfull = ns / (ZU * WG);
    for (int k = 0; k < nm; ++k) {
        const double* __restrict__ Zk = P.Z + (size_t)k * ns;
        double mu[CT], acc[CT], zt[ZU];
        if (VAR != 2) sim_load_chunk(Zk, ns, nfull, tid, zt, 0);
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            mu[c] = s_theta[c * P.np + k];
            if (VAR == 3) {
                union { double d; int i[2]; } v; v.d = mu[c];
                v.i[0] = __builtin_amdgcn_readfirstlane(v.i[0]); v.i[1] = __builtin_amdgcn_readfirstlane(v.i[1]);
                mu[c] = v.d;
            }
            acc[c] = 0.0;
        }
        for (int ch = 0; ch < nfull; ++ch) {
            double zn[ZU];
            const bool last = (ch + 1 == nfull);
            const double* __restrict__ Zn = (last && k + 1 < nm) ? Zk + ns : Zk;
            sim_load_chunk(Zn, ns, last ? 0 : ch + 1, tid, zn, 0);
#pragma unroll
            for (int u = 0; u < ZU; ++u) {
#pragma unroll
                for (int c = 0; c < CT; ++c) { const double x = zc[u] + mu[c]; acc[c] = acc[c] + x; }
            }
#pragma unroll
            for (int u = 0; u < ZU; ++u) zc[u] = zn[u];
        }
        if (VAR != 2) {
            const int s0 = nfull * ZU * WG + tid;
#pragma unroll
            for (int u = 0; u < ZU; ++u) {
                if (s0 + u * WG < ns) {
#pragma unroll
                    for (int c = 0; c < CT; ++c) { const double x = zt[u] + mu[c]; acc[c] = acc[c] + x; }
                }
            }
        }
        if (VAR == 1) {
            double t = 0;
#pragma unroll
            for (int c = 0; c < CT; ++c) t += acc[c];
            if (t == 12345.678) s_part[tid] = t;
        } else {
            const double tot = wave_reduce_transposed<CT>(acc, lane);
            if (acc_writer<CT>(lane)) s_part[(wave * CT + acc_index<CT>(lane)) * nm + k] = tot;
        }
    }
}

template <int CT, int VAR>
__global__ __launch_bounds__(WG, 4) void k_sim(const KParams P, double* out, unsigned long long* ts) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* s_theta = smem;
    double* s_part = smem + CT * P.np;
    const int tid = threadIdx.x;
    if (tid < CT * P.np) s_theta[tid] = 0.001 * tid + blockIdx.x;
    double za[ZU];
    sim_load_chunk(P.Z, P.ns, 0, tid, za, 0);
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    if (VAR == 0) simulate_tile<CT>(P, s_theta, s_part, tid, za);
    else sim_variant<CT, VAR>(P, s_theta, s_part, tid, za);
    __syncthreads();
    const unsigned long long t1 = wall_clock64();
    if (tid < CT * P.nm) out[blockIdx.x * CT * P.nm + tid] = s_part[tid];
    if (tid == 0) { ts[2 * blockIdx.x] = t0; ts[2 * blockIdx.x + 1] = t1; }
}


// V4: balanced row chunks (ZB-row register buffers, double buffered), G moments reduced together
constexpr int ZB = 12;
template <int CT, int G>
__device__ inline void sim_v4(const KParams& P, const double* s_theta, double* s_part, int tid, double (&za)[ZB]) {
    const int lane = tid & 63, wave = tid >> 6;
    const int ns = P.ns, nm = P.nm;
    const int rows = (ns + WG - 1) / WG;
    const int nch = (rows + ZB - 1) / ZB;
    const int rpc = (rows + nch - 1) / nch;  // rows per chunk (balanced)
    const int nitems = nm * nch;
    double zb[ZB];
    int item = 0;
    for (int k0 = 0; k0 < nm; k0 += G) {
        double acc[G * CT];
#pragma unroll
        for (int i = 0; i < G * CT; ++i) acc[i] = 0.0;
#pragma unroll
        for (int kk = 0; kk < G; ++kk) {
            const int k = k0 + kk;
            if (k < nm) {
                double mu[CT];
#pragma unroll
                for (int c = 0; c < CT; ++c) mu[c] = s_theta[c * P.np + k];
                for (int ch = 0; ch < nch; ch += 2) {
                    // ---- item in za; prefetch next into zb
                    {
                        const int nit = item + 1;
                        const int nk = min(nit / nch, nm - 1), nc = nit - (nit / nch) * nch;
                        const double* __restrict__ Zn = P.Z + (size_t)nk * ns;
                        const int s0 = nc * rpc * WG + tid;
#pragma unroll
                        for (int u = 0; u < ZB; ++u) if (u < rpc) zb[u] = Zn[min(s0 + u * WG, ns - 1)];
                        const int c0 = ch * rpc * WG + tid;
#pragma unroll
                        for (int u = 0; u < ZB; ++u) {
                            if (u < rpc && c0 + u * WG < ns) {
#pragma unroll
                                for (int c = 0; c < CT; ++c) { const double x = za[u] + mu[c]; acc[kk * CT + c] = acc[kk * CT + c] + x; }
                            }
                        }
                        ++item;
                    }
                    if (ch + 1 < nch) {
                        const int nit = item + 1;
                        const int nk = min(nit / nch, nm - 1), nc = nit - (nit / nch) * nch;
                        const double* __restrict__ Zn = P.Z + (size_t)nk * ns;
                        const int s0 = nc * rpc * WG + tid;
#pragma unroll
                        for (int u = 0; u < ZB; ++u) if (u < rpc) za[u] = Zn[min(s0 + u * WG, ns - 1)];
                        const int c0 = (ch + 1) * rpc * WG + tid;
#pragma unroll
                        for (int u = 0; u < ZB; ++u) {
                            if (u < rpc && c0 + u * WG < ns) {
#pragma unroll
                                for (int c = 0; c < CT; ++c) { const double x = zb[u] + mu[c]; acc[kk * CT + c] = acc[kk * CT + c] + x; }
                            }
                        }
                        ++item;
                    } else {
                        // odd chunk count: the prefetched buffer is zb, move it to za for the next moment
#pragma unroll
                        for (int u = 0; u < ZB; ++u) za[u] = zb[u];
                    }
                }
            }
        }
        const double tot = wave_reduce_transposed<G * CT>(acc, lane);
        if (acc_writer<G * CT>(lane)) {
            const int a = acc_index<G * CT>(lane);
            const int kk = a / CT, c = a - kk * CT;
            if (k0 + kk < nm) s_part[(wave * CT + c) * nm + k0 + kk] = tot;
        }
    }
    (void)nitems;
}

template <int CT, int G>
__global__ __launch_bounds__(WG, 4) void k_sim4(const KParams P, double* out, unsigned long long* ts) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* s_theta = smem;
    double* s_part = smem + CT * P.np;
    const int tid = threadIdx.x;
    if (tid < CT * P.np) s_theta[tid] = 0.001 * tid + blockIdx.x;
    double za[ZB];
    {
        const int rows = (P.ns + WG - 1) / WG, nch = (rows + ZB - 1) / ZB, rpc = (rows + nch - 1) / nch;
#pragma unroll
        for (int u = 0; u < ZB; ++u) if (u < rpc) za[u] = P.Z[min(tid + u * WG, P.ns - 1)];
    }
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    sim_v4<CT, G>(P, s_theta, s_part, tid, za);
    __syncthreads();
    const unsigned long long t1 = wall_clock64();
    if (tid < CT * P.nm) out[blockIdx.x * CT * P.nm + tid] = s_part[tid];
    if (tid == 0) { ts[2 * blockIdx.x] = t0; ts[2 * blockIdx.x + 1] = t1; }
}

template <int CT, int VAR>
void run(const char* name, int ntiles) {
    KParams P{};
    P.np = 2; P.nm = 2; P.ns = 10000;
    std::vector<double> Z((size_t)P.nm * P.ns);
    for (size_t i = 0; i < Z.size(); ++i) Z[i] = (double)(i % 97) * 0.01 - 0.5;
    double* dZ; (void)hipMalloc(&dZ, Z.size() * 8); (void)hipMemcpy(dZ, Z.data(), Z.size() * 8, hipMemcpyHostToDevice);
    P.Z = dZ;
    double* out; (void)hipMalloc(&out, (size_t)ntiles * CT * 2 * 8);
    unsigned long long* ts; (void)hipMalloc(&ts, (size_t)ntiles * 16);
    const size_t smem = (size_t)(CT * 2 + 8 * CT * 2 + 512) * 8;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
#define LAUNCH() do { if (VAR == 4) hipLaunchKernelGGL((k_sim4<CT, 2>), dim3(ntiles), dim3(WG), smem, 0, P, out, ts); else if (VAR == 5) hipLaunchKernelGGL((k_sim4<CT, 1>), dim3(ntiles), dim3(WG), smem, 0, P, out, ts); else hipLaunchKernelGGL((k_sim<CT, VAR>), dim3(ntiles), dim3(WG), smem, 0, P, out, ts); } while (0)
    for (int i = 0; i < 3; ++i) LAUNCH();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    const int reps = 50;
    for (int i = 0; i < reps; ++i) LAUNCH();
    std::vector<double> ho((size_t)ntiles * CT * 2);
    (void)hipMemcpy(ho.data(), out, ho.size() * 8, hipMemcpyDeviceToHost);
    double chk = 0; for (double v : ho) chk += v;
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)ntiles * 2);
    (void)hipMemcpy(h.data(), ts, h.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0, mx = 0; unsigned long long tmin = ~0ull, tmax = 0;
    for (int b = 0; b < ntiles; ++b) { double d = (h[2*b+1] - h[2*b]) / 100.0; mean += d; if (d > mx) mx = d; if (h[2*b] < tmin) tmin = h[2*b]; if (h[2*b+1] > tmax) tmax = h[2*b+1]; }
    mean /= ntiles;
    const double adds = (double)ntiles * CT * 2.0 * 10000 * 2;
    printf("%-34s chk=%.6f tiles=%d  launch-to-launch %.2f us | in-kernel sim mean %.2f max %.2f span %.2f us -> %.1f T add/s (span)\n", name, chk, ntiles, ms * 1e3 / reps,
           mean, mx, (tmax - tmin) / 100.0, adds / ((tmax - tmin) / 100.0 * 1e-6) / 1e12);
    (void)hipFree(dZ); (void)hipFree(out); (void)hipFree(ts);
}
}  // namespace

int main() {
    run<8, 0>("V0 simulate_tile CT=8", 512);
    run<8, 1>("V1 no wave reduction", 512);
    run<8, 2>("V2 no ragged tail", 512);
    run<8, 3>("V3 mu via readfirstlane", 512);
    run<8, 4>("V4 balanced chunks, G=2", 512);
    run<8, 5>("V5 balanced chunks, G=1", 512);
    run<8, 4>("V4 256 tiles", 256);
    run<8, 0>("V0 CT=8, 256 tiles (1 WG/CU)", 256);
    run<16, 0>("V0 CT=16, 256 tiles", 256);
    run<4, 0>("V0 CT=4, 1024 tiles", 1024);
    return 0;
}
