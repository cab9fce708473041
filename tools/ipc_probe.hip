// tools/ipc_probe.hip — the transport of the p2p sharded form, on its own: G processes, one window each, every process maps every
// other's window through hipIpcGetMemHandle / hipIpcOpenMemHandle and talks to it with plain stores, a system-scope release and
// one atomic per tile (the same protocol as k_chain_iter_norm_p2p: data -> fence -> arrival counter; readers spin on their own
// counters).  All G processes may sit on ONE device (a 1-GPU lease: RCCL refuses duplicate devices, raw HIP IPC does not) or on
// device rank % ndev.
//   hipcc --offload-arch=gfx950 -O2 -o ipc_probe ipc_probe.hip && ./ipc_probe [G=2] [mode=1] [iters=2000] [wgs=64]
//   mode 0: hipMalloc, 1: hipExtMallocWithFlags(hipDeviceMallocFinegrained), 2: hipDeviceMallocUncached
// Prints: one-way flag latency (ping-pong inside one kernel), and the per-iteration time + data errors of the launch-per-iteration
// protocol (every rank: wait for everybody's iteration t-1, check it, publish iteration t to everybody).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <sys/wait.h>
#include <sys/stat.h>
#include <time.h>
#include <string>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "[rank %d] %s: %s (line %d)\n", g_rank, #x, hipGetErrorString(e_), __LINE__); _exit(3); } } while (0)
static int g_rank = -1;

constexpr int MAXG = 8;
constexpr int ROWS = 4096;           // "chains" per rank
constexpr int RW = 8;                // doubles per row
struct Window {                      // layout of one rank's window
    unsigned long long arrived[MAXG * 16];    // one counter per source rank, 128 bytes apart
    unsigned long long ping[16];
    double data[2][MAXG][ROWS][RW];           // [parity][source rank][row][RW]
};
struct Peers { Window* w[MAXG]; };

__device__ inline unsigned long long now100() { return wall_clock64(); }   // 100 MHz

// ping-pong: rank 0 and rank 1, one lane each
__global__ void k_pingpong(Peers P, int rank, int n, unsigned long long* out) {
    Window* mine = P.w[rank];
    Window* other = P.w[rank ^ 1];
    const unsigned long long t0 = now100();
    for (int i = 1; i <= n; ++i) {
        if (rank == 0) __hip_atomic_store(&other->ping[0], (unsigned long long)i, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        unsigned long long spins = 0;
        while (__hip_atomic_load(&mine->ping[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < (unsigned long long)i) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 1023) == 0 && now100() - t0 > 300000000ull) { out[1] = 1; return; }   // 3 s: give up
        }
        if (rank == 1) __hip_atomic_store(&other->ping[0], (unsigned long long)i, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    out[0] = now100() - t0;
}

// one iteration of the protocol.  grid = wgs, block = 256.  Rows of this rank are split over the workgroups.
__global__ void k_iter(Peers P, int rank, int G, int t, int wgs, unsigned long long* errs) {
    Window* mine = P.w[rank];
    const int tid = threadIdx.x;
    __shared__ int s_fail;
    if (tid == 0) s_fail = 0;
    __syncthreads();
    // wait for everybody's iteration t-1 (arrival counters are monotone: wgs per iteration and source)
    if (tid < G && t > 1) {
        const unsigned long long want = (unsigned long long)wgs * (unsigned long long)(t - 1);
        const unsigned long long t0 = now100();
        unsigned long long spins = 0;
        while (__hip_atomic_load(&mine->arrived[tid * 16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
            __builtin_amdgcn_s_sleep(2);
            if ((++spins & 255) == 0 && now100() - t0 > 300000000ull) { s_fail = 1; break; }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    __syncthreads();
    if (s_fail) { if (tid == 0) atomicAdd(&errs[1], 1ull); }
    // check what the others wrote for iteration t-1 (a sample: the rows this workgroup owns, of every source)
    const int rows_per = ROWS / wgs;
    const int r0 = blockIdx.x * rows_per;
    if (t > 1 && !s_fail) {
        const int pb = (t - 1) & 1;
        for (int s = 0; s < G; ++s)
            for (int i = tid; i < rows_per * RW; i += blockDim.x) {
                const int row = r0 + i / RW, f = i % RW;
                const double got = mine->data[pb][s][row][f];
                const double want = (double)(t - 1) * 1000003.0 + s * 65537.0 + row * 8.0 + f;
                if (got != want) atomicAdd(&errs[0], 1ull);
            }
    }
    // publish iteration t to everybody (my own window included), then arrive
    const int b = t & 1;
    for (int p = 0; p < G; ++p) {
        Window* w = P.w[p];
        for (int i = tid; i < rows_per * RW; i += blockDim.x) {
            const int row = r0 + i / RW, f = i % RW;
            w->data[b][rank][row][f] = (double)t * 1000003.0 + rank * 65537.0 + row * 8.0 + f;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");   // system scope: the stores above are visible before the arrival below
    __syncthreads();
    if (tid < G) __hip_atomic_fetch_add(&P.w[tid]->arrived[rank * 16], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

static void file_barrier(const std::string& dir, const char* tag, int rank, int G) {
    char b[256];
    snprintf(b, sizeof b, "%s/%s_%d", dir.c_str(), tag, rank);
    FILE* f = fopen(b, "w"); fclose(f);
    for (int r = 0; r < G; ++r) {
        snprintf(b, sizeof b, "%s/%s_%d", dir.c_str(), tag, r);
        int tries = 0;
        while (access(b, F_OK) != 0) { usleep(1000); if (++tries > 120000) { fprintf(stderr, "barrier %s timed out\n", tag); _exit(4); } }
    }
}

static int child(int rank, int G, int mode, int iters, int wgs, const std::string& dir) {
    g_rank = rank;
    int ndev = 0;
    CHK(hipGetDeviceCount(&ndev));
    CHK(hipSetDevice(rank % ndev));
    Window* mine = nullptr;
    if (mode == 0) CHK(hipMalloc((void**)&mine, sizeof(Window)));
    else CHK(hipExtMallocWithFlags((void**)&mine, sizeof(Window), mode == 1 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached));
    CHK(hipMemset(mine, 0, sizeof(Window)));
    CHK(hipDeviceSynchronize());
    hipIpcMemHandle_t h;
    CHK(hipIpcGetMemHandle(&h, mine));
    char b[256];
    snprintf(b, sizeof b, "%s/handle_%d.tmp", dir.c_str(), rank);
    FILE* f = fopen(b, "wb"); fwrite(&h, sizeof h, 1, f); fclose(f);
    char b2[256];
    snprintf(b2, sizeof b2, "%s/handle_%d", dir.c_str(), rank);
    rename(b, b2);
    Peers P{};
    for (int r = 0; r < G; ++r) {
        if (r == rank) { P.w[r] = mine; continue; }
        snprintf(b2, sizeof b2, "%s/handle_%d", dir.c_str(), r);
        int tries = 0;
        while (access(b2, F_OK) != 0) { usleep(1000); if (++tries > 120000) { fprintf(stderr, "no handle of rank %d\n", r); _exit(4); } }
        hipIpcMemHandle_t hr;
        FILE* g = fopen(b2, "rb"); if (fread(&hr, sizeof hr, 1, g) != 1) _exit(5); fclose(g);
        void* p = nullptr;
        CHK(hipIpcOpenMemHandle(&p, hr, hipIpcMemLazyEnablePeerAccess));
        P.w[r] = (Window*)p;
    }
    unsigned long long* out = nullptr;
    CHK(hipMalloc((void**)&out, 64));
    CHK(hipMemset(out, 0, 64));
    hipStream_t st;
    CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    file_barrier(dir, "open", rank, G);
    // (1) flag latency
    if (rank < 2 && G >= 2) {
        const int n = 2000;
        hipLaunchKernelGGL(k_pingpong, dim3(1), dim3(1), 0, st, P, rank, n, out);
        CHK(hipStreamSynchronize(st));
        unsigned long long o[2];
        CHK(hipMemcpy(o, out, 16, hipMemcpyDeviceToHost));
        if (rank == 0) printf("ping-pong: %s, one way %.2f us (%d round trips)\n", o[1] ? "TIMED OUT" : "ok", o[0] * 0.01 / n / 2.0, n);
        CHK(hipMemset(out, 0, 64));
    }
    file_barrier(dir, "pp", rank, G);
    // (2) the launch-per-iteration protocol
    struct timespec a, z;
    clock_gettime(CLOCK_MONOTONIC, &a);
    for (int t = 1; t <= iters; ++t) hipLaunchKernelGGL(k_iter, dim3(wgs), dim3(256), 0, st, P, rank, G, t, wgs, out);
    CHK(hipStreamSynchronize(st));
    clock_gettime(CLOCK_MONOTONIC, &z);
    unsigned long long o[2];
    CHK(hipMemcpy(o, out, 16, hipMemcpyDeviceToHost));
    const double us = ((z.tv_sec - a.tv_sec) * 1e9 + (z.tv_nsec - a.tv_nsec)) / 1e3 / iters;
    printf("[rank %d] protocol: %d iterations, %.2f us per iteration, data errors %llu, wait time-outs %llu\n", rank, iters, us, o[0], o[1]);
    file_barrier(dir, "done", rank, G);   // nobody unmaps a window somebody still writes
    for (int r = 0; r < G; ++r) if (r != rank) CHK(hipIpcCloseMemHandle(P.w[r]));
    CHK(hipFree(mine));
    fflush(stdout);
    return (o[0] || o[1]) ? 1 : 0;
}

int main(int argc, char** argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 2, mode = argc > 2 ? atoi(argv[2]) : 1, iters = argc > 3 ? atoi(argv[3]) : 2000,
              wgs = argc > 4 ? atoi(argv[4]) : 64;
    if (G < 1 || G > MAXG || ROWS % wgs) { fprintf(stderr, "bad arguments\n"); return 2; }
    char tmpl[] = "/tmp/ipc_probe_XXXXXX";
    const std::string dir = mkdtemp(tmpl);
    printf("ipc_probe: %d processes, allocation mode %d, %d workgroups per launch\n", G, mode, wgs);
    fflush(stdout);
    std::vector<pid_t> pids;
    for (int r = 0; r < G; ++r) {   // fork BEFORE any HIP call
        pid_t p = fork();
        if (p == 0) _exit(child(r, G, mode, iters, wgs, dir));
        pids.push_back(p);
    }
    int rc = 0;
    for (pid_t p : pids) { int s = 0; waitpid(p, &s, 0); if (!WIFEXITED(s) || WEXITSTATUS(s)) rc = 1; }
    printf("ipc_probe: %s\n", rc ? "FAILED" : "ok");
    return rc;
}
