// Probe for the peer-store communication backend (ldu_comm.cpp "peer"): N processes, each with a window of
// fine-grained device memory exported with hipIpcGetMemHandle and mapped by every other process
// (hipIpcOpenMemHandle) - on one GPU (N processes share device 0) or on N GPUs (process r on device r % count).
// Measures (1) that kernels of different processes exchange tagged 16-byte granules through the windows while both
// are running, (2) the time of one all-reduce of 4 doubles done that way (one single-wavefront kernel per rank and
// reduction: store to every peer, poll the own window, sum in rank order), (3) a halo-style pack-and-store of
// n faces followed by the consumer's poll.
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/ipc_probe tools/ipc_probe.hip ; tools/bin/ipc_probe 2 [finegrained=1]
#include <hip/hip_runtime.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "[rank %d] %s: %s\n", g_rank, #e, hipGetErrorString(_e)); exit(3); } } while (0)
static int g_rank = 0;
#define MAXR 8
struct Peers { uint4* win[MAXR]; };
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void store_granule(uint4* p, double v, unsigned tag)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    u32x4 g;
    g.x = (unsigned)b; g.y = tag; g.z = (unsigned)(b >> 32); g.w = tag;
    // system scope: write-through past this device's L2 (the window may live on another GPU)
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p), "v"(g) : "memory");
}
__device__ __forceinline__ bool load_granule(const uint4* p, unsigned tag, double& v)
{
    u32x4 g;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(g) : "v"(p) : "memory");
    if (g.y != tag || g.w != tag) return false;
    v = __longlong_as_double((long long)(((unsigned long long)g.z << 32) | g.x));
    return true;
}

// all-reduce of `count` doubles: slot layout in every window: [parity][rank][16]
__global__ void allreduce_kernel(Peers P, int me, int n, int count, unsigned seq, const double* in, double* out, int* fail)
{
    __shared__ double v[MAXR][16];
    const int lane = threadIdx.x;
    const int r = lane / 16, i = lane % 16;
    const int par = seq & 1;
    if (r < n && i < count) store_granule(P.win[r] + (par * MAXR + me) * 16 + i, in[i], seq);
    if (r < n && i < count)
    {
        double x = 0;
        unsigned spins = 0;
        while (!load_granule(P.win[me] + (par * MAXR + r) * 16 + i, seq, x))
        {
            if (++spins > 5000000u) { *fail = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        v[r][i] = x;
    }
    __syncthreads();
    if (lane < count)
    {
        double t = v[0][lane];
        for (int q = 1; q < n; q++) t += v[q][lane];
        out[lane] = t;
    }
}

// persistent ping-pong between rank 0 and rank 1 inside ONE launch per process
__global__ void pingpong_kernel(Peers P, int me, int iters, unsigned long long* ticks, int* fail)
{
    if (threadIdx.x) return;
    uint4* mine = P.win[me] + 4096;
    uint4* theirs = P.win[1 - me] + 4096;
    const unsigned long long t0 = wall_clock64();
    for (int it = 1; it <= iters; it++)
    {
        double x;
        unsigned spins = 0;
        if (me == 0)
        {
            store_granule(theirs, (double)it, (unsigned)it);
            while (!load_granule(mine, (unsigned)it, x)) if (++spins > 200000000u) { *fail = 1; return; }
        }
        else
        {
            while (!load_granule(mine, (unsigned)it, x)) if (++spins > 200000000u) { *fail = 1; return; }
            store_granule(theirs, x + 1.0, (unsigned)it);
        }
    }
    *ticks = wall_clock64() - t0;
}

__global__ void halo_send_kernel(uint4* dst, const double* x, int n, unsigned seq)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) store_granule(dst + i, x[i], seq);
}
__global__ void halo_recv_kernel(const uint4* src, double* y, int n, unsigned seq, int* fail)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = 0;
    unsigned spins = 0;
    while (!load_granule(src + i, seq, v)) { if (++spins > 5000000u) { *fail = 1; break; } __builtin_amdgcn_s_sleep(1); }
    y[i] = v;
}

__global__ void halo_fused_kernel(uint4* dst, const uint4* src, const double* x, double* y, int n, unsigned seq, int* fail)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    store_granule(dst + i, x[i], seq);
    double v = 0;
    unsigned spins = 0;
    while (!load_granule(src + i, seq, v)) { if (++spins > 5000000u) { *fail = 1; break; } __builtin_amdgcn_s_sleep(1); }
    y[i] = v;
}

static void xwrite(int fd, const void* p, size_t n) { if (write(fd, p, n) != (ssize_t)n) { perror("write"); exit(4); } }
static void xread(int fd, void* p, size_t n)
{
    size_t got = 0;
    while (got < n) { ssize_t k = read(fd, (char*)p + got, n - got); if (k <= 0) { perror("read"); exit(4); } got += k; }
}

int main(int argc, char** argv)
{
    setvbuf(stdout, nullptr, _IOLBF, 0);
    const int n = argc > 1 ? atoi(argv[1]) : 2;
    const int fine = argc > 2 ? atoi(argv[2]) : 1;
    const int arIters = argc > 3 ? atoi(argv[3]) : 2000;
    if (n < 1 || n > MAXR) return 2;
    // fork BEFORE any HIP call; star topology of pipes through rank 0 for the bootstrap and the barriers
    int up[MAXR][2], down[MAXR][2];
    for (int r = 1; r < n; r++) { if (pipe(up[r]) || pipe(down[r])) return 2; }
    for (int r = 1; r < n; r++)
    {
        pid_t pid = fork();
        if (pid == 0) { g_rank = r; break; }
    }
    auto barrier = [&]() {
        char c = 1;
        if (g_rank == 0) { for (int r = 1; r < n; r++) xread(up[r][0], &c, 1); for (int r = 1; r < n; r++) xwrite(down[r][1], &c, 1); }
        else { xwrite(up[g_rank][1], &c, 1); xread(down[g_rank][0], &c, 1); }
    };
    int ndev = 0;
    CK(hipGetDeviceCount(&ndev));
    CK(hipSetDevice(g_rank % ndev));
    const size_t winBytes = 8u << 20;
    uint4* win = nullptr;
    if (fine) CK(hipExtMallocWithFlags((void**)&win, winBytes, hipDeviceMallocFinegrained));
    else CK(hipMalloc((void**)&win, winBytes));
    CK(hipMemset(win, 0, winBytes));
    CK(hipDeviceSynchronize());
    hipIpcMemHandle_t h[MAXR];
    CK(hipIpcGetMemHandle(&h[g_rank], win));
    if (g_rank == 0)
    {
        for (int r = 1; r < n; r++) xread(up[r][0], &h[r], sizeof(h[r]));
        for (int r = 1; r < n; r++) xwrite(down[r][1], h, sizeof(hipIpcMemHandle_t) * n);
    }
    else
    {
        xwrite(up[g_rank][1], &h[g_rank], sizeof(h[0]));
        xread(down[g_rank][0], h, sizeof(hipIpcMemHandle_t) * n);
    }
    printf("[rank %d] window exported, handles exchanged\n", g_rank);
    Peers P;
    memset(&P, 0, sizeof(P));
    for (int r = 0; r < n; r++)
    {
        if (r == g_rank) { P.win[r] = win; continue; }
        void* p = nullptr;
        CK(hipIpcOpenMemHandle(&p, h[r], hipIpcMemLazyEnablePeerAccess));
        P.win[r] = (uint4*)p;
    }
    if (g_rank == 0) printf("ipc_probe: %d ranks on %d device(s), %s windows mapped\n", n, ndev, fine ? "fine-grained" : "coarse-grained");
    double *d_in, *d_out; int* d_fail; unsigned long long* d_ticks;
    CK(hipMalloc((void**)&d_in, 128)); CK(hipMalloc((void**)&d_out, 128)); CK(hipMalloc((void**)&d_fail, 4)); CK(hipMalloc((void**)&d_ticks, 8));
    CK(hipMemset(d_fail, 0, 4));
    double hin[4] = {1.0 + g_rank, 0.5 * g_rank, -2.0, 1e-3 * g_rank};
    CK(hipMemcpy(d_in, hin, 32, hipMemcpyHostToDevice));
    hipStream_t s; CK(hipStreamCreate(&s));
    unsigned seq = 0;
    // (2) all-reduce: correctness + time per operation, back to back on a stream
    barrier();
    for (int rep = 0; rep < 2; rep++)
    {
        const int iters = arIters;
        barrier();
        auto t0 = std::chrono::steady_clock::now();
        for (int it = 0; it < iters; it++) allreduce_kernel<<<1, 128, 0, s>>>(P, g_rank, n, 4, ++seq, d_in, d_out, d_fail);
        CK(hipStreamSynchronize(s));
        const double us = 1e6 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / iters;
        double hout[4]; int fail = 0;
        CK(hipMemcpy(hout, d_out, 32, hipMemcpyDeviceToHost)); CK(hipMemcpy(&fail, d_fail, 4, hipMemcpyDeviceToHost));
        double e0 = 0; for (int r = 0; r < n; r++) e0 += 1.0 + r;
        printf("[rank %d] allreduce rep %d: %.2f us per op, sum0 %.3f (expect %.3f) fail %d\n", g_rank, rep, us, hout[0], e0, fail);
    }
    // (1) ping-pong inside one launch
    if (n >= 2)
    {
        barrier();
        if (g_rank < 2)
        {
            pingpong_kernel<<<1, 64, 0, s>>>(P, g_rank, 2000, d_ticks, d_fail);
            CK(hipStreamSynchronize(s));
            unsigned long long t; int fail;
            CK(hipMemcpy(&t, d_ticks, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&fail, d_fail, 4, hipMemcpyDeviceToHost));
            printf("[rank %d] ping-pong: %.2f us per round trip (fail %d)\n", g_rank, t / 100.0 / 2000, fail);
        }
        barrier();
    }
    // (3) halo patterns.  mode 0: ring, two kernels (send to r+1, then poll what r-1 sent); mode 1: the pairwise pattern of
    // processor patches (send to r+1 AND r-1, then poll both); mode 2: ring, one fused kernel (store, then poll)
    const int haloIters = argc > 4 ? atoi(argv[4]) : 200;
    for (int mode = 0; mode < 3; mode++)
    for (int nf : {64, 46656})
    {
        double *x, *y;
        CK(hipMalloc((void**)&x, nf * 8)); CK(hipMalloc((void**)&y, 2 * nf * 8));
        std::vector<double> hx(nf);
        for (int i = 0; i < nf; i++) hx[i] = g_rank * 1000.0 + i;
        CK(hipMemcpy(x, hx.data(), nf * 8, hipMemcpyHostToDevice));
        const int to = (g_rank + 1) % n, from = (g_rank + n - 1) % n;
        const int grid = (nf + 255) / 256;
        barrier();
        auto t0 = std::chrono::steady_clock::now();
        for (int it = 0; it < haloIters; it++)
        {
            ++seq;
            // slot A (from the left neighbour) at 8192, slot B (from the right neighbour) at 8192 + 131072; parity doubles
            const size_t offA = 8192 + (seq & 1) * 65536, offB = offA + 131072;
            if (mode == 2) halo_fused_kernel<<<grid, 256, 0, s>>>(P.win[to] + offA, P.win[g_rank] + offA, x, y, nf, seq, d_fail);
            else
            {
                halo_send_kernel<<<grid, 256, 0, s>>>(P.win[to] + offA, x, nf, seq);
                if (mode == 1) halo_send_kernel<<<grid, 256, 0, s>>>(P.win[from] + offB, x, nf, seq);
                halo_recv_kernel<<<grid, 256, 0, s>>>(P.win[g_rank] + offA, y, nf, seq, d_fail);
                if (mode == 1) halo_recv_kernel<<<grid, 256, 0, s>>>(P.win[g_rank] + offB, y + nf, nf, seq, d_fail);
            }
        }
        CK(hipStreamSynchronize(s));
        const double us = 1e6 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / haloIters;
        std::vector<double> hy(nf); int fail;
        CK(hipMemcpy(hy.data(), y, nf * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&fail, d_fail, 4, hipMemcpyDeviceToHost));
        int bad = 0; for (int i = 0; i < nf; i++) bad += hy[i] != from * 1000.0 + i;
        printf("[rank %d] halo mode %d, %d faces: %.2f us per exchange, %d wrong, fail %d\n", g_rank, mode, nf, us, bad, fail);
        CK(hipFree(x)); CK(hipFree(y));
    }
    barrier();
    for (int r = 0; r < n; r++) if (r != g_rank) CK(hipIpcCloseMemHandle(P.win[r]));
    if (g_rank == 0) { int st; while (wait(&st) > 0) {} printf("ipc_probe: done\n"); }
    return 0;
}
