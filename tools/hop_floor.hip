// TOOL: what ONE dependency hop can cost on this chip - the floor the sweep engines' hand-offs
// (DESIGN: 0.64 us inside gs_wg_kernel, ~1.2 us inside a block of the block engine, ~3 us across blocks) are compared with.
//
// A "hop": producer finishes a row -> publishes {value, stamp} -> the consumer, which polls, sees it -> (optionally) runs the
// arithmetic chain of a GaussSeidel row on it (8 dependent f64 subtractions of products + one divide, GaussSeidelSmoother.C:
// 151-176) -> publishes its own result.  N hops back and forth between two parties = 2 N hand-offs; time by the 100 MHz wall
// clock (wall_clock64) of the first party.
//   lds      two wavefronts of ONE workgroup through LDS (value 8 B + stamp 4 B, value written first, stamp after it)
//   lds1     ONE wavefront, two lanes, through LDS (what a chain inside one wavefront costs: no cross-wave visibility wait)
//   l2       two workgroups on the SAME XCD through a 16-byte {value, tag} granule, sc1 store / sc1 polling load
//   xcd      two workgroups on DIFFERENT XCDs, the same granule
//   +chain   each of the above with the row arithmetic between receive and send
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/hop_floor tools/hop_floor.hip ; run: /tmp/hop_floor [N]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ double row_chain(double x, const double* c, double rhs, double rD)
{
    // eight neighbours: bPrime -= coeff * psi (dependent subtractions), then psi = bPrime / diag
    double b = rhs;
#pragma unroll
    for (int k = 0; k < 8; k++) b -= c[k] * x;
    return b / rD;
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void st_granule(uint4* p, double v, unsigned tag)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    u32x4 d;
    d.x = (unsigned)u; d.y = tag; d.z = (unsigned)(u >> 32); d.w = tag;
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(d) : "memory");
}
__device__ __forceinline__ u32x4 ld_granule(const uint4* p)
{
    u32x4 g;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(g) : "v"(p) : "memory");
    return g;
}

// ---- LDS, two wavefronts of one workgroup
template <bool CHAIN>
__global__ void __launch_bounds__(128) k_lds(int n, const double* coef, long long* out)
{
    __shared__ volatile double val[2];
    __shared__ volatile unsigned stamp[2];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x < 2) { val[threadIdx.x] = 1.0; stamp[threadIdx.x] = 0; }
    __syncthreads();
    double c[8];
    for (int k = 0; k < 8; k++) c[k] = coef[k];
    const double rhs = coef[8], rD = coef[9];
    double x = 1.0;
    const long long t0 = wall_clock64();
    // wave 0 publishes hop 1, 3, 5 ... into slot 0; wave 1 answers with 2, 4, ... into slot 1
    for (int i = 1; i <= n; i++)
    {
        if (w == 0)
        {
            if (lane == 0) { val[0] = x; __threadfence_block(); stamp[0] = (unsigned)i; }
            while (stamp[1] != (unsigned)i) {}
            x = val[1];
            if (CHAIN) x = row_chain(x, c, rhs, rD);
        }
        else
        {
            while (stamp[0] != (unsigned)i) {}
            x = val[0];
            if (CHAIN) x = row_chain(x, c, rhs, rD);
            if (lane == 0) { val[1] = x; __threadfence_block(); stamp[1] = (unsigned)i; }
        }
    }
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = (long long)x; }
}

// ---- LDS, inside ONE wavefront: lane 0 -> LDS -> all lanes read -> lane 0 (the chain of a block whose rows sit in one wave)
template <bool CHAIN>
__global__ void __launch_bounds__(64) k_lds1(int n, const double* coef, long long* out)
{
    __shared__ volatile double val[2];
    if (threadIdx.x < 2) val[threadIdx.x] = 1.0;
    __syncthreads();
    double c[8];
    for (int k = 0; k < 8; k++) c[k] = coef[k];
    const double rhs = coef[8], rD = coef[9];
    double x = 1.0;
    const long long t0 = wall_clock64();
    for (int i = 1; i <= 2 * n; i++)
    {
        if (threadIdx.x == 0) val[i & 1] = x;
        __builtin_amdgcn_s_waitcnt(0xc07f);        // lgkmcnt(0): the write has reached LDS; a wavefront sees its own LDS writes in order
        x = val[i & 1];
        if (CHAIN) x = row_chain(x, c, rhs, rD);
    }
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = (long long)x; }
}

// ---- global granule between two workgroups; which two is decided on the host from the XCC ids of a first pass
template <bool CHAIN>
__global__ void __launch_bounds__(64) k_granule(int n, int wgA, int wgB, uint4* gran, const double* coef, long long* out, int* xccOut, unsigned epoch)
{
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0 && xccOut) xccOut[blockIdx.x] = (int)(xcc & 0xf);
    if ((int)blockIdx.x != wgA && (int)blockIdx.x != wgB) return;
    const bool first = (int)blockIdx.x == wgA;
    double c[8];
    for (int k = 0; k < 8; k++) c[k] = coef[k];
    const double rhs = coef[8], rD = coef[9];
    double x = 1.0;
    uint4* mine = gran + (first ? 0 : 16);         // 256 bytes apart: two lines
    const uint4* theirs = gran + (first ? 16 : 0);
    long long spins = 0;
    const long long t0 = wall_clock64();
    for (int i = 1; i <= n; i++)
    {
        const unsigned tag = epoch + (unsigned)i;
        if (first)
        {
            if (threadIdx.x == 0) st_granule(mine, x, tag);
            u32x4 g;
            do { g = ld_granule(theirs); spins++; } while (g.y != tag || g.w != tag);
            x = __longlong_as_double((long long)(((unsigned long long)g.z << 32) | g.x));
            if (CHAIN) x = row_chain(x, c, rhs, rD);
        }
        else
        {
            u32x4 g;
            do { g = ld_granule(theirs); spins++; } while (g.y != tag || g.w != tag);
            x = __longlong_as_double((long long)(((unsigned long long)g.z << 32) | g.x));
            if (CHAIN) x = row_chain(x, c, rhs, rD);
            if (threadIdx.x == 0) st_granule(mine, x, tag);
        }
    }
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0 && first) { out[0] = t1 - t0; out[1] = (long long)x; out[2] = spins; }
}

// the arithmetic chain alone (no hand-off): n rows one after the other in one wavefront
__global__ void __launch_bounds__(64) k_chain(int n, const double* coef, long long* out)
{
    double c[8];
    for (int k = 0; k < 8; k++) c[k] = coef[k];
    const double rhs = coef[8], rD = coef[9];
    double x = 1.0;
    const long long t0 = wall_clock64();
    for (int i = 0; i < n; i++) x = row_chain(x, c, rhs, rD);
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = (long long)x; }
}

static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 2000;
    const int reps = 9;
    double hc[10] = {0.11, -0.12, 0.13, -0.14, 0.15, -0.16, 0.17, -0.18, 0.3, 1.7};
    double* coef; CHECK(hipMalloc(&coef, sizeof(hc))); CHECK(hipMemcpy(coef, hc, sizeof(hc), hipMemcpyHostToDevice));
    long long* out; CHECK(hipMalloc(&out, 64)); CHECK(hipMemset(out, 0, 64));
    uint4* gran; CHECK(hipMalloc(&gran, 4096)); CHECK(hipMemset(gran, 0, 4096));
    const int nWG = 64;
    int* xcc; CHECK(hipMalloc(&xcc, sizeof(int) * nWG));
    long long h[3];
    auto tick_us = [](long long t) { return (double)t / 100.0; };     // 100 MHz
    printf("# hop floor on this device: N = %d round trips per run, median of %d runs; us per HAND-OFF (a round trip = 2)\n", n, reps);
    {
        std::vector<double> v;
        for (int r = 0; r < reps; r++) { k_chain<<<1, 64>>>(2 * n, coef, out); CHECK(hipDeviceSynchronize()); CHECK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost)); v.push_back(tick_us(h[0]) / (2.0 * n)); }
        printf("%-34s %8.3f us per row\n", "row arithmetic alone (8 fma-free sub-mul + divide)", med(v));
    }
    for (int chain = 0; chain < 2; chain++)
    {
        std::vector<double> v;
        for (int r = 0; r < reps; r++)
        {
            if (chain) k_lds1<true><<<1, 64>>>(n, coef, out); else k_lds1<false><<<1, 64>>>(n, coef, out);
            CHECK(hipDeviceSynchronize()); CHECK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost)); v.push_back(tick_us(h[0]) / (2.0 * n));
        }
        printf("%-34s %8.3f us per hand-off\n", chain ? "lds1 (one wavefront) + row chain" : "lds1 (one wavefront)", med(v));
    }
    for (int chain = 0; chain < 2; chain++)
    {
        std::vector<double> v;
        for (int r = 0; r < reps; r++)
        {
            if (chain) k_lds<true><<<1, 128>>>(n, coef, out); else k_lds<false><<<1, 128>>>(n, coef, out);
            CHECK(hipDeviceSynchronize()); CHECK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost)); v.push_back(tick_us(h[0]) / (2.0 * n));
        }
        printf("%-34s %8.3f us per hand-off\n", chain ? "lds (two wavefronts) + row chain" : "lds (two wavefronts)", med(v));
    }
    // where do the workgroups of a 64-workgroup launch land?
    unsigned epoch = 1000;
    k_granule<false><<<nWG, 64>>>(0, 0, 1, gran, coef, out, xcc, epoch);
    CHECK(hipDeviceSynchronize());
    std::vector<int> hx(nWG);
    CHECK(hipMemcpy(hx.data(), xcc, sizeof(int) * nWG, hipMemcpyDeviceToHost));
    printf("# XCC of workgroups 0..15:");
    for (int i = 0; i < 16; i++) printf(" %d", hx[i]);
    printf("\n");
    int same = -1, other = -1;
    for (int i = 1; i < nWG && (same < 0 || other < 0); i++)
    {
        if (hx[i] == hx[0] && same < 0) same = i;
        if (hx[i] != hx[0] && other < 0) other = i;
    }
    for (int where = 0; where < 2; where++)
    {
        const int b = where ? other : same;
        if (b < 0) { printf("%s: no such pair of workgroups in this launch\n", where ? "xcd" : "l2"); continue; }
        for (int chain = 0; chain < 2; chain++)
        {
            std::vector<double> v, sp;
            for (int r = 0; r < reps; r++)
            {
                epoch += 4 * n + 16;
                // (the placement is re-read every run: it is the hardware's choice)
                if (chain) k_granule<true><<<nWG, 64>>>(n, 0, b, gran, coef, out, xcc, epoch);
                else k_granule<false><<<nWG, 64>>>(n, 0, b, gran, coef, out, xcc, epoch);
                CHECK(hipDeviceSynchronize());
                CHECK(hipMemcpy(h, out, 24, hipMemcpyDeviceToHost));
                CHECK(hipMemcpy(hx.data(), xcc, sizeof(int) * nWG, hipMemcpyDeviceToHost));
                const bool ok = where ? hx[0] != hx[b] : hx[0] == hx[b];
                if (ok) { v.push_back(tick_us(h[0]) / (2.0 * n)); sp.push_back((double)h[2] / n); }
            }
            char name[80];
            snprintf(name, sizeof(name), "%s granule%s", where ? "xcd (different XCDs)" : "l2 (same XCD)", chain ? " + row chain" : "");
            if (v.empty()) printf("%-34s placement changed between runs\n", name);
            else printf("%-34s %8.3f us per hand-off (%.1f polls per receive, %zu runs)\n", name, med(v), med(sp), v.size());
        }
    }
    return 0;
}
