// What shader clock does a latency-bound, nearly idle GPU run at?  One wavefront (and 256 x 4 wavefronts) spin for
// a fixed number of shader cycles (clock64 / s_memtime) while the 100 MHz wall clock measures the time.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(long long cycles, long long* out)
{
    const long long w0 = wall_clock64();
    const long long c0 = clock64();
    double a = threadIdx.x;
    while (clock64() - c0 < cycles) a = a * 1.0000001 + 1e-9;
    const long long w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = w1 - w0; out[1] = (long long)a; }
}
// dependent f64 chain: n iterations of x = x*a + b (mul, add: 2 dependent ops)
__global__ void chain(int n, double* io, long long* out)
{
    double x = io[threadIdx.x], a = 1.0000001, b = 1e-9;
    const long long w0 = wall_clock64();
    const long long c0 = clock64();
    for (int i = 0; i < n; i++) { x = x * a; x = x + b; }
    const long long c1 = clock64();
    const long long w1 = wall_clock64();
    io[threadIdx.x] = x;
    if (threadIdx.x == 0) { out[0] = w1 - w0; out[1] = c1 - c0; }
}
__global__ void chain_add(int n, double* io, long long* out)
{
    double x = io[threadIdx.x], b = io[threadIdx.x + 1] + 1e-9;
    const long long w0 = wall_clock64();
    for (int i = 0; i < n; i++) x = x - b;
    const long long w1 = wall_clock64();
    io[threadIdx.x] = x;
    if (threadIdx.x == 0) out[0] = w1 - w0;
}
__global__ void chain_div(int n, double* io, long long* out)
{
    double x = io[threadIdx.x] + 3.0, d = io[threadIdx.x + 1] + 1.0000001;
    const long long w0 = wall_clock64();
    for (int i = 0; i < n; i++) x = x / d;
    const long long w1 = wall_clock64();
    io[threadIdx.x] = x;
    if (threadIdx.x == 0) out[0] = w1 - w0;
}
__global__ void chain_lds(int n, int* io, long long* out)
{
    __shared__ int a[256];
    for (int i = threadIdx.x; i < 256; i += 64) a[i] = (i * 7 + 3) & 255;
    __syncthreads();
    int j = io[threadIdx.x] & 255;
    const long long w0 = wall_clock64();
    for (int i = 0; i < n; i++) j = a[j];
    const long long w1 = wall_clock64();
    io[threadIdx.x] = j;
    if (threadIdx.x == 0) out[0] = w1 - w0;
}
int main()
{
    long long* d; hipMalloc(&d, 64); double* io; hipMalloc(&io, 64 * 8); hipMemset(io, 0, 512);
    long long h[2];
    for (int rep = 0; rep < 3; rep++)
        for (int grid : {1, 1024})
        {
            spin<<<grid, 64>>>(20000000LL, d);
            hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
            printf("grid %4d: 2e7 shader cycles took %.3f ms wall -> clock64 runs at %.0f MHz\n", grid, h[0] / 1e5, 2e7 / (h[0] / 100.0));
        }
    chain<<<1, 64>>>(100000, io, d);
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("dependent f64 mul+add x 1e5: %.3f ms wall, %.1f ns per op pair, %.1f clock64 ticks per pair\n", h[0] / 1e5, h[0] * 10.0 / 1e5, (double)h[1] / 1e5);
    chain_add<<<1, 64>>>(100000, io, d);
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("dependent f64 sub x 1e5: %.1f ns per op\n", h[0] * 10.0 / 1e5);
    chain_div<<<1, 64>>>(100000, io, d);
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("dependent f64 div x 1e5: %.1f ns per division\n", h[0] * 10.0 / 1e5);
    chain_lds<<<1, 64>>>(100000, (int*)io, d);
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("dependent LDS read x 1e5: %.1f ns per read\n", h[0] * 10.0 / 1e5);
    return 0;
}
