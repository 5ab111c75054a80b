// latency / throughput of the per-sample embedding gather: NL loads per lane issued back to back, then one wait
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
template <int NL, int W>   // W = dwords per load (1 or 4)
__global__ __launch_bounds__(512) void k(const float *tab, const int *idx, float *out, unsigned long long *clk, int iters, int rows)
{
    typedef float vec __attribute__((ext_vector_type(W)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = 0.f;
    unsigned long long tot = 0;
    int seed = blockIdx.x * 7 + 1;
    for (int it = 0; it < iters; ++it) {
        vec v[NL];
        __syncthreads();
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            seed = (seed * 1103515245 + 12345) & 0x7fffffff;
            const int r = __builtin_amdgcn_readfirstlane((seed >> 8) % rows);           // uniform row
            v[i] = *(const vec *)(tab + ((size_t)r * 1536 + (i % (12 / W)) * 512 * W / 4 * 0 + threadIdx.x * W));
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) asm volatile("" :: "v"(v[i][0]));
        asm volatile("s_waitcnt vmcnt(0)");
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        tot += t1 - t0;
#pragma unroll
        for (int i = 0; i < NL; ++i) { acc += v[i][0]; }
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc;
    if (lane == 0 && blockIdx.x == 0) clk[wave] = tot / iters;
}
int main()
{
    const int rows = 768;                       // 768 rows x 6 KB = 4.7 MB table set
    float *tab; int *idx; float *out; unsigned long long *clk;
    hipMalloc(&tab, (size_t)rows * 1536 * 4 + 65536); hipMemset(tab, 0, (size_t)rows * 1536 * 4 + 65536);
    hipMalloc(&idx, 4096); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 64);
    unsigned long long c[8];
    auto run = [&](const char *name, auto kern, int grid, int r) {
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, 0, tab, idx, out, clk, 200, r);
        hipDeviceSynchronize();
        hipMemcpy(c, clk, sizeof(c), hipMemcpyDeviceToHost);
        printf("%-34s grid=%3d rows=%3d clk:", name, grid, r);
        for (int w = 0; w < 8; ++w) printf(" %5llu", c[w]);
        printf("\n");
    };
    for (int grid : {1, 256}) for (int r : {16, 768}) {
        run("36 x dword / lane", k<36, 1>, grid, r);
        run("12 x dword / lane", k<12, 1>, grid, r);
        run(" 9 x dwordx4 / lane", k<9, 4>, grid, r);
        run(" 1 x dword / lane", k<1, 1>, grid, r);
    }
    return 0;
}
