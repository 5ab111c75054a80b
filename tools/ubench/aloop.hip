// Isolated model of the GRU-A item loop: per item 1 ds_read_b128 + 16 (mul [+dpp]) + 16 add, 4 chains.
#include <hip/hip_runtime.h>
#include <stdio.h>
#pragma clang diagnostic ignored "-Wunused-value"
template <int SEL> __device__ __forceinline__ float qb(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), SEL * 0x55, 0xf, 0xf, true));
}
template <bool DPP, bool LDS, int NW>
__global__ __launch_bounds__(512) void k(float *out, unsigned long long *clk, int n, const float *win)
{
    __shared__ float4 h[512];
    for (int i = threadIdx.x; i < 512; i += blockDim.x) h[i] = make_float4(i * 0.001f, 0.5f, 0.25f, 0.125f);
    float4 w[NW];
#pragma unroll
    for (int j = 0; j < NW; ++j) w[j] = ((const float4 *)win)[j * 512 + threadIdx.x];
    __syncthreads();
    float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    const int lane = threadIdx.x & 63;
    float4 hreg = h[lane];
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            float4 hv = LDS ? h[(lane * 7 + j * 13 + it) & 511] : hreg;
            const float hk[4] = {hv.x, hv.y, hv.z, hv.w};
            const float wk[4] = {w[j].x, w[j].y, w[j].z, w[j].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float t0, t1, t2, t3;
                if (DPP) { t0 = wk[c] * qb<0>(hk[c]); t1 = wk[c] * qb<1>(hk[c]); t2 = wk[c] * qb<2>(hk[c]); t3 = wk[c] * qb<3>(hk[c]); }
                else { t0 = wk[c] * hk[c]; t1 = wk[c] * hk[(c + 1) & 3]; t2 = wk[c] * hk[(c + 2) & 3]; t3 = wk[c] * hk[(c + 3) & 3]; }
                a0 = a0 + t0; a1 = a1 + t1; a2 = a2 + t2; a3 = a3 + t3;
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = a0 + a1 + a2 + a3;
    if (lane == 0) clk[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}
int main()
{
    float *d_out, *d_w; unsigned long long *d_clk;
    hipMalloc(&d_out, 4096 * 4); hipMalloc(&d_clk, 4096 * 8); hipMalloc(&d_w, 40 * 512 * 16); hipMemset(d_w, 0, 40 * 512 * 16);
    const int N = 200;
    unsigned long long c[8];
    auto run = [&](const char *name, auto kern, int threads, int blocks) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d_out, d_clk, N, d_w);
        hipDeviceSynchronize();
        hipMemcpy(c, d_clk, sizeof(c), hipMemcpyDeviceToHost);
        printf("%-28s thr=%3d blocks=%3d  clk/item per wave:", name, threads, blocks);
        for (int w = 0; w < threads / 64; ++w) printf(" %6.1f", (double)c[w] / (N * 32.0));
        printf("\n");
    };
    for (int blocks : {1, 256}) {
        run("dpp+lds", k<true, true, 32>, 64, blocks);
        run("dpp+lds", k<true, true, 32>, 256, blocks);
        run("dpp+lds", k<true, true, 32>, 512, blocks);
        run("dpp, no lds", k<true, false, 32>, 512, blocks);
        run("no dpp, lds", k<false, true, 32>, 512, blocks);
        run("no dpp, no lds", k<false, false, 32>, 512, blocks);
        run("no dpp, no lds", k<false, false, 32>, 64, blocks);
    }
    return 0;
}
