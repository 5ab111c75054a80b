// GRU-B inner loop in isolation: per block 1 ds_read_b128 (row weights, distinct addresses) + 4 dependent adds
// with 4 DPP multiplies in their shadow.  Variants separate the LDS cost from the VALU chain.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int SEL> __device__ __forceinline__ float qb(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), SEL * 0x55, 0xf, 0xf, true));
}
#define SB __builtin_amdgcn_sched_barrier(0)
// MODE 0: loads + chain, 1: chain only (weights constant), 2: loads only, 3: chain without DPP, 4: adds only,
// 5: loads + chain without DPP (the state operand as a plain register / SGPR: what an s_load-fed GRU-B would cost)
template <int MODE>
__global__ __launch_bounds__(512) void k(float *out, unsigned long long *clk, int nblk, int active_waves)
{
    extern __shared__ float4 lds[];
    for (int i = threadIdx.x; i < 4800; i += blockDim.x) lds[i] = make_float4(1.0f + i * 1e-6f, 1.f, 1.f, 1.f);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float z = out[threadIdx.x];
    unsigned long long t0 = 0, t1 = 0;
    if (wave < active_waves) {
        const float4 *wp = lds + (lane & 7) + 8 * (lane >> 3);       // 64 distinct 16-byte addresses
        float4 h = lds[4000 + (lane & 3)];
        float4 w0 = wp[0], w1 = wp[64], w2 = wp[128], w3;
        float p0 = w0.x, p1 = w0.y, p2 = w0.z, p3 = w0.w;
        if (MODE == 1 || MODE == 3 || MODE == 4) w3 = w2;
        t0 = __builtin_amdgcn_s_memtime();
        for (int b = 0; b < nblk; b += 4) {
#define STEP(WL, OFF, WN, K)                                                                    \
            {                                                                                   \
                if (MODE == 0 || MODE == 2 || MODE == 5) WL = wp[OFF];                                       \
                SB;                                                                             \
                if (MODE != 2) {                                                                \
                    float t0_, t1_, t2_, t3_;                                                   \
                    z = z + p0; SB; if (MODE == 4) t0_ = p0; else if (MODE == 3 || MODE == 5) t0_ = WN.x * h.x; else t0_ = WN.x * qb<K>(h.x); SB; \
                    z = z + p1; SB; if (MODE == 4) t1_ = p1; else if (MODE == 3 || MODE == 5) t1_ = WN.y * h.y; else t1_ = WN.y * qb<K>(h.y); SB; \
                    z = z + p2; SB; if (MODE == 4) t2_ = p2; else if (MODE == 3 || MODE == 5) t2_ = WN.z * h.z; else t2_ = WN.z * qb<K>(h.z); SB; \
                    z = z + p3; SB; if (MODE == 4) t3_ = p3; else if (MODE == 3 || MODE == 5) t3_ = WN.w * h.w; else t3_ = WN.w * qb<K>(h.w); SB; \
                    p0 = t0_; p1 = t1_; p2 = t2_; p3 = t3_;                                     \
                } else { asm volatile("" :: "v"(WL.x), "v"(WL.w)); }                            \
            }
            STEP(w3, 192, w1, 1)
            STEP(w0, 256, w2, 2)
            STEP(w1, 320, w3, 3)
            STEP(w2, 384, w0, 0)
            wp += 256;
            if ((b & 31) == 28) wp -= 2048;
        }
        t1 = __builtin_amdgcn_s_memtime();
    }
    out[threadIdx.x] = z;
    if (lane == 0) clk[wave] = t1 - t0;
}

int main()
{
    float *d_out; unsigned long long *d_clk;
    hipMalloc(&d_out, 4096 * 4); hipMalloc(&d_clk, 64 * 8); hipMemset(d_out, 0, 4096 * 4);
    unsigned long long c[8];
    const int N = 960;
    auto run = [&](const char *name, auto kern, int waves) {
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 80000);
        hipLaunchKernelGGL(kern, dim3(1), dim3(512), 80000, 0, d_out, d_clk, N, waves);
        hipDeviceSynchronize();
        hipMemcpy(c, d_clk, sizeof(c), hipMemcpyDeviceToHost);
        printf("%-36s active=%d clk/block:", name, waves);
        for (int w = 0; w < waves; ++w) printf(" %6.1f", (double)c[w] / N);
        printf("\n");
    };
    for (int waves : {1, 4, 8}) {
        run("loads + dpp chain", k<0>, waves);
        run("dpp chain only", k<1>, waves);
        run("loads only", k<2>, waves);
        run("plain mul chain only", k<3>, waves);
        run("adds only", k<4>, waves);
        run("loads + plain mul chain", k<5>, waves);
    }
    return 0;
}
