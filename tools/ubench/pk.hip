// packed-f32 VALU and LDS broadcast-read rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#pragma clang diagnostic ignored "-Wunused-value"
typedef float float2v __attribute__((ext_vector_type(2)));

// 8 independent packed chains: pk_mul (broadcast weight via op_sel) + pk_add
__global__ __launch_bounds__(512) void k_pk(float *out, unsigned long long *clk, int n)
{
    float2v acc[8], h[8];
    float2v w = {1.0001f, 0.9999f};
#pragma unroll
    for (int i = 0; i < 8; ++i) { acc[i] = (float2v){0.f, 0.f}; h[i] = (float2v){out[threadIdx.x] + i, 1.f}; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float2v t;
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(h[i]), "v"(w));
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(acc[i]) : "v"(acc[i]), "v"(t));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
    out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) clk[threadIdx.x >> 6] = t1 - t0;
}
// same with scalar ops (16 chains) for reference
__global__ __launch_bounds__(512) void k_sc(float *out, unsigned long long *clk, int n)
{
    float acc[16], h[16];
    float w = 1.0001f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[i] = 0.f; h[i] = out[threadIdx.x] + i; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float t;
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t) : "v"(h[i]), "v"(w));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(acc[i]) : "v"(acc[i]), "v"(t));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) clk[threadIdx.x >> 6] = t1 - t0;
}
// dpp-mul chains for reference
__global__ __launch_bounds__(512) void k_dpp(float *out, unsigned long long *clk, int n)
{
    float acc[16], h[16];
    float w = 1.0001f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[i] = 0.f; h[i] = out[threadIdx.x] + i; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float t;
            asm volatile("v_mul_f32_dpp %0, %1, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(t) : "v"(h[i]), "v"(w));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(acc[i]) : "v"(acc[i]), "v"(t));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) clk[threadIdx.x >> 6] = t1 - t0;
}
// LDS reads: 16 reads issued back-to-back then one wait; groups of 8 lanes share an address
template <int WIDTH>
__global__ __launch_bounds__(512) void k_lds(float *out, unsigned long long *clk, int n)
{
    __shared__ float4 buf[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) buf[i] = make_float4(i, 1, 2, 3);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    int base = (lane >> 3) * 37;
    float4 a = make_float4(0, 0, 0, 0);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n; ++it) {
        float4 v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (WIDTH == 16) v[k] = buf[(base + k * 5 + it) & 4095];
            else { v[k].x = ((const float *)buf)[(base + k * 5 + it) & 16383]; v[k].y = v[k].z = v[k].w = 0; }
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) asm volatile("" :: "v"(v[k].x), "v"(v[k].w));
        a.x += v[0].x;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = a.x;
    if ((threadIdx.x & 63) == 0) clk[threadIdx.x >> 6] = t1 - t0;
}
int main()
{
    float *d_out; unsigned long long *d_clk;
    hipMalloc(&d_out, 4096 * 4); hipMalloc(&d_clk, 64 * 8); hipMemset(d_out, 0, 4096 * 4);
    unsigned long long c[8];
    const int N = 2000;
    auto run = [&](const char *name, auto kern, int threads, double per) {
        hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, d_out, d_clk, N);
        hipDeviceSynchronize();
        hipMemcpy(c, d_clk, sizeof(c), hipMemcpyDeviceToHost);
        printf("%-44s thr=%3d:", name, threads);
        for (int w = 0; w < threads / 64; ++w) printf(" %6.2f", (double)c[w] / (N * per));
        printf("\n");
    };
    for (int thr : {64, 256, 512}) {
        run("pk_mul(bcast)+pk_add, clk per pk-instr", k_pk, thr, 16.0);
        run("mul+add scalar,       clk per instr", k_sc, thr, 32.0);
        run("mul_dpp+add,          clk per instr", k_dpp, thr, 32.0);
        run("ds_read_b128 x16 per wait, clk per read", k_lds<16>, thr, 16.0);
        run("ds_read_b32  x16 per wait, clk per read", k_lds<4>, thr, 16.0);
    }
    return 0;
}
