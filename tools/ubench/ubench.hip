// Micro-benchmarks that calibrate the cost model of the sample kernel on gfx950 (shader clocks via s_memtime).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#pragma clang diagnostic ignored "-Wunused-value"
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_dep_chain(float *out, unsigned long long *clk, int n, float a, float b)
{
    float acc = out[threadIdx.x];
    float w0 = a, w1 = b;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) { float t = w0 * w1; asm volatile("" : "+v"(t)); acc = acc + t; w0 = w0 + 1.0f; }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = acc + w0;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

// pure dependent add chain
__global__ void k_add_chain(float *out, unsigned long long *clk, int n, float a)
{
    float acc = out[threadIdx.x];
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) { acc = acc + a; asm volatile("" : "+v"(acc)); }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

// 4 independent add chains
__global__ void k_add_indep(float *out, unsigned long long *clk, int n, float a)
{
    float a0 = out[threadIdx.x], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { a0 = a0 + a; a1 = a1 + a; a2 = a2 + a; a3 = a3 + a; asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)); }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = a0 + a1 + a2 + a3;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

// LDS read -> use -> address dependent chain (pointer chase), b128
__global__ void k_lds_chase(float *out, unsigned long long *clk, int n)
{
    __shared__ float4 buf[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) buf[i] = make_float4(__int_as_float((i * 17 + 5) & 1023), 0, 0, 0);
    __syncthreads();
    int idx = threadIdx.x & 63;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) { float4 v = buf[idx]; idx = __float_as_int(v.x); }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = (float)idx;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

// independent LDS b128 reads, 8 in flight
__global__ void k_lds_stream(float *out, unsigned long long *clk, int n)
{
    __shared__ float4 buf[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) buf[i] = make_float4(i, 1, 2, 3);
    __syncthreads();
    float acc = 0;
    const int base = threadIdx.x & 63;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = buf[(base + k * 64 + i) & 2047];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += v[k].x + v[k].w;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

__global__ void k_barrier(float *out, unsigned long long *clk, int n)
{
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) __syncthreads();
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
    out[threadIdx.x] = 0;
}

// mul with DPP quad broadcast + add, 4 independent chains (the GRU-A inner loop shape)
__global__ void k_dpp_mac(float *out, unsigned long long *clk, int n, float a)
{
    float acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0, h = out[threadIdx.x], w = a;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float h0 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, h), 0x00, 0xf, 0xf, true));
            float h1 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, h), 0x55, 0xf, 0xf, true));
            float h2 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, h), 0xAA, 0xf, 0xf, true));
            float h3 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, h), 0xFF, 0xf, 0xf, true));
            acc0 = acc0 + w * h0; acc1 = acc1 + w * h1; acc2 = acc2 + w * h2; acc3 = acc3 + w * h3;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = acc0 + acc1 + acc2 + acc3;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

// global (L2-resident) dependent load chain
__global__ void k_gl_chase(const int *tab, float *out, unsigned long long *clk, int n)
{
    int idx = threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) idx = tab[idx];
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = (float)idx;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

// LDS b128 / b32 independent reads with only `active` lanes enabled (cost vs active lanes)
template <int WIDTH>
__global__ void k_lds_active(float *out, unsigned long long *clk, int n, int active)
{
    __shared__ float4 buf[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) buf[i] = make_float4(i, 1, 2, 3);
    __syncthreads();
    float acc = 0;
    const int lane = threadIdx.x & 63;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (lane < active) {
        for (int i = 0; i < n; ++i) {
            if (WIDTH == 16) {
                float4 v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = buf[(lane + k * 64 + i) & 2047];
#pragma unroll
                for (int k = 0; k < 8; ++k) acc += v[k].x + v[k].w;
            } else {
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = ((const float *)buf)[(lane + k * 64 + i) & 8191];
#pragma unroll
                for (int k = 0; k < 8; ++k) acc += v[k];
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
// all lanes read the same address (broadcast)
__global__ void k_lds_bcast(float *out, unsigned long long *clk, int n)
{
    __shared__ float4 buf[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) buf[i] = make_float4(i, 1, 2, 3);
    __syncthreads();
    float acc = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = buf[(k * 64 + i) & 2047];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += v[k].x + v[k].w;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
// integer / f64 latency chains
__global__ void k_imul_chain(float *out, unsigned long long *clk, int n, unsigned a)
{
    unsigned x = threadIdx.x + 1;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) { x = x * a + 12345u; }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = (float)x;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
__global__ void k_f64_chain(float *out, unsigned long long *clk, int n, double a)
{
    double x = out[threadIdx.x];
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) { x = x + a; asm volatile("" : "+v"(x)); }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = (float)x;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

int main()
{
    float *d_out; unsigned long long *d_clk; int *d_tab;
    CHK(hipMalloc(&d_out, 4096 * 4)); CHK(hipMalloc(&d_clk, 64 * 8)); CHK(hipMemset(d_out, 0, 4096 * 4));
    std::vector<int> tab(1 << 20);
    for (size_t i = 0; i < tab.size(); ++i) tab[i] = (int)((i * 1021 + 77) & (tab.size() - 1));
    CHK(hipMalloc(&d_tab, tab.size() * 4)); CHK(hipMemcpy(d_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
    unsigned long long c;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    auto report = [&](const char *name, double per, const char *unit) {
        CHK(hipMemcpy(&c, d_clk, 8, hipMemcpyDeviceToHost));
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %8.2f clk/%s   (kernel %.3f ms)\n", name, (double)c / per, unit, ms);
        return 0;
    };
    const int N = 4096;
    for (int threads : {64, 512}) {
        printf("--- %d threads/block, 1 block\n", threads);
        hipEventRecord(e0); hipLaunchKernelGGL(k_dep_chain, dim3(1), dim3(threads), 0, 0, d_out, d_clk, N, 1.0001f, 0.5f); hipEventRecord(e1); hipDeviceSynchronize();
        report("dependent mul->add MAC (mul feeds add)", N * 16.0, "MAC");
        hipEventRecord(e0); hipLaunchKernelGGL(k_add_chain, dim3(1), dim3(threads), 0, 0, d_out, d_clk, N, 1.0001f); hipEventRecord(e1); hipDeviceSynchronize();
        report("dependent v_add chain", N * 16.0, "add");
        hipEventRecord(e0); hipLaunchKernelGGL(k_add_indep, dim3(1), dim3(threads), 0, 0, d_out, d_clk, N, 1.0001f); hipEventRecord(e1); hipDeviceSynchronize();
        report("4 independent v_add chains", N * 16.0, "add");
        hipEventRecord(e0); hipLaunchKernelGGL(k_dpp_mac, dim3(1), dim3(threads), 0, 0, d_out, d_clk, N, 1.0001f); hipEventRecord(e1); hipDeviceSynchronize();
        report("dpp-mul + add, 4 chains (per MAC)", N * 16.0, "MAC");
        hipEventRecord(e0); hipLaunchKernelGGL(k_lds_chase, dim3(1), dim3(threads), 0, 0, d_out, d_clk, N); hipEventRecord(e1); hipDeviceSynchronize();
        report("LDS b128 dependent read latency", (double)N, "read");
        hipEventRecord(e0); hipLaunchKernelGGL(k_lds_stream, dim3(1), dim3(threads), 0, 0, d_out, d_clk, N); hipEventRecord(e1); hipDeviceSynchronize();
        report("LDS b128 x8 independent (per read)", N * 8.0, "read");
        hipEventRecord(e0); hipLaunchKernelGGL(k_barrier, dim3(1), dim3(threads), 0, 0, d_out, d_clk, N); hipEventRecord(e1); hipDeviceSynchronize();
        report("__syncthreads", (double)N, "barrier");
        hipEventRecord(e0); hipLaunchKernelGGL(k_gl_chase, dim3(1), dim3(threads), 0, 0, d_tab, d_out, d_clk, 2048); hipEventRecord(e1); hipDeviceSynchronize();
        report("global dependent load (4 MB table, L2)", 2048.0, "load");
    }

    printf("--- LDS cost vs active lanes (64 threads, 8 independent reads per wait)\n");
    for (int act : {64, 48, 32, 24, 16, 8}) {
        hipEventRecord(e0); hipLaunchKernelGGL(k_lds_active<16>, dim3(1), dim3(64), 0, 0, d_out, d_clk, N, act); hipEventRecord(e1); hipDeviceSynchronize();
        char nm[64]; snprintf(nm, sizeof nm, "b128 read, %d active lanes", act); report(nm, N * 8.0, "read");
        hipEventRecord(e0); hipLaunchKernelGGL(k_lds_active<4>, dim3(1), dim3(64), 0, 0, d_out, d_clk, N, act); hipEventRecord(e1); hipDeviceSynchronize();
        snprintf(nm, sizeof nm, "b32  read, %d active lanes", act); report(nm, N * 8.0, "read");
    }
    hipEventRecord(e0); hipLaunchKernelGGL(k_lds_bcast, dim3(1), dim3(64), 0, 0, d_out, d_clk, N); hipEventRecord(e1); hipDeviceSynchronize();
    report("b128 broadcast read (all lanes same addr)", N * 8.0, "read");
    hipEventRecord(e0); hipLaunchKernelGGL(k_imul_chain, dim3(1), dim3(64), 0, 0, d_out, d_clk, N, 69069u); hipEventRecord(e1); hipDeviceSynchronize();
    report("dependent u32 mul-add chain", N * 16.0, "op");
    hipEventRecord(e0); hipLaunchKernelGGL(k_f64_chain, dim3(1), dim3(64), 0, 0, d_out, d_clk, N, 1.5); hipEventRecord(e1); hipDeviceSynchronize();
    report("dependent f64 add chain", N * 16.0, "add");
    // clock calibration: s_memtime ticks per microsecond
    hipEventRecord(e0); hipLaunchKernelGGL(k_add_chain, dim3(1), dim3(64), 0, 0, d_out, d_clk, 1 << 16, 1.0f); hipEventRecord(e1); hipDeviceSynchronize();
    CHK(hipMemcpy(&c, d_clk, 8, hipMemcpyDeviceToHost));
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("s_memtime: %.1f ticks/us (kernel %.3f ms, %llu ticks)\n", (double)c / (ms * 1e3), ms, c);
    return 0;
}
