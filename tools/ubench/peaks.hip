// Measured ceilings of this box for the roofline denominators: LDS read bandwidth of the whole chip
// (ds_read_b128, conflict-free, all CUs busy), HBM copy bandwidth (read + write streams), fp32 VALU rate without FMA
// (separately rounded multiply + add, the arithmetic the PARITY engine is allowed to use).  Prints one JSON object.
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ __launch_bounds__(512) void lds_read(float *out, int iters)
{
    __shared__ float4 buf[4096];                                   // 64 KB
    for (int i = threadIdx.x; i < 4096; i += 512) buf[i] = make_float4(i, 1, 2, 3);
    __syncthreads();
    float4 a = make_float4(0, 0, 0, 0);
    int idx = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float4 v = buf[(idx + k * 512) & 4095];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        idx = (idx + 1) & 4095;
    }
    out[blockIdx.x * 512 + threadIdx.x] = a.x + a.y + a.z + a.w;
}

__global__ __launch_bounds__(256) void copy4(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

__global__ __launch_bounds__(512) void valu_mul_add(float *out, int iters)
{
    float a[16], h = out[threadIdx.x] + 1.0001f;
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float t;
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t) : "v"(a[i]), "v"(h));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(a[i]) : "v"(t), "v"(h));
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

// the same arithmetic as packed math: v_pk_mul_f32 + v_pk_add_f32, every half rounded on its own -- PARITY may use it
// (the no-FMA ceiling has to include it)
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(512) void valu_pk_mul_add(float *out, int iters)
{
    f2 a[8], h = {out[threadIdx.x] + 1.0001f, out[threadIdx.x] + 0.9999f};
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (f2){(float)i, (float)i + 0.5f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            f2 t;
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(a[i]), "v"(h));
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(a[i]) : "v"(t), "v"(h));
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i][0] + a[i][1];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    float *out;
    hipMalloc(&out, (size_t)cus * 8 * 512 * 4);
    hipMemset(out, 0, (size_t)cus * 8 * 512 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    // LDS: 2 workgroups per CU, 16 b128 reads per thread and iteration
    const int it_l = 4000;
    hipLaunchKernelGGL(lds_read, dim3(cus * 2), dim3(512), 0, 0, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(lds_read, dim3(cus * 2), dim3(512), 0, 0, out, it_l);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    const double lds_tbs = (double)cus * 2 * 512 * 16.0 * 16 * it_l / (ms * 1e-3) / 1e12;
    // HBM copy: 4 GiB read + 4 GiB written
    const size_t n4 = (size_t)1 << 28;                               // float4 elements = 4 GiB
    float4 *src, *dst;
    hipMalloc(&src, n4 * 16); hipMalloc(&dst, n4 * 16);
    hipMemset(src, 1, n4 * 16);
    hipLaunchKernelGGL(copy4, dim3(cus * 16), dim3(256), 0, 0, src, dst, n4);
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(copy4, dim3(cus * 16), dim3(256), 0, 0, src, dst, n4);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    const double hbm_tbs = 3.0 * 2.0 * n4 * 16 / (ms * 1e-3) / 1e12;
    // VALU: separately rounded multiply + add
    const int it_v = 20000;
    hipLaunchKernelGGL(valu_mul_add, dim3(cus * 4), dim3(512), 0, 0, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(valu_mul_add, dim3(cus * 4), dim3(512), 0, 0, out, it_v);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    const double valu_tflops = (double)cus * 4 * 512 * 32.0 * it_v / (ms * 1e-3) / 1e12;
    hipLaunchKernelGGL(valu_pk_mul_add, dim3(cus * 4), dim3(512), 0, 0, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(valu_pk_mul_add, dim3(cus * 4), dim3(512), 0, 0, out, it_v);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    const double pk_tflops = (double)cus * 4 * 512 * 32.0 * it_v / (ms * 1e-3) / 1e12;
    printf("{\"device\": \"%s\", \"compute_units\": %d, \"clock_mhz\": %d, \"lds_read_TBs\": %.1f, \"hbm_copy_TBs\": %.2f, "
           "\"fp32_mul_add_no_fma_TFLOPs\": %.1f, \"fp32_packed_mul_add_no_fma_TFLOPs\": %.1f, \"note\": \"measured by tools/ubench/peaks.hip; the guide's peaks (150 TB/s LDS, "
           "8 TB/s HBM, 157.3 TFLOP/s fp32 with packed FMA) are the denominators bench.py uses\"}\n",
           p.name, cus, p.clockRate / 1000, lds_tbs, hbm_tbs, valu_tflops, pk_tflops);
    return 0;
}
