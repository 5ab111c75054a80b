// GRU-A item forms for four streams, PARITY arithmetic (every product and every sum rounded separately), clk per item and wave at
// 1 / 2 / 4 waves per SIMD.  One item = (this lane's row) x (one 4-wide input block) x 4 streams = 16 products + 16 sequential sums.
//   MODE 0  16 v_mul_f32_dpp (quad broadcast of the state) + 16 v_add_f32          -- what round 3 ships
//   MODE 1  4 v_mfma_f32_4x4x1 with C = -0.0 (exact products of 4 streams x 4 rows per quad) + 16 v_add_f32
//   MODE 2  4 v_mfma_f32_4x4x1 (C = -0.0) + 8 v_pk_add_f32 (stream pairs)
//   MODE 3  every lane reads all four streams' values (4 ds_read_b128, [block][column][stream]) + 8 v_pk_mul_f32 (weight
//           broadcast through op_sel) + 8 v_pk_add_f32: no cross-lane operand at all
//   MODE 4  as MODE 2 with the weights ALSO read from LDS (2 ds_read_b128 per item): the GRU-B block loop as "items"
// The state block address is per row group (8 lanes) as in the kernel; weights stay in VGPRs (NW items per lane).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
template <int SEL> __device__ __forceinline__ float qb(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), SEL * 0x55, 0xf, 0xf, true));
}
constexpr int NW = 24;
template <int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void k(float *out, unsigned long long *clk, int n, const float *win)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4 *h = (float4 *)smem;                       // 96 blocks x 4 streams x float4 = 6 KB (+ pad)
    float4 *wl = (float4 *)(smem + 8192);             // MODE 4: [item][lane] weights, 24 KB
    for (int i = threadIdx.x; i < 512; i += THREADS) h[i] = make_float4(i * 0.001f, 0.5f, 0.25f, 0.125f);
    for (int i = threadIdx.x; i < NW * 64; i += THREADS) wl[i] = make_float4(1.f, 0.5f, 0.25f, 0.125f);
    f4 w[NW];
#pragma unroll
    for (int j = 0; j < NW; ++j) { const float4 t = ((const float4 *)win)[j * 64 + (threadIdx.x & 63)]; w[j] = (f4){t.x, t.y, t.z, t.w}; }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    f2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
    f4 nz = {-0.f, -0.f, -0.f, -0.f};
    asm volatile("" : "+v"(nz));
    // block of item j for this lane's row group (8 lanes share a block), [block][stream][4] with the quad lane picking the stream
    auto blk = [&](int j, int it) { return ((lane >> 3) * 11 + j * 5 + it) % 96; };
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n; ++it) {
        if constexpr (MODE == 3) {
            float4 hq[2][4];
#pragma unroll
            for (int c = 0; c < 4; ++c) hq[0][c] = h[blk(0, it) * 4 + c];
#pragma unroll
            for (int j = 0; j < NW; ++j) {
                if (j + 1 < NW) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) hq[(j + 1) & 1][c] = h[blk(j + 1, it) * 4 + c];
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float4 hv = hq[j & 1][c];
                    f2 h01 = {hv.x, hv.y}, h23 = {hv.z, hv.w}, ww = {w[j][c], w[j][c]};
                    const f2 p01 = h01 * ww, p23 = h23 * ww;          // v_pk_mul_f32
                    a01 = a01 + p01; a23 = a23 + p23;                 // v_pk_add_f32
                }
            }
        } else {
            float4 hq[3], wq[3];
            hq[0] = h[blk(0, it) * 4 + (lane & 3)]; hq[1] = h[blk(1, it) * 4 + (lane & 3)];
            if constexpr (MODE == 4) { wq[0] = wl[lane]; wq[1] = wl[64 + lane]; }
#pragma unroll
            for (int j = 0; j < NW; ++j) {
                if (j + 2 < NW) {
                    hq[(j + 2) % 3] = h[blk(j + 2, it) * 4 + (lane & 3)];
                    if constexpr (MODE == 4) wq[(j + 2) % 3] = wl[(j + 2) * 64 + lane];
                }
                const float4 hv = hq[j % 3];
                const float hk[4] = {hv.x, hv.y, hv.z, hv.w};
                float wk[4] = {w[j][0], w[j][1], w[j][2], w[j][3]};
                if constexpr (MODE == 4) { const float4 t = wq[j % 3]; wk[0] = t.x; wk[1] = t.y; wk[2] = t.z; wk[3] = t.w; }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if constexpr (MODE == 0) {
                        const float t0 = wk[c] * qb<0>(hk[c]), t1 = wk[c] * qb<1>(hk[c]), t2 = wk[c] * qb<2>(hk[c]), t3 = wk[c] * qb<3>(hk[c]);
                        a0 = a0 + t0; a1 = a1 + t1; a2 = a2 + t2; a3 = a3 + t3;
                    } else if constexpr (MODE == 1) {
                        const f4 p = __builtin_amdgcn_mfma_f32_4x4x1f32(hk[c], wk[c], nz, 0, 0, 0);
                        a0 = a0 + p[0]; a1 = a1 + p[1]; a2 = a2 + p[2]; a3 = a3 + p[3];
                    } else {
                        const f4 p = __builtin_amdgcn_mfma_f32_4x4x1f32(hk[c], wk[c], nz, 0, 0, 0);
                        a01 = a01 + __builtin_shufflevector(p, p, 0, 1);
                        a23 = a23 + __builtin_shufflevector(p, p, 2, 3);
                    }
                }
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = a0 + a1 + a2 + a3 + a01[0] + a01[1] + a23[0] + a23[1];
    if (lane == 0) clk[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}
// exactness of MODE 2 against MODE 0 on the same data (bitwise)
template <int MODE>
__global__ __launch_bounds__(64) void val(const float *hin, const float *win, float *res)
{
    const int lane = threadIdx.x;
    f4 nz = {-0.f, -0.f, -0.f, -0.f};
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    f2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
    for (int j = 0; j < 256; ++j) {
        const int b = ((lane >> 3) * 11 + j * 5) % 96;
        const float4 hv = ((const float4 *)hin)[b * 4 + (lane & 3)];
        const float4 wv = ((const float4 *)win)[j * 64 + lane];
        const float hk[4] = {hv.x, hv.y, hv.z, hv.w}, wk[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if constexpr (MODE == 0) {
                const float t0 = wk[c] * qb<0>(hk[c]), t1 = wk[c] * qb<1>(hk[c]), t2 = wk[c] * qb<2>(hk[c]), t3 = wk[c] * qb<3>(hk[c]);
                a[0] = a[0] + t0; a[1] = a[1] + t1; a[2] = a[2] + t2; a[3] = a[3] + t3;
            } else {
                const f4 p = __builtin_amdgcn_mfma_f32_4x4x1f32(hk[c], wk[c], nz, 0, 0, 0);
                a01 = a01 + __builtin_shufflevector(p, p, 0, 1);
                a23 = a23 + __builtin_shufflevector(p, p, 2, 3);
            }
        }
    }
    if constexpr (MODE != 0) { a[0] = a01[0]; a[1] = a01[1]; a[2] = a23[0]; a[3] = a23[1]; }
    for (int s = 0; s < 4; ++s) res[lane * 4 + s] = a[s];
}
int main()
{
    float *d_out, *d_w, *d_h, *d_r0, *d_r1; unsigned long long *d_clk;
    hipMalloc(&d_out, 4096 * 4); hipMalloc(&d_clk, 4096 * 8); hipMalloc(&d_w, 256 * 64 * 16); hipMalloc(&d_h, 96 * 64); hipMalloc(&d_r0, 1024); hipMalloc(&d_r1, 1024);
    {
        static float hw[256 * 64 * 4], hh[96 * 16];
        unsigned s = 12345u;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((int)(s >> 8) % 20001 - 10000) * 1e-4f; };
        for (auto &v : hw) v = rnd() * 0.37f;
        for (auto &v : hh) v = rnd();
        hipMemcpy(d_w, hw, sizeof(hw), hipMemcpyHostToDevice); hipMemcpy(d_h, hh, sizeof(hh), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(val<0>, dim3(1), dim3(64), 0, 0, d_h, d_w, d_r0);
        hipLaunchKernelGGL(val<2>, dim3(1), dim3(64), 0, 0, d_h, d_w, d_r1);
        static unsigned r0[256], r1[256];
        hipMemcpy(r0, d_r0, 1024, hipMemcpyDeviceToHost); hipMemcpy(r1, d_r1, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 256; ++i) bad += r0[i] != r1[i];
        printf("mfma + pk_add vs dpp-mul + add over 256 items x 64 rows x 4 streams: %d of 256 sums differ bitwise\n", bad);
    }
    const int N = 400;
    unsigned long long c[16];
    auto run = [&](const char *name, auto kern, int threads, int blocks) {
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 40960);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 40960, 0, d_out, d_clk, N, d_w);
        hipDeviceSynchronize();
        hipMemcpy(c, d_clk, sizeof(c), hipMemcpyDeviceToHost);
        printf("%-34s waves/SIMD=%d blocks=%3d  clk/item per wave:", name, threads / 256, blocks);
        double inv = 0;
        for (int w = 0; w < threads / 64; ++w) { printf(" %5.1f", (double)c[w] / (N * (double)NW)); inv += (N * (double)NW) / (double)c[w]; }
        printf("   | per SIMD: %.1f clk/item\n", 4.0 / inv);
    };
#define RUNALL(MODE, NAME) \
    run(NAME, k<MODE, 256>, 256, 1); run(NAME, k<MODE, 512>, 512, 1); run(NAME, k<MODE, 1024>, 1024, 1); run(NAME, k<MODE, 512>, 512, 256);
    RUNALL(0, "0: 16 dpp-mul + 16 add")
    RUNALL(1, "1: 4 mfma(-0) + 16 add")
    RUNALL(2, "2: 4 mfma(-0) + 8 pk_add")
    RUNALL(3, "3: 4 reads + 8 pk_mul + 8 pk_add")
    RUNALL(4, "4: mode 2, weights from LDS too")
    return 0;
}
