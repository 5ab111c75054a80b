// v_mfma_f32_4x4x1_16B_f32 as a quad outer-product multiplier: layout, bit-equality of (a*b + 0) with v_mul_f32, and the cost of
// "4 MFMA + 16 adds" against "16 DPP multiplies + 16 adds" per GRU-A item (4 streams x 4 columns).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int SEL> __device__ __forceinline__ float qb(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), SEL * 0x55, 0xf, 0xf, true));
}
__global__ void layout(const float *a, const float *b, float *d)
{
    const int l = threadIdx.x;
    f4 c = {0.f, 0.f, 0.f, 0.f};
    f4 r = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
    for (int k = 0; k < 4; ++k) d[l * 4 + k] = r[k];
}
__global__ void exact(const float *a, const float *b, float *viamfma, float *viamul, int n)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    f4 c = {0.f, 0.f, 0.f, 0.f};
    f4 r = __builtin_amdgcn_mfma_f32_4x4x1f32(a[i], b[i], c, 0, 0, 0);
    // lane (quad base + i) register j = a_i * b_j : compare the diagonal and one off-diagonal with plain multiplies
    float prod[4];
    // (layout, see main: register k of lane (quad base + j) = a[quad base + k] * b[quad base + j])
    prod[0] = qb<0>(a[i]) * b[i]; prod[1] = qb<1>(a[i]) * b[i]; prod[2] = qb<2>(a[i]) * b[i]; prod[3] = qb<3>(a[i]) * b[i];
    for (int k = 0; k < 4; ++k) { viamfma[i * 4 + k] = r[k]; viamul[i * 4 + k] = prod[k]; }
}
template <int MODE>   // 0: 16 dpp mul + 16 add, 1: 4 mfma (c = 0) + 16 add, 2: 4 mfma accumulating (FAST)
__global__ __launch_bounds__(512) void cost(float *out, unsigned long long *clk, int n)
{
    float w[32][4];
    for (int j = 0; j < 32; ++j) for (int c = 0; c < 4; ++c) w[j][c] = out[(threadIdx.x + j * 7 + c) & 1023];
    float h[4] = {out[threadIdx.x] + 1.f, 0.5f, 0.25f, 0.125f};
    float acc[4] = {0, 0, 0, 0};
    f4 accv = {0, 0, 0, 0};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int j = 0; j < 32; ++j) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (MODE == 0) {
                    acc[0] += w[j][c] * qb<0>(h[c]); acc[1] += w[j][c] * qb<1>(h[c]); acc[2] += w[j][c] * qb<2>(h[c]); acc[3] += w[j][c] * qb<3>(h[c]);
                } else if (MODE == 1) {
                    f4 z = {0.f, 0.f, 0.f, 0.f};
                    f4 p = __builtin_amdgcn_mfma_f32_4x4x1f32(w[j][c], h[c], z, 0, 0, 0);
                    acc[0] += p[0]; acc[1] += p[1]; acc[2] += p[2]; acc[3] += p[3];
                } else {
                    accv = __builtin_amdgcn_mfma_f32_4x4x1f32(w[j][c], h[c], accv, 0, 0, 0);
                }
            }
            h[0] += 1e-7f;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3] + accv[0] + accv[1] + accv[2] + accv[3];
    if ((threadIdx.x & 63) == 0) clk[threadIdx.x >> 6] = t1 - t0;
}
int main()
{
    float ha[64], hb[64], hd[256], *a, *b, *d;
    for (int i = 0; i < 64; ++i) { ha[i] = 1.f + i; hb[i] = 1.f + (i + 1) * 0.015625f + i * 3.0517578125e-5f; }   // products a[x]*b[y] all distinct and exact
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, a, b, d);
    hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
    for (int l = 0; l < 8; ++l) {
        printf("lane %d:", l);
        for (int k = 0; k < 4; ++k) { float v = hd[l * 4 + k]; int ai = -1, bi = -1; for (int x = 0; x < 64; ++x) for (int y = 0; y < 64; ++y) if (ha[x] * hb[y] == v) { ai = x; bi = y; } printf("  d[%d] = a[%d]*b[%d]", k, ai, bi); }
        printf("\n");
    }
    // exactness on random bit patterns (finite, incl. tiny and huge)
    const int N = 1 << 22;
    float *xa = (float *)malloc(N * 4), *xb = (float *)malloc(N * 4), *m1 = (float *)malloc(N * 16), *m2 = (float *)malloc(N * 16);
    srand(1);
    for (int i = 0; i < N; ++i) {
        unsigned ua = ((unsigned)rand() << 16) ^ rand(), ub = ((unsigned)rand() << 16) ^ rand();
        if (i & 1) { ua = (ua & 0x807FFFFFu) | ((100u + (rand() % 60)) << 23); ub = (ub & 0x807FFFFFu) | ((100u + (rand() % 60)) << 23); }   // moderate exponents
        memcpy(&xa[i], &ua, 4); memcpy(&xb[i], &ub, 4);
        if (!isfinite(xa[i])) xa[i] = 1.5f; if (!isfinite(xb[i])) xb[i] = -2.5f;
    }
    float *da, *db, *d1, *d2;
    hipMalloc(&da, N * 4); hipMalloc(&db, N * 4); hipMalloc(&d1, N * 16); hipMalloc(&d2, N * 16);
    hipMemcpy(da, xa, N * 4, hipMemcpyHostToDevice); hipMemcpy(db, xb, N * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(exact, dim3(N / 64), dim3(64), 0, 0, da, db, d1, d2, N);
    hipMemcpy(m1, d1, N * 16, hipMemcpyDeviceToHost); hipMemcpy(m2, d2, N * 16, hipMemcpyDeviceToHost);
    long bad = 0, badnz = 0, sub = 0;
    for (long i = 0; i < (long)N * 4; ++i) if (memcmp(&m1[i], &m2[i], 4)) { ++bad; if (m1[i] != m2[i] || (m1[i] == 0 && m2[i] != 0)) ++badnz; if (fabsf(m2[i]) < 1.2e-38f) ++sub; if (bad < 6) printf("  mismatch: mfma %a mul %a\n", m1[i], m2[i]); }
    printf("exactness: %ld of %ld products differ bitwise (%ld differ in value, %ld of them where the multiply result is subnormal/zero)\n", bad, (long)N * 4, badnz, sub);
    // cost
    float *o; unsigned long long *c, hc[8];
    hipMalloc(&o, 4096); hipMalloc(&c, 64); hipMemset(o, 0, 4096);
    const int n = 200;
    for (int waves : {1, 8}) {
        hipLaunchKernelGGL(cost<0>, dim3(1), dim3(64 * waves), 0, 0, o, c, n); hipDeviceSynchronize(); hipMemcpy(hc, c, 64, hipMemcpyDeviceToHost);
        printf("waves=%d  16 dpp-mul + 16 add : %.1f clk/item (wave0) %.1f (wave %d)\n", waves, hc[0] / (n * 32.0), hc[waves - 1] / (n * 32.0), waves - 1);
        hipLaunchKernelGGL(cost<1>, dim3(1), dim3(64 * waves), 0, 0, o, c, n); hipDeviceSynchronize(); hipMemcpy(hc, c, 64, hipMemcpyDeviceToHost);
        printf("waves=%d  4 mfma(c=0) + 16 add: %.1f clk/item (wave0) %.1f (wave %d)\n", waves, hc[0] / (n * 32.0), hc[waves - 1] / (n * 32.0), waves - 1);
        hipLaunchKernelGGL(cost<2>, dim3(1), dim3(64 * waves), 0, 0, o, c, n); hipDeviceSynchronize(); hipMemcpy(hc, c, 64, hipMemcpyDeviceToHost);
        printf("waves=%d  4 mfma accumulating  : %.1f clk/item (wave0) %.1f (wave %d)\n", waves, hc[0] / (n * 32.0), hc[waves - 1] / (n * 32.0), waves - 1);
    }
    return 0;
}
