// which SIMD does wave w of a 512-thread workgroup land on -- and where do the waves of a SECOND workgroup on the same CU go?
//   hwid [lds_bytes [blocks]]     (HW_ID: wave_id[3:0], simd_id[5:4], cu_id[11:8], sh_id[12], se_id[15:13] on gfx9; XCC_ID[3:0])
// 150000 B of LDS -> one workgroup per CU (the float kernels); 66000 B -> two (the int8 PACK2 kernels).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
__global__ __launch_bounds__(512, 2) void k(unsigned *out, int spin)
{
    extern __shared__ unsigned char smem[];
    unsigned v, x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    // stay resident long enough for every CU to receive its full share of workgroups
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) smem[threadIdx.x] = (unsigned char)v;
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2] = v;
        out[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2 + 1] = x;
    }
}
int main(int argc, char **argv)
{
    const int lds = argc > 1 ? atoi(argv[1]) : 150000, nb = argc > 2 ? atoi(argv[2]) : 256;
    unsigned *d;
    std::vector<unsigned> h((size_t)nb * 16);
    hipMalloc(&d, h.size() * 4);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(k, dim3(nb), dim3(512), lds, 0, d, 2000000);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    struct B { unsigned key; int blk; unsigned simd[8]; };
    std::vector<B> bs;
    int pattern_ok = 0;
    for (int b = 0; b < nb; ++b) {
        B e;
        const unsigned v0 = h[(size_t)b * 16], x0 = h[(size_t)b * 16 + 1];
        e.key = ((x0 & 15) << 16) | (((v0 >> 13) & 7) << 12) | (((v0 >> 12) & 1) << 8) | ((v0 >> 8) & 15);
        e.blk = b;
        bool rr = true;
        for (int w = 0; w < 8; ++w) { e.simd[w] = (h[((size_t)b * 8 + w) * 2] >> 4) & 3; if (e.simd[w] != (e.simd[0] + w) % 4) rr = false; }
        pattern_ok += rr;
        bs.push_back(e);
    }
    std::sort(bs.begin(), bs.end(), [](const B &a, const B &b) { return a.key != b.key ? a.key < b.key : a.blk < b.blk; });
    int same_start = 0, pairs = 0, cus = 0;
    for (size_t i = 0; i < bs.size(); ++i) {
        if (i == 0 || bs[i].key != bs[i - 1].key) ++cus;
        else { ++pairs; same_start += bs[i].simd[0] == bs[i - 1].simd[0]; }
    }
    printf("lds %d blocks %d: %d distinct CUs; %d of %d blocks place wave w on SIMD (s0 + w) %% 4; co-resident pairs %d, of which wave 0 on the SAME simd: %d\n",
           lds, nb, cus, pattern_ok, nb, pairs, same_start);
    for (size_t i = 0; i < bs.size() && i < 12; ++i) {
        printf("  xcc %u se %u sh %u cu %2u  block %3d: simd of waves 0..7 =", (bs[i].key >> 16) & 15, (bs[i].key >> 12) & 7, (bs[i].key >> 8) & 1, bs[i].key & 15, bs[i].blk);
        for (int w = 0; w < 8; ++w) printf(" %u", bs[i].simd[w]);
        printf("\n");
    }
    return 0;
}
