// GRU-B input mat-vec candidates with the STATE operand delivered through SGPRs (scalar loads from an L2-resident
// mirror of the state that the gate stage writes) instead of LDS + DPP quad broadcasts.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o grub2 grub2.hip && ./grub2
// Every mode computes, for 48 rows (lane = row) and its streams, zrh = start + sum over 96 blocks x 4 columns of
// w[row][blk][c] * h[stream][blk][c] in that order, product and sum rounded separately (the PARITY contract), and is
// checked bit for bit against the host.  One 512-thread workgroup; `active` waves run the loop, each step preceded by
// the whole workgroup re-writing the state mirror with vector stores (s_waitcnt vmcnt(0) + barrier), like the gate stage
// would, and (cold = 1) an s_dcache_inv in every consumer wave.
//   MODE 0  reference: what the kernel does today -- weights ds_read_b128, state quad from LDS + DPP multiplies (1 stream / wave)
//   MODE 1  S1 : 1 stream / wave, weights ds_read_b128, state s_load_dwordx16 per 4 blocks, v_mul (SGPR operand) + v_add
//   MODE 2  S1p: as 1, products as v_pk_mul_f32 (2 per block)
//   MODE 3  S2 : 2 streams / wave in packed lanes: s_load_dwordx16 per 2 blocks, 4 v_pk_mul_f32 + 4 v_pk_add_f32 per block
//   MODE 4  S4 : 4 streams / wave: s_load_dwordx16 per block, 8 + 8 packed ops per block
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float i16v __attribute__((ext_vector_type(16)));   // (a float tuple: element access through __builtin_bit_cast of an int vector element miscompiles)
typedef int i8v __attribute__((ext_vector_type(8)));

#define NBLK 96
#define NROW 48
#define GLOBAL_AS __attribute__((address_space(1)))

template <int SEL> __device__ __forceinline__ float qb(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), SEL * 0x55, 0xf, 0xf, true));
}
__device__ __forceinline__ i16v sload16(const float *base, int byte_off)
{
    i16v v;
    asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(v) : "s"(base), "s"(byte_off) : "memory");
    return v;
}
// the wait "modifies" the loaded tuples, so every consumer is ordered behind it
__device__ __forceinline__ void wait0(i16v &a) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a) :: "memory"); }
__device__ __forceinline__ void wait0(i16v &a, i16v &b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b) :: "memory"); }
__device__ __forceinline__ void wait0(i16v &a, i16v &b, i16v &c, i16v &d) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b), "+s"(c), "+s"(d) :: "memory"); }

// (h.lo * w.lo, h.hi * w.lo) and (h.lo * w.hi, h.hi * w.hi): the SGPR pair holds one column of two streams
__device__ __forceinline__ f2 pkmul_lo(f2 h, f2 w) { f2 p; asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(p) : "s"(h), "v"(w)); return p; }
__device__ __forceinline__ f2 pkmul_hi(f2 h, f2 w) { f2 p; asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(p) : "s"(h), "v"(w)); return p; }
__device__ __forceinline__ f2 pkmul_el(f2 h, f2 w) { f2 p; asm("v_pk_mul_f32 %0, %1, %2" : "=v"(p) : "s"(h), "v"(w)); return p; }
__device__ __forceinline__ f2 pkadd(f2 a, f2 b) { f2 r; asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
#define HP(v, k) ((f2){(v)[2 * (k)], (v)[2 * (k) + 1]})
#define HS(v, k) ((v)[k])

// host and device agree on the synthetic operands
__host__ __device__ inline float wval(int row, int blk, int c) { return (float)(((row * 131 + blk * 17 + c * 7) * 73) % 251 - 125) / 128.f; }
__host__ __device__ inline float hval(int s, int n, int t) { return (float)(((s * 389 + n * 37 + t * 101) % 257) - 128) / 256.f; }

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float *hmir, float *out, unsigned long long *clk, int steps, int active, int cold, int partner)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f4 *lw = (f4 *)smem;                                     // [blk][row 48] float4
    f4 *lh = (f4 *)(smem + NBLK * NROW * 16);                // MODE 0: [blk][stream 4] float4 state blocks
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < NBLK * NROW; i += 512) {
        const int blk = i / NROW, row = i % NROW;
        lw[i] = (f4){wval(row, blk, 0), wval(row, blk, 1), wval(row, blk, 2), wval(row, blk, 3)};
    }
    __syncthreads();
    const int row = lane < NROW ? lane : NROW - 1;
    constexpr int SPW = MODE == 3 ? 2 : (MODE == 4 ? 4 : 1);   // streams per wave
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    unsigned long long tsum = 0, tfirst = 0;
    auto GLOBAL_AS *hm = (GLOBAL_AS float *)(uintptr_t)hmir;
    float dummy = 0.f;
    for (int t = 0; t < steps; ++t) {
        // ---- "gate stage": everybody writes the new state: LDS copy (MODE 0) and the global mirror in the mode's layout
        for (int i = tid; i < 4 * 384; i += 512) {
            const int s = i & 3, n = i >> 2, blk = n >> 2, c = n & 3;
            const float v = hval(s, n, t);
            ((float *)lh)[(blk * 4 + s) * 4 + c] = v;
            int o;
            if (MODE == 1 || MODE == 2) o = (s * NBLK + blk) * 4 + c;                       // [stream][blk][c]
            else if (MODE == 3) o = (((s >> 1) * NBLK + blk) * 4 + c) * 2 + (s & 1);       // [pair][blk][c][2]
            else o = (blk * 4 + c) * 4 + s;                                                // [blk][c][4]
            hm[o] = v;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        if (wave < active) {
            if (cold) { asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" ::: "memory"); }
            const f4 *wp = lw + row;
            if constexpr (MODE == 0) {
                const int s = wave & 3;
                const f4 *hp = lh + s + 4 * (lane & 3);                           // lane k of a quad fetches block 4q+k
                float z = 0.f;
                f4 w0 = wp[0], w1 = wp[NROW], w2 = wp[2 * NROW], w3;
                f4 hq = hp[0], hn;
                float p0 = w0.x * qb<0>(hq.x), p1 = w0.y * qb<0>(hq.y), p2 = w0.z * qb<0>(hq.z), p3 = w0.w * qb<0>(hq.w);
#define SB __builtin_amdgcn_sched_barrier(0)
#define STEP(WL, OFF, WN, HQ, HK, EXTRA)                                                            \
                {                                                                                   \
                    WL = wp[(OFF) * NROW];                                                          \
                    EXTRA                                                                           \
                    SB;                                                                             \
                    z = z + p0; SB; const float t0_ = WN.x * qb<HK>(HQ.x); SB;                      \
                    z = z + p1; SB; const float t1_ = WN.y * qb<HK>(HQ.y); SB;                      \
                    z = z + p2; SB; const float t2_ = WN.z * qb<HK>(HQ.z); SB;                      \
                    z = z + p3; SB; const float t3_ = WN.w * qb<HK>(HQ.w); SB;                      \
                    p0 = t0_; p1 = t1_; p2 = t2_; p3 = t3_;                                         \
                }
                for (int q = 0; q < NBLK / 4; ++q) {
                    const int qn = q + 1 < NBLK / 4 ? q + 1 : q;
                    STEP(w3, 3, w1, hq, 1, hn = hp[16 * qn];)
                    STEP(w0, 4, w2, hq, 2, )
                    STEP(w1, 5, w3, hq, 3, )
                    STEP(w2, 6, w0, hn, 0, )
                    hq = hn;
                    wp += 4 * NROW;
                }
#undef STEP
                acc[0] = z;
            } else if constexpr (MODE == 1 || MODE == 2) {
                const int s = wave & 3;
                const float *hb = hmir + s * NBLK * 4;
                float z = 0.f;
                i16v ha = sload16(hb, 0), hbn;
                f4 wa[4], wb[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) wa[j] = wp[j * NROW];
                auto blk4 = [&](const i16v &h, const f4 (&w)[4]) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if constexpr (MODE == 1) {
                            const float q0 = w[j].x * HS(h, 4 * j), q1 = w[j].y * HS(h, 4 * j + 1), q2 = w[j].z * HS(h, 4 * j + 2), q3 = w[j].w * HS(h, 4 * j + 3);
                            z = z + q0; z = z + q1; z = z + q2; z = z + q3;
                        } else {
                            const f2 wl = {w[j].x, w[j].y}, wh = {w[j].z, w[j].w};
                            const f2 q01 = pkmul_el(HP(h, 2 * j), wl), q23 = pkmul_el(HP(h, 2 * j + 1), wh);
                            z = z + q01.x; z = z + q01.y; z = z + q23.x; z = z + q23.y;
                        }
                    }
                };
                for (int q = 0; q < NBLK / 4; q += 2) {
                    wait0(ha);
                    hbn = sload16(hb, (q + 1) * 64);
#pragma unroll
                    for (int j = 0; j < 4; ++j) wb[j] = wp[((q + 1) * 4 + j) * NROW];
                    blk4(ha, wa);
                    wait0(hbn);
                    const int q2 = q + 2 < NBLK / 4 ? q + 2 : q;
                    ha = sload16(hb, q2 * 64);
#pragma unroll
                    for (int j = 0; j < 4; ++j) wa[j] = wp[(q2 * 4 + j) * NROW];
                    blk4(hbn, wb);
                }
                wait0(ha);
                acc[0] = z;
            } else if constexpr (MODE == 3) {
                const int pr = wave & 1;
                const float *hb = hmir + pr * NBLK * 8;
                f2 z = {0.f, 0.f};
                i16v ha0 = sload16(hb, 0), ha1 = sload16(hb, 64), hb0, hb1;
                f4 wa[4], wb[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) wa[j] = wp[j * NROW];
                auto blk2 = [&](const i16v &h, const f4 &w0, const f4 &w1) {
                    const f2 a0 = {w0.x, w0.y}, a1 = {w0.z, w0.w}, b0 = {w1.x, w1.y}, b1 = {w1.z, w1.w};
                    f2 p0 = pkmul_lo(HP(h, 0), a0), p1 = pkmul_hi(HP(h, 1), a0), p2 = pkmul_lo(HP(h, 2), a1), p3 = pkmul_hi(HP(h, 3), a1);
                    z = pkadd(z, p0); z = pkadd(z, p1); z = pkadd(z, p2); z = pkadd(z, p3);
                    p0 = pkmul_lo(HP(h, 4), b0); p1 = pkmul_hi(HP(h, 5), b0); p2 = pkmul_lo(HP(h, 6), b1); p3 = pkmul_hi(HP(h, 7), b1);
                    z = pkadd(z, p0); z = pkadd(z, p1); z = pkadd(z, p2); z = pkadd(z, p3);
                };
                for (int q = 0; q < NBLK / 4; q += 2) {
                    wait0(ha0, ha1);
                    hb0 = sload16(hb, (q + 1) * 128); hb1 = sload16(hb, (q + 1) * 128 + 64);
#pragma unroll
                    for (int j = 0; j < 4; ++j) wb[j] = wp[((q + 1) * 4 + j) * NROW];
                    blk2(ha0, wa[0], wa[1]); blk2(ha1, wa[2], wa[3]);
                    wait0(hb0, hb1);
                    const int q2 = q + 2 < NBLK / 4 ? q + 2 : q;
                    ha0 = sload16(hb, q2 * 128); ha1 = sload16(hb, q2 * 128 + 64);
#pragma unroll
                    for (int j = 0; j < 4; ++j) wa[j] = wp[(q2 * 4 + j) * NROW];
                    blk2(hb0, wb[0], wb[1]); blk2(hb1, wb[2], wb[3]);
                }
                wait0(ha0, ha1);
                acc[0] = z.x; acc[1] = z.y;
            } else {
                const float *hb = hmir;
                f2 z01 = {0.f, 0.f}, z23 = {0.f, 0.f};
                i16v ha0 = sload16(hb, 0), ha1 = sload16(hb, 64), hb0, hb1;
                f4 wa[2], wb[2];
                wa[0] = wp[0]; wa[1] = wp[NROW];
                auto blk1 = [&](const i16v &h, const f4 &w0) {
                    const f2 a0 = {w0.x, w0.y}, a1 = {w0.z, w0.w};
                    f2 p0 = pkmul_lo(HP(h, 0), a0), r0 = pkmul_lo(HP(h, 1), a0), p1 = pkmul_hi(HP(h, 2), a0), r1 = pkmul_hi(HP(h, 3), a0);
                    z01 = pkadd(z01, p0); z23 = pkadd(z23, r0); z01 = pkadd(z01, p1); z23 = pkadd(z23, r1);
                    p0 = pkmul_lo(HP(h, 4), a1); r0 = pkmul_lo(HP(h, 5), a1); p1 = pkmul_hi(HP(h, 6), a1); r1 = pkmul_hi(HP(h, 7), a1);
                    z01 = pkadd(z01, p0); z23 = pkadd(z23, r0); z01 = pkadd(z01, p1); z23 = pkadd(z23, r1);
                };
                for (int b = 0; b < NBLK; b += 4) {
                    wait0(ha0, ha1);
                    hb0 = sload16(hb, (b + 2) * 64); hb1 = sload16(hb, (b + 3) * 64);
                    wb[0] = wp[(b + 2) * NROW]; wb[1] = wp[(b + 3) * NROW];
                    blk1(ha0, wa[0]); blk1(ha1, wa[1]);
                    wait0(hb0, hb1);
                    const int b2 = b + 4 < NBLK ? b + 4 : b;
                    ha0 = sload16(hb, b2 * 64); ha1 = sload16(hb, (b2 + 1) * 64);
                    wa[0] = wp[b2 * NROW]; wa[1] = wp[(b2 + 1) * NROW];
                    blk1(hb0, wb[0]); blk1(hb1, wb[1]);
                }
                wait0(ha0, ha1);
                acc[0] = z01.x; acc[1] = z01.y; acc[2] = z23.x; acc[3] = z23.y;
            }
        } else if (partner && wave >= 4) {
            // the wave sharing the SIMD runs GRU-A-like items meanwhile: ds_read_b128 + 16 DPP multiplies + 16 adds
            float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
            for (int it = 0; it < partner; ++it) {
                const f4 hv = lh[(it * 5 + (lane >> 3)) & 255];
                const float hk[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    a0 = a0 + hk[c] * qb<0>(hk[c]); a1 = a1 + hk[c] * qb<1>(hk[c]); a2 = a2 + hk[c] * qb<2>(hk[c]); a3 = a3 + hk[c] * qb<3>(hk[c]);
                }
            }
            dummy += a0 + a1 + a2 + a3;
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        tsum += t1 - t0;
        if (t == 0) tfirst = t1 - t0;
        __syncthreads();
    }
    if (wave < active && lane < NROW) {
#pragma unroll
        for (int s = 0; s < SPW; ++s) out[(wave * 64 + lane) * 4 + s] = acc[s];
    }
    if (dummy == 12345.f) out[4095] = dummy;
    if (lane == 0) { clk[wave] = tsum; clk[8 + wave] = tfirst; }
}

static float host_ref(int row, int s, int t)
{
    volatile float z = 0.f;
    for (int blk = 0; blk < NBLK; ++blk)
        for (int c = 0; c < 4; ++c) {
            volatile float p = wval(row, blk, c) * hval(s, blk * 4 + c, t);
            z = z + p;
        }
    return z;
}

int main()
{
    float *d_h, *d_out; unsigned long long *d_clk;
    hipMalloc(&d_h, 8192 * 4); hipMalloc(&d_out, 4096 * 4); hipMalloc(&d_clk, 64 * 8);
    const int STEPS = 200;
    auto run = [&](const char *name, auto kern, int mode, int active, int cold, int partner) {
        hipMemset(d_out, 0, 4096 * 4);
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 90000);
        hipLaunchKernelGGL(kern, dim3(1), dim3(512), 90000, 0, d_h, d_out, d_clk, STEPS, active, cold, partner);
        if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", name); return; }
        unsigned long long c[16]; std::vector<float> o(4096);
        hipMemcpy(c, d_clk, sizeof(c), hipMemcpyDeviceToHost);
        hipMemcpy(o.data(), d_out, 4096 * 4, hipMemcpyDeviceToHost);
        const int spw = mode == 3 ? 2 : (mode == 4 ? 4 : 1);
        int bad = 0;
        for (int w = 0; w < active; ++w)
            for (int r = 0; r < NROW; ++r)
                for (int s = 0; s < spw; ++s) {
                    const int stream = mode == 3 ? (w & 1) * 2 + s : (mode == 4 ? s : (w & 3));
                    if (o[(w * 64 + r) * 4 + s] != host_ref(r, stream, STEPS - 1)) ++bad;
                }
        printf("%-34s active=%d cold=%d partner=%3d  clk/step:", name, active, cold, partner);
        for (int w = 0; w < active; ++w) printf(" %6.0f", (double)c[w] / STEPS);
        printf("  | per block %5.1f | first step %llu | mismatches %d\n", (double)c[0] / STEPS / NBLK, c[8], bad);
    };
    for (int partner : {0, 30}) {
        for (int active : {1, 2, 4}) {
            run("0 LDS state + DPP (today)", k<0>, 0, active, 0, partner);
            for (int cold : {0, 1}) {
                run("1 S1  s_load + v_mul/v_add", k<1>, 1, active, cold, partner);
                run("2 S1p s_load + pk_mul/v_add", k<2>, 2, active, cold, partner);
                if (active <= 2) run("3 S2  s_load + pk_mul/pk_add", k<3>, 3, active, cold, partner);
                if (active == 1) run("4 S4  s_load + pk x16", k<4>, 4, active, cold, partner);
            }
        }
    }
    return 0;
}
