// Persistent per-sample kernel of the LPCNet synthesis engine (gfx950 / CDNA4).
//
// Replaces the reference's per-sample path: lpcnet_synthesize_tail_impl (src/lpcnet.c:235-271)
// -> run_sample_network (:146-167) -> compute_gru_a_input (src/nnet.c:484-491),
// compute_sparse_gru (:410-448), compute_gruB (:326-372), sample_mdense (:163-214), and the
// generic mat-vec kernels of src/vec.h:131-162 / :347-404.
//
// Arithmetic contract ("PARITY"): every output row accumulates its products in the reference's
// generic-C order (block order, then the 4 columns of a block), each product and each sum
// rounded separately (no FMA: build with -ffp-contract=off), table tanh.  Rows are independent,
// so rows go to lanes and the per-row order is kept inside a lane -> results are bit-identical
// to the reference's generic-C float build.
//
// Mapping (one workgroup = 8 waves = 512 lanes, S interleaved streams, S in {1,2,4}):
//   * GRU-A recurrent weights (177 KB fp32 > 160 KB LDS) live in VGPRs for the whole launch:
//     each lane owns <=3 output rows ("slots", dealt by model_pack.c so that the 64 rows of a
//     wave-slot have similar block counts) as NW items (float4, or one dword of 4 int8).
//   * GRU-A state h (S x 384) lives in LDS as [block p][stream][4]; the 8 lanes of a row group
//     need the same 16*S bytes, so each lane of a quad fetches ONE stream's 16 B (a single
//     ds_read_b128 per item) and the other three streams arrive through DPP quad_perm
//     broadcasts folded into the multiplies -- LDS traffic is 1/S of the naive scheme.
//     int8 blobs: the state is quantised once per sample; v_dot4_i32_i8 per (item, stream).
//   * GRU-B input weights (<=73.7 KB) + recurrent matrix live in LDS; one wave per stream, one lane per output row, gates
//     exchanged with wave shuffles (no barrier).  Float blobs with a dense GRU-B input matrix: its state operand -- the same
//     384 values for every row -- is a BROADCAST LDS read of the block the gate stage has just written, and the whole 96-block
//     loop is one hand-scheduled assembly block (grub_lds_loop_s{1,2,4}.inc, tools/gen_grub_asm.py --lds S; round 4).
//     int8 blobs and block-sparse matrices use the LDS + DPP / dot4 forms below.  (Round 3's form -- the state mirrored into an L2-resident buffer
//     and pulled through the scalar cache into SGPR-pair operands -- and the other measured alternatives are in the history: EXPERIMENTS.md.)
//   * dual-FC tree: all 255 nodes x 2 channels are evaluated in parallel (lane = node,channel;
//     18 weight VGPRs), a ballot per wave yields the 255 decision bits.  Speculative evaluation
//     is exact: every node's logit is a pure function of the GRU-B state.
//   * wave 0 leads: the 16 lanes of row s walk stream s's tree and hold its 16-sample LPC history
//     (lane = tap; DPP row shift / row broadcast for the prediction), one mu-law pass, PCM.
//     Wave 1 draws the KISS99 thresholds of the next sample.
//
// Schedule of one sample (4 workgroup barriers B1..B4):
//   P1 GRU-A rows | B1 | P2 gates | B2 | P3 GRU-B (waves < S)  ||  the HEAD of the next sample's
//   candidate chains on waves 4..7 | B3 | P4 tree | B4 | leader publishes the next sample's mu-law indices through an LDS
//   flag -- no barrier: the other waves are already in P1, running what needs neither the indices nor the gathered
//   embedding rows (the TAIL of their candidate chains), then poll the flag and gather.
//   A candidate row's sum is sequential, but it need not be formed in one go: model_pack.c stores the first `head` blocks of
//   every row of a wave's candidate slot end-aligned in the item array; waves 4..7 run them one sample ahead in GRU-B's
//   shadow, park the partial sums in LDS, and the slot continues from there in P1.
//
// Round 5:
//   * int8 blobs, <= 2 streams per workgroup (two workgroups per CU): GRU-B's chain waves are 2 and 3, not 0 and 1 -- wave 0 also leads the
//     streams, waves w and w + 4 share a SIMD, and two co-resident workgroups are bound by their busiest SIMD; the six other waves all carry
//     candidate heads of up to 22 items (lpcnet_engine.h: LPCN_I8_GBWA / _GBWB, LPCN_DEAL_HMASK_I8, LPCN_DEAL_EH_I8).
//   * float blobs with more than 32 items per lane keep 28 in VGPRs and STREAM the rest from L2 every sample (see NR / fetch_w): no variant
//     has a scratch access inside the sample loop any more, and 48 items per lane (1.83 x the benchmark model's blocks) load.
//
// FAST (lpcnet_batch_set_fast) frees the order of a row's sum, and the kernel then uses the matrix pipe: a GRU-A item is
// four v_mfma_f32_4x4x1 (4 rows' weights x 4 streams' state values per quad), and -- float blobs, dense GRU-B -- the GRU-B
// input mat-vec of all S streams is one [48 x 384] x [384 x S] GEMM on the gate waves (see gb_mfma in P3).
//
// Two things the compiler does to this kernel that cost more than any instruction in it (both found in the assembly):
//   * a FLAT store anywhere in the sample loop (the test trace, through a generic pointer) makes SIInsertWaitcnts force
//     every later LDS wait to lgkmcnt(0) until both counters have been drained: the state prefetch of the item chains no
//     longer overlapped anything.  All trace stores go through address-space-1 pointers.
//   * a member of the argument block read inside the loop is a scalar load, and a scalar load in flight forces lgkmcnt(0)
//     as well (SMEM returns out of order); see the argument-block members fetched once below.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "lpcnet_engine.h"
#include "lpcnet_math.h"

struct LpcnSampleArgs {
    // model (device pointers)
    const float *emb_sig, *emb_pred, *emb_exc;      // [256][3 slots][512 threads]: lane-ordered embedding tables
    const float4 *a_w;                              // [8][NW][64] float4 (fp32 blobs) or [8][NW][64] dwords of 4 int8 (int8 blobs)
    const uint8_t *a_blk;                           // [8][NW][64]
    const int *a_row;                               // [8][3][64]
    const int *a_bound;                             // [8][4]
    const int *a_allh;                              // [8][3]
    const int *a_head;                              // [8] float blobs: items [NW - head, NW) = the first `head` chain items of slot 0's rows, computed one sample ahead
    const float *a_bias1;                           // [1152] recurrent bias row
    const float *a_diag;                            // [1152]
    const float *b_w;                               // [nb_b][8][4] fp32, or [nb_b/4][8][4 blocks] dwords of 4 int8
    const int *b_start;                             // [7]
    const uint8_t *b_blk;                           // [nb_b]
    const float *b_rec;                             // [16][48] fp32, or [48 rows][4] dwords of 4 int8
    const float *b_bias;                            // [2][48]
    const float *fc_w, *fc_b, *fc_f;                // [256][2][16], [2][256], [2][256]
    const float *tab_tansig, *tab_ulaw2lin, *tab_logit;
    int nb_b;
    int b_dense;                                    // GRU-B lists all 96 input blocks for every row group, in order
    // work
    int n_streams, n_frames, preload, fc_advance;
    int frame_len;                                  // samples synthesised per frame (1..160)
    const float *cond_a;                            // [stream][frame][1152]
    const float *cond_b;                            // [stream][frame][48]
    const float *lpc;                               // [stream][frame][16]
    const int *fc_base;                             // [stream] frame_count reference value
    short *pcm;                                     // [stream] x pcm_stride samples, frame f at +f*160
    long long pcm_stride;
    lpcn_stream_state *state;                       // [stream]
    float *dbg;                                     // optional per-sample trace (tests)
    unsigned long long *prof;                       // optional: [8] shader-clock totals per phase, workgroup 0 wave 0
    const uint32_t *fc_wh;                          // FAST sub-option: dual-FC weights as fp16 pairs [256][2][8] (BASELINE config 4: "fp16 dual-FC")
    int fc_f16;                                     // non-zero: the tree phase runs on fc_wh with v_dot2_f32_f16 (FAST only)
    const float *emb_nat_sig, *emb_nat_pred, *emb_nat_exc;      // [256][1152]: the embedding tables in the blob's own row order (float blobs; sample_kernel_x2.hip.h)
};

#ifndef LPCN_ENABLE_PROF
#define LPCN_ENABLE_PROF 0
#endif
#define LPCN_DBG_STRIDE 1600    // floats per (sample) trace record: hA 384, hB 16, exc,sig,pred,pcm,pred, leader clocks barrier->publish, the tree's own decision; [448..1600) GRU-A pre-activations

namespace lpcn {

constexpr int NA = LPCN_N_A, NB = LPCN_N_B, RA = LPCN_ROWS_A, RB = LPCN_ROWS_B;

// pointers fetched from the argument block are generic; tell the compiler they are global memory
#define LPCN_GLOBAL __attribute__((address_space(1)))
template <typename T> __device__ __forceinline__ const LPCN_GLOBAL T *as_global(const T *p)
{
    return (const LPCN_GLOBAL T *)(uintptr_t)p;
}
template <typename T> __device__ __forceinline__ LPCN_GLOBAL T *as_global_rw(T *p)
{
    return (LPCN_GLOBAL T *)(uintptr_t)p;
}
// Loop-invariant values that hipcc would otherwise hoist out of the 160-sample loop and keep in
// VGPRs for the whole launch; the weights need that register space.
#define LPCN_REMAT_V(x) asm volatile("" : "+v"(x))
#define LPCN_REMAT_S(x) asm volatile("" : "+s"(x))

// ---- LDS carve-up (bytes), all offsets multiples of 16 ---------------------------------------
#define LPCN_PROD_BLOCKS 48      // single stream per workgroup: blocks of GRU-B whose products the helper waves form (the two assembly loops are generated for this split)
template <int S> struct Lds {
    static constexpr int HA_STRIDE = 16 * S;                       // bytes per 4-neuron block
    static constexpr int hA     = 0;
    static constexpr int hA_sz  = 96 * HA_STRIDE + 24 * 16;         // 16 B pad every 4 blocks
    static constexpr int pre    = hA + hA_sz;                       // [S][1152] f32 pre-activations
    static constexpr int inh    = pre + S * RA * 4;                 // [S][384]  input part of candidate rows
    static constexpr int cond   = inh + S * NA * 4;                 // [S][1152] frame conditioning (GRU-A)
    static constexpr int abias  = cond + S * RA * 4;                // [1152]{bias, diag}: recurrent bias and diagonal weight per row
    static constexpr int adiag  = abias + RA * 4;                   // (second half of the interleaved {bias, diag} table)
    static constexpr int hT     = adiag + RA * 4;                   // [384][S] second copy of the GRU-A state, stream-interleaved
    static constexpr int hB     = hT + NA * S * 4;                  // [S][16]
    static constexpr int idx    = hB + S * NB * 4;                  // [S][4] i32 (sig,pred,exc,live)
    static constexpr int thr    = idx + S * 16;                     // [S][8] f32
    static constexpr int mask   = thr + S * 32;                     // [S][8] u64
    static constexpr int lead   = mask + S * 64;                    // [S][8] leader scalars (pred,deemph,exc,head,rng4)
    static constexpr int flag   = lead + S * 32;                    // [4] i32: sequence number of the newest published sample indices
    static constexpr int condb  = flag + 16;                        // [S][48] f32
    static constexpr int lpc    = condb + S * RB * 4;               // [S][16] f32
    static constexpr int sig    = lpc + S * 64;                     // [S][16] f32 ring of past samples
    static constexpr int pcmbuf = sig + S * 64;                     // [S][160] i16
    static constexpr int tansig = pcmbuf + S * 320;                 // [204] f32
    static constexpr int ulaw   = tansig + 816;                     // [256] f32 mu-law decode table
    static constexpr int logit  = ulaw + 1024;                      // [256] f32 sampling thresholds
    static constexpr int brec   = logit + 1024;                     // [16][48]
    static constexpr int bbias  = brec + NB * RB * 4;               // [2][48]
    static constexpr int bstart = bbias + 2 * RB * 4;               // [8] i32
    static constexpr int bblk   = bstart + 32;                      // [<=608] u8, groups padded to x4
    static constexpr int boff   = bblk + 608;                       // [<=608] u16 LDS offsets of the GRU-B input blocks
    static constexpr int bw     = boff + 1216;                      // [nb_b padded][8][4] f32
    static constexpr int hBh(int nb_b, bool i8) { return bw + (nb_b + (i8 ? 28 : 8)) * (i8 ? 32 : 128); }     // [S][16] f16: GRU-B state as halves (FAST fp16 dual FC), behind everything else
    // single-stream PARITY float: idle waves hand GRU-B's chain wave the PRODUCTS of the last PROD_BLOCKS blocks through LDS
    // ([block][48 rows] float4, 768 B per block) -- the chain then costs one read + four adds per block instead of two reads, two
    // packed multiplies and four adds.  Only S = 1 has the room (50 KB free of the 160 KB).
    static constexpr int PROD_BLOCKS = LPCN_PROD_BLOCKS, PROD_FIRST = 96 - PROD_BLOCKS;      // (the two assembly loops are generated for this split)
    static constexpr int prod_sz = S == 1 ? (PROD_BLOCKS + 7) * RB * 16 : 0;      // (S = 1, + 7 blocks: the chain wave's ring of eight reads ahead past the last block on its last trip)
    static constexpr int prod(int nb_b, bool i8) { return hBh(nb_b, i8) + S * 32; }
    static constexpr int total(int nb_b, bool i8) { return hBh(nb_b, i8) + S * 32 + (i8 ? 0 : prod_sz); }      // (bw pad: the GRU-B pipeline reads ahead)
    // int8 engine: the quantised states overlay the region of the fp32 engine's block-ordered float state
    static constexpr int xq     = hA;                               // [96 blocks][S] dwords: 4 int8 of one stream's block
    static constexpr int xqT    = hA + 384 * S;                     // [S][96] dwords: the same, stream-major (GRU-B input)
    static constexpr int hBq    = hA + 768 * S;                     // [S][4] dwords: quantised GRU-B state
    __host__ __device__ static constexpr int ha_off(int p) { return p * HA_STRIDE + (p >> 2) * 16; }
};

template <int SEL> __device__ __forceinline__ float quad_bcast(float v)
{
    // value held by lane (quad base + SEL); the compiler folds this into v_mul_f32_dpp
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), SEL * 0x55, 0xf, 0xf, true));
}

// FAST float path: acc += w * (h held by lane quad_base + SEL) as ONE v_fmac_f32_dpp (VOP2 with a DPP source).  hipcc
// does not fold a DPP move into a fused multiply-add (it emits v_mov_b32_dpp + v_fma_f32), hence inline assembly.
// DPP hazard: a VGPR written by a VALU instruction must not be read through DPP in the next two issue slots; `h` is
// always the destination of an LDS read here, and tools/kernel_resources.py --check-dpp verifies it on the assembly.
template <int SEL> __device__ __forceinline__ float fmac_quad(float acc, const float w, const float h)
{
    if constexpr (SEL == 0) asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(h), "v"(w));
    if constexpr (SEL == 1) asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(h), "v"(w));
    if constexpr (SEL == 2) asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(h), "v"(w));
    if constexpr (SEL == 3) asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(h), "v"(w));
    return acc;
}

// int8 (DOT_PROD) arithmetic of the reference's generic-C build, src/vec.h:274-339:
//   x_q = (signed char)(int)floor(.5 + 127*x)   (float product, double sum)
//   out = out*(128*127);  out += (w0*x0 + w1*x1 + w2*x2 + w3*x3) per block (exact integer);  out *= 1/128/127
constexpr float QS = 128.f * 127.f, QS1 = 1.f / 128.f / 127.f;
__device__ __forceinline__ int quant_s8(float x)
{
    const float t = 127.f * x;
    // floor(.5 + t) as ONE instruction, v_cvt_rpi_i32_f32 (round to nearest, ties toward +infinity): == (int)floor(.5 + (double)t) for every finite t
    // (lpcnet_hip_quant_sweep_device: all 2^32 bit patterns on the device)
    int q;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(q) : "v"(t));
    return q & 0xFF;
}
// Integer dot products converted to float.  v_dot4_i32_i8 with a literal-zero accumulator saves the
// v_mov the compiler's v_dot4c selection needs, but the hazard recogniser cannot see inside inline
// asm: gfx90a+ requires 3 wait states between a DOT write and a different VALU op reading the
// result (and 4 before a different VALU op overwrites it), so every block below carries its own
// spacing -- the dots of the other streams, or s_nop.
__device__ __forceinline__ void dot4_cvt_x4(float (&f)[4], int w0, int w1, int w2, int w3, int x0, int x1, int x2, int x3)
{
    int t0, t1, t2, t3;
    asm("v_dot4_i32_i8 %4, %8, %12, 0\n\t"
        "v_dot4_i32_i8 %5, %9, %13, 0\n\t"
        "v_dot4_i32_i8 %6, %10, %14, 0\n\t"
        "v_dot4_i32_i8 %7, %11, %15, 0\n\t"
        "v_cvt_f32_i32 %0, %4\n\t"
        "v_cvt_f32_i32 %1, %5\n\t"
        "v_cvt_f32_i32 %2, %6\n\t"
        "v_cvt_f32_i32 %3, %7"
        : "=&v"(f[0]), "=&v"(f[1]), "=&v"(f[2]), "=&v"(f[3]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(x0), "v"(x1), "v"(x2), "v"(x3));
}
__device__ __forceinline__ float dot4_cvt(int w, int x)
{
    float f;
    int t0;
    asm("v_dot4_i32_i8 %1, %2, %3, 0\n\t"
        "s_nop 2\n\t"
        "v_cvt_f32_i32 %0, %1"
        : "=&v"(f), "=&v"(t0)
        : "v"(w), "v"(x));
    return f;
}
template <int S> struct XVec;
template <> struct XVec<1> { typedef int type; };
template <> struct XVec<2> { typedef int type __attribute__((ext_vector_type(2))); };
template <> struct XVec<4> { typedef int type __attribute__((ext_vector_type(4))); };

// value of lane J of the caller's 16-lane row (DPP row_newbcast); the compiler folds it into the consuming VALU op
template <int J> __device__ __forceinline__ float row_bcast(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150 + J, 0xf, 0xf, false));
}
template <int J> __device__ __forceinline__ float lpc_chain(float r, float prod)
{
    if constexpr (J < LPCN_LPC_ORDER) return lpc_chain<J + 1>(r - row_bcast<J>(prod), prod);
    else return r;
}

// PACK2 variants: the register allocation leaves room for 4 waves per SIMD (128 VGPRs per lane), i.e. TWO workgroups per
// CU, which fill each other's barrier / latency bubbles (measured: a second resident workgroup slows the first by only
// ~15 %).  Only the int8 kernels with <= 32 items per lane and S <= 2 get there without a scratch access inside the
// sample loop (tools/kernel_resources.py), and only their LDS footprint (~66 KB) lets two workgroups share a CU; S = 4 would
// need 128 VGPRs AND <= 80 KB (it has ~92 KB): tried, 119 vs 137 M samples/s.  The engine picks a PACK2 variant when a
// batch has more workgroups than the device has CUs.

// Accumulator of one GRU row while its items run, kept in a float VGPR:
//   PARITY, float blob : the float sum itself (src/vec.h:355-401)
//   PARITY, int8 blob  : the float sum scaled by 128*127, one exactly-integer block product added at a time (src/vec.h:306-339)
//   FAST,   int8 blob  : the bit pattern of an int32 -- start value rounded once to the 1/(128*127) grid, blocks
//                        accumulated exactly by v_dot4_i32_i8, converted back at the end: the arithmetic of the
//                        reference's AVX2 int8 build (src/vec_avx.h:700-705,742-744 and :803-858)
template <bool I8, bool FAST> __device__ __forceinline__ float acc_start(float v)
{
    if constexpr (I8 && FAST) return __builtin_bit_cast(float, (int)__builtin_rintf(v * QS));
    else if constexpr (I8) return v * QS;
    else return v;
}
template <bool I8, bool FAST> __device__ __forceinline__ float acc_final(float a)
{
    if constexpr (I8 && FAST) return (float)__builtin_bit_cast(int, a) * QS1;
    else if constexpr (I8) return a * QS1;
    else return a;
}

// Activations.  PARITY: the reference's generic-C table tanh (src/vec.h:82-104), one LDS lookup each.  FAST: the hardware
// exponential and reciprocal -- tanh(x) = 1 - 2/(2^(2x log2 e) + 1), sigmoid(x) = 1/(1 + 2^(-x log2 e)); like the reference's
// AVX2 build, which also replaces the table by an approximation of its own (src/vec_avx.h:393-440), accurate to a few 1e-7
// where the table version is accurate to ~1e-5.  Saturates correctly: 2^(+big) = inf -> 1, 2^(-big) = 0 -> -1.
template <bool FAST> __device__ __forceinline__ float act_tanh(const float x, const float *tab)
{
    if constexpr (FAST) return 1.f - 2.f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(x * 2.885390082f) + 1.f);
    else return lpcn_tanh(x, tab);
}
template <bool FAST> __device__ __forceinline__ float act_sigmoid(const float x, const float *tab)
{
    if constexpr (FAST) return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * -1.442695041f));
    else return lpcn_sigmoid(x, tab);
}

// FAST = the arithmetic of the reference's SIMD builds instead of its generic-C order: fused multiply-add for float
// blobs (src/vec_avx.h:790-858 _mm256_fmadd_ps), int32 block accumulation for int8 blobs (see acc_start).  Results are
// no longer bit-identical to the generic-C build; tests/test_gpu_fast.py bounds the deviation teacher-forced.
template <int S, int NW, bool I8, bool FAST, bool PACK2 = false>
__global__ __launch_bounds__(LPCN_WG_THREADS, PACK2 ? 4 : 2) void sample_kernel(const LpcnSampleArgs *__restrict__ Ap)
{
    using L = Lds<S>;
    using WT = typename std::conditional<I8, int, float4>::type;          // one resident item
    constexpr bool I8M = I8 && !FAST && S >= 2;             // PARITY, int8 blobs, S >= 2: an item's block products of all streams from one v_mfma_i32_4x4x4i8 (exact integers, like v_dot4)
    using HT = typename std::conditional<I8, typename std::conditional<I8M, int, typename XVec<S>::type>::type, float4>::type;   // one fetched state block
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *const sm_pre = (float *)(smem + L::pre);
    float *const sm_inh = (float *)(smem + L::inh);
    float *const sm_cond = (float *)(smem + L::cond);
    const float *const sm_abias = (const float *)(smem + L::abias);
    const float *const sm_adiag = (const float *)(smem + L::adiag);
    float *const sm_hT = (float *)(smem + L::hT);
    float *const sm_hB = (float *)(smem + L::hB);
    int *const sm_idx = (int *)(smem + L::idx);
    float *const sm_thr = (float *)(smem + L::thr);
    unsigned long long *const sm_mask = (unsigned long long *)(smem + L::mask);
    float *const sm_lead = (float *)(smem + L::lead);
    float *const sm_condb = (float *)(smem + L::condb);
    float *const sm_lpc = (float *)(smem + L::lpc);
    short *const sm_pcm = (short *)(smem + L::pcmbuf);
    const float *const sm_tansig = (const float *)(smem + L::tansig);
    const float *const sm_ulaw = (const float *)(smem + L::ulaw);
    const float *const sm_logit = (const float *)(smem + L::logit);
    const float *const sm_brec = (const float *)(smem + L::brec);
    const float *const sm_bbias = (const float *)(smem + L::bbias);
    const int *const sm_bstart = (const int *)(smem + L::bstart);
    const unsigned char *const sm_bblk = smem + L::bblk;
    const unsigned short *const sm_boff = (const unsigned short *)(smem + L::boff);
    const float *const sm_bw = (const float *)(smem + L::bw);

    const int tid0 = threadIdx.x;
    const int n_streams = Ap->n_streams, n_frames = Ap->n_frames, preload = Ap->preload, frame_len = Ap->frame_len;
    const int s0 = blockIdx.x * S;                          // first stream of this workgroup
    // streams past the end are computed on a clamped copy and never written back
    auto stream_of = [&](int s) { return (s0 + s < n_streams) ? s0 + s : n_streams - 1; };
    const int n_valid = (n_streams - s0 < S) ? n_streams - s0 : S;
    const size_t nf = (size_t)n_frames;
    auto *const states = as_global_rw(Ap->state);
    const auto *const emb_sig = as_global(Ap->emb_sig);
    const auto *const emb_pred = as_global(Ap->emb_pred);
    const auto *const emb_exc = as_global(Ap->emb_exc);

    // ------------------------------------------------------------------ resident weights ----
    // Float blobs denser than 32 items per lane: the first 28 items of a lane stay in VGPRs, the others are STREAMED -- re-fetched from
    // the L2-resident item array every sample, WSD - 1 items ahead of their use, into a ring of WSD registers quads.  (Round 4 kept all 36 / 40
    // items resident: the allocator then spilled into the sample loop -- 94 scratch accesses per sample at 40 items, 91 M samples/s for a model
    // with 1.53 x the benchmark model's blocks.  A spilled weight costs the same 1 KB per wave and sample as a streamed one, but its reload sits
    // right in front of its use.)  Only the waves that own more than NR items ever execute a streamed one.
    constexpr int NR = (!I8 && NW > 32) ? 28 : NW;          // resident items (ring depth 4 / 6 and 24 / 28 / 32 resident items measured within 1 %)
    constexpr int WSD = 4;
    WT w[NR];
    WT ws[NR < NW ? WSD : 1] = {};
    // LDS offsets of the items' state blocks, two per register -- of the RESIDENT items; a streamed item's block index is re-fetched with it (one byte
    // per lane and item from a_blk, WSOD - 1 items ahead of its state read): 64 / 80 items per lane would otherwise hold 32 / 40 registers of offsets
    // (round 6: heavy-tailed, trained-like sparsity needs those variants)
    constexpr int WSOD = 8;
    constexpr bool OFFB = I8 && NW > 64;                     // int8 blobs, 96 items per lane: the block INDICES, four per register (48 registers of offsets do not fit beside 96 of weights)
    uint32_t offp[OFFB ? (NW + 3) / 4 : (NR + 1) / 2];
    int wso[NR < NW ? WSOD : 1] = {};
    int row_reg[3];
#define LPCN_ROW(k) (row_reg[k])
    {
        const int lane = tid0 & 63, wave = tid0 >> 6;
        const size_t base = (size_t)wave * NW * 64 + lane;
        const int lane_sel = (S >= 4 ? (lane & 3) : (S == 2 ? (lane & 1) : 0)) * 16;
        const auto *ab = as_global(Ap->a_blk);
        if constexpr (I8) {
            const auto *aq = (const LPCN_GLOBAL int *)as_global(Ap->a_w);
#pragma unroll
            for (int j = 0; j < NR; ++j) w[j] = aq[base + (size_t)j * 64];
        } else {
            const auto *aw = (const LPCN_GLOBAL float *)as_global(Ap->a_w);
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const auto *v = aw + (base + (size_t)j * 64) * 4;
                w[j] = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
        if constexpr (OFFB) {
#pragma unroll
            for (int j = 0; j < NW; j += 4) {
                uint32_t pk = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) pk |= (uint32_t)((j + k < NW) ? ab[base + (size_t)(j + k) * 64] : 0) << (8 * k);
                offp[j >> 2] = pk;
            }
        } else
#pragma unroll
        for (int j = 0; j < NR; j += 2) {
            const int p0 = ab[base + (size_t)j * 64];
            const int p1 = (j + 1 < NR) ? ab[base + (size_t)(j + 1) * 64] : 0;
            // (I8M: lane k of a quad fetches only stream k's dword -- the matrix pipe forms all (stream, row) pairs of the quad)
            if constexpr (I8M) offp[j >> 1] = (uint32_t)(p0 * 4 * S + (lane & (S - 1)) * 4) | ((uint32_t)(p1 * 4 * S + (lane & (S - 1)) * 4) << 16);
            else if constexpr (I8) offp[j >> 1] = (uint32_t)(p0 * 4 * S) | ((uint32_t)(p1 * 4 * S) << 16);
            else offp[j >> 1] = (uint32_t)(L::ha_off(p0) + lane_sel) | ((uint32_t)(L::ha_off(p1) + lane_sel) << 16);
        }
        const auto *ar = as_global(Ap->a_row);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            row_reg[k] = ar[(wave * 3 + k) * 64 + lane];
        }
    }
    int b1 = __builtin_amdgcn_readfirstlane(as_global(Ap->a_bound)[(tid0 >> 6) * 4 + 1]);
    int b2 = __builtin_amdgcn_readfirstlane(as_global(Ap->a_bound)[(tid0 >> 6) * 4 + 2]);
    int b3 = __builtin_amdgcn_readfirstlane(as_global(Ap->a_bound)[(tid0 >> 6) * 4 + 3]);   // this wave's item count
    const bool allh0 = __builtin_amdgcn_readfirstlane(as_global(Ap->a_allh)[(tid0 >> 6) * 3]) != 0;
    const bool b_dense = Ap->b_dense != 0;
    // PARITY, float blob, dense GRU-B input matrix: GRU-B reads the GRU-A state from LDS as a broadcast (grub_lds_loop_s*.inc)
    const bool gb_lds = !I8 && !FAST && b_dense;
    const bool gb_prod = S == 1 && gb_lds;                   // single stream per workgroup: waves 1..3 form the products of GRU-B's last PROD_BLOCKS blocks for the chain wave
    // Argument-block members the sample loop needs on its critical path are fetched ONCE and made opaque: left to itself the compiler re-reads them
    // with a scalar load at every use, and a scalar load in flight forces every LDS wait behind it to lgkmcnt(0) -- right behind a barrier that is a
    // stall for every wave (int8: 148 -> 154.5 M samples/s; float: 124.8 -> 125.4 M once GRU-B's scalar-state form had freed its 36 SGPRs).
    int tracing_s = 0;                                       // tests only: workgroup 0 writes the per-sample trace
    { tracing_s = __builtin_amdgcn_readfirstlane((Ap->dbg != nullptr && blockIdx.x == 0) ? 1 : 0); LPCN_REMAT_S(tracing_s); }
#define tracing (tracing_s != 0)
#define tracing_any (tracing_s != 0)
#define tracing_lane0(extra) (tracing_s != 0 && tid == 0 && (extra))
    const LPCN_GLOBAL float *fc_w_s = nullptr, *fc_b_s = nullptr, *fc_f_s = nullptr;
    {
        fc_w_s = as_global(Ap->fc_w); fc_b_s = as_global(Ap->fc_b); fc_f_s = as_global(Ap->fc_f);
        asm volatile("" : "+s"(fc_w_s), "+s"(fc_b_s), "+s"(fc_f_s));
    }
    const LPCN_GLOBAL char *aw_s = (const LPCN_GLOBAL char *)as_global(Ap->a_w);       // item array, for the streamed items (NR < NW)
    const LPCN_GLOBAL uint8_t *ab_s = as_global(Ap->a_blk);                             // ... and their block indices
    if constexpr (NR < NW) asm volatile("" : "+s"(aw_s), "+s"(ab_s));
#define fc_w_g fc_w_s
#define fc_b_g fc_b_s
#define fc_f_g fc_f_s
    // bit k: this wave owns rows in slot k (wave-uniform)
    const int has_slot = __builtin_amdgcn_readfirstlane((__ballot(row_reg[0] >= 0) != 0ull ? 1 : 0) | (__ballot(row_reg[1] >= 0) != 0ull ? 2 : 0) |
                                                        (__ballot(row_reg[2] >= 0) != 0ull ? 4 : 0));
    const bool has2 = (has_slot & 4) != 0;
    // Waves that do not run GRU-B compute the HEAD of their candidate slot's chains (the first `hl` blocks of every
    // row of slot 0, stored end-aligned at items [NW - hl, NW) by model_pack.c) one sample ahead, in GRU-B's shadow, and park
    // the partial sums (the raw accumulators: scaled by 128*127 for int8 blobs, an int32 pattern in FAST) in the rows' sm_pre
    // cells; the slot then starts from those in the next sample.
    const int hl = __builtin_amdgcn_readfirstlane(as_global(Ap->a_head)[tid0 >> 6]);
    const bool early_wave = hl > 0;                          // wave-uniform (model_pack.c: float blobs -- waves 4..7 only, never a GRU-B wave; int8 blobs at <= 2 streams -- every wave except GRU-B's chain waves 2, 3: LPCN_DEAL_HMASK_I8)

    // ------------------------------------------------------------------ LDS residents -------
    {
        const int tid = tid0;
        const int nb_b = Ap->nb_b;
        const auto *t0 = as_global(Ap->tab_tansig), *t1 = as_global(Ap->tab_ulaw2lin), *t2 = as_global(Ap->tab_logit);
        for (int i = tid; i < 201; i += LPCN_WG_THREADS) ((float *)(smem + L::tansig))[i] = t0[i];
        for (int i = tid; i < 256; i += LPCN_WG_THREADS) {
            ((float *)(smem + L::ulaw))[i] = t1[i];
            ((float *)(smem + L::logit))[i] = t2[i];
        }
        const auto *ab1 = as_global(Ap->a_bias1), *adg = as_global(Ap->a_diag);
        for (int i = tid; i < RA; i += LPCN_WG_THREADS) {
            ((float *)(smem + L::abias))[2 * i] = ab1[i];          // [row]{bias, diag}: one 8-byte read per row
            ((float *)(smem + L::abias))[2 * i + 1] = adg[i];
        }
        const auto *br = as_global(Ap->b_rec), *bb = as_global(Ap->b_bias);
        for (int i = tid; i < (I8 ? RB * 4 : NB * RB); i += LPCN_WG_THREADS) ((uint32_t *)(smem + L::brec))[i] = ((const LPCN_GLOBAL uint32_t *)br)[i];   // bit copy (dwords of 4 int8 for I8)
        for (int i = tid; i < 2 * RB; i += LPCN_WG_THREADS) ((float *)(smem + L::bbias))[i] = bb[i];
        if (tid < 7) ((int *)(smem + L::bstart))[tid] = as_global(Ap->b_start)[tid];
        const auto *bk = as_global(Ap->b_blk);
        for (int i = tid; i < 608; i += LPCN_WG_THREADS) {
            const int pblk = i < nb_b ? bk[i] : 0;
            smem[L::bblk + i] = (unsigned char)pblk;
            ((unsigned short *)(smem + L::boff))[i] = (unsigned short)(I8 ? pblk * 4 : L::ha_off(pblk));
        }
        const auto *bw = as_global(Ap->b_w);
        constexpr int BW_DW = I8 ? 8 : 32;                  // dwords per GRU-B block
        for (int i = tid; i < (nb_b + (I8 ? 28 : 8)) * BW_DW; i += LPCN_WG_THREADS) {
            int di = i;
            if constexpr (FAST && !I8 && S >= 2) {
                // the matrix-pipe GRU-B (dense input matrix) reads 4 rows x 16 consecutive blocks per instruction: [row quad][block][row][4]
                // makes that one contiguous KB (the common [block][8 rows][4] order costs a two-way bank conflict there)
                if (b_dense && i < nb_b * BW_DW) {
                    const int blk_abs = i >> 5, ri = (i >> 2) & 7, c = i & 3;
                    di = ((((blk_abs / 96) * 2 + (ri >> 2)) * 96 + blk_abs % 96) * 4 + (ri & 3)) * 4 + c;
                }
            }
            if constexpr (I8) {
                // (int8 blobs, dense matrix: a row group is 24 quads of 4 blocks = 3 072 B -- the same bank phase for all six; same cure)
                if (b_dense) {
                    if (i < nb_b * BW_DW) di = i + ((0x321100 >> (4 * ((i >> 5) / 24))) & 15) * 32;
                    else if (i < (nb_b + 4) * BW_DW) di = i + 3 * 32;
                    else continue;
                }
            }
            if constexpr (!FAST && !I8) {
                // gb_lds: a wave's weight read fetches 16 B per row from six row groups 12 288 B apart -- the same 32 banks for
                // every group, a two-way conflict inside each 16-lane service group of ds_read_b128.  Row groups 2, 3 and 5 are
                // shifted by one more block (128 B = the other half of the banks); the pad blocks behind the matrix absorb it.
                if (gb_lds) {
                    if (i < nb_b * BW_DW) di = i + ((0x321100 >> (4 * ((i >> 5) / 96))) & 15) * 32;
                    else if (i < (nb_b + 5) * BW_DW) di = i + 3 * 32;     // zero blocks behind the shifted last group
                    else continue;
                }
            }
            ((uint32_t *)(smem + L::bw))[di] = i < nb_b * BW_DW ? ((const LPCN_GLOBAL uint32_t *)bw)[i] : 0u;
        }
        for (int i = tid; i < S * NA; i += LPCN_WG_THREADS) {
            const int s = i / NA, n = i % NA;
            const float hv0 = states[stream_of(s)].gru_a[n];
            sm_hT[n * S + s] = hv0;
            if constexpr (I8) {
                const unsigned char q = (unsigned char)quant_s8(hv0);
                smem[L::xq + ((n >> 2) * S + s) * 4 + (n & 3)] = q;
                smem[L::xqT + (s * 96 + (n >> 2)) * 4 + (n & 3)] = q;
            } else {
                *(float *)(smem + L::hA + L::ha_off(n >> 2) + s * 16 + (n & 3) * 4) = hv0;
            }
        }
        for (int i = tid; i < S * NB; i += LPCN_WG_THREADS) {
            const float hv0 = states[stream_of(i / NB)].gru_b[i % NB];
            sm_hB[i] = hv0;
            if constexpr (I8) smem[L::hBq + i] = (unsigned char)quant_s8(hv0);
            if constexpr (FAST) ((_Float16 *)(smem + L::hBh(Ap->nb_b, I8)))[i] = (_Float16)hv0;
        }
        // leader-lane state (lane s of wave 0 leads stream s); kept in LDS between samples
        if (tid < S) {
            const auto *st = &states[stream_of(tid)];
            sm_idx[tid] = 0;
            sm_lead[tid * 8 + 0] = 0.f;                                       // pred
            sm_lead[tid * 8 + 1] = st->deemph_mem;
            ((int *)sm_lead)[tid * 8 + 2] = st->last_exc;
#pragma unroll
            for (int j = 0; j < 4; ++j) ((uint32_t *)sm_lead)[tid * 8 + 4 + j] = st->rng[j];
        }
    }
    __syncthreads();

    // Opening a sample has two independent halves that run on different waves:
    //  * wave 0, lane s ("leader" of stream s): LPC prediction and the three mu-law indices of the
    //    embedding gather (src/lpcnet.c:252-254), published through sm_idx + sm_flag;
    //  * wave 1, lane s: the two KISS99 words that become the 8 tree thresholds (src/nnet.c:178-184).
    // LPC predictor state of wave 0: lane 16*s + j holds sample j of stream s's history (j = 0 newest,
    // src/lpcnet.c:252-263) and, per frame, LPC coefficient j.
    // (lrow/tap are recomputed from the thread id and the coefficient re-read from LDS where needed: every
    // VGPR that stays live across the GRU-A item loop costs the fp32 engine a spill)
#define LPCN_LROW ((tid0 & 63) >> 4)
#define LPCN_TAP (tid0 & 15)
    float hist = (tid0 < 16 * S) ? states[stream_of(LPCN_LROW)].last_sig[LPCN_TAP] : 0.f;
    auto row_shr1 = [](float v, float fill) {               // value of the previous lane of the 16-lane row; lane 0 gets `fill`
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill), __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, false));
    };
    // `prod` = this lane's term s_j*a_j of the prediction (tap j); `per_frame`: also (re)write the stream's live flag
    auto open_sample = [&](const bool live, const float newest, const float prod, const int exc, const bool per_frame) {   // wave 0, lanes < 16*S
        int t_ = tid0;
        LPCN_REMAT_V(t_);
        const int lrow = (t_ & 63) >> 4, tap = t_ & 15;
        // pred = ((0 - s0*a0) - s1*a1) - ... in tap order (src/lpcnet.c:252): every lane of the row runs the whole chain,
        // taking product j from lane j of its row through a DPP row broadcast folded into the subtract -- the
        // broadcast operand does not depend on the chain, so the 16 steps cost only the add latency
        const float r = lpc_chain<0>(0.f, prod);
        // mu-law index of the newest sample (even taps) and of the prediction (odd taps) in one pass; tap 0 collects both
        const int u = lpcn_lin2ulaw((tap & 1) ? r : newest);
        const int u_pred = __builtin_amdgcn_mov_dpp(u, 0xB1, 0xf, 0xf, true);      // neighbour lane (quad_perm [1,0,3,2])
        if (tap == 0) {
            if (live) {
                sm_lead[lrow * 8 + 0] = r;
                sm_idx[lrow] = u | (u_pred << 8) | (exc << 16);     // (sig, pred, exc) indices packed into one word per stream
            } else {
                sm_idx[lrow] = 0;
            }
            if (per_frame) sm_idx[S + lrow] = live ? 1 : 0;
        }
    };
    auto draw_thresholds = [&](const int ls) {
        int *li = (int *)sm_lead + ls * 8;
        uint32_t rng[4] = {(uint32_t)li[4], (uint32_t)li[5], (uint32_t)li[6], (uint32_t)li[7]};
        const uint32_t r0 = lpcn_kiss99(rng), r1 = lpcn_kiss99(rng);
        li[4] = (int)rng[0]; li[5] = (int)rng[1]; li[6] = (int)rng[2]; li[7] = (int)rng[3];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            sm_thr[ls * 8 + b] = sm_logit[(r0 >> (8 * b)) & 0xFF];
            sm_thr[ls * 8 + 4 + b] = sm_logit[(r1 >> (8 * b)) & 0xFF];
        }
    };
    // The sample loop has no barrier between the leader's work and the next sample's GRU-A: the
    // other waves run ahead into the rows that need no gathered input and pick the new indices
    // up through this flag (LDS operations of one wave complete in order).
    int seq = 0;                                             // samples opened so far (identical in every wave)
    // Waves that do not run GRU-B (wave >= S) and whose last slot holds only candidate rows compute that
    // slot for the NEXT sample while GRU-B runs: those rows start from bias + diag*h, which is final
    // once the gate stage is done, and they make up most of GRU-A's blocks.  (model_pack.c puts the
    // candidate-only slot of waves 4..7 last, so the rest of the sample simply ends at item bound[2].)
    bool head_ready = false;                                 // wave-uniform: the parked partial sums belong to the sample about to start
    const uint32_t flag_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)(smem + L::flag);
    auto publish_indices = [&]() {                           // after the sm_idx writes of the same lane
        asm volatile("ds_write_b32 %0, %1" :: "v"(flag_addr), "v"(seq) : "memory");
    };
    auto wait_indices = [&]() {
        int v;
        do {
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(flag_addr) : "memory");
            v = __builtin_amdgcn_readfirstlane(v);
            if (v != seq) __builtin_amdgcn_s_sleep(1);
        } while (v != seq);
    };
    // Arrival counter in LDS for hand-offs without a workgroup barrier (FAST float GRU-B on the matrix pipe: every wave bumps it behind its LDS stores -- a wave's
    // LDS operations complete in order -- and only the gate waves wait for all eight)
    int gbseq = 0;                                           // samples handed off so far (identical in every wave)
    const uint32_t arrive_addr = flag_addr + 4;
    auto lds_arrive = [&]() {                                 // the same counter for hand-offs through LDS: no store round trip to wait for
        int one = 1;
        unsigned long long ex;
        asm volatile("s_mov_b64 %0, exec\n\t"
                     "s_mov_b64 exec, 1\n\t"
                     "ds_add_u32 %1, %2\n\t"
                     "s_mov_b64 exec, %0"
                     : "=&s"(ex) : "v"(arrive_addr), "v"(one) : "memory");
    };
    auto mirror_wait = [&]() {
        int v;
        const int want = gbseq * LPCN_WAVES;
        do {
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(arrive_addr) : "memory");
            v = __builtin_amdgcn_readfirstlane(v);
            if (v != want) __builtin_amdgcn_s_sleep(1);
        } while (v != want);
    };
    // products hand-off (gb_prod): the three producer waves bump a second counter once their blocks are written (a wave's LDS
    // instructions complete in order: the add is behind the stores)
    int prseq = 0;
    const uint32_t prod_cnt_addr = flag_addr + 8;
    auto prod_arrive = [&]() {
        int one = 1;
        unsigned long long ex;
        asm volatile("s_mov_b64 %0, exec\n\t"
                     "s_mov_b64 exec, 1\n\t"
                     "ds_add_u32 %1, %2\n\t"
                     "s_mov_b64 exec, %0"
                     : "=&s"(ex) : "v"(prod_cnt_addr), "v"(one) : "memory");
    };
    auto prod_wait = [&]() {
        int v;
        const int want = prseq * 3;                          // (producers: waves 1..3 of a single-stream workgroup)
        do {
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(prod_cnt_addr) : "memory");
            v = __builtin_amdgcn_readfirstlane(v);
            if (v != want) __builtin_amdgcn_s_sleep(1);
        } while (v != want);
    };
    if (tid0 == 0) { *(int *)(smem + L::flag) = 0; *(int *)(smem + L::flag + 4) = 0; *(int *)(smem + L::flag + 8) = 0; }

#if LPCN_ENABLE_PROF      // per-phase shader-clock accounting (profiling builds only: it costs VGPRs)
    unsigned long long *const prof = Ap->prof;
    unsigned long long pt[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
    const bool profiling = prof != nullptr && blockIdx.x == 0;      // wave-uniform: the counters stay in SGPRs
#ifndef LPCN_PROF_MASK
#define LPCN_PROF_MASK 0x7FF        // which of the 12 slots are compiled in (every live counter costs two SGPRs)
#endif
#define LPCN_PROF(slot) do { if (((LPCN_PROF_MASK >> (slot)) & 1) && profiling) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); pt[slot] += now_ - tprev; tprev = now_; } } while (0)
#else
#define LPCN_PROF(slot) do { } while (0)
#endif
    // ====================================================================== frame loop ======
    for (int f = 0; f < n_frames; ++f) {
        // ---- frame-rate inputs -> LDS
        bool live = false;
        {
            const int tid = tid0;
            const auto *ca = as_global(Ap->cond_a), *cb = as_global(Ap->cond_b), *lp = as_global(Ap->lpc);
            for (int i = tid; i < S * RA; i += LPCN_WG_THREADS) {
                const int s = i / RA, r = i % RA;
                sm_cond[r * S + s] = ca[((size_t)stream_of(s) * nf + f) * RA + r];
            }
            if (tid < S * RB) sm_condb[tid] = cb[((size_t)stream_of(tid / RB) * nf + f) * RB + tid % RB];
            if (tid < S * LPCN_LPC_ORDER)
                sm_lpc[tid] = lp[((size_t)stream_of(tid / LPCN_LPC_ORDER) * nf + f) * LPCN_LPC_ORDER + tid % LPCN_LPC_ORDER];
            if (tid < 16 * S || (tid >= 64 && tid < 64 + S)) {    // wave 0: predictor lanes; wave 1: threshold lanes
                const int lstream = stream_of(tid < 64 ? LPCN_LROW : tid - 64);
                const int fc_ref = Ap->fc_base ? as_global(Ap->fc_base)[lstream] : states[lstream].frame_count;
                int fc = Ap->fc_advance ? fc_ref + f + 1 : fc_ref;
                if (fc > 1000) fc = 1000;
                live = fc > LPCN_FEATURES_DELAY;                 // src/lpcnet.c:239-243
            }
            if (preload > 0 && tid < S) {                         // teacher forcing reads the caller's samples
                const auto *pin = as_global(Ap->pcm) + (size_t)stream_of(tid) * (size_t)Ap->pcm_stride + (size_t)f * LPCN_FRAME_SIZE;
                for (int i = 0; i < preload; ++i) sm_pcm[tid * LPCN_FRAME_SIZE + i] = pin[i];
            }
        }
        __syncthreads();        // sm_lpc visible to the leaders
        ++seq;
        if (tid0 < 16 * S) {
            open_sample(live, hist, hist * sm_lpc[tid0], ((const int *)sm_lead)[LPCN_LROW * 8 + 2], true);   // sm_lpc is [stream][16] = [lane] for wave 0
            publish_indices();
        }
        if (tid0 >= 64 && tid0 < 64 + S && live) draw_thresholds(tid0 - 64);
        __syncthreads();
        int live_mask = 0;                                   // bit s: stream s produces samples in this frame
#pragma unroll
        for (int s = 0; s < S; ++s) live_mask |= (sm_idx[S + s] ? 1 : 0) << s;
        live_mask = __builtin_amdgcn_readfirstlane(live_mask);
        const int any_live = live_mask;
        if (!any_live) {                                     // start-up frames: zeros, no state change
            for (int i = tid0; i < S * LPCN_FRAME_SIZE; i += LPCN_WG_THREADS) sm_pcm[i] = 0;
            __syncthreads();
        }

        // ================================================================== sample loop ====
#if LPCN_ENABLE_PROF
        if (profiling) tprev = __builtin_amdgcn_s_memtime();
#endif
        for (int smp = 0; any_live && smp < frame_len; ++smp) {
            // ---------------------------------------------------------------- P1: GRU-A ----
            // ---- embedding gather: one 16-byte row per table and stream holds the entries of this
            // lane's three rows.  Register pressure is the binding constraint of this kernel (the
            // weights own 4*NW VGPRs), so the gathered values are consumed slot by slot.
            // (zero-initialised: otherwise the compiler carries the arrays around the sample loop as live values)
            float ge[3][3][S] = {};                          // sig / pred / exc entries (third set: light waves only)
            int gi[S] = {};                                  // this sample's mu-law indices, packed (scalar registers)
            auto load_indices = [&]() {                      // one LDS read for all streams, then the loads go back to back
                typename XVec<S>::type v = *(const typename XVec<S>::type *)sm_idx;
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    if constexpr (S == 1) gi[s] = __builtin_amdgcn_readfirstlane(v);
                    else gi[s] = __builtin_amdgcn_readfirstlane(v[s]);
                }
            };
            auto gather = [&](const int k, const int set) {
                if (!((has_slot >> k) & 1)) return;          // this wave owns no rows in slot k: nothing to fetch (ge stays 0)
                uint32_t toff = (uint32_t)tid0 * 4u;
                LPCN_REMAT_V(toff);                          // keep the lane offset a 32-bit VGPR (no hoisted 64-bit per-lane table bases)
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    // the mu-law indices are workgroup-uniform: scalar row base + lane offset
                    // scalar row base (table + level*6 KB + slot*2 KB) + 32-bit lane offset: one SGPR-pair add per load
                    const uint32_t i0 = (uint32_t)(gi[s] & 0xFF), i1 = (uint32_t)((gi[s] >> 8) & 0xFF), i2 = (uint32_t)(gi[s] >> 16) & 0xFFu;
                    const LPCN_GLOBAL char *r0 = (const LPCN_GLOBAL char *)emb_sig + (size_t)((i0 * LPCN_MAX_SLOTS + k) * (LPCN_WG_THREADS * 4u));
                    const LPCN_GLOBAL char *r1 = (const LPCN_GLOBAL char *)emb_pred + (size_t)((i1 * LPCN_MAX_SLOTS + k) * (LPCN_WG_THREADS * 4u));
                    const LPCN_GLOBAL char *r2 = (const LPCN_GLOBAL char *)emb_exc + (size_t)((i2 * LPCN_MAX_SLOTS + k) * (LPCN_WG_THREADS * 4u));
                    // (opaque scalar pointers: otherwise LLVM re-associates to per-lane 64-bit VGPR addresses)
                    asm volatile("" : "+s"(r0), "+s"(r1), "+s"(r2));
                    ge[set][0][s] = *(const LPCN_GLOBAL float *)(r0 + toff);
                    ge[set][1][s] = *(const LPCN_GLOBAL float *)(r1 + toff);
                    ge[set][2][s] = *(const LPCN_GLOBAL float *)(r2 + toff);
                }
            };
            LPCN_PROF(5);
            float acc[S] = {};                               // (initialised for the same reason as ge)
            // state blocks are fetched PF items ahead of their use
            constexpr int PF = 2;                            // (1 / 2 / 3 items ahead: 124.4 / 124.8 / 125.1 M in round 4 -- noise)
            HT hq[PF + 1] = {};
            auto fetch_o = [&](const int j) {                // streamed item j's block index -> its ring slot (one byte per lane, from L2)
                if constexpr (NR < NW) {
                    uint32_t bo = (uint32_t)tid0;
                    LPCN_REMAT_V(bo);
                    bo = (bo >> 6) * (uint32_t)(NW * 64) + (bo & 63u) + (uint32_t)j * 64u;
                    wso[(j - NR) % WSOD] = ab_s[bo];
                }
            };
            auto fetch_h = [&](const int j) {
                uint32_t off;
                if (NR < NW && j >= NR) {                    // (float blobs only: int8 items are all resident)
                    uint32_t pb = (uint32_t)wso[(j >= NR ? j - NR : 0) % WSOD], ls = (uint32_t)tid0;
                    LPCN_REMAT_V(ls);
                    ls = (S >= 4 ? (ls & 3u) : (S == 2 ? (ls & 1u) : 0u)) * 16u;
                    off = pb * (uint32_t)L::HA_STRIDE + (pb >> 2) * 16u + ls;
                } else if constexpr (OFFB) {
                    uint32_t pk = offp[j >> 2];
                    LPCN_REMAT_V(pk);
                    const uint32_t pb = (pk >> (8 * (j & 3))) & 0xFFu;
                    if constexpr (I8M) { uint32_t ls = (uint32_t)tid0; LPCN_REMAT_V(ls); off = pb * (4u * S) + (ls & (uint32_t)(S - 1)) * 4u; }
                    else off = pb * (4u * S);
                } else {
                    uint32_t pk = offp[(j < NR ? j : 0) >> 1];
                    LPCN_REMAT_V(pk);                        // keep the unpack inside the sample loop
                    off = (j & 1) ? (pk >> 16) : (pk & 0xFFFFu);
                }
                hq[j % (PF + 1)] = *(const HT *)(smem + L::hA + off);
            };
            auto fetch_w = [&](const int j) {                // streamed item j -> its ring slot (one global_load_dwordx4 per lane: 1 KB per wave, from L2)
                if constexpr (NR < NW) {
                    uint32_t wo = (uint32_t)tid0;
                    LPCN_REMAT_V(wo);                        // (the lane's byte offset is rebuilt from the thread id: no 64-bit per-lane pointer kept across the loop)
                    wo = (wo >> 6) * (uint32_t)(NW * 1024) + (wo & 63u) * 16u + (uint32_t)j * 1024u;
                    const LPCN_GLOBAL float *v = (const LPCN_GLOBAL float *)(aw_s + wo);      // ONE scalar base + a 32-bit lane offset (a scalar base per item cost 60 spilled SGPRs at 48 items)
                    if constexpr (!I8) ws[(j - NR) % WSD] = make_float4(v[0], v[1], v[2], v[3]);
                }
            };
            auto wsel = [&](const int j) -> WT {             // item j's weights: resident or from the ring (j is a constant after unrolling)
                if constexpr (NR < NW) return j < NR ? w[j < NR ? j : 0] : ws[(j >= NR ? j - NR : 0) % WSD];
                else return w[j];
            };
            // the matrix pipe's addend (PARITY float items, S >= 2): four registers of -0.0, re-materialised where an item chain starts so
            // that they are not live across the other phases
            typedef float negz_t __attribute__((ext_vector_type(4)));
            negz_t negz = {-0.f, -0.f, -0.f, -0.f};
            auto load_negz = [&]() { negz = (negz_t){-0.f, -0.f, -0.f, -0.f}; asm volatile("" : "+v"(negz)); };
            if constexpr (!I8 && !FAST && S >= 2) load_negz();
            // one item = (this lane's row) x (one 4-wide input block) for all S streams
            auto mac = [&](const int j) {
                // one item = (this lane's row) x (one 4-wide input block) for all S streams; per output
                // the products are added in block order, columns 0..3 (src/vec.h:355-401)
                if constexpr (I8 && FAST) {
                    // int32 accumulation across the row's blocks (exact; src/vec_avx.h:823-845)
                    const HT xv = hq[j % (PF + 1)];
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        int xs;
                        if constexpr (S == 1) xs = xv; else xs = xv[s];
                        acc[s] = __builtin_bit_cast(float, __builtin_amdgcn_sdot4(w[j], xs, __builtin_bit_cast(int, acc[s]), false));
                    }
                } else if constexpr (I8M) {
                    // register k of the result = (stream k's block) . (this lane's row) as an exact int32, for the quad's streams at once
                    typedef int i4v __attribute__((ext_vector_type(4)));
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    const i4v zero = {0, 0, 0, 0};
                    const i4v d = __builtin_amdgcn_mfma_i32_4x4x4i8(hq[j % (PF + 1)], w[j], zero, 0, 0, 0);
                    f2 a01 = {acc[0], acc[1]};
                    a01 = a01 + (f2){(float)d[0], (float)d[1]};
                    acc[0] = a01[0]; acc[1] = a01[1];
                    if constexpr (S == 4) {
                        f2 a23 = {acc[2], acc[3]};
                        a23 = a23 + (f2){(float)d[2], (float)d[3]};
                        acc[2] = a23[0]; acc[3] = a23[1];
                    }
                } else if constexpr (I8) {
                    // one dot4 = the row's block product for one stream, exact in int32 (src/vec.h:329-334)
                    static_assert(S == 1, "int8 PARITY items of two or four streams run on the matrix pipe (I8M)");
                    acc[0] = acc[0] + dot4_cvt(w[j], hq[j % (PF + 1)]);
                } else {
                const float4 hv = hq[j % (PF + 1)];
                const float hk[4] = {hv.x, hv.y, hv.z, hv.w};
                const float wk[4] = {wsel(j).x, wsel(j).y, wsel(j).z, wsel(j).w};
                if constexpr (FAST && S >= 2) {
                    // FAST: the quad broadcast as a 4x4 outer product on the matrix pipe.  v_mfma_f32_4x4x1_16B_f32 gives lane j
                    // of a quad, in register k, A(lane k) * B(lane j) + C: with A = this lane's state value (lane k fetched
                    // stream k's block), B = this lane's weight and C = the running sums, ONE instruction does the fused
                    // multiply-adds of a column for all four streams (tools/ubench/mfma4.hip: 61 clk per item against
                    // 155 for 16 DPP multiplies + 16 adds).  With C = -0.0 the result is the separately rounded product
                    // bit for bit, so PARITY could use it as a multiplier too -- measured: no faster in this kernel (two
                    // waves per SIMD contend for the matrix pipe, and the item loop is bound by its LDS read and control).
                    typedef float f4 __attribute__((ext_vector_type(4)));
                    f4 av;
                    av[0] = acc[0]; av[1] = acc[1]; av[2] = S == 4 ? acc[2] : 0.f; av[3] = S == 4 ? acc[3] : 0.f;
#pragma unroll
                    for (int c = 0; c < 4; ++c) av = __builtin_amdgcn_mfma_f32_4x4x1f32(hk[c], wk[c], av, 0, 0, 0);
                    acc[0] = av[0]; acc[1] = av[1];
                    if constexpr (S == 4) { acc[2] = av[2]; acc[3] = av[3]; }
                }
                if constexpr (!FAST && S >= 2) {
                    // the products of a column for all streams of the quad from ONE matrix-pipe instruction (C = -0.0: bit for bit the
                    // separately rounded product), the sums as v_pk_add_f32 over stream pairs -- each half of a packed add is
                    // rounded on its own, so the order of every row's sum is still the reference's: 4 MFMA + 8 packed adds per
                    // item instead of 16 DPP multiplies + 16 adds (S = 2: lanes 2, 3 of a quad repeat streams 0, 1; 4 packed adds)
                    typedef float f4 __attribute__((ext_vector_type(4)));
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    f2 a01 = {acc[0], acc[1]}, a23 = {0.f, 0.f};
                    if constexpr (S == 4) a23 = (f2){acc[2], acc[3]};
                    // (the four products of an item are issued back to back into four result tuples: hipcc otherwise re-uses one
                    // tuple and every column waits out the matrix pipe's latency behind s_nop)
                    f4 pv[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) pv[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(hk[c], wk[c], negz, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        a01 = a01 + __builtin_shufflevector(pv[c], pv[c], 0, 1);
                        if constexpr (S == 4) a23 = a23 + __builtin_shufflevector(pv[c], pv[c], 2, 3);
                    }
                    acc[0] = a01[0]; acc[1] = a01[1];
                    if constexpr (S == 4) { acc[2] = a23[0]; acc[3] = a23[1]; }
                } else
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if constexpr (FAST) {
                        if constexpr (S == 1) acc[0] = __builtin_fmaf(wk[c], hk[c], acc[0]);
                        // (S >= 2: all four columns at once below, on the matrix pipe)
                    } else {
                        static_assert(FAST || S == 1, "PARITY float items of two or four streams run on the matrix pipe");
                        acc[0] = acc[0] + wk[c] * hk[c];
                    }
                }
                }
                        };
            // start value of row slot k: bias + diag*h (+ gathered input for the update/reset rows),
            // gather sum in the reference's order ((cond + sig) + pred) + exc (src/nnet.c:431-440,
            // :487-489).  Candidate rows park the gathered input in sm_inh for the gate stage.  The
            // value is parked in the row's own sm_pre cell until the row becomes the running one.
            // The gather-independent half of a start value (bias + diag*h and the frame condition) is read and formed
            // for all three slots before the wave waits for the indices / the gathered rows.
            float pre_b[3][S] = {}, pre_c[3][S] = {};
            auto row_pre = [&](const int k, const int slot) {
                int r = LPCN_ROW(k);
                LPCN_REMAT_V(r);
                r = r < 0 ? 0 : r;
                const int n = r >= 2 * NA ? r - 2 * NA : (r >= NA ? r - NA : r);
                const float bias = sm_abias[2 * r], diag = sm_abias[2 * r + 1];
                const bool parked = k == 0 && early_wave;      // slot 0 continues from the partial sums of its early head
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    pre_b[slot][s] = parked ? sm_pre[r * S + s] : bias + diag * sm_hT[n * S + s];
                    pre_c[slot][s] = sm_cond[r * S + s];
                }
            };
            auto row_init = [&](const int k, const int set, const bool to_acc, const bool park = true) {
                const int slot = k;
                int r = LPCN_ROW(k);
                LPCN_REMAT_V(r);
                const bool live_row = r >= 0;
                r = r < 0 ? 0 : r;
                const int n = r >= 2 * NA ? r - 2 * NA : (r >= NA ? r - NA : r);
                const bool candidate = r >= 2 * NA;
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const float b = pre_b[slot][s];
                    const float g = ((pre_c[slot][s] + ge[set][0][s]) + ge[set][1][s]) + ge[set][2][s];
                    if (candidate && live_row) sm_inh[n * S + s] = g;
                    float v = candidate ? b : b + g;
                    if (!(k == 0 && early_wave)) v = acc_start<I8, FAST>(v);       // (a parked partial sum already is an accumulator)
                    if (to_acc) acc[s] = v; else if (live_row && park) sm_pre[r * S + s] = v;
                }
            };
            auto row_swap = [&](const int k_done, const int k_next, const bool store_done = true) {   // finished row out, next row in
                int r = LPCN_ROW(k_done), r2 = LPCN_ROW(k_next);
                LPCN_REMAT_V(r);
                LPCN_REMAT_V(r2);
                if (r >= 0 && store_done) {
#pragma unroll
                    for (int s = 0; s < S; ++s) sm_pre[r * S + s] = acc_final<I8, FAST>(acc[s]);
                }
                r2 = r2 < 0 ? 0 : r2;
#pragma unroll
                for (int s = 0; s < S; ++s) acc[s] = sm_pre[r2 * S + s];
            };
            auto row_store = [&](const int k) {
                int r = LPCN_ROW(k);
                LPCN_REMAT_V(r);
                if (r >= 0) {
#pragma unroll
                    for (int s = 0; s < S; ++s) sm_pre[r * S + s] = acc_final<I8, FAST>(acc[s]);
                }
            };
            // Waves whose first slot holds only candidate rows (the big ones) start it from
            // bias + diag*h alone: they need neither the new sample's indices nor the gathered rows
            // for that, so they run ahead while wave 0 still finishes the previous sample.  They
            // poll for the indices at item JG, issue the gathers, and resolve the gather-dependent
            // start values at item JSTAR (<= their first slot boundary).  The other waves wait for
            // the indices and resolve everything up front.  jmode is wave-uniform.
            // JSTAR is the largest of 10 / 14 / 18 that does not exceed the first slot's length (wave-uniform, fixed per launch:
            // a handful of possible positions keeps the resolve code from being inlined at every item of the chain)
            constexpr int JG = 8, JSTAR_MAX = 18, JSTAR_MIN = I8 ? 18 : 10;   // (JG: the leader publishes ~2 k clk after the tree barrier = ~8 items)
            int jstar = b1 >= 18 ? 18 : (b1 >= 14 ? 14 : 10);
            LPCN_REMAT_S(jstar);
            // The early head of slot 0 (see `hl`): one compile-time chain over the last LPCN_EARLY_MAX items, entered at NW - hl.
            auto run_head = [&]() __attribute__((always_inline)) {
                if constexpr (!I8 && !FAST && S >= 2) load_negz();
                {
                    int r = LPCN_ROW(0);
                    LPCN_REMAT_V(r);
                    r = r < 0 ? 0 : r;
                    const int n = r - 2 * NA;
                    const float bias = sm_abias[2 * r], diag = sm_abias[2 * r + 1];
#pragma unroll
                    for (int s = 0; s < S; ++s) acc[s] = acc_start<I8, FAST>(bias + diag * sm_hT[n * S + s]);
                }
                int e0 = NW - hl;
                LPCN_REMAT_S(e0);
                // (every step issues the same LDS read whether its item runs or not: otherwise the compiler's s_waitcnt
                // bookkeeping degrades to full waits and the state prefetch no longer overlaps anything)
                constexpr int J0 = NW - LPCN_EARLY_MAX < 0 ? 0 : NW - LPCN_EARLY_MAX;
#pragma unroll
                for (int j = J0; j < J0 + WSOD - 1; ++j) if (NR < NW && j >= NR && j < NW) fetch_o(j);
#pragma unroll
                for (int j = 0; j < PF; ++j) if (J0 + j < NW) fetch_h(J0 + j);
#pragma unroll
                for (int j = J0; j < J0 + WSD - 1; ++j) if (NR < NW && j >= NR && j < NW) fetch_w(j);
                auto step = [&](auto self, auto jc) __attribute__((always_inline)) -> void {
                    constexpr int j = decltype(jc)::value;
                    if constexpr (j < NW) {
                        if constexpr (NR < NW && j + WSOD - 1 >= NR && j + WSOD - 1 < NW) fetch_o(j + WSOD - 1);
                        if constexpr (j + PF < NW) fetch_h(j + PF);
                        if constexpr (NR < NW && j + WSD - 1 >= NR && j + WSD - 1 < NW) fetch_w(j + WSD - 1);
                        if (j >= e0) { asm volatile(""); mac(j); }       // (the empty statement keeps this a scalar branch: if-converted, int8 items became
                                                                        // selects on 24 precomputed lane masks -- 48 SGPRs, most of them spilled)
                        self(self, std::integral_constant<int, j + 1>{});
                    }
                };
                step(step, std::integral_constant<int, J0>{});
                {                                            // park the partial sums as they are (no final scaling)
                    int r = LPCN_ROW(0);
                    LPCN_REMAT_V(r);
                    if (r >= 0) {
#pragma unroll
                        for (int s = 0; s < S; ++s) sm_pre[r * S + s] = acc[s];
                    }
                }
            };
            if (early_wave && !head_ready) { run_head(); head_ready = true; }     // first sample of the launch only
            const int jend = b3;
            const int jmode = __builtin_amdgcn_readfirstlane((allh0 && b1 >= JSTAR_MIN) ? 1 : 0);
            if (jmode == 0) {
                row_pre(0, 0); row_pre(1, 1); row_pre(2, 2);
                wait_indices();
                load_indices();
                gather(1, 0);
                gather(2, 1);
                gather(0, 2);
                row_init(1, 0, false);
                row_init(2, 1, false);
                row_init(0, 2, true);
            } else {
                int r = LPCN_ROW(0);
                LPCN_REMAT_V(r);
                r = r < 0 ? 0 : r;
                const int n = r - 2 * NA;
                const float bias = sm_abias[2 * r], diag = sm_abias[2 * r + 1];
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    acc[s] = early_wave ? sm_pre[r * S + s] : acc_start<I8, FAST>(bias + diag * sm_hT[n * S + s]);
                }
            }
            // issue priority for the part of GRU-A that everybody waits for: int8 -- the younger wave of each SIMD pair sees
            // its gather last; fp32 -- the wave that still has a whole candidate slot in front of it (jmode)
            if (I8 ? tid0 >= 256 : jmode != 0) __builtin_amdgcn_s_setprio(2);
            LPCN_PROF(10);     // gather issue + (light waves) wait + start values
#pragma unroll
            for (int j = 0; j < PF && j < NW; ++j) fetch_h(j);
            LPCN_REMAT_S(b1);
            LPCN_REMAT_S(b2);
            LPCN_REMAT_S(b3);
            // All tests below are wave-uniform scalar branches.  A taken branch costs ~35 clk of refetch, so the
            // common case (an ordinary item) must fall through every one of them: hence the expectations.
            int nextb = b1;
            LPCN_REMAT_S(nextb);
            auto item = [&](const int j) -> bool {           // false: this wave has no more items
                if ((j == 10 || j == 14 || j == 18) && j >= JSTAR_MIN && j <= JSTAR_MAX && __builtin_expect(j == jstar, 0)) {
                    if (jmode) {
                        row_init(1, 0, false);
                        if (has2) { row_init(2, 1, false); gather(0, 0); }     // (two-slot waves already hold slot 0's rows in set 1)
                        __builtin_amdgcn_s_waitcnt(0xC07F);
                    }
                }
                if (__builtin_expect(j >= jend, 0)) return false;
                if (NR < NW && j + WSOD - 1 >= NR && j + WSOD - 1 < NW) fetch_o(j + WSOD - 1);
                if (j + PF < NW) fetch_h(j + PF);
                if (NR < NW && j + WSD - 1 >= NR && j + WSD - 1 < NW) fetch_w(j + WSD - 1);
                // slot boundaries (a slot may be empty: b1 == b2, or b1 == 0): ONE scalar compare per item against the next one
                // (the rare block drains its own LDS traffic -- s_waitcnt lgkmcnt(0) -- so that the join with the
                // common path keeps its precise wait counts)
                if (__builtin_expect(j == nextb, 0)) {
                    if (j == b1) row_swap(0, 1);
                    if (j == b2) row_swap(1, 2);
                    __builtin_amdgcn_s_waitcnt(0xC07F);
                    nextb = b1 > j ? b1 : (b2 > j ? b2 : NW);
                }
                mac(j);
                return true;
            };
#pragma unroll
            for (int j = 0; j < JG && j < NW; ++j) item(j);  // (no wave ends before JG: jmode needs b1 >= JSTAR_MIN > JG, others just fall through)
            if (jmode) {                                     // between two fully unrolled halves
                row_pre(1, 1); row_pre(2, 2);
                wait_indices();
                load_indices();
                gather(1, 0);
                if (has2) gather(2, 1); else gather(0, 1);   // a third slot's rows are fetched when set 0 is free again (JSTAR)
            }
            // straight-line items JG..NW-1 with ONE exit branch (compile-time recursion instead of an unrolled
            // loop with a break, which the unroller refuses)
            auto run_items = [&](auto self, auto jc) __attribute__((always_inline)) -> void {
                constexpr int j = decltype(jc)::value;
                if constexpr (j < NW) {
                    if (!item(j)) return;
                    self(self, std::integral_constant<int, j + 1>{});
                }
            };
            run_items(run_items, std::integral_constant<int, JG>{});
            LPCN_PROF(9);      // item loop (incl. the mid-phase start values of the heavy waves)
            // close whichever slot is still open; slots that start exactly at NW have no items
            // (b1 <= b2 <= NW; items past a wave's last real item carry zero weights)
            if (b1 >= jend) {
                row_swap(0, 1);
                row_swap(1, 2);
                row_store(2);
            } else if (b2 >= jend) {
                row_swap(1, 2);
                row_store(2);
            } else {
                row_store(2);
            }
            if (jmode) {                                     // input part of the candidate rows of slot 0
                int r = LPCN_ROW(0);
                LPCN_REMAT_V(r);
                if (r >= 0) {
                    const int n = r - 2 * NA;
                    if (has2) {
#pragma unroll
                        for (int s = 0; s < S; ++s)
                            sm_inh[n * S + s] = ((sm_cond[r * S + s] + ge[0][0][s]) + ge[0][1][s]) + ge[0][2][s];
                    } else {
#pragma unroll
                        for (int s = 0; s < S; ++s)
                            sm_inh[n * S + s] = ((sm_cond[r * S + s] + ge[1][0][s]) + ge[1][1][s]) + ge[1][2][s];
                    }
                }
            }
            __builtin_amdgcn_s_setprio(0);
            LPCN_PROF(6);      // slots: begin + items + end
            __syncthreads();                                                   // B1
            LPCN_PROF(0);
            if (tracing) {                                                     // tests: recurrent pre-activations of stream 0
                LPCN_GLOBAL float *d = as_global_rw(Ap->dbg) + ((size_t)f * LPCN_FRAME_SIZE + smp) * LPCN_DBG_STRIDE + 448;
                int t_ = tid0;
                LPCN_REMAT_V(t_);                            // (otherwise the LDS address is hoisted out of the sample loop and spilled)
                for (int i = t_; i < RA; i += LPCN_WG_THREADS) d[i] = sm_pre[i * S];
            }

            int tid = tid0;
            LPCN_REMAT_V(tid);
            // ------------------------------------------------------------ P2: GRU-A gates --
            // One work item per (neuron, stream): update/reset sigmoids, candidate tanh and the blend
            // (src/nnet.c:441-447).  sm_pre / sm_inh / sm_hT are [row][stream], so item i = n*S + s is
            // simply element i of each gate's third: consecutive lanes touch consecutive words.
            // S = 2: 768 items on 512 lanes -- the second round is empty for waves 4..7; S = 1: 384 items -- waves 6, 7 have none.  A wave
            // runs only the rounds in which it has items (wave-uniform count, one scalar branch): the instructions of an empty round
            // are issue slots taken from the waves that share the SIMD (two workgroups per CU for the int8 kernels).
            // (tried in round 5: the activations of a lane's first two items as packed math on 2-vectors -- same operations, same order, 35-68 fewer
            // instructions in the loop, bit-exact, and 2-5 % slower: a packed fp32 instruction costs more here than the two scalar ones it replaces)
            auto gate_stage = [&](auto nqc) __attribute__((always_inline)) {
                constexpr int NI = NA * S;                                     // items
                constexpr int NQ = decltype(nqc)::value;                       // rounds of this wave
                constexpr bool FULL = NI % LPCN_WG_THREADS == 0;               // (S = 4: every lane has an item in every round -- no range tests, no selects)
                if constexpr (NQ > 0) {
                float z[NQ], rg[NQ], a[NQ], hold[NQ];
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int i = tid + q * LPCN_WG_THREADS;
                    const int ic = (FULL || i < NI) ? i : 0;
                    z[q] = sm_pre[ic];
                    rg[q] = sm_pre[NI + ic];
                    a[q] = sm_pre[2 * NI + ic];
                    hold[q] = sm_hT[ic];
                }
#pragma unroll
                for (int q = 0; q < NQ; ++q) { z[q] = act_sigmoid<FAST>(z[q], sm_tansig); rg[q] = act_sigmoid<FAST>(rg[q], sm_tansig); }
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int i = tid + q * LPCN_WG_THREADS;
                    a[q] = a[q] * rg[q] + sm_inh[(FULL || i < NI) ? i : 0];
                }
#pragma unroll
                for (int q = 0; q < NQ; ++q) a[q] = act_tanh<FAST>(a[q], sm_tansig);
                // item i = tid + 512 q is (neuron tid / S + (512 / S) q, stream tid % S): the stream is the lane's own in every round, and the block-ordered copy's
                // address advances by a constant per round (128 / S blocks and their pads)
                const unsigned ut = (unsigned)tid;
                const bool live_s = ((live_mask >> (ut % (unsigned)S)) & 1) != 0;
                constexpr int HA_ADV = L::ha_off(LPCN_WG_THREADS / 4 / S);
                unsigned char *const ha0 = smem + L::hA + L::ha_off((int)((ut / (unsigned)S) >> 2)) + (ut % (unsigned)S) * 16u + ((ut / (unsigned)S) & 3u) * 4u;
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int i = tid + q * LPCN_WG_THREADS;
                    const int n = i / S, s = i % S;
                    const float hnew = z[q] * hold[q] + (1.f - z[q]) * a[q];      // src/nnet.c:447
                    const float hv = live_s ? hnew : hold[q];
                    if (FULL || i < NI) {
                        sm_hT[i] = hv;
                        if constexpr (I8) {                  // next sample's activations, quantised once (src/vec.h:311)
                            const unsigned char qv = (unsigned char)quant_s8(hv);
                            smem[L::xq + ((n >> 2) * S + s) * 4 + (n & 3)] = qv;
                            smem[L::xqT + (s * 96 + (n >> 2)) * 4 + (n & 3)] = qv;
                        } else {
                            *(float *)(ha0 + q * HA_ADV) = hv;
                        }
                    }
                }
                }
            };
            {
                constexpr int NI = NA * S, NQ_MAX = (NI + LPCN_WG_THREADS - 1) / LPCN_WG_THREADS;
                // (every wave runs every round: skipping a wave's empty rounds at S = 1, 2 was measured in round 5 and is slower -- 156.8 vs 157.9 M int8;
                // the wave-divergent paths cost more at the barrier than the empty round's issue slots return)
                gate_stage(std::integral_constant<int, NQ_MAX>{});
            }
            __syncthreads();                                                   // B2
            LPCN_PROF(1);

            LPCN_REMAT_V(tid);
            const int lane = tid & 63;
            const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
            // prefetch this lane's dual-FC row (node = tid>>1, channel = tid&1); it lands while GRU-B
            // runs.  Re-fetched every sample: 18 VGPRs that must not stay live across the GRU-A loop.
            // (tried: the GRU-B waves issue these loads only after their mat-vec -- the ~0.1 k clk saved in front of GRU-B
            // come back as a later tree phase: 102.3 vs 102.2 M samples/s, not kept)
            const int node = tid >> 1, chan = tid & 1;
            const auto *fcw_ptr = fc_w_g + node * 2 * NB + chan * NB;
            float fcw[NB], fcb = 0.f, fcf = 0.f;
            bool fc_f16 = false;
            auto load_fc = [&]() __attribute__((always_inline)) {
                if constexpr (FAST) {
                    // fp16 sub-option: 8 dwords of fp16 pairs per row instead of 16 floats.  Both forms issue the same 16 loads from two
                    // row halves (the second half is a dead copy of the first for fp16): a branch here would merge the two register
                    // sets with copies, and the copies would wait for the loads -- in front of GRU-B.
                    fc_f16 = Ap->fc_f16 != 0;
                    const auto *lo = fc_f16 ? (const LPCN_GLOBAL float *)(as_global(Ap->fc_wh) + (node * 2 + chan) * (NB / 2)) : fcw_ptr;
                    const auto *hi = fc_f16 ? lo : fcw_ptr + NB / 2;
#pragma unroll
                    for (int j = 0; j < NB / 2; ++j) { fcw[j] = lo[j]; fcw[NB / 2 + j] = hi[j]; }
                } else {
#pragma unroll
                    for (int j = 0; j < NB; ++j) fcw[j] = fcw_ptr[j];
                }
                fcb = fc_b_g[chan * 256 + node]; fcf = fc_f_g[chan * 256 + node];
            };
            // (tried on the 128-VGPR FAST kernels: fetching the row only when the tree needs it -- 160 vs 168 M samples/s on int8)
            load_fc();
            LPCN_PROF(7);      // dual-FC prefetch issue
            // ----------------------------------------------------- P3: GRU-B (wave = stream)
            // FAST, int8 blobs, dense input matrix: integer block sums are exact in any order, so the 96 input blocks of a
            // stream are split over GB_W = 8/S waves (each reads 1/GB_W of the weights), the partial sums meet in LDS and
            // the first wave of each stream's group runs the gate stage.  (PARITY keeps one wave per stream: a row's 384-term
            // float sum has to stay in the reference's order.)
            constexpr int GB_W = LPCN_WAVES / S;
            bool gb_split = false;                           // wave-uniform
            if constexpr (I8 && FAST) {
                if (b_dense) {
                    typedef int i4 __attribute__((ext_vector_type(4)));
                    constexpr int QP = 24 / GB_W;            // quads of 4 blocks per wave: 3, 6 or 12
                    const int s2 = wave / GB_W, part = wave - s2 * GB_W;
                    const int r = lane < RB ? lane : RB - 1;
                    const i4 *wq = (const i4 *)(smem + L::bw) + (sm_bstart[r >> 3] >> 2) * 8 + (r & 7) + part * QP * 8 + ((0x321100 >> (4 * (r >> 3))) & 15) * 8;
                    const i4 *xq4 = (const i4 *)(smem + L::xqT + s2 * 384) + part * QP;
                    int z0 = 0, z1 = 0, z2 = 0, z3 = 0;
#pragma unroll
                    for (int q = 0; q < QP; ++q) {
                        const i4 w4 = wq[q * 8], x4 = xq4[q];
                        z0 = __builtin_amdgcn_sdot4(w4[0], x4[0], z0, false);
                        z1 = __builtin_amdgcn_sdot4(w4[1], x4[1], z1, false);
                        z2 = __builtin_amdgcn_sdot4(w4[2], x4[2], z2, false);
                        z3 = __builtin_amdgcn_sdot4(w4[3], x4[3], z3, false);
                    }
                    ((int *)sm_pre)[wave * 64 + lane] = (z0 + z1) + (z2 + z3);      // (sm_pre is idle in this phase: int8 blobs have no early slots)
                    __syncthreads();
                    gb_split = true;
                }
            }
            // FAST, float blobs, dense input matrix: wave s + 4 (it shares stream s's SIMD) takes the last quads of the 96 input
            // blocks once its early GRU-A slot is done; fused multiply-adds into four independent sums per part (the order of
            // a row's sum is free in FAST), the parts meet in LDS (sm_inh is idle in this phase).  The helper's early slot
            // (e items of ~80 clk on the matrix pipe) comes first, a quad costs ~290 clk here: both finish together with
            // gb_qa = (24 + e * 0.28) / 2 quads on the stream's own wave.
            // FAST, float blobs, dense input matrix, S >= 2: the input mat-vec of ALL streams of the workgroup is one small GEMM,
            // [48 rows x 384] x [384 x S], on the matrix pipe.  v_mfma_f32_4x4x1_16B: each of the 16 quads of a wave forms a
            // (4 rows) x (4 streams) outer product for ITS input column and accumulates it, so one instruction covers 16
            // columns; a unit = one row quad over all 384 columns = 6 weight reads + 6 state reads (16 bytes per lane, the
            // state straight out of the layout the other paths use) + 24 MFMAs in two chains.  The 12 units go to the waves
            // that have no candidate head to run (S = 4: three per gate wave).  The 16 quads' sums are folded with two DPP
            // row shifts, the remaining 4 lane rows meet in LDS (sm_inh is idle in this phase) and the stream's gate wave adds
            // them up.  No workgroup barrier: every wave bumps the arrival counter behind its LDS stores (LDS operations of a
            // wave complete in order) and goes on; only the gate waves wait for all eight.  ~1.2 k clk of GRU-B mat-vec
            // instead of ~6 k for one FMA chain per stream.
            const bool gb_mfma = FAST && !I8 && b_dense && S >= 2;
            if constexpr (FAST && !I8 && S >= 2) {
                if (b_dense) {                               // (workgroup-uniform)
                    typedef float f4 __attribute__((ext_vector_type(4)));
                    ++gbseq;
                    int n0, n1;                              // this wave's row quads
                    if constexpr (S == 4) { n0 = 3 * wave; n1 = wave < 4 ? n0 + 3 : n0; }
                    else { n0 = wave < 2 ? 2 * wave : 4 * wave - 4; n1 = wave < 2 ? n0 + 2 : (wave < 4 ? n0 + 4 : n0); }
                    if (n0 < n1) {                           // (wave-uniform)
                        const int quad = lane >> 2, l4 = lane & 3;
                        // lane (quad, j): stream j's values of input block 16 u + quad; lane (quad, i): row 4 rq + i of the same block
                        // in the swizzled weight image (see the LDS fill)
                        const unsigned char *hq0 = smem + L::hA + quad * L::HA_STRIDE + (quad >> 2) * 16 + l4 * 16;
                        const unsigned char *wq0 = smem + L::bw + (quad * 4 + l4) * 16;
                        for (int n = n0; n < n1; ++n) {
                            const unsigned char *wq = wq0 + n * (96 * 64);
                            f4 da = {0.f, 0.f, 0.f, 0.f}, db = {0.f, 0.f, 0.f, 0.f};     // two chains: a dependent MFMA waits for its predecessor
                            // reads two steps ahead of the matrix pipe (scheduling barriers: hipcc otherwise sinks every read to its use)
                            float4 wr[3], hr[3];
                            wr[0] = *(const float4 *)(wq); hr[0] = *(const float4 *)(hq0);
                            wr[1] = *(const float4 *)(wq + 1024); hr[1] = *(const float4 *)(hq0 + (16 * L::HA_STRIDE + 64));
#pragma unroll
                            for (int u = 0; u < 6; ++u) {
                                if (u + 2 < 6) {
                                    wr[(u + 2) % 3] = *(const float4 *)(wq + (u + 2) * 1024);
                                    hr[(u + 2) % 3] = *(const float4 *)(hq0 + (u + 2) * (16 * L::HA_STRIDE + 64));
                                }
                                __builtin_amdgcn_sched_barrier(0);
                                const float4 w4 = wr[u % 3], hv = hr[u % 3];
                                da = __builtin_amdgcn_mfma_f32_4x4x1f32(w4.x, hv.x, da, 0, 0, 0);
                                db = __builtin_amdgcn_mfma_f32_4x4x1f32(w4.y, hv.y, db, 0, 0, 0);
                                da = __builtin_amdgcn_mfma_f32_4x4x1f32(w4.z, hv.z, da, 0, 0, 0);
                                db = __builtin_amdgcn_mfma_f32_4x4x1f32(w4.w, hv.w, db, 0, 0, 0);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                            // register k of lane (quad, j) = row 4 rq + k x stream j, this quad's columns: fold the four quads of a 16-lane row
                            // (element-wise copies first: hipcc's __builtin_bit_cast of an ext-vector ELEMENT reads element 0 whatever the index)
                            float dk[4] = {da[0] + db[0], da[1] + db[1], da[2] + db[2], da[3] + db[3]};
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                dk[k] = dk[k] + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, dk[k]), 0x114, 0xf, 0xf, true));
                                dk[k] = dk[k] + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, dk[k]), 0x118, 0xf, 0xf, true));
                            }
                            if ((lane & 15) >= 12 && l4 < S) {   // [row][stream][lane row]: 4 partial sums side by side
                                float *o = sm_inh + ((n * 4) * S + l4) * 4 + (lane >> 4);
#pragma unroll
                                for (int k = 0; k < 4; ++k) o[k * S * 4] = dk[k];
                            }
                        }
                    }
                    lds_arrive();
                }
            }
            const bool gb_fsplit = FAST && !I8 && b_dense && S <= LPCN_WAVES / 2 && !gb_mfma;
            int gb_qa = 24;
            if constexpr (FAST && !I8) {
                const int hw = wave < LPCN_WAVES / 2 ? wave + LPCN_WAVES / 2 : wave;       // the helper wave of this pair
                const int e = hw >= S ? as_global(Ap->a_head)[hw] : 0;       // items of the helper's early head
                gb_qa = __builtin_amdgcn_readfirstlane((24 + (9 * e) / 32) / 2);
                gb_qa = gb_qa > 24 ? 24 : gb_qa;
            }
            auto gb_part_fast = [&](const int gs_, const int q_lo, const int q_hi) -> float {
                const int r_ = lane < RB ? lane : RB - 1;
                const int bbeg_ = sm_bstart[r_ >> 3];
                const unsigned char *wptr = smem + L::bw + ((bbeg_ + 4 * q_lo) * 8 + (r_ & 7)) * 16;
                const unsigned char *hbase = smem + L::hA + gs_ * 16;
                const unsigned short *offk = sm_boff + bbeg_ + (lane & 3);
                auto ldw = [&](int byte_off) { return *(const float4 *)(wptr + byte_off); };
                auto ldh = [&](int q) { return *(const float4 *)(hbase + offk[4 * q]); };
                float4 wr0 = ldw(0), wr1 = ldw(128), wr2 = ldw(256), wr3;
                float4 hq = ldh(q_lo), hn;
                float z0 = 0.f, z1 = 0.f, z2 = 0.f, z3 = 0.f;
                for (int q = q_lo; q < q_hi; ++q) {          // (reads one quad / three blocks past the end: valid, unused LDS data)
                    wr3 = ldw(384); hn = ldh(q + 1);
                    z0 = fmac_quad<0>(z0, wr0.x, hq.x); z1 = fmac_quad<0>(z1, wr0.y, hq.y); z2 = fmac_quad<0>(z2, wr0.z, hq.z); z3 = fmac_quad<0>(z3, wr0.w, hq.w);
                    wr0 = ldw(512);
                    z0 = fmac_quad<1>(z0, wr1.x, hq.x); z1 = fmac_quad<1>(z1, wr1.y, hq.y); z2 = fmac_quad<1>(z2, wr1.z, hq.z); z3 = fmac_quad<1>(z3, wr1.w, hq.w);
                    wr1 = ldw(640);
                    z0 = fmac_quad<2>(z0, wr2.x, hq.x); z1 = fmac_quad<2>(z1, wr2.y, hq.y); z2 = fmac_quad<2>(z2, wr2.z, hq.z); z3 = fmac_quad<2>(z3, wr2.w, hq.w);
                    wr2 = ldw(768);
                    z0 = fmac_quad<3>(z0, wr3.x, hq.x); z1 = fmac_quad<3>(z1, wr3.y, hq.y); z2 = fmac_quad<3>(z2, wr3.z, hq.z); z3 = fmac_quad<3>(z3, wr3.w, hq.w);
                    hq = hn;
                    wptr += 512;
                }
                return (z0 + z1) + (z2 + z3);
            };
            // int8 PARITY, S <= 2 (two workgroups per CU): GRU-B's chain waves are LPCN_I8_GBWA (stream 0) and LPCN_I8_GBWB (stream 1) instead of
            // waves 0 and 1 -- SIMD balance, see lpcnet_engine.h
            constexpr bool GBMOVE = I8 && !FAST && S <= 2;
            constexpr int GBWA = GBMOVE ? LPCN_I8_GBWA : 0, GBWB = GBMOVE ? LPCN_I8_GBWB : 1;
            const bool gate_wave = gb_split ? (wave % GB_W == 0) : (GBMOVE ? (wave == GBWA || (S == 2 && wave == GBWB)) : wave < S);      // wave-uniform
            float zrh = 0.f, rec = 0.f;
            const int s = gb_split ? wave / GB_W : (GBMOVE ? (wave == GBWA ? 0 : 1) : wave);     // (stream of a gate wave)
            const int r = lane < RB ? lane : RB - 1;
            if (gb_prod) ++prseq;
            if (gate_wave) {
                // the longest chain of the sample: win issue arbitration against the early GRU-A slot sharing the SIMD
                __builtin_amdgcn_s_setprio(3);                 // (priority 1 or none: the same step time, measured)
                const int g = r >> 3, ri = r & 7;
                zrh = sm_bbias[r] + sm_condb[s * RB + r];                     // src/nnet.c:351
                rec = sm_bbias[RB + r];
                if constexpr (I8 && FAST) {
                  if (gb_split) {
                    typedef int i4 __attribute__((ext_vector_type(4)));
                    int rsum = (int)__builtin_rintf(rec * QS), zsum = (int)__builtin_rintf(zrh * QS);     // src/vec_avx.h:703-705
                    const i4 wr = ((const i4 *)(smem + L::brec))[r];
                    const i4 hb = *(const i4 *)(smem + L::hBq + s * 16);
#pragma unroll
                    for (int k = 0; k < 4; ++k) rsum = __builtin_amdgcn_sdot4(wr[k], hb[k], rsum, false);
#pragma unroll
                    for (int k = 0; k < GB_W; ++k) zsum += ((const int *)sm_pre)[(s * GB_W + k) * 64 + lane];
                    rec = (float)rsum * QS1;
                    zrh = (float)zsum * QS1;
                    (void)g; (void)ri;
                  } else {
                    // int32 accumulation like the reference's AVX2 int8 build (src/vec_avx.h:690-750): start values rounded
                    // to the 1/(128*127) grid, exact block sums; integer addition is associative, so the four blocks of a
                    // 16-byte weight read feed four independent chains
                    typedef int i4 __attribute__((ext_vector_type(4)));
                    int rsum = (int)__builtin_rintf(rec * QS);
                    int z0 = (int)__builtin_rintf(zrh * QS), z1 = 0, z2 = 0, z3 = 0;
                    const i4 wr = ((const i4 *)(smem + L::brec))[r];
                    const i4 hb = *(const i4 *)(smem + L::hBq + s * 16);
#pragma unroll
                    for (int k = 0; k < 4; ++k) rsum = __builtin_amdgcn_sdot4(wr[k], hb[k], rsum, false);
                    rec = (float)rsum * QS1;
                    const int bbeg = sm_bstart[g], bend = sm_bstart[g + 1];
                    const int nq = (bend - bbeg) >> 2;
                    const i4 *wq = (const i4 *)(smem + L::bw) + (bbeg >> 2) * 8 + ri + (b_dense ? ((0x321100 >> (4 * g)) & 15) * 8 : 0);
                    const unsigned char *xb = smem + L::xqT + s * 384;
                    if (b_dense) {
                        const i4 *xq4 = (const i4 *)xb;      // all 96 input blocks in order: 24 quads, 8 quads of reads in flight
#pragma unroll 3
                        for (int q = 0; q < 24; ++q) {
                            const i4 w4 = wq[q * 8], x4 = xq4[q];
                            z0 = __builtin_amdgcn_sdot4(w4[0], x4[0], z0, false);
                            z1 = __builtin_amdgcn_sdot4(w4[1], x4[1], z1, false);
                            z2 = __builtin_amdgcn_sdot4(w4[2], x4[2], z2, false);
                            z3 = __builtin_amdgcn_sdot4(w4[3], x4[3], z3, false);
                        }
                    } else {
                        const uint2 *offs = (const uint2 *)(sm_boff + bbeg);
                        for (int q = 0; q < nq; ++q) {
                            const i4 w4 = wq[q * 8];
                            const uint2 o = offs[q];
                            z0 = __builtin_amdgcn_sdot4(w4[0], *(const int *)(xb + (o.x & 0xFFFFu)), z0, false);
                            z1 = __builtin_amdgcn_sdot4(w4[1], *(const int *)(xb + (o.x >> 16)), z1, false);
                            z2 = __builtin_amdgcn_sdot4(w4[2], *(const int *)(xb + (o.y & 0xFFFFu)), z2, false);
                            z3 = __builtin_amdgcn_sdot4(w4[3], *(const int *)(xb + (o.y >> 16)), z3, false);
                        }
                    }
                    zrh = (float)((z0 + z1) + (z2 + z3)) * QS1;
                  }
                } else if constexpr (I8) {
                    typedef int i4 __attribute__((ext_vector_type(4)));
                    zrh = zrh * QS;
                    rec = rec * QS;
                    // recurrent part: 16 quantised inputs = 4 blocks (src/vec.h:274-304)
                    const i4 wr = ((const i4 *)(smem + L::brec))[r];
                    const i4 hb = *(const i4 *)(smem + L::hBq + s * 16);
                    {
                        float d[4];
                        dot4_cvt_x4(d, wr[0], wr[1], wr[2], wr[3], hb[0], hb[1], hb[2], hb[3]);
#pragma unroll
                        for (int k = 0; k < 4; ++k) rec = rec + d[k];
                    }
                    rec = rec * QS1;
                    // input part: four blocks of this lane's row per 16-byte read; groups are padded to x4
                    const int bbeg = sm_bstart[g], bend = sm_bstart[g + 1];
                    const int nq = (bend - bbeg) >> 2;
                    const i4 *wq = (const i4 *)(smem + L::bw) + (bbeg >> 2) * 8 + ri + (b_dense ? ((0x321100 >> (4 * g)) & 15) * 8 : 0);
                    const unsigned char *xb = smem + L::xqT + s * 384;
                    if (b_dense) {
                        // all 96 input blocks in order: 24 quads, software-pipelined in batches of 4 quads
                        // (8 LDS reads in flight while the previous batch feeds the dependent add chain)
                        const i4 *xq4 = (const i4 *)xb;
                        if constexpr (PACK2) {
                            // 128-VGPR variants (two workgroups per CU): the other workgroup hides the LDS latency, a
                            // two-quad pipeline is enough and keeps the kernel out of scratch memory
                            // ring of three quads (reads two quads ahead): 2 / 3 / 4 quads -> 169.2 / 172.0 / 172.1 M samples/s in round 5; the hand-scheduled
                            // assembly form of this loop (tools/gen_grub_asm.py --i8) was measured too: 168.4 vs 169.2 M, 15 more spilled VGPRs
                            {
                            i4 w0 = wq[0], x0 = xq4[0], w1 = wq[8], x1 = xq4[1];
#pragma unroll 3
                                for (int q = 0; q < 24; ++q) {
                                    const i4 w2 = wq[(q + 2) * 8], x2 = xq4[q + 2];     // (past the end on the last trips: padded / unused)
                                    float d[4];
                                    dot4_cvt_x4(d, w0[0], w0[1], w0[2], w0[3], x0[0], x0[1], x0[2], x0[3]);
                                    zrh = zrh + d[0]; zrh = zrh + d[1]; zrh = zrh + d[2]; zrh = zrh + d[3];
                                    w0 = w1; x0 = x1; w1 = w2; x1 = x2;
                                }
                            }
                        } else {
                        i4 wA[4], xA[4], wB[4], xB[4];
#define LPCN_LDQ(W, X, Q0) _Pragma("unroll") for (int k = 0; k < 4; ++k) { W[k] = wq[((Q0) + k) * 8]; X[k] = xq4[(Q0) + k]; }
#define LPCN_CPQ(W, X) _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                          \
                            float d[4];                                                                      \
                            dot4_cvt_x4(d, W[k][0], W[k][1], W[k][2], W[k][3], X[k][0], X[k][1], X[k][2], X[k][3]); \
                            zrh = zrh + d[0]; zrh = zrh + d[1]; zrh = zrh + d[2]; zrh = zrh + d[3]; }
                        LPCN_LDQ(wA, xA, 0)
#pragma unroll 1
                        for (int q = 0; q < 24; q += 8) {
                            LPCN_LDQ(wB, xB, q + 4)
                            __builtin_amdgcn_sched_barrier(0);
                            LPCN_CPQ(wA, xA)
                            __builtin_amdgcn_sched_barrier(0);
                            LPCN_LDQ(wA, xA, q + 8)          // past the end on the last trip: padded / unused
                            __builtin_amdgcn_sched_barrier(0);
                            LPCN_CPQ(wB, xB)
                            __builtin_amdgcn_sched_barrier(0);
                        }
#undef LPCN_LDQ
#undef LPCN_CPQ
                        }
                    } else {
                        const uint2 *offs = (const uint2 *)(sm_boff + bbeg);
                        for (int q = 0; q < nq; ++q) {
                            const i4 w4 = wq[q * 8];
                            const uint2 o = offs[q];
                            const int x0 = *(const int *)(xb + (o.x & 0xFFFFu)), x1 = *(const int *)(xb + (o.x >> 16));
                            const int x2 = *(const int *)(xb + (o.y & 0xFFFFu)), x3 = *(const int *)(xb + (o.y >> 16));
                            float d[4];
                            dot4_cvt_x4(d, w4[0], w4[1], w4[2], w4[3], x0, x1, x2, x3);
#pragma unroll
                            for (int k = 0; k < 4; ++k) zrh = zrh + d[k];
                        }
                    }
                    zrh = zrh * QS1;
                } else if (gb_mfma) {
#pragma unroll
                    for (int j = 0; j < NB; ++j) rec = __builtin_fmaf(sm_brec[j * RB + r], sm_hB[s * NB + j], rec);
                    mirror_wait();                           // all eight waves' partial sums are in LDS
                    const float4 pa = *(const float4 *)(sm_inh + (r * S + s) * 4);
                    zrh = zrh + ((pa.x + pa.y) + (pa.z + pa.w));
                    (void)g; (void)ri;
                } else if (gb_fsplit) {
#pragma unroll
                    for (int j = 0; j < NB; ++j) rec = __builtin_fmaf(sm_brec[j * RB + r], sm_hB[s * NB + j], rec);
                    zrh = zrh + gb_part_fast(s, 0, gb_qa);
                    (void)g; (void)ri;
                } else if (gb_lds) {
                    // state as a broadcast LDS read per block, weights as before: one hand-scheduled assembly block (tools/gen_grub_asm.py --lds S)
#pragma unroll
                    for (int j = 0; j < NB; ++j) rec = rec + sm_brec[j * RB + r] * sm_hB[s * NB + j];
                    uint32_t wp32 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)(smem + L::bw + (sm_bstart[g] * 8 + ri) * 16 + ((0x321100 >> (4 * g)) & 15) * 128);
                    uint32_t hp32 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)(smem + L::hA + s * 16);
                    if constexpr (S == 4) {
                        asm volatile(
#include "grub_lds_loop_s4.inc"
                            : [z] "+v"(zrh), [wp] "+v"(wp32), [hp] "+v"(hp32) : : LPCN_GRUB_LDS_CLOBBERS);
                    } else if constexpr (S == 2) {
                        asm volatile(
#include "grub_lds_loop_s2.inc"
                            : [z] "+v"(zrh), [wp] "+v"(wp32), [hp] "+v"(hp32) : : LPCN_GRUB_LDS_CLOBBERS);
                    } else if (gb_prod) {
                        // the first blocks as above, then the products of the others that waves 1..3 have formed meanwhile
                        asm volatile(
#include "grub_lds_loop_s1_first.inc"
                            : [z] "+v"(zrh), [wp] "+v"(wp32), [hp] "+v"(hp32) : : LPCN_GRUB_LDS32_CLOBBERS);
                        prod_wait();
                        uint32_t pp32 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)(smem + L::prod(Ap->nb_b, I8) + r * 16);
                        asm volatile(
#include "grub_prod_loop.inc"
                            : [z] "+v"(zrh), [pp] "+v"(pp32) : : LPCN_GRUB_PROD_CLOBBERS);
                    } else {
                        asm volatile(
#include "grub_lds_loop_s1.inc"
                            : [z] "+v"(zrh), [wp] "+v"(wp32), [hp] "+v"(hp32) : : LPCN_GRUB_LDS_CLOBBERS);
                    }
                } else {
                // Each group's block list is padded to a multiple of 4 (zero weights) by the host.
                // This phase runs one wave per SIMD, so latency must be hidden by software, and the
                // per-wave LDS issue rate (~25 clk per instruction) is as scarce as the dependent add
                // chain (4 adds per block).  The four rows of a lane quad belong to the same row group
                // and the same stream, so they need the same state blocks: lane k of the quad fetches
                // block 4q+k and the other three rows take it through DPP quad broadcasts folded into
                // the multiplies -- 1.25 LDS instructions per block instead of 2.
                // Pipeline: weights are fetched 3 blocks ahead (ring of 4), the state quad one quad
                // ahead, products are formed one block ahead of the running sum; scheduling barriers
                // pin one multiply into the shadow of every dependent add.  Reads past the end of a
                // group fetch valid (unused) LDS data; the host pads the arrays.
                const int bbeg = sm_bstart[g], bend = sm_bstart[g + 1];
                const int nq = (bend - bbeg) >> 2;                          // groups of 4 blocks
                // recurrent part first: an independent 16-term chain (src/nnet.c:356-361)
#pragma unroll
                for (int j = 0; j < NB; ++j) rec = rec + sm_brec[j * RB + r] * sm_hB[s * NB + j];
                const unsigned char *wptr = smem + L::bw + (bbeg * 8 + ri) * 16;   // this lane's row, block 0 of its group
                const unsigned char *hbase = smem + L::hA + s * 16;
                const int k4 = lane & 3;
                const unsigned short *offk = sm_boff + bbeg + k4;           // LDS offset of the block this lane fetches, per quad
                auto ldw = [&](int byte_off) { return *(const float4 *)(wptr + byte_off); };
                auto ldh = [&](int q) { return *(const float4 *)(hbase + offk[4 * q]); };
                float4 wr0 = ldw(0), wr1 = ldw(128), wr2 = ldw(256), wr3;
                float4 hq = ldh(0), hn;
                unsigned off_n = offk[4];
                float p0 = wr0.x * quad_bcast<0>(hq.x), p1 = wr0.y * quad_bcast<0>(hq.y);
                float p2 = wr0.z * quad_bcast<0>(hq.z), p3 = wr0.w * quad_bcast<0>(hq.w);
#define LPCN_SB __builtin_amdgcn_sched_barrier(0)
                // accumulate block b (p0..p3), form products of block b+1 from (WN, HK-th lane of HQ), fetch block b+3 into WL
#define LPCN_B_STEP(WL, WOFF, WN, HQ, HK, EXTRA)                                                        \
                {                                                                                       \
                    WL = ldw(WOFF);                                                                     \
                    EXTRA                                                                               \
                    LPCN_SB;                                                                            \
                    zrh = zrh + p0; LPCN_SB; const float t0_ = WN.x * quad_bcast<HK>(HQ.x); LPCN_SB;    \
                    zrh = zrh + p1; LPCN_SB; const float t1_ = WN.y * quad_bcast<HK>(HQ.y); LPCN_SB;    \
                    zrh = zrh + p2; LPCN_SB; const float t2_ = WN.z * quad_bcast<HK>(HQ.z); LPCN_SB;    \
                    zrh = zrh + p3; LPCN_SB; const float t3_ = WN.w * quad_bcast<HK>(HQ.w); LPCN_SB;    \
                    p0 = t0_; p1 = t1_; p2 = t2_; p3 = t3_;                                             \
                }
                for (int q = 0; q < nq; ++q) {
                    // entering: p = products of block 4q; wr1, wr2 = weights of blocks 4q+1, 4q+2; hq = state quad q
                    LPCN_B_STEP(wr3, 384, wr1, hq, 1, hn = *(const float4 *)(hbase + off_n); off_n = offk[4 * q + 8];)
                    LPCN_B_STEP(wr0, 512, wr2, hq, 2, )
                    LPCN_B_STEP(wr1, 640, wr3, hq, 3, )
                    LPCN_B_STEP(wr2, 768, wr0, hn, 0, )
                    hq = hn;
                    wptr += 512;
                }
#undef LPCN_B_STEP
#undef LPCN_SB
                }
                LPCN_PROF(8);      // GRU-B input mat-vec
                __builtin_amdgcn_s_setprio(0);
            } else if (gb_prod && wave < LPCN_WAVES / 2) {
                // ---- single stream: waves 1..3 (idle in this phase) multiply weight x state for GRU-B's last PROD_BLOCKS blocks, every third block
                // each, and leave the products in LDS for the chain wave (same rounding as its own v_pk_mul_f32: one multiply per term)
                const int g = r >> 3, ri = r & 7;
                const unsigned char *wrow = smem + L::bw + (sm_bstart[g] * 8 + ri) * 16 + ((0x321100 >> (4 * g)) & 15) * 128;
                float4 *pout = (float4 *)(smem + L::prod(Ap->nb_b, I8)) + r;
                // batches of four blocks: all eight reads first (the stores may alias them as far as the compiler knows), two batches in flight
                constexpr int PB = 4, NBATCH = ((L::PROD_BLOCKS + 2) / 3 + PB - 1) / PB;   // (the tail of the last batch is clamped to block 95)
                float4 wv[2][PB], hv[2][PB];
                auto blk = [&](const int k) { const int b = L::PROD_FIRST + wave - 1 + 3 * k; return b < 96 ? b : 95; };      // (a clamped block is written twice with the same value)
                auto fetch = [&](const int t) {
#pragma unroll
                    for (int k = 0; k < PB; ++k) {
                        const int b = blk(t * PB + k);
                        wv[t & 1][k] = *(const float4 *)(wrow + b * 128);
                        hv[t & 1][k] = *(const float4 *)(smem + L::hA + L::ha_off(b));
                    }
                };
                fetch(0);
#pragma unroll
                for (int t = 0; t < NBATCH; ++t) {
                    if (t + 1 < NBATCH) fetch(t + 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < PB; ++k) {
                        const float4 w4 = wv[t & 1][k], h4 = hv[t & 1][k];
                        pout[(blk(t * PB + k) - L::PROD_FIRST) * RB] = make_float4(h4.x * w4.x, h4.y * w4.y, h4.z * w4.z, h4.w * w4.w);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                prod_arrive();
            } else if (early_wave) {
                // ---- the head of the next sample's candidate chains (runs in the shadow of GRU-B)
                run_head();
            }
            if constexpr (FAST && !I8) {
                if (gb_fsplit) {                             // (workgroup-uniform)
                    if (wave >= LPCN_WAVES / 2 && wave - LPCN_WAVES / 2 < S)
                        sm_inh[(wave - LPCN_WAVES / 2) * 64 + lane] = gb_part_fast(wave - LPCN_WAVES / 2, gb_qa, 24);
                    __syncthreads();
                    if (gate_wave) zrh = zrh + sm_inh[s * 64 + lane];
                }
            }
            if (gate_wave) {
                // gates: rows [0,16) update, [16,32) reset, [32,48) candidate (src/nnet.c:362-371)
                const float sg = act_sigmoid<FAST>(zrh + rec, sm_tansig);
                // (ds_bpermute with an index formed from the re-materialised lane id: __shfl builds its own lane id, which the compiler
                // hoists out of the sample loop and spills -- a scratch load in front of GRU-B's gates)
                const float r_gate = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((16 + (lane & 15)) << 2, __builtin_bit_cast(int, sg)));
                const float hc = act_tanh<FAST>(zrh + rec * r_gate, sm_tansig);
                const float hc_i = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((32 + (lane & 15)) << 2, __builtin_bit_cast(int, hc)));
                if (lane < NB) {
                    const float hold = sm_hB[s * NB + lane];
                    const float hnew = sg * hold + (1.f - sg) * hc_i;
                    if ((live_mask >> s) & 1) {
                        sm_hB[s * NB + lane] = hnew;
                        if constexpr (I8) smem[L::hBq + s * NB + lane] = (unsigned char)quant_s8(hnew);
                        if constexpr (FAST) ((_Float16 *)(smem + L::hBh(Ap->nb_b, I8)))[s * NB + lane] = (_Float16)hnew;
                    }
                }
            }
            if constexpr (I8) {
                // (int8 blobs: waves 2 and 3 carry candidate heads too -- at S <= 2 they run no GRU-B -- and FAST splits GRU-B over
                // all waves with gate waves 0, GB_W, 2 GB_W, ...: a wave can be both a gate wave and an early wave; it computes its
                // candidate heads behind the gate stage)
                if (gate_wave && early_wave) run_head();
            }
            LPCN_PROF(11);     // GRU-B gates (gate waves) / early GRU-A slot (the others)
            __syncthreads();                                                   // B3
            LPCN_PROF(2);

            // ------------------------------------------ P4: dual-FC tree, all nodes at once --
            {
                const int node_level = node > 0 ? 31 - __clz(node) : 0;
                // (tried: the S x 16 GRU-B state values through one LDS read per lane + v_readlane into SGPR operands instead of
                // S x 4 broadcast ds_read_b128 per wave: P4 2.2 k -> 2.8 k clk, the 64 v_readlane cost more than the reads)
                // With the ballot's store in the loop body every stream is a basic block of its own, and hipcc runs the streams one after the
                // other (state reads -> 16 dependent sums -> table lookup -> ballot).  Evaluating them STAGE BY STAGE instead -- S
                // interleaved chains, S overlapping activations, the masks stored at the end -- was measured in round 4: the phase is bound by
                // VALU issue of the two waves per SIMD, not by latency, where the sums are separate multiplies and adds (PARITY fp32 124.2
                // vs 125.6 M, int8 155.7 vs 157.5 M: slower); FAST float, whose sums are fused and half as many instructions, gains
                // (161.9 vs 158.9 M) and uses the staged form.
                constexpr bool P4_STAGED = FAST && !I8;
                float sumv[S], logit[S];
                if constexpr (P4_STAGED) {
#pragma unroll
                    for (int s = 0; s < S; ++s) sumv[s] = fcb;
                    if constexpr (FAST) {
                        if (fc_f16) {
                            // fp16 dual FC (FAST sub-option): weights and GRU-B state as halves, fp32 accumulation, two MACs per
                            // v_dot2_f32_f16 -- half the instructions and half the operand bytes of the tree phase
                            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#pragma unroll
                            for (int j = 0; j < NB / 2; ++j)
#pragma unroll
                                for (int s = 0; s < S; ++s)
                                    sumv[s] = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, fcw[j]), __builtin_bit_cast(h2, ((const uint32_t *)(smem + L::hBh(Ap->nb_b, I8) + s * 32))[j]), sumv[s], false);
                        } else {
#pragma unroll
                            for (int j = 0; j < NB; ++j)
#pragma unroll
                                for (int s = 0; s < S; ++s) sumv[s] = __builtin_fmaf(fcw[j], sm_hB[s * NB + j], sumv[s]);
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < NB; ++j)
#pragma unroll
                            for (int s = 0; s < S; ++s) sumv[s] = sumv[s] + fcw[j] * sm_hB[s * NB + j];           // src/nnet.c:194-199 (per stream: j = 0..15 in order)
                    }
#pragma unroll
                    for (int s = 0; s < S; ++s) logit[s] = fcf * act_tanh<FAST>(sumv[s], sm_tansig);
                    unsigned long long mk[S];
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        // partner channel sits in the neighbouring lane: quad_perm [1,0,3,2]
                        const float vo = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, logit[s]), 0xB1, 0xf, 0xf, true));
                        const float lg = logit[s] + vo;                                 // sum1 += sum2
                        mk[s] = __ballot(sm_thr[s * 8 + node_level] < lg) & (wave == 0 ? 0x5555555555555554ull : 0x5555555555555555ull);
                    }
                    if (lane == 0) {
#pragma unroll
                        for (int s = 0; s < S; ++s) sm_mask[s * 8 + wave] = mk[s];
                    }
                } else {
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    float sum = fcb;
                    if constexpr (FAST) {
                        if (fc_f16) {
                            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                            const uint32_t *hh = (const uint32_t *)(smem + L::hBh(Ap->nb_b, I8) + s * 32);
#pragma unroll
                            for (int j = 0; j < NB / 2; ++j)
                                sum = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, fcw[j]), __builtin_bit_cast(h2, hh[j]), sum, false);
                        } else {
#pragma unroll
                            for (int j = 0; j < NB; ++j) sum = __builtin_fmaf(fcw[j], sm_hB[s * NB + j], sum);
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < NB; ++j) sum = sum + fcw[j] * sm_hB[s * NB + j];                   // src/nnet.c:194-199
                    }
                    const float v = fcf * act_tanh<FAST>(sum, sm_tansig);
                    // partner channel sits in the neighbouring lane: quad_perm [1,0,3,2]
                    const float vo = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
                    const float lg = v + vo;                                        // sum1 += sum2
                    // (the lanes that carry a decision -- channel 0, node > 0 -- are a constant of the wave: the even lanes, without lane 0 on wave 0)
                    const unsigned long long m = __ballot(sm_thr[s * 8 + node_level] < lg) & (wave == 0 ? 0x5555555555555554ull : 0x5555555555555555ull);
                    if (lane == 0) sm_mask[s * 8 + wave] = m;
                }
                }
            }
            // wave 0, before it waits at the barrier: the prediction terms of the next sample that do not involve the
            // sample about to be drawn -- tap j >= 1 will hold today's tap j-1 (src/lpcnet.c:252,262)
            float lpc_tap = 0.f, prod_old = 0.f;
            if (tid < 16 * S) {
                lpc_tap = sm_lpc[tid];
                prod_old = row_shr1(hist, 0.f) * lpc_tap;
            }
            __syncthreads();                                                   // B4
            LPCN_PROF(3);

            // ------------------------------------------------ P5: leader finishes the sample --
            // Not a workgroup phase: only wave 0 (lane s = stream s) and wave 1 (thresholds of the next
            // sample) work here; everybody else is already in the next sample's GRU-A.
            const bool more = smp + 1 < frame_len;
            if (more) ++seq;
            unsigned long long t_b4 = 0;
            if (tracing_any) t_b4 = __builtin_amdgcn_s_memtime();                 // tests: leader latency barrier -> publish
            // the leader's dependent chain is what every other wave waits for: it wins issue arbitration against the wave that
            // shares its SIMD (which is already running next sample's candidate items at priority 2) until the indices are out
            if (tid < 64) __builtin_amdgcn_s_setprio(3);
            if (tid < 16 * S) {
                const int lrow = (tid & 63) >> 4, tap = tid & 15;
                // (tried in round 5: the walk by the row's 16 lanes in two dependent steps -- lane c tests the c-th root-to-leaf path of a 4-level
                // subtree, a ballot names the lane that matched -- a third of the dependent depth, bit-exact, and slower: fp32 123.1 vs 126.3 M,
                // int8 164.1 vs 170.4 M.  And for int8 blobs at two streams per workgroup: levels 0..6 only with one stream per lane, the last level
                // evaluated here from an LDS table -- +4.5 % with the tree phase halved, -0.4 % once this wave pays for the extra level)
                auto walk_tree = [&](int lrow) {             // the sampler's 8 decisions from the 255 ballot bits of stream row lrow
                    typedef unsigned u4 __attribute__((ext_vector_type(4)));
                    const u4 *mk = (const u4 *)(sm_mask + lrow * 8);
                    const u4 qa = mk[0], qb = mk[1], qc = mk[2], qd = mk[3];
                    auto bit_of = [](unsigned word, int k) { return (int)((word >> (2 * k)) & 1u); };
                    int val = 0;
#pragma unroll
                    for (int b = 0; b < 4; ++b) val = (val << 1) | bit_of(qa[0], (1 << b) | val);        // nodes 1..15
                    val = (val << 1) | bit_of(qa[1], val);                                               // nodes 16..31
                    val = (val << 1) | bit_of((val & 16) ? qa[3] : qa[2], val & 15);                     // nodes 32..63
                    {
                        const int k = val >> 4;                                                          // nodes 64..127: dwords 4..7
                        const unsigned lo = (k & 1) ? qb[1] : qb[0], hi = (k & 1) ? qb[3] : qb[2];
                        val = (val << 1) | bit_of((k & 2) ? hi : lo, val & 15);
                    }
                    {
                        const int k = val >> 4;                                                          // nodes 128..255: dwords 8..15
                        const unsigned a0 = (k & 1) ? qc[1] : qc[0], a1 = (k & 1) ? qc[3] : qc[2];
                        const unsigned a2 = (k & 1) ? qd[1] : qd[0], a3 = (k & 1) ? qd[3] : qd[2];
                        const unsigned b0 = (k & 2) ? a1 : a0, b1 = (k & 2) ? a3 : a2;
                        val = (val << 1) | bit_of((k & 4) ? b1 : b0, val & 15);
                    }
                    return val;
                };
                float pcm = 0.f, deemph = 0.f;
                int exc = 0;                   // (tree_val: the tree's own decision -- teacher forcing overrides exc)
                if (live) {                                  // (all 16 lanes of a stream's row do the same walk)
                    const float pred = sm_lead[lrow * 8 + 0];           // (issued together with the mask reads)
                    deemph = sm_lead[lrow * 8 + 1];
                    const int val = walk_tree(lrow);
                    exc = val;
                    if (smp < preload) {                                        // src/lpcnet.c:256-258
                        const float x = (float)sm_pcm[lrow * LPCN_FRAME_SIZE + smp];
                        exc = lpcn_lin2ulaw(x - 0.85f * deemph - pred);
                        pcm = x - 0.85f * deemph;
                    } else {
                        pcm = pred + sm_ulaw[exc];                              // src/lpcnet.c:260
                    }
                }
                {                                            // history shifts by one, the new sample enters at tap 0 (src/lpcnet.c:262-263)
                    const float shifted = row_shr1(hist, pcm);
                    hist = live ? shifted : hist;
                }
                if (tap == 0 && live) ((int *)sm_lead)[lrow * 8 + 2] = exc;
                if (tracing_lane0(live)) {
                    LPCN_GLOBAL float *d = as_global_rw(Ap->dbg) + ((size_t)f * LPCN_FRAME_SIZE + smp) * LPCN_DBG_STRIDE + 400;
                    d[1] = (float)(sm_idx[0] & 0xFF); d[2] = (float)((sm_idx[0] >> 8) & 0xFF);
                }
                // the next sample's indices first: the other waves are waiting for them
                if (more) { open_sample(live, pcm, tap == 0 ? pcm * lpc_tap : prod_old, exc, false); publish_indices(); }
                __builtin_amdgcn_s_setprio(0);
                if (tracing_lane0(live))
                    as_global_rw(Ap->dbg)[((size_t)f * LPCN_FRAME_SIZE + smp) * LPCN_DBG_STRIDE + 405] = (float)(unsigned)(__builtin_amdgcn_s_memtime() - t_b4);
                if (tap == 0) {
                    if (live) {
                        if (tracing_lane0(true)) {
                            LPCN_GLOBAL float *d = as_global_rw(Ap->dbg) + ((size_t)f * LPCN_FRAME_SIZE + smp) * LPCN_DBG_STRIDE + 400;
                            d[0] = (float)exc; d[3] = pcm + 0.85f * deemph; d[4] = pcm - sm_ulaw[exc];
                            // the tree's own decision (teacher forcing overrides exc) is walked AGAIN here, in the cold trace branch:
                            // any use of the leader's value this far down (a register, an extra LDS store) cost 17 spilled
                            // SGPRs in the sample loop and 2.3 % of the float kernel
                            asm volatile("" ::: "memory");
                            d[6] = (float)walk_tree(lrow);
                        }
                        pcm = pcm + 0.85f * deemph;
                        sm_lead[lrow * 8 + 1] = pcm;                            // de-emphasis memory
                        if (smp >= preload) sm_pcm[lrow * LPCN_FRAME_SIZE + smp] = (short)lpcn_round_pcm(pcm);
                    } else {
                        sm_pcm[lrow * LPCN_FRAME_SIZE + smp] = 0;
                    }
                }
            }
            if (more && tid >= 64 && tid < 64 + S && live) draw_thresholds(tid - 64);
            if (tracing) {
                LPCN_GLOBAL float *d = as_global_rw(Ap->dbg) + ((size_t)f * LPCN_FRAME_SIZE + smp) * LPCN_DBG_STRIDE;
                if (tid < NA) d[tid] = sm_hT[tid * S];
                if (tid < NB) d[384 + tid] = sm_hB[tid];
            }
            // no workgroup barrier here (see above); keep the compiler from mixing the two samples' code
            // (the comment marks the bottom of the sample loop in the assembly: tools/kernel_resources.py)
            asm volatile("; LPCN_SAMPLE_LOOP_END" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            LPCN_PROF(4);
        }

        // ---- flush the frame's PCM (S*160 samples, coalesced)
        __syncthreads();        // the leaders' last samples
        {
            auto *out = as_global_rw(Ap->pcm);
            const size_t pstride = (size_t)Ap->pcm_stride;
            for (int i = tid0; i < S * LPCN_FRAME_SIZE; i += LPCN_WG_THREADS) {
                const int s = i / LPCN_FRAME_SIZE, k = i % LPCN_FRAME_SIZE;
                if (s < n_valid && k < frame_len) out[(size_t)(s0 + s) * pstride + (size_t)f * LPCN_FRAME_SIZE + k] = sm_pcm[i];
            }
        }
        __syncthreads();
    }

#if LPCN_ENABLE_PROF
    if (profiling && (tid0 & 63) == 0) {
#pragma unroll
        for (int i = 0; i < 12; ++i) prof[(tid0 >> 6) * 12 + i] += pt[i];
    }
#endif
    // ------------------------------------------------------------------ write state back ----
    {
        const int tid = tid0;
        for (int i = tid; i < S * NA; i += LPCN_WG_THREADS) {
            const int s = i / NA, n = i % NA;
            if (s < n_valid)
                states[s0 + s].gru_a[n] = sm_hT[n * S + s];
        }
        for (int i = tid; i < S * NB; i += LPCN_WG_THREADS)
            if (i / NB < n_valid) states[s0 + i / NB].gru_b[i % NB] = sm_hB[i];
        if (tid < 16 * S && LPCN_LROW < n_valid) states[s0 + LPCN_LROW].last_sig[LPCN_TAP] = hist;
        if (tid < n_valid) {
            auto *st = &states[s0 + tid];
            const int *li = (const int *)sm_lead + tid * 8;
#pragma unroll
            for (int j = 0; j < LPCN_LPC_ORDER; ++j) st->lpc[j] = sm_lpc[tid * LPCN_LPC_ORDER + j];
            st->deemph_mem = sm_lead[tid * 8 + 1];
            st->last_exc = li[2];
#pragma unroll
            for (int j = 0; j < 4; ++j) st->rng[j] = (uint32_t)li[4 + j];
        }
    }
}

#undef tracing
#undef tracing_any
#undef tracing_lane0
#undef fc_w_g
#undef fc_b_g
#undef fc_f_g

}  // namespace lpcn
