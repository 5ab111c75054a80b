// Frame-rate (100 Hz) kernels of the LPCNet HIP engine.
//
// Replaces run_frame_network (src/lpcnet.c:82-120): pitch embedding, two causal conv1d layers,
// two dense layers, the 128->1152 / 128->48 conditioning projections (src/nnet.c:122-135,
// :452-470, mat-vec order of src/vec.h:131-162) and lpc_from_cepstrum (src/freq.c:310-320 with
// its 320-point kiss FFT, src/kiss_fft.c:518-586, and Levinson recursion, src/freq.c:86-127).
//
// PARITY arithmetic: every output accumulates its inputs in ascending input order with separately
// rounded multiply and add (-ffp-contract=off), table tanh -> bit-identical to the reference's
// generic-C build.  Parallelism comes from (stream x frame x output neuron); all frames of a
// chunk are known up front, so weights are reused across FT frames per load.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "lpcnet_engine.h"
#include "lpcnet_math.h"
#include "lpcnet_exp10.h"

struct LpcnFrameModel {
    const float *conv1_w, *conv1_b, *conv2_w, *conv2_b;   // [3][in][128]
    const float *pitch_emb;                               // [256][64]
    const float *dense1_w, *dense1_b, *dense2_w, *dense2_b;
    const float *a_dense_w, *a_dense_b;                   // [128][1152]
    const float *b_dense_w, *b_dense_b;                   // [128][48]
    const float *tab_tansig, *tab_idct, *tab_tw;
    const short *tab_bitrev;
    float lpc_gamma;
    int end2end;                 // END2END model: LPC = rc2lpc(first 16 conditioning outputs), no cepstral LPC, no delay line
};

namespace lpcn {

constexpr int FT = 8;            // frames per tile (weight reuse factor)
#define LPCN_FRAME_UNROLL 16     // trips of a weight loop unrolled together in the one-frame kernels: that many L2 loads of a lane in flight (the sums keep their order)
#define LPCN_FRAME_UNROLL_F1 1   // ... in the chunk kernel (FT frames of one stream per tile: throughput-bound, the hint costs it 0.3 ms per 25-frame step)
#define LPCN_FRAME_UNROLL_PJ8 1  // ... in the projection kernel with full tiles
#define LPCN_STR2(x) #x
#define LPCN_STR(x) LPCN_STR2(x)
#if LPCN_FRAME_UNROLL_F1 > 1
#define LPCN_PRAGMA_F1 _Pragma(LPCN_STR(unroll LPCN_FRAME_UNROLL_F1))
#else
#define LPCN_PRAGMA_F1           /* (the compiler's own choice, as before round 5) */
#endif
#define LPCN_PROJ_SPLIT_MAX 2048 // (stream, frame) items up to which the projection's ten row blocks are separate workgroups
constexpr int FIN = LPCN_FRAME_IN, CN = LPCN_COND;

// ---------------------------------------------------------------------------------------------
// Kernel F1: features -> cond[128] for every frame of the chunk.  One workgroup (128 lanes, lane =
// output neuron) per stream walks the chunk in tiles of FT frames, carrying the conv histories.
// Also: LPC delay-line shift (src/lpcnet.c:110-111) and frame_count bookkeeping (:119).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void frame_cond_kernel(LpcnFrameModel M, int n_frames, const float *feat, int feat_stride,
                                                         size_t feat_stream_stride, lpcn_stream_state *states, int *fc_base,
                                                         float *cond_out /*[stream][n_frames][128]*/, float *lpc_out /*[stream][n_frames][16]*/)
{
    __shared__ float x1[(FT + 2) * FIN];      // conv1 window: 2 history frames + FT new frames
    __shared__ float x2[(FT + 2) * CN];       // conv2 window (conv1 outputs after the start-up zeroing)
    __shared__ float y[FT * CN];              // layer outputs of the tile
    __shared__ float tansig[204];
    const int i = threadIdx.x;
    const int stream = blockIdx.x;
    lpcn_stream_state *st = &states[stream];
    const float *f = feat + (size_t)stream * feat_stream_stride;
    for (int k = i; k < 201; k += 128) tansig[k] = M.tab_tansig[k];
    const int fc0 = st->frame_count;
    for (int k = i; k < 2 * FIN; k += 128) x1[k] = st->conv1_mem[k];
    for (int k = i; k < 2 * CN; k += 128) x2[k] = st->conv2_mem[k];
    // LPC delay line: frames 0,1 of the chunk use the coefficients of the two previous frames
    if (i < LPCN_LPC_ORDER) {
        float g = M.lpc_gamma, gi = g;
        for (int k = 0; k < i; ++k) gi *= g;
        if (!M.end2end) {
            lpc_out[((size_t)stream * n_frames + 0) * LPCN_LPC_ORDER + i] = st->old_lpc[1][i] * gi;
            if (n_frames >= 2) lpc_out[((size_t)stream * n_frames + 1) * LPCN_LPC_ORDER + i] = st->old_lpc[0][i] * gi;
            else st->old_lpc[1][i] = st->old_lpc[0][i];
        }
    }
    __syncthreads();

    for (int t0 = 0; t0 < n_frames; t0 += FT) {
        const int nt = n_frames - t0 < FT ? n_frames - t0 : FT;
        // inputs of the tile: 20 features + 64-d pitch embedding (src/lpcnet.c:93-97)
        for (int k = i; k < nt * FIN; k += 128) {
            const int t = k / FIN, c = k % FIN;
            const float *ft = f + (size_t)(t0 + t) * feat_stride;
            float v;
            if (c < LPCN_NB_FEAT) v = ft[c];
            else {
                int pitch = (int)floor(.1 + (double)(50.f * ft[LPCN_NB_BANDS]) + 100);
                pitch = pitch < 33 ? 33 : (pitch > 255 ? 255 : pitch);
                v = M.pitch_emb[pitch * LPCN_PITCH_EMB + (c - LPCN_NB_FEAT)];
            }
            x1[(2 + t) * FIN + c] = v;
        }
        __syncthreads();
        float acc[FT];
        // conv1: out[t] = b + sum_j W[j][i] * window_t[j], window_t = x1[t*FIN .. t*FIN+3*FIN)
#pragma unroll
        for (int t = 0; t < FT; ++t) acc[t] = M.conv1_b[i];
        LPCN_PRAGMA_F1
        for (int j = 0; j < 3 * FIN; ++j) {
            const float wv = M.conv1_w[j * CN + i];
#pragma unroll
            for (int t = 0; t < FT; ++t) acc[t] = acc[t] + wv * x1[t * FIN + j];
        }
#pragma unroll
        for (int t = 0; t < FT; ++t) {
            float v = lpcn_tanh(acc[t], tansig);
            int fc = fc0 + t0 + t; if (fc > 1000) fc = 1000;
            if (fc < 1) v = 0.f;                                   // src/lpcnet.c:99
            x2[(2 + t) * CN + i] = v;
        }
        __syncthreads();
        // conv2
#pragma unroll
        for (int t = 0; t < FT; ++t) acc[t] = M.conv2_b[i];
        LPCN_PRAGMA_F1
        for (int j = 0; j < 3 * CN; ++j) {
            const float wv = M.conv2_w[j * CN + i];
#pragma unroll
            for (int t = 0; t < FT; ++t) acc[t] = acc[t] + wv * x2[t * CN + j];
        }
#pragma unroll
        for (int t = 0; t < FT; ++t) {
            float v = lpcn_tanh(acc[t], tansig);
            int fc = fc0 + t0 + t; if (fc > 1000) fc = 1000;
            if (fc < LPCN_FEATURES_DELAY) v = 0.f;                 // src/lpcnet.c:101
            y[t * CN + i] = v;
        }
        __syncthreads();
        // dense1
#pragma unroll
        for (int t = 0; t < FT; ++t) acc[t] = M.dense1_b[i];
        LPCN_PRAGMA_F1
        for (int j = 0; j < CN; ++j) {
            const float wv = M.dense1_w[j * CN + i];
#pragma unroll
            for (int t = 0; t < FT; ++t) acc[t] = acc[t] + wv * y[t * CN + j];
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < FT; ++t) y[t * CN + i] = lpcn_tanh(acc[t], tansig);
        __syncthreads();
        // dense2
#pragma unroll
        for (int t = 0; t < FT; ++t) acc[t] = M.dense2_b[i];
        LPCN_PRAGMA_F1
        for (int j = 0; j < CN; ++j) {
            const float wv = M.dense2_w[j * CN + i];
#pragma unroll
            for (int t = 0; t < FT; ++t) acc[t] = acc[t] + wv * y[t * CN + j];
        }
#pragma unroll
        for (int t = 0; t < FT; ++t)
            if (t < nt) cond_out[((size_t)stream * n_frames + t0 + t) * CN + i] = lpcn_tanh(acc[t], tansig);
        __syncthreads();
        // slide the histories: keep the last two frames of this tile (src/nnet.c:469)
        float h1[2], h2[2];
        for (int k = i; k < 2 * FIN; k += 128) h1[(k - i) / 128] = x1[nt * FIN + k];
        for (int k = i; k < 2 * CN; k += 128) h2[(k - i) / 128] = x2[nt * CN + k];
        __syncthreads();
        for (int k = i; k < 2 * FIN; k += 128) x1[k] = h1[(k - i) / 128];
        for (int k = i; k < 2 * CN; k += 128) x2[k] = h2[(k - i) / 128];
        __syncthreads();
    }
    for (int k = i; k < 2 * FIN; k += 128) st->conv1_mem[k] = x1[k];
    for (int k = i; k < 2 * CN; k += 128) st->conv2_mem[k] = x2[k];
    if (i == 0) {
        fc_base[stream] = fc0;
        int fc = fc0 + n_frames;                                    // src/lpcnet.c:119 (saturates at 1000)
        st->frame_count = fc > 1000 ? 1000 : fc;
    }
}

// ---------------------------------------------------------------------------------------------
// Kernel F1', n_frames == 1 (the real-time operating point: one 10-ms frame per stream and step, src/lpcnet_demo.c:203-219, and every
// pass of the legacy API's combining dispatcher): the tile runs over ST STREAMS instead of FT frames of one stream, so a weight
// load still feeds ST accumulators.  F1 with one frame per stream reads the 453 KB of conv / dense weights once per stream and
// throws 7 of its 8 accumulators away: 0.81 ms for 8 192 streams (8 % of the 10-ms step), L2-bandwidth bound.  Same arithmetic per
// output (inputs in ascending order, multiply and add rounded separately): bit-identical to F1.
// ---------------------------------------------------------------------------------------------
// (round 5: ST -- streams per tile, 8 for a full batch -- is a template parameter: a batch of one to four streams, i.e. every call of the
// reference's per-frame API that finds no company in the combining dispatcher, ran eight accumulators for one stream, instruction-bound on
// two waves; and the weight loops carry an unroll hint so that a lane has LPCN_FRAME_UNROLL L2 loads in flight instead of one per trip.
// rocprofv3 on the reference's demo, one stream, per frame: this kernel 57.7 us and frame_proj 75.6 us before, ~40 us for all frame
// kernels together after)
template <int ST>
__global__ __launch_bounds__(128) void frame_cond_t1_kernel(LpcnFrameModel M, int n_streams, const float *feat, size_t feat_stream_stride,
                                                            lpcn_stream_state *states, int *fc_base, float *cond_out /*[stream][128]*/,
                                                            float *lpc_out /*[stream][16]*/)
{
    __shared__ float x1[ST][3 * FIN];         // per stream: conv1 window = 2 history frames + the new frame
    __shared__ float x2[ST][3 * CN];          // conv2 window
    __shared__ float y[ST][CN];
    __shared__ float tansig[204];
    __shared__ int fcs[ST];
    const int i = threadIdx.x;
    const int s0 = blockIdx.x * ST;
    const int ns = n_streams - s0 < ST ? n_streams - s0 : ST;
    for (int k = i; k < 201; k += 128) tansig[k] = M.tab_tansig[k];
    for (int k = i; k < ST * 2 * FIN; k += 128) {
        const int t = k / (2 * FIN), c = k % (2 * FIN);
        x1[t][c] = t < ns ? states[s0 + t].conv1_mem[c] : 0.f;
    }
    for (int k = i; k < ST * 2 * CN; k += 128) {
        const int t = k / (2 * CN), c = k % (2 * CN);
        x2[t][c] = t < ns ? states[s0 + t].conv2_mem[c] : 0.f;
    }
    // inputs: 20 features + 64-d pitch embedding (src/lpcnet.c:93-97)
    for (int k = i; k < ST * FIN; k += 128) {
        const int t = k / FIN, c = k % FIN;
        float v = 0.f;
        if (t < ns) {
            const float *ft = feat + (size_t)(s0 + t) * feat_stream_stride;
            if (c < LPCN_NB_FEAT) v = ft[c];
            else {
                int pitch = (int)floor(.1 + (double)(50.f * ft[LPCN_NB_BANDS]) + 100);
                pitch = pitch < 33 ? 33 : (pitch > 255 ? 255 : pitch);
                v = M.pitch_emb[pitch * LPCN_PITCH_EMB + (c - LPCN_NB_FEAT)];
            }
        }
        x1[t][2 * FIN + c] = v;
    }
    if (i < ST) fcs[i] = i < ns ? states[s0 + i].frame_count : 0;
    // LPC delay line with a single frame: the frame uses the coefficients of two frames ago, the line shifts (src/lpcnet.c:110-111)
    if (!M.end2end) {
        for (int k = i; k < ns * LPCN_LPC_ORDER; k += 128) {
            const int t = k / LPCN_LPC_ORDER, c = k % LPCN_LPC_ORDER;
            lpcn_stream_state *st = &states[s0 + t];
            float g = M.lpc_gamma, gi = g;
            for (int q = 0; q < c; ++q) gi *= g;
            lpc_out[(size_t)(s0 + t) * LPCN_LPC_ORDER + c] = st->old_lpc[1][c] * gi;
            st->old_lpc[1][c] = st->old_lpc[0][c];
        }
    }
    __syncthreads();
    float acc[ST];
#pragma unroll
    for (int t = 0; t < ST; ++t) acc[t] = M.conv1_b[i];
#pragma unroll LPCN_FRAME_UNROLL
    for (int j = 0; j < 3 * FIN; ++j) {
        const float wv = M.conv1_w[j * CN + i];
#pragma unroll
        for (int t = 0; t < ST; ++t) acc[t] = acc[t] + wv * x1[t][j];
    }
#pragma unroll
    for (int t = 0; t < ST; ++t) {
        float v = lpcn_tanh(acc[t], tansig);
        if (fcs[t] < 1) v = 0.f;                                      // src/lpcnet.c:99
        x2[t][2 * CN + i] = v;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < ST; ++t) acc[t] = M.conv2_b[i];
#pragma unroll LPCN_FRAME_UNROLL
    for (int j = 0; j < 3 * CN; ++j) {
        const float wv = M.conv2_w[j * CN + i];
#pragma unroll
        for (int t = 0; t < ST; ++t) acc[t] = acc[t] + wv * x2[t][j];
    }
#pragma unroll
    for (int t = 0; t < ST; ++t) {
        float v = lpcn_tanh(acc[t], tansig);
        if (fcs[t] < LPCN_FEATURES_DELAY) v = 0.f;                    // src/lpcnet.c:101
        y[t][i] = v;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < ST; ++t) acc[t] = M.dense1_b[i];
#pragma unroll LPCN_FRAME_UNROLL
    for (int j = 0; j < CN; ++j) {
        const float wv = M.dense1_w[j * CN + i];
#pragma unroll
        for (int t = 0; t < ST; ++t) acc[t] = acc[t] + wv * y[t][j];
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < ST; ++t) y[t][i] = lpcn_tanh(acc[t], tansig);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < ST; ++t) acc[t] = M.dense2_b[i];
#pragma unroll LPCN_FRAME_UNROLL
    for (int j = 0; j < CN; ++j) {
        const float wv = M.dense2_w[j * CN + i];
#pragma unroll
        for (int t = 0; t < ST; ++t) acc[t] = acc[t] + wv * y[t][j];
    }
#pragma unroll
    for (int t = 0; t < ST; ++t)
        if (t < ns) cond_out[(size_t)(s0 + t) * CN + i] = lpcn_tanh(acc[t], tansig);
    // histories: the last two frames of the window (src/nnet.c:469); frame_count (src/lpcnet.c:119)
    for (int k = i; k < ns * 2 * FIN; k += 128) states[s0 + k / (2 * FIN)].conv1_mem[k % (2 * FIN)] = x1[k / (2 * FIN)][FIN + k % (2 * FIN)];
    for (int k = i; k < ns * 2 * CN; k += 128) states[s0 + k / (2 * CN)].conv2_mem[k % (2 * CN)] = x2[k / (2 * CN)][CN + k % (2 * CN)];
    if (i < ns) {
        fc_base[s0 + i] = fcs[i];
        states[s0 + i].frame_count = fcs[i] + 1 > 1000 ? 1000 : fcs[i] + 1;
    }
}

// ---------------------------------------------------------------------------------------------
// Kernel F2: cond[128] -> GRU-A conditioning (1152, bias included) and GRU-B conditioning (48)
// (src/lpcnet.c:105-106).  Workgroup = tile of FT consecutive (stream, frame) items of the flattened [stream][frame] index -- all
// three arrays are dense in it -- so a tile is full whatever the number of frames per stream (round 5: with one frame per stream,
// the real-time case, a tile per stream used one accumulator of eight).  Lane = output column.
// ---------------------------------------------------------------------------------------------
// (round 5: for small batches the nine 128-row blocks of the GRU-A conditioning and the GRU-B conditioning are blockIdx.y = 0..9 -- one
// workgroup walked all ten in turn, 76 us for a single (stream, frame) item; and the tile size is a template parameter, see frame_cond_t1_kernel)
template <int FTT>
__global__ __launch_bounds__(128) void frame_proj_kernel(LpcnFrameModel M, size_t n_items, const float *cond,
                                                         float *cond_a, float *cond_b)
{
    __shared__ float c[FTT * CN];
    const int i = threadIdx.x;
    const size_t item0 = (size_t)blockIdx.x * FTT;
    const int nt = n_items - item0 < (size_t)FTT ? (int)(n_items - item0) : FTT;
    for (int k = i; k < FTT * CN; k += 128) {
        const int t = k / CN;
        c[k] = t < nt ? cond[(item0 + t) * CN + k % CN] : 0.f;
    }
    __syncthreads();
    float acc[FTT];
    constexpr int NRB = LPCN_ROWS_A / 128;             // row blocks of the GRU-A conditioning; block NRB = the GRU-B conditioning
    constexpr int UNR = FTT == FT ? LPCN_FRAME_UNROLL_PJ8 : LPCN_FRAME_UNROLL;
    // gridDim.y == NRB + 1: one row block per workgroup (few items: latency matters); gridDim.y == 1: this workgroup walks all of them
    const int rb0 = gridDim.y > 1 ? blockIdx.y : 0, rb1 = gridDim.y > 1 ? blockIdx.y + 1 : NRB + 1;
    for (int rb = rb0; rb < rb1; ++rb) {
        if (rb < NRB) {
            const int r = rb * 128 + i;
#pragma unroll
            for (int t = 0; t < FTT; ++t) acc[t] = M.a_dense_b[r];
            if constexpr (UNR > 1) {
#pragma unroll UNR
                for (int j = 0; j < CN; ++j) {
                    const float wv = M.a_dense_w[j * LPCN_ROWS_A + r];
#pragma unroll
                    for (int t = 0; t < FTT; ++t) acc[t] = acc[t] + wv * c[t * CN + j];
                }
            } else {
                for (int j = 0; j < CN; ++j) {
                    const float wv = M.a_dense_w[j * LPCN_ROWS_A + r];
#pragma unroll
                    for (int t = 0; t < FTT; ++t) acc[t] = acc[t] + wv * c[t * CN + j];
                }
            }
#pragma unroll
            for (int t = 0; t < FTT; ++t)
                if (t < nt) cond_a[(item0 + t) * LPCN_ROWS_A + r] = acc[t];
        } else if (i < LPCN_ROWS_B) {
#pragma unroll
            for (int t = 0; t < FTT; ++t) acc[t] = M.b_dense_b[i];
            if constexpr (UNR > 1) {
#pragma unroll UNR
                for (int j = 0; j < CN; ++j) {
                    const float wv = M.b_dense_w[j * LPCN_ROWS_B + i];
#pragma unroll
                    for (int t = 0; t < FTT; ++t) acc[t] = acc[t] + wv * c[t * CN + j];
                }
            } else {
                for (int j = 0; j < CN; ++j) {
                    const float wv = M.b_dense_w[j * LPCN_ROWS_B + i];
#pragma unroll
                    for (int t = 0; t < FTT; ++t) acc[t] = acc[t] + wv * c[t * CN + j];
                }
            }
#pragma unroll
            for (int t = 0; t < FTT; ++t)
                if (t < nt) cond_b[(item0 + t) * LPCN_ROWS_B + i] = acc[t];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Kernel F3: LPC coefficients from the 18 cepstral features of each frame.  One wavefront per
// (stream, frame); the 320-point FFT lives in LDS and reproduces the reference's butterfly
// arithmetic exactly (radix 4,4,4 then 5; 1/320 folded into the digit-reversal copy).
// ---------------------------------------------------------------------------------------------
struct cpx { float r, i; };
__device__ __forceinline__ cpx cmul(cpx a, cpx b) { cpx m; m.r = a.r * b.r - a.i * b.i; m.i = a.r * b.i + a.i * b.r; return m; }
__device__ __forceinline__ cpx cadd(cpx a, cpx b) { cpx m; m.r = a.r + b.r; m.i = a.i + b.i; return m; }
__device__ __forceinline__ cpx csub(cpx a, cpx b) { cpx m; m.r = a.r - b.r; m.i = a.i - b.i; return m; }

constexpr int LPC_WAVES = 4;     // wavefronts (= frames) per workgroup

__global__ __launch_bounds__(64 * LPC_WAVES) void lpc_kernel(LpcnFrameModel M, int n_streams, int n_frames, const float *feat, int feat_stride,
                                                             size_t feat_stream_stride, lpcn_stream_state *states, float *lpc_out)
{
    __shared__ cpx fbuf[LPC_WAVES][320];
    __shared__ float ex[LPC_WAVES][LPCN_NB_BANDS];
    __shared__ float xr[LPC_WAVES][164];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t item = (size_t)blockIdx.x * LPC_WAVES + wv;
    const bool valid = item < (size_t)n_streams * n_frames;
    const int stream = valid ? (int)(item / n_frames) : 0, t = valid ? (int)(item % n_frames) : 0;
    const float *c = feat + (size_t)stream * feat_stream_stride + (size_t)t * feat_stride;
    cpx *F = fbuf[wv];
    const cpx *TW = (const cpx *)M.tab_tw;
    static const short band_edge[LPCN_NB_BANDS] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28, 34, 40};   // src/freq.c:46-49
    static const float band_comp[LPCN_NB_BANDS] = {0.8f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 0.666667f, 0.5f, 0.5f, 0.5f,
                                                   0.333333f, 0.25f, 0.25f, 0.2f, 0.166667f, 0.173913f};             // src/freq.c:51-53
    // inverse DCT of the cepstrum (+4 on c0) and 10^x with the band compensation (src/freq.c:230-240, :317-318)
    if (lane < LPCN_NB_BANDS) {
        float sum = 0.f;
        for (int j = 0; j < LPCN_NB_BANDS; ++j) {
            float cj = c[j];
            if (j == 0) cj = cj + 4.f;
            sum = sum + cj * M.tab_idct[lane * LPCN_NB_BANDS + j];
        }
        const float e = (float)((double)sum * sqrt(2. / LPCN_NB_BANDS));
        // pow(10.f, e) in double, then the product with the band compensation rounded to float (src/freq.c:317-318);
        // lpcn_exp10 is this engine's own correctly rounded 10^e (lpcnet_exp10.h), not the device math library's pow
        ex[wv][lane] = (float)(lpcn_exp10(e) * (double)band_comp[lane]);
    }
    __syncthreads();
    // band interpolation (src/freq.c:202-215); bin 160 forced to 0 (:286)
    for (int k = lane; k < 161; k += 64) {
        float v = 0.f;
        if (k < 160) {
            int b = 0;
            while (b < LPCN_NB_BANDS - 2 && k >= band_edge[b + 1] * 4) ++b;
            const int size = (band_edge[b + 1] - band_edge[b]) * 4, j = k - band_edge[b] * 4;
            const float frac = (float)j / (float)size;
            v = (1.f - frac) * ex[wv][b] + frac * ex[wv][b + 1];
        }
        xr[wv][k] = v;
    }
    __syncthreads();
    // Hermitian extension + digit-reversal copy with the 1/320 scale (src/freq.c:260-266, src/kiss_fft.c:579-584)
    for (int k = lane; k < 320; k += 64) {
        const float re = k < 161 ? xr[wv][k] : xr[wv][320 - k];
        const float im = k < 161 ? 0.f : -0.f;
        cpx v; v.r = 0.0031250000f * re; v.i = 0.0031250000f * im;
        F[M.tab_bitrev[k]] = v;
    }
    __syncthreads();
    // radix-4, m = 1: 80 butterflies with unit twiddles (src/kiss_fft.c:111-131)
    for (int b = lane; b < 80; b += 64) {
        cpx *f = F + 4 * b;
        cpx f0 = f[0], f1 = f[1], f2 = f[2], f3 = f[3];
        cpx s0 = csub(f0, f2);
        f0 = cadd(f0, f2);
        cpx s1 = cadd(f1, f3);
        f2 = csub(f0, s1);
        f0 = cadd(f0, s1);
        s1 = csub(f1, f3);
        f1.r = s0.r + s1.i; f1.i = s0.i - s1.r;
        f3.r = s0.r - s1.i; f3.i = s0.i + s1.r;
        f[0] = f0; f[1] = f1; f[2] = f2; f[3] = f3;
    }
    __syncthreads();
    // radix-4 stages m = 4 (fstride 20) and m = 16 (fstride 5) (src/kiss_fft.c:132-168)
#pragma unroll
    for (int stage = 0; stage < 2; ++stage) {
        const int m = stage == 0 ? 4 : 16, fs = stage == 0 ? 20 : 5;
        for (int b = lane; b < 80; b += 64) {
            const int blk = b / m, j = b % m;
            cpx *f = F + blk * 4 * m + j;
            const cpx a = cmul(f[m], TW[j * fs]), bb = cmul(f[2 * m], TW[2 * j * fs]), cc = cmul(f[3 * m], TW[3 * j * fs]);
            cpx f0 = f[0];
            const cpx d = csub(f0, bb);
            f0 = cadd(f0, bb);
            const cpx e = cadd(a, cc), g = csub(a, cc);
            f[2 * m] = csub(f0, e);
            f[0] = cadd(f0, e);
            cpx o1, o3;
            o1.r = d.r + g.i; o1.i = d.i - g.r;
            o3.r = d.r - g.i; o3.i = d.i + g.r;
            f[m] = o1; f[3 * m] = o3;
        }
        __syncthreads();
    }
    // radix-5, m = 64 (src/kiss_fft.c:232-305)
    {
        const cpx ya = TW[64], yb = TW[128];
        const int u = lane;
        cpx *F0 = F + u, *F1 = F0 + 64, *F2 = F0 + 128, *F3 = F0 + 192, *F4 = F0 + 256;
        const cpx s0 = *F0;
        const cpx s1 = cmul(*F1, TW[u]), s2 = cmul(*F2, TW[2 * u]), s3 = cmul(*F3, TW[3 * u]), s4 = cmul(*F4, TW[4 * u]);
        const cpx s7 = cadd(s1, s4), s10 = csub(s1, s4), s8 = cadd(s2, s3), s9 = csub(s2, s3);
        cpx o0, s5, s6, s11, s12;
        o0.r = s0.r + (s7.r + s8.r);
        o0.i = s0.i + (s7.i + s8.i);
        s5.r = s0.r + (s7.r * ya.r + s8.r * yb.r);
        s5.i = s0.i + (s7.i * ya.r + s8.i * yb.r);
        s6.r = s10.i * ya.i + s9.i * yb.i;
        s6.i = -(s10.r * ya.i + s9.r * yb.i);
        s11.r = s0.r + (s7.r * yb.r + s8.r * ya.r);
        s11.i = s0.i + (s7.i * yb.r + s8.i * ya.r);
        s12.r = s9.i * ya.i - s10.i * yb.i;
        s12.i = s10.r * yb.i - s9.r * ya.i;
        __syncthreads();
        *F0 = o0; *F1 = csub(s5, s6); *F4 = cadd(s5, s6); *F2 = cadd(s11, s12); *F3 = csub(s11, s12);
    }
    __syncthreads();
    // autocorrelation lags 0..16 (reversed read, src/freq.c:268-272), noise floor, lag window, Levinson
    if (lane == 0 && valid) {
        float ac[LPCN_LPC_ORDER + 1], lpc[LPCN_LPC_ORDER];
        ac[0] = 320.f * F[0].r;
        for (int k = 1; k <= LPCN_LPC_ORDER; ++k) ac[k] = 320.f * F[320 - k].r;
        ac[0] = (float)((double)ac[0] + ((double)ac[0] * 1e-4 + 320 / 12 / 38.));      // src/freq.c:291
        for (int k = 1; k <= LPCN_LPC_ORDER; ++k) ac[k] = (float)((double)ac[k] * (1 - 6e-5 * k * k));
        for (int k = 0; k < LPCN_LPC_ORDER; ++k) lpc[k] = 0.f;
        float err = ac[0];
        if (ac[0] != 0.f) {
            for (int k = 0; k < LPCN_LPC_ORDER; ++k) {                                   // src/freq.c:86-127, float build
                float rr = 0.f;
                for (int j = 0; j < k; ++j) rr = rr + lpc[j] * ac[k - j];
                rr = rr + ac[k + 1];
                const float r = -rr / err;
                lpc[k] = r;
                for (int j = 0; j < (k + 1) >> 1; ++j) {
                    const float t1 = lpc[j], t2 = lpc[k - 1 - j];
                    lpc[j] = t1 + r * t2;
                    lpc[k - 1 - j] = t2 + r * t1;
                }
                err = err - (r * r) * err;
                if (err < .001f * ac[0]) break;
            }
        }
        // two-frame delay (src/lpcnet.c:110-112) + LPC_GAMMA weighting of the consumed copy (src/freq.c:299-308)
        lpcn_stream_state *st = &states[stream];
        float g = M.lpc_gamma, gi = g;
        for (int k = 0; k < LPCN_LPC_ORDER; ++k) {
            if (t + 2 < n_frames) lpc_out[((size_t)stream * n_frames + t + 2) * LPCN_LPC_ORDER + k] = lpc[k] * gi;
            if (t == n_frames - 1) st->old_lpc[0][k] = lpc[k];
            if (t == n_frames - 2) st->old_lpc[1][k] = lpc[k];
            gi *= g;
        }
    }
}

// test seam: lpcn_exp10 on the device for an array of arguments (tests/test_exp10.py sweeps it against glibc)
__global__ __launch_bounds__(256) void exp10_kernel(const float *x, double *out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = lpcn_exp10(x[i]);
}

// END2END models (src/lpcnet.c:56-80,107-108): the first 16 outputs of the conditioning network are reflection
// coefficients; step-up recursion to LPC, then the LPC_GAMMA weighting.  One thread per (stream, frame).
__global__ __launch_bounds__(64) void rc2lpc_kernel(LpcnFrameModel M, int n_items, const float *cond /*[item][128]*/, float *lpc_out /*[item][16]*/)
{
    const int item = blockIdx.x * 64 + threadIdx.x;
    if (item >= n_items) return;
    float tmp[LPCN_LPC_ORDER], ntmp[LPCN_LPC_ORDER];
#pragma unroll
    for (int i = 0; i < LPCN_LPC_ORDER; ++i) { tmp[i] = cond[(size_t)item * CN + i]; ntmp[i] = 0.f; }
#pragma unroll
    for (int i = 0; i < LPCN_LPC_ORDER; ++i) {
#pragma unroll
        for (int j = 0; j <= i - 1; ++j) ntmp[j] = tmp[j] + tmp[i] * tmp[i - j - 1];
#pragma unroll
        for (int k = 0; k <= i - 1; ++k) tmp[k] = ntmp[k];
    }
    float g = M.lpc_gamma, gi = g;
#pragma unroll
    for (int k = 0; k < LPCN_LPC_ORDER; ++k) { lpc_out[(size_t)item * LPCN_LPC_ORDER + k] = tmp[k] * gi; gi *= g; }
}

}  // namespace lpcn

static inline int lpcn_launch_frame_kernels(const LpcnFrameModel &M, hipStream_t st, int n, int n_frames, const float *d_feat,
                                            int feat_stride, size_t feat_stream_stride, lpcn_stream_state *d_state, int *d_fc_base,
                                            float *d_cond, float *d_cond_a, float *d_cond_b, float *d_lpc, char *err, size_t errlen)
{
    const size_t items = (size_t)n * n_frames;
    if (n_frames == 1) {     // one frame per stream (real-time steps, the legacy API's passes): tiles of up to 8 streams
#define LPCN_T1(STV) hipLaunchKernelGGL(lpcn::frame_cond_t1_kernel<STV>, dim3((n + STV - 1) / STV), dim3(128), 0, st, M, n, d_feat, feat_stream_stride, d_state, d_fc_base, d_cond, d_lpc)
        if (n >= 8 || n > 4) LPCN_T1(8); else if (n > 2) LPCN_T1(4); else if (n == 2) LPCN_T1(2); else LPCN_T1(1);
#undef LPCN_T1
    } else
        hipLaunchKernelGGL(lpcn::frame_cond_kernel, dim3(n), dim3(128), 0, st, M, n_frames, d_feat, feat_stride, feat_stream_stride,
                           d_state, d_fc_base, d_cond, d_lpc);
    {
#define LPCN_PJ(FTV) hipLaunchKernelGGL(lpcn::frame_proj_kernel<FTV>, dim3((unsigned)((items + FTV - 1) / FTV), items <= LPCN_PROJ_SPLIT_MAX ? LPCN_ROWS_A / 128 + 1 : 1), dim3(128), 0, st, M, items, (const float *)d_cond, d_cond_a, d_cond_b)
        if (items > 4) LPCN_PJ(lpcn::FT); else if (items > 2) LPCN_PJ(4); else if (items == 2) LPCN_PJ(2); else LPCN_PJ(1);
#undef LPCN_PJ
    }
    if (M.end2end)
        hipLaunchKernelGGL(lpcn::rc2lpc_kernel, dim3((unsigned)((items + 63) / 64)), dim3(64), 0, st, M, (int)items, (const float *)d_cond, d_lpc);
    else
    hipLaunchKernelGGL(lpcn::lpc_kernel, dim3((unsigned)((items + lpcn::LPC_WAVES - 1) / lpcn::LPC_WAVES)), dim3(64 * lpcn::LPC_WAVES), 0, st,
                       M, n, n_frames, d_feat, feat_stride, feat_stream_stride, d_state, d_lpc);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(err, errlen, "frame kernels: %s", hipGetErrorString(e)); return LPCN_E_HIP; }
    return 0;
}
