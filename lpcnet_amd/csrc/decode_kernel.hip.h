// Codec path on the device: 8-byte packets -> 4 feature vectors per packet, for every stream of a batch.
// Replaces decode_packet (src/lpcnet_dec.c:81-155) + perform_double_interp / interp_band_gain-free
// band interpolation (src/common.c:37-65).  Bit layout 7+6+3+2+10+10+10+13+3, MSB first.
// One 32-lane group per stream: lane i < 18 owns band i (the VQ memory of a band never leaves its lane),
// lane 18 writes the pitch feature, lane 19 the pitch correlation.  Packets of a stream are sequential
// (the VQ memory carries over); float expressions are written exactly like the reference's, and the
// translation unit is compiled with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "lpcnet_engine.h"

namespace lpcn {

struct DecodeTables {
    const float *cb1, *cb2, *cb3;     // [1024][17] each (ceps_codebook1..3)
    const float *cb_diff4;            // [4096][18]
    const float *pitch;               // [64] = (float)(pow(2.f, k/21.)*32), evaluated by the host libm (src/lpcnet_dec.c:107)
};

__device__ __forceinline__ unsigned dec_bits(const unsigned long long word, int &pos, const int n)
{
    const unsigned v = (unsigned)((word >> (64 - pos - n)) & ((1ull << n) - 1ull));
    pos += n;
    return v;
}

__global__ __launch_bounds__(64) void decode_kernel(DecodeTables T, const unsigned char *__restrict__ packets, int n_streams,
                                                    int n_packets, float *__restrict__ vq_mem, float *__restrict__ feat, int feat_stride)
{
    const int stream = blockIdx.x * 2 + (threadIdx.x >> 5), lane = threadIdx.x & 31;
    if (stream >= n_streams) return;
    const int nb = LPCN_NB_BANDS;
    float mem = lane < nb ? vq_mem[(size_t)stream * nb + lane] : 0.f;
    for (int p = 0; p < n_packets; ++p) {
        const unsigned char *buf = packets + ((size_t)stream * n_packets + p) * 8;
        unsigned long long word = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) word = (word << 8) | buf[k];
        int pos = 0;
        const int c0_id = (int)dec_bits(word, pos, 7), main_pitch = (int)dec_bits(word, pos, 6);
        int modulation = (int)dec_bits(word, pos, 3);
        const int corr_id = (int)dec_bits(word, pos, 2);
        const int e0 = (int)dec_bits(word, pos, 10), e1 = (int)dec_bits(word, pos, 10), e2 = (int)dec_bits(word, pos, 10);
        int vq_mid = (int)dec_bits(word, pos, 13), interp_id = (int)dec_bits(word, pos, 3);
        int voiced = 1;
        modulation -= 4;
        if (modulation == -4) { voiced = 0; modulation = 0; }
        float *out = feat + ((size_t)stream * n_packets * 4 + (size_t)p * 4) * feat_stride;
        if (lane < nb) {
            const int i = lane;
            float f3;
            if (i == 0) f3 = (c0_id - 64) / 4.f;
            else f3 = T.cb1[e0 * 17 + i - 1] + T.cb2[e1 * 17 + i - 1] + T.cb3[e2 * 17 + i - 1];
            float sign = 1;
            if (vq_mid >= 4096) { vq_mid -= 4096; sign = -1; }
            float f1 = sign * T.cb_diff4[vq_mid * nb + i];
            if ((vq_mid & 3) < 2) f1 += .5f * (mem + f3);
            else if ((vq_mid & 3) == 2) f1 += mem;
            else f1 += f3;
            interp_id += (interp_id >= 7);
            const int ma = interp_id / 3, mb = interp_id % 3;
            const float f0 = ma == 0 ? .5f * (mem + f1) : (ma == 1 ? mem : f1);
            const float f2 = mb == 0 ? .5f * (f1 + f3) : (mb == 1 ? f1 : f3);
            out[0 * feat_stride + i] = f0;
            out[1 * feat_stride + i] = f1;
            out[2 * feat_stride + i] = f2;
            out[3 * feat_stride + i] = f3;
            mem = f3;
        } else if (lane == nb || lane == nb + 1) {
            const float frame_corr = voiced ? 0.3875f + .175f * corr_id : 0.0375f + .075f * corr_id;
#pragma unroll
            for (int sub = 0; sub < 4; ++sub) {
                float pp = T.pitch[main_pitch];
                pp *= 1.f + modulation / 16.f / 7.f * (2 * sub - 3);
                if (pp < 33) pp = 33;
                if (pp > 255) pp = 255;
                out[sub * feat_stride + lane] = lane == nb ? .02f * (pp - 100.f) : frame_corr - .5f;
            }
        } else if (lane < feat_stride) {
#pragma unroll
            for (int sub = 0; sub < 4; ++sub) out[sub * feat_stride + lane] = 0.f;
        }
    }
    if (lane < nb) vq_mem[(size_t)stream * nb + lane] = mem;
}

}  // namespace lpcn
