// Persistent per-sample kernel, EIGHT float streams per CU as two groups of four that run HALF A STEP APART (gfx950 / CDNA4).
//
// Same path and same arithmetic contract as sample_kernel.hip.h (PARITY: the reference's generic-C order, src/vec.h:347-404,
// src/nnet.c:326-372,410-448,163-214, src/lpcnet.c:146-167,235-271; bit-identical to the generic-C float build) -- a different
// SCHEDULE.  The four-stream kernel's step is one stream-sample's dependency chain (leader -> gather -> update / reset rows ->
// gates -> GRU-B -> tree: 17.3 k of its 19.5 k clk) and its vector units sit idle 39 % of the time; the only independent work
// that could fill them is OTHER STREAMS.  Here one workgroup (8 waves, 256 VGPRs: GRU-A's weights stay register-resident and
// are shared by all eight streams, GRU-B's weights stay in LDS and are shared too) carries groups X and Y of four streams each.
// The sample of a group is the four phases P1 (GRU-A rows) | P2 (gates) | P3 (GRU-B chains, candidate heads of the next sample) |
// P4 (tree); the workgroup runs HALF-STEPS of two intervals and two barriers:
//
//      interval A:  P3 of the one group (its GRU-B chains on waves 0..3, its candidate heads on the head waves)
//                   followed, on every wave, by P1 of the OTHER group, then P4 (the tree) of the first one   | barrier
//      interval B:  P2 of that other group                                                                   | barrier, the groups swap roles
//
// so a wave is never short of work that does not depend on the chain it has just fed: the leader / gather latency of one group's
// P1 is covered by the other group's chain and heads, and the waves that used to wait ~2 k clk for GRU-B run the other group's
// rows.  The two phases of an interval belong to different groups, hence touch different LDS cells; the update / reset
// pre-activations and the candidate inputs are dead outside P1..P2 of their group, so ONE copy serves both groups, and the frame
// conditioning (1152 x 4 floats per group and frame) is read from L2 by the lanes that need it (one row = one dword per stream and
// sample) instead of living in LDS: 152.7 KB of the CU's 160.
//
// Scope: float blobs, PARITY arithmetic, dense GRU-B input matrix, <= 32 items per lane.  Everything else runs on the four-stream kernel.
#pragma once
#include "sample_kernel.hip.h"

namespace lpcn {

#define LPCN_X2_LW 4            // the wave that leads the streams (tree walk, LPC predictor, mu-law): a head wave (model_pack.c gives it the shortest candidate slot) -- waves 0..3 start GRU-B's chains at once
#define LPCN_X2_TW 0            // the wave that draws the KISS99 thresholds: a chain wave -- they have the most slack at barrier 1 (wave 0 / 1 / 3 / 5 / 6: 149.5 / 148.9 / 145.0 / 147.4 / 143.5 M; leader on wave 5 / 7: 139.3 / 138.2 M against 146.7 M on wave 4, round 6)
#define LPCN_X2_HG 10           // head items a row wave runs before it polls the leader's indices for the start-value pass (6 / 10 / 14 / 18: 145.0 / 146.7 / 146.2 / 144.3 M)

struct LdsX2 {
    static constexpr int S = 4;
    static constexpr int HA_STRIDE = 16 * S;
    // ---- one block per group, identical layout
    static constexpr int g_hA    = 0;                               // [96 blocks][4 streams][4] + 16 B pad per 4 blocks: the GRU-A state for the items and GRU-B
    static constexpr int g_hT    = g_hA + 96 * HA_STRIDE + 24 * 16; // [384][4] the same, neuron-major (gates, start values)
    static constexpr int g_prec  = g_hT + NA * S * 4;               // [384][4] candidate pre-activations / the parked sums of their heads
    static constexpr int g_hB    = g_prec + NA * S * 4;             // [4][16]
    static constexpr int g_idx   = g_hB + S * NB * 4;               // [4] packed indices, [4] live flags
    static constexpr int g_thr   = g_idx + S * 16;                  // [4][8]
    static constexpr int g_mask  = g_thr + S * 32;                  // [4][8] u64
    static constexpr int g_lead  = g_mask + S * 64;                 // [4][8]
    static constexpr int g_flag  = g_lead + S * 32;                 // [4] i32
    static constexpr int g_condb = g_flag + 16;                     // [4][48]
    static constexpr int g_lpc   = g_condb + S * RB * 4;            // [4][16]
    static constexpr int g_pcm   = g_lpc + S * 64;                  // [4][160] i16
    static constexpr int G_SZ    = g_pcm + S * 320;
    // ---- shared by both groups
    static constexpr int pre_ur  = 2 * G_SZ;                        // [768][4] update / reset pre-activations of the group that is in P1..P2
    static constexpr int inh     = pre_ur + 2 * NA * S * 4;         // [384][4] input part of its candidate rows
    static constexpr int abias   = inh + NA * S * 4;                // [1152]{bias, diag}
    static constexpr int tansig  = abias + 2 * RA * 4;
    static constexpr int ulaw    = tansig + 816;
    static constexpr int logit   = ulaw + 1024;
    static constexpr int brec    = logit + 1024;                    // [16][48]
    static constexpr int bbias   = brec + NB * RB * 4;              // [2][48]
    static constexpr int bstart  = bbias + 2 * RB * 4;              // [8] i32
    static constexpr int bw      = bstart + 32;                     // [nb_b + 8][8][4] f32
    static constexpr int total(int nb_b) { return bw + (nb_b + 8) * 128; }
    __host__ __device__ static constexpr int ha_off(int p) { return p * HA_STRIDE + (p >> 2) * 16; }
};

template <int NW>
__global__ __launch_bounds__(LPCN_WG_THREADS, 2) void sample_kernel_x2(const LpcnSampleArgs *__restrict__ Ap)
{
    using L = LdsX2;
    constexpr int S = 4;
    constexpr int LW = LPCN_X2_LW, TW = LPCN_X2_TW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *const sm_pre_ur = (float *)(smem + L::pre_ur);
    float *const sm_inh = (float *)(smem + L::inh);
    const float *const sm_abias = (const float *)(smem + L::abias);
    const float *const sm_tansig = (const float *)(smem + L::tansig);
    const float *const sm_ulaw = (const float *)(smem + L::ulaw);
    const float *const sm_logit = (const float *)(smem + L::logit);
    const float *const sm_brec = (const float *)(smem + L::brec);
    const float *const sm_bbias = (const float *)(smem + L::bbias);
    const int *const sm_bstart = (const int *)(smem + L::bstart);

    const int tid0 = threadIdx.x;
    const int n_streams = Ap->n_streams, n_frames = Ap->n_frames, preload = Ap->preload, frame_len = Ap->frame_len;
    const int s0 = blockIdx.x * 2 * S;                      // first stream of this workgroup; group g holds streams s0 + 4 g + {0..3}
    auto stream_of = [&](int gs) __attribute__((always_inline)) { return (s0 + gs < n_streams) ? s0 + gs : n_streams - 1; };      // streams past the end: a clamped copy, never written back
    const int n_valid = (n_streams - s0 < 2 * S) ? n_streams - s0 : 2 * S;
    const size_t nf = (size_t)n_frames;
    auto *const states = as_global_rw(Ap->state);
    const LPCN_GLOBAL float *emb_nat_sig = as_global(Ap->emb_nat_sig), *emb_nat_pred = as_global(Ap->emb_nat_pred), *emb_nat_exc = as_global(Ap->emb_nat_exc);      // [256][1152]
    asm volatile("" : "+s"(emb_nat_sig), "+s"(emb_nat_pred), "+s"(emb_nat_exc));

    // ------------------------------------------------------------------ resident weights ----
    float4 w[NW];
    uint32_t offp[(NW + 1) / 2];
    int row_reg[3];
#define LPCN_ROW(k) (row_reg[k])
    {
        const int lane = tid0 & 63, wave = tid0 >> 6;
        const size_t base = (size_t)wave * NW * 64 + lane;
        const int lane_sel = (lane & 3) * 16;
        const auto *ab = as_global(Ap->a_blk);
        const auto *aw = (const LPCN_GLOBAL float *)as_global(Ap->a_w);
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const auto *v = aw + (base + (size_t)j * 64) * 4;
            w[j] = make_float4(v[0], v[1], v[2], v[3]);
        }
#pragma unroll
        for (int j = 0; j < NW; j += 2) {
            const int p0 = ab[base + (size_t)j * 64];
            const int p1 = (j + 1 < NW) ? ab[base + (size_t)(j + 1) * 64] : 0;
            offp[j >> 1] = (uint32_t)(L::ha_off(p0) + lane_sel) | ((uint32_t)(L::ha_off(p1) + lane_sel) << 16);
        }
        const auto *ar = as_global(Ap->a_row);
#pragma unroll
        for (int k = 0; k < 3; ++k) row_reg[k] = ar[(wave * 3 + k) * 64 + lane];
    }
    int b1 = __builtin_amdgcn_readfirstlane(as_global(Ap->a_bound)[(tid0 >> 6) * 4 + 1]);
    int b2 = __builtin_amdgcn_readfirstlane(as_global(Ap->a_bound)[(tid0 >> 6) * 4 + 2]);
    int b3 = __builtin_amdgcn_readfirstlane(as_global(Ap->a_bound)[(tid0 >> 6) * 4 + 3]);   // this wave's item count
    const bool allh0 = __builtin_amdgcn_readfirstlane(as_global(Ap->a_allh)[(tid0 >> 6) * 3]) != 0;
    const LPCN_GLOBAL float *fc_w_s = as_global(Ap->fc_w), *fc_b_s = as_global(Ap->fc_b), *fc_f_s = as_global(Ap->fc_f);
    asm volatile("" : "+s"(fc_w_s), "+s"(fc_b_s), "+s"(fc_f_s));
    const LPCN_GLOBAL float *cond_a_s = as_global(Ap->cond_a);
    asm volatile("" : "+s"(cond_a_s));
    const int has_slot = __builtin_amdgcn_readfirstlane((__ballot(row_reg[0] >= 0) != 0ull ? 1 : 0) | (__ballot(row_reg[1] >= 0) != 0ull ? 2 : 0) |
                                                        (__ballot(row_reg[2] >= 0) != 0ull ? 4 : 0));
    const bool has2 = (has_slot & 4) != 0;
    const int hl = __builtin_amdgcn_readfirstlane(as_global(Ap->a_head)[tid0 >> 6]);
    const bool early_wave = hl > 0;                          // wave-uniform: this wave computes the head of its candidate slot one sample ahead
    // wave-uniform: no row of this wave starts from bias + diag*h formed in P1 -- slot 0 continues from its head's partial sums (or holds update / reset rows only), slots 1, 2
    // hold no candidate rows: the slot start is then ONE cell read (the row waves of the benchmark model's dealing)
    const bool plain_start = __builtin_amdgcn_readfirstlane(((early_wave || __ballot(row_reg[0] >= 2 * NA) == 0ull) && __ballot(row_reg[1] >= 2 * NA || row_reg[2] >= 2 * NA) == 0ull) ? 1 : 0) != 0;

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
            ((float *)(smem + L::abias))[2 * i] = ab1[i];
            ((float *)(smem + L::abias))[2 * i + 1] = adg[i];
        }
        const auto *br = as_global(Ap->b_rec), *bb = as_global(Ap->b_bias);
        for (int i = tid; i < NB * RB; i += LPCN_WG_THREADS) ((float *)(smem + L::brec))[i] = br[i];
        for (int i = tid; i < 2 * RB; i += LPCN_WG_THREADS) ((float *)(smem + L::bbias))[i] = bb[i];
        if (tid < 7) ((int *)(smem + L::bstart))[tid] = as_global(Ap->b_start)[tid];
        const auto *bwg = as_global(Ap->b_w);
        // (row groups 2, 3 and 5 shifted by one more block: the six groups' rows then cover both halves of the banks -- see sample_kernel.hip.h)
        for (int i = tid; i < (nb_b + 8) * 32; i += LPCN_WG_THREADS) {
            int di;
            if (i < nb_b * 32) di = i + ((0x321100 >> (4 * ((i >> 5) / 96))) & 15) * 32;
            else if (i < (nb_b + 5) * 32) di = i + 3 * 32;
            else continue;
            ((uint32_t *)(smem + L::bw))[di] = i < nb_b * 32 ? ((const LPCN_GLOBAL uint32_t *)bwg)[i] : 0u;
        }
        for (int i = tid; i < 2 * S * NA; i += LPCN_WG_THREADS) {
            const int gs = i / NA, n = i % NA, g = gs >> 2, s = gs & 3;
            const float hv0 = states[stream_of(gs)].gru_a[n];
            unsigned char *gb = smem + g * L::G_SZ;
            ((float *)(gb + L::g_hT))[n * S + s] = hv0;
            *(float *)(gb + L::g_hA + L::ha_off(n >> 2) + s * 16 + (n & 3) * 4) = hv0;
        }
        for (int i = tid; i < 2 * S * NB; i += LPCN_WG_THREADS) {
            const int gs = i / NB, g = gs >> 2, s = gs & 3;
            ((float *)(smem + g * L::G_SZ + L::g_hB))[s * NB + i % NB] = states[stream_of(gs)].gru_b[i % NB];
        }
        if (tid < 2 * S) {
            const int g = tid >> 2, s = tid & 3;
            const auto *st = &states[stream_of(tid)];
            unsigned char *gb = smem + g * L::G_SZ;
            ((int *)(gb + L::g_idx))[s] = 0;
            float *ld = (float *)(gb + L::g_lead);
            ld[s * 8 + 0] = 0.f;
            ld[s * 8 + 1] = st->deemph_mem;
            ((int *)ld)[s * 8 + 2] = st->last_exc;
#pragma unroll
            for (int j = 0; j < 4; ++j) ((uint32_t *)ld)[s * 8 + 4 + j] = st->rng[j];
        }
        if (tid < 2) { int *fl = (int *)(smem + tid * L::G_SZ + L::g_flag); fl[0] = 0; fl[1] = 0; fl[2] = 0; }
    }
    __syncthreads();

    // ---- leader state: wave LW, lane 16 s + j holds sample j of stream s's LPC history, one register per group
    const bool is_lw = (tid0 >> 6) == LW;                    // (wave-uniform)
    const bool is_tw_lane = tid0 >= 64 * TW && tid0 < 64 * TW + S;
#define LPCN_LROW ((tid0 & 63) >> 4)
#define LPCN_TAP (tid0 & 15)
    // "P" variables belong to the group that runs P1 / P2 in the current half-step, "Q" to the one that runs P3 / P4; they swap at its end.
    // The loop starts at h = -1 with P = group 1, Q = group 0.
    float histP = is_lw ? states[stream_of(S + LPCN_LROW)].last_sig[LPCN_TAP] : 0.f;
    float histQ = is_lw ? states[stream_of(LPCN_LROW)].last_sig[LPCN_TAP] : 0.f;
    bool liveP = false, liveQ = false;                       // per lane: leader lanes (their stream), threshold lanes
    int live_maskP = 0, live_maskQ = 0;                      // bit s: stream s of the group produces samples in its current frame
    int seqP = 0, seqQ = 0;                                  // samples opened so far, per group (identical in every wave)
    int chnP = 0, chnQ = 0;                                  // GRU-B phases run so far, per group
    int smpP = 0, smpQ = 0, fP = 0, fQ = 0;                  // position of the group's NEXT P1 sample: sample within the frame, frame
    float lpc_tap = 0.f, prod_old = 0.f;                     // leader lanes: computed behind the tree of a group, used when its sample is finished
    auto row_shr1 = [](float v, float fill) __attribute__((always_inline)) {               // value of the previous lane of the 16-lane row; lane 0 gets `fill`
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill), __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, false));
    };
    auto lds_addr = [](const void *ptr) __attribute__((always_inline)) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)(unsigned char *)ptr; };

    const int T = n_frames * frame_len;                      // samples per stream in this launch
#if LPCN_ENABLE_PROF
#ifndef LPCN_PROF_MASK
#define LPCN_PROF_MASK 0xFFF          // which of the 12 slots are compiled in (a run with only the two barrier waits, 0x480, perturbs the kernel least)
#endif
    // per-phase shader-clock accounting of workgroup 0 (profiling builds only), clk summed over the half-steps of the launch:
    // 0 leader / thresholds / frame boundary, 1 GRU-B mat-vec, 2 GRU-B gates, 3 candidate heads, 4 P1 start values (wait for the indices, gather, cond),
    // 5 P1 items, 6 P1 close, 7 wait at barrier 1, 8 dual-FC prefetch + GRU-A gate stage, 9 tree, 10 wait at barrier 2
    unsigned long long *const prof = Ap->prof;
    unsigned long long pt[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
    const bool profiling = prof != nullptr && blockIdx.x == 0;
#define LPCN_X2_PROF(slot) do { if (((LPCN_PROF_MASK >> (slot)) & 1) && profiling) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); pt[slot] += now_ - tprev; tprev = now_; } } while (0)
    if (profiling) tprev = __builtin_amdgcn_s_memtime();
#else
#define LPCN_X2_PROF(slot) do { } while (0)
#endif
    // ====================================================================== half-steps ======
    for (int h = -1; h <= 2 * T + 1; ++h) {
        const int p = h & 1, q = p ^ 1;
        unsigned char *const gp = smem + p * L::G_SZ, *const gq = smem + q * L::G_SZ;
        const bool p_active = h >= 0 && (h >> 1) < T;        // group P starts a sample in this half-step
        const bool p_prev = h >= 2;                          // group P's previous sample has been through its tree: the leader finishes it now
        const bool q_chain = h >= 1 && ((h - 1) >> 1) < T;   // group Q has a sample in GRU-B / the tree
        const bool q_heads = ((h + 1) >> 1) < T;             // group Q starts another sample in the next half-step: its candidate heads run now
        const bool new_frame = p_active && smpP == 0;
        const bool more = p_active && smpP != 0;             // P's new sample continues the frame of the one just finished
        float *const hT_p = (float *)(gp + L::g_hT);
        float *const hB_q = (float *)(gq + L::g_hB);
        int *const idx_p = (int *)(gp + L::g_idx);
        float *const lead_p = (float *)(gp + L::g_lead);
        float *const thr_p = (float *)(gp + L::g_thr);
        float *const lpc_p = (float *)(gp + L::g_lpc);
        short *const pcm_p = (short *)(gp + L::g_pcm);
        const uint32_t flag_p = lds_addr(gp + L::g_flag);

        // prediction + mu-law indices of group P's next sample (wave LW; src/lpcnet.c:252-254), published through idx_p + flag_p
        auto open_sample = [&](const bool live, const float newest, const float prod, const int exc, const bool per_frame) __attribute__((always_inline)) {
            int t_ = tid0;
            LPCN_REMAT_V(t_);
            const int lrow = (t_ & 63) >> 4, tap = t_ & 15;
            const float r = lpc_chain<0>(0.f, prod);
            const int u = lpcn_lin2ulaw((tap & 1) ? r : newest);
            const int u_pred = __builtin_amdgcn_mov_dpp(u, 0xB1, 0xf, 0xf, true);
            if (tap == 0) {
                if (live) {
                    lead_p[lrow * 8 + 0] = r;
                    idx_p[lrow] = u | (u_pred << 8) | (exc << 16);
                } else {
                    idx_p[lrow] = 0;
                }
                if (per_frame) idx_p[S + lrow] = live ? 1 : 0;
            }
        };
        auto publish_indices = [&]() __attribute__((always_inline)) { asm volatile("ds_write_b32 %0, %1" :: "v"(flag_p), "v"(seqP) : "memory"); };
        auto draw_thresholds = [&](const int ls) __attribute__((always_inline)) {           // src/nnet.c:178-184
            int *li = (int *)lead_p + ls * 8;
            uint32_t rng[4] = {(uint32_t)li[4], (uint32_t)li[5], (uint32_t)li[6], (uint32_t)li[7]};
            const uint32_t r0 = lpcn_kiss99(rng), r1 = lpcn_kiss99(rng);
            li[4] = (int)rng[0]; li[5] = (int)rng[1]; li[6] = (int)rng[2]; li[7] = (int)rng[3];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                thr_p[ls * 8 + b] = sm_logit[(r0 >> (8 * b)) & 0xFF];
                thr_p[ls * 8 + 4 + b] = sm_logit[(r1 >> (8 * b)) & 0xFF];
            }
        };

        // ------------------------------------------------ the leader finishes group P's previous sample --
        if (more) ++seqP;
        if (p_prev && is_lw) {
            __builtin_amdgcn_s_setprio(3);
            const int smp_done = smpP == 0 ? frame_len - 1 : smpP - 1;      // index of the finished sample in its frame
            int t_ = tid0;
            LPCN_REMAT_V(t_);                                // (lane-derived indices are rebuilt here: hoisted out of the loop they are spilled, and a scratch reload on the leader's path is ~0.5 k clk)
            const int lrow = (t_ & 63) >> 4, tap = t_ & 15;
            auto walk_tree = [&](int lrow_) __attribute__((always_inline)) {                // the sampler's 8 decisions from the 255 ballot bits of stream row lrow_
                typedef unsigned u4 __attribute__((ext_vector_type(4)));
                const u4 *mk = (const u4 *)((const unsigned long long *)(gp + L::g_mask) + lrow_ * 8);
                const u4 qa = mk[0], qb = mk[1], qc = mk[2], qd = mk[3];
                auto bit_of = [](unsigned word, int k) __attribute__((always_inline)) { return (int)((word >> (2 * k)) & 1u); };
                int val = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) val = (val << 1) | bit_of(qa[0], (1 << b) | val);
                val = (val << 1) | bit_of(qa[1], val);
                val = (val << 1) | bit_of((val & 16) ? qa[3] : qa[2], val & 15);
                {
                    const int k = val >> 4;
                    const unsigned lo = (k & 1) ? qb[1] : qb[0], hi = (k & 1) ? qb[3] : qb[2];
                    val = (val << 1) | bit_of((k & 2) ? hi : lo, val & 15);
                }
                {
                    const int k = val >> 4;
                    const unsigned a0 = (k & 1) ? qc[1] : qc[0], a1 = (k & 1) ? qc[3] : qc[2];
                    const unsigned a2 = (k & 1) ? qd[1] : qd[0], a3 = (k & 1) ? qd[3] : qd[2];
                    const unsigned b0 = (k & 2) ? a1 : a0, b1_ = (k & 2) ? a3 : a2;
                    val = (val << 1) | bit_of((k & 4) ? b1_ : b0, val & 15);
                }
                return val;
            };
            float pcm = 0.f, deemph = 0.f;
            int exc = 0;
            if (liveP) {
                const float pred = lead_p[lrow * 8 + 0];
                deemph = lead_p[lrow * 8 + 1];
                exc = walk_tree(lrow);
                if (smp_done < preload) {                                       // src/lpcnet.c:256-258
                    const float x = (float)pcm_p[lrow * LPCN_FRAME_SIZE + smp_done];
                    exc = lpcn_lin2ulaw(x - 0.85f * deemph - pred);
                    pcm = x - 0.85f * deemph;
                } else {
                    pcm = pred + sm_ulaw[exc];                                  // src/lpcnet.c:260
                }
            }
            {
                const float shifted = row_shr1(histP, pcm);                     // src/lpcnet.c:262-263
                histP = liveP ? shifted : histP;
            }
            if (tap == 0 && liveP) ((int *)lead_p)[lrow * 8 + 2] = exc;
            if (more) { open_sample(liveP, pcm, tap == 0 ? pcm * lpc_tap : prod_old, exc, false); publish_indices(); }
            __builtin_amdgcn_s_setprio(0);
            if (tap == 0) {
                if (liveP) {
                    pcm = pcm + 0.85f * deemph;
                    lead_p[lrow * 8 + 1] = pcm;
                    if (smp_done >= preload) pcm_p[lrow * LPCN_FRAME_SIZE + smp_done] = (short)lpcn_round_pcm(pcm);
                } else {
                    pcm_p[lrow * LPCN_FRAME_SIZE + smp_done] = 0;
                }
            }
        }
        if (more && is_tw_lane && liveP) { int t_ = tid0; LPCN_REMAT_V(t_); draw_thresholds(t_ - 64 * TW); }

        // ------------------------------------------------ group P enters a new frame --------------------
        if (new_frame) {
            int tid = tid0;
            LPCN_REMAT_V(tid);                                // (lane-derived values are rebuilt here: hoisted out of the half-step loop they are spilled)
            const bool lw_ = (tid >> 6) == LW, twl_ = tid >= 64 * TW && tid < 64 * TW + S;
            __syncthreads();                                  // the leader's last sample of the previous frame
            if (fP > 0) {                                     // flush the finished frame's PCM (4 x 160 samples, coalesced)
                auto *out = as_global_rw(Ap->pcm);
                const size_t pstride = (size_t)Ap->pcm_stride;
                for (int i = tid; i < S * LPCN_FRAME_SIZE; i += LPCN_WG_THREADS) {
                    const int s = i / LPCN_FRAME_SIZE, k = i % LPCN_FRAME_SIZE;
                    if (S * p + s < n_valid && k < frame_len) out[(size_t)(s0 + S * p + s) * pstride + (size_t)(fP - 1) * LPCN_FRAME_SIZE + k] = pcm_p[i];
                }
            }
            __syncthreads();                                  // (preload below overwrites the buffer)
            {
                const auto *cb = as_global(Ap->cond_b), *lp = as_global(Ap->lpc);
                float *const condb_p = (float *)(gp + L::g_condb);
                if (tid < S * RB) condb_p[tid] = cb[((size_t)stream_of(S * p + tid / RB) * nf + fP) * RB + tid % RB];
                if (tid >= 256 && tid < 256 + S * LPCN_LPC_ORDER) {
                    const int i = tid - 256;
                    lpc_p[i] = lp[((size_t)stream_of(S * p + i / LPCN_LPC_ORDER) * nf + fP) * LPCN_LPC_ORDER + i % LPCN_LPC_ORDER];
                }
                if (lw_ || twl_) {
                    const int lstream = stream_of(S * p + (lw_ ? (tid & 63) >> 4 : tid - 64 * TW));
                    const int fc_ref = Ap->fc_base ? as_global(Ap->fc_base)[lstream] : states[lstream].frame_count;
                    int fc = Ap->fc_advance ? fc_ref + fP + 1 : fc_ref;
                    if (fc > 1000) fc = 1000;
                    liveP = fc > LPCN_FEATURES_DELAY;         // src/lpcnet.c:239-243
                }
                if (preload > 0 && tid < S) {
                    const auto *pin = as_global(Ap->pcm) + (size_t)stream_of(S * p + tid) * (size_t)Ap->pcm_stride + (size_t)fP * LPCN_FRAME_SIZE;
                    for (int i = 0; i < preload; ++i) pcm_p[tid * LPCN_FRAME_SIZE + i] = pin[i];
                }
            }
            __syncthreads();                                  // lpc_p visible to the leaders
            ++seqP;
            if (lw_) {
                open_sample(liveP, histP, histP * lpc_p[tid & 63], ((const int *)lead_p)[((tid & 63) >> 4) * 8 + 2], true);
                publish_indices();
            }
            if (twl_ && liveP) draw_thresholds(tid - 64 * TW);
            __syncthreads();
            int lm = 0;
#pragma unroll
            for (int s = 0; s < S; ++s) lm |= (idx_p[S + s] ? 1 : 0) << s;
            live_maskP = __builtin_amdgcn_readfirstlane(lm);
        }

        LPCN_X2_PROF(0);
        // ================================================================ interval A ===========
        // ---- shared pieces of the item chains (P1 of group P, candidate heads of group Q)
        float acc[S] = {};
        constexpr int PF = 2;                                // (state blocks fetched 1 / 2 / 3 / 4 items ahead: 150.0 / 151.0 / 149.6 / 148.0 M)
        float4 hq[PF + 1] = {};
        const unsigned char *hA_cur = gp + L::g_hA;           // state blocks of the group whose items are running
        auto fetch_h = [&](const int j) __attribute__((always_inline)) {
            uint32_t pk = offp[j >> 1];
            LPCN_REMAT_V(pk);
            const uint32_t off = (j & 1) ? (pk >> 16) : (pk & 0xFFFFu);
            hq[j % (PF + 1)] = *(const float4 *)(hA_cur + off);
        };
        typedef float negz_t __attribute__((ext_vector_type(4)));
        negz_t negz = {-0.f, -0.f, -0.f, -0.f};
        auto load_negz = [&]() __attribute__((always_inline)) { negz = (negz_t){-0.f, -0.f, -0.f, -0.f}; asm volatile("" : "+v"(negz)); };
        // one item = (this lane's row) x (one 4-wide input block) for the four streams of a group: the products of a column from ONE
        // v_mfma_f32_4x4x1 with C = -0.0 (bit for bit the separately rounded product), the sums as v_pk_add_f32 over stream pairs, columns
        // 0..3 in order (src/vec.h:355-401)
        auto mac = [&](const int j) __attribute__((always_inline)) {
            typedef float f4 __attribute__((ext_vector_type(4)));
            typedef float f2 __attribute__((ext_vector_type(2)));
            const float4 hv = hq[j % (PF + 1)];
            const float hk[4] = {hv.x, hv.y, hv.z, hv.w};
            const float wk[4] = {w[j].x, w[j].y, w[j].z, w[j].w};
            f2 a01 = {acc[0], acc[1]}, a23 = {acc[2], acc[3]};
            f4 pv[4];
            // (the four products of an item are issued back to back into four result tuples, then the sums.  Round 6 tried to software-pipeline the
            // items by one -- the matrix-pipe instructions of item j + 1 between / behind the adds of item j, in one or in two sets of product
            // registers: bit-exact and far slower, 113.8 vs 145.6 M samples/s -- an MFMA between dependent packed adds stalls both)
#pragma unroll
            for (int c = 0; c < 4; ++c) pv[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(hk[c], wk[c], negz, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                a01 = a01 + __builtin_shufflevector(pv[c], pv[c], 0, 1);
                a23 = a23 + __builtin_shufflevector(pv[c], pv[c], 2, 3);
            }
            acc[0] = a01[0]; acc[1] = a01[1]; acc[2] = a23[0]; acc[3] = a23[1];
        };
        // pre-activation cell of GRU-A row r: update / reset rows share one copy, candidate rows have one per group
        auto pre_cell = [&](const int r, unsigned char *gb) __attribute__((always_inline)) -> float * {
            return (float *)(r < 2 * NA ? smem + L::pre_ur + r * (S * 4) : gb + L::g_prec + (r - 2 * NA) * (S * 4));
        };

        const int lane = tid0 & 63;
        const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
        // this lane's dual-FC row (node = tid >> 1, channel = tid & 1) for group Q's tree, fetched behind the wave's items of P (it lands while the slots are closed)
        float fcw[NB], fcb = 0.f, fcf = 0.f;
        auto load_fc = [&]() __attribute__((always_inline)) {
            int t_ = tid0;
            LPCN_REMAT_V(t_);
            const int node_ = t_ >> 1, chan_ = t_ & 1;
            const auto *fcw_ptr = fc_w_s + node_ * 2 * NB + chan_ * NB;
#pragma unroll
            for (int j = 0; j < NB; ++j) fcw[j] = fcw_ptr[j];
            fcb = fc_b_s[chan_ * 256 + node_]; fcf = fc_f_s[chan_ * 256 + node_];
        };
        const uint32_t chcnt_q = lds_addr(gq + L::g_flag) + 8;   // arrival counter of Q's GRU-B chains (the tree of a wave must not start before all four have written their state)
        if (q_chain) ++chnQ;

        // Order of an interval on one wave (round 6, third form.  The first ran P3 of Q, then all of P1 of P, and every wave then sat ~4 k clk behind
        // its own embedding gather; the second issued each wave's gather before Q's chain / heads and the 60 values in flight pushed GRU-A's weights
        // into scratch -- profiles/r06_x2_phase_v1.txt):
        //   2. head waves: the first LPCN_X2_HG items of Q's candidate heads (the leader needs ~1.5 k clk to publish P's indices)
        //   3. waves 4..7: poll the indices, then the START VALUES of ALL of P's GRU-A rows as one element-wise pass ("P0"): row r of the natural
        //      [256][1152] tables per lane -- every load a contiguous 256 B per wave, the conditioning row straight from cond_a -- written to the
        //      rows' pre-activation cells (update / reset rows) or the candidate inputs; an arrival counter tells the row owners
        //   4. Q's GRU-B chains (waves 0..3) / the rest of Q's heads
        //   5. P's items from the parked start values, close.
        // The waves that carry GRU-B's chains -- the longest link of an interval -- neither gather nor wait for a gather.
        // ---------------------------------------------------------------- 2..4: P3 of group Q around P's start-value pass ----
        // The HEAD of group Q's next candidate chains: the first `hl` blocks of every row of this wave's candidate slot (items [NW - hl, NW)) from
        // bias + diag*h -- final once Q's gate stage is done -- parked in the rows' pre-activation cells; the slot continues from there in Q's next P1.
        constexpr int J0 = NW - LPCN_EARLY_MAX < 0 ? 0 : NW - LPCN_EARLY_MAX;
        constexpr int JM = J0 + LPCN_X2_HG < NW ? J0 + LPCN_X2_HG : NW;
        const bool do_heads = early_wave && q_heads;
        int e0 = NW - hl;
        LPCN_REMAT_S(e0);
        auto head_step = [&](auto self, auto jc, auto jend_c) __attribute__((always_inline)) -> void {
            constexpr int j = decltype(jc)::value, je = decltype(jend_c)::value;
            if constexpr (j < je) {
                if constexpr (j + PF < NW) fetch_h(j + PF);
                if (j >= e0) { asm volatile(""); mac(j); }   // (every step issues the same LDS read whether its item runs or not: exact wait counts)
                self(self, std::integral_constant<int, j + 1>{}, jend_c);
            }
        };
        const uint32_t p0cnt_p = flag_p + 4;                 // arrival counter of P's start-value pass
        // ---- 3: P0 of group P on waves 4..7, in five stages that are interleaved with the rest of Q's heads (the loads of a round land while head items run)
        constexpr int CW = LPCN_X2_CHAIN_WAVES, P0W = LPCN_X2_P0_FIRST, NP0 = LPCN_WAVES - P0W;      // waves 0..3 carry GRU-B's chains (one stream each), waves 4..7 the heads and the start-value pass
        static_assert(NP0 * 64 * 3 == 2 * NA && CW == S, "three rounds of the row waves' lanes cover GRU-A's update / reset rows");
        const bool p0_wave = p_active && wave >= P0W;
        uint32_t o_sig[S] = {}, o_pred[S] = {}, o_exc[S] = {}, o_cond[S] = {};      // byte offsets of the streams' table rows (scalar) -- the lane adds its row
        constexpr int P0PERM = 0x3102;                       // waves 4, 5, 6, 7 take lanes 128.., 0.., 64.., 192.. of the pass: the half round on waves 5, 6 (on 6, 7 / 5, 7 / 4, 5: 149.5 / 149.8 / 147.5 vs 150.2 M)
        int i0 = 0;
        float ld[3][4 * S] = {};
        auto p0_open = [&]() __attribute__((always_inline)) {
            int gi[S];
            {   // poll the flag and fetch the four index words in ONE LDS round trip (a wave's LDS operations complete in order: indices read behind a flag that has the
                // new sequence number are the new ones): 156.9 -> 158.0 M.  The polls of this kernel do not sleep between reads (156.1 vs 155.3 M with s_sleep 1).  (The same merge for the chains' counter + the tree's state reads, and for P0's counter +
                // slot 0's cell: 157.5 / 157.1 vs 157.7 M, not kept.)
                typedef int i4 __attribute__((ext_vector_type(4)));
                i4 v;
                int f;
                const uint32_t idx_a = lds_addr(idx_p);
                do {
                    asm volatile("ds_read_b32 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(f), "=&v"(v) : "v"(flag_p), "v"(idx_a) : "memory");
                    f = __builtin_amdgcn_readfirstlane(f);
                } while (f != seqP);
#pragma unroll
                for (int s = 0; s < S; ++s) gi[s] = __builtin_amdgcn_readfirstlane(v[s]);
            }
#pragma unroll
            for (int s = 0; s < S; ++s) {
                o_sig[s] = (uint32_t)(gi[s] & 0xFF) * (uint32_t)(RA * 4);
                o_pred[s] = (uint32_t)((gi[s] >> 8) & 0xFF) * (uint32_t)(RA * 4);
                o_exc[s] = ((uint32_t)(gi[s] >> 16) & 0xFFu) * (uint32_t)(RA * 4);
                o_cond[s] = (uint32_t)(((size_t)stream_of(S * p + s) * nf + (size_t)fP) * RA * 4);     // (< 4 GB: the engine bounds the chunk)
            }
            int t_ = tid0;
            LPCN_REMAT_V(t_);
            i0 = (((P0PERM >> (4 * ((t_ >> 6) & 3))) & 3) << 6) | (t_ & 63);   // lane i0 of the 256 takes rows i0 + 256 k; the half round (k = 4: rows 1024..1151) goes to the two waves with i0 < 128
        };
        const bool fifth = ((P0PERM >> (4 * (wave & 3))) & 3) < 2;      // (wave-uniform)
        auto issue = [&](const int k, const int buf) __attribute__((always_inline)) {
            const uint32_t rb = (uint32_t)(i0 + 256 * k) * 4u;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                ld[buf][4 * s + 0] = *(const LPCN_GLOBAL float *)((const LPCN_GLOBAL char *)cond_a_s + (o_cond[s] + rb));
                ld[buf][4 * s + 1] = *(const LPCN_GLOBAL float *)((const LPCN_GLOBAL char *)emb_nat_sig + (o_sig[s] + rb));
                ld[buf][4 * s + 2] = *(const LPCN_GLOBAL float *)((const LPCN_GLOBAL char *)emb_nat_pred + (o_pred[s] + rb));
                ld[buf][4 * s + 3] = *(const LPCN_GLOBAL float *)((const LPCN_GLOBAL char *)emb_nat_exc + (o_exc[s] + rb));
            }
        };
        auto reduce = [&](const int k, const int buf) __attribute__((always_inline)) {
            const int r = i0 + 256 * k;
            float g[S];
#pragma unroll
            for (int s = 0; s < S; ++s) g[s] = ((ld[buf][4 * s + 0] + ld[buf][4 * s + 1]) + ld[buf][4 * s + 2]) + ld[buf][4 * s + 3];      // src/nnet.c:487-489
            if (k < 3) {                                     // update / reset rows: start value = (bias + diag*h) + input (src/nnet.c:431-440)
                const int n = r >= NA ? r - NA : r;
                const float2 bd = *(const float2 *)(sm_abias + 2 * r);
                const float4 hv = *(const float4 *)(hT_p + n * S);
                *(float4 *)(sm_pre_ur + r * S) = make_float4((bd.x + bd.y * hv.x) + g[0], (bd.x + bd.y * hv.y) + g[1], (bd.x + bd.y * hv.z) + g[2], (bd.x + bd.y * hv.w) + g[3]);
            } else {                                         // candidate rows: the input part goes to the gate stage
                *(float4 *)(sm_inh + (r - 2 * NA) * S) = make_float4(g[0], g[1], g[2], g[3]);
            }
        };
        auto p0_arrive = [&]() __attribute__((always_inline)) {     // the stores above are ahead of this add in the wave's LDS queue
            int one = 1;
            unsigned long long ex;
            asm volatile("s_mov_b64 %0, exec\n\t"
                         "s_mov_b64 exec, 1\n\t"
                         "ds_add_u32 %1, %2\n\t"
                         "s_mov_b64 exec, %0"
                         : "=&s"(ex) : "v"(p0cnt_p), "v"(one) : "memory");
        };
        // The two kinds of waves take disjoint paths (so that the 64 registers GRU-B's assembly block names and the 48 loads of the start-value pass in
        // flight never count against each other in the register allocation):
        if (wave < CW) {
            if (q_chain) {
                // GRU-B of stream `wave` of group Q: one lane per output row, 384 dependent adds per row in the reference's order (src/nnet.c:326-372); the
                // state operand is a broadcast LDS read of the block the gate stage has written (grub_lds_loop_s4.inc, tools/gen_grub_asm.py --lds 4).
                // Round 6 measured three other forms of this link on the two-group kernel, all bit-exact, all slower (EXPERIMENTS.md): two streams per
                // wave with 3 reads per block (92 clk per block), the same packed over the stream pair (90), all four streams on one wave with the
                // products on the matrix pipe (135) -- per stream-block cheaper, but the interval waits for its longest chain.
                __builtin_amdgcn_s_setprio(3);
                const int s = wave;
                int ln_ = tid0;
                LPCN_REMAT_V(ln_);                           // (lane-derived addresses are rebuilt here: hoisted out of the loop they are spilled, and their scratch reloads sit in front of the chain)
                ln_ &= 63;
                const int r = ln_ < RB ? ln_ : RB - 1;
                const int g6 = r >> 3, ri = r & 7;
                const float *const condb_q = (const float *)(gq + L::g_condb);
                float zrh = sm_bbias[r] + condb_q[s * RB + r];                  // src/nnet.c:351
                float rec = sm_bbias[RB + r];
#pragma unroll
                for (int j = 0; j < NB; ++j) rec = rec + sm_brec[j * RB + r] * hB_q[s * NB + j];
                uint32_t wp32 = lds_addr(smem + L::bw + (sm_bstart[g6] * 8 + ri) * 16 + ((0x321100 >> (4 * g6)) & 15) * 128);
                uint32_t hp32 = lds_addr(gq + L::g_hA + s * 16);
                asm volatile(
#include "grub_lds_loop_s4.inc"
                    : [z] "+v"(zrh), [wp] "+v"(wp32), [hp] "+v"(hp32) : : LPCN_GRUB_LDS_CLOBBERS);
                __builtin_amdgcn_s_setprio(0);
                LPCN_X2_PROF(1);
                // gates: rows [0,16) update, [16,32) reset, [32,48) candidate (src/nnet.c:362-371)
                const int ln = ln_ & 15;
                const float sg = lpcn_sigmoid(zrh + rec, sm_tansig);
                const float r_gate = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((16 + ln) << 2, __builtin_bit_cast(int, sg)));
                const float hc = lpcn_tanh(zrh + rec * r_gate, sm_tansig);
                const float hc_i = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((32 + ln) << 2, __builtin_bit_cast(int, hc)));
                if (ln_ < NB) {
                    const float hold = hB_q[s * NB + ln_];
                    const float hnew = sg * hold + (1.f - sg) * hc_i;
                    if ((live_maskQ >> s) & 1) hB_q[s * NB + ln_] = hnew;
                }
                {                                            // (the store above is ahead of this add in the wave's LDS queue)
                    int one = 1;
                    unsigned long long ex;
                    asm volatile("s_mov_b64 %0, exec\n\t"
                                 "s_mov_b64 exec, 1\n\t"
                                 "ds_add_u32 %1, %2\n\t"
                                 "s_mov_b64 exec, %0"
                                 : "=&s"(ex) : "v"(chcnt_q), "v"(one) : "memory");
                }
                LPCN_X2_PROF(2);
            }
            // (Round 6 also gave these waves a share of P's start-value pass -- a round of update / reset rows and one of candidate inputs, or the candidate
            // inputs alone, issued above the gates: 144.0 / 145.7 vs 147.0 M.  With only the barrier waits instrumented the chain waves have 1.3-2.2 k clk
            // of slack per half-step, not the 4 k the full phase table shows; EXPERIMENTS.md.)
        } else {
            if (do_heads) {
                hA_cur = gq + L::g_hA;
                load_negz();
                {
                    int r = LPCN_ROW(0);
                    LPCN_REMAT_V(r);
                    r = r < 0 ? 0 : r;
                    const int n = r - 2 * NA;
                    const float bias = sm_abias[2 * r], diag = sm_abias[2 * r + 1];
                    const float *hT_q = (const float *)(gq + L::g_hT);
#pragma unroll
                    for (int s = 0; s < S; ++s) acc[s] = bias + diag * hT_q[n * S + s];
                }
#pragma unroll
                for (int j = 0; j < PF; ++j) if (J0 + j < NW) fetch_h(J0 + j);
                head_step(head_step, std::integral_constant<int, J0>{}, std::integral_constant<int, JM>{});
            }
            LPCN_X2_PROF(3);
            // all three rounds are issued at once (48 loads per lane in flight), the update / reset rows are reduced and announced first -- the row owners
            // wait for those --, the candidate inputs, which only the gate stage behind the barrier needs, after the rest of the heads
            if (p0_wave) {
                p0_open();
                LPCN_X2_PROF(1);                             // (head waves: slot 1 = wait for the indices, slot 2 = issue of the rounds)
                issue(0, 0); issue(1, 1); issue(2, 2);
                LPCN_X2_PROF(2);
                reduce(0, 0); reduce(1, 1); reduce(2, 2);
                p0_arrive();
                issue(3, 0);
                if (fifth) issue(4, 1);
            }
            LPCN_X2_PROF(4);
            if (do_heads) {
                head_step(head_step, std::integral_constant<int, JM>{}, std::integral_constant<int, NW>{});
                int r = LPCN_ROW(0);
                LPCN_REMAT_V(r);
                if (r >= 0) {
                    float *c = pre_cell(r, gq);
#pragma unroll
                    for (int s = 0; s < S; ++s) c[s] = acc[s];
                }
            }
            if (p0_wave) { reduce(3, 0); if (fifth) reduce(4, 1); }
            LPCN_X2_PROF(3);
        }

        // ---------------------------------------------------------------- 5: P1 of group P ----
        if (p_active) {
            hA_cur = gp + L::g_hA;
            load_negz();
            {                                                // the start values of the update / reset rows and the candidate inputs come from P0
                int v;
                const int want = seqP * NP0;
                do {
                    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(p0cnt_p) : "memory");
                    v = __builtin_amdgcn_readfirstlane(v);
                } while (v != want);
            }
            LPCN_X2_PROF(11);                                // wait for the start-value pass of the four row waves
            // slot 0 becomes the running row: candidate rows start from bias + diag*h -- or from the sums their head has parked --, update / reset
            // rows from their P0 cell; candidate rows further down park bias + diag*h in their own cell
            if (plain_start) {
                int r = LPCN_ROW(0);
                LPCN_REMAT_V(r);
                r = r < 0 ? 0 : r;
                const float *c = pre_cell(r, gp);
#pragma unroll
                for (int s = 0; s < S; ++s) acc[s] = c[s];
            } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                int r = LPCN_ROW(k);
                LPCN_REMAT_V(r);
                const bool live_row = r >= 0;
                r = r < 0 ? 0 : r;
                const bool candidate = r >= 2 * NA;
                const int n = candidate ? r - 2 * NA : 0;
                const float bias = sm_abias[2 * r], diag = sm_abias[2 * r + 1];
                float *c = pre_cell(r, gp);
                const bool parked = k == 0 && early_wave;
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const float bv = bias + diag * hT_p[n * S + s];
                    if (k == 0) acc[s] = (candidate && !parked) ? bv : c[s];
                    else if (candidate && live_row) c[s] = bv;
                }
            }
            }
            auto row_swap = [&](const int k_done, const int k_next) __attribute__((always_inline)) {   // finished row out, next row in
                int r = LPCN_ROW(k_done), r2 = LPCN_ROW(k_next);
                LPCN_REMAT_V(r);
                LPCN_REMAT_V(r2);
                if (r >= 0) {
                    float *c = pre_cell(r, gp);
#pragma unroll
                    for (int s = 0; s < S; ++s) c[s] = acc[s];
                }
                r2 = r2 < 0 ? 0 : r2;
                const float *c2 = pre_cell(r2, gp);
#pragma unroll
                for (int s = 0; s < S; ++s) acc[s] = c2[s];
            };
            auto row_store = [&](const int k) __attribute__((always_inline)) {
                int r = LPCN_ROW(k);
                LPCN_REMAT_V(r);
                if (r >= 0) {
                    float *c = pre_cell(r, gp);
#pragma unroll
                    for (int s = 0; s < S; ++s) c[s] = acc[s];
                }
            };
            const int jend = b3;
            LPCN_X2_PROF(4);
#pragma unroll
            for (int j = 0; j < PF && j < NW; ++j) fetch_h(j);
            LPCN_REMAT_S(b1);
            LPCN_REMAT_S(b2);
            LPCN_REMAT_S(b3);
            // All tests below are wave-uniform scalar branches; an ordinary item falls through every one of them.
            int nextb = b1;
            LPCN_REMAT_S(nextb);
            auto item = [&](const int j) __attribute__((always_inline)) -> bool {           // false: this wave has no more items
                if (__builtin_expect(j >= jend, 0)) return false;
                if (j + PF < NW) fetch_h(j + PF);
                if (__builtin_expect(j == nextb, 0)) {       // slot boundaries (a slot may be empty): ONE compare per item against the next one
                    if (j == b1) row_swap(0, 1);
                    if (j == b2) row_swap(1, 2);
                    __builtin_amdgcn_s_waitcnt(0xC07F);
                    nextb = b1 > j ? b1 : (b2 > j ? b2 : NW);
                }
                mac(j);
                return true;
            };
            auto run_items = [&](auto self, auto jc) __attribute__((always_inline)) -> void {
                constexpr int j = decltype(jc)::value;
                if constexpr (j < NW) {
                    if (!item(j)) return;
                    self(self, std::integral_constant<int, j + 1>{});
                }
            };
            run_items(run_items, std::integral_constant<int, 0>{});
            if (q_chain) load_fc();
            LPCN_X2_PROF(5);
            // close whichever slot is still open; slots that start exactly at the end have no items
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
            LPCN_X2_PROF(6);
        }
        // ------------------------------------------------------------ P4 of group Q: dual-FC tree, all nodes at once (src/nnet.c:163-214) --
        // Round 6: the tree runs on each wave BEHIND its own part of interval A, in front of the barrier -- a wave that is done early evaluates its nodes
        // while others are still in their (latency-bound) items, instead of all eight saturating the vector units at once behind the barrier.
        if (q_chain) {
            if (!p_active) load_fc();
            {
                int v;
                const int want = chnQ * S;
                do {
                    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(chcnt_q) : "memory");
                    v = __builtin_amdgcn_readfirstlane(v);
                } while (v != want);
            }
            int tid = tid0;
            LPCN_REMAT_V(tid);
            const int node = tid >> 1;
            const int node_level = node > 0 ? 31 - __clz(node) : 0;
            const float *const thr_q = (const float *)(gq + L::g_thr);
            unsigned long long *const mask_q = (unsigned long long *)(gq + L::g_mask);
            // the node's 16 products for all four streams from the matrix pipe, four columns at a time, like a GRU-A item: lane k of a quad holds stream k's
            // state, v_mfma_f32_4x4x1 with C = -0.0 returns (stream k's value) x (this lane's weight) in register k, rounded once; the sums stay in the
            // reference's order as packed adds over stream pairs (src/nnet.c:194-199) -- 16 MFMA + 32 packed adds instead of 64 multiplies + 64 adds
            // (146.7 -> 148.3 M samples/s)
            float sums[S];
            {
                typedef float f4 __attribute__((ext_vector_type(4)));
                typedef float f2 __attribute__((ext_vector_type(2)));
                float hs[NB];
                const float4 *hp = (const float4 *)(hB_q + (tid & 3) * NB);
#pragma unroll
                for (int qd = 0; qd < NB / 4; ++qd) { const float4 v4 = hp[qd]; hs[4 * qd] = v4.x; hs[4 * qd + 1] = v4.y; hs[4 * qd + 2] = v4.z; hs[4 * qd + 3] = v4.w; }
                load_negz();
                f2 s01 = {fcb, fcb}, s23 = {fcb, fcb};
#pragma unroll
                for (int jb = 0; jb < NB; jb += 4) {
                    f4 pv[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) pv[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(hs[jb + c], fcw[jb + c], negz, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        s01 = s01 + __builtin_shufflevector(pv[c], pv[c], 0, 1);
                        s23 = s23 + __builtin_shufflevector(pv[c], pv[c], 2, 3);
                    }
                }
                sums[0] = s01[0]; sums[1] = s01[1]; sums[2] = s23[0]; sums[3] = s23[1];
            }
            // the four streams stage by stage -- four table lookups in flight, the masks stored at the end (a store per stream makes every stream a basic block
            // of its own, and the compiler then runs them one behind the other)
            float vq[S], thq[S];
            unsigned long long mq[S];
#pragma unroll
            for (int s = 0; s < S; ++s) { vq[s] = lpcn_tanh(sums[s], sm_tansig); thq[s] = thr_q[s * 8 + node_level]; }
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const float v = fcf * vq[s];
                const float vo = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
                const float lg = v + vo;
                mq[s] = __ballot(thq[s] < lg) & (wave == 0 ? 0x5555555555555554ull : 0x5555555555555555ull);
            }
            if (lane == 0) {
#pragma unroll
                for (int s = 0; s < S; ++s) mask_q[s * 8 + wave] = mq[s];
            }
            // wave LW: the prediction terms of Q's next sample that do not involve the sample about to be drawn (src/lpcnet.c:252,262)
            if (is_lw) {
                lpc_tap = ((const float *)(gq + L::g_lpc))[tid & 63];
                prod_old = row_shr1(histQ, 0.f) * lpc_tap;
            }
        }
        LPCN_X2_PROF(9);
        __syncthreads();                                                       // B1
        LPCN_X2_PROF(7);

        // ================================================================ interval B ===========
        int tid = tid0;
        LPCN_REMAT_V(tid);
        // ------------------------------------------------------------ P2 of group P: GRU-A gates (src/nnet.c:441-447) --
        if (p_active) {
            constexpr int NI = NA * S, NQ = NI / LPCN_WG_THREADS;
            const float *const prec_p = (const float *)(gp + L::g_prec);
            float z[NQ], rg[NQ], a[NQ], hold[NQ];
#pragma unroll
            for (int k = 0; k < NQ; ++k) {
                const int i = tid + k * LPCN_WG_THREADS;
                z[k] = sm_pre_ur[i];
                rg[k] = sm_pre_ur[NI + i];
                a[k] = prec_p[i];
                hold[k] = hT_p[i];
            }
#pragma unroll
            for (int k = 0; k < NQ; ++k) { z[k] = lpcn_sigmoid(z[k], sm_tansig); rg[k] = lpcn_sigmoid(rg[k], sm_tansig); }
#pragma unroll
            for (int k = 0; k < NQ; ++k) a[k] = a[k] * rg[k] + sm_inh[tid + k * LPCN_WG_THREADS];
#pragma unroll
            for (int k = 0; k < NQ; ++k) a[k] = lpcn_tanh(a[k], sm_tansig);
            // item i = tid + 512 k is (neuron (tid >> 2) + 128 k, stream tid & 3): the stream is the lane's own for all three, and the block-ordered copy's address
            // advances by a constant (32 blocks of 64 B + 8 pads of 16 B per 128 neurons)
            const unsigned ut = (unsigned)tid;
            const bool live_s = ((live_maskP >> (ut & 3u)) & 1) != 0;
            unsigned char *const ha0 = gp + L::g_hA + L::ha_off((int)(ut >> 4)) + (ut & 3u) * 16u + ((ut >> 2) & 3u) * 4u;
            static_assert(L::ha_off(32) == 32 * L::HA_STRIDE + 8 * 16 && LPCN_WG_THREADS == 512 && S == 4, "gate-stage address stride");
#pragma unroll
            for (int k = 0; k < NQ; ++k) {
                const float hnew = z[k] * hold[k] + (1.f - z[k]) * a[k];      // src/nnet.c:447
                const float hv = live_s ? hnew : hold[k];
                hT_p[tid + k * LPCN_WG_THREADS] = hv;
                *(float *)(ha0 + k * L::ha_off(32)) = hv;
            }
        }
        LPCN_X2_PROF(8);
        __syncthreads();                                                       // B2
        LPCN_X2_PROF(10);

        // ---- the groups swap roles; P's position advances by the sample it has just started
        if (p_active) { if (++smpP == frame_len) { smpP = 0; ++fP; } }
        { const float t = histP; histP = histQ; histQ = t; }
        { const bool t = liveP; liveP = liveQ; liveQ = t; }
        { const int t = live_maskP; live_maskP = live_maskQ; live_maskQ = t; }
        { const int t = seqP; seqP = seqQ; seqQ = t; }
        { const int t = chnP; chnP = chnQ; chnQ = t; }
        { const int t = smpP; smpP = smpQ; smpQ = t; }
        { const int t = fP; fP = fQ; fQ = t; }
        asm volatile("; LPCN_SAMPLE_LOOP_END" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }

#if LPCN_ENABLE_PROF
    if (profiling && (tid0 & 63) == 0) {
#pragma unroll
        for (int i = 0; i < 12; ++i) prof[(tid0 >> 6) * 12 + i] += pt[i];
    }
#endif
    // ---- flush the last frame's PCM of both groups, write the state back.  (After the loop the P variables belong to group 0:
    // 2 T + 3 half-steps = an odd number of swaps from P = group 1.)
    __syncthreads();
    {
        auto *out = as_global_rw(Ap->pcm);
        const size_t pstride = (size_t)Ap->pcm_stride;
        for (int i = tid0; i < 2 * S * LPCN_FRAME_SIZE; i += LPCN_WG_THREADS) {
            const int gs = i / LPCN_FRAME_SIZE, k = i % LPCN_FRAME_SIZE;
            const short *pb = (const short *)(smem + (gs >> 2) * L::G_SZ + L::g_pcm);
            if (gs < n_valid && k < frame_len) out[(size_t)(s0 + gs) * pstride + (size_t)(n_frames - 1) * LPCN_FRAME_SIZE + k] = pb[(gs & 3) * LPCN_FRAME_SIZE + k];
        }
    }
    {
        const int tid = tid0;
        for (int i = tid; i < 2 * S * NA; i += LPCN_WG_THREADS) {
            const int gs = i / NA, n = i % NA;
            if (gs < n_valid) states[s0 + gs].gru_a[n] = ((const float *)(smem + (gs >> 2) * L::G_SZ + L::g_hT))[n * S + (gs & 3)];
        }
        for (int i = tid; i < 2 * S * NB; i += LPCN_WG_THREADS) {
            const int gs = i / NB;
            if (gs < n_valid) states[s0 + gs].gru_b[i % NB] = ((const float *)(smem + (gs >> 2) * L::G_SZ + L::g_hB))[(gs & 3) * NB + i % NB];
        }
        if (is_lw) {
            if (LPCN_LROW < n_valid) states[s0 + LPCN_LROW].last_sig[LPCN_TAP] = histP;
            if (S + LPCN_LROW < n_valid) states[s0 + S + LPCN_LROW].last_sig[LPCN_TAP] = histQ;
        }
        if (tid < n_valid) {
            auto *st = &states[s0 + tid];
            const unsigned char *gb = smem + (tid >> 2) * L::G_SZ;
            const int *li = (const int *)(gb + L::g_lead) + (tid & 3) * 8;
            const float *lp = (const float *)(gb + L::g_lpc);
#pragma unroll
            for (int j = 0; j < LPCN_LPC_ORDER; ++j) st->lpc[j] = lp[(tid & 3) * LPCN_LPC_ORDER + j];
            st->deemph_mem = ((const float *)li)[1];
            st->last_exc = li[2];
#pragma unroll
            for (int j = 0; j < 4; ++j) st->rng[j] = (uint32_t)li[4 + j];
        }
    }
#undef LPCN_ROW
#undef LPCN_LROW
#undef LPCN_TAP
}

}  // namespace lpcn
