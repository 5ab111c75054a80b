// Device runtime of the LPCNet HIP engine: model upload, per-batch device buffers, kernel launches.
// Only the C ABI of lpcnet_engine.h is visible to the C host shell.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "lpcnet_engine.h"
#include "lpcnet_tables_gen.h"
#include "sample_kernel.hip.h"
#include "frame_kernels.hip.h"
#include "decode_kernel.hip.h"
#include <math.h>

static thread_local char g_err[512] = "";
extern "C" const char *lpcn_last_error(void) { return g_err; }

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            snprintf(g_err, sizeof(g_err), "%s:%d: %s -> %s", __FILE__, __LINE__, #expr,      \
                     hipGetErrorString(_e));                                                  \
            return LPCN_E_HIP;                                                                \
        }                                                                                     \
    } while (0)

// Every entry point selects the engine's device and restores the caller's current device on return.
struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; if (prev != dev) (void)hipSetDevice(dev); else prev = -1; }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

struct lpcn_engine {
    int device = 0;
    int nw = 0, nw_variant = 0, nb_b = 0;
    bool is_int8 = false;
    bool fc_f16 = false;               // FAST sub-option: fp16 dual FC (lpcn_engine_set_fast(e, 2))
    bool fast = false;                 // FAST arithmetic (lpcn_engine_set_fast): fused / integer accumulation instead of the reference's generic-C order
    float lpc_gamma = 1.f;
    hipStream_t stream = nullptr;
    std::vector<void *> allocs;
    LpcnSampleArgs sargs{};        // model part filled at creation
    LpcnSampleArgs sargs_fast{};   // the same with the FAST arithmetic's own GRU-A packing (int8 blobs: dealt without candidate heads)
    bool has_fast_image = false;
    int nw_variant_fast = 0;
    LpcnSampleArgs sargs_x2{};     // the same with the two-group kernel's own GRU-A packing (float blobs; model_pack.c: lpcn_model_pack_x2)
    bool has_x2_image = false;
    int nw_variant_x2 = 0;
    LpcnFrameModel fmodel{};
    lpcn::DecodeTables dec{};      // codec path: VQ codebooks + pitch table (set by lpcn_engine_set_codebooks)
    bool has_codebooks = false;
};

struct lpcn_batch_dev {
    lpcn_engine *e = nullptr;
    int n = 0, max_chunk = 0, S = 0, frame_len = LPCN_FRAME_SIZE;
    bool S_auto = true;                // streams per workgroup are chosen by the engine (measured on this batch, see autotune_streams_per_wg)
    bool tuned = false;                // ... and have been measured for the current arithmetic flavour
    bool pack2 = false;                // 128-VGPR variant: two workgroups per CU (int8, <= 32 items per lane, more workgroups than CUs)
    lpcn_stream_state *d_state = nullptr;
    int *d_fc_base = nullptr;
    float *d_cond_a = nullptr, *d_cond_b = nullptr, *d_lpc = nullptr, *d_cond = nullptr;
    float *d_feat = nullptr;           // staging for host-pointer runs / decoded feature vectors
    float *d_vq_mem = nullptr;         // [n][18] VQ memory of the codec path (src/lpcnet_private.h:52)
    lpcn_stream_state *d_state_tmp = nullptr;   // per-stream-arguments step (lpcn_batch_dev_step_host): the compacted group's states
    int *d_map = nullptr;              //   ... and its stream indices
    float *d_keep_a = nullptr, *d_keep_b = nullptr, *d_keep_lpc = nullptr;   //   ... and every stream's most recent frame products
    std::vector<char> keep_ok;         //   ... which exist only for streams whose last frame step went through the step call (mode 1)
    unsigned char *d_packets = nullptr;
    size_t packets_cap = 0;
    short *d_pcm = nullptr;
    size_t feat_cap = 0, pcm_cap = 0;
    LpcnSampleArgs *d_args = nullptr;
    float *d_dbg = nullptr;
    unsigned long long *d_prof = nullptr;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    // Ordering across caller streams: the batch's scratch buffers and state are shared by every call, so each enqueue
    // records ev_last on its stream; a call on a DIFFERENT stream first waits for it, and every host-side access
    // (sync, state get/set, reset, destroy, buffer growth) waits for it as well.
    hipEvent_t ev_last = nullptr;
    hipStream_t last_stream = nullptr;
    bool pending = false;
    void *h_pin = nullptr;             // pinned host staging of the single-stream fast path (state | features | pcm)
    bool timing = false;
    float ms_sample = 0.f, ms_frame = 0.f;
};

// HIP-graph capture: a captured copy node re-reads its HOST source at every replay, so the argument block of a captured launch cannot come from the
// caller's stack -- it travels as the by-value parameter of a one-lane kernel instead, which the graph's kernel node owns (round 6, ADVICE r5: round 5's
// pinned pool of 32 slots leaked a slot per captured launch and ended every capture after the 32nd)
__global__ void lpcn_set_args_kernel(LpcnSampleArgs *dst, const LpcnSampleArgs a) { *dst = a; }
static bool stream_is_capturing(hipStream_t st)
{
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    return st != nullptr && hipStreamIsCapturing(st, &cs) == hipSuccess && cs == hipStreamCaptureStatusActive;
}
// (while the caller's stream is being captured into a HIP graph nothing is executed: the batch's event chain is left alone -- an event
// recorded inside a capture cannot be waited for from the host -- and ordering replays against other work on the batch is the caller's job)
static int order_begin(lpcn_batch_dev *b, hipStream_t st)
{
    if (stream_is_capturing(st)) return 0;
    if (b->pending && st != b->last_stream) HIP_TRY(hipStreamWaitEvent(st, b->ev_last, 0));
    return 0;
}
static int order_end(lpcn_batch_dev *b, hipStream_t st)
{
    if (stream_is_capturing(st)) return 0;
    HIP_TRY(hipEventRecord(b->ev_last, st));
    b->last_stream = st;
    b->pending = true;
    return 0;
}
// host-side wait for everything enqueued for this batch, whichever stream it went to
static int wait_all(lpcn_batch_dev *b)
{
    if (b->pending) { HIP_TRY(hipEventSynchronize(b->ev_last)); b->pending = false; }
    HIP_TRY(hipStreamSynchronize(b->e->stream));
    return 0;
}

template <typename T>
static int upload(lpcn_engine *e, const T **dst, const void *src, size_t count)
{
    void *p = nullptr;
    HIP_TRY(hipMalloc(&p, count * sizeof(T) ? count * sizeof(T) : 16));
    e->allocs.push_back(p);
    if (count) HIP_TRY(hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice));
    *dst = (const T *)p;
    return 0;
}

extern "C" int lpcn_engine_create(lpcn_engine **out, int device, const lpcn_model_host *m)
{
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        snprintf(g_err, sizeof(g_err), "no HIP device visible: the LPCNet HIP engine has no CPU fallback");
        return LPCN_E_NODEVICE;
    }
    if (device < 0 || device >= ndev) { snprintf(g_err, sizeof(g_err), "bad device %d (have %d)", device, ndev); return LPCN_E_ARG; }
    // compiled item counts per lane: fp32 items are 4 VGPRs each, int8 items 1 VGPR
    static const int variants_f32[] = {24, 28, 30, 32, 36, 40, 48, 64, 80, 96, 0};
    static const int variants_i8[] = {32, 48, 64, 96, 0};
    int nwv = 0;
    for (const int *v = m->is_int8 ? variants_i8 : variants_f32; *v; ++v) if (m->nw <= *v) { nwv = *v; break; }
    if (!nwv) {
        snprintf(g_err, sizeof(g_err), "GRU-A does not fit the kernel's item variants (needs %d items on one lane, max 96: more than three full rows' worth of blocks on one wave)", m->nw);
        return LPCN_E_MODEL;
    }
    DeviceGuard guard(device);
    lpcn_engine *e = new lpcn_engine();
    e->device = device;
    e->nw = m->nw; e->nw_variant = nwv; e->nb_b = m->nb_b_padded; e->lpc_gamma = m->lpc_gamma;
    e->is_int8 = m->is_int8 != 0;
    int rc = 0;
    auto fail = [&](int code) { lpcn_engine_destroy(e); return code; };
    if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) return fail(LPCN_E_HIP);

    LpcnSampleArgs &a = e->sargs;
    // GRU-A as dealt to waves and lanes (model_pack.c) + the embedding tables in that lane order: everything that depends on the dealing
    auto upload_gru_a = [&](const lpcn_model_host *mm, LpcnSampleArgs &dst, int nwv_) -> int {
        // re-pad the item arrays from the model's nw to the compiled variant
        const size_t item_dw = mm->is_int8 ? 1 : 4;            // dwords per (lane, item)
        const uint32_t *src = mm->is_int8 ? (const uint32_t *)mm->pk_a_wq : (const uint32_t *)mm->pk_a_w;
        std::vector<uint32_t> w((size_t)LPCN_WAVES * nwv_ * 64 * item_dw, 0u);
        std::vector<uint8_t> b((size_t)LPCN_WAVES * nwv_ * 64, 0);
        for (int wv = 0; wv < LPCN_WAVES; ++wv)
            for (int j = 0; j < mm->nw; ++j) {
                // the early head of slot 0, [nw - head, nw) in the model, stays end-aligned in the compiled variant's item array
                const int jd = j >= mm->nw - mm->pk_a_head[wv] ? j + (nwv_ - mm->nw) : j;
                memcpy(&w[((size_t)wv * nwv_ + jd) * 64 * item_dw], &src[((size_t)wv * mm->nw + j) * 64 * item_dw], 64 * item_dw * 4);
                memcpy(&b[((size_t)wv * nwv_ + jd) * 64], &mm->pk_a_blk[((size_t)wv * mm->nw + j) * 64], 64);
            }
        const uint32_t *d = nullptr;
        int rc_ = 0;
        if ((rc_ = upload<uint32_t>(e, &d, w.data(), w.size()))) return rc_;
        dst.a_w = (const float4 *)d;
        if ((rc_ = upload<uint8_t>(e, &dst.a_blk, b.data(), b.size()))) return rc_;
        int bound[LPCN_WAVES * 4], allh[LPCN_WAVES * 3];
        for (int wv = 0; wv < LPCN_WAVES; ++wv) {
            for (int k = 0; k < 4; ++k) bound[wv * 4 + k] = mm->pk_a_bound[wv][k];
            for (int k = 0; k < 3; ++k) allh[wv * 3 + k] = mm->pk_a_allh[wv][k];
        }
#define UPD(T, field, srcp, count) if ((rc_ = upload<T>(e, &dst.field, srcp, count))) return rc_
        UPD(int, a_row, mm->pk_a_row, LPCN_WAVES * 3 * 64);
        UPD(int, a_bound, bound, LPCN_WAVES * 4);
        UPD(int, a_allh, allh, LPCN_WAVES * 3);
        UPD(int, a_head, mm->pk_a_head, LPCN_WAVES);
        if (mm->pk_emb[0]) {                                  // (the two-group kernel's image has no lane-ordered tables)
            UPD(float, emb_sig, mm->pk_emb[0], (size_t)256 * LPCN_MAX_SLOTS * LPCN_WG_THREADS);
            UPD(float, emb_pred, mm->pk_emb[1], (size_t)256 * LPCN_MAX_SLOTS * LPCN_WG_THREADS);
            UPD(float, emb_exc, mm->pk_emb[2], (size_t)256 * LPCN_MAX_SLOTS * LPCN_WG_THREADS);
        }
#undef UPD
        return 0;
    };
    if ((rc = upload_gru_a(m, a, nwv))) return fail(rc);
#define UP(T, field, src, count) if ((rc = upload<T>(e, &a.field, src, count))) return fail(rc)
    if (!m->is_int8) {                                       // the tables in the blob's own [256][1152] order: the start-value pass of the two-group kernel
        UP(float, emb_nat_sig, m->emb_sig, (size_t)256 * LPCN_ROWS_A);
        UP(float, emb_nat_pred, m->emb_pred, (size_t)256 * LPCN_ROWS_A);
        UP(float, emb_nat_exc, m->emb_exc, (size_t)256 * LPCN_ROWS_A);
    }
    UP(float, a_bias1, m->a_bias + LPCN_ROWS_A, LPCN_ROWS_A);
    UP(float, a_diag, m->a_diag, LPCN_ROWS_A);
    if (m->is_int8) {
        const uint32_t *d = nullptr;
        if ((rc = upload<uint32_t>(e, &d, m->pk_b_wq, (size_t)8 * m->nb_b_padded))) return fail(rc);
        a.b_w = (const float *)d;
        // recurrent matrix: blob layout [6 groups][4 column blocks][8 rows][4] -> [48 rows][4 blocks] dwords
        uint32_t rq[LPCN_ROWS_B * 4];
        const unsigned char *src = (const unsigned char *)m->b_rec;
        for (int r = 0; r < LPCN_ROWS_B; ++r)
            for (int jb = 0; jb < 4; ++jb) memcpy(&rq[r * 4 + jb], src + (((r >> 3) * 4 + jb) * 8 + (r & 7)) * 4, 4);
        if ((rc = upload<uint32_t>(e, &d, rq, LPCN_ROWS_B * 4))) return fail(rc);
        a.b_rec = (const float *)d;
    } else {
        UP(float, b_w, m->pk_b_w, (size_t)32 * m->nb_b_padded);
        UP(float, b_rec, m->b_rec, LPCN_N_B * LPCN_ROWS_B);
    }
    UP(int, b_start, m->pk_b_start, 7);
    UP(uint8_t, b_blk, m->pk_b_blk, m->nb_b_padded + 4);
    UP(float, b_bias, m->b_bias, 2 * LPCN_ROWS_B);
    UP(float, fc_w, m->fc_w, 256 * 2 * LPCN_N_B);
    UP(float, fc_b, m->fc_b, 512);
    UP(float, fc_f, m->fc_f, 512);
    {   // fp16 image of the dual-FC weights, packed in pairs (FAST sub-option)
        std::vector<uint32_t> wh(256 * 2 * LPCN_N_B / 2);
        for (size_t i = 0; i < wh.size(); ++i) {
            const _Float16 lo = (_Float16)m->fc_w[2 * i], hi = (_Float16)m->fc_w[2 * i + 1];
            uint16_t l, h;
            memcpy(&l, &lo, 2); memcpy(&h, &hi, 2);
            wh[i] = (uint32_t)l | ((uint32_t)h << 16);
        }
        UP(uint32_t, fc_wh, wh.data(), wh.size());
    }
    UP(float, tab_tansig, lpcn_tansig, 201);
    UP(float, tab_ulaw2lin, lpcn_ulaw2lin_tab, 256);
    UP(float, tab_logit, lpcn_logit_tab, 256);
#undef UP
    a.nb_b = m->nb_b_padded;
    a.b_dense = m->b_dense;
    if (!m->is_int8 && m->b_dense && !getenv("LPCNET_HIP_NO_X2_IMAGE")) {
        // the two-group kernel (eight float streams per workgroup) runs on its own dealing of GRU-A: chains on waves 0, 1, rows on waves 2..7
        lpcn_model_host mx;
        if (lpcn_model_pack_x2(m, &mx) == 0) {
            static const int variants_x2[] = {24, 28, 30, 32, 0};
            int nwx = 0;
            for (const int *v = variants_x2; *v; ++v) if (mx.nw <= *v) { nwx = *v; break; }
            mx.pk_b_w = m->pk_b_w; mx.pk_b_start = m->pk_b_start; mx.pk_b_blk = m->pk_b_blk;      // (the check reads GRU-B's packing too: shared)
            const int stx = lpcn_model_selftest(&mx);
            mx.pk_b_w = nullptr; mx.pk_b_start = nullptr; mx.pk_b_blk = nullptr;
            if (nwx && stx == 0) {
                e->sargs_x2 = e->sargs;
                rc = upload_gru_a(&mx, e->sargs_x2, nwx);
                e->nw_variant_x2 = nwx;
                e->has_x2_image = rc == 0;
            }
            lpcn_model_release(&mx);
            if (rc) return fail(rc);
        }
    }
    {   // the FAST arithmetic's own GRU-A image where its best dealing is not PARITY's (int8 blobs)
        lpcn_model_host mf;
        const int pf = getenv("LPCNET_HIP_NO_FAST_IMAGE") ? 1 : lpcn_model_pack_fast(m, &mf);
        if (pf < 0) { snprintf(g_err, sizeof(g_err), "packing the FAST image of GRU-A failed"); return fail(LPCN_E_MODEL); }
        if (pf == 0) {
            int nwf = 0;
            for (const int *v = variants_i8; *v; ++v) if (mf.nw <= *v) { nwf = *v; break; }
            if (nwf) {
                e->sargs_fast = e->sargs;
                rc = upload_gru_a(&mf, e->sargs_fast, nwf);
                e->nw_variant_fast = nwf;
                e->has_fast_image = rc == 0;
            }
            lpcn_model_release(&mf);
            if (rc) return fail(rc);
        }
    }

    LpcnFrameModel &fm = e->fmodel;
#define UPF(field, src, count) if ((rc = upload<float>(e, &fm.field, src, count))) return fail(rc)
    UPF(conv1_w, m->conv1_w, 3 * LPCN_FRAME_IN * LPCN_COND);
    UPF(conv1_b, m->conv1_b, LPCN_COND);
    UPF(conv2_w, m->conv2_w, 3 * LPCN_COND * LPCN_COND);
    UPF(conv2_b, m->conv2_b, LPCN_COND);
    UPF(pitch_emb, m->pitch_emb, 256 * LPCN_PITCH_EMB);
    UPF(dense1_w, m->dense1_w, LPCN_COND * LPCN_COND);
    UPF(dense1_b, m->dense1_b, LPCN_COND);
    UPF(dense2_w, m->dense2_w, LPCN_COND * LPCN_COND);
    UPF(dense2_b, m->dense2_b, LPCN_COND);
    UPF(a_dense_w, m->a_dense_w, LPCN_COND * LPCN_ROWS_A);
    UPF(a_dense_b, m->a_dense_b, LPCN_ROWS_A);
    UPF(b_dense_w, m->b_dense_w, LPCN_COND * LPCN_ROWS_B);
    UPF(b_dense_b, m->b_dense_b, LPCN_ROWS_B);
    UPF(tab_tansig, lpcn_tansig, 201);
    UPF(tab_idct, lpcn_idct_tab, 324);
    UPF(tab_tw, lpcn_fft_tw, 640);
#undef UPF
    {
        const short *src = lpcn_fft_bitrev;
        if ((rc = upload<short>(e, &fm.tab_bitrev, src, 320))) return fail(rc);
    }
    fm.lpc_gamma = m->lpc_gamma;
    fm.end2end = 0;
    *out = e;
    return 0;
}

extern "C" void lpcn_engine_destroy(lpcn_engine *e)
{
    if (!e) return;
    DeviceGuard guard(e->device);
    for (void *p : e->allocs) (void)hipFree(p);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}
extern "C" int lpcn_engine_device(const lpcn_engine *e) { return e->device; }

// VQ codebooks of the codec path (the reference's generated ceps_codebooks.c): cb1..3 [1024][17], cb_diff4 [4096][18]
extern "C" int lpcn_engine_set_codebooks(lpcn_engine *e, const float *cb1, const float *cb2, const float *cb3, const float *cb_diff4)
{
    DeviceGuard guard(e->device);
    int rc = 0;
    float pitch[64];
    for (int k = 0; k < 64; ++k) pitch[k] = (float)(pow(2.f, k / 21.) * 32);      // src/lpcnet_dec.c:107 (PITCH_MIN_PERIOD 32)
    if (e->has_codebooks) {            // a newer codebook version replaces the contents of the buffers already on the device
        HIP_TRY(hipStreamSynchronize(e->stream));
        HIP_TRY(hipMemcpy((void *)e->dec.cb1, cb1, sizeof(float) * 1024 * 17, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy((void *)e->dec.cb2, cb2, sizeof(float) * 1024 * 17, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy((void *)e->dec.cb3, cb3, sizeof(float) * 1024 * 17, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy((void *)e->dec.cb_diff4, cb_diff4, sizeof(float) * 4096 * 18, hipMemcpyHostToDevice));
        return 0;
    }
    if ((rc = upload<float>(e, &e->dec.cb1, cb1, 1024 * 17))) return rc;
    if ((rc = upload<float>(e, &e->dec.cb2, cb2, 1024 * 17))) return rc;
    if ((rc = upload<float>(e, &e->dec.cb3, cb3, 1024 * 17))) return rc;
    if ((rc = upload<float>(e, &e->dec.cb_diff4, cb_diff4, 4096 * 18))) return rc;
    if ((rc = upload<float>(e, &e->dec.pitch, pitch, 64))) return rc;
    e->has_codebooks = true;
    return 0;
}
extern "C" int lpcn_engine_has_codebooks(const lpcn_engine *e) { return e->has_codebooks ? 1 : 0; }
// LPC_GAMMA is a compile-time constant of the reference's generated nnet_data.h (lpc_weighting, src/freq.c:299-308), not
// part of the weight blob: models trained with --lpc-gamma != 1 set it here.
// END2END is a compile-time switch of the reference (src/lpcnet.c:56-80,107-108), not part of the weight blob
extern "C" int lpcn_engine_set_end2end(lpcn_engine *e, int on)
{
    e->fmodel.end2end = on != 0;
    return 0;
}
// FAST arithmetic: what the reference's own SIMD builds do (src/vec_avx.h) -- fused multiply-add for float blobs, exact
// int32 block accumulation for int8 blobs -- instead of the generic-C order.  Not bit-exact; validated teacher-forced.
extern "C" int lpcn_engine_set_fast(lpcn_engine *e, int on)
{
    e->fast = on != 0;
    e->fc_f16 = on == 2;              // 2 = FAST with the dual FC in fp16 (BASELINE config 4's wording; weights converted at engine creation)
    return 0;
}
extern "C" int lpcn_engine_set_lpc_gamma(lpcn_engine *e, float gamma)
{
    if (!(gamma > 0.f && gamma <= 1.f)) { snprintf(g_err, sizeof(g_err), "lpc_gamma must be in (0, 1]"); return LPCN_E_ARG; }
    e->lpc_gamma = gamma;
    e->fmodel.lpc_gamma = gamma;
    return 0;
}

// ------------------------------------------------------------------------------------ batches --
// Streams per workgroup: one workgroup occupies a CU, so a batch runs in ceil(workgroups / CUs) rounds; a round with S
// interleaved streams costs step[S] (measured us per sample step, tests/tools/gpu_sweep.py).  Pick the cheapest.
// (measured, us per sample step: tests/tools/gpu_sweep.py, tools/gpu_call_*.sh)
static int device_cus(const lpcn_engine *e)
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, e->device) == hipSuccess && prop.multiProcessorCount > 0) return prop.multiProcessorCount;
    return 256;
}
// items per lane of the GRU-A image the engine's current arithmetic runs on
static int cur_nw_variant(const lpcn_engine *e) { return (e->fast && e->has_fast_image) ? e->nw_variant_fast : e->nw_variant; }
static bool pack2_available(const lpcn_engine *e) { return e->is_int8 && cur_nw_variant(e) <= 32; }
// Two groups of four float streams per workgroup, half a step apart (sample_kernel_x2.hip.h): float blobs, PARITY arithmetic, dense GRU-B input
// matrix, <= 32 items per lane.  b->S == 8 selects it.
static bool x2_available(const lpcn_engine *e)
{
    const char *off = getenv("LPCNET_HIP_NO_X2");            // tools / tests: "1" = never
    if (off && *off == '1') return false;
    return e->has_x2_image && !e->fast;
}
static bool use_pack2(const lpcn_engine *e, int n, int S)
{
    const char *force = getenv("LPCNET_HIP_PACK2");          // tools / tests: "0" never, "1" whenever the variant exists
    // (S = 4 needs ~92 KB of LDS per workgroup: two do not fit a CU, and the 128-VGPR code alone is slower -- measured 119 vs 137 M)
    if (S > 2 || !pack2_available(e)) return false;
    if (force && *force) return *force == '1';
    return (n + S - 1) / S > device_cus(e);
}
static int auto_streams_per_wg(const lpcn_engine *e, int n)
{
    const int cus = device_cus(e);
    static const float step_f32[3] = {6.5f, 7.6f, 9.9f}, step_i8[3] = {4.4f, 5.5f, 7.3f};
    static const float step_f32_fast[3] = {6.0f, 7.0f, 8.9f}, step_i8_fast[3] = {4.4f, 5.0f, 7.0f};
    // PACK2 (two int8 workgroups per CU): time of a round in which every CU carries two workgroups
    static const float pair_i8[3] = {5.8f, 7.3f, 8.6f}, pair_i8_fast[3] = {5.2f, 6.3f, 7.6f};
    const float *step = e->fast ? (e->is_int8 ? step_i8_fast : step_f32_fast) : (e->is_int8 ? step_i8 : step_f32);
    const float *pair = e->fast ? pair_i8_fast : pair_i8;
    int best = 1;
    float best_t = 0.f;
    for (int k = 0; k < 3; ++k) {
        const int S = 1 << k, wgs = (n + S - 1) / S;
        float t;
        if (use_pack2(e, n, S)) t = (float)((wgs + 2 * cus - 1) / (2 * cus)) * pair[k];
        else t = (float)((wgs + cus - 1) / cus) * step[k];
        if (k == 0 || t < best_t) { best = S; best_t = t; }
    }
    if (x2_available(e)) {                                   // eight streams per workgroup: a round of the two-group kernel (us per sample step)
        static const float step_x2 = 13.9f;
        const int wgs = (n + 7) / 8;
        const float t = (float)((wgs + cus - 1) / cus) * step_x2;
        if (t < best_t) { best = 8; best_t = t; }
    }
    return best;
}

extern "C" int lpcn_batch_dev_create(lpcn_batch_dev **out, lpcn_engine *e, int n, int max_chunk)
{
    *out = nullptr;
    if (!e || n <= 0 || max_chunk <= 0) { snprintf(g_err, sizeof(g_err), "bad batch arguments"); return LPCN_E_ARG; }
    DeviceGuard guard(e->device);
    lpcn_batch_dev *b = new lpcn_batch_dev();
    b->e = e; b->n = n; b->max_chunk = max_chunk;
    b->S = auto_streams_per_wg(e, n);
    b->pack2 = use_pack2(e, n, b->S);
    auto fail = [&](int code) { lpcn_batch_dev_destroy(b); return code; };
#define AL(ptr, bytes) if (hipMalloc((void **)&ptr, (bytes)) != hipSuccess) { snprintf(g_err, sizeof(g_err), "hipMalloc(%zu) failed", (size_t)(bytes)); return fail(LPCN_E_HIP); }
    AL(b->d_state, sizeof(lpcn_stream_state) * n);
    AL(b->d_fc_base, sizeof(int) * n);
    AL(b->d_cond_a, sizeof(float) * (size_t)n * max_chunk * LPCN_ROWS_A);
    AL(b->d_cond_b, sizeof(float) * (size_t)n * max_chunk * LPCN_ROWS_B);
    AL(b->d_lpc, sizeof(float) * (size_t)n * max_chunk * LPCN_LPC_ORDER);
    AL(b->d_cond, sizeof(float) * (size_t)n * (max_chunk + 4) * LPCN_COND * 2);
    AL(b->d_args, sizeof(LpcnSampleArgs));
    AL(b->d_vq_mem, sizeof(float) * (size_t)n * LPCN_NB_BANDS);
#undef AL
    for (auto &ev : b->ev) if (hipEventCreate(&ev) != hipSuccess) return fail(LPCN_E_HIP);
    if (hipEventCreateWithFlags(&b->ev_last, hipEventDisableTiming) != hipSuccess) return fail(LPCN_E_HIP);
    *out = b;
    int rc = lpcn_batch_dev_reset(b, 0, n);
    if (rc) return fail(rc);
    return 0;
}

extern "C" void lpcn_batch_dev_destroy(lpcn_batch_dev *b)
{
    if (!b) return;
    DeviceGuard guard(b->e->device);
    (void)wait_all(b);
    if (b->h_pin) (void)hipHostFree(b->h_pin);
    if (b->ev_last) (void)hipEventDestroy(b->ev_last);
    void *ptrs[] = {b->d_state, b->d_fc_base, b->d_cond_a, b->d_cond_b, b->d_lpc, b->d_cond, b->d_feat, b->d_pcm, b->d_args, b->d_dbg, b->d_prof,
                    b->d_vq_mem, b->d_packets, b->d_state_tmp, b->d_map, b->d_keep_a, b->d_keep_b, b->d_keep_lpc};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    for (auto &ev : b->ev) if (ev) (void)hipEventDestroy(ev);
    delete b;
}

// lpcnet_reset semantics (src/lpcnet.c:174-182): zero everything, last_exc = lin2ulaw(0) = 128,
// RNG seeded from the string "LPCNet" (src/kiss99.c:34-57, evaluated on the host).
static void host_reset_state(lpcn_stream_state *st)
{
    memset(st, 0, sizeof(*st));
    st->last_exc = 128;
    uint32_t c[4] = {362436069u, 521288629u, 123456789u, 380116160u};
    const unsigned char d[6] = {'L', 'P', 'C', 'N', 'e', 't'};
    c[0] ^= d[0]; c[1] ^= d[1]; c[2] ^= d[2]; c[3] ^= d[3];
    lpcn_kiss99(c);
    c[0] ^= d[4]; c[1] ^= d[5];
    if (c[0] == 0 || c[0] == 0x9068FFFFu) c[0]++;
    if (c[1] == 0 || c[1] == 0x464FFFFFu) c[1]++;
    if (c[2] == 0) c[2]++;
    memcpy(st->rng, c, sizeof(c));
}

extern "C" int lpcn_batch_dev_reset(lpcn_batch_dev *b, int first, int count)
{
    if (!b->keep_ok.empty()) b->keep_ok.assign(b->keep_ok.size(), 0);      // (lpcn_batch_dev_step_host's per-stream frame products are stale now)
    if (first < 0 || count < 0 || first + count > b->n) { snprintf(g_err, sizeof(g_err), "reset range"); return LPCN_E_ARG; }
    DeviceGuard guard(b->e->device);
    std::vector<lpcn_stream_state> h(count);
    for (auto &s : h) host_reset_state(&s);
    { int rcw = wait_all(b); if (rcw) return rcw; }
    HIP_TRY(hipMemcpy(b->d_state + first, h.data(), sizeof(lpcn_stream_state) * count, hipMemcpyHostToDevice));
    if (count) HIP_TRY(hipMemset(b->d_vq_mem + (size_t)first * LPCN_NB_BANDS, 0, sizeof(float) * (size_t)count * LPCN_NB_BANDS));
    return 0;
}

extern "C" int lpcn_batch_dev_get_state(lpcn_batch_dev *b, int s, lpcn_stream_state *host)
{
    if (s < 0 || s >= b->n) { snprintf(g_err, sizeof(g_err), "stream index"); return LPCN_E_ARG; }
    DeviceGuard guard(b->e->device);
    { int rcw = wait_all(b); if (rcw) return rcw; }
    HIP_TRY(hipMemcpy(host, b->d_state + s, sizeof(*host), hipMemcpyDeviceToHost));
    return 0;
}
extern "C" int lpcn_batch_dev_set_state(lpcn_batch_dev *b, int s, const lpcn_stream_state *host)
{
    if (!b->keep_ok.empty()) b->keep_ok.assign(b->keep_ok.size(), 0);      // (lpcn_batch_dev_step_host's per-stream frame products are stale now)
    if (s < 0 || s >= b->n) { snprintf(g_err, sizeof(g_err), "stream index"); return LPCN_E_ARG; }
    DeviceGuard guard(b->e->device);
    { int rcw = wait_all(b); if (rcw) return rcw; }
    HIP_TRY(hipMemcpy(b->d_state + s, host, sizeof(*host), hipMemcpyHostToDevice));
    return 0;
}
extern "C" int lpcn_batch_dev_streams_per_wg(const lpcn_batch_dev *b) { return b->S; }
// the engine's arithmetic flavour changed: re-run the cost model unless the caller pinned the value
extern "C" int lpcn_batch_dev_retune(lpcn_batch_dev *b)
{
    if (b->S_auto) { b->S = auto_streams_per_wg(b->e, b->n); b->tuned = false; }      // measured again at the next run
    b->pack2 = use_pack2(b->e, b->n, b->S);
    return 0;
}
extern "C" int lpcn_batch_dev_set_streams_per_wg(lpcn_batch_dev *b, int s)
{
    b->S_auto = s == 0;
    if (s == 0) { s = auto_streams_per_wg(b->e, b->n); b->tuned = false; }
    if (s == 8 && !x2_available(b->e)) { snprintf(g_err, sizeof(g_err), "eight streams per workgroup need a float blob with a dense GRU-B matrix and <= 32 items per lane, PARITY arithmetic"); return LPCN_E_ARG; }
    if (s != 1 && s != 2 && s != 4 && s != 8) { snprintf(g_err, sizeof(g_err), "streams per workgroup must be 1, 2, 4 or 8"); return LPCN_E_ARG; }
    b->S = s;
    b->pack2 = use_pack2(b->e, b->n, b->S);
    return 0;
}
extern "C" int lpcn_batch_dev_set_frame_len(lpcn_batch_dev *b, int n)
{
    if (n < 1 || n > LPCN_FRAME_SIZE) { snprintf(g_err, sizeof(g_err), "frame length must be 1..160"); return LPCN_E_ARG; }
    b->frame_len = n;
    return 0;
}
extern "C" int lpcn_batch_dev_enable_timing(lpcn_batch_dev *b, int on) { b->timing = on != 0; return 0; }
extern "C" int lpcn_batch_dev_last_timing(lpcn_batch_dev *b, float *ms_sample, float *ms_frame)
{
    if (ms_sample) *ms_sample = b->ms_sample;
    if (ms_frame) *ms_frame = b->ms_frame;
    return 0;
}
extern "C" int lpcn_batch_dev_sync(lpcn_batch_dev *b)
{
    DeviceGuard guard(b->e->device);
    return wait_all(b);
}

// ------------------------------------------------------------------------------- launches -----
// The sample kernel's variants (streams per workgroup x items per lane x blob flavour x arithmetic) are compiled in
// separate translation units, one per streams-per-workgroup value (sample_variants.hip), so they build in parallel.
extern "C" int lpcn_launch_sample_s1(int nw, int is_int8, int flags, int grid, int lds, hipStream_t st, const LpcnSampleArgs *d_args);
extern "C" int lpcn_launch_sample_s2(int nw, int is_int8, int flags, int grid, int lds, hipStream_t st, const LpcnSampleArgs *d_args);
extern "C" int lpcn_launch_sample_s4(int nw, int is_int8, int flags, int grid, int lds, hipStream_t st, const LpcnSampleArgs *d_args);
extern "C" int lpcn_launch_sample_x2(int nw, int grid, int lds, hipStream_t st, const LpcnSampleArgs *d_args);      // sample_x2.hip: two groups of four streams
extern "C" int lpcn_x2_lds_bytes(int nb_b);

// one chunk of the per-sample kernel; cond_a/cond_b/lpc for the chunk are already in the batch buffers
static int launch_sample(lpcn_batch_dev *b, hipStream_t st, short *d_pcm, size_t pcm_stride, int n_frames,
                         int preload, bool fc_from_frames)
{
    if (b->S == 8 && !x2_available(b->e)) { b->S = 4; b->pack2 = false; }      // (the arithmetic flavour changed under a pinned value)
    // (the two-group kernel addresses the launch's conditioning rows with 32-bit byte offsets: launches beyond 4 GB of them -- more than ~9 300 streams x
    // 100 frames -- run on the four-stream kernel)
    const int S_keep = b->S;
    if (b->S == 8 && (unsigned long long)b->n * (unsigned long long)n_frames * LPCN_ROWS_A * 4ull >= (1ull << 32)) b->S = 4;
    struct SRestore { lpcn_batch_dev *b; int s; ~SRestore() { b->S = s; } } s_restore{b, S_keep};
    LpcnSampleArgs a = b->S == 8 ? b->e->sargs_x2 : ((b->e->fast && b->e->has_fast_image) ? b->e->sargs_fast : b->e->sargs);
    a.n_streams = b->n; a.n_frames = n_frames; a.preload = preload; a.frame_len = b->frame_len;
    a.fc_advance = fc_from_frames ? 1 : 0;
    a.cond_a = b->d_cond_a; a.cond_b = b->d_cond_b; a.lpc = b->d_lpc;
    a.fc_base = fc_from_frames ? b->d_fc_base : nullptr;
    a.pcm = d_pcm; a.pcm_stride = (long long)pcm_stride;
    a.state = b->d_state; a.dbg = b->d_dbg; a.prof = b->d_prof;
    a.fc_f16 = (b->e->fast && b->e->fc_f16) ? 1 : 0;
    if (stream_is_capturing(st)) {
        hipLaunchKernelGGL(lpcn_set_args_kernel, dim3(1), dim3(1), 0, st, b->d_args, a);
        HIP_TRY(hipGetLastError());
    } else {
        HIP_TRY(hipMemcpyAsync(b->d_args, &a, sizeof(a), hipMemcpyHostToDevice, st));
    }
    const int i8 = b->e->is_int8 ? 1 : 0;
    const int fast = (b->e->fast ? 1 : 0) | (b->pack2 ? 2 : 0);
    const int nwv = cur_nw_variant(b->e);
    int lds = 0, rc = 0;
    const int grid = (b->n + b->S - 1) / b->S;
    switch (b->S) {
    case 8: lds = lpcn_x2_lds_bytes(b->e->nb_b); rc = lpcn_launch_sample_x2(b->e->nw_variant_x2, grid, lds, st, b->d_args); break;
    case 1: lds = lpcn::Lds<1>::total(b->e->nb_b, b->e->is_int8); rc = lpcn_launch_sample_s1(nwv, i8, fast, grid, lds, st, b->d_args); break;
    case 2: lds = lpcn::Lds<2>::total(b->e->nb_b, b->e->is_int8); rc = lpcn_launch_sample_s2(nwv, i8, fast, grid, lds, st, b->d_args); break;
    default: lds = lpcn::Lds<4>::total(b->e->nb_b, b->e->is_int8); rc = lpcn_launch_sample_s4(nwv, i8, fast, grid, lds, st, b->d_args); break;
    }
    if (rc) { snprintf(g_err, sizeof(g_err), "sample kernel launch failed: %s", hipGetErrorString((hipError_t)rc)); return LPCN_E_HIP; }
    return 0;
}

static int launch_frames(lpcn_batch_dev *b, hipStream_t st, const float *d_feat, int feat_stride,
                         size_t feat_stream_stride, int n_frames)
{
    return lpcn_launch_frame_kernels(b->e->fmodel, st, b->n, n_frames, d_feat, feat_stride, feat_stream_stride,
                                     b->d_state, b->d_fc_base, b->d_cond, b->d_cond_a, b->d_cond_b, b->d_lpc, g_err, sizeof(g_err));
}

// Streams per workgroup, measured instead of looked up (the table in auto_streams_per_wg holds THIS model's step
// times; a denser / sparser blob or another item-count variant has other optima).  Before the first run of a batch whose
// value is not pinned, each candidate S runs live frames on the batch's own buffers -- zeroed frame products, the stream
// states saved and restored around it -- timed with HIP events: a warm-up launch, then one frame and five frames; the
// DIFFERENCE is the steady-state cost of four frames (a launch's prologue -- filling registers and LDS with the weights --
// is ~4 % of a two-frame launch and grows with the workgroup count, which biased a plain two-frame timing against the
// two-workgroups-per-CU form of the int8 kernel).  ~9 launches, ~12 ms, once per batch and arithmetic flavour;
// LPCNET_HIP_NO_AUTOTUNE=1 keeps the table value.
// Round 4: (a) the FAST flavour is never timed -- its float kernels switch GRU-B's algorithm with S (matrix-pipe GEMM at S >= 2,
// fused DPP chains at S = 1: different summation orders), so a measured S would make FAST output depend on timing noise;
// it takes the table's value, a pure function of (blob kind, stream count, device).  (b) the measurement is the best of
// three timing pairs.  (c) it never runs inside the enqueue-only device-pointer calls on a caller's stream (those take the
// table's value unless lpcnet_batch_tune() has been called): it allocates, synchronises and would break stream capture.
static int autotune_streams_per_wg(lpcn_batch_dev *b, hipStream_t st)
{
    b->tuned = true;
    const char *off = getenv("LPCNET_HIP_NO_AUTOTUNE");
    if ((off && *off == '1') || b->n < 2 || b->e->fast) return 0;     // (one stream: one workgroup whatever S is)
    const int nf = 5 < b->max_chunk ? 5 : b->max_chunk;
    int rc = 0;
    lpcn_stream_state *saved = nullptr;
    short *pcm = nullptr;                                    // (a buffer of its own: the caller may be holding the staging buffer's address)
    HIP_TRY(hipMalloc((void **)&saved, sizeof(lpcn_stream_state) * b->n));
    if (hipMalloc((void **)&pcm, sizeof(short) * (size_t)b->n * nf * LPCN_FRAME_SIZE) != hipSuccess) { (void)hipFree(saved); snprintf(g_err, sizeof(g_err), "auto-tune: hipMalloc failed"); return LPCN_E_HIP; }
    auto done = [&](int code) { (void)hipFree(saved); (void)hipFree(pcm); return code; };
    std::vector<int> fc((size_t)b->n, LPCN_FEATURES_DELAY + 3);                     // every stream live
    if (hipMemcpyAsync(saved, b->d_state, sizeof(lpcn_stream_state) * b->n, hipMemcpyDeviceToDevice, st) != hipSuccess ||
        hipMemsetAsync(b->d_cond_a, 0, sizeof(float) * (size_t)b->n * nf * LPCN_ROWS_A, st) != hipSuccess ||
        hipMemsetAsync(b->d_cond_b, 0, sizeof(float) * (size_t)b->n * nf * LPCN_ROWS_B, st) != hipSuccess ||
        hipMemsetAsync(b->d_lpc, 0, sizeof(float) * (size_t)b->n * nf * LPCN_LPC_ORDER, st) != hipSuccess ||
        hipMemcpyAsync(b->d_fc_base, fc.data(), sizeof(int) * b->n, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) { snprintf(g_err, sizeof(g_err), "auto-tune: buffer setup failed"); return done(LPCN_E_HIP); }
    const int keepS = b->S;
    const bool keepP = b->pack2;
    int best = keepS;
    float best_ms = -1.f;
    for (int S = 1; S <= 8; S *= 2) {
        if (S == 8 && (!x2_available(b->e) || b->n <= 4)) break;
        b->S = S; b->pack2 = use_pack2(b->e, b->n, S);
        float ms = -1.f, ms1 = 0.f;
        for (int pass = 0; pass < 7 && !rc; ++pass) {          // warm-up (1 frame), then three pairs of (1 frame, nf frames): the smallest difference
            const int k = (pass && !(pass & 1)) ? nf : 1;
            float t = 0.f;
            if (hipEventRecord(b->ev[1], st) != hipSuccess) rc = LPCN_E_HIP;
            if (!rc) rc = launch_sample(b, st, pcm, (size_t)nf * LPCN_FRAME_SIZE, k, 0, true);
            if (!rc && (hipEventRecord(b->ev[2], st) != hipSuccess || hipEventSynchronize(b->ev[2]) != hipSuccess ||
                        hipEventElapsedTime(&t, b->ev[1], b->ev[2]) != hipSuccess)) rc = LPCN_E_HIP;
            if (pass & 1) ms1 = t;
            else if (pass) { const float d = nf > 1 ? t - ms1 : t; if (ms < 0.f || d < ms) ms = d; }      // nf - 1 frames in steady state
        }
        if (rc) break;
        if (best_ms < 0.f || ms < best_ms) { best_ms = ms; best = S; }
    }
    // the measurement ran on the real state: put it back
    if (hipMemcpyAsync(b->d_state, saved, sizeof(lpcn_stream_state) * b->n, hipMemcpyDeviceToDevice, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) rc = rc ? rc : LPCN_E_HIP;
    if (rc) { b->S = keepS; b->pack2 = keepP; return done(rc); }
    b->S = best; b->pack2 = use_pack2(b->e, b->n, best);
    return done(0);
}

// lpcnet_batch_tune(): measure now, on the engine's own stream (e.g. right after lpcnet_batch_load_model), so that the first
// synthesize call -- in particular an enqueue-only one on a caller's stream -- carries no measurement
extern "C" int lpcn_batch_dev_tune(lpcn_batch_dev *b)
{
    DeviceGuard guard(b->e->device);
    if (!b->S_auto) return 0;
    hipStream_t st = b->e->stream;
    int rc = order_begin(b, st);
    if (rc) return rc;
    if ((rc = autotune_streams_per_wg(b, st))) return rc;
    return order_end(b, st);
}

// may_tune: the call may measure the streams per workgroup first (allocates, synchronises, launches trial kernels).  Only the
// host-pointer wrappers pass true; the enqueue-only device-pointer entry points never do, whichever stream they name (ADVICE r4:
// a NULL hip_stream used to select the engine's own stream AND the measurement).
static int run_impl(lpcn_batch_dev *b, const float *d_features, int feat_stride, short *d_pcm, int n_frames, int preload, void *hip_stream, bool may_tune)
{
    if (n_frames <= 0 || feat_stride < LPCN_NB_FEAT || preload < 0 || preload > LPCN_FRAME_SIZE) {
        snprintf(g_err, sizeof(g_err), "bad run arguments"); return LPCN_E_ARG;
    }
    DeviceGuard guard(b->e->device);
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : b->e->stream;
    float tf = 0.f, ts = 0.f;
    { int rco = order_begin(b, st); if (rco) return rco; }
    if (!b->keep_ok.empty()) b->keep_ok.assign(b->keep_ok.size(), 0);      // (the step call's per-stream frame products are stale now)
    if (b->S_auto && !b->tuned && may_tune && st == b->e->stream) { int rct = autotune_streams_per_wg(b, st); if (rct) return rct; }      // (never inside an enqueue-only call, never on a caller's stream)
    for (int f0 = 0; f0 < n_frames; f0 += b->max_chunk) {
        const int nf = n_frames - f0 < b->max_chunk ? n_frames - f0 : b->max_chunk;
        if (b->timing) HIP_TRY(hipEventRecord(b->ev[0], st));
        int rc = launch_frames(b, st, d_features + (size_t)f0 * feat_stride, feat_stride, (size_t)n_frames * feat_stride, nf);
        if (rc) return rc;
        if (b->timing) HIP_TRY(hipEventRecord(b->ev[1], st));
        rc = launch_sample(b, st, d_pcm + (size_t)f0 * LPCN_FRAME_SIZE, (size_t)n_frames * LPCN_FRAME_SIZE, nf, preload, true);
        if (rc) return rc;
        if (b->timing) {
            HIP_TRY(hipEventRecord(b->ev[2], st));
            HIP_TRY(hipEventSynchronize(b->ev[2]));
            float a = 0.f, c = 0.f;
            HIP_TRY(hipEventElapsedTime(&a, b->ev[0], b->ev[1]));
            HIP_TRY(hipEventElapsedTime(&c, b->ev[1], b->ev[2]));
            tf += a; ts += c;
        }
    }
    if (b->timing) { b->ms_frame = tf; b->ms_sample = ts; }
    return order_end(b, st);
}

extern "C" int lpcn_batch_dev_run(lpcn_batch_dev *b, const float *d_features, int feat_stride,
                                  short *d_pcm, int n_frames, int preload, void *hip_stream)
{
    return run_impl(b, d_features, feat_stride, d_pcm, n_frames, preload, hip_stream, false);
}

static int ensure_staging(lpcn_batch_dev *b, size_t feat_floats, size_t pcm_samples)
{
    if (feat_floats > b->feat_cap || pcm_samples > b->pcm_cap) { int rcw = wait_all(b); if (rcw) return rcw; }   // the old buffers may still be in use
    if (feat_floats > b->feat_cap) {
        if (b->d_feat) (void)hipFree(b->d_feat);
        b->d_feat = nullptr; b->feat_cap = 0;
        HIP_TRY(hipMalloc((void **)&b->d_feat, feat_floats * sizeof(float)));
        b->feat_cap = feat_floats;
    }
    if (pcm_samples > b->pcm_cap) {
        if (b->d_pcm) (void)hipFree(b->d_pcm);
        b->d_pcm = nullptr; b->pcm_cap = 0;
        HIP_TRY(hipMalloc((void **)&b->d_pcm, pcm_samples * sizeof(short)));
        b->pcm_cap = pcm_samples;
    }
    return 0;
}

extern "C" int lpcn_batch_dev_run_host(lpcn_batch_dev *b, const float *features, int feat_stride,
                                       short *pcm, int n_frames, int preload)
{
    if (n_frames <= 0) { snprintf(g_err, sizeof(g_err), "bad run arguments"); return LPCN_E_ARG; }
    DeviceGuard guard(b->e->device);
    const size_t nfeat = (size_t)b->n * n_frames * feat_stride, npcm = (size_t)b->n * n_frames * LPCN_FRAME_SIZE;
    int rc = ensure_staging(b, nfeat, npcm);
    if (rc) return rc;
    hipStream_t st = b->e->stream;
    if ((rc = order_begin(b, st))) return rc;      // the staging buffers may still be read by work on a caller stream
    HIP_TRY(hipMemcpyAsync(b->d_feat, features, nfeat * sizeof(float), hipMemcpyHostToDevice, st));
    if (preload > 0) HIP_TRY(hipMemcpyAsync(b->d_pcm, pcm, npcm * sizeof(short), hipMemcpyHostToDevice, st));
    rc = run_impl(b, b->d_feat, feat_stride, b->d_pcm, n_frames, preload, st, true);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(pcm, b->d_pcm, npcm * sizeof(short), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

// Single-stream fast path of the legacy per-frame API (a batch of one stream): one frame-network step + frame_len samples
// with ONE host synchronisation.  The caller's POD state is uploaded only when it differs from the device copy
// (st_in == NULL: the device copy is current); features, state and PCM travel through one pinned buffer.
extern "C" int lpcn_batch_dev_run_single(lpcn_batch_dev *b, const lpcn_stream_state *st_in, const float *feat, short *pcm,
                                         lpcn_stream_state *st_out)
{
    if (b->n != 1) { snprintf(g_err, sizeof(g_err), "run_single needs a batch of one stream"); return LPCN_E_ARG; }
    DeviceGuard guard(b->e->device);
    const size_t off_feat = sizeof(lpcn_stream_state), off_pcm = off_feat + LPCN_NB_FEAT * sizeof(float);
    int rc = ensure_staging(b, LPCN_NB_FEAT, LPCN_FRAME_SIZE);
    if (rc) return rc;
    if (!b->h_pin) HIP_TRY(hipHostMalloc(&b->h_pin, off_pcm + LPCN_FRAME_SIZE * sizeof(short), hipHostMallocDefault));
    hipStream_t st = b->e->stream;
    if ((rc = order_begin(b, st))) return rc;
    unsigned char *pin = (unsigned char *)b->h_pin;
    if (st_in) {
        memcpy(pin, st_in, sizeof(*st_in));
        HIP_TRY(hipMemcpyAsync(b->d_state, pin, sizeof(*st_in), hipMemcpyHostToDevice, st));
    }
    memcpy(pin + off_feat, feat, LPCN_NB_FEAT * sizeof(float));
    HIP_TRY(hipMemcpyAsync(b->d_feat, pin + off_feat, LPCN_NB_FEAT * sizeof(float), hipMemcpyHostToDevice, st));
    rc = run_impl(b, b->d_feat, LPCN_NB_FEAT, b->d_pcm, 1, 0, st, true);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(pin, b->d_state, sizeof(lpcn_stream_state), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(pin + off_pcm, b->d_pcm, LPCN_FRAME_SIZE * sizeof(short), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    b->pending = false;
    memcpy(st_out, pin, sizeof(*st_out));
    memcpy(pcm, pin + off_pcm, (size_t)b->frame_len * sizeof(short));
    return 0;
}

// Combined pass of the legacy per-frame API (api.c: callers of one model that arrive while a pass is in flight are served together):
// k <= capacity independent streams, each with its caller's POD state -- states up, kernels, states and results down, ONE
// synchronisation.  The batch was created with the capacity as its stream count; a pass uses its first k slots.  `kind`:
//   LPCN_GROUP_FRAME_SAMPLES  frame network on feat[i] + frame_len samples (lpcnet_synthesize / lpcnet_synthesize_impl, src/lpcnet.c:273-281);
//                             the first `preload` samples of pcm[i] are imposed (teacher forcing); the frame products come back in
//                             ga[i] / gb[i] (the reference keeps them in the state for a later tail call; lpc is part of the state record)
//   LPCN_GROUP_TAIL           frame_len samples from the caller's products ga[i] / gb[i] / lpc[i] (lpcnet_synthesize_tail_impl, :235-271)
//   LPCN_GROUP_FRAMES         frame network only; products returned in ga[i] / gb[i] / lpc[i] (run_frame_network, :82-120)
// Nothing is written to the callers' memory unless the whole pass succeeded.
extern "C" int lpcn_batch_dev_run_group(lpcn_batch_dev *b, int k, int kind, int frame_len, int preload, const lpcn_stream_state *const *st_in,
                                        const float *const *feat, short *const *pcm, lpcn_stream_state *const *st_out,
                                        float *const *ga, float *const *gb, float *const *lpc)
{
    const bool samples = kind != LPCN_GROUP_FRAMES, frames = kind != LPCN_GROUP_TAIL;
    if (k < 1 || kind < 0 || kind > 2 || (samples && (frame_len < 1 || frame_len > LPCN_FRAME_SIZE || preload < 0 || preload > frame_len))) {
        snprintf(g_err, sizeof(g_err), "bad group arguments"); return LPCN_E_ARG;
    }
    DeviceGuard guard(b->e->device);
    const int cap = b->n;
    if (k > cap) { snprintf(g_err, sizeof(g_err), "group of %d exceeds the batch's %d streams", k, cap); return LPCN_E_ARG; }
    int rc = ensure_staging(b, (size_t)cap * LPCN_NB_FEAT, (size_t)cap * LPCN_FRAME_SIZE);
    if (rc) return rc;
    // pinned staging: states | features | pcm | cond_a | cond_b | lpc, each for `cap` streams
    const size_t sz_st = sizeof(lpcn_stream_state), off_feat = sz_st * cap, off_pcm = off_feat + sizeof(float) * LPCN_NB_FEAT * cap;
    const size_t off_ga = off_pcm + sizeof(short) * LPCN_FRAME_SIZE * cap, off_gb = off_ga + sizeof(float) * LPCN_ROWS_A * cap;
    const size_t off_lpc = off_gb + sizeof(float) * LPCN_ROWS_B * cap, pin_bytes = off_lpc + sizeof(float) * LPCN_LPC_ORDER * cap;
    if (!b->h_pin) HIP_TRY(hipHostMalloc(&b->h_pin, pin_bytes, hipHostMallocDefault));
    hipStream_t st = b->e->stream;
    if ((rc = order_begin(b, st))) return rc;
    unsigned char *pin = (unsigned char *)b->h_pin;
    for (int i = 0; i < k; ++i) {
        memcpy(pin + sz_st * i, st_in[i], sz_st);
        if (frames) memcpy(pin + off_feat + sizeof(float) * LPCN_NB_FEAT * i, feat[i], sizeof(float) * LPCN_NB_FEAT);
        if (samples && preload > 0) memcpy(pin + off_pcm + sizeof(short) * LPCN_FRAME_SIZE * i, pcm[i], sizeof(short) * (size_t)preload);
        if (kind == LPCN_GROUP_TAIL) {
            memcpy(pin + off_ga + sizeof(float) * LPCN_ROWS_A * i, ga[i], sizeof(float) * LPCN_ROWS_A);
            memcpy(pin + off_gb + sizeof(float) * LPCN_ROWS_B * i, gb[i], sizeof(float) * LPCN_ROWS_B);
            memcpy(pin + off_lpc + sizeof(float) * LPCN_LPC_ORDER * i, lpc[i], sizeof(float) * LPCN_LPC_ORDER);
        }
    }
    HIP_TRY(hipMemcpyAsync(b->d_state, pin, sz_st * k, hipMemcpyHostToDevice, st));
    if (frames) HIP_TRY(hipMemcpyAsync(b->d_feat, pin + off_feat, sizeof(float) * LPCN_NB_FEAT * k, hipMemcpyHostToDevice, st));
    if (samples && preload > 0) HIP_TRY(hipMemcpyAsync(b->d_pcm, pin + off_pcm, sizeof(short) * LPCN_FRAME_SIZE * k, hipMemcpyHostToDevice, st));
    if (kind == LPCN_GROUP_TAIL) {     // (one frame per stream: the chunk buffers are dense [stream][...])
        HIP_TRY(hipMemcpyAsync(b->d_cond_a, pin + off_ga, sizeof(float) * LPCN_ROWS_A * k, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(b->d_cond_b, pin + off_gb, sizeof(float) * LPCN_ROWS_B * k, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(b->d_lpc, pin + off_lpc, sizeof(float) * LPCN_LPC_ORDER * k, hipMemcpyHostToDevice, st));
    }
    const int keepS = b->S, keep_len = b->frame_len;
    const bool keepP = b->pack2, keepA = b->S_auto, keepT = b->tuned;
    b->n = k; b->frame_len = samples ? frame_len : keep_len; b->S = auto_streams_per_wg(b->e, k); b->pack2 = use_pack2(b->e, k, b->S); b->S_auto = false; b->tuned = true;
    if (!b->keep_ok.empty()) b->keep_ok.assign(b->keep_ok.size(), 0);
    if (kind == LPCN_GROUP_FRAME_SAMPLES) rc = run_impl(b, b->d_feat, LPCN_NB_FEAT, b->d_pcm, 1, preload, st, false);      // (S comes from the table for the group's size)
    else if (kind == LPCN_GROUP_TAIL) rc = launch_sample(b, st, b->d_pcm, (size_t)LPCN_FRAME_SIZE, 1, preload, false);
    else rc = launch_frames(b, st, b->d_feat, LPCN_NB_FEAT, (size_t)LPCN_NB_FEAT, 1);
    b->n = cap; b->frame_len = keep_len; b->S = keepS; b->pack2 = keepP; b->S_auto = keepA; b->tuned = keepT;
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(pin, b->d_state, sz_st * k, hipMemcpyDeviceToHost, st));
    if (samples) HIP_TRY(hipMemcpyAsync(pin + off_pcm, b->d_pcm, sizeof(short) * LPCN_FRAME_SIZE * k, hipMemcpyDeviceToHost, st));
    if (frames) {
        HIP_TRY(hipMemcpyAsync(pin + off_ga, b->d_cond_a, sizeof(float) * LPCN_ROWS_A * k, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(pin + off_gb, b->d_cond_b, sizeof(float) * LPCN_ROWS_B * k, hipMemcpyDeviceToHost, st));
        if (kind == LPCN_GROUP_FRAMES) HIP_TRY(hipMemcpyAsync(pin + off_lpc, b->d_lpc, sizeof(float) * LPCN_LPC_ORDER * k, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    b->pending = false;
    for (int i = 0; i < k; ++i) {
        memcpy(st_out[i], pin + sz_st * i, sz_st);
        // live frames return the imposed samples unchanged; start-up frames are cleared entirely (src/lpcnet.c:239-243)
        if (samples) memcpy(pcm[i], pin + off_pcm + sizeof(short) * LPCN_FRAME_SIZE * i, sizeof(short) * (size_t)frame_len);
        if (frames && ga[i]) {                               // (ga[i] == NULL: this caller does not take the frame products back)
            memcpy(ga[i], pin + off_ga + sizeof(float) * LPCN_ROWS_A * i, sizeof(float) * LPCN_ROWS_A);
            memcpy(gb[i], pin + off_gb + sizeof(float) * LPCN_ROWS_B * i, sizeof(float) * LPCN_ROWS_B);
            if (kind == LPCN_GROUP_FRAMES) memcpy(lpc[i], pin + off_lpc + sizeof(float) * LPCN_LPC_ORDER * i, sizeof(float) * LPCN_LPC_ORDER);
        }
    }
    return 0;
}

// Codec path: 8-byte packets [stream][packet][8] -> 4 frames each.  Device pointers, work only enqueued.
static int decode_impl(lpcn_batch_dev *b, const unsigned char *d_packets, short *d_pcm, int n_packets, void *hip_stream, bool may_tune)
{
    if (!b->keep_ok.empty()) b->keep_ok.assign(b->keep_ok.size(), 0);      // (lpcn_batch_dev_step_host's per-stream frame products are stale now)
    if (n_packets <= 0) { snprintf(g_err, sizeof(g_err), "bad decode arguments"); return LPCN_E_ARG; }
    if (!b->e->has_codebooks) { snprintf(g_err, sizeof(g_err), "no VQ codebooks installed (lpcnet_hip_set_codebooks)"); return LPCN_E_MODEL; }
    DeviceGuard guard(b->e->device);
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : b->e->stream;
    const int T = 4 * n_packets;
    {   // (growing the staging buffer waits for the batch's outstanding work first; it happens once per size)
        int rc = ensure_staging(b, (size_t)b->n * T * LPCN_NB_FEAT, 0);
        if (rc) return rc;
        if ((rc = order_begin(b, st))) return rc;
    }
    hipLaunchKernelGGL(lpcn::decode_kernel, dim3((b->n + 1) / 2), dim3(64), 0, st, b->e->dec, d_packets, b->n, n_packets, b->d_vq_mem,
                       b->d_feat, LPCN_NB_FEAT);
    HIP_TRY(hipGetLastError());
    return run_impl(b, b->d_feat, LPCN_NB_FEAT, d_pcm, T, 0, st, may_tune);
}
extern "C" int lpcn_batch_dev_decode(lpcn_batch_dev *b, const unsigned char *d_packets, short *d_pcm, int n_packets, void *hip_stream)
{
    return decode_impl(b, d_packets, d_pcm, n_packets, hip_stream, false);
}

extern "C" int lpcn_batch_dev_decode_host(lpcn_batch_dev *b, const unsigned char *packets, short *pcm, int n_packets)
{
    if (n_packets <= 0) { snprintf(g_err, sizeof(g_err), "bad decode arguments"); return LPCN_E_ARG; }
    DeviceGuard guard(b->e->device);
    const size_t nbytes = (size_t)b->n * n_packets * 8, npcm = (size_t)b->n * n_packets * 4 * LPCN_FRAME_SIZE;
    if (nbytes > b->packets_cap) {
        { int rcw = wait_all(b); if (rcw) return rcw; }
        if (b->d_packets) (void)hipFree(b->d_packets);
        b->d_packets = nullptr; b->packets_cap = 0;
        HIP_TRY(hipMalloc((void **)&b->d_packets, nbytes));
        b->packets_cap = nbytes;
    }
    int rc = ensure_staging(b, 0, npcm);
    if (rc) return rc;
    hipStream_t st = b->e->stream;
    if ((rc = order_begin(b, st))) return rc;
    HIP_TRY(hipMemcpyAsync(b->d_packets, packets, nbytes, hipMemcpyHostToDevice, st));
    rc = decode_impl(b, b->d_packets, b->d_pcm, n_packets, nullptr, true);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(pcm, b->d_pcm, npcm * sizeof(short), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

extern "C" int lpcn_batch_dev_run_tail_host(lpcn_batch_dev *b, const float *cond_a, const float *cond_b,
                                            const float *lpc, short *pcm, int n_frames, int preload)
{
    if (!b->keep_ok.empty()) b->keep_ok.assign(b->keep_ok.size(), 0);      // (lpcn_batch_dev_step_host's per-stream frame products are stale now)
    if (n_frames <= 0 || preload < 0 || preload > LPCN_FRAME_SIZE) { snprintf(g_err, sizeof(g_err), "bad run arguments"); return LPCN_E_ARG; }
    DeviceGuard guard(b->e->device);
    const size_t npcm = (size_t)b->n * n_frames * LPCN_FRAME_SIZE;
    int rc = ensure_staging(b, 0, npcm);
    if (rc) return rc;
    hipStream_t st = b->e->stream;
    if ((rc = order_begin(b, st))) return rc;
    if (b->S_auto && !b->tuned && (rc = autotune_streams_per_wg(b, st))) return rc;
    if (preload > 0) HIP_TRY(hipMemcpyAsync(b->d_pcm, pcm, npcm * sizeof(short), hipMemcpyHostToDevice, st));
    for (int f0 = 0; f0 < n_frames; f0 += b->max_chunk) {
        const int nf = n_frames - f0 < b->max_chunk ? n_frames - f0 : b->max_chunk;
        // gather the chunk [stream][f0..f0+nf) into the dense chunk buffers
        HIP_TRY(hipMemcpy2DAsync(b->d_cond_a, (size_t)nf * LPCN_ROWS_A * 4, cond_a + (size_t)f0 * LPCN_ROWS_A, (size_t)n_frames * LPCN_ROWS_A * 4,
                                 (size_t)nf * LPCN_ROWS_A * 4, b->n, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpy2DAsync(b->d_cond_b, (size_t)nf * LPCN_ROWS_B * 4, cond_b + (size_t)f0 * LPCN_ROWS_B, (size_t)n_frames * LPCN_ROWS_B * 4,
                                 (size_t)nf * LPCN_ROWS_B * 4, b->n, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpy2DAsync(b->d_lpc, (size_t)nf * LPCN_LPC_ORDER * 4, lpc + (size_t)f0 * LPCN_LPC_ORDER, (size_t)n_frames * LPCN_LPC_ORDER * 4,
                                 (size_t)nf * LPCN_LPC_ORDER * 4, b->n, hipMemcpyHostToDevice, st));
        if (b->timing) HIP_TRY(hipEventRecord(b->ev[1], st));
        rc = launch_sample(b, st, b->d_pcm + (size_t)f0 * LPCN_FRAME_SIZE, (size_t)n_frames * LPCN_FRAME_SIZE, nf, preload, false);
        if (rc) return rc;
        if (b->timing) {
            HIP_TRY(hipEventRecord(b->ev[2], st));
            HIP_TRY(hipEventSynchronize(b->ev[2]));
            HIP_TRY(hipEventElapsedTime(&b->ms_sample, b->ev[1], b->ev[2]));
        }
        HIP_TRY(hipStreamSynchronize(st));
    }
    HIP_TRY(hipMemcpyAsync(pcm, b->d_pcm, npcm * sizeof(short), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

extern "C" int lpcn_batch_dev_run_frames_host(lpcn_batch_dev *b, const float *features, int feat_stride,
                                              float *cond_a, float *cond_b, float *lpc, int n_frames)
{
    if (!b->keep_ok.empty()) b->keep_ok.assign(b->keep_ok.size(), 0);      // (lpcn_batch_dev_step_host's per-stream frame products are stale now)
    if (n_frames <= 0 || feat_stride < LPCN_NB_FEAT) { snprintf(g_err, sizeof(g_err), "bad run arguments"); return LPCN_E_ARG; }
    DeviceGuard guard(b->e->device);
    const size_t nfeat = (size_t)b->n * n_frames * feat_stride;
    int rc = ensure_staging(b, nfeat, 0);
    if (rc) return rc;
    hipStream_t st = b->e->stream;
    if ((rc = order_begin(b, st))) return rc;
    HIP_TRY(hipMemcpyAsync(b->d_feat, features, nfeat * sizeof(float), hipMemcpyHostToDevice, st));
    for (int f0 = 0; f0 < n_frames; f0 += b->max_chunk) {
        const int nf = n_frames - f0 < b->max_chunk ? n_frames - f0 : b->max_chunk;
        rc = launch_frames(b, st, b->d_feat + (size_t)f0 * feat_stride, feat_stride, (size_t)n_frames * feat_stride, nf);
        if (rc) return rc;
        if (cond_a) HIP_TRY(hipMemcpy2DAsync(cond_a + (size_t)f0 * LPCN_ROWS_A, (size_t)n_frames * LPCN_ROWS_A * 4, b->d_cond_a, (size_t)nf * LPCN_ROWS_A * 4,
                                             (size_t)nf * LPCN_ROWS_A * 4, b->n, hipMemcpyDeviceToHost, st));
        if (cond_b) HIP_TRY(hipMemcpy2DAsync(cond_b + (size_t)f0 * LPCN_ROWS_B, (size_t)n_frames * LPCN_ROWS_B * 4, b->d_cond_b, (size_t)nf * LPCN_ROWS_B * 4,
                                             (size_t)nf * LPCN_ROWS_B * 4, b->n, hipMemcpyDeviceToHost, st));
        if (lpc) HIP_TRY(hipMemcpy2DAsync(lpc + (size_t)f0 * LPCN_LPC_ORDER, (size_t)n_frames * LPCN_LPC_ORDER * 4, b->d_lpc, (size_t)nf * LPCN_LPC_ORDER * 4,
                                          (size_t)nf * LPCN_LPC_ORDER * 4, b->n, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    return 0;
}

// ---- one frame step with PER-STREAM arguments (a batched packet-loss concealment needs it: src/lpcnet_plc.c calls
// lpcnet_synthesize_impl(N = TRAINING_OFFSET, preload), lpcnet_synthesize_tail_impl(N, preload) and plain frames per
// stream, depending on which of its streams lost a packet).  mode[s]: 0 = leave the stream alone, 1 = frame network on
// features[s] + n_samples[s] samples (lpcnet_synthesize_impl, src/lpcnet.c:273-277), 2 = n_samples[s] samples from the
// products of the stream's most recent frame (lpcnet_synthesize_tail_impl, src/lpcnet.c:235-271); the first preload[s]
// samples of pcm[s] are imposed (teacher forcing).  Streams with equal (mode, n_samples, preload) form a group; a group is
// compacted (states gathered by a small kernel, features / PCM on the host side), runs through the ordinary kernels and is
// scattered back -- the hot kernels stay untouched; this path is meant for occasional use, not for throughput.
__global__ void lpcn_state_move_kernel(lpcn_stream_state *dst, const lpcn_stream_state *src, const int *map, int count, int scatter)
{
    const int words = (int)(sizeof(lpcn_stream_state) / 4);
    const int i = blockIdx.x;
    if (i >= count) return;
    const uint32_t *s = (const uint32_t *)(scatter ? &src[i] : &src[map[i]]);
    uint32_t *d = (uint32_t *)(scatter ? &dst[map[i]] : &dst[i]);
    for (int k = threadIdx.x; k < words; k += blockDim.x) d[k] = s[k];
}
__global__ void lpcn_rows_move_kernel(float *dst, const float *src, const int *map, int count, int width, int scatter)
{
    const int i = blockIdx.x;
    if (i >= count) return;
    const float *s = src + (size_t)(scatter ? i : map[i]) * width;
    float *d = dst + (size_t)(scatter ? map[i] : i) * width;
    for (int k = threadIdx.x; k < width; k += blockDim.x) d[k] = s[k];
}

extern "C" int lpcn_batch_dev_step_host(lpcn_batch_dev *b, const float *features, int feat_stride, short *pcm,
                                        const int *n_samples, const int *preload, const int *mode)
{
    if (!features || !pcm || !n_samples || !preload || !mode || feat_stride < LPCN_NB_FEAT) { snprintf(g_err, sizeof(g_err), "bad step arguments"); return LPCN_E_ARG; }
    for (int s = 0; s < b->n; ++s) {
        if (mode[s] < 0 || mode[s] > 2) { snprintf(g_err, sizeof(g_err), "stream %d: mode must be 0, 1 or 2", s); return LPCN_E_ARG; }
        if (mode[s] && (n_samples[s] < 1 || n_samples[s] > LPCN_FRAME_SIZE || preload[s] < 0 || preload[s] > n_samples[s])) {
            snprintf(g_err, sizeof(g_err), "stream %d: n_samples must be 1..160 and preload 0..n_samples", s); return LPCN_E_ARG;
        }
    }
    // a tail-only step continues the frame of the stream's last mode-1 step: the reference's lpcnet_synthesize_tail_impl uses
    // whatever the last run_frame_network left in the state, but this engine keeps frame products per stream only for
    // steps that went through this call -- a stream advanced by the ordinary synthesize / decode calls has none
    if (b->keep_ok.size() != (size_t)b->n) b->keep_ok.assign((size_t)b->n, 0);
    for (int s = 0; s < b->n; ++s)
        if (mode[s] == 2 && !b->keep_ok[s]) {
            snprintf(g_err, sizeof(g_err), "stream %d: a tail-only step (mode 2) needs a preceding frame step (mode 1) of lpcnet_batch_synthesize_step", s);
            return LPCN_E_ARG;
        }
    DeviceGuard guard(b->e->device);
    int rc = ensure_staging(b, (size_t)b->n * LPCN_NB_FEAT, (size_t)b->n * LPCN_FRAME_SIZE);
    if (rc) return rc;
    hipStream_t st = b->e->stream;
    if ((rc = order_begin(b, st))) return rc;
    if (b->S_auto && !b->tuned && (rc = autotune_streams_per_wg(b, st))) return rc;
#define ALX(ptr, bytes) if (!(ptr) && hipMalloc((void **)&(ptr), (bytes)) != hipSuccess) { snprintf(g_err, sizeof(g_err), "hipMalloc(%zu) failed", (size_t)(bytes)); return LPCN_E_HIP; }
    ALX(b->d_state_tmp, sizeof(lpcn_stream_state) * b->n);
    ALX(b->d_map, sizeof(int) * b->n);
    if (!b->d_keep_a) {
        ALX(b->d_keep_a, sizeof(float) * (size_t)b->n * LPCN_ROWS_A);
        ALX(b->d_keep_b, sizeof(float) * (size_t)b->n * LPCN_ROWS_B);
        ALX(b->d_keep_lpc, sizeof(float) * (size_t)b->n * LPCN_LPC_ORDER);
        HIP_TRY(hipMemsetAsync(b->d_keep_a, 0, sizeof(float) * (size_t)b->n * LPCN_ROWS_A, st));
        HIP_TRY(hipMemsetAsync(b->d_keep_b, 0, sizeof(float) * (size_t)b->n * LPCN_ROWS_B, st));
        HIP_TRY(hipMemsetAsync(b->d_keep_lpc, 0, sizeof(float) * (size_t)b->n * LPCN_LPC_ORDER, st));
    }
#undef ALX
    std::vector<char> done((size_t)b->n, 0);
    std::vector<int> map;
    std::vector<float> fc((size_t)b->n * LPCN_NB_FEAT);
    std::vector<short> pc((size_t)b->n * LPCN_FRAME_SIZE);
    const int keep_n = b->n, keep_len = b->frame_len;
    lpcn_stream_state *const keep_state = b->d_state;
    auto restore = [&]() { b->n = keep_n; b->frame_len = keep_len; b->d_state = keep_state; };
    for (int s0 = 0; s0 < keep_n; ++s0) {
        if (done[s0] || mode[s0] == 0) continue;
        map.clear();
        for (int s = s0; s < keep_n; ++s)
            if (!done[s] && mode[s] == mode[s0] && n_samples[s] == n_samples[s0] && preload[s] == preload[s0]) { map.push_back(s); done[s] = 1; }
        const int cnt = (int)map.size(), N = n_samples[s0], pre = preload[s0], md = mode[s0];
        for (int i = 0; i < cnt; ++i) {
            memcpy(&fc[(size_t)i * LPCN_NB_FEAT], features + (size_t)map[i] * feat_stride, sizeof(float) * LPCN_NB_FEAT);
            memcpy(&pc[(size_t)i * LPCN_FRAME_SIZE], pcm + (size_t)map[i] * LPCN_FRAME_SIZE, sizeof(short) * LPCN_FRAME_SIZE);
        }
        HIP_TRY(hipMemcpyAsync(b->d_map, map.data(), sizeof(int) * cnt, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(b->d_feat, fc.data(), sizeof(float) * (size_t)cnt * LPCN_NB_FEAT, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(b->d_pcm, pc.data(), sizeof(short) * (size_t)cnt * LPCN_FRAME_SIZE, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(lpcn_state_move_kernel, dim3(cnt), dim3(256), 0, st, b->d_state_tmp, (const lpcn_stream_state *)keep_state, (const int *)b->d_map, cnt, 0);
        b->n = cnt; b->d_state = b->d_state_tmp; b->frame_len = N;
        if (md == 1) {
            rc = launch_frames(b, st, b->d_feat, LPCN_NB_FEAT, (size_t)LPCN_NB_FEAT, 1);
            if (!rc) {      // remember the products per stream (a later tail-only step of the stream uses them)
                hipLaunchKernelGGL(lpcn_rows_move_kernel, dim3(cnt), dim3(256), 0, st, b->d_keep_a, (const float *)b->d_cond_a, (const int *)b->d_map, cnt, LPCN_ROWS_A, 1);
                hipLaunchKernelGGL(lpcn_rows_move_kernel, dim3(cnt), dim3(64), 0, st, b->d_keep_b, (const float *)b->d_cond_b, (const int *)b->d_map, cnt, LPCN_ROWS_B, 1);
                hipLaunchKernelGGL(lpcn_rows_move_kernel, dim3(cnt), dim3(64), 0, st, b->d_keep_lpc, (const float *)b->d_lpc, (const int *)b->d_map, cnt, LPCN_LPC_ORDER, 1);
            }
        } else {
            hipLaunchKernelGGL(lpcn_rows_move_kernel, dim3(cnt), dim3(256), 0, st, b->d_cond_a, (const float *)b->d_keep_a, (const int *)b->d_map, cnt, LPCN_ROWS_A, 0);
            hipLaunchKernelGGL(lpcn_rows_move_kernel, dim3(cnt), dim3(64), 0, st, b->d_cond_b, (const float *)b->d_keep_b, (const int *)b->d_map, cnt, LPCN_ROWS_B, 0);
            hipLaunchKernelGGL(lpcn_rows_move_kernel, dim3(cnt), dim3(64), 0, st, b->d_lpc, (const float *)b->d_keep_lpc, (const int *)b->d_map, cnt, LPCN_LPC_ORDER, 0);
        }
        if (!rc) rc = launch_sample(b, st, b->d_pcm, (size_t)LPCN_FRAME_SIZE, 1, pre, md == 1);
        restore();
        if (rc) return rc;
        hipLaunchKernelGGL(lpcn_state_move_kernel, dim3(cnt), dim3(256), 0, st, keep_state, (const lpcn_stream_state *)b->d_state_tmp, (const int *)b->d_map, cnt, 1);
        if (hipGetLastError() != hipSuccess) { snprintf(g_err, sizeof(g_err), "per-stream step: kernel launch failed"); return LPCN_E_HIP; }
        HIP_TRY(hipMemcpyAsync(pc.data(), b->d_pcm, sizeof(short) * (size_t)cnt * LPCN_FRAME_SIZE, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        // only now do the group's streams own frame products a later tail-only step may continue from (the flag used to be
        // set before the sample launch and the state scatter had succeeded)
        if (md == 1) for (int i = 0; i < cnt; ++i) b->keep_ok[(size_t)map[i]] = 1;
        for (int i = 0; i < cnt; ++i) memcpy(pcm + (size_t)map[i] * LPCN_FRAME_SIZE, &pc[(size_t)i * LPCN_FRAME_SIZE], sizeof(short) * (size_t)N);
    }
    return 0;
}

// debug trace (tests only): allocate / fetch the per-sample trace of workgroup 0, stream 0
extern "C" int lpcn_batch_dev_debug_trace(lpcn_batch_dev *b, int n_samples, float *host_out)
{
    DeviceGuard guard(b->e->device);
    if (host_out == nullptr) {
        { int rcw = wait_all(b); if (rcw) return rcw; }
        if (b->d_dbg) { (void)hipFree(b->d_dbg); b->d_dbg = nullptr; }
        if (n_samples > 0) {
            HIP_TRY(hipMalloc((void **)&b->d_dbg, sizeof(float) * (size_t)n_samples * LPCN_DBG_STRIDE));
            HIP_TRY(hipMemset(b->d_dbg, 0, sizeof(float) * (size_t)n_samples * LPCN_DBG_STRIDE));
        }
        return 0;
    }
    if (!b->d_dbg) { snprintf(g_err, sizeof(g_err), "trace not enabled"); return LPCN_E_ARG; }
    { int rcw = wait_all(b); if (rcw) return rcw; }
    HIP_TRY(hipMemcpy(host_out, b->d_dbg, sizeof(float) * (size_t)n_samples * LPCN_DBG_STRIDE, hipMemcpyDeviceToHost));
    return 0;
}

// per-phase shader-clock totals of workgroup 0 / wave 0 (out == NULL: enable + zero; else fetch 8 values)
extern "C" int lpcn_batch_dev_profile(lpcn_batch_dev *b, unsigned long long *out)
{
    DeviceGuard guard(b->e->device);
    if (!b->d_prof) HIP_TRY(hipMalloc((void **)&b->d_prof, 96 * sizeof(unsigned long long)));
    { int rcw = wait_all(b); if (rcw) return rcw; }
    if (!out) { HIP_TRY(hipMemset(b->d_prof, 0, 96 * sizeof(unsigned long long))); return 0; }
    HIP_TRY(hipMemcpy(out, b->d_prof, 96 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return 0;
}

// test seam: the device's 10^x (lpcnet_exp10.h) for host arrays
extern "C" int lpcn_debug_exp10(int device, const float *x, double *out, size_t n)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { snprintf(g_err, sizeof(g_err), "no such HIP device"); return LPCN_E_NODEVICE; }
    DeviceGuard guard(device);
    float *dx = nullptr;
    double *dy = nullptr;
    HIP_TRY(hipMalloc((void **)&dx, n * sizeof(float)));
    if (hipMalloc((void **)&dy, n * sizeof(double)) != hipSuccess) { (void)hipFree(dx); snprintf(g_err, sizeof(g_err), "hipMalloc failed"); return LPCN_E_HIP; }
    int rc = 0;
    if (hipMemcpy(dx, x, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) rc = LPCN_E_HIP;
    if (!rc) {
        hipLaunchKernelGGL(lpcn::exp10_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (const float *)dx, dy, n);
        if (hipGetLastError() != hipSuccess || hipMemcpy(out, dy, n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) rc = LPCN_E_HIP;
    }
    (void)hipFree(dx); (void)hipFree(dy);
    if (rc) snprintf(g_err, sizeof(g_err), "exp10 test kernel failed");
    return rc;
}

// test seam: the arithmetic identities PARITY rests on, evaluated with THIS library's compile flags and float mode.
//   * v_mfma_f32_4x4x1(A, B, C = -0.0): register k of lane j of a quad == v_mul_f32(A of lane k, B of lane j), bit for bit
//     (the GRU-A items of the float PARITY kernels form their products there, sample_kernel.hip.h: mac());
//   * each half of v_pk_mul_f32 / v_pk_add_f32 == v_mul_f32 / v_add_f32 (GRU-B's block loop, the items' sums).
// n lanes (a multiple of 64).  out_mfma / out_mul: [n][4] bit patterns (k = 0..3: A from lane 4*(i/4) + k, B from lane i);
// out_pk / out_sc: [n][4] = {pk_mul half 0, half 1, pk_add half 0, half 1} and the scalar instructions' results on the same operands
// (half 0: (a[i], b[i]), half 1: (a[i^1], b[i^1])).
__global__ void lpcn_arith_identity_kernel(const float *a, const float *b, uint32_t *out_mfma, uint32_t *out_mul, uint32_t *out_pk, uint32_t *out_sc)
{
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    f4 negz = {-0.f, -0.f, -0.f, -0.f};
    asm volatile("" : "+v"(negz));                         // (the same guard as the kernel's: the addend must reach the instruction as -0.0)
    const float av = a[i], bv = b[i];
    const f4 p = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, negz, 0, 0, 0);
    // (element-wise copies first: hipcc's __builtin_bit_cast of an ext-vector ELEMENT reads element 0 whatever the index)
    const float pe[4] = {p[0], p[1], p[2], p[3]};
    const size_t q = i & ~(size_t)3;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float ak = a[q + k], m;
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m) : "v"(ak), "v"(bv));
        out_mfma[i * 4 + k] = __float_as_uint(pe[k]);
        out_mul[i * 4 + k] = __float_as_uint(m);
    }
    const float a2 = a[i ^ 1], b2 = b[i ^ 1];
    f2 x = {av, a2}, y = {bv, b2}, pm, pa;
    asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(pm) : "v"(x), "v"(y));
    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(pa) : "v"(x), "v"(y));
    float m0, m1, s0, s1;
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m0) : "v"(av), "v"(bv));
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m1) : "v"(a2), "v"(b2));
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(s0) : "v"(av), "v"(bv));
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(s1) : "v"(a2), "v"(b2));
    const float pk0 = pm[0], pk1 = pm[1], pk2 = pa[0], pk3 = pa[1];
    out_pk[i * 4 + 0] = __float_as_uint(pk0); out_pk[i * 4 + 1] = __float_as_uint(pk1);
    out_pk[i * 4 + 2] = __float_as_uint(pk2); out_pk[i * 4 + 3] = __float_as_uint(pk3);
    out_sc[i * 4 + 0] = __float_as_uint(m0); out_sc[i * 4 + 1] = __float_as_uint(m1);
    out_sc[i * 4 + 2] = __float_as_uint(s0); out_sc[i * 4 + 3] = __float_as_uint(s1);
}

extern "C" int lpcn_debug_arith_identities(int device, const float *a, const float *b, uint32_t *out_mfma, uint32_t *out_mul,
                                           uint32_t *out_pk, uint32_t *out_sc, size_t n)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { snprintf(g_err, sizeof(g_err), "no such HIP device"); return LPCN_E_NODEVICE; }
    if (!n || n % 64) { snprintf(g_err, sizeof(g_err), "operand count must be a positive multiple of 64"); return LPCN_E_ARG; }
    DeviceGuard guard(device);
    float *d_in = nullptr;
    uint32_t *d_out = nullptr;
    HIP_TRY(hipMalloc((void **)&d_in, 2 * n * sizeof(float)));
    if (hipMalloc((void **)&d_out, 16 * n * sizeof(uint32_t)) != hipSuccess) { (void)hipFree(d_in); snprintf(g_err, sizeof(g_err), "hipMalloc failed"); return LPCN_E_HIP; }
    int rc = 0;
    if (hipMemcpy(d_in, a, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_in + n, b, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) rc = LPCN_E_HIP;
    if (!rc) {
        hipLaunchKernelGGL(lpcn_arith_identity_kernel, dim3((unsigned)(n / 64)), dim3(64), 0, 0, (const float *)d_in, (const float *)(d_in + n),
                           d_out, d_out + 4 * n, d_out + 8 * n, d_out + 12 * n);
        uint32_t *const dst[4] = {out_mfma, out_mul, out_pk, out_sc};
        if (hipGetLastError() != hipSuccess) rc = LPCN_E_HIP;
        for (int k = 0; k < 4 && !rc; ++k)
            if (hipMemcpy(dst[k], d_out + (size_t)k * 4 * n, 4 * n * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) rc = LPCN_E_HIP;
    }
    (void)hipFree(d_in); (void)hipFree(d_out);
    if (rc) snprintf(g_err, sizeof(g_err), "arithmetic identity test kernel failed");
    return rc;
}

// test seam: the state re-quantisation of the int8 kernels.  The reference computes (int)floor(.5 + t) with t = 127 x rounded to float and the
// sum in DOUBLE (src/vec.h:311-316: exact, 0.5 + t needs at most 31 bits); the kernels use ONE instruction, v_cvt_rpi_i32_f32 ("round to
// nearest, ties toward +infinity" = floor(t + 0.5) evaluated exactly), when LPCN_QUANT_RPI is set.  This sweep compares both on ALL 2^32 bit
// patterns: out[0] = mismatches among the finite t with |t| < 2^31, out[1] = mismatches inside the reachable range |t| <= 127.5 (|x| <= 1),
// out[2] = one mismatching bit pattern (if any).
__global__ void lpcn_quant_sweep_kernel(unsigned long long *out)
{
    const uint32_t base = (blockIdx.x * blockDim.x + threadIdx.x) * 256u;
    unsigned bad = 0, bad_in = 0;
    for (uint32_t k = 0; k < 256u; ++k) {
        const uint32_t u = base + k;
        const float t = __uint_as_float(u);
        if (!(fabsf(t) < 2147483648.f)) continue;            // NaN, infinities and |t| >= 2^31: the C conversion is undefined there
        const int want = (int)floor(.5 + (double)t);
        int got;
        asm volatile("v_cvt_rpi_i32_f32 %0, %1" : "=v"(got) : "v"(t));
        if (got != want) { ++bad; if (fabsf(t) <= 127.5f) ++bad_in; out[2] = u; }
    }
    if (bad) atomicAdd(&out[0], (unsigned long long)bad);
    if (bad_in) atomicAdd(&out[1], (unsigned long long)bad_in);
}

extern "C" int lpcn_debug_quant_sweep(int device, unsigned long long *out3)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { snprintf(g_err, sizeof(g_err), "no such HIP device"); return LPCN_E_NODEVICE; }
    DeviceGuard guard(device);
    unsigned long long *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, 3 * sizeof(unsigned long long)));
    int rc = 0;
    if (hipMemset(d, 0, 3 * sizeof(unsigned long long)) != hipSuccess) rc = LPCN_E_HIP;
    if (!rc) {
        hipLaunchKernelGGL(lpcn_quant_sweep_kernel, dim3(65536), dim3(256), 0, 0, d);      // 2^16 x 2^8 threads x 2^8 patterns
        if (hipGetLastError() != hipSuccess || hipMemcpy(out3, d, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) rc = LPCN_E_HIP;
    }
    (void)hipFree(d);
    if (rc) snprintf(g_err, sizeof(g_err), "quantisation sweep kernel failed");
    return rc;
}
