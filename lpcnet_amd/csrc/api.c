/* C host shell of the LPCNet HIP engine: the public entry points of include/lpcnet.h and
 * include/lpcnet_batch.h on top of the device runtime declared in lpcnet_engine.h.
 *
 * Single-stream API ownership model (SURVEY.md §7 hard part 7): the reference lets callers
 * allocate LPCNetState themselves (lpcnet_get_size + lpcnet_init, no de-init hook) and copy it
 * by value (PLC snapshot / rollback, src/lpcnet_plc.c:216-231).  LPCNetState therefore stays a
 * self-contained POD: it carries the complete per-stream state plus an integer handle of the
 * model; device memory belongs to a process-global registry (one engine + one 1-stream device
 * batch per distinct blob).  Every lpcnet_synthesize() call uploads the POD state, runs one frame
 * on the device and downloads state + PCM.
 */
#include <pthread.h>
#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "lpcnet.h"
#include "lpcnet_batch.h"
#include "lpcnet_engine.h"

#define LPCN_MAGIC 0x4C50434Eu   /* "LPCN" */

#define MAX_FEATURE_BUFFER_SIZE 4     /* src/lpcnet_private.h:26 */

struct LPCNetState {
    uint32_t magic;
    int32_t model_id;                 /* index into the registry, -1 = no model bound */
    lpcn_stream_state s;
    /* the frame products live in the state like in the reference (src/lpcnet_private.h:36-38), so that
     * run_frame_network and lpcnet_synthesize_tail_impl can be called separately (PLC, src/lpcnet_plc.c) */
    float gru_a_condition[LPCN_ROWS_A];
    float gru_b_condition[LPCN_ROWS_B];
    float feature_buffer[NB_FEATURES * MAX_FEATURE_BUFFER_SIZE];   /* run_frame_network_deferred queue */
    int32_t feature_buffer_fill;
};

_Static_assert(sizeof(struct LPCNetState) == LPCNET_HIP_STATE_BYTES, "include/lpcnet.h: LPCNET_HIP_STATE_BYTES is stale");

struct LPCNetDecState {
    LPCNetState lpcnet_state;         /* reference: src/lpcnet_private.h:50-53 */
    float vq_mem[LPCN_NB_BANDS];
};

struct LPCNetBatch {
    int n, device;
    lpcn_engine *engine;
    lpcn_batch_dev *dev;
    int cb_version;                   /* g_cb_version of the codebooks on the device (0 = none) */
};

static __thread char tl_err[512];
static void set_err(const char *msg) { snprintf(tl_err, sizeof(tl_err), "%s", msg); }
static void take_engine_err(void) { snprintf(tl_err, sizeof(tl_err), "%s", lpcn_last_error()); }
const char *lpcnet_hip_last_error(void) { return tl_err; }
const char *lpcnet_batch_last_error(void) { return tl_err; }

/* ---- VQ codebooks for the codec path (absent generated file ceps_codebooks.c) ---------------- */
static float *g_cb[4];
static int g_cb_version;              /* bumped by every lpcnet_hip_set_codebooks: batches re-upload lazily */
static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;

void lpcnet_hip_set_codebooks(const float *cb1, const float *cb2, const float *cb3, const float *cbd)
{
    const float *src[4] = {cb1, cb2, cb3, cbd};
    const size_t cnt[4] = {1024 * 17, 1024 * 17, 1024 * 17, 4096 * 18};
    pthread_mutex_lock(&g_lock);
    for (int i = 0; i < 4; i++) {
        free(g_cb[i]);
        g_cb[i] = (float *)malloc(cnt[i] * sizeof(float));
        memcpy(g_cb[i], src[i], cnt[i] * sizeof(float));
    }
    g_cb_version++;
    pthread_mutex_unlock(&g_lock);
}

/* packet -> 4 feature vectors.  Follows src/lpcnet_dec.c:81-155 and src/common.c:37-65:
 * 7+6+3+2+10+10+10+13+3 bits, MSB first. */
static unsigned get_bits(const unsigned char *buf, int *pos, int n)
{
    unsigned d = 0;
    for (; n > 0; n--, (*pos)++) d = (d << 1) | ((buf[*pos >> 3] >> (7 - (*pos & 7))) & 1u);
    return d;
}

static void band_interp(float *x, const float *left, const float *right, int mode)
{
    for (int i = 0; i < LPCN_NB_BANDS; i++)
        x[i] = mode == 0 ? .5f * (left[i] + right[i]) : (mode == 1 ? left[i] : right[i]);
}

static int packet_to_features(float feat[4][NB_TOTAL_FEATURES], float *vq_mem, const unsigned char *buf)
{
    if (!g_cb[0]) return -1;
    int pos = 0;
    const int c0_id = (int)get_bits(buf, &pos, 7), main_pitch = (int)get_bits(buf, &pos, 6);
    int modulation = (int)get_bits(buf, &pos, 3);
    const int corr_id = (int)get_bits(buf, &pos, 2);
    const int e0 = (int)get_bits(buf, &pos, 10), e1 = (int)get_bits(buf, &pos, 10), e2 = (int)get_bits(buf, &pos, 10);
    int vq_mid = (int)get_bits(buf, &pos, 13), interp_id = (int)get_bits(buf, &pos, 3);
    int voiced = 1;
    memset(feat, 0, sizeof(float) * 4 * NB_TOTAL_FEATURES);
    modulation -= 4;
    if (modulation == -4) { voiced = 0; modulation = 0; }
    const float frame_corr = voiced ? 0.3875f + .175f * corr_id : 0.0375f + .075f * corr_id;
    for (int sub = 0; sub < 4; sub++) {
        float p = (float)(pow(2.f, main_pitch / 21.) * 32);
        p *= 1.f + modulation / 16.f / 7.f * (2 * sub - 3);
        if (p < 33) p = 33;
        if (p > 255) p = 255;
        feat[sub][LPCN_NB_BANDS] = .02f * (p - 100.f);
        feat[sub][LPCN_NB_BANDS + 1] = frame_corr - .5f;
    }
    feat[3][0] = (c0_id - 64) / 4.f;
    for (int i = 0; i < LPCN_NB_BANDS - 1; i++)
        feat[3][i + 1] = g_cb[0][e0 * 17 + i] + g_cb[1][e1 * 17 + i] + g_cb[2][e2 * 17 + i];
    float sign = 1;
    if (vq_mid >= 4096) { vq_mid -= 4096; sign = -1; }
    for (int i = 0; i < LPCN_NB_BANDS; i++) feat[1][i] = sign * g_cb[3][vq_mid * LPCN_NB_BANDS + i];
    if ((vq_mid & 3) < 2)       for (int i = 0; i < LPCN_NB_BANDS; i++) feat[1][i] += .5f * (vq_mem[i] + feat[3][i]);
    else if ((vq_mid & 3) == 2) for (int i = 0; i < LPCN_NB_BANDS; i++) feat[1][i] += vq_mem[i];
    else                        for (int i = 0; i < LPCN_NB_BANDS; i++) feat[1][i] += feat[3][i];
    interp_id += (interp_id >= 7);
    band_interp(feat[0], vq_mem, feat[1], interp_id / 3);
    band_interp(feat[2], feat[1], feat[3], interp_id % 3);
    memcpy(vq_mem, feat[3], sizeof(float) * LPCN_NB_BANDS);
    return 0;
}

/* ---- model registry for the single-stream API ------------------------------------------------ */
typedef struct {
    int used;
    const unsigned char *blob; int len; uint64_t hash;
    lpcn_engine *engine;
    lpcn_batch_dev *dev;              /* 1 stream */
} registry_entry;
#define MAX_MODELS 16
static registry_entry g_reg[MAX_MODELS];

static uint64_t fnv1a(const unsigned char *p, int n)
{
    uint64_t h = 1469598103934665603ull;
    for (int i = 0; i < n; i++) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}

static int registry_bind(const unsigned char *blob, int len)
{
    const uint64_t h = fnv1a(blob, len);
    int slot = -1;
    for (int i = 0; i < MAX_MODELS; i++) {
        if (g_reg[i].used && g_reg[i].len == len && g_reg[i].hash == h) return i;
        if (!g_reg[i].used && slot < 0) slot = i;
    }
    if (slot < 0) { set_err("too many distinct models bound through lpcnet_load_model"); return -1; }
    lpcn_model_host m;
    if (lpcn_model_parse(&m, blob, len) != 0) { set_err("malformed or incomplete DNNw weight blob"); return -1; }
    lpcn_engine *e = NULL;
    lpcn_batch_dev *d = NULL;
    int rc = lpcn_engine_create(&e, 0, &m);
    lpcn_model_release(&m);
    if (rc) { take_engine_err(); return -1; }
    rc = lpcn_batch_dev_create(&d, e, 1, 1);
    if (rc) { take_engine_err(); lpcn_engine_destroy(e); return -1; }
    g_reg[slot].used = 1; g_reg[slot].blob = blob; g_reg[slot].len = len; g_reg[slot].hash = h;
    g_reg[slot].engine = e; g_reg[slot].dev = d;
    return slot;
}

void lpcnet_hip_shutdown(void)
{
    pthread_mutex_lock(&g_lock);
    for (int i = 0; i < MAX_MODELS; i++)
        if (g_reg[i].used) {
            lpcn_batch_dev_destroy(g_reg[i].dev);
            lpcn_engine_destroy(g_reg[i].engine);
            g_reg[i].used = 0;
        }
    pthread_mutex_unlock(&g_lock);
}

/* ---- single-stream API ------------------------------------------------------------------------ */
/* KISS99 seeding from the string "LPCNet" (src/lpcnet.c:176,181; src/kiss99.c:34-57) */
static void seed_rng(uint32_t c[4])
{
    static const unsigned char d[6] = {'L', 'P', 'C', 'N', 'e', 't'};
    c[0] = 362436069u ^ d[0]; c[1] = 521288629u ^ d[1]; c[2] = 123456789u ^ d[2]; c[3] = 380116160u ^ d[3];
    {   /* one generator step (src/kiss99.c:59-81) */
        uint32_t z = 36969u * (c[0] & 0xFFFF) + (c[0] >> 16), w = 18000u * (c[1] & 0xFFFF) + (c[1] >> 16);
        uint32_t j = c[2];
        j ^= j << 13; j ^= j >> 17; j ^= j << 5;
        c[0] = z; c[1] = w; c[2] = j; c[3] = 69069u * c[3] + 1234567u;
    }
    c[0] ^= d[4]; c[1] ^= d[5];
    if (c[0] == 0 || c[0] == 0x9068FFFFu) c[0]++;
    if (c[1] == 0 || c[1] == 0x464FFFFFu) c[1]++;
    if (c[2] == 0) c[2]++;
}

int lpcnet_get_size(void) { return (int)sizeof(LPCNetState); }

void lpcnet_reset(LPCNetState *st)
{
    memset(&st->s, 0, sizeof(*st) - offsetof(LPCNetState, s));      /* everything behind the model handle, src/lpcnet.c:177-179 */
    st->s.last_exc = 128;             /* lin2ulaw(0), src/lpcnet.c:180 */
    seed_rng(st->s.rng);
}

int lpcnet_init(LPCNetState *st)
{
    st->magic = LPCN_MAGIC;
    st->model_id = -1;
    lpcnet_reset(st);
    return 0;
}

LPCNetState *lpcnet_create(void)
{
    LPCNetState *st = (LPCNetState *)calloc(1, sizeof(*st));
    if (st) lpcnet_init(st);
    return st;
}

void lpcnet_destroy(LPCNetState *st) { free(st); }

int lpcnet_load_model(LPCNetState *st, const unsigned char *data, int len)
{
    if (!st || !data || len <= 0) { set_err("lpcnet_load_model: bad arguments"); return -1; }
    pthread_mutex_lock(&g_lock);
    const int id = registry_bind(data, len);
    pthread_mutex_unlock(&g_lock);
    if (id < 0) return -1;
    st->model_id = id;
    return 0;
}

static lpcn_batch_dev *bound_device(const LPCNetState *st, const char *who)
{
    if (st->magic != LPCN_MAGIC || st->model_id < 0 || st->model_id >= MAX_MODELS || !g_reg[st->model_id].used) {
        fprintf(stderr, "%s: no model bound to this state (call lpcnet_load_model first); "
                        "the HIP engine has no built-in model and no CPU fallback\n", who);
        abort();
    }
    return g_reg[st->model_id].dev;
}

static void device_failure(const char *who)
{
    fprintf(stderr, "%s: device failure: %s\n", who, lpcn_last_error());
    abort();
}

/* ---- the reference's internal entry points (src/lpcnet_private.h:125-132), which src/lpcnet_plc.c links to ---- */

/* src/lpcnet.c:226-233 */
void lpcnet_reset_signal(LPCNetState *st)
{
    st->s.deemph_mem = 0;
    st->s.last_exc = 128;             /* lin2ulaw(0.f) */
    memset(st->s.last_sig, 0, sizeof(st->s.last_sig));
    memset(st->s.gru_a, 0, sizeof(st->s.gru_a));
    memset(st->s.gru_b, 0, sizeof(st->s.gru_b));
}

/* src/lpcnet.c:82-120: one step of the 100 Hz network; products go to the caller's arrays */
void run_frame_network(LPCNetState *st, float *gru_a_condition, float *gru_b_condition, float *lpc, const float *features)
{
    lpcn_batch_dev *d = bound_device(st, "run_frame_network");
    float lpc_new[LPCN_LPC_ORDER];      /* `lpc` may point into st->s, which the state download overwrites */
    pthread_mutex_lock(&g_lock);
    int rc = lpcn_batch_dev_set_state(d, 0, &st->s);
    if (!rc) rc = lpcn_batch_dev_run_frames_host(d, features, NB_FEATURES, gru_a_condition, gru_b_condition, lpc_new, 1);
    if (!rc) rc = lpcn_batch_dev_get_state(d, 0, &st->s);
    pthread_mutex_unlock(&g_lock);
    if (rc) device_failure("run_frame_network");
    memcpy(lpc, lpc_new, sizeof(lpc_new));
}

/* src/lpcnet.c:122-132: queue a frame (at most kernel_size1 + kernel_size2 - 2 = 4, oldest dropped) */
void run_frame_network_deferred(LPCNetState *st, const float *features)
{
    const int max_buffer_size = MAX_FEATURE_BUFFER_SIZE;
    if (st->feature_buffer_fill == max_buffer_size)
        memmove(st->feature_buffer, &st->feature_buffer[NB_FEATURES], sizeof(float) * (size_t)(max_buffer_size - 1) * NB_FEATURES);
    else
        st->feature_buffer_fill++;
    memcpy(&st->feature_buffer[(st->feature_buffer_fill - 1) * NB_FEATURES], features, sizeof(float) * NB_FEATURES);
}

/* src/lpcnet.c:134-144: run the queued frames, products discarded */
void run_frame_network_flush(LPCNetState *st)
{
    for (int i = 0; i < st->feature_buffer_fill; i++) {
        float lpc[LPCN_LPC_ORDER], ga[LPCN_ROWS_A], gb[LPCN_ROWS_B];
        run_frame_network(st, ga, gb, lpc, &st->feature_buffer[i * NB_FEATURES]);
    }
    st->feature_buffer_fill = 0;
}

/* src/lpcnet.c:235-271: N samples from the products held in the state; the first `preload` samples of
 * `output` are imposed on the synthesis filter (teacher forcing) instead of being written */
void lpcnet_synthesize_tail_impl(LPCNetState *st, short *output, int N, int preload)
{
    if (N <= 0) return;
    lpcn_batch_dev *d = bound_device(st, "lpcnet_synthesize_tail_impl");
    if (N > LPCN_FRAME_SIZE || preload < 0 || preload > N) {
        fprintf(stderr, "lpcnet_synthesize_tail_impl: N=%d preload=%d outside 1..%d / 0..N\n", N, preload, LPCN_FRAME_SIZE);
        abort();
    }
    short frame[LPCN_FRAME_SIZE] = {0};
    memcpy(frame, output, sizeof(short) * (size_t)preload);
    pthread_mutex_lock(&g_lock);
    int rc = lpcn_batch_dev_set_state(d, 0, &st->s);
    if (!rc) rc = lpcn_batch_dev_set_frame_len(d, N);
    if (!rc) rc = lpcn_batch_dev_run_tail_host(d, st->gru_a_condition, st->gru_b_condition, st->s.lpc, frame, 1, preload);
    if (!rc) rc = lpcn_batch_dev_get_state(d, 0, &st->s);
    pthread_mutex_unlock(&g_lock);
    if (rc) device_failure("lpcnet_synthesize_tail_impl");
    /* live frames return the imposed samples unchanged; start-up frames are cleared entirely (src/lpcnet.c:239-243) */
    memcpy(output, frame, sizeof(short) * (size_t)N);
}

/* src/lpcnet.c:273-277 */
void lpcnet_synthesize_impl(LPCNetState *st, const float *features, short *output, int N, int preload)
{
    run_frame_network(st, st->gru_a_condition, st->gru_b_condition, st->s.lpc, features);
    lpcnet_synthesize_tail_impl(st, output, N, preload);
}

/* src/lpcnet.c:279-281.  (One fused device pass: frame kernels and sample kernel back to back; equals
 * lpcnet_synthesize_impl(..., 0) bit for bit, tests/test_gpu_parity.py.) */
void lpcnet_synthesize(LPCNetState *st, const float *features, short *output, int N)
{
    if (N <= 0) return;
    lpcn_batch_dev *d = bound_device(st, "lpcnet_synthesize");
    if (N > LPCN_FRAME_SIZE) {
        fprintf(stderr, "lpcnet_synthesize: N=%d > %d samples per call is not supported by the HIP engine\n", N, LPCN_FRAME_SIZE);
        abort();
    }
    pthread_mutex_lock(&g_lock);
    short frame[LPCN_FRAME_SIZE];
    int rc = lpcn_batch_dev_set_state(d, 0, &st->s);
    if (!rc) rc = lpcn_batch_dev_set_frame_len(d, N);
    if (!rc) rc = lpcn_batch_dev_run_host(d, features, NB_FEATURES, frame, 1, 0);
    if (!rc) rc = lpcn_batch_dev_get_state(d, 0, &st->s);
    pthread_mutex_unlock(&g_lock);
    if (rc) device_failure("lpcnet_synthesize");
    memcpy(output, frame, sizeof(short) * (size_t)N);
}

/* ---- decoder ----------------------------------------------------------------------------------- */
int lpcnet_decoder_get_size(void) { return (int)sizeof(LPCNetDecState); }

int lpcnet_decoder_init(LPCNetDecState *st)
{
    memset(st, 0, sizeof(*st));
    lpcnet_init(&st->lpcnet_state);
    return 0;
}

LPCNetDecState *lpcnet_decoder_create(void)
{
    LPCNetDecState *st = (LPCNetDecState *)malloc(sizeof(*st));
    if (st) lpcnet_decoder_init(st);
    return st;
}

void lpcnet_decoder_destroy(LPCNetDecState *st) { free(st); }

int lpcnet_decode(LPCNetDecState *st, const unsigned char *buf, short *pcm)
{
    float feat[4][NB_TOTAL_FEATURES];
    if (packet_to_features(feat, st->vq_mem, buf) != 0) {
        set_err("lpcnet_decode: no VQ codebooks installed (lpcnet_hip_set_codebooks)");
        return -1;
    }
    for (int k = 0; k < 4; k++)
        lpcnet_synthesize(&st->lpcnet_state, feat[k], &pcm[k * LPCN_FRAME_SIZE], LPCN_FRAME_SIZE);
    return 0;
}

/* ---- batch API ----------------------------------------------------------------------------------- */
LPCNetBatch *lpcnet_batch_create(int n_streams, int device)
{
    if (n_streams <= 0 || device < 0) { set_err("lpcnet_batch_create: bad arguments"); return NULL; }
    LPCNetBatch *b = (LPCNetBatch *)calloc(1, sizeof(*b));
    if (!b) return NULL;
    b->n = n_streams; b->device = device;
    return b;
}

void lpcnet_batch_destroy(LPCNetBatch *b)
{
    if (!b) return;
    if (b->dev) lpcn_batch_dev_destroy(b->dev);
    if (b->engine) lpcn_engine_destroy(b->engine);
    free(b);
}

int lpcnet_batch_streams(const LPCNetBatch *b) { return b->n; }

#define BATCH_CHUNK_FRAMES 100       /* frame products are staged per chunk: 100 frames = 1 s of audio */

int lpcnet_batch_load_model(LPCNetBatch *b, const unsigned char *data, int len)
{
    lpcn_model_host m;
    if (lpcn_model_parse(&m, data, len) != 0) { set_err("malformed or incomplete DNNw weight blob"); return -1; }
    lpcn_engine *e = NULL;
    lpcn_batch_dev *d = NULL;
    int rc = lpcn_engine_create(&e, b->device, &m);
    lpcn_model_release(&m);
    if (rc) { take_engine_err(); return -1; }
    rc = lpcn_batch_dev_create(&d, e, b->n, BATCH_CHUNK_FRAMES);
    if (rc) { take_engine_err(); lpcn_engine_destroy(e); return -1; }
    if (b->dev) lpcn_batch_dev_destroy(b->dev);
    if (b->engine) lpcn_engine_destroy(b->engine);
    b->engine = e; b->dev = d;
    return 0;
}

#define NEED_MODEL(b) do { if (!(b) || !(b)->dev) { set_err("batch has no model (lpcnet_batch_load_model)"); return LPCN_E_MODEL; } } while (0)
#define FWD(call) do { int rc_ = (call); if (rc_) take_engine_err(); return rc_; } while (0)

int lpcnet_batch_reset(LPCNetBatch *b, int first, int count)
{
    NEED_MODEL(b);
    FWD(lpcn_batch_dev_reset(b->dev, first, count));
}

int lpcnet_batch_synthesize(LPCNetBatch *b, const float *features, int feat_stride, short *pcm, int n_frames)
{
    NEED_MODEL(b);
    FWD(lpcn_batch_dev_run_host(b->dev, features, feat_stride, pcm, n_frames, 0));
}

int lpcnet_batch_synthesize_preload(LPCNetBatch *b, const float *features, int feat_stride, short *pcm, int n_frames, int preload)
{
    NEED_MODEL(b);
    FWD(lpcn_batch_dev_run_host(b->dev, features, feat_stride, pcm, n_frames, preload));
}

int lpcnet_batch_synthesize_device(LPCNetBatch *b, const float *d_features, int feat_stride, short *d_pcm, int n_frames, void *hip_stream)
{
    NEED_MODEL(b);
    FWD(lpcn_batch_dev_run(b->dev, d_features, feat_stride, d_pcm, n_frames, 0, hip_stream));
}

int lpcnet_batch_sync(LPCNetBatch *b) { NEED_MODEL(b); FWD(lpcn_batch_dev_sync(b->dev)); }

/* the device copy of the VQ codebooks follows lpcnet_hip_set_codebooks() */
static int batch_codebooks(LPCNetBatch *b)
{
    pthread_mutex_lock(&g_lock);
    int rc = 0;
    if (!g_cb[0]) { set_err("lpcnet_batch_decode: no VQ codebooks installed (lpcnet_hip_set_codebooks)"); rc = LPCN_E_MODEL; }
    else if (b->cb_version != g_cb_version) {
        rc = lpcn_engine_set_codebooks(b->engine, g_cb[0], g_cb[1], g_cb[2], g_cb[3]);
        if (rc) take_engine_err(); else b->cb_version = g_cb_version;
    }
    pthread_mutex_unlock(&g_lock);
    return rc;
}

/* packets [n][n_packets][8] (host) -> pcm [n][n_packets*640]; unpacking, VQ lookup and interpolation run on the device */
int lpcnet_batch_decode(LPCNetBatch *b, const unsigned char *packets, short *pcm, int n_packets)
{
    NEED_MODEL(b);
    if (n_packets <= 0) { set_err("lpcnet_batch_decode: bad arguments"); return LPCN_E_ARG; }
    int rc = batch_codebooks(b);
    if (rc) return rc;
    FWD(lpcn_batch_dev_decode_host(b->dev, packets, pcm, n_packets));
}

/* the same with device pointers, enqueued on the caller's stream (NULL = the batch's own) */
int lpcnet_batch_decode_device(LPCNetBatch *b, const unsigned char *d_packets, short *d_pcm, int n_packets, void *hip_stream)
{
    NEED_MODEL(b);
    if (n_packets <= 0) { set_err("lpcnet_batch_decode_device: bad arguments"); return LPCN_E_ARG; }
    int rc = batch_codebooks(b);
    if (rc) return rc;
    FWD(lpcn_batch_dev_decode(b->dev, d_packets, d_pcm, n_packets, hip_stream));
}

/* LPC_GAMMA of the model (a #define of the reference's generated nnet_data.h, not in the blob); default 1 */
int lpcnet_batch_set_lpc_gamma(LPCNetBatch *b, float gamma)
{
    NEED_MODEL(b);
    FWD(lpcn_engine_set_lpc_gamma(b->engine, gamma));
}

/* END2END models (LPC from the network's reflection coefficients); a #define of the reference, not in the blob */
int lpcnet_batch_set_end2end(LPCNetBatch *b, int on)
{
    NEED_MODEL(b);
    FWD(lpcn_engine_set_end2end(b->engine, on));
}

int lpcnet_batch_export_state(LPCNetBatch *b, int stream, LPCNetState *st)
{
    NEED_MODEL(b);
    if (st->magic != LPCN_MAGIC) { st->magic = LPCN_MAGIC; st->model_id = -1; }
    FWD(lpcn_batch_dev_get_state(b->dev, stream, &st->s));
}

int lpcnet_batch_import_state(LPCNetBatch *b, int stream, const LPCNetState *st)
{
    NEED_MODEL(b);
    FWD(lpcn_batch_dev_set_state(b->dev, stream, &st->s));
}

int lpcnet_batch_set_streams_per_workgroup(LPCNetBatch *b, int s) { NEED_MODEL(b); FWD(lpcn_batch_dev_set_streams_per_wg(b->dev, s)); }
int lpcnet_batch_get_streams_per_workgroup(const LPCNetBatch *b) { return b && b->dev ? lpcn_batch_dev_streams_per_wg(b->dev) : 0; }
int lpcnet_batch_enable_timing(LPCNetBatch *b, int on) { NEED_MODEL(b); FWD(lpcn_batch_dev_enable_timing(b->dev, on)); }
int lpcnet_batch_last_timing(LPCNetBatch *b, float *ms_s, float *ms_f) { NEED_MODEL(b); FWD(lpcn_batch_dev_last_timing(b->dev, ms_s, ms_f)); }

int lpcnet_batch_run_tail(LPCNetBatch *b, const float *cond_a, const float *cond_b, const float *lpc, short *pcm, int n_frames, int preload)
{
    NEED_MODEL(b);
    FWD(lpcn_batch_dev_run_tail_host(b->dev, cond_a, cond_b, lpc, pcm, n_frames, preload));
}

int lpcnet_batch_run_frames(LPCNetBatch *b, const float *features, int feat_stride, float *cond_a, float *cond_b, float *lpc, int n_frames)
{
    NEED_MODEL(b);
    FWD(lpcn_batch_dev_run_frames_host(b->dev, features, feat_stride, cond_a, cond_b, lpc, n_frames));
}

int lpcnet_batch_state_size(void) { return (int)sizeof(lpcn_stream_state); }
int lpcnet_batch_get_raw_state(LPCNetBatch *b, int stream, void *out) { NEED_MODEL(b); FWD(lpcn_batch_dev_get_state(b->dev, stream, (lpcn_stream_state *)out)); }
int lpcnet_batch_set_raw_state(LPCNetBatch *b, int stream, const void *in) { NEED_MODEL(b); FWD(lpcn_batch_dev_set_state(b->dev, stream, (const lpcn_stream_state *)in)); }
int lpcnet_batch_debug_trace(LPCNetBatch *b, int n_samples, float *host_out) { NEED_MODEL(b); FWD(lpcn_batch_dev_debug_trace(b->dev, n_samples, host_out)); }
int lpcnet_batch_profile(LPCNetBatch *b, unsigned long long *out) { NEED_MODEL(b); FWD(lpcn_batch_dev_profile(b->dev, out)); }

/* Host-only model check (no GPU needed): parses the blob with the loader's rules, builds the device
 * packings and verifies them.  info[0..5] = {is_int8, blocks GRU-A, blocks GRU-B, items per lane,
 * padded GRU-B blocks, selftest code}.  Returns 0 if the blob is loadable by the engine (float or
 * int8 flavour, see info[0]), -1 if it is malformed (lpcnet_load_model would fail). */
int lpcnet_hip_check_model(const unsigned char *data, int len, int *info)
{
    lpcn_model_host m;
    if (lpcn_model_parse(&m, data, len) != 0) { set_err("malformed or incomplete DNNw weight blob"); return -1; }
    int st = lpcn_model_selftest(&m);
    if (info) { info[0] = m.is_int8; info[1] = m.nb_a; info[2] = m.nb_b; info[3] = m.nw; info[4] = m.nb_b_padded; info[5] = st; }
    lpcn_model_release(&m);
    if (st) { set_err("internal error: device packing inconsistent with blob"); return -1; }
    return 0;
}
