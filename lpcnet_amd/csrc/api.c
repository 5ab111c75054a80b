/* C host shell of the LPCNet HIP engine: the public entry points of include/lpcnet.h and
 * include/lpcnet_batch.h on top of the device runtime declared in lpcnet_engine.h.
 *
 * Single-stream API ownership model (SURVEY.md §7 hard part 7): the reference lets callers
 * allocate LPCNetState themselves (lpcnet_get_size + lpcnet_init, no de-init hook) and copy it
 * by value (PLC snapshot / rollback, src/lpcnet_plc.c:216-231).  LPCNetState therefore stays a
 * self-contained POD: it carries the complete per-stream state plus an integer handle of the
 * model; device memory belongs to a process-global registry (one engine + one 1-stream device
 * batch per distinct blob, created lazily on the device chosen with lpcnet_hip_set_device()).  The registry
 * remembers the POD state it last downloaded: a lpcnet_synthesize() call whose state still equals that copy
 * (the normal frame-after-frame case) skips the upload; the new state and the PCM come back in one pinned
 * transfer with one synchronisation per frame.
 *
 * A state without an explicitly loaded model uses the process-default model, which mirrors the reference's
 * compiled-in model: lpcnet_hip_set_default_model(), else the file named by $LPCNET_HIP_MODEL, else
 * ./weights_blob.bin (the file name the reference's demo hard-codes, src/lpcnet_demo.c:113).  The same applies
 * to the decoder's VQ codebooks ($LPCNET_HIP_CODEBOOKS, ./ceps_codebooks.bin).
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

/* include/lpcnet_hip_state.h publishes this layout with the reference's member names (the unmodified src/lpcnet_plc.c is
 * compiled against it, integration/lpcnet_private_hip.h): every member the PLC touches must sit where that header says */
#define LPCNetState LPCNetState_published
#include "../../include/lpcnet_hip_state.h"
#undef LPCNetState
#define SAME_OFF(pub, mine) _Static_assert(offsetof(struct LPCNetState_published, pub) == offsetof(struct LPCNetState, mine), "include/lpcnet_hip_state.h is stale: " #pub)
_Static_assert(sizeof(struct LPCNetState_published) == sizeof(struct LPCNetState), "include/lpcnet_hip_state.h is stale");
SAME_OFF(nnet.gru_a_state, s.gru_a); SAME_OFF(nnet.gru_b_state, s.gru_b); SAME_OFF(nnet.feature_conv1_state, s.conv1_mem);
SAME_OFF(nnet.feature_conv2_state, s.conv2_mem); SAME_OFF(old_lpc, s.old_lpc); SAME_OFF(last_sig, s.last_sig);
SAME_OFF(deemph_mem, s.deemph_mem); SAME_OFF(last_exc, s.last_exc); SAME_OFF(frame_count, s.frame_count); SAME_OFF(hip_rng, s.rng);
SAME_OFF(lpc, s.lpc); SAME_OFF(gru_a_condition, gru_a_condition); SAME_OFF(gru_b_condition, gru_b_condition);
SAME_OFF(feature_buffer, feature_buffer); SAME_OFF(feature_buffer_fill, feature_buffer_fill);
#undef SAME_OFF

struct LPCNetDecState {
    LPCNetState lpcnet_state;         /* reference: src/lpcnet_private.h:50-53 */
    float vq_mem[LPCN_NB_BANDS];
};

/* A batch is one or more shards: contiguous blocks of streams, each on its own HIP device with its own engine,
 * device buffers and stream (streams are independent, so shards never communicate). */
#define LPCN_MAX_SHARDS 16
typedef struct {
    int first, count, device;
    lpcn_engine *engine;
    lpcn_batch_dev *dev;
    int cb_version;                   /* g_cb_version of the codebooks on the device (0 = none) */
} batch_shard;

struct LPCNetBatch {
    int n, n_shards;
    batch_shard sh[LPCN_MAX_SHARDS];
};

static __thread char tl_err[512];
static __thread int tl_status;        /* sticky: code of the first failed void entry point on this thread since lpcnet_hip_clear_error() */
static void set_err(const char *msg) { snprintf(tl_err, sizeof(tl_err), "%s", msg); }
static void take_engine_err(void) { snprintf(tl_err, sizeof(tl_err), "%s", lpcn_last_error()); }
const char *lpcnet_hip_last_error(void) { return tl_err; }
const char *lpcnet_batch_last_error(void) { return tl_err; }
int lpcnet_hip_status(void) { return tl_status; }
void lpcnet_hip_clear_error(void) { tl_status = 0; tl_err[0] = 0; }

#ifndef LPCN_SOURCE_HASH
#define LPCN_SOURCE_HASH "unknown"
#endif
#ifndef LPCN_DEVICE_SOURCE_HASH
#define LPCN_DEVICE_SOURCE_HASH "unknown"
#endif
/* identity of the sources this library was built from (lpcnet_amd/build.py bakes both in; the marker makes the string
 * findable in the file without loading it): src = every file under csrc/ + include/, dev = the device sources only */
static const char g_build_info[] = "LPCN_BUILD_INFO: src=" LPCN_SOURCE_HASH " dev=" LPCN_DEVICE_SOURCE_HASH;
const char *lpcnet_hip_build_info(void) { return g_build_info + sizeof("LPCN_BUILD_INFO: ") - 1; }

/* ---- VQ codebooks for the codec path (absent generated file ceps_codebooks.c) ---------------- */
static float *g_cb[4];
static int g_cb_version;              /* bumped by every lpcnet_hip_set_codebooks: batches re-upload lazily */
static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;      /* model registry */
/* The codebooks have a lock of their own, and readers share it: decoder threads unpack their packets concurrently (the
 * per-stream VQ memory lives in the caller's LPCNetDecState) and never meet the registry lock; only
 * lpcnet_hip_set_codebooks() -- and the one-time lookup of the default file -- excludes them. */
static pthread_rwlock_t g_cb_lock = PTHREAD_RWLOCK_INITIALIZER;
static int g_cb_default_tried;

static const size_t g_cb_count[4] = {1024 * 17, 1024 * 17, 1024 * 17, 4096 * 18};

static int install_codebooks_locked(const float *const src[4])
{
    float *fresh[4] = {NULL, NULL, NULL, NULL};
    for (int i = 0; i < 4; i++) {
        fresh[i] = (float *)malloc(g_cb_count[i] * sizeof(float));
        if (!fresh[i]) { for (int k = 0; k < i; k++) free(fresh[k]); return -1; }
        memcpy(fresh[i], src[i], g_cb_count[i] * sizeof(float));
    }
    for (int i = 0; i < 4; i++) { free(g_cb[i]); g_cb[i] = fresh[i]; }
    g_cb_version++;
    return 0;
}

void lpcnet_hip_set_codebooks(const float *cb1, const float *cb2, const float *cb3, const float *cbd)
{
    const float *const src[4] = {cb1, cb2, cb3, cbd};
    pthread_rwlock_wrlock(&g_cb_lock);
    if (install_codebooks_locked(src) != 0) set_err("lpcnet_hip_set_codebooks: out of memory");
    pthread_rwlock_unlock(&g_cb_lock);
}

/* whole file into malloc'ed memory (NULL if absent / unreadable / empty) */
static unsigned char *read_file(const char *path, long *len)
{
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    unsigned char *buf = NULL;
    long n = -1;
    if (fseek(f, 0, SEEK_END) == 0 && (n = ftell(f)) > 0 && fseek(f, 0, SEEK_SET) == 0) {
        buf = (unsigned char *)malloc((size_t)n);
        if (buf && fread(buf, 1, (size_t)n, f) != (size_t)n) { free(buf); buf = NULL; }
    }
    fclose(f);
    *len = n;
    return buf;
}

/* Process-default codebooks (the reference compiles ceps_codebooks.c in): $LPCNET_HIP_CODEBOOKS, else ./ceps_codebooks.bin;
 * raw little-endian float32: ceps_codebook1..3 [1024][17] each, then ceps_codebook_diff4 [4096][18]. */
static void default_codebooks_locked(void)
{
    if (g_cb[0] || g_cb_default_tried) return;
    g_cb_default_tried = 1;
    const char *path = getenv("LPCNET_HIP_CODEBOOKS");
    long len = 0;
    unsigned char *buf = read_file(path && *path ? path : "ceps_codebooks.bin", &len);
    if (!buf) return;
    const size_t want = (g_cb_count[0] + g_cb_count[1] + g_cb_count[2] + g_cb_count[3]) * sizeof(float);
    if ((size_t)len == want) {
        const float *f = (const float *)buf;
        const float *const src[4] = {f, f + g_cb_count[0], f + g_cb_count[0] + g_cb_count[1], f + g_cb_count[0] + g_cb_count[1] + g_cb_count[2]};
        (void)install_codebooks_locked(src);
    }
    free(buf);
}

/* the codebooks, read-locked (shared); the default file is looked for once, under the write lock */
static void codebooks_rdlock(void)
{
    pthread_rwlock_rdlock(&g_cb_lock);
    if (g_cb[0] || g_cb_default_tried) return;
    pthread_rwlock_unlock(&g_cb_lock);
    pthread_rwlock_wrlock(&g_cb_lock);
    default_codebooks_locked();
    pthread_rwlock_unlock(&g_cb_lock);
    pthread_rwlock_rdlock(&g_cb_lock);
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

/* (called with the codebooks read-locked, codebooks_rdlock(): another thread may replace them) */
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

/* ---- model registry for the single-stream API ------------------------------------------------
 * One slot per distinct weight blob (the registry keeps its own copy: dedupe compares the bytes, and a slot can
 * rebuild its device side at any time).  lpcnet_hip_shutdown() only releases the device resources, so a state bound
 * before the shutdown simply re-creates them at its next call.  Two limits (round 4): at most MAX_RESIDENT slots hold a
 * device side -- materialising one more releases the device side of the least recently used idle slot, whose handle and
 * blob copy stay valid (its states re-create it transparently at their next call, like after a shutdown) -- and MAX_MODELS
 * slots in all (the handle's low byte).  Only when that many DISTINCT blobs are bound does binding another one evict a slot
 * altogether (states are PODs that may be copied or dropped at will, so there is nothing to count references with); a
 * state still holding such a handle stops with a message. */
typedef struct {
    int used;
    unsigned gen;                     /* bumped when the slot is evicted: a state's handle carries the generation it was bound with */
    unsigned long last_use;           /* g_use_clock at the last bind / run (least recently used slot is evicted when the table is full) */
    int pins;                         /* callers between "slot chosen" and "run_lock taken" (g_lock protects it) */
    unsigned char *blob; int len; uint64_t hash;
    lpcn_engine *engine;
    lpcn_batch_dev *dev;              /* 1 stream */
    int device;
    int cache_valid;                  /* `cached` equals the device copy of the stream state */
    lpcn_stream_state cached;
    pthread_mutex_t run_lock;         /* one device round trip at a time per model */
    /* combining dispatcher of lpcnet_synthesize (see comb_synthesize) */
    lpcn_batch_dev *gdev;             /* COMB_MAX streams on the slot's engine, created when two callers first meet */
    pthread_mutex_t q_lock;
    pthread_cond_t q_cv;
    struct comb_req *q_head, *q_tail;
    int q_leader;                     /* a caller is dispatching (on the device or about to) */
    int fail_code, fail_count;        /* sticky: first failed pass on this model (lpcnet_hip_model_status) */
} registry_entry;
#define COMB_MAX 256
/* one waiting call of an entry point */
typedef struct comb_req {
    LPCNetState *st; const float *feat; short *pcm; int N;
    int kind, preload;                /* LPCN_GROUP_FRAME_SAMPLES / _TAIL / _FRAMES (lpcnet_engine.h) */
    int want_products;                /* FRAME_SAMPLES: the frame products must come back (lpcnet_synthesize_impl; a combined pass always returns them) */
    float *ga, *gb, *lpc;             /* the products: in (TAIL), out (FRAMES, FRAME_SAMPLES) */
    int done, rc;
    struct comb_req *next;
    char err[256];                    /* the leader's message for this caller (tl_err is thread-local) */
} comb_req;
#ifndef LPCN_MAX_MODELS
#define LPCN_MAX_MODELS 256           /* slots (the handle's low byte); tests build a library with a handful to reach the eviction paths */
#endif
#ifndef LPCN_MAX_RESIDENT
#define LPCN_MAX_RESIDENT 16          /* slots that hold a device side at the same time */
#endif
#define MAX_MODELS LPCN_MAX_MODELS
#define MAX_RESIDENT LPCN_MAX_RESIDENT
_Static_assert(MAX_MODELS >= 2 && MAX_MODELS <= 256 && MAX_RESIDENT >= 1, "the handle keeps the slot in its low byte");
static registry_entry g_reg[MAX_MODELS];
static int g_device = -1;             /* device of the single-stream API: lpcnet_hip_set_device(), $LPCNET_HIP_DEVICE, else 0 */
static int g_default_model = -1;      /* handle of the process-default model, -1 = not resolved yet */
static unsigned long g_use_clock;
/* A state's model handle: slot in the low byte, the slot's generation above it (a POD state cannot be told that its
 * model went away; a stale handle is detected instead and stops with a message). */
#define HANDLE(slot) ((int)(((g_reg[slot].gen & 0x7FFFFFu) << 8) | (unsigned)(slot)))
static int handle_slot(int h) { const int slot = h & 0xFF; return (h >= 0 && slot < MAX_MODELS && g_reg[slot].used && HANDLE(slot) == h) ? slot : -1; }

static uint64_t fnv1a(const unsigned char *p, int n)
{
    uint64_t h = 1469598103934665603ull;
    for (int i = 0; i < n; i++) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}

static int single_stream_device(void)
{
    if (g_device >= 0) return g_device;
    const char *e = getenv("LPCNET_HIP_DEVICE");
    return (e && *e) ? atoi(e) : 0;
}

int lpcnet_hip_set_device(int device)
{
    if (device < 0) { set_err("lpcnet_hip_set_device: bad device"); return -1; }
    pthread_mutex_lock(&g_lock);
    g_device = device;
    pthread_mutex_unlock(&g_lock);
    return 0;
}

/* (g_lock held) device side of a slot, created on first use, after lpcnet_hip_shutdown() and after it was released to make
 * room: the number of resident device sides is bounded, the least recently used idle one goes first */
static int registry_materialize(registry_entry *r)
{
    if (r->dev) return 0;
    {
        int resident = 0, victim = -1;
        for (int i = 0; i < MAX_MODELS; i++) if (g_reg[i].used && g_reg[i].dev) resident++;
        while (resident >= MAX_RESIDENT) {
            victim = -1;
            for (int i = 0; i < MAX_MODELS; i++)
                if (g_reg[i].used && g_reg[i].dev && &g_reg[i] != r && __atomic_load_n(&g_reg[i].pins, __ATOMIC_SEQ_CST) == 0 &&
                    (victim < 0 || g_reg[i].last_use < g_reg[victim].last_use)) victim = i;
            if (victim < 0 || pthread_mutex_trylock(&g_reg[victim].run_lock) != 0) break;       /* everything resident is in a call: go over the limit for now */
            registry_entry *v = &g_reg[victim];
            if (v->gdev) { lpcn_batch_dev_destroy(v->gdev); v->gdev = NULL; }
            lpcn_batch_dev_destroy(v->dev);
            lpcn_engine_destroy(v->engine);
            __atomic_store_n(&v->dev, (lpcn_batch_dev *)NULL, __ATOMIC_SEQ_CST);
            v->engine = NULL; v->cache_valid = 0;
            pthread_mutex_unlock(&v->run_lock);
            resident--;
        }
    }
    lpcn_model_host m;
    if (lpcn_model_parse(&m, r->blob, r->len) != 0) { set_err("malformed or incomplete DNNw weight blob"); return -1; }
    lpcn_engine *e = NULL;
    lpcn_batch_dev *d = NULL;
    const int device = single_stream_device();
    int rc = lpcn_engine_create(&e, device, &m);
    lpcn_model_release(&m);
    if (rc) { take_engine_err(); return -1; }
    rc = lpcn_batch_dev_create(&d, e, 1, 1);
    if (rc) { take_engine_err(); lpcn_engine_destroy(e); return -1; }
    r->engine = e; r->dev = d; r->device = device; r->cache_valid = 0;
    return 0;
}

/* (g_lock held) handle of this blob's slot, new or existing; -1 on a malformed blob / no device / every slot busy */
static int registry_bind(const unsigned char *blob, int len)
{
    const uint64_t h = fnv1a(blob, len);
    int slot = -1;
    for (int i = 0; i < MAX_MODELS; i++) {
        if (g_reg[i].used && g_reg[i].len == len && g_reg[i].hash == h && memcmp(g_reg[i].blob, blob, (size_t)len) == 0) {
            g_reg[i].last_use = ++g_use_clock;
            return registry_materialize(&g_reg[i]) == 0 ? HANDLE(i) : -1;
        }
        if (!g_reg[i].used && slot < 0) slot = i;
    }
    if (slot < 0) {                                          /* table full: evict the least recently used slot that nobody is running on */
        for (int i = 0; i < MAX_MODELS; i++)
            if (__atomic_load_n(&g_reg[i].pins, __ATOMIC_SEQ_CST) == 0 && i != handle_slot(g_default_model) && (slot < 0 || g_reg[i].last_use < g_reg[slot].last_use) &&
                pthread_mutex_trylock(&g_reg[i].run_lock) == 0) {
                if (slot >= 0) pthread_mutex_unlock(&g_reg[slot].run_lock);
                slot = i;
            }
        if (slot < 0) { set_err("too many distinct models in use at once through lpcnet_load_model"); return -1; }
        registry_entry *v = &g_reg[slot];
        if (v->gdev) { lpcn_batch_dev_destroy(v->gdev); v->gdev = NULL; }
        if (v->dev) { lpcn_batch_dev_destroy(v->dev); lpcn_engine_destroy(v->engine); }
        free(v->blob);
        const unsigned gen = v->gen + 1;
        pthread_mutex_unlock(&v->run_lock);
        pthread_mutex_destroy(&v->run_lock);
        pthread_mutex_destroy(&v->q_lock);
        pthread_cond_destroy(&v->q_cv);
        memset(v, 0, sizeof(*v));
        v->gen = gen;
    }
    registry_entry *r = &g_reg[slot];
    { const unsigned gen = r->gen; memset(r, 0, sizeof(*r)); r->gen = gen; }
    r->blob = (unsigned char *)malloc((size_t)len);
    if (!r->blob) { set_err("out of memory"); return -1; }
    memcpy(r->blob, blob, (size_t)len);
    r->len = len; r->hash = h;
    if (registry_materialize(r) != 0) { free(r->blob); r->blob = NULL; return -1; }
    pthread_mutex_init(&r->run_lock, NULL);
    pthread_mutex_init(&r->q_lock, NULL);
    pthread_cond_init(&r->q_cv, NULL);
    r->used = 1;
    r->last_use = ++g_use_clock;
    return HANDLE(slot);
}

/* Device resources of the single-stream API are released; bound states stay valid and re-create them on demand. */
void lpcnet_hip_shutdown(void)
{
    pthread_mutex_lock(&g_lock);
    for (int i = 0; i < MAX_MODELS; i++)
        if (g_reg[i].used && g_reg[i].dev) {
            pthread_mutex_lock(&g_reg[i].run_lock);
            if (g_reg[i].gdev) { lpcn_batch_dev_destroy(g_reg[i].gdev); g_reg[i].gdev = NULL; }
            lpcn_batch_dev_destroy(g_reg[i].dev);
            lpcn_engine_destroy(g_reg[i].engine);
            g_reg[i].dev = NULL; g_reg[i].engine = NULL; g_reg[i].cache_valid = 0;
            pthread_mutex_unlock(&g_reg[i].run_lock);
        }
    pthread_mutex_unlock(&g_lock);
}

int lpcnet_hip_set_default_model(const unsigned char *data, int len)
{
    if (!data || len <= 0) { set_err("lpcnet_hip_set_default_model: bad arguments"); return -1; }
    pthread_mutex_lock(&g_lock);
    const int id = registry_bind(data, len);
    if (id >= 0) g_default_model = id;
    pthread_mutex_unlock(&g_lock);
    return id >= 0 ? 0 : -1;
}

/* (g_lock held) the process-default model: explicit, else $LPCNET_HIP_MODEL, else ./weights_blob.bin */
static int default_model_locked(void)
{
    if (handle_slot(g_default_model) >= 0) return g_default_model;
    g_default_model = -1;
    /* (a failed lookup is not remembered: the file may appear, or the working directory change, before the next call) */
    const char *path = getenv("LPCNET_HIP_MODEL");
    const int from_cwd = !(path && *path);
    long len = 0;
    unsigned char *buf = read_file(from_cwd ? "weights_blob.bin" : path, &len);
    if (!buf) return -1;
    if (len > 0 && len < 0x7FFFFFFF) g_default_model = registry_bind(buf, (int)len);
    free(buf);
    if (g_default_model >= 0 && from_cwd && !getenv("LPCNET_HIP_QUIET"))
        fprintf(stderr, "lpcnet_hip: no model given (lpcnet_load_model / lpcnet_hip_set_default_model / $LPCNET_HIP_MODEL): using ./weights_blob.bin "
                        "from the current directory, like the reference's demo does (src/lpcnet_demo.c:113)\n");
    return g_default_model;
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
    st->model_id = -1;                /* resolved to the process-default model at first use (the reference binds its compiled-in model here) */
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
    st->magic = LPCN_MAGIC;
    st->model_id = id;
    return 0;
}

/* The reference has no public way to give an LPCNetDecState a model other than its compiled-in one
 * (src/lpcnet.c:284-290: lpcnet_decoder_init -> lpcnet_init); this is the explicit form for weight-file users. */
int lpcnet_hip_decoder_load_model(LPCNetDecState *st, const unsigned char *data, int len)
{
    if (!st) { set_err("lpcnet_hip_decoder_load_model: bad arguments"); return -1; }
    return lpcnet_load_model(&st->lpcnet_state, data, len);
}

/* ---- failure of an entry point that returns void in the reference (SURVEY.md §8b "Errors") -------------------------------
 * The reference's lpcnet_synthesize & co. cannot fail; here a device can.  Such a call does NOT abort the process: it
 * zero-fills what it was asked to produce (silence / zero frame products), leaves the caller's state as it was, and records
 * the failure in a sticky per-thread status (lpcnet_hip_status(), message in lpcnet_hip_last_error(), until
 * lpcnet_hip_clear_error()) and a sticky per-model status (lpcnet_hip_model_status()).  Under the combining dispatcher every
 * caller of a failed pass gets its own zero-filled output and its own status.  The first few failures of a process are also
 * written to stderr (LPCNET_HIP_QUIET=1: never); LPCNET_HIP_ABORT_ON_ERROR=1 restores the old stop-with-a-message. */
static __thread unsigned tl_fail_seq;                        /* failures recorded on this thread so far */
static void entry_failed(const char *who, int code, registry_entry *r, const char *msg)
{
    tl_fail_seq++;
    char buf[sizeof(tl_err)];
    snprintf(buf, sizeof(buf), "%s: %s", who, msg ? msg : "");
    snprintf(tl_err, sizeof(tl_err), "%s", buf);
    if (!tl_status) tl_status = code;
    (void)r;                                                 /* (the per-model status is recorded by dispatch() while the slot is pinned) */
    static int reported, env_read, env_fatal, env_quiet;     /* the two environment switches are read once (this is the per-frame path of a PLC that lost its device) */
    if (!__atomic_load_n(&env_read, __ATOMIC_ACQUIRE)) {
        const char *ab = getenv("LPCNET_HIP_ABORT_ON_ERROR");
        env_fatal = ab && *ab == '1';
        env_quiet = getenv("LPCNET_HIP_QUIET") != NULL;
        __atomic_store_n(&env_read, 1, __ATOMIC_RELEASE);
    }
    const int fatal = env_fatal;
    if (fatal || (!env_quiet && __atomic_fetch_add(&reported, 1, __ATOMIC_SEQ_CST) < 8))
        fprintf(stderr, "%s%s\n", tl_err, fatal ? "" : " -- output zero-filled, see lpcnet_hip_status()");
    if (fatal) abort();
}

/* (g_lock held) registry slot of this state's model -- its own handle, else the process-default model -- with a device
 * side; -1 with *code and tl_err set.  `msg` receives the explanation (the engine's message is thread-local too). */
static int resolve_slot_locked(LPCNetState *st, int *code)
{
    int h = (st->magic == LPCN_MAGIC) ? st->model_id : -1;
    if (h >= 0 && handle_slot(h) < 0) {
        snprintf(tl_err, sizeof(tl_err), "the model bound to this state was evicted (more than %d distinct models bound in this process); "
                                         "call lpcnet_load_model again", MAX_MODELS);
        *code = LPCN_E_MODEL;
        return -1;
    }
    if (h < 0) {
        h = default_model_locked();
        if (h >= 0 && st->magic == LPCN_MAGIC) st->model_id = h;
    }
    const int id = handle_slot(h);
    if (id < 0 || registry_materialize(&g_reg[id]) != 0) {
        char why[256];
        snprintf(why, sizeof(why), "%.255s", (id >= 0 && tl_err[0]) ? tl_err : "none found");
        snprintf(tl_err, sizeof(tl_err), "no model bound to this state and no default model (lpcnet_load_model, lpcnet_hip_set_default_model, "
                                         "$LPCNET_HIP_MODEL or ./weights_blob.bin): %s; the HIP engine has no built-in model and no CPU fallback", why);
        *code = id >= 0 ? LPCN_E_HIP : LPCN_E_MODEL;
        return -1;
    }
    return id;
}

/* The registry slot this state runs on, PINNED (neither evicted nor recycled until unpinned) but not locked: resolves the
 * handle / the default model and makes sure the slot has a device side.  Only g_lock is taken.  NULL: *code, tl_err. */
static registry_entry *pin_entry(LPCNetState *st, int *code)
{
    pthread_mutex_lock(&g_lock);
    const int id = resolve_slot_locked(st, code);
    if (id < 0) { pthread_mutex_unlock(&g_lock); return NULL; }
    registry_entry *r = &g_reg[id];
    r->last_use = ++g_use_clock;
    __atomic_add_fetch(&r->pins, 1, __ATOMIC_SEQ_CST);
    pthread_mutex_unlock(&g_lock);
    return r;
}
static void unpin_entry(registry_entry *r) { __atomic_sub_fetch(&r->pins, 1, __ATOMIC_SEQ_CST); }

/* A PINNED slot, locked for one device round trip (unlock with release_entry).  The lock is taken BY SLOT: the caller's
 * queue, model and device side are this slot's whatever the state's handle or the default model have become meanwhile
 * (ADVICE r4).  lpcnet_hip_shutdown() may have released the device side while we waited: it is re-created (lock order
 * g_lock -> run_lock everywhere).  0, or a negative code with tl_err set. */
static int acquire_slot(registry_entry *r)
{
    for (;;) {
        pthread_mutex_lock(&g_lock);
        const int rc = registry_materialize(r);
        pthread_mutex_unlock(&g_lock);
        if (rc) return LPCN_E_HIP;
        pthread_mutex_lock(&r->run_lock);
        if (__atomic_load_n(&r->dev, __ATOMIC_SEQ_CST) != NULL) return 0;
        pthread_mutex_unlock(&r->run_lock);
    }
}
static void release_entry(registry_entry *r) { pthread_mutex_unlock(&r->run_lock); }

/* upload the caller's POD state unless it is the copy the device already holds */
static int push_state(registry_entry *r, const lpcn_stream_state *s)
{
    if (r->cache_valid && memcmp(&r->cached, s, sizeof(*s)) == 0) return 0;
    r->cache_valid = 0;
    return lpcn_batch_dev_set_state(r->dev, 0, s);
}
static int pull_state(registry_entry *r, lpcn_stream_state *s)
{
    int rc = lpcn_batch_dev_get_state(r->dev, 0, s);
    if (!rc) { r->cached = *s; r->cache_valid = 1; }
    return rc;
}

/* sticky status of the model a state is bound to (0, or the code of its first failed pass since the last clear) */
int lpcnet_hip_model_status(const LPCNetState *st, int clear)
{
    if (!st || st->magic != LPCN_MAGIC) return LPCN_E_ARG;
    pthread_mutex_lock(&g_lock);
    const int slot = handle_slot(st->model_id >= 0 ? st->model_id : g_default_model);
    int code = slot < 0 ? LPCN_E_MODEL : __atomic_load_n(&g_reg[slot].fail_code, __ATOMIC_SEQ_CST);
    if (slot >= 0 && clear) __atomic_store_n(&g_reg[slot].fail_code, 0, __ATOMIC_SEQ_CST);
    pthread_mutex_unlock(&g_lock);
    return code;
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

/* ---- combining dispatcher (round 4: lpcnet_synthesize; round 5: every entry point that touches the device) ---------------------------
 * The reference is re-entrant per state: a server with one thread per stream scales over its cores.  Here every state of a model
 * shares one device, so concurrent callers are COMBINED instead of serialised: a call queues itself on its model's slot; if no pass
 * is being dispatched it becomes the leader, takes every queued call of its own shape -- same entry point, N and preload -- up to
 * COMB_MAX, and runs them as ONE multi-stream pass (each caller's POD state up, kernels, states and results down, one
 * synchronisation); calls that arrive meanwhile queue up and form the next pass, led by one of them -- group commit, no timer: a lone
 * caller never waits for company, and under load a pass carries as many streams as arrived during the previous one.  Results are
 * bit-identical to running the calls one by one (streams are independent).  The PLC-facing entry points (run_frame_network,
 * lpcnet_synthesize_impl, lpcnet_synthesize_tail_impl: src/lpcnet_private.h:125-132, what a threaded PLC server calls through
 * src/lpcnet_plc.c:216-239,378-421) go through the same queue as lpcnet_synthesize. */

/* (run_lock held) the 100 Hz network for one state, serially; nothing is written unless it succeeded */
static int frame_network_locked(registry_entry *r, LPCNetState *st, float *gru_a_condition, float *gru_b_condition, float *lpc, const float *features)
{
    float lpc_new[LPCN_LPC_ORDER];      /* `lpc` may point into st->s, which the state download overwrites */
    float ga[LPCN_ROWS_A], gb[LPCN_ROWS_B];
    lpcn_stream_state snew;
    int rc = push_state(r, &st->s);
    r->cache_valid = 0;
    if (!rc) rc = lpcn_batch_dev_run_frames_host(r->dev, features, NB_FEATURES, ga, gb, lpc_new, 1);
    if (!rc) rc = pull_state(r, &snew);
    if (rc) { r->cache_valid = 0; return rc; }
    st->s = snew;
    memcpy(gru_a_condition, ga, sizeof(ga));
    memcpy(gru_b_condition, gb, sizeof(gb));
    memcpy(lpc, lpc_new, sizeof(lpc_new));
    return 0;
}

/* (run_lock held) N samples (any N: consecutive pieces of <= 160) from the products held in the state, serially */
static int tail_locked(registry_entry *r, LPCNetState *st, short *output, int N, int preload)
{
    lpcn_stream_state snew;
    int rc = push_state(r, &st->s);
    r->cache_valid = 0;
    short one[LPCN_FRAME_SIZE];                              /* (the caller's buffer is written only when every piece has succeeded; N <= 160 -- every caller in
                                                              * the reference -- needs no heap) */
    short *tmp = N <= LPCN_FRAME_SIZE ? one : (short *)malloc(sizeof(short) * (size_t)N);
    if (!tmp) { set_err("out of memory"); return LPCN_E_HIP; }
    for (int done = 0; !rc && done < N; done += LPCN_FRAME_SIZE) {
        const int n = N - done < LPCN_FRAME_SIZE ? N - done : LPCN_FRAME_SIZE;
        const int pre = preload - done < 0 ? 0 : (preload - done > n ? n : preload - done);
        short frame[LPCN_FRAME_SIZE] = {0};
        memcpy(frame, output + done, sizeof(short) * (size_t)pre);
        rc = lpcn_batch_dev_set_frame_len(r->dev, n);
        if (!rc) rc = lpcn_batch_dev_run_tail_host(r->dev, st->gru_a_condition, st->gru_b_condition, st->s.lpc, frame, 1, pre);
        /* live frames return the imposed samples unchanged; start-up frames are cleared entirely (src/lpcnet.c:239-243) */
        if (!rc) memcpy(tmp + done, frame, sizeof(short) * (size_t)n);
    }
    if (!rc) rc = pull_state(r, &snew);
    if (rc) { r->cache_valid = 0; if (tmp != one) free(tmp); return rc; }
    st->s = snew;
    memcpy(output, tmp, sizeof(short) * (size_t)N);
    if (tmp != one) free(tmp);
    return 0;
}

/* One device pass for `k` queued calls of one shape on one model (run_lock held by the caller through acquire_slot). */
static int comb_run(registry_entry *r, comb_req **grp, int k)
{
    comb_req *q = grp[0];
    if (k == 1) {                                            /* nobody to share with */
        if (q->kind == LPCN_GROUP_FRAME_SAMPLES && q->preload == 0 && !q->want_products) {
            /* lpcnet_synthesize alone: the single-stream fast path (state upload skipped when the device copy is current) */
            const int fresh = !(r->cache_valid && memcmp(&r->cached, &q->st->s, sizeof(q->st->s)) == 0);
            r->cache_valid = 0;
            lpcn_stream_state snew;
            int rc = lpcn_batch_dev_set_frame_len(r->dev, q->N);
            if (!rc) rc = lpcn_batch_dev_run_single(r->dev, fresh ? &q->st->s : NULL, q->feat, q->pcm, &snew);
            if (!rc) { q->st->s = snew; r->cached = snew; r->cache_valid = 1; }
            return rc;
        }
        if (q->kind == LPCN_GROUP_FRAMES) return frame_network_locked(r, q->st, q->ga, q->gb, q->lpc, q->feat);
        if (q->kind == LPCN_GROUP_TAIL) return tail_locked(r, q->st, q->pcm, q->N, q->preload);
        /* lpcnet_synthesize_impl alone: the reference's two steps (src/lpcnet.c:273-277); atomic for the caller -- the state is put back
         * if the second step fails */
        const LPCNetState keep = *q->st;
        int rc = frame_network_locked(r, q->st, q->ga, q->gb, q->lpc, q->feat);
        if (!rc) rc = tail_locked(r, q->st, q->pcm, q->N, q->preload);
        if (rc) *q->st = keep;
        return rc;
    }
    if (!r->gdev) {
        int rc = lpcn_batch_dev_create(&r->gdev, r->engine, COMB_MAX, 1);
        if (rc) return rc;
    }
    const lpcn_stream_state *sin[COMB_MAX];
    lpcn_stream_state *sout[COMB_MAX];
    const float *ft[COMB_MAX];
    short *pc[COMB_MAX];
    float *ga[COMB_MAX], *gb[COMB_MAX], *lp[COMB_MAX];
    for (int i = 0; i < k; i++) {
        sin[i] = &grp[i]->st->s; sout[i] = &grp[i]->st->s; ft[i] = grp[i]->feat; pc[i] = grp[i]->pcm;
        /* a plain lpcnet_synthesize never refreshes the frame products kept in the state -- not on the lone fast path, and therefore not here either
         * (whether it did used to depend on whether another thread happened to call at the same time) */
        const int takes = grp[i]->kind != LPCN_GROUP_FRAME_SAMPLES || grp[i]->want_products;
        ga[i] = takes ? grp[i]->ga : NULL; gb[i] = takes ? grp[i]->gb : NULL; lp[i] = grp[i]->lpc;
    }
    r->cache_valid = 0;                                      /* (the one-stream batch's device copy is not what these states continue from) */
    return lpcn_batch_dev_run_group(r->gdev, k, q->kind, q->N, q->preload, sin, ft, pc, sout, ga, gb, lp);      /* (writes only after the pass has succeeded) */
}

/* dispatcher statistics (process-wide, relaxed atomics): calls served, device passes that served them, the largest pass */
static unsigned long long g_disp_calls, g_disp_passes, g_disp_max;
int lpcnet_hip_dispatch_stats(unsigned long long *out3, int reset)
{
    if (out3) {
        out3[0] = __atomic_load_n(&g_disp_calls, __ATOMIC_RELAXED);
        out3[1] = __atomic_load_n(&g_disp_passes, __ATOMIC_RELAXED);
        out3[2] = __atomic_load_n(&g_disp_max, __ATOMIC_RELAXED);
    }
    if (reset) {
        __atomic_store_n(&g_disp_calls, 0ull, __ATOMIC_RELAXED);
        __atomic_store_n(&g_disp_passes, 0ull, __ATOMIC_RELAXED);
        __atomic_store_n(&g_disp_max, 0ull, __ATOMIC_RELAXED);
    }
    return 0;
}

/* Queue one call on its model's slot and wait until a pass has served it (possibly leading that pass).  Returns 0 or a negative
 * code (tl_err set; *rout = the slot, if one was resolved); nothing is written to the caller's memory on failure. */
static int dispatch(comb_req *me, registry_entry **rout)
{
    /* the slot of this state's model, pinned while the call is queued (a pinned slot is neither evicted nor recycled) */
    int code = 0;
    registry_entry *r = pin_entry(me->st, &code);            /* (g_lock only: a pass in flight on this model must not keep other callers from queueing) */
    *rout = r;
    if (!r) return code;
    pthread_mutex_lock(&r->q_lock);
    if (r->q_tail) r->q_tail->next = me; else r->q_head = me;
    r->q_tail = me;
    while (!me->done) {
        if (r->q_leader) { pthread_cond_wait(&r->q_cv, &r->q_lock); continue; }
        /* lead one pass: my own call and everything queued with the same shape */
        r->q_leader = 1;
        comb_req *grp[COMB_MAX];
        int k = 0;
        grp[k++] = me;
        comb_req **pp = &r->q_head, *last = NULL;
        while (*pp) {
            comb_req *q = *pp;
            if (q == me || (q->kind == me->kind && q->N == me->N && q->preload == me->preload && q->N <= LPCN_FRAME_SIZE && k < COMB_MAX && q->st != me->st)) {
                if (q != me) grp[k++] = q;
                *pp = q->next;                               /* unlink */
            } else { last = q; pp = &q->next; }
        }
        r->q_tail = last;
        pthread_mutex_unlock(&r->q_lock);
        __atomic_fetch_add(&g_disp_calls, (unsigned long long)k, __ATOMIC_RELAXED);
        __atomic_fetch_add(&g_disp_passes, 1ull, __ATOMIC_RELAXED);
        {
            unsigned long long mx = __atomic_load_n(&g_disp_max, __ATOMIC_RELAXED);
            while ((unsigned long long)k > mx && !__atomic_compare_exchange_n(&g_disp_max, &mx, (unsigned long long)k, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) { }
        }
        int rc = acquire_slot(r);                            /* run_lock of THIS slot; re-creates the device side if it was released meanwhile */
        if (!rc) {
            rc = comb_run(r, grp, k);
            if (rc) { r->cache_valid = 0; take_engine_err(); }
            release_entry(r);
        }
        pthread_mutex_lock(&r->q_lock);
        for (int i = 0; i < k; i++) {
            grp[i]->rc = rc;
            if (rc) snprintf(grp[i]->err, sizeof(grp[i]->err), "%.255s", tl_err);      /* (the message is thread-local: hand every caller its copy) */
            grp[i]->done = 1;
        }
        r->q_leader = 0;
        pthread_cond_broadcast(&r->q_cv);
    }
    pthread_mutex_unlock(&r->q_lock);
    if (me->rc) {                                            /* the model's sticky status, recorded while the slot is still pinned (behind the unpin
                                                              * the slot may already belong to another model) */
        int zero = 0;
        __atomic_compare_exchange_n(&r->fail_code, &zero, me->rc, 0, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
        __atomic_add_fetch(&r->fail_count, 1, __ATOMIC_SEQ_CST);
    }
    unpin_entry(r);
    *rout = NULL;                                            /* (not to be touched behind the unpin) */
    if (me->rc) snprintf(tl_err, sizeof(tl_err), "%s", me->err);
    return me->rc;
}

/* src/lpcnet.c:82-120: one step of the 100 Hz network; products go to the caller's arrays (zeros on failure) */
void run_frame_network(LPCNetState *st, float *gru_a_condition, float *gru_b_condition, float *lpc, const float *features)
{
    comb_req me = {st, features, NULL, LPCN_FRAME_SIZE, LPCN_GROUP_FRAMES, 0, 0, gru_a_condition, gru_b_condition, lpc, 0, 0, NULL, {0}};
    registry_entry *r = NULL;
    const int rc = dispatch(&me, &r);
    if (rc) {
        char why[sizeof(tl_err)];
        snprintf(why, sizeof(why), "%s", tl_err);
        memset(gru_a_condition, 0, sizeof(float) * LPCN_ROWS_A);
        memset(gru_b_condition, 0, sizeof(float) * LPCN_ROWS_B);
        memset(lpc, 0, sizeof(float) * LPCN_LPC_ORDER);
        entry_failed("run_frame_network", rc, r, why);
    }
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
 * `output` are imposed on the synthesis filter (teacher forcing) instead of being written.  Any N: the
 * sample loop does not care where a call ends, so N > 160 runs as consecutive pieces of <= 160 samples (never combined).
 * On failure the samples behind the imposed ones are zero-filled and the state is left as it was. */
void lpcnet_synthesize_tail_impl(LPCNetState *st, short *output, int N, int preload)
{
    if (N <= 0) return;
    if (preload < 0 || preload > N) {
        char why[96];
        snprintf(why, sizeof(why), "preload=%d outside 0..N=%d", preload, N);
        memset(output, 0, sizeof(short) * (size_t)N);
        entry_failed("lpcnet_synthesize_tail_impl", LPCN_E_ARG, NULL, why);
        return;
    }
    comb_req me = {st, NULL, output, N, LPCN_GROUP_TAIL, preload, 0, st->gru_a_condition, st->gru_b_condition, st->s.lpc, 0, 0, NULL, {0}};
    registry_entry *r = NULL;
    const int rc = dispatch(&me, &r);
    if (rc) {
        char why[sizeof(tl_err)];
        snprintf(why, sizeof(why), "%s", tl_err);
        memset(output + preload, 0, sizeof(short) * (size_t)(N - preload));
        entry_failed("lpcnet_synthesize_tail_impl", rc, r, why);
    }
}

/* src/lpcnet.c:273-277: the frame network, then N samples; the frame products stay in the state for a later tail call */
void lpcnet_synthesize_impl(LPCNetState *st, const float *features, short *output, int N, int preload)
{
    if (N > LPCN_FRAME_SIZE || N <= 0 || preload < 0 || preload > N) {      /* the reference's two steps, each with its own checks */
        const LPCNetState keep = *st;                        /* atomic for the caller like the one-pass form: a failed second step puts the state back */
        const unsigned seq0 = tl_fail_seq;
        run_frame_network(st, st->gru_a_condition, st->gru_b_condition, st->s.lpc, features);
        if (N > 0 && tl_fail_seq == seq0) lpcnet_synthesize_tail_impl(st, output, N, preload);
        else if (N > 0) memset(output + (preload > 0 && preload <= N ? preload : 0), 0, sizeof(short) * (size_t)(N - (preload > 0 && preload <= N ? preload : 0)));
        if (tl_fail_seq != seq0) *st = keep;
        return;
    }
    comb_req me = {st, features, output, N, LPCN_GROUP_FRAME_SAMPLES, preload, 1, st->gru_a_condition, st->gru_b_condition, st->s.lpc, 0, 0, NULL, {0}};
    registry_entry *r = NULL;
    const int rc = dispatch(&me, &r);
    if (rc) {
        char why[sizeof(tl_err)];
        snprintf(why, sizeof(why), "%s", tl_err);
        memset(output + preload, 0, sizeof(short) * (size_t)(N - preload));
        entry_failed("lpcnet_synthesize_impl", rc, r, why);
    }
}

/* src/lpcnet.c:279-281.  N <= 160 (every caller in the reference): one fused device pass -- frame kernels and sample
 * kernel back to back, one synchronisation; bit-identical to lpcnet_synthesize_impl(..., 0) (tests/test_gpu_parity.py).
 * N > 160: the reference runs the frame network once and then N samples; so does the two-step form.
 * One difference a caller of the INTERNAL entry points could observe: a lone lpcnet_synthesize call does not refresh the frame
 * products kept in the state (gru_a_condition / gru_b_condition: 4.8 KB more to download per frame on the single-stream fast path);
 * a lpcnet_synthesize_tail_impl that continues a frame must follow lpcnet_synthesize_impl, as it does in src/lpcnet_plc.c.
 * Returns 0 or a negative code (tl_err set; *rout = the slot, if one was resolved); nothing is written on failure. */
static int synthesize_rc(LPCNetState *st, const float *features, short *output, int N, registry_entry **rout)
{
    comb_req me = {st, features, output, N, LPCN_GROUP_FRAME_SAMPLES, 0, 0, st->gru_a_condition, st->gru_b_condition, st->s.lpc, 0, 0, NULL, {0}};
    return dispatch(&me, rout);
}

void lpcnet_synthesize(LPCNetState *st, const float *features, short *output, int N)
{
    if (N <= 0) return;
    if (N > LPCN_FRAME_SIZE) { lpcnet_synthesize_impl(st, features, output, N, 0); return; }
    registry_entry *r = NULL;
    const int rc = synthesize_rc(st, features, output, N, &r);
    if (rc) {
        char why[sizeof(tl_err)];
        snprintf(why, sizeof(why), "%s", tl_err);
        memset(output, 0, sizeof(short) * (size_t)N);
        entry_failed("lpcnet_synthesize", rc, r, why);
    }
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

/* src/lpcnet.c:310-319.  -1 (the reference always returns 0) when no codebooks are installed or a frame's device pass
 * failed -- the packet's remaining samples are zero-filled in that case. */
int lpcnet_decode(LPCNetDecState *st, const unsigned char *buf, short *pcm)
{
    float feat[4][NB_TOTAL_FEATURES];
    codebooks_rdlock();                                      /* shared: decoder threads unpack concurrently, the VQ memory is the caller's */
    const int rc = packet_to_features(feat, st->vq_mem, buf);
    pthread_rwlock_unlock(&g_cb_lock);
    if (rc != 0) {
        set_err("lpcnet_decode: no VQ codebooks installed (lpcnet_hip_set_codebooks, $LPCNET_HIP_CODEBOOKS or ./ceps_codebooks.bin)");
        return -1;
    }
    for (int k = 0; k < 4; k++) {
        registry_entry *r = NULL;
        const int rs = synthesize_rc(&st->lpcnet_state, feat[k], &pcm[k * LPCN_FRAME_SIZE], LPCN_FRAME_SIZE, &r);
        if (rs) {
            char why[sizeof(tl_err)];
            snprintf(why, sizeof(why), "%s", tl_err);
            memset(&pcm[k * LPCN_FRAME_SIZE], 0, sizeof(short) * (size_t)(4 - k) * LPCN_FRAME_SIZE);
            entry_failed("lpcnet_decode", rs, r, why);
            return -1;
        }
    }
    return 0;
}

/* ---- batch API ----------------------------------------------------------------------------------- */
/* contiguous block partition: shard k of K gets n/K streams, the first n%K shards one more */
static void shard_partition(int n, int k, int K, int *first, int *count)
{
    const int base = n / K, extra = n % K;
    *count = base + (k < extra ? 1 : 0);
    *first = k * base + (k < extra ? k : extra);
}

LPCNetBatch *lpcnet_batch_create_sharded(int n_streams, const int *devices, int n_devices)
{
    if (n_streams <= 0 || !devices || n_devices <= 0 || n_devices > LPCN_MAX_SHARDS) { set_err("lpcnet_batch_create_sharded: bad arguments"); return NULL; }
    for (int k = 0; k < n_devices; k++) if (devices[k] < 0) { set_err("lpcnet_batch_create_sharded: bad device"); return NULL; }
    LPCNetBatch *b = (LPCNetBatch *)calloc(1, sizeof(*b));
    if (!b) return NULL;
    b->n = n_streams;
    for (int k = 0; k < n_devices; k++) {
        int first, count;
        shard_partition(n_streams, k, n_devices, &first, &count);
        if (count == 0) continue;          /* more devices than streams */
        batch_shard *s = &b->sh[b->n_shards++];
        s->first = first; s->count = count; s->device = devices[k];
    }
    return b;
}

LPCNetBatch *lpcnet_batch_create(int n_streams, int device)
{
    if (n_streams <= 0 || device < 0) { set_err("lpcnet_batch_create: bad arguments"); return NULL; }
    return lpcnet_batch_create_sharded(n_streams, &device, 1);
}

void lpcnet_batch_destroy(LPCNetBatch *b)
{
    if (!b) return;
    for (int k = 0; k < b->n_shards; k++) {
        if (b->sh[k].dev) lpcn_batch_dev_destroy(b->sh[k].dev);
        if (b->sh[k].engine) lpcn_engine_destroy(b->sh[k].engine);
    }
    free(b);
}

int lpcnet_batch_streams(const LPCNetBatch *b) { return b->n; }
int lpcnet_batch_shards(const LPCNetBatch *b) { return b ? b->n_shards : 0; }

int lpcnet_batch_shard_info(const LPCNetBatch *b, int shard, int *first, int *count, int *device)
{
    if (!b || shard < 0 || shard >= b->n_shards) { set_err("shard index"); return LPCN_E_ARG; }
    if (first) *first = b->sh[shard].first;
    if (count) *count = b->sh[shard].count;
    if (device) *device = b->sh[shard].device;
    return 0;
}

#define BATCH_CHUNK_FRAMES 100       /* frame products are staged per chunk: 100 frames = 1 s of audio */

int lpcnet_batch_load_model(LPCNetBatch *b, const unsigned char *data, int len)
{
    lpcn_model_host m;
    if (!b) { set_err("lpcnet_batch_load_model: bad arguments"); return -1; }
    if (lpcn_model_parse(&m, data, len) != 0) { set_err("malformed or incomplete DNNw weight blob"); return -1; }
    lpcn_engine *e[LPCN_MAX_SHARDS] = {0};
    lpcn_batch_dev *d[LPCN_MAX_SHARDS] = {0};
    int rc = 0;
    for (int k = 0; k < b->n_shards && !rc; k++) {          /* the model is replicated on every shard's device */
        rc = lpcn_engine_create(&e[k], b->sh[k].device, &m);
        if (!rc) rc = lpcn_batch_dev_create(&d[k], e[k], b->sh[k].count, BATCH_CHUNK_FRAMES);
    }
    lpcn_model_release(&m);
    if (rc) {
        take_engine_err();
        for (int k = 0; k < b->n_shards; k++) { if (d[k]) lpcn_batch_dev_destroy(d[k]); if (e[k]) lpcn_engine_destroy(e[k]); }
        return -1;
    }
    for (int k = 0; k < b->n_shards; k++) {
        if (b->sh[k].dev) lpcn_batch_dev_destroy(b->sh[k].dev);
        if (b->sh[k].engine) lpcn_engine_destroy(b->sh[k].engine);
        b->sh[k].engine = e[k]; b->sh[k].dev = d[k]; b->sh[k].cb_version = 0;
    }
    return 0;
}

#define NEED_MODEL(b) do { if (!(b) || !(b)->n_shards || !(b)->sh[0].dev) { set_err("batch has no model (lpcnet_batch_load_model)"); return LPCN_E_MODEL; } } while (0)
#define FWD(call) do { int rc_ = (call); if (rc_) take_engine_err(); return rc_; } while (0)
#define NEED_ONE_SHARD(b, what) do { if ((b)->n_shards != 1) { set_err(what ": device pointers belong to one device; use the _shard form on a sharded batch"); return LPCN_E_ARG; } } while (0)

/* ---- one host thread per shard: the shards of a host-pointer call run concurrently, each on its own
 * device and stream; with a single shard the work runs on the calling thread. */
typedef struct {
    LPCNetBatch *b; int shard;
    int kind;                                     /* 0 synthesize(preload), 1 decode */
    const float *features; int feat_stride; short *pcm; int n_frames, preload;
    const unsigned char *packets; int n_packets;
    int rc; char err[256];
} shard_job;

static void *shard_worker(void *arg)
{
    shard_job *j = (shard_job *)arg;
    batch_shard *s = &j->b->sh[j->shard];
    if (j->kind == 0)
        j->rc = lpcn_batch_dev_run_host(s->dev, j->features + (size_t)s->first * j->n_frames * j->feat_stride, j->feat_stride,
                                        j->pcm + (size_t)s->first * j->n_frames * LPCN_FRAME_SIZE, j->n_frames, j->preload);
    else
        j->rc = lpcn_batch_dev_decode_host(s->dev, j->packets + (size_t)s->first * j->n_packets * 8,
                                           j->pcm + (size_t)s->first * j->n_packets * 4 * LPCN_FRAME_SIZE, j->n_packets);
    if (j->rc) snprintf(j->err, sizeof(j->err), "%s", lpcn_last_error());      /* (the engine's message is thread-local) */
    return NULL;
}

static int run_shards(LPCNetBatch *b, const shard_job *proto)
{
    shard_job job[LPCN_MAX_SHARDS];
    pthread_t th[LPCN_MAX_SHARDS];
    int started[LPCN_MAX_SHARDS] = {0};
    for (int k = 0; k < b->n_shards; k++) { job[k] = *proto; job[k].b = b; job[k].shard = k; job[k].rc = 0; job[k].err[0] = 0; }
    for (int k = 1; k < b->n_shards; k++) started[k] = pthread_create(&th[k], NULL, shard_worker, &job[k]) == 0;
    shard_worker(&job[0]);
    for (int k = 1; k < b->n_shards; k++) { if (started[k]) pthread_join(th[k], NULL); else shard_worker(&job[k]); }
    for (int k = 0; k < b->n_shards; k++) if (job[k].rc) { set_err(job[k].err); return job[k].rc; }
    return 0;
}

int lpcnet_batch_reset(LPCNetBatch *b, int first, int count)
{
    NEED_MODEL(b);
    if (first < 0 || count < 0 || first + count > b->n) { set_err("reset range"); return LPCN_E_ARG; }
    for (int k = 0; k < b->n_shards; k++) {
        const batch_shard *s = &b->sh[k];
        const int lo = first > s->first ? first : s->first;
        const int hi = first + count < s->first + s->count ? first + count : s->first + s->count;
        if (hi <= lo) continue;
        int rc = lpcn_batch_dev_reset(s->dev, lo - s->first, hi - lo);
        if (rc) { take_engine_err(); return rc; }
    }
    return 0;
}

int lpcnet_batch_synthesize_preload(LPCNetBatch *b, const float *features, int feat_stride, short *pcm, int n_frames, int preload)
{
    NEED_MODEL(b);
    shard_job j;
    memset(&j, 0, sizeof(j));
    j.kind = 0; j.features = features; j.feat_stride = feat_stride; j.pcm = pcm; j.n_frames = n_frames; j.preload = preload;
    return run_shards(b, &j);
}

/* one frame step with per-stream arguments (see include/lpcnet_batch.h); shards run one after the other */
int lpcnet_batch_synthesize_step(LPCNetBatch *b, const float *features, int feat_stride, short *pcm,
                                 const int *n_samples, const int *preload, const int *mode)
{
    NEED_MODEL(b);
    if (!features || !pcm || !n_samples || !preload || !mode) { set_err("lpcnet_batch_synthesize_step: bad arguments"); return LPCN_E_ARG; }
    for (int k = 0; k < b->n_shards; k++) {
        const batch_shard *s = &b->sh[k];
        int rc = lpcn_batch_dev_step_host(s->dev, features + (size_t)s->first * feat_stride, feat_stride, pcm + (size_t)s->first * LPCN_FRAME_SIZE,
                                          n_samples + s->first, preload + s->first, mode + s->first);
        if (rc) { take_engine_err(); return rc; }
    }
    return 0;
}

int lpcnet_batch_synthesize(LPCNetBatch *b, const float *features, int feat_stride, short *pcm, int n_frames)
{
    return lpcnet_batch_synthesize_preload(b, features, feat_stride, pcm, n_frames, 0);
}

int lpcnet_batch_synthesize_device_shard(LPCNetBatch *b, int shard, const float *d_features, int feat_stride, short *d_pcm,
                                         int n_frames, void *hip_stream)
{
    NEED_MODEL(b);
    if (shard < 0 || shard >= b->n_shards) { set_err("shard index"); return LPCN_E_ARG; }
    FWD(lpcn_batch_dev_run(b->sh[shard].dev, d_features, feat_stride, d_pcm, n_frames, 0, hip_stream));
}

int lpcnet_batch_synthesize_device(LPCNetBatch *b, const float *d_features, int feat_stride, short *d_pcm, int n_frames, void *hip_stream)
{
    NEED_MODEL(b);
    NEED_ONE_SHARD(b, "lpcnet_batch_synthesize_device");
    return lpcnet_batch_synthesize_device_shard(b, 0, d_features, feat_stride, d_pcm, n_frames, hip_stream);
}

int lpcnet_batch_sync(LPCNetBatch *b)
{
    NEED_MODEL(b);
    for (int k = 0; k < b->n_shards; k++) { int rc = lpcn_batch_dev_sync(b->sh[k].dev); if (rc) { take_engine_err(); return rc; } }
    return 0;
}

/* the device copies of the VQ codebooks follow lpcnet_hip_set_codebooks() */
static int batch_codebooks(LPCNetBatch *b)
{
    codebooks_rdlock();
    int rc = 0;
    if (!g_cb[0]) { set_err("lpcnet_batch_decode: no VQ codebooks installed (lpcnet_hip_set_codebooks)"); rc = LPCN_E_MODEL; }
    for (int k = 0; !rc && k < b->n_shards; k++)
        if (b->sh[k].cb_version != g_cb_version) {
            rc = lpcn_engine_set_codebooks(b->sh[k].engine, g_cb[0], g_cb[1], g_cb[2], g_cb[3]);
            if (rc) take_engine_err(); else b->sh[k].cb_version = g_cb_version;
        }
    pthread_rwlock_unlock(&g_cb_lock);
    return rc;
}

/* packets [n][n_packets][8] (host) -> pcm [n][n_packets*640]; unpacking, VQ lookup and interpolation run on the device */
int lpcnet_batch_decode(LPCNetBatch *b, const unsigned char *packets, short *pcm, int n_packets)
{
    NEED_MODEL(b);
    if (n_packets <= 0) { set_err("lpcnet_batch_decode: bad arguments"); return LPCN_E_ARG; }
    int rc = batch_codebooks(b);
    if (rc) return rc;
    shard_job j;
    memset(&j, 0, sizeof(j));
    j.kind = 1; j.packets = packets; j.pcm = pcm; j.n_packets = n_packets;
    return run_shards(b, &j);
}

/* the same with device pointers, enqueued on the caller's stream (NULL = the batch's own) */
int lpcnet_batch_decode_device(LPCNetBatch *b, const unsigned char *d_packets, short *d_pcm, int n_packets, void *hip_stream)
{
    NEED_MODEL(b);
    NEED_ONE_SHARD(b, "lpcnet_batch_decode_device");
    if (n_packets <= 0) { set_err("lpcnet_batch_decode_device: bad arguments"); return LPCN_E_ARG; }
    int rc = batch_codebooks(b);
    if (rc) return rc;
    FWD(lpcn_batch_dev_decode(b->sh[0].dev, d_packets, d_pcm, n_packets, hip_stream));
}

#define EACH_SHARD(call) do { for (int k_ = 0; k_ < b->n_shards; k_++) { batch_shard *s = &b->sh[k_]; int rc_ = (call); if (rc_) { take_engine_err(); return rc_; } } return 0; } while (0)

/* LPC_GAMMA of the model (a #define of the reference's generated nnet_data.h, not in the blob); default 1 */
int lpcnet_batch_set_lpc_gamma(LPCNetBatch *b, float gamma) { NEED_MODEL(b); EACH_SHARD(lpcn_engine_set_lpc_gamma(s->engine, gamma)); }
/* END2END models (LPC from the network's reflection coefficients); a #define of the reference, not in the blob */
int lpcnet_batch_set_end2end(LPCNetBatch *b, int on) { NEED_MODEL(b); EACH_SHARD(lpcn_engine_set_end2end(s->engine, on)); }

/* arithmetic flavour: 0 = PARITY (default, bit-identical to the reference's generic-C build), 1 = FAST (what the
 * reference's own SIMD builds do: fused multiply-add / int32 block accumulation; validated teacher-forced) */
int lpcnet_batch_set_fast(LPCNetBatch *b, int on)
{
    NEED_MODEL(b);
    for (int k = 0; k < b->n_shards; k++) {
        int rc = lpcn_engine_set_fast(b->sh[k].engine, on);
        if (!rc) rc = lpcn_batch_dev_retune(b->sh[k].dev);
        if (rc) { take_engine_err(); return rc; }
    }
    return 0;
}

static batch_shard *shard_of(LPCNetBatch *b, int stream)
{
    for (int k = 0; k < b->n_shards; k++)
        if (stream >= b->sh[k].first && stream < b->sh[k].first + b->sh[k].count) return &b->sh[k];
    return NULL;
}
#define SHARD_OF(sv, b, stream) batch_shard *sv = shard_of(b, stream); do { if (!sv) { set_err("stream index"); return LPCN_E_ARG; } } while (0)

int lpcnet_batch_export_state(LPCNetBatch *b, int stream, LPCNetState *st)
{
    NEED_MODEL(b);
    SHARD_OF(s, b, stream);
    if (st->magic != LPCN_MAGIC) { st->magic = LPCN_MAGIC; st->model_id = -1; }
    FWD(lpcn_batch_dev_get_state(s->dev, stream - s->first, &st->s));
}

int lpcnet_batch_import_state(LPCNetBatch *b, int stream, const LPCNetState *st)
{
    NEED_MODEL(b);
    SHARD_OF(s, b, stream);
    FWD(lpcn_batch_dev_set_state(s->dev, stream - s->first, &st->s));
}

int lpcnet_batch_set_streams_per_workgroup(LPCNetBatch *b, int spw) { NEED_MODEL(b); EACH_SHARD(lpcn_batch_dev_set_streams_per_wg(s->dev, spw)); }
int lpcnet_batch_tune(LPCNetBatch *b) { NEED_MODEL(b); EACH_SHARD(lpcn_batch_dev_tune(s->dev)); }
int lpcnet_batch_get_streams_per_workgroup(const LPCNetBatch *b) { return b && b->n_shards && b->sh[0].dev ? lpcn_batch_dev_streams_per_wg(b->sh[0].dev) : 0; }
int lpcnet_batch_enable_timing(LPCNetBatch *b, int on) { NEED_MODEL(b); EACH_SHARD(lpcn_batch_dev_enable_timing(s->dev, on)); }
/* kernel milliseconds of the most recent run: the slowest shard */
int lpcnet_batch_last_timing(LPCNetBatch *b, float *ms_s, float *ms_f)
{
    NEED_MODEL(b);
    float bs = 0.f, bf = 0.f;
    for (int k = 0; k < b->n_shards; k++) {
        float a = 0.f, c = 0.f;
        lpcn_batch_dev_last_timing(b->sh[k].dev, &a, &c);
        if (a > bs) bs = a;
        if (c > bf) bf = c;
    }
    if (ms_s) *ms_s = bs;
    if (ms_f) *ms_f = bf;
    return 0;
}

int lpcnet_batch_run_tail(LPCNetBatch *b, const float *cond_a, const float *cond_b, const float *lpc, short *pcm, int n_frames, int preload)
{
    NEED_MODEL(b);
    for (int k = 0; k < b->n_shards; k++) {
        const batch_shard *s = &b->sh[k];
        const size_t o = (size_t)s->first * n_frames;
        int rc = lpcn_batch_dev_run_tail_host(s->dev, cond_a + o * LPCN_ROWS_A, cond_b + o * LPCN_ROWS_B, lpc + o * LPCN_LPC_ORDER,
                                              pcm + o * LPCN_FRAME_SIZE, n_frames, preload);
        if (rc) { take_engine_err(); return rc; }
    }
    return 0;
}

int lpcnet_batch_run_frames(LPCNetBatch *b, const float *features, int feat_stride, float *cond_a, float *cond_b, float *lpc, int n_frames)
{
    NEED_MODEL(b);
    for (int k = 0; k < b->n_shards; k++) {
        const batch_shard *s = &b->sh[k];
        const size_t o = (size_t)s->first * n_frames;
        int rc = lpcn_batch_dev_run_frames_host(s->dev, features + o * feat_stride, feat_stride, cond_a ? cond_a + o * LPCN_ROWS_A : NULL,
                                                cond_b ? cond_b + o * LPCN_ROWS_B : NULL, lpc ? lpc + o * LPCN_LPC_ORDER : NULL, n_frames);
        if (rc) { take_engine_err(); return rc; }
    }
    return 0;
}

int lpcnet_batch_state_size(void) { return (int)sizeof(lpcn_stream_state); }
int lpcnet_batch_get_raw_state(LPCNetBatch *b, int stream, void *out)
{
    NEED_MODEL(b);
    SHARD_OF(s, b, stream);
    FWD(lpcn_batch_dev_get_state(s->dev, stream - s->first, (lpcn_stream_state *)out));
}
int lpcnet_batch_set_raw_state(LPCNetBatch *b, int stream, const void *in)
{
    NEED_MODEL(b);
    SHARD_OF(s, b, stream);
    FWD(lpcn_batch_dev_set_state(s->dev, stream - s->first, (const lpcn_stream_state *)in));
}
int lpcnet_batch_debug_trace(LPCNetBatch *b, int n_samples, float *host_out) { NEED_MODEL(b); FWD(lpcn_batch_dev_debug_trace(b->sh[0].dev, n_samples, host_out)); }
int lpcnet_batch_profile(LPCNetBatch *b, unsigned long long *out) { NEED_MODEL(b); FWD(lpcn_batch_dev_profile(b->sh[0].dev, out)); }

/* tools: how GRU-A's rows were dealt to the 8 waves of the sample kernel: out[0] = items per lane, then per wave
 * {bound[0..3], allh[0..2]} (item index where each slot starts / slot holds only candidate rows), then out[57 + wave] =
 * items of slot 0's chains computed one sample ahead (stored end-aligned, [nw - head, nw)) */
int lpcnet_hip_model_layout(const unsigned char *data, int len, int *out)
{
    lpcn_model_host m;
    if (lpcn_model_parse(&m, data, len) != 0) { set_err("malformed or incomplete DNNw weight blob"); return -1; }
    out[0] = m.nw;
    for (int w = 0; w < LPCN_WAVES; w++) {
        for (int k = 0; k < 4; k++) out[1 + w * 7 + k] = m.pk_a_bound[w][k];
        for (int k = 0; k < 3; k++) out[1 + w * 7 + 4 + k] = m.pk_a_allh[w][k];
        out[57 + w] = m.pk_a_head[w];
    }
    lpcn_model_release(&m);
    return 0;
}

int lpcnet_hip_exp10_device(const float *x, double *out, int n)
{
    if (!x || !out || n <= 0) { set_err("lpcnet_hip_exp10_device: bad arguments"); return LPCN_E_ARG; }
    FWD(lpcn_debug_exp10(single_stream_device(), x, out, (size_t)n));
}

/* test seam (include/lpcnet_batch.h): the arithmetic identities the PARITY kernels rest on, on the device */
int lpcnet_hip_arith_identities_device(const float *a, const float *b, unsigned *out_mfma, unsigned *out_mul, unsigned *out_pk, unsigned *out_sc, int n)
{
    if (!a || !b || !out_mfma || !out_mul || !out_pk || !out_sc || n <= 0) { set_err("lpcnet_hip_arith_identities_device: bad arguments"); return LPCN_E_ARG; }
    FWD(lpcn_debug_arith_identities(single_stream_device(), a, b, out_mfma, out_mul, out_pk, out_sc, (size_t)n));
}

/* test seam (include/lpcnet_batch.h): the int8 kernels' one-instruction re-quantisation against the reference's formula, exhaustively */
int lpcnet_hip_quant_sweep_device(unsigned long long *out3)
{
    if (!out3) { set_err("lpcnet_hip_quant_sweep_device: bad arguments"); return LPCN_E_ARG; }
    FWD(lpcn_debug_quant_sweep(single_stream_device(), out3));
}

/* Host-only model check (no GPU needed): parses the blob with the loader's rules, builds the device
 * packings and verifies them.  info[0..5] = {is_int8, blocks GRU-A, blocks GRU-B, items per lane,
 * padded GRU-B blocks, selftest code}.  Returns 0 if the blob is loadable by the engine (float or
 * int8 flavour, see info[0]), -1 if it is malformed (lpcnet_load_model would fail). */
int lpcnet_hip_check_model(const unsigned char *data, int len, int *info)
{
    lpcn_model_host m;
    if (lpcn_model_parse(&m, data, len) != 0) { set_err("malformed or incomplete DNNw weight blob"); return -1; }
    int st = lpcn_model_selftest(&m);
    if (!st) {                                           /* the FAST arithmetic's own GRU-A packing, where there is one (int8 blobs) */
        lpcn_model_host f;
        const int pf = lpcn_model_pack_fast(&m, &f);
        if (pf < 0) st = 100;
        else if (pf == 0) {
            f.pk_b_w = m.pk_b_w; f.pk_b_wq = m.pk_b_wq; f.pk_b_start = m.pk_b_start; f.pk_b_blk = m.pk_b_blk;      /* (the check reads GRU-B's packing too: shared) */
            const int sf = lpcn_model_selftest(&f);
            if (sf) st = 100 + sf;
            f.pk_b_w = NULL; f.pk_b_wq = NULL; f.pk_b_start = NULL; f.pk_b_blk = NULL;
            lpcn_model_release(&f);
        }
    }
    if (!st && !m.is_int8) {                             /* the two-group kernel's own GRU-A packing (float blobs that fit it) */
        lpcn_model_host f;
        if (lpcn_model_pack_x2(&m, &f) == 0) {
            f.pk_b_w = m.pk_b_w; f.pk_b_start = m.pk_b_start; f.pk_b_blk = m.pk_b_blk;
            const int sf = lpcn_model_selftest(&f);
            if (sf) st = 200 + sf;
            f.pk_b_w = NULL; f.pk_b_start = NULL; f.pk_b_blk = NULL;
            lpcn_model_release(&f);
        }
    }
    if (info) { info[0] = m.is_int8; info[1] = m.nb_a; info[2] = m.nb_b; info[3] = m.nw; info[4] = m.nb_b_padded; info[5] = st; }
    lpcn_model_release(&m);
    if (st) { set_err("internal error: device packing inconsistent with blob"); return -1; }
    return 0;
}
