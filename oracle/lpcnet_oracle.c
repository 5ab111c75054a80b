/* TEST INFRASTRUCTURE ONLY -- see lpcnet_oracle.h for scope, pinning and usage rules.
 *
 * CPU restatement of the LPCNet synthesis hot path.  Written from the algorithm, organised
 * around flat arrays rather than the reference's layer structs; every routine names the
 * reference lines whose arithmetic (operation order, float/double promotion) it reproduces.
 * Must be built with -ffp-contract=off: the reference's reproducible flavour performs every
 * multiply and add as a separately rounded IEEE operation.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "lpcnet_oracle.h"
#include "orc_tables_gen.h"

#define N_A   ORC_N_A
#define N_B   ORC_N_B
#define COND  ORC_COND
#define NFEAT ORC_NB_FEAT
#define FIN   ORC_FRAME_IN
#define LPCO  ORC_LPC_ORDER
#define NBANDS ORC_NB_BANDS

/* ======================================================================================
 * model: named arrays located inside the caller's DNNw blob
 * ====================================================================================== */
struct orc_model {
    int is_int8;
    int end2end;
    int nb_a, nb_b;                     /* number of 8x4 blocks in GRU-A / GRU-B input matrix */
    float lpc_gamma;
    const float *emb_sig, *emb_pred, *emb_exc;         /* [256][3*N_A] pre-multiplied tables  */
    const float *a_dense_w, *a_dense_b;                /* [COND][3*N_A], [3*N_A]              */
    const float *b_dense_w, *b_dense_b;                /* [COND][3*N_B], [3*N_B]              */
    const float *conv1_w, *conv1_b, *conv2_w, *conv2_b;/* [3][in][COND]                       */
    const float *pitch_emb;                            /* [256][64]                           */
    const float *dense1_w, *dense1_b, *dense2_w, *dense2_b;
    const float *fc_w, *fc_b, *fc_f;                   /* [256][2][N_B], [2][256], [2][256]   */
    const float *a_bias, *a_diag;                      /* [2][3*N_A], [3*N_A]                 */
    const void  *a_w; const int *a_idx;                /* block-sparse recurrent weights      */
    const float *b_bias;                               /* [2][3*N_B]                          */
    const void  *b_w; const int *b_idx;                /* GRU-B input weights (block-sparse)  */
    const void  *b_rec;                                /* GRU-B recurrent weights             */
};

typedef struct { const char *name; int type, size; const void *data; } rec_t;

/* record walk: src/parse_lpcnet_weights.c:36-77 (64-byte header, payload padded to 64) */
static int walk_blob(const unsigned char *p, int len, rec_t *out, int cap)
{
    int n = 0;
    while (len > 0) {
        int32_t hdr[5];
        if (len < 64) return -1;
        memcpy(hdr, p, sizeof(hdr));                    /* "DNNw", version, type, size, block */
        if (hdr[4] < hdr[3] || hdr[4] > len - 64 || hdr[3] < 0) return -1;
        if (p[20 + 43] != 0) return -1;                 /* name must be NUL-terminated        */
        if (hdr[3] <= 0) return -1;                     /* the reference rejects empty arrays */
        if (n >= cap) return -1;
        out[n].name = (const char *)p + 20;
        out[n].type = hdr[2];
        out[n].size = hdr[3];
        out[n].data = p + 64;
        n++;
        p += 64 + hdr[4];
        len -= 64 + hdr[4];
    }
    return n;
}

static const rec_t *find_rec(const rec_t *r, int n, const char *name)
{
    int i;
    for (i = 0; i < n; i++) if (strcmp(r[i].name, name) == 0) return &r[i];
    return NULL;
}

static const void *need(const rec_t *r, int n, const char *name, int bytes)
{
    const rec_t *e = find_rec(r, n, name);
    return (e && e->size == bytes) ? e->data : NULL;
}

/* index stream check: src/parse_lpcnet_weights.c:90-113 */
static const int *need_idx(const rec_t *r, int n, const char *name, int nb_in, int nb_out, int *blocks)
{
    const rec_t *e = find_rec(r, n, name);
    const int *idx;
    int remain;
    *blocks = 0;
    if (!e) return NULL;
    idx = (const int *)e->data;
    remain = e->size / (int)sizeof(int);
    while (remain > 0) {
        int cnt = *idx++, i;
        if (remain < cnt + 1) return NULL;
        for (i = 0; i < cnt; i++) {
            int pos = *idx++;
            if (pos + 3 >= nb_in || (pos & 3)) return NULL;
        }
        nb_out -= 8;
        remain -= cnt + 1;
        *blocks += cnt;
    }
    return nb_out == 0 ? (const int *)e->data : NULL;
}

orc_model *orc_model_parse(const unsigned char *blob, int len, float lpc_gamma)
{
    rec_t recs[64];
    int n = walk_blob(blob, len, recs, 64);
    orc_model *m;
    const rec_t *qa;
    int qsz;
    if (n <= 0) return NULL;
    m = (orc_model *)calloc(1, sizeof(*m));
    m->lpc_gamma = lpc_gamma;
    m->end2end = 0;
#define F(field, name, count) if (!(m->field = (const float *)need(recs, n, name, (count) * 4))) goto fail
    F(emb_sig,  "gru_a_embed_sig_weights",  256 * 3 * N_A);
    F(emb_pred, "gru_a_embed_pred_weights", 256 * 3 * N_A);
    F(emb_exc,  "gru_a_embed_exc_weights",  256 * 3 * N_A);
    F(a_dense_w, "gru_a_dense_feature_weights", COND * 3 * N_A);
    F(a_dense_b, "gru_a_dense_feature_bias", 3 * N_A);
    F(b_dense_w, "gru_b_dense_feature_weights", COND * 3 * N_B);
    F(b_dense_b, "gru_b_dense_feature_bias", 3 * N_B);
    F(conv1_w, "feature_conv1_weights", 3 * FIN * COND);
    F(conv1_b, "feature_conv1_bias", COND);
    F(conv2_w, "feature_conv2_weights", 3 * COND * COND);
    F(conv2_b, "feature_conv2_bias", COND);
    F(pitch_emb, "embed_pitch_weights", 256 * ORC_PITCH_EMB);
    F(dense1_w, "feature_dense1_weights", COND * COND);
    F(dense1_b, "feature_dense1_bias", COND);
    F(dense2_w, "feature_dense2_weights", COND * COND);
    F(dense2_b, "feature_dense2_bias", COND);
    F(fc_w, "dual_fc_weights", 256 * 2 * N_B);
    F(fc_b, "dual_fc_bias", 2 * 256);
    F(fc_f, "dual_fc_factor", 2 * 256);
    F(a_bias, "sparse_gru_a_bias", 6 * N_A);
    F(a_diag, "sparse_gru_a_recurrent_weights_diag", 3 * N_A);
    F(b_bias, "gru_b_bias", 6 * N_B);
#undef F
    /* arrays the reference binds although inference never reads them */
    if (!need(recs, n, "embed_sig_weights", 256 * 128 * 4)) goto fail;
    if (!need(recs, n, "sparse_gru_a_subias", 6 * N_A * 4)) goto fail;
    if (!need(recs, n, "gru_b_subias", 6 * N_B * 4)) goto fail;
    if (!(m->a_idx = need_idx(recs, n, "sparse_gru_a_recurrent_weights_idx", N_A, 3 * N_A, &m->nb_a))) goto fail;
    if (!(m->b_idx = need_idx(recs, n, "gru_b_weights_idx", N_A, 3 * N_B, &m->nb_b))) goto fail;
    qa = find_rec(recs, n, "sparse_gru_a_recurrent_weights");
    if (!qa) goto fail;
    if (qa->size == 32 * m->nb_a * 4) m->is_int8 = 0;
    else if (qa->size == 32 * m->nb_a) m->is_int8 = 1;
    else goto fail;
    qsz = m->is_int8 ? 1 : 4;
    m->a_w = qa->data;
    if (!(m->b_w = need(recs, n, "gru_b_weights", 32 * m->nb_b * qsz))) goto fail;
    if (!(m->b_rec = need(recs, n, "gru_b_recurrent_weights", 3 * N_B * N_B * qsz))) goto fail;
    return m;
fail:
    free(m);
    return NULL;
}

void orc_model_free(orc_model *m) { free(m); }
int  orc_model_is_int8(const orc_model *m) { return m->is_int8; }
void orc_model_set_end2end(orc_model *m, int on) { m->end2end = on != 0; }      /* the reference's compile-time END2END */
int  orc_model_nb_blocks(const orc_model *m, int which) { return which ? m->nb_b : m->nb_a; }

/* ======================================================================================
 * scalar helpers
 * ====================================================================================== */

/* src/vec.h:82-99: nearest table knot (spacing 0.04, 201 knots), then one correction step
 * y + dx*(1-y^2)*(1 - y*dx), all in float; floor() is taken on the double-promoted value. */
float orc_tanh_approx(float x)
{
    float sign = 1, y, dy;
    int i;
    if (x < 0) { x = -x; sign = -1; }
    i = (int)floor(.5f + 25 * x);
    if (i > 200) i = 200;
    if (i < 0) i = 0;
    x -= .04f * i;
    y = orc_tansig[i];
    dy = 1 - y * y;
    y = y + x * dy * (1 - y * x);
    return sign * y;
}

/* src/vec.h:101-104 */
float orc_sigmoid_approx(float x) { return .5f + .5f * orc_tanh_approx(.5f * x); }

/* src/common.h:18-33: exponent split + cubic on the mantissa */
static float log2_approx(float x)
{
    union { float f; int i; } in;
    int integer;
    float frac;
    in.f = x;
    integer = (in.i >> 23) - 127;
    in.i -= integer << 23;
    frac = in.f - 1.5f;
    frac = -0.41445418f + frac * (0.95909232f + frac * (-0.33951290f + frac * 0.16541097f));
    return 1 + integer + frac;
}

/* src/common.h:47-58.  Float throughout except the final floor(.5 + u) in double. */
int orc_lin2ulaw(float x)
{
    float u;
    float scale = 255.f / 32768.f;
    int s = x >= 0 ? 1 : -1;
    x = (float)fabs(x);
    u = (s * (128 * (0.69315f * log2_approx(1 + scale * x)) / 5.5451774445f));
    u = 128 + u;
    if (u < 0) u = 0;
    if (u > 255) u = 255;
    return (int)floor(.5 + u);
}

/* src/common.h:37-45 for integer codes (tabulated by tools/gen_tables.py with the same
 * double-precision exp) */
float orc_ulaw2lin(int code) { return orc_ulaw2lin_tab[code & 255]; }
float orc_logit_table(int i) { return orc_logit_tab[i & 255]; }

/* src/kiss99.c:59-81 */
unsigned orc_kiss99_rand(unsigned *c)
{
    uint32_t z = c[0], w = c[1], jsr = c[2], jcong = c[3];
    uint32_t znew = 36969 * (z & 0xFFFF) + (z >> 16);
    uint32_t wnew = 18000 * (w & 0xFFFF) + (w >> 16);
    uint32_t mwc = (znew << 16) + wnew;
    jsr ^= jsr << 13;
    jsr ^= jsr >> 17;
    jsr ^= jsr << 5;
    jcong = 69069 * jcong + 1234567;
    c[0] = znew; c[1] = wnew; c[2] = jsr; c[3] = jcong;
    return (mwc ^ jcong) + jsr;
}

/* src/kiss99.c:34-57 */
void orc_kiss99_srand(unsigned *c, const unsigned char *d, int n)
{
    int i;
    c[0] = 362436069u; c[1] = 521288629u; c[2] = 123456789u; c[3] = 380116160u;
    for (i = 3; i < n; i += 4) {
        c[0] ^= d[i - 3]; c[1] ^= d[i - 2]; c[2] ^= d[i - 1]; c[3] ^= d[i];
        orc_kiss99_rand(c);
    }
    if (i - 3 < n) c[0] ^= d[i - 3];
    if (i - 2 < n) c[1] ^= d[i - 2];
    if (i - 1 < n) c[2] ^= d[i - 1];
    if (c[0] == 0 || c[0] == 0x9068FFFFu) c[0]++;
    if (c[1] == 0 || c[1] == 0x464FFFFFu) c[1]++;
    if (c[2] == 0) c[2]++;
}

/* ======================================================================================
 * mat-vec kernels, generic-C order
 * ====================================================================================== */

/* out[i] += sum_j W[j*stride+i]*x[j], j ascending, one rounded mul and one rounded add per term
 * (src/nnet.c:73-86 -> src/vec.h:131-162; rows are independent so the 16-row tiling is moot) */
static void gemv_cols(float *out, const float *W, int rows, int cols, int stride, const float *x)
{
    int i, j;
    for (i = 0; i < rows; i++) {
        float acc = out[i];
        for (j = 0; j < cols; j++) acc += W[j * stride + i] * x[j];
        out[i] = acc;
    }
}

/* float block-sparse accumulate, src/vec.h:347-404: block = [in 4][out 8]; per output row the
 * blocks are visited in index order and the 4 columns of a block in order 0..3 */
static void sparse_f32(float *out, const float *w, int rows, const int *idx, const float *x)
{
    int g, j, r, k;
    for (g = 0; g < rows; g += 8) {
        int cnt = *idx++;
        for (j = 0; j < cnt; j++) {
            int pos = *idx++;
            for (k = 0; k < 4; k++) {
                float xk = x[pos + k];
                for (r = 0; r < 8; r++) out[g + r] += w[k * 8 + r] * xk;
            }
            w += 32;
        }
    }
}

/* signed-activation int8 quantiser, src/vec.h:280,311: floor(.5 + 127*x) in double */
static void quant_s8(signed char *q, const float *x, int n)
{
    int i;
    for (i = 0; i < n; i++) q[i] = (signed char)(int)floor(.5 + 127 * x[i]);
}

#define Q_SCALE   (128.f * 127.f)
#define Q_SCALE_1 (1.f / 128.f / 127.f)

/* int8 block-sparse accumulate, generic flavour, src/vec.h:306-339: block = [out 8][in 4];
 * out is pre-scaled by 128*127, each block adds an exact integer dot to the float accumulator */
static void sparse_s8(float *out, const signed char *w, int rows, int cols, const int *idx, const float *xf)
{
    signed char x[2048];
    int g, j, r, i;
    for (i = 0; i < rows; i++) out[i] *= Q_SCALE;
    quant_s8(x, xf, cols);
    for (g = 0; g < rows; g += 8) {
        int cnt = *idx++;
        for (j = 0; j < cnt; j++) {
            int pos = *idx++;
            int x0 = x[pos], x1 = x[pos + 1], x2 = x[pos + 2], x3 = x[pos + 3];
            for (r = 0; r < 8; r++)
                out[g + r] += (w[4 * r] * x0 + w[4 * r + 1] * x1 + w[4 * r + 2] * x2 + w[4 * r + 3] * x3);
            w += 32;
        }
    }
    for (i = 0; i < rows; i++) out[i] *= Q_SCALE_1;
}

/* int8 dense accumulate, src/vec.h:274-304: weights pre-blocked [rows/8][cols/4][8][4]; the
 * activations are converted to float before the products, which are exact, so the bracketed
 * 4-term sum equals the integer dot */
static void dense_s8(float *out, const signed char *w, int rows, int cols, const float *xf)
{
    signed char x[2048];
    int g, j, r, i;
    for (i = 0; i < rows; i++) out[i] *= Q_SCALE;
    quant_s8(x, xf, cols);
    for (g = 0; g < rows; g += 8) {
        for (j = 0; j < cols; j += 4) {
            float x0 = x[j], x1 = x[j + 1], x2 = x[j + 2], x3 = x[j + 3];
            for (r = 0; r < 8; r++)
                out[g + r] += (w[4 * r] * x0 + w[4 * r + 1] * x1 + w[4 * r + 2] * x2 + w[4 * r + 3] * x3);
            w += 32;
        }
    }
    for (i = 0; i < rows; i++) out[i] *= Q_SCALE_1;
}

/* ======================================================================================
 * sample-rate network
 * ====================================================================================== */

/* src/nnet.c:484-491: ((cond + E_sig[sig]) + E_pred[pred]) + E_exc[exc] */
void orc_gru_a_input(const orc_model *m, float *out, const float *cond, int sig, int pred, int exc)
{
    int i;
    const float *a = m->emb_sig + sig * 3 * N_A;
    const float *b = m->emb_pred + pred * 3 * N_A;
    const float *c = m->emb_exc + exc * 3 * N_A;
    for (i = 0; i < 3 * N_A; i++) out[i] = cond[i] + a[i] + b[i] + c[i];
}

/* src/nnet.c:410-448.  `input` already contains the input-side pre-activations. */
void orc_sparse_gru_a(const orc_model *m, float *state, const float *input)
{
    float recur[3 * N_A];
    const float *bias = m->a_bias + 3 * N_A;            /* recurrent bias row (no USE_SU_BIAS) */
    int i, k;
    for (k = 0; k < 3; k++)
        for (i = 0; i < N_A; i++) {
            float v = bias[k * N_A + i] + m->a_diag[k * N_A + i] * state[i];
            recur[k * N_A + i] = (k < 2) ? v + input[k * N_A + i] : v;
        }
    if (m->is_int8) sparse_s8(recur, (const signed char *)m->a_w, 3 * N_A, N_A, m->a_idx, state);
    else            sparse_f32(recur, (const float *)m->a_w, 3 * N_A, m->a_idx, state);
    for (i = 0; i < 2 * N_A; i++) recur[i] = orc_sigmoid_approx(recur[i]);
    for (i = 0; i < N_A; i++) {
        float z = recur[i], r = recur[N_A + i];
        float h = orc_tanh_approx(recur[2 * N_A + i] * r + input[2 * N_A + i]);
        state[i] = z * state[i] + (1 - z) * h;
    }
}

/* src/nnet.c:326-372 */
void orc_gru_b(const orc_model *m, const float *cond_b, float *state, const float *input)
{
    float zrh[3 * N_B], recur[3 * N_B];
    int i;
    for (i = 0; i < 3 * N_B; i++) zrh[i] = m->b_bias[i] + cond_b[i];
    if (m->is_int8) sparse_s8(zrh, (const signed char *)m->b_w, 3 * N_B, N_A, m->b_idx, input);
    else            sparse_f32(zrh, (const float *)m->b_w, 3 * N_B, m->b_idx, input);
    for (i = 0; i < 3 * N_B; i++) recur[i] = m->b_bias[3 * N_B + i];
    if (m->is_int8) dense_s8(recur, (const signed char *)m->b_rec, 3 * N_B, N_B, state);
    else            gemv_cols(recur, (const float *)m->b_rec, 3 * N_B, N_B, 3 * N_B, state);
    for (i = 0; i < 2 * N_B; i++) zrh[i] = orc_sigmoid_approx(zrh[i] + recur[i]);
    for (i = 0; i < N_B; i++) {
        float z = zrh[i], r = zrh[N_B + i];
        float h = orc_tanh_approx(zrh[2 * N_B + i] + recur[2 * N_B + i] * r);
        state[i] = z * state[i] + (1 - z) * h;
    }
}

/* src/nnet.c:163-214: 8-level binary tree; level b evaluates dual-FC row i=(1<<b)|prefix and
 * compares the logit with a pre-drawn threshold (2 RNG words = 8 threshold bytes) */
int orc_sample_mdense(const orc_model *m, const float *input, unsigned *rng)
{
    float thr[8];
    int b, j, val = 0;
    for (b = 0; b < 8; b += 4) {
        unsigned r = orc_kiss99_rand(rng);
        thr[b]     = orc_logit_tab[r & 0xFF];
        thr[b + 1] = orc_logit_tab[(r >> 8) & 0xFF];
        thr[b + 2] = orc_logit_tab[(r >> 16) & 0xFF];
        thr[b + 3] = orc_logit_tab[(r >> 24) & 0xFF];
    }
    for (b = 0; b < 8; b++) {
        int i = (1 << b) | val;
        const float *w = m->fc_w + i * 2 * N_B;
        float s1 = m->fc_b[i], s2 = m->fc_b[256 + i];
        for (j = 0; j < N_B; j++) {
            s1 += w[j] * input[j];
            s2 += w[N_B + j] * input[j];
        }
        s1 = m->fc_f[i] * orc_tanh_approx(s1);
        s2 = m->fc_f[256 + i] * orc_tanh_approx(s2);
        s1 += s2;
        val = (val << 1) | (thr[b] < s1);
    }
    return val;
}

/* ---- the engine's "fp16 dual FC" sub-option (BASELINE.json config 4; lpcnet_batch_set_fast(b, 2)) -----------------------
 * The reference has no fp16 arithmetic; this is the oracle-side statement of what the option computes, so that the engine's
 * tree can be checked against something other than itself: dual-FC weights and the GRU-B state rounded to binary16
 * (round to nearest even), products exact (11 x 11 significand bits fit fp32), sums accumulated in fp32 in index order, two
 * terms per step -- the hardware's v_dot2_f32_f16 does not document whether the two products of a step are summed exactly
 * before the accumulator is added (variant 1) or added one after the other with an fp32 rounding each (variant 0): both
 * are provided, the test's tolerance covers their difference.  tanh is evaluated in double (the engine's FAST flavour
 * uses the hardware exponential and reciprocal, a few 1e-7 off).  logits[b] = the logit of the node on `path` at level b. */
static float orc_f16_round(float x)
{
    union { float f; unsigned u; } v, r;
    v.f = x;
    const unsigned sign = v.u & 0x80000000u, a = v.u & 0x7FFFFFFFu;
    if (a >= 0x7F800000u) return x;                              /* inf / nan */
    if (a >= 0x477FF000u) { r.u = sign | 0x7F800000u; return r.f; }   /* rounds to >= 65520: overflow to inf */
    if (a < 0x33000000u) { r.u = sign; return r.f; }            /* < 2^-25: rounds to zero */
    if (a < 0x38800000u) {                                       /* subnormal half: multiples of 2^-24 */
        const double q = (double)fabsf(x) * 16777216.0;         /* / 2^-24 */
        double fl = floor(q), fr = q - fl;
        if (fr > 0.5 || (fr == 0.5 && ((long long)fl & 1))) fl += 1.0;
        r.f = (float)(fl / 16777216.0);
        r.u |= sign;
        return r.f;
    }
    {                                                            /* normal: keep 10 fraction bits, RNE on the 13 dropped */
        unsigned m = a, rem = m & 0x1FFFu;
        m &= ~0x1FFFu;
        if (rem > 0x1000u || (rem == 0x1000u && (m & 0x2000u))) m += 0x2000u;
        r.u = sign | m;
        return r.f;
    }
}

float orc_f16(float x) { return orc_f16_round(x); }         /* (exported for the test that pins it to IEEE binary16) */

void orc_mdense_f16_path(const orc_model *m, const float *input, int path, int variant, float *logits)
{
    float h[N_B];
    int b, j, c;
    for (j = 0; j < N_B; j++) h[j] = orc_f16_round(input[j]);
    for (b = 0; b < 8; b++) {
        const int i = (1 << b) | (path >> (8 - b));
        float s[2];
        for (c = 0; c < 2; c++) {
            const float *w = m->fc_w + i * 2 * N_B + c * N_B;
            float acc = m->fc_b[c * 256 + i];
            for (j = 0; j < N_B; j += 2) {
                const float w0 = orc_f16_round(w[j]), w1 = orc_f16_round(w[j + 1]);
                if (variant) acc = (float)((double)acc + ((double)w0 * h[j] + (double)w1 * h[j + 1]));
                else { acc = acc + w0 * h[j]; acc = acc + w1 * h[j + 1]; }
            }
            s[c] = m->fc_f[c * 256 + i] * (float)tanh((double)acc);
        }
        logits[b] = s[0] + s[1];
    }
}

/* ======================================================================================
 * per-stream state and the two rates
 * ====================================================================================== */
struct orc_state {
    const orc_model *m;
    unsigned rng[4];
    /* everything below is cleared by reset (src/lpcnet_private.h:33, src/lpcnet.c:177-179) */
    float conv1_mem[2 * FIN], conv2_mem[2 * COND], gru_a[N_A], gru_b[N_B];
    int last_exc;
    float last_sig[LPCO];
    float old_lpc[ORC_FEATURES_DELAY][LPCO];
    float cond_a[3 * N_A], cond_b[3 * N_B];
    int frame_count;
    float deemph_mem;
    float lpc[LPCO];
};

void orc_state_reset(orc_state *st)
{
    const orc_model *m = st->m;
    memset(st, 0, sizeof(*st));
    st->m = m;
    st->last_exc = orc_lin2ulaw(0.f);
    orc_kiss99_srand(st->rng, (const unsigned char *)"LPCNet", 6);
}

orc_state *orc_state_create(const orc_model *m)
{
    orc_state *st = (orc_state *)calloc(1, sizeof(*st));
    st->m = m;
    orc_state_reset(st);
    return st;
}
void orc_state_destroy(orc_state *st) { free(st); }

static void dense_layer(float *out, const float *W, const float *b, int n_in, int n_out,
                        const float *x, int do_tanh)
{
    int i;
    for (i = 0; i < n_out; i++) out[i] = b[i];
    gemv_cols(out, W, n_out, n_in, n_out, x);
    if (do_tanh) for (i = 0; i < n_out; i++) out[i] = orc_tanh_approx(out[i]);
}

/* src/nnet.c:452-470: window = [older | old | new], weights [tap][in][out] */
static void conv_layer(float *out, float *mem, const float *W, const float *b, int n_in, const float *x)
{
    float win[3 * 128];
    memcpy(win, mem, 2 * n_in * sizeof(float));
    memcpy(win + 2 * n_in, x, n_in * sizeof(float));
    dense_layer(out, W, b, 3 * n_in, COND, win, 1);
    memcpy(mem, win + n_in, 2 * n_in * sizeof(float));
}

/* src/lpcnet.c:82-120 */
void orc_frame_network(orc_state *st, const float *features, float *cond_a, float *cond_b, float *lpc)
{
    const orc_model *m = st->m;
    float in[FIN], c1[COND], c2[COND], d1[COND], cond[COND];
    int pitch, i;
    pitch = (int)floor(.1 + 50 * features[NBANDS] + 100);
    if (pitch < 33) pitch = 33;
    if (pitch > 255) pitch = 255;
    memcpy(in, features, NFEAT * sizeof(float));
    memcpy(in + NFEAT, m->pitch_emb + pitch * ORC_PITCH_EMB, ORC_PITCH_EMB * sizeof(float));
    conv_layer(c1, st->conv1_mem, m->conv1_w, m->conv1_b, FIN, in);
    if (st->frame_count < 1) memset(c1, 0, sizeof(c1));
    conv_layer(c2, st->conv2_mem, m->conv2_w, m->conv2_b, COND, c1);
    if (st->frame_count < ORC_FEATURES_DELAY) memset(c2, 0, sizeof(c2));
    dense_layer(d1, m->dense1_w, m->dense1_b, COND, COND, c2, 1);
    dense_layer(cond, m->dense2_w, m->dense2_b, COND, COND, d1, 1);
    dense_layer(cond_a, m->a_dense_w, m->a_dense_b, COND, 3 * N_A, cond, 0);
    dense_layer(cond_b, m->b_dense_w, m->b_dense_b, COND, 3 * N_B, cond, 0);
    if (m->end2end) {
        /* END2END models: the first 16 conditioning outputs are reflection coefficients, src/lpcnet.c:56-80,107-108 */
        float tmp[LPCO], ntmp[LPCO] = {0};
        int j, k;
        memcpy(tmp, cond, sizeof(tmp));
        for (i = 0; i < LPCO; i++) {
            for (j = 0; j <= i - 1; j++) ntmp[j] = tmp[j] + tmp[i] * tmp[i - j - 1];
            for (k = 0; k <= i - 1; k++) tmp[k] = ntmp[k];
        }
        memcpy(lpc, tmp, sizeof(tmp));
    } else {
        /* two-frame LPC delay line, src/lpcnet.c:110-112 */
        memcpy(lpc, st->old_lpc[ORC_FEATURES_DELAY - 1], LPCO * sizeof(float));
        memmove(st->old_lpc[1], st->old_lpc[0], (ORC_FEATURES_DELAY - 1) * LPCO * sizeof(float));
        orc_lpc_from_cepstrum(st->old_lpc[0], features);
    }
    {   /* src/freq.c:299-308 */
        float g = m->lpc_gamma, gi = g;
        for (i = 0; i < LPCO; i++) { lpc[i] *= gi; gi *= g; }
    }
    if (st->frame_count < 1000) st->frame_count++;
}

/* src/lpcnet.c:146-167 */
int orc_sample_network(orc_state *st, const float *cond_a, const float *cond_b,
                       int last_exc, int last_sig, int pred)
{
    float a_in[3 * N_A];
    orc_gru_a_input(st->m, a_in, cond_a, last_sig, pred, last_exc);
    orc_sparse_gru_a(st->m, st->gru_a, a_in);
    orc_gru_b(st->m, cond_b, st->gru_b, st->gru_a);
    return orc_sample_mdense(st->m, st->gru_b, st->rng);
}

/* src/lpcnet.c:235-271 */
static void tail(orc_state *st, short *output, int N, int preload)
{
    int i, j;
    if (st->frame_count <= ORC_FEATURES_DELAY) {
        memset(output, 0, N * sizeof(short));
        return;
    }
    for (i = 0; i < N; i++) {
        float pred = 0, pcm;
        int exc, sig_u, pred_u;
        for (j = 0; j < LPCO; j++) pred -= st->last_sig[j] * st->lpc[j];
        sig_u = orc_lin2ulaw(st->last_sig[0]);
        pred_u = orc_lin2ulaw(pred);
        exc = orc_sample_network(st, st->cond_a, st->cond_b, st->last_exc, sig_u, pred_u);
        if (i < preload) {
            exc = orc_lin2ulaw(output[i] - 0.85f * st->deemph_mem - pred);
            pcm = output[i] - 0.85f * st->deemph_mem;
        } else {
            pcm = pred + orc_ulaw2lin(exc);
        }
        memmove(&st->last_sig[1], &st->last_sig[0], (LPCO - 1) * sizeof(float));
        st->last_sig[0] = pcm;
        st->last_exc = exc;
        pcm += 0.85f * st->deemph_mem;
        st->deemph_mem = pcm;
        if (pcm < -32767) pcm = -32767;
        if (pcm > 32767) pcm = 32767;
        if (i >= preload) output[i] = (short)(int)floor(.5 + pcm);
    }
}

void orc_synthesize(orc_state *st, const float *features, short *output, int N, int preload)
{
    orc_frame_network(st, features, st->cond_a, st->cond_b, st->lpc);
    tail(st, output, N, preload);
}

void orc_synthesize_tail(orc_state *st, const float *cond_a, const float *cond_b, const float *lpc,
                         short *output, int N, int preload)
{
    memcpy(st->cond_a, cond_a, sizeof(st->cond_a));
    memcpy(st->cond_b, cond_b, sizeof(st->cond_b));
    memcpy(st->lpc, lpc, sizeof(st->lpc));
    tail(st, output, N, preload);
}

void orc_get_nnet_state(const orc_state *st, float *conv1, float *conv2, float *gru_a, float *gru_b)
{
    if (conv1) memcpy(conv1, st->conv1_mem, sizeof(st->conv1_mem));
    if (conv2) memcpy(conv2, st->conv2_mem, sizeof(st->conv2_mem));
    if (gru_a) memcpy(gru_a, st->gru_a, sizeof(st->gru_a));
    if (gru_b) memcpy(gru_b, st->gru_b, sizeof(st->gru_b));
}
void orc_set_gru_state(orc_state *st, const float *gru_a, const float *gru_b)
{
    if (gru_a) memcpy(st->gru_a, gru_a, sizeof(st->gru_a));
    if (gru_b) memcpy(st->gru_b, gru_b, sizeof(st->gru_b));
}
void orc_get_frame_products(const orc_state *st, float *lpc, float *cond_a, float *cond_b)
{
    memcpy(lpc, st->lpc, sizeof(st->lpc));
    memcpy(cond_a, st->cond_a, sizeof(st->cond_a));
    memcpy(cond_b, st->cond_b, sizeof(st->cond_b));
}
void orc_get_signal_state(const orc_state *st, float *last_sig, int *last_exc, float *deemph_mem,
                          int *frame_count, unsigned *rng4)
{
    memcpy(last_sig, st->last_sig, sizeof(st->last_sig));
    *last_exc = st->last_exc;
    *deemph_mem = st->deemph_mem;
    *frame_count = st->frame_count;
    memcpy(rng4, st->rng, sizeof(st->rng));
}
void orc_force_frame_count(orc_state *st, int fc) { st->frame_count = fc; }

/* ======================================================================================
 * LPC from cepstrum (100 Hz): src/freq.c:310-320 -> idct :230, lpc_from_bands :275-297
 * ====================================================================================== */
static const short band_edge[NBANDS] = {   /* src/freq.c:46-49, in 5 ms bins (x4 FFT bins) */
    0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28, 34, 40
};
static const float band_comp[NBANDS] = {   /* src/freq.c:51-53 */
    0.8f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 0.666667f, 0.5f, 0.5f, 0.5f, 0.333333f, 0.25f,
    0.25f, 0.2f, 0.166667f, 0.173913f
};

typedef struct { float r, i; } cpx;

static cpx cmul(cpx a, cpx b)
{
    cpx m;
    m.r = a.r * b.r - a.i * b.i;
    m.i = a.r * b.i + a.i * b.r;
    return m;
}
static cpx cadd(cpx a, cpx b) { cpx m; m.r = a.r + b.r; m.i = a.i + b.i; return m; }
static cpx csub(cpx a, cpx b) { cpx m; m.r = a.r - b.r; m.i = a.i - b.i; return m; }
#define TW(k) (*(const cpx *)&orc_fft_tw[2 * (k)])

/* radix-4 pass over N sub-transforms of length 4*m laid out with pitch mm
 * (src/kiss_fft.c:101-170; the m==1 branch has unit twiddles) */
static void pass4(cpx *f, int fstride, int m, int N, int mm)
{
    int i, j;
    if (m == 1) {
        for (i = 0; i < N; i++, f += 4) {
            cpx s0 = csub(f[0], f[2]), s1;
            f[0] = cadd(f[0], f[2]);
            s1 = cadd(f[1], f[3]);
            f[2] = csub(f[0], s1);
            f[0] = cadd(f[0], s1);
            s1 = csub(f[1], f[3]);
            f[1].r = s0.r + s1.i;
            f[1].i = s0.i - s1.r;
            f[3].r = s0.r - s1.i;
            f[3].i = s0.i + s1.r;
        }
        return;
    }
    for (i = 0; i < N; i++) {
        cpx *F = f + i * mm;
        for (j = 0; j < m; j++, F++) {
            cpx a = cmul(F[m], TW(j * fstride));
            cpx b = cmul(F[2 * m], TW(2 * j * fstride));
            cpx c = cmul(F[3 * m], TW(3 * j * fstride));
            cpx d = csub(F[0], b), e, g;
            F[0] = cadd(F[0], b);
            e = cadd(a, c);
            g = csub(a, c);
            F[2 * m] = csub(F[0], e);
            F[0] = cadd(F[0], e);
            F[m].r = d.r + g.i;
            F[m].i = d.i - g.r;
            F[3 * m].r = d.r - g.i;
            F[3 * m].i = d.i + g.r;
        }
    }
}

/* radix-5 pass (src/kiss_fft.c:232-305) */
static void pass5(cpx *f, int fstride, int m, int N, int mm)
{
    cpx ya = TW(fstride * m), yb = TW(fstride * 2 * m);
    int i, u;
    for (i = 0; i < N; i++) {
        cpx *F0 = f + i * mm, *F1 = F0 + m, *F2 = F0 + 2 * m, *F3 = F0 + 3 * m, *F4 = F0 + 4 * m;
        for (u = 0; u < m; u++, F0++, F1++, F2++, F3++, F4++) {
            cpx s0 = *F0;
            cpx s1 = cmul(*F1, TW(u * fstride));
            cpx s2 = cmul(*F2, TW(2 * u * fstride));
            cpx s3 = cmul(*F3, TW(3 * u * fstride));
            cpx s4 = cmul(*F4, TW(4 * u * fstride));
            cpx s7 = cadd(s1, s4), s10 = csub(s1, s4), s8 = cadd(s2, s3), s9 = csub(s2, s3);
            cpx s5, s6, s11, s12;
            F0->r = F0->r + (s7.r + s8.r);
            F0->i = F0->i + (s7.i + s8.i);
            s5.r = s0.r + (s7.r * ya.r + s8.r * yb.r);
            s5.i = s0.i + (s7.i * ya.r + s8.i * yb.r);
            s6.r = s10.i * ya.i + s9.i * yb.i;
            s6.i = -(s10.r * ya.i + s9.r * yb.i);
            *F1 = csub(s5, s6);
            *F4 = cadd(s5, s6);
            s11.r = s0.r + (s7.r * yb.r + s8.r * ya.r);
            s11.i = s0.i + (s7.i * yb.r + s8.i * ya.r);
            s12.r = s9.i * ya.i - s10.i * yb.i;
            s12.i = s10.r * yb.i - s9.r * ya.i;
            *F2 = cadd(s11, s12);
            *F3 = csub(s11, s12);
        }
    }
}

/* forward 320-point complex FFT with 1/320 scaling folded into the digit-reversal copy
 * (src/kiss_fft.c:566-586 + :518-564; factors 5,4,4,4 per src/lpcnet_tables.c:197-205).
 * in/out are interleaved re,im. */
void orc_fft320(const float *in_ri, float *out_ri)
{
    cpx *out = (cpx *)out_ri;
    const float scale = 0.0031250000f;
    int i;
    for (i = 0; i < 320; i++) {
        out[orc_fft_bitrev[i]].r = scale * in_ri[2 * i];
        out[orc_fft_bitrev[i]].i = scale * in_ri[2 * i + 1];
    }
    /* stages run from the innermost factor outwards: fstride = 80,20,5,1 */
    pass4(out, 80, 1, 80, 4);
    pass4(out, 20, 4, 20, 16);
    pass4(out, 5, 16, 5, 64);
    pass5(out, 1, 64, 1, 1);
}

/* Levinson-Durbin, float build of src/freq.c:86-127 (the fixed-point shift macros are no-ops) */
static void levinson(float *lpc, const float *ac, int p)
{
    float err = ac[0];
    int i, j;
    memset(lpc, 0, p * sizeof(float));
    if (ac[0] == 0) return;
    for (i = 0; i < p; i++) {
        float rr = 0, r;
        for (j = 0; j < i; j++) rr += lpc[j] * ac[i - j];
        rr += ac[i + 1];
        r = -rr / err;
        lpc[i] = r;
        for (j = 0; j < (i + 1) >> 1; j++) {
            float t1 = lpc[j], t2 = lpc[i - 1 - j];
            lpc[j] = t1 + r * t2;
            lpc[i - 1 - j] = t2 + r * t1;
        }
        err = err - (r * r) * err;
        if (err < .001f * ac[0]) break;
    }
}

void orc_lpc_from_cepstrum(float *lpc, const float *cepstrum)
{
    float tmp[NBANDS], Ex[NBANDS], Xr[161];
    float x[640], y[640], ac[LPCO + 1];
    int i, j;
    memcpy(tmp, cepstrum, sizeof(tmp));
    tmp[0] += 4;
    for (i = 0; i < NBANDS; i++) {                      /* idct, src/freq.c:230-240 */
        float sum = 0;
        for (j = 0; j < NBANDS; j++) sum += tmp[j] * orc_idct_tab[i * NBANDS + j];
        Ex[i] = (float)(sum * sqrt(2. / NBANDS));
    }
    for (i = 0; i < NBANDS; i++) Ex[i] = (float)(pow(10.f, Ex[i]) * band_comp[i]);
    /* interp_band_gain, src/freq.c:202-215: linear interpolation across each band */
    for (i = 0; i < NBANDS - 1; i++) {
        int size = (band_edge[i + 1] - band_edge[i]) * 4;
        for (j = 0; j < size; j++) {
            float frac = (float)j / size;
            Xr[band_edge[i] * 4 + j] = (1 - frac) * Ex[i] + frac * Ex[i + 1];
        }
    }
    Xr[160] = 0;
    /* inverse_transform, src/freq.c:256-273: Hermitian extension, forward FFT, reversed read */
    for (i = 0; i < 161; i++) { x[2 * i] = Xr[i]; x[2 * i + 1] = 0; }
    for (; i < 320; i++) { x[2 * i] = x[2 * (320 - i)]; x[2 * i + 1] = -x[2 * (320 - i) + 1]; }
    orc_fft320(x, y);
    ac[0] = 320 * y[0];
    for (i = 1; i <= LPCO; i++) ac[i] = 320 * y[2 * (320 - i)];
    ac[0] += ac[0] * 1e-4 + 320 / 12 / 38.;              /* -40 dB floor, src/freq.c:291 */
    for (i = 1; i <= LPCO; i++) ac[i] *= (1 - 6e-5 * i * i);
    levinson(lpc, ac, LPCO);
}

/* ======================================================================================
 * codec front-end, src/lpcnet_dec.c:81-155 + src/common.c:37-65
 * ====================================================================================== */
static unsigned take_bits(const unsigned char *buf, int *bitpos, int n)
{
    unsigned d = 0;
    while (n--) {
        d = (d << 1) | ((buf[*bitpos >> 3] >> (7 - (*bitpos & 7))) & 1);
        (*bitpos)++;
    }
    return d;
}

static void interp1(float *x, const float *left, const float *right, int id)
{
    int i;
    for (i = 0; i < NBANDS; i++)
        x[i] = id == 0 ? .5f * (left[i] + right[i]) : (id == 1 ? left[i] : right[i]);
}

void orc_decode_packet(float *F, float *vq_mem, const unsigned char *buf,
                       const float *cb1, const float *cb2, const float *cb3, const float *cbd)
{
    int bp = 0, i, sub, voiced = 1;
    int c0_id = take_bits(buf, &bp, 7), main_pitch = take_bits(buf, &bp, 6);
    int modulation = take_bits(buf, &bp, 3), corr_id = take_bits(buf, &bp, 2);
    int e0 = take_bits(buf, &bp, 10), e1 = take_bits(buf, &bp, 10), e2 = take_bits(buf, &bp, 10);
    int vq_mid = take_bits(buf, &bp, 13), interp_id = take_bits(buf, &bp, 3);
    float frame_corr, sign = 1;
    float (*f)[36] = (float (*)[36])F;
    memset(F, 0, 4 * 36 * sizeof(float));
    modulation -= 4;
    if (modulation == -4) { voiced = 0; modulation = 0; }
    frame_corr = voiced ? 0.3875f + .175f * corr_id : 0.0375f + .075f * corr_id;
    for (sub = 0; sub < 4; sub++) {
        float p = (float)(pow(2.f, main_pitch / 21.) * 32);
        p *= 1.f + modulation / 16.f / 7.f * (2 * sub - 3);
        p = 255 < (33 > p ? 33 : p) ? 255 : (33 > p ? 33 : p);
        f[sub][NBANDS] = .02f * (p - 100.f);
        f[sub][NBANDS + 1] = frame_corr - .5f;
    }
    f[3][0] = (c0_id - 64) / 4.f;
    for (i = 0; i < NBANDS - 1; i++)
        f[3][i + 1] = cb1[e0 * 17 + i] + cb2[e1 * 17 + i] + cb3[e2 * 17 + i];
    if (vq_mid >= 4096) { vq_mid -= 4096; sign = -1; }
    for (i = 0; i < NBANDS; i++) f[1][i] = sign * cbd[vq_mid * NBANDS + i];
    if ((vq_mid & 3) < 2)       for (i = 0; i < NBANDS; i++) f[1][i] += .5f * (vq_mem[i] + f[3][i]);
    else if ((vq_mid & 3) == 2) for (i = 0; i < NBANDS; i++) f[1][i] += vq_mem[i];
    else                        for (i = 0; i < NBANDS; i++) f[1][i] += f[3][i];
    interp_id += (interp_id >= 7);
    interp1(f[0], vq_mem, f[1], interp_id / 3);
    interp1(f[2], f[1], f[3], interp_id % 3);
    memcpy(vq_mem, f[3], NBANDS * sizeof(float));
}
