/* TEST INFRASTRUCTURE ONLY -- never linked into the product library.
 *
 * Thin accessor layer compiled *together with* the unmodified reference sources
 * (/root/reference/src, read-only) into oracle/_ref/liblpcnet_ref_<flavour>.so so that Python
 * tests can reach the reference's internal functions and its private LPCNetState without
 * hard-coding struct offsets.  Everything here only forwards to reference code; no algorithm
 * is restated in this file (the restatement lives in oracle/lpcnet_oracle.c).
 */
#include <string.h>
#include "lpcnet_private.h"   /* reference: struct LPCNetState (src/lpcnet_private.h:28-48) */
#include "nnet.h"
#include "common.h"
#include "freq.h"

extern float ceps_codebook1[], ceps_codebook2[], ceps_codebook3[], ceps_codebook_diff4[];

int ref_flavour(void)
{
    int f = 0;
#ifdef DOT_PROD
    f |= 1;            /* int8 weights */
#endif
#ifdef USE_SU_BIAS
    f |= 2;            /* unsigned activations + subias (AVX2 int8) */
#endif
#ifdef NO_OPTIMIZATIONS
    f |= 4;            /* generic C kernels (vec.h:42-409) */
#endif
    return f;
}

void ref_set_codebooks(const float *c1, const float *c2, const float *c3, const float *d4)
{
    memcpy(ceps_codebook1, c1, sizeof(float)*1024*17);
    memcpy(ceps_codebook2, c2, sizeof(float)*1024*17);
    memcpy(ceps_codebook3, c3, sizeof(float)*1024*17);
    memcpy(ceps_codebook_diff4, d4, sizeof(float)*4096*18);
}

/* ---- state access ------------------------------------------------------------------ */
void ref_get_nnet_state(const LPCNetState *st, float *conv1, float *conv2, float *gru_a, float *gru_b)
{
    if (conv1) memcpy(conv1, st->nnet.feature_conv1_state, sizeof(st->nnet.feature_conv1_state));
    if (conv2) memcpy(conv2, st->nnet.feature_conv2_state, sizeof(st->nnet.feature_conv2_state));
    if (gru_a) memcpy(gru_a, st->nnet.gru_a_state, sizeof(st->nnet.gru_a_state));
    if (gru_b) memcpy(gru_b, st->nnet.gru_b_state, sizeof(st->nnet.gru_b_state));
}

void ref_set_gru_state(LPCNetState *st, const float *gru_a, const float *gru_b)
{
    if (gru_a) memcpy(st->nnet.gru_a_state, gru_a, sizeof(st->nnet.gru_a_state));
    if (gru_b) memcpy(st->nnet.gru_b_state, gru_b, sizeof(st->nnet.gru_b_state));
}

/* frame-side products as left in the state by run_frame_network (src/lpcnet.c:275) */
void ref_get_frame_products(const LPCNetState *st, float *lpc, float *cond_a, float *cond_b)
{
    memcpy(lpc, st->lpc, sizeof(st->lpc));
    memcpy(cond_a, st->gru_a_condition, sizeof(st->gru_a_condition));
    memcpy(cond_b, st->gru_b_condition, sizeof(st->gru_b_condition));
}

void ref_get_signal_state(const LPCNetState *st, float *last_sig, int *last_exc, float *deemph_mem,
                          int *frame_count, unsigned *rng4)
{
    memcpy(last_sig, st->last_sig, sizeof(st->last_sig));
    *last_exc = st->last_exc;
    *deemph_mem = st->deemph_mem;
    *frame_count = st->frame_count;
    rng4[0] = st->rng.z; rng4[1] = st->rng.w; rng4[2] = st->rng.jsr; rng4[3] = st->rng.jcong;
}

void ref_get_logit_table(const LPCNetState *st, float *table256)
{
    memcpy(table256, st->sampling_logit_table, sizeof(st->sampling_logit_table));
}

/* ---- entry points with internal linkage-visible names -------------------------------- */
void ref_synthesize_impl(LPCNetState *st, const float *features, short *output, int N, int preload)
{
    lpcnet_synthesize_impl(st, features, output, N, preload);   /* src/lpcnet.c:273 */
}

void ref_run_frame_network(LPCNetState *st, const float *features, float *cond_a, float *cond_b, float *lpc)
{
    run_frame_network(st, cond_a, cond_b, lpc, features);        /* src/lpcnet.c:82 */
}

int ref_run_sample_network(LPCNetState *st, const float *cond_a, const float *cond_b,
                           int last_exc, int last_sig, int pred)
{
    return run_sample_network(st, cond_a, cond_b, last_exc, last_sig, pred,
                              st->sampling_logit_table, &st->rng);  /* src/lpcnet.c:146 */
}

/* ---- layer-level known-answer hooks (use the model bound inside `st`) ----------------- */
void ref_gru_a_input(const LPCNetState *st, float *out, const float *cond, int sig, int pred, int exc)
{
    compute_gru_a_input(out, cond, GRU_A_STATE_SIZE, &st->model.gru_a_embed_sig, sig,
                        &st->model.gru_a_embed_pred, pred, &st->model.gru_a_embed_exc, exc);
}
void ref_sparse_gru_a(const LPCNetState *st, float *state, const float *input)
{
    compute_sparse_gru(&st->model.sparse_gru_a, state, input);
}
void ref_gru_b(const LPCNetState *st, const float *cond_b, float *state, const float *input)
{
    compute_gruB(&st->model.gru_b, cond_b, state, input);
}
int ref_sample_mdense(const LPCNetState *st, const float *input, unsigned *rng4)
{
    kiss99_ctx r; int v;
    r.z = rng4[0]; r.w = rng4[1]; r.jsr = rng4[2]; r.jcong = rng4[3];
    v = sample_mdense(&st->model.dual_fc, input, st->sampling_logit_table, &r);
    rng4[0] = r.z; rng4[1] = r.w; rng4[2] = r.jsr; rng4[3] = r.jcong;
    return v;
}
void ref_mdense_logits(const LPCNetState *st, float *out256, const float *input)
{
    compute_mdense(&st->model.dual_fc, out256, input);   /* SOFTMAX_HACK: sigmoid'ed outputs */
}

/* ---- scalar helpers ------------------------------------------------------------------ */
int   ref_lin2ulaw(float x)       { return lin2ulaw(x); }
float ref_ulaw2lin(float u)       { return ulaw2lin(u); }
float ref_tanh_approx(float x)    { float y; vec_tanh(&y, &x, 1); return y; }
float ref_sigmoid_approx(float x) { float y; vec_sigmoid(&y, &x, 1); return y; }
void  ref_vec_tanh(float *y, const float *x, int n)    { vec_tanh(y, x, n); }
void  ref_vec_sigmoid(float *y, const float *x, int n) { vec_sigmoid(y, x, n); }
float ref_lpc_from_cepstrum(float *lpc, const float *ceps) { return lpc_from_cepstrum(lpc, ceps); }
void  ref_lpc_weighting(float *lpc, float gamma) { lpc_weighting(lpc, gamma); }
void  ref_decode_packet(float *features4x36, float *vq_mem, const unsigned char *buf)
{
    decode_packet((float (*)[NB_TOTAL_FEATURES])features4x36, vq_mem, buf);
}
