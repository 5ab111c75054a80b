/* TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of the LPCNet synthesis hot path (reference: /root/reference, xiph/LPCNet
 * tag 2024_10_08), used as the checker for the HIP engine.  It follows the reference's
 * *generic-C* arithmetic (src/vec.h:42-409: table tanh, sequential non-FMA accumulation), the
 * only flavour of the reference that is bit-reproducible across compilers.
 *
 * Pinning: the reference holds no golden vectors for this path ("parity unpinned" by its own
 * tests).  This restatement is therefore pinned against the reference itself,
 * compiled from its own sources by oracle/Makefile into oracle/_ref (flavours gf/gi), and
 * against fixtures generated from those builds (tests/golden/, tests/tools/make_golden.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this code.
 * It must be compiled with -ffp-contract=off (the Makefile does).
 */
#ifndef LPCNET_ORACLE_H
#define LPCNET_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_N_A        384
#define ORC_N_B        16
#define ORC_COND       128
#define ORC_NB_FEAT    20
#define ORC_PITCH_EMB  64
#define ORC_FRAME_IN   (ORC_NB_FEAT + ORC_PITCH_EMB)
#define ORC_LPC_ORDER  16
#define ORC_NB_BANDS   18
#define ORC_FEATURES_DELAY 2

typedef struct orc_model orc_model;
typedef struct orc_state orc_state;

/* blob = "DNNw" weight blob (src/nnet.h:54-61).  The blob must outlive the model (the
 * reference keeps pointers into it too, src/parse_lpcnet_weights.c:46).  Returns NULL if a
 * record is malformed or an array is missing / mis-sized (src/parse_lpcnet_weights.c:124-220).
 * int8 (DOT_PROD) vs float flavour is inferred from the size of the qweight arrays. */
orc_model *orc_model_parse(const unsigned char *blob, int len, float lpc_gamma);
void orc_model_set_end2end(orc_model *m, int on);   /* END2END of the generated nnet_data.h: LPC from reflection coefficients */
void orc_model_free(orc_model *m);
int  orc_model_is_int8(const orc_model *m);
int  orc_model_nb_blocks(const orc_model *m, int which /*0 = GRU-A, 1 = GRU-B*/);

orc_state *orc_state_create(const orc_model *m);
void orc_state_destroy(orc_state *st);
void orc_state_reset(orc_state *st);                       /* src/lpcnet.c:174-182 */

/* src/lpcnet.c:82-120; results are also stored in the state like lpcnet_synthesize_impl does */
void orc_frame_network(orc_state *st, const float *features, float *cond_a, float *cond_b, float *lpc);
/* src/lpcnet.c:146-167 */
int  orc_sample_network(orc_state *st, const float *cond_a, const float *cond_b,
                        int last_exc, int last_sig, int pred);
/* src/lpcnet.c:235-281 (preload>0 = teacher forcing, :256-259) */
void orc_synthesize(orc_state *st, const float *features, short *output, int N, int preload);
/* tail only, with frame products supplied by the caller (parity seam for the sample loop) */
void orc_synthesize_tail(orc_state *st, const float *cond_a, const float *cond_b, const float *lpc,
                         short *output, int N, int preload);

/* state access for tests */
void orc_get_nnet_state(const orc_state *st, float *conv1, float *conv2, float *gru_a, float *gru_b);
void orc_set_gru_state(orc_state *st, const float *gru_a, const float *gru_b);
void orc_get_frame_products(const orc_state *st, float *lpc, float *cond_a, float *cond_b);
void orc_get_signal_state(const orc_state *st, float *last_sig, int *last_exc, float *deemph_mem,
                          int *frame_count, unsigned *rng4);
void orc_force_frame_count(orc_state *st, int frame_count);

/* layer-level hooks */
void orc_gru_a_input(const orc_model *m, float *out, const float *cond, int sig, int pred, int exc);
void orc_sparse_gru_a(const orc_model *m, float *state, const float *input);
void orc_gru_b(const orc_model *m, const float *cond_b, float *state, const float *input);
int  orc_sample_mdense(const orc_model *m, const float *input, unsigned *rng4);
/* the engine's fp16 dual-FC sub-option restated (no reference counterpart): logits[8] of the nodes on the 8-bit `path`;
 * variant 0 / 1 = the two possible rounding orders of a v_dot2_f32_f16 step (see lpcnet_oracle.c) */
float orc_f16(float x);                                     /* float -> binary16 -> float, round to nearest even */
void orc_mdense_f16_path(const orc_model *m, const float *input, int path, int variant, float *logits);

/* scalar helpers */
int   orc_lin2ulaw(float x);                                /* src/common.h:47-58 */
float orc_ulaw2lin(int code);                               /* src/common.h:37-45 */
float orc_tanh_approx(float x);                             /* src/vec.h:82-99    */
float orc_sigmoid_approx(float x);                          /* src/vec.h:101-104  */
void  orc_kiss99_srand(unsigned *ctx4, const unsigned char *data, int n);   /* src/kiss99.c:34-57 */
unsigned orc_kiss99_rand(unsigned *ctx4);                   /* src/kiss99.c:59-81 */
void  orc_lpc_from_cepstrum(float *lpc, const float *cepstrum);             /* src/freq.c:310-320 */
void  orc_fft320(const float *in_ri, float *out_ri);        /* src/kiss_fft.c:566-586, float build */
float orc_logit_table(int i);                               /* src/lpcnet.c:188-191 */

/* codec front-end (src/lpcnet_dec.c:81-155) with caller-supplied VQ codebooks */
void orc_decode_packet(float *features4x36, float *vq_mem, const unsigned char *buf8,
                       const float *cb1, const float *cb2, const float *cb3, const float *cb_diff4);

#ifdef __cplusplus
}
#endif
#endif
