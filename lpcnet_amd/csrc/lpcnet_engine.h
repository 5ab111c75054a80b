/* The one header that crosses from the C host shell into the HIP device runtime.
 * Plain C ABI: pointers, sizes and ints only.
 *
 * Layers:
 *   api.c (C)          public include/lpcnet.h + include/lpcnet_batch.h entry points
 *   model_pack.c (C)   DNNw blob -> validated host model + device-friendly packings
 *   engine.hip (HIP)   device memory, uploads, kernel launches  <-- declared here
 */
#ifndef LPCNET_ENGINE_H
#define LPCNET_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- fixed architecture of the default model (what the reference's generated nnet_data.h
 *      hard-codes) ---------------------------------------------------- */
#define LPCN_N_A        384
#define LPCN_N_B        16
#define LPCN_COND       128
#define LPCN_NB_FEAT    20
#define LPCN_PITCH_EMB  64
#define LPCN_FRAME_IN   (LPCN_NB_FEAT + LPCN_PITCH_EMB)
#define LPCN_LPC_ORDER  16
#define LPCN_NB_BANDS   18
#define LPCN_FRAME_SIZE 160
#define LPCN_FEATURES_DELAY 2
#define LPCN_ROWS_A     (3 * LPCN_N_A)
#define LPCN_ROWS_B     (3 * LPCN_N_B)

/* ---- sample-kernel geometry --------------------------------------------------------------
 * One workgroup = 8 wavefronts = 512 lanes runs S interleaved streams.  GRU-A's recurrent rows
 * are dealt to lanes in "slots" of 64 rows (8 row-groups of the 8x4 block structure); each wave
 * owns up to LPCN_MAX_SLOTS slots and keeps their weights resident in VGPRs as `nw` items of
 * float4 (one item = the lane's row x one 4-wide input block). */
#define LPCN_WG_THREADS 512
#define LPCN_WAVES      8
#define LPCN_MAX_SLOTS  3
#define LPCN_EARLY_MAX  24     /* most items of a candidate slot that waves 4..7 may compute one sample ahead (float blobs) */
#define LPCN_DEAL_EH_F32 24    /* float blobs: head length of the early candidate items (round 4, matrix-pipe items: 20 / 22 / 24 -> 121.3 / 123.5 / 123.9 M samples/s) */
#define LPCN_DEAL_HW_I8 3      /* int8 blobs: first wave that may carry a candidate head (model_pack.c; 4 / 3 / 2 -> 153.4 / 156.2 / 154.1 M samples/s; round 5: superseded by the mask below) */
/* int8 blobs, PARITY, <= 2 streams per workgroup (two workgroups per CU -- the operating point of a full GPU): GRU-B's chain waves are
 * LPCN_I8_GBWA (stream 0) and LPCN_I8_GBWB (stream 1), NOT waves 0 and 1.  Waves w and w + 4 share a SIMD and wave 0 also leads the
 * streams (tree walk, LPC, mu-law: ~380 VALU instructions per sample that no other wave has); with GRU-B (~410 per stream) on waves
 * 0 / 1 as well, the SIMD of waves 0 / 4 issues 1.9 k VALU instructions per step against 1.2 k on the SIMDs of waves 2 / 6 and 3 / 7
 * (tools/valu_census.py --int8), and two co-resident workgroups are bound by the busiest SIMD (tools/ubench/hwid.hip: the second
 * workgroup's wave w lands one SIMD further than the first one's).  Every other wave is idle in GRU-B's phase and carries a
 * candidate head (LPCN_DEAL_HMASK_I8: bit w = wave w may), LPCN_DEAL_EH_I8 items long.  Measured (1024 streams, bit-exact):
 * waves 0,1 / 2,3 / 1,3 / 1,2 -> 157.6 / 163.3 / 160.7 / 160.1 M samples/s; head 10 / 14 / 16 / 18 / 20 / 22 / 24 items with 2,3 ->
 * 157.3 / 163.3 / 166.1 / 167.9 / 164.6 / 169.0 / 168.3 M (gpurun_out of round 5, two runs each within 0.2 M). */
#define LPCN_I8_GBWA 2
#define LPCN_I8_GBWB 3
#define LPCN_DEAL_HMASK_I8 (0xFFu & ~((1u << LPCN_I8_GBWA) | (1u << LPCN_I8_GBWB)))
#define LPCN_DEAL_EH_I8 22
/* the two-group float kernel (sample_kernel_x2.hip.h): GRU-B's chains -- one stream each -- run on waves 0 .. LPCN_X2_CHAIN_WAVES - 1; waves LPCN_X2_P0_FIRST .. 7
 * carry one candidate slot each (its head runs in the chains' shadow) and run the start-value pass, wave LPCN_X2_P0_FIRST also leads the streams; at most
 * LPCN_X2_NW_MAX register-resident items per lane */
#define LPCN_X2_CHAIN_WAVES 4
#define LPCN_X2_P0_FIRST 4
#define LPCN_X2_NW_MAX 32
#define LPCN_DEAL_EH_FAST_I8 0 /* head length of the FAST arithmetic's own image of int8 blobs (model_pack.c: lpcn_model_pack_fast) */

typedef struct lpcn_model_host {
    int is_int8;                 /* blob flavour: 0 = float qweights (DISABLE_DOT_PROD), 1 = int8 (DOT_PROD)  */
    int b_dense;                 /* GRU-B input matrix lists every input block for every row group */
    int nb_a, nb_b;              /* 8x4 blocks in GRU-A recurrent / GRU-B input matrices       */
    int nw;                      /* items per lane needed by the register-resident packing     */
    int nb_b_padded;             /* GRU-B blocks after padding every row group to a multiple of 4 */
    float lpc_gamma;

    /* plain arrays (pointers into the caller's blob, reference layouts) */
    const float *emb_sig, *emb_pred, *emb_exc;      /* [256][1152]                             */
    const float *a_dense_w, *a_dense_b;             /* [128][1152], [1152]                     */
    const float *b_dense_w, *b_dense_b;             /* [128][48], [48]                         */
    const float *conv1_w, *conv1_b, *conv2_w, *conv2_b;
    const float *pitch_emb;                         /* [256][64]                               */
    const float *dense1_w, *dense1_b, *dense2_w, *dense2_b;
    const float *fc_w, *fc_b, *fc_f;                /* [256][2][16], [2][256], [2][256]        */
    const float *a_bias, *a_diag;                   /* [2][1152], [1152]                       */
    const float *a_w; const int *a_idx;             /* float blocks [in4][out8] + index stream */
    const float *b_bias;                            /* [2][48]                                 */
    const float *b_w; const int *b_idx;
    const float *b_rec;                             /* [16][48]                                */

    /* device-oriented packings built by lpcn_model_pack() (malloc'ed, owned by the model)     */
    float   *pk_a_w;      /* [8 waves][nw][64 lanes][4]     row weights per item (float blobs)  */
    int32_t *pk_a_wq;     /* [8 waves][nw][64 lanes]        4 int8 row weights per item (int8)  */
    int32_t *pk_b_wq;     /* [blk/4][row-in-group 8][blk%4] 4 int8 per row and block (int8)     */
    uint8_t *pk_a_blk;    /* [8][nw][64]                    input block index p = pos/4        */
    int32_t *pk_a_row;    /* [8][3 slots][64]               GRU-A row (0..1151) or -1          */
    int32_t  pk_a_bound[LPCN_WAVES][LPCN_MAX_SLOTS + 1];    /* item index where each slot starts */
    int32_t  pk_a_allh[LPCN_WAVES][LPCN_MAX_SLOTS];        /* 1 if every live row of the slot is a candidate-state row */
    int32_t  pk_a_head[LPCN_WAVES];   /* float blobs: chain items [0, head) of slot 0's rows sit at items [nw - head, nw) (computed one sample ahead) */
    float   *pk_b_w;      /* GRU-B input weights re-blocked [blk][row-in-group 8][k 4] per group */
    int32_t *pk_b_start;  /* [6 groups + 1] first block of each group in pk_b_w                */
    uint8_t *pk_b_blk;    /* [nb_b] input block index per block                                */
    float   *pk_emb[3];   /* sig/pred/exc tables re-ordered to [256][3 slots][512 threads] */
} lpcn_model_host;

/* model_pack.c -------------------------------------------------------------------------------
 * Parse + validate a DNNw blob exactly like the reference loader does
 * (src/parse_lpcnet_weights.c:53-113, :124-220) and build the packings.  Returns 0, or -1 on a
 * malformed / incomplete blob (same condition under which lpcnet_load_model returns -1). */
int  lpcn_model_parse(lpcn_model_host *m, const unsigned char *blob, int len);
void lpcn_model_release(lpcn_model_host *m);
int  lpcn_model_pack_x2(const lpcn_model_host *m, lpcn_model_host *f);     /* 0: f holds the two-group kernel's own GRU-A packing; -1: the model does not fit it */
int  lpcn_model_pack_fast(const lpcn_model_host *m, lpcn_model_host *f);   /* 1: FAST shares PARITY's image; 0: f holds FAST's own GRU-A packing */
/* re-expand the packings and compare them with the blob (0 = consistent) */
int  lpcn_model_selftest(const lpcn_model_host *m);

/* ---- per-stream state as the device keeps it (AoS, one record per stream) ------------------ */
typedef struct lpcn_stream_state {
    float gru_a[LPCN_N_A];
    float gru_b[LPCN_N_B];
    float conv1_mem[2 * LPCN_FRAME_IN];
    float conv2_mem[2 * LPCN_COND];
    float old_lpc[LPCN_FEATURES_DELAY][LPCN_LPC_ORDER];
    float last_sig[LPCN_LPC_ORDER];
    float deemph_mem;
    int32_t last_exc;
    int32_t frame_count;
    uint32_t rng[4];
    float lpc[LPCN_LPC_ORDER];          /* products of the most recent frame (for export)       */
    int32_t pad[3];
} lpcn_stream_state;

/* ---- engine.hip ----------------------------------------------------------------------------- */
typedef struct lpcn_engine lpcn_engine;      /* one per (process, HIP device, model)            */

/* All functions return 0 on success or a negative LPCN_E_* code; the message of the last failure
 * on the calling thread is available from lpcn_last_error(). */
#define LPCN_E_NODEVICE  (-2)
#define LPCN_E_HIP       (-3)
#define LPCN_E_ARG       (-4)
#define LPCN_E_MODEL     (-5)
const char *lpcn_last_error(void);

int  lpcn_engine_create(lpcn_engine **out, int device, const lpcn_model_host *m);
void lpcn_engine_destroy(lpcn_engine *e);
int  lpcn_engine_device(const lpcn_engine *e);

/* Device buffers for a set of n streams processed together. */
typedef struct lpcn_batch_dev lpcn_batch_dev;
int  lpcn_batch_dev_create(lpcn_batch_dev **out, lpcn_engine *e, int n_streams, int max_chunk_frames);
void lpcn_batch_dev_destroy(lpcn_batch_dev *b);
int  lpcn_batch_dev_reset(lpcn_batch_dev *b, int first, int count);           /* lpcnet_reset   */
int  lpcn_batch_dev_get_state(lpcn_batch_dev *b, int stream, lpcn_stream_state *host);
/* k independent streams with their callers' POD states in ONE pass and one synchronisation (engine.hip): frame network + samples,
 * samples from the callers' frame products, or the frame network alone */
#define LPCN_GROUP_FRAME_SAMPLES 0
#define LPCN_GROUP_TAIL          1
#define LPCN_GROUP_FRAMES        2
int  lpcn_batch_dev_run_group(lpcn_batch_dev *b, int k, int kind, int frame_len, int preload, const lpcn_stream_state *const *st_in,
                              const float *const *feat, short *const *pcm, lpcn_stream_state *const *st_out,
                              float *const *ga, float *const *gb, float *const *lpc);
int  lpcn_batch_dev_tune(lpcn_batch_dev *b);              /* measure the streams per workgroup now, on the engine's own stream */
int  lpcn_batch_dev_set_state(lpcn_batch_dev *b, int stream, const lpcn_stream_state *host);
int  lpcn_batch_dev_streams_per_wg(const lpcn_batch_dev *b);
int  lpcn_batch_dev_set_streams_per_wg(lpcn_batch_dev *b, int s);              /* 1,2,4 (0=auto) */
int  lpcn_batch_dev_retune(lpcn_batch_dev *b);                                 /* after lpcn_engine_set_fast: auto value again */

/* Run n_frames frames (frame network + LPC + 160-sample loop each) for every stream.
 *   features : [n_streams][n_frames][feat_stride] floats, only [0..19] of each frame are read
 *   pcm      : [n_streams][n_frames*160] int16
 *   preload  : teacher forcing, 0..160 leading samples of every frame are taken from pcm
 *              (semantics of lpcnet_synthesize_impl, src/lpcnet.c:256-259)
 * *_dev variants take device pointers and only enqueue work on `hip_stream` (NULL = the
 * engine's own stream); the host variant copies in/out and synchronises. */
int  lpcn_batch_dev_run(lpcn_batch_dev *b, const float *d_features, int feat_stride,
                        short *d_pcm, int n_frames, int preload, void *hip_stream);
int  lpcn_batch_dev_run_host(lpcn_batch_dev *b, const float *features, int feat_stride,
                             short *pcm, int n_frames, int preload);
int  lpcn_batch_dev_sync(lpcn_batch_dev *b);
/* Legacy per-frame API on a batch of ONE stream: one frame-network step on feat[0..19] + frame_len samples with a single
 * host synchronisation.  st_in == NULL: the device copy of the state is current (no upload); st_out receives the new state. */
int  lpcn_batch_dev_run_single(lpcn_batch_dev *b, const lpcn_stream_state *st_in, const float *feat, short *pcm,
                               lpcn_stream_state *st_out);

/* Codec path (src/lpcnet_dec.c:81-155 on the device): packets [n][n_packets][8] -> pcm [n][n_packets*640].
 * The VQ memory of every stream lives on the device and is cleared by lpcn_batch_dev_reset. */
int  lpcn_engine_set_codebooks(lpcn_engine *e, const float *cb1, const float *cb2, const float *cb3, const float *cb_diff4);
int  lpcn_engine_has_codebooks(const lpcn_engine *e);
int  lpcn_engine_set_end2end(lpcn_engine *e, int on);              /* END2END of the model's nnet_data.h (default off) */
int  lpcn_engine_set_lpc_gamma(lpcn_engine *e, float gamma);     /* LPC_GAMMA of the model's nnet_data.h (default 1) */
int  lpcn_engine_set_fast(lpcn_engine *e, int on);                /* FAST arithmetic (not bit-exact; default off = PARITY) */
int  lpcn_batch_dev_decode(lpcn_batch_dev *b, const unsigned char *d_packets, short *d_pcm, int n_packets, void *hip_stream);
int  lpcn_batch_dev_decode_host(lpcn_batch_dev *b, const unsigned char *packets, short *pcm, int n_packets);

/* Parity seam (SURVEY.md §7 hard part 9): run only the sample loop with caller-provided frame
 * products (host pointers): cond_a [n][f][1152], cond_b [n][f][48], lpc [n][f][16]. */
int  lpcn_batch_dev_run_tail_host(lpcn_batch_dev *b, const float *cond_a, const float *cond_b,
                                  const float *lpc, short *pcm, int n_frames, int preload);
/* Frame network + LPC only (host pointers); outputs like above (any may be NULL). */
int  lpcn_batch_dev_run_frames_host(lpcn_batch_dev *b, const float *features, int feat_stride,
                                    float *cond_a, float *cond_b, float *lpc, int n_frames);

/* One frame step with per-stream arguments (host pointers, see engine.hip): mode[s] 0 = skip, 1 = frame network + n_samples[s]
 * samples, 2 = n_samples[s] samples from the stream's most recent frame products; preload[s] leading samples of pcm[s] imposed.
 * features [n][feat_stride], pcm [n][160]. */
int  lpcn_batch_dev_step_host(lpcn_batch_dev *b, const float *features, int feat_stride, short *pcm,
                              const int *n_samples, const int *preload, const int *mode);

/* Timing of the most recent run: kernel-only milliseconds measured with HIP events on the
 * stream the kernels were launched on (sample kernel, frame kernels). */
int  lpcn_batch_dev_last_timing(lpcn_batch_dev *b, float *ms_sample, float *ms_frame);
int  lpcn_batch_dev_enable_timing(lpcn_batch_dev *b, int on);
/* samples produced per frame (default 160); lpcnet_synthesize(st, f, out, N) maps to N here */
int  lpcn_batch_dev_set_frame_len(lpcn_batch_dev *b, int n);
/* tests only: per-sample trace of workgroup 0 / stream 0 (host_out == NULL: (re)allocate n_samples records) */
int  lpcn_batch_dev_debug_trace(lpcn_batch_dev *b, int n_samples, float *host_out);
/* per-phase shader-clock totals of workgroup 0 (out == NULL: enable and zero; else fetch 8 values) */
int  lpcn_batch_dev_profile(lpcn_batch_dev *b, unsigned long long *out);

/* test seam: this engine's own 10^x (lpcnet_exp10.h) evaluated on the device for host arrays */
int  lpcn_debug_exp10(int device, const float *x, double *out, size_t n);
/* test seam: v_mfma_f32_4x4x1 with C = -0.0 against v_mul_f32, and the halves of v_pk_mul_f32 / v_pk_add_f32 against the scalar
 * instructions, as bit patterns (engine.hip: lpcn_arith_identity_kernel); n = operand count, a multiple of 64; outputs [n][4] each */
/* test seam: v_cvt_rpi_i32_f32 against (int)floor(.5 + (double)t) on all 2^32 bit patterns (engine.hip: lpcn_quant_sweep_kernel);
 * out3 = {mismatches among finite |t| < 2^31, mismatches with |t| <= 127.5, one mismatching pattern} */
int  lpcn_debug_quant_sweep(int device, unsigned long long *out3);
int  lpcn_debug_arith_identities(int device, const float *a, const float *b, uint32_t *out_mfma, uint32_t *out_mul,
                                 uint32_t *out_pk, uint32_t *out_sc, size_t n);

#ifdef __cplusplus
}
#endif
#endif
