/* lpcnet.h -- public C API of the MI355X-native LPCNet synthesis engine.
 *
 * Drop-in for the synthesis + decoder subset of xiph/LPCNet's include/lpcnet.h (tag 2024_10_08):
 * same names, signatures, argument meaning and return conventions, so a program written against
 * the reference header (e.g. src/lpcnet_demo.c -synthesis / -decode) links against
 * liblpcnet_hip.so unchanged.  Each declaration cites the reference declaration it replaces.
 * The encoder / feature-extraction / PLC entry points of the reference header are outside this
 * engine's scope and are not exported.
 *
 * Differences a caller can observe:
 *   - The trained model is never compiled in.  A state uses the blob given to lpcnet_load_model()
 *     (the reference's -DUSE_WEIGHTS_FILE flow, src/lpcnet.c:192-196, src/lpcnet_demo.c:205-207); a
 *     state without one -- lpcnet_create() alone, or an LPCNetDecState, which the reference can only
 *     run on its compiled-in model (src/lpcnet_demo.c:176-188) -- uses the process-default model:
 *     lpcnet_hip_set_default_model(), else the file $LPCNET_HIP_MODEL, else ./weights_blob.bin.
 *     The decoder's VQ codebooks (the reference's generated ceps_codebooks.c) come from
 *     lpcnet_hip_set_codebooks(), else $LPCNET_HIP_CODEBOOKS, else ./ceps_codebooks.bin.
 *   - Arithmetic follows the reference's generic-C path bit for bit (not its AVX2/NEON
 *     approximations): float ("DISABLE_DOT_PROD") blobs like its generic float build, int8
 *     ("DOT_PROD") blobs like its generic int8 build; the flavour is detected from the blob.
 *   - All compute runs on a HIP device; there is no CPU fallback.  If no device is usable,
 *     lpcnet_load_model() returns -1 and lpcnet_hip_last_error() says why.  Entry points that return
 *     void in the reference (lpcnet_synthesize, run_frame_network, lpcnet_synthesize_impl/_tail_impl)
 *     never abort: a call that fails -- device error, no model and no default model, evicted model --
 *     zero-fills its output (silence / zero frame products), leaves the caller's state untouched and
 *     records the failure in a sticky per-thread status (lpcnet_hip_status(), lpcnet_hip_last_error())
 *     and a sticky per-model status (lpcnet_hip_model_status()); lpcnet_decode returns -1.
 *     LPCNET_HIP_ABORT_ON_ERROR=1 in the environment turns such a failure into abort() with the message.
 */
#ifndef LPCNET_H_
#define LPCNET_H_

#ifdef __cplusplus
extern "C" {
#endif

#ifndef LPCNET_EXPORT
# if defined(__GNUC__)
#  define LPCNET_EXPORT __attribute__ ((visibility ("default")))
# else
#  define LPCNET_EXPORT
# endif
#endif

#define NB_FEATURES 20                      /* reference include/lpcnet.h:45 */
#define NB_TOTAL_FEATURES 36                /* :46 */
#define LPCNET_COMPRESSED_SIZE 8            /* :49 bytes per packet */
#define LPCNET_PACKET_SAMPLES (4*160)       /* :51 */
#define LPCNET_FRAME_SIZE (160)             /* :53 */

typedef struct LPCNetState LPCNetState;         /* :55 */
typedef struct LPCNetDecState LPCNetDecState;   /* :57 */

/* ---- decoder (reference include/lpcnet.h:67-96) ------------------------------------------ */
LPCNET_EXPORT int lpcnet_decoder_get_size(void);                                   /* :67 */
LPCNET_EXPORT int lpcnet_decoder_init(LPCNetDecState *st);                         /* :76, returns 0 */
LPCNET_EXPORT LPCNetDecState *lpcnet_decoder_create(void);                         /* :83 */
LPCNET_EXPORT void lpcnet_decoder_destroy(LPCNetDecState *st);                     /* :88 */
/* 8 bytes in -> 640 samples out; returns 0 (:96, src/lpcnet.c:310-319) */
LPCNET_EXPORT int lpcnet_decode(LPCNetDecState *st, const unsigned char *buf, short *pcm);

/* ---- synthesis (reference include/lpcnet.h:78, :160-188, :214) ---------------------------- */
LPCNET_EXPORT void lpcnet_reset(LPCNetState *lpcnet);                              /* :78 */
LPCNET_EXPORT int lpcnet_get_size(void);                                           /* :160 */
LPCNET_EXPORT int lpcnet_init(LPCNetState *st);                                    /* :169, returns 0 */
LPCNET_EXPORT LPCNetState *lpcnet_create(void);                                    /* :174 */
LPCNET_EXPORT void lpcnet_destroy(LPCNetState *st);                                /* :179 */
/* one frame-network step on features[0..19] followed by N output samples (:188; any N > 0 like the reference,
 * N <= 160 is the single-launch fast path) */
LPCNET_EXPORT void lpcnet_synthesize(LPCNetState *st, const float *features, short *output, int N);
/* bind a "DNNw" weight blob; 0 on success, -1 on a malformed blob or missing device (:214).
 * Unlike the reference the blob is copied to the device and need not outlive the state. */
LPCNET_EXPORT int lpcnet_load_model(LPCNetState *st, const unsigned char *data, int len);

/* sizeof(LPCNetState) of this engine (== lpcnet_get_size()).  The state is a relocatable POD without pointers; a
 * caller that embeds it by value (struct LPCNetPLCState, src/lpcnet_private.h:86) can declare
 *     struct LPCNetState { _Alignas(4) unsigned char opaque[LPCNET_HIP_STATE_BYTES]; };                              */
#define LPCNET_HIP_STATE_BYTES 8724

/* ---- internal entry points of the reference (src/lpcnet_private.h:125-132).  src/lpcnet_plc.c links to them:
 * exporting them with the same names and meaning lets the unmodified PLC drive this engine.
 * LPCNetState holds the frame products between run_frame_network and lpcnet_synthesize_tail_impl, as in the
 * reference, and may be copied by value (snapshot / rollback, src/lpcnet_plc.c:216-231). */
LPCNET_EXPORT void lpcnet_reset_signal(LPCNetState *lpcnet);                                                  /* src/lpcnet.c:226-233 */
LPCNET_EXPORT void run_frame_network(LPCNetState *lpcnet, float *gru_a_condition, float *gru_b_condition,
                                     float *lpc, const float *features);                                      /* src/lpcnet.c:82-120 */
LPCNET_EXPORT void run_frame_network_deferred(LPCNetState *lpcnet, const float *features);                    /* src/lpcnet.c:122-132 */
LPCNET_EXPORT void run_frame_network_flush(LPCNetState *lpcnet);                                              /* src/lpcnet.c:134-144 */
LPCNET_EXPORT void lpcnet_synthesize_tail_impl(LPCNetState *lpcnet, short *output, int N, int preload);      /* src/lpcnet.c:235-271 */
LPCNET_EXPORT void lpcnet_synthesize_impl(LPCNetState *lpcnet, const float *features, short *output, int N,
                                          int preload);                                                       /* src/lpcnet.c:273-277 */

/* ---- additions (not in the reference) ------------------------------------------------------ */
/* message of the last failure on the calling thread ("" if none) */
LPCNET_EXPORT const char *lpcnet_hip_last_error(void);
/* Sticky status of the calling thread: 0, or the code of the FIRST void entry point that failed on this thread since
 * lpcnet_hip_clear_error() (the call zero-filled its output; see the header comment).  Codes: */
#define LPCNET_HIP_E_NODEVICE (-2)          /* no usable HIP device */
#define LPCNET_HIP_E_DEVICE   (-3)          /* a HIP call or kernel launch failed */
#define LPCNET_HIP_E_ARG      (-4)          /* bad argument (e.g. preload outside 0..N) */
#define LPCNET_HIP_E_MODEL    (-5)          /* no model bound / no default model / model evicted / malformed blob */
LPCNET_EXPORT int lpcnet_hip_status(void);
LPCNET_EXPORT void lpcnet_hip_clear_error(void);
/* Sticky status of the MODEL a state is bound to: 0, or the code of the first failed pass on that model since it was last
 * cleared (clear != 0 resets it).  A server thread that shares a model with others learns here that some caller's pass
 * failed -- under the combining dispatcher one failed pass fails every call it carried. */
LPCNET_EXPORT int lpcnet_hip_model_status(const LPCNetState *st, int clear);
/* Combining dispatcher of the per-state entry points (lpcnet_synthesize, lpcnet_synthesize_impl / _tail_impl, run_frame_network):
 * out3 = {calls served, device passes that served them, calls in the largest pass} since the process started or the last reset.
 * calls / passes > 1 means concurrent callers of one model were served together. */
LPCNET_EXPORT int lpcnet_hip_dispatch_stats(unsigned long long *out3, int reset);
/* "src=<hash> dev=<hash>": sha1 prefixes of the sources this library was built from (all of lpcnet_amd/csrc + include, and
 * the device sources alone); bench.py echoes them so that a measurement can be tied to the tree it claims */
LPCNET_EXPORT const char *lpcnet_hip_build_info(void);
/* install the VQ codebooks used by lpcnet_decode (the reference compiles them in from
 * ceps_codebooks.c, a generated file that is not part of its tree): cb1..3 [1024][17], diff4 [4096][18] */
LPCNET_EXPORT void lpcnet_hip_set_codebooks(const float *cb1, const float *cb2, const float *cb3, const float *cb_diff4);
/* process-default model for states that never saw lpcnet_load_model() (see the header comment); the blob is copied.
 * 0 on success, -1 on a malformed blob or missing device. */
LPCNET_EXPORT int lpcnet_hip_set_default_model(const unsigned char *data, int len);
/* lpcnet_load_model() for a decoder state (LPCNetDecState is opaque in the reference and has no model entry point) */
LPCNET_EXPORT int lpcnet_hip_decoder_load_model(LPCNetDecState *st, const unsigned char *data, int len);
/* HIP device used by the single-stream API for models bound from now on (default: $LPCNET_HIP_DEVICE, else 0) */
LPCNET_EXPORT int lpcnet_hip_set_device(int device);
/* host-only blob validation (no GPU needed): 0 = loadable (float or int8 flavour), -1 = malformed.
 * info (may be NULL) receives {is_int8, GRU-A blocks, GRU-B blocks, items/lane, padded GRU-B blocks, selftest} */
LPCNET_EXPORT int lpcnet_hip_check_model(const unsigned char *data, int len, int *info);
/* release every device resource held for the single-stream API (optional, e.g. before exit); states stay bound and
 * re-create the device side at their next call */
LPCNET_EXPORT void lpcnet_hip_shutdown(void);

#ifdef __cplusplus
}
#endif
#endif
