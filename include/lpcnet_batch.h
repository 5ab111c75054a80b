/* lpcnet_batch.h -- batched multi-stream extension of the LPCNet C API (additive).
 *
 * The reference library is one-state-one-stream and single threaded (src/lpcnet.c has no
 * threading); the MI355X engine gets its throughput from thousands of independent streams, so
 * this header adds a batch object that owns the per-stream state on the device.  A batch is
 * semantically n independent LPCNetState objects driven in lock step:
 *   lpcnet_batch_synthesize(b, F, stride, P, T)  ==  for every stream s, for t < T:
 *       lpcnet_synthesize(state[s], &F[(s*T + t)*stride], &P[(s*T + t)*160], 160)
 * Streams never interact, so several batches (one per GPU, one process per GPU) shard a workload
 * with no communication.
 */
#ifndef LPCNET_BATCH_H_
#define LPCNET_BATCH_H_

#include "lpcnet.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct LPCNetBatch LPCNetBatch;

/* n_streams independent synthesis states on HIP device `device`; NULL on failure. */
LPCNET_EXPORT LPCNetBatch *lpcnet_batch_create(int n_streams, int device);
/* The same, sharded over several HIP devices of one node: device k of `devices[0..n_devices)` owns a
 * contiguous block of streams (sizes differ by at most one), with its own copy of the model, its own device buffers
 * and HIP stream; host-pointer calls (synthesize, decode) run one host thread per shard concurrently and involve no
 * communication between devices.  A device may be listed more than once (several shards on one GPU). */
LPCNET_EXPORT LPCNetBatch *lpcnet_batch_create_sharded(int n_streams, const int *devices, int n_devices);
LPCNET_EXPORT int lpcnet_batch_shards(const LPCNetBatch *b);
/* block of streams [first, first+count) and device of one shard (any output may be NULL) */
LPCNET_EXPORT int lpcnet_batch_shard_info(const LPCNetBatch *b, int shard, int *first, int *count, int *device);
LPCNET_EXPORT void lpcnet_batch_destroy(LPCNetBatch *b);
LPCNET_EXPORT int lpcnet_batch_streams(const LPCNetBatch *b);
/* like lpcnet_load_model(): 0 or -1.  The blob is copied; it may be freed afterwards. */
LPCNET_EXPORT int lpcnet_batch_load_model(LPCNetBatch *b, const unsigned char *data, int len);
/* lpcnet_reset() on streams [first, first+count) */
LPCNET_EXPORT int lpcnet_batch_reset(LPCNetBatch *b, int first, int count);

/* Host-pointer synthesis: features [n_streams][n_frames][feat_stride] (only [0..19] of a frame are
 * read, feat_stride >= 20; 36 = `lpcnet_demo -synthesis` file layout), pcm [n_streams][n_frames*160].
 * Copies in, runs, copies out, synchronises.  Returns 0 or a negative error code. */
LPCNET_EXPORT int lpcnet_batch_synthesize(LPCNetBatch *b, const float *features, int feat_stride, short *pcm, int n_frames);
/* Device-pointer synthesis: same layouts in device memory of the batch's device; only enqueues on
 * `hip_stream` (a hipStream_t; NULL = the batch's own stream).  lpcnet_batch_sync() waits.
 * Ordering: calls on different streams are ordered by the batch (an event chain); the batch's buffers and stream states are
 * shared by all of them.  HIP graphs: the call may be issued on a stream that is being CAPTURED (nothing runs, nothing is
 * allocated or synchronised; any number of launches may be captured, the kernels' arguments travel inside the graph).  The
 * event chain does not see a capture, so REPLAYS of such a graph are not ordered against other work on the same batch --
 * eager calls, other graphs, state export / import, reset: put them on the replay's stream or order them with events, exactly
 * as for two kernels that share a buffer. */
LPCNET_EXPORT int lpcnet_batch_synthesize_device(LPCNetBatch *b, const float *d_features, int feat_stride, short *d_pcm,
                                                 int n_frames, void *hip_stream);
/* device-pointer synthesis on ONE shard of a sharded batch: pointers are on that shard's device and cover only its
 * `count` streams ([count][n_frames][feat_stride] / [count][n_frames*160]) */
LPCNET_EXPORT int lpcnet_batch_synthesize_device_shard(LPCNetBatch *b, int shard, const float *d_features, int feat_stride,
                                                       short *d_pcm, int n_frames, void *hip_stream);
/* waits for everything enqueued for this batch, on whichever stream(s) it was enqueued */
LPCNET_EXPORT int lpcnet_batch_sync(LPCNetBatch *b);
/* Teacher-forced variant = lpcnet_synthesize_impl(..., preload) of the reference
 * (src/lpcnet.c:256-259,273): the first `preload` samples of every frame are read from pcm. */
LPCNET_EXPORT int lpcnet_batch_synthesize_preload(LPCNetBatch *b, const float *features, int feat_stride, short *pcm,
                                                  int n_frames, int preload);
/* ONE frame step with PER-STREAM arguments -- what a batched packet-loss concealment needs (src/lpcnet_plc.c drives each of
 * its streams with lpcnet_synthesize_impl(N, preload), lpcnet_synthesize_tail_impl(N, preload) or nothing, depending on that
 * stream's losses).  features [n_streams][feat_stride], pcm [n_streams][160]; per stream s:
 *   mode[s] = 0  leave the stream alone
 *   mode[s] = 1  frame network on features[s], then n_samples[s] samples      (lpcnet_synthesize_impl, src/lpcnet.c:273-277)
 *   mode[s] = 2  n_samples[s] samples from the products of the stream's most recent mode-1 step
 *                                                                          (lpcnet_synthesize_tail_impl, src/lpcnet.c:235-271)
 *   n_samples[s] in 1..160; the first preload[s] <= n_samples[s] samples of pcm[s] are imposed on the synthesis filter
 *   (teacher forcing, src/lpcnet.c:256-259) and returned unchanged; samples n_samples[s]..159 of pcm[s] are not touched.
 * Streams with equal (mode, n_samples, preload) run together; results are bit-identical to driving each stream alone through
 * the single-stream entry points.  Meant for occasional use (it compacts and scatters the groups), not for throughput.
 * A mode-2 step needs a mode-1 step of THIS call as the stream's most recent frame step: the products are kept per stream
 * only here, so after lpcnet_batch_synthesize* / _decode* / a reset / a state import the call returns -4 (bad argument) for
 * such a stream (checked for all streams before anything runs).  The call is not atomic across groups: if a later group
 * fails on the device, earlier groups of the same call have already advanced. */
LPCNET_EXPORT int lpcnet_batch_synthesize_step(LPCNetBatch *b, const float *features, int feat_stride, short *pcm,
                                               const int *n_samples, const int *preload, const int *mode);
/* Codec path: packets [n_streams][n_packets][8] -> pcm [n_streams][n_packets*640] (lpcnet_decode per stream) */
LPCNET_EXPORT int lpcnet_batch_decode(LPCNetBatch *b, const unsigned char *packets, short *pcm, int n_packets);
/* the same with device pointers (packets [n][n_packets][8], pcm [n][n_packets*640]), only enqueued on `hip_stream`
 * (NULL = the batch's own stream): bit unpacking, VQ lookup and interpolation (decode_packet, src/lpcnet_dec.c:81-155)
 * run in a device kernel; the per-stream VQ memory lives on the device and is cleared by lpcnet_batch_reset */
/* LPC_GAMMA (bandwidth expansion of the LPC filter, lpc_weighting src/freq.c:299-308) is a compile-time constant of the
 * reference's generated nnet_data.h, not part of the weight blob; models trained with --lpc-gamma != 1 set it after
 * lpcnet_batch_load_model.  gamma in (0, 1], default 1. */
LPCNET_EXPORT int lpcnet_batch_set_lpc_gamma(LPCNetBatch *b, float gamma);
/* END2END models: the reference's compile-time END2END (src/lpcnet.c:56-80,107-108) -- the LPC filter comes from the
 * first 16 conditioning outputs (reflection coefficients, rc2lpc) instead of the cepstrum; set after lpcnet_batch_load_model */
LPCNET_EXPORT int lpcnet_batch_set_end2end(LPCNetBatch *b, int on);
/* Arithmetic flavour of the sample loop.  0 = PARITY (default): every product and sum rounded like the reference's
 * generic-C build, results bit-identical to it.  1 = FAST: the arithmetic of the reference's own SIMD builds -- fused
 * multiply-add for float blobs (src/vec_avx.h:790-858), int32 block accumulation for int8 blobs (src/vec_avx.h:690-750) --
 * not bit-identical to any reference build (those differ among themselves as well); tests/test_gpu_fast.py keeps its
 * teacher-forced deviation inside the reference's own AVX2-vs-generic envelope.
 * 2 = FAST with the dual fully-connected layer of the sampler in fp16 (weights and GRU-B state as halves, fp32 accumulation,
 * v_dot2_f32_f16): BASELINE.json config 4's "fp16 dual-FC".  The reference has no fp16 arithmetic, so the pin is to a stated
 * definition: the oracle restates it (oracle/lpcnet_oracle.c: orc_mdense_f16_path -- binary16 weights and GRU-B state, fp32
 * sums, both v_dot2 rounding orders) and replays every tree decision of the engine from the traced GRU-B state against the
 * reference RNG's thresholds, within a 2e-4 x max(1, |logit|) band (tests/test_gpu_fast.py::
 * test_fp16_dual_fc_against_the_oracle_restatement; fewer than 1 % of the decisions may fall inside the band). */
LPCNET_EXPORT int lpcnet_batch_set_fast(LPCNetBatch *b, int on);
LPCNET_EXPORT int lpcnet_batch_decode_device(LPCNetBatch *b, const unsigned char *d_packets, short *d_pcm, int n_packets,
                                             void *hip_stream);

/* State interchange with the single-stream API (PLC-style snapshot / rollback). */
LPCNET_EXPORT int lpcnet_batch_export_state(LPCNetBatch *b, int stream, LPCNetState *st);
LPCNET_EXPORT int lpcnet_batch_import_state(LPCNetBatch *b, int stream, const LPCNetState *st);

/* Tuning / introspection */
LPCNET_EXPORT int lpcnet_batch_set_streams_per_workgroup(LPCNetBatch *b, int s);      /* 1, 2, 4, 8; 0 = auto.  8 = the two-group kernel (float blobs with a dense GRU-B matrix and
                                                                                        * <= 32 GRU-A items per lane, bit-exact arithmetic): chosen automatically beyond four streams per CU */
LPCNET_EXPORT int lpcnet_batch_get_streams_per_workgroup(const LPCNetBatch *b);
/* Streams per workgroup are measured on the batch itself (PARITY arithmetic; FAST takes a table value so that its output
 * never depends on timing): lpcnet_batch_tune() does it now, on the engine's own stream (~10 ms).  Without it the first
 * host-pointer call measures; the enqueue-only *_device calls on a caller's stream never do (they use the table value). */
LPCNET_EXPORT int lpcnet_batch_tune(LPCNetBatch *b);
LPCNET_EXPORT int lpcnet_batch_enable_timing(LPCNetBatch *b, int on);
LPCNET_EXPORT int lpcnet_batch_last_timing(LPCNetBatch *b, float *ms_sample_kernel, float *ms_frame_kernels);
LPCNET_EXPORT const char *lpcnet_batch_last_error(void);

/* Parity seams used by the test-suite (SURVEY.md §7 hard part 9): the two halves of the path alone.
 *   tail:   sample loop only, frame products given: cond_a [n][T][1152], cond_b [n][T][48], lpc [n][T][16]
 *   frames: frame network + LPC only, products returned in the same layouts (any may be NULL) */
LPCNET_EXPORT int lpcnet_batch_run_tail(LPCNetBatch *b, const float *cond_a, const float *cond_b, const float *lpc,
                                        short *pcm, int n_frames, int preload);
LPCNET_EXPORT int lpcnet_batch_run_frames(LPCNetBatch *b, const float *features, int feat_stride,
                                          float *cond_a, float *cond_b, float *lpc, int n_frames);
/* tools: GRU-A row dealing of a blob (out[65]: items per lane, then per wave bound[4] + candidate-only flags[3], then
 * per wave the number of early head items of slot 0) */
LPCNET_EXPORT int lpcnet_hip_model_layout(const unsigned char *data, int len, int *out);
/* the engine's own correctly rounded 10^x of the LPC path (pow(10.f, x) of src/freq.c:317) evaluated on the device */
LPCNET_EXPORT int lpcnet_hip_exp10_device(const float *x, double *out, int n);
/* The arithmetic identities the bit-exact (PARITY) kernels rest on, evaluated on the device with the library's own compile flags
 * and float mode, as bit patterns: v_mfma_f32_4x4x1(A, B, C = -0.0) == v_mul_f32 (out_mfma / out_mul, [n][4]: product k of lane i =
 * a[4*(i/4) + k] * b[i]) and the halves of v_pk_mul_f32 / v_pk_add_f32 == v_mul_f32 / v_add_f32 (out_pk / out_sc, [n][4] =
 * {mul (a[i], b[i]), mul (a[i^1], b[i^1]), add (a[i], b[i]), add (a[i^1], b[i^1])}).  n a multiple of 64.
 * tests/test_gpu_parity.py::test_matrix_pipe_multiplier_and_packed_math_are_exact feeds it stratified operands (subnormal
 * inputs and products, underflow, +-0, the largest finite products, infinities). */
LPCNET_EXPORT int lpcnet_hip_arith_identities_device(const float *a, const float *b, unsigned *out_mfma, unsigned *out_mul,
                                                     unsigned *out_pk, unsigned *out_sc, int n);
/* The int8 kernels re-quantise the GRU states with ONE instruction (v_cvt_rpi_i32_f32) where the reference evaluates
 * (int)floor(.5 + 127 x) with the sum in double (src/vec.h:311-316): this runs both over ALL 2^32 float bit patterns on the device.
 * out3 = {mismatches among finite |t| < 2^31, mismatches inside the reachable |t| <= 127.5, one mismatching pattern}. */
LPCNET_EXPORT int lpcnet_hip_quant_sweep_device(unsigned long long *out3);
/* raw per-stream state record (layout = struct lpcn_stream_state in lpcnet_amd/csrc/lpcnet_engine.h) */
LPCNET_EXPORT int lpcnet_batch_state_size(void);
LPCNET_EXPORT int lpcnet_batch_get_raw_state(LPCNetBatch *b, int stream, void *out);
LPCNET_EXPORT int lpcnet_batch_set_raw_state(LPCNetBatch *b, int stream, const void *in);
LPCNET_EXPORT int lpcnet_batch_debug_trace(LPCNetBatch *b, int n_samples, float *host_out);
/* shader-clock totals per phase of the sample kernel (workgroup 0): out == NULL enables/zeros, else 8 values */
LPCNET_EXPORT int lpcnet_batch_profile(LPCNetBatch *b, unsigned long long *out);

#ifdef __cplusplus
}
#endif
#endif
