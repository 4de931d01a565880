/* lc3plus_batch.h -- batched multi-stream extension of the lc3_enc_* ABI (new in this engine).
 *
 * The reference encodes one frame of one stream per call (R/lc3.c:226-234 -> R/enc_lc3_fl.c:162-174).
 * Frames of ONE stream are not independent (MDCT overlap, pitch / LTPF memories and the rate-control
 * loop persist, R/setup_enc_lc3.h:17-62), but streams (and channels, R/enc_lc3_fl.c:167-171) are.
 * A batch therefore is n_streams independent encoder instances with identical (samplerate, frame_ms,
 * hrmode, channels) and a per-stream bitrate; each encode() call advances every stream by n_frames
 * frames with the per-stream state resident in HBM between calls.  One wavefront encodes one
 * channel-stream.  Plain pointers and sizes only.
 */
#ifndef LC3PLUS_BATCH_H
#define LC3PLUS_BATCH_H
#include "lc3.h"
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lc3plus_batch lc3plus_batch;

/* Creates n_streams encoders on HIP device `device` (-1 = current).  bitrates[n_streams] is the total
 * bitrate per stream (all channels), validated like lc3_enc_set_bitrate.  Error codes as lc3_enc_*. */
LC3_Error lc3plus_enc_batch_create(lc3plus_batch** batch, int n_streams, int samplerate, int channels,
                                   float frame_ms, int hrmode, const int* bitrates, int device);
LC3_Error lc3plus_enc_batch_destroy(lc3plus_batch* batch);

int lc3plus_enc_batch_input_samples(const lc3plus_batch* batch);    /* samples per channel per frame */
int lc3plus_enc_batch_num_bytes(const lc3plus_batch* batch, int stream); /* bytes per frame of that stream */
int lc3plus_enc_batch_stride(const lc3plus_batch* batch);           /* max over streams of num_bytes */
LC3_Error lc3plus_enc_batch_set_bitrate(lc3plus_batch* batch, int stream, int bitrate);
LC3_Error lc3plus_enc_batch_set_bandwidth(lc3plus_batch* batch, int stream, int bandwidth);

/* Advances every stream by n_frames.
 *   pcm : [n_streams][n_frames][channels][input_samples] samples, int16_t (bitdepth 16) or int32_t (24/32)
 *   out : [n_streams][n_frames][out_stride] bytes; frame payload = num_bytes(stream) bytes, rest untouched
 *   *_on_device : 0 = host pointer, 1 = device pointer used in place.  With BOTH on the host the call is cut into runs of frames
 *                 whose H2D copy, kernels and D2H copy overlap on three HIP streams; pinned caller memory (hipHostMalloc /
 *                 hipHostRegister) is copied in place, pageable memory is staged through the batch's pinned buffers
 *   hip_stream  : hipStream_t to enqueue on (NULL = the batch's own stream); the call returns after the
 *                 work is complete when sync != 0, otherwise right after enqueueing.                  */
LC3_Error lc3plus_enc_batch_encode(lc3plus_batch* batch, const void* pcm, int pcm_on_device, int bitdepth,
                                   int n_frames, void* out, int out_stride, int out_on_device,
                                   void* hip_stream, int sync);

/* Checkpoint / resume.  The cross-frame state of every channel-stream of the batch (MDCT / resampler memory, pitch and LTPF histories,
 * rate-control and attack-detector words; R/setup_enc_lc3.h:17-62) as one opaque host array of state_size() bytes.  A batch created with
 * the same (n_streams, samplerate, channels, frame_ms, hrmode, bitrates, bandwidths) that is given the state continues the streams
 * exactly where the first one stopped - on another GPU or in another process.  Both calls wait for the last encode() to finish. */
size_t    lc3plus_enc_batch_state_size(const lc3plus_batch* batch);
LC3_Error lc3plus_enc_batch_get_state(lc3plus_batch* batch, void* state, size_t size);
LC3_Error lc3plus_enc_batch_set_state(lc3plus_batch* batch, const void* state, size_t size);

/* A promise about device-pointer calls, off by default: with ready != 0 the caller guarantees that the PCM passed to every following
 * encode() call is COMPLETE in device memory when the call is made - not merely queued earlier on hip_stream (so it is wrong for PCM
 * that a kernel or copy queued on hip_stream is still producing).  The batch then lets the frame-parallel and pitch kernels of a
 * call start on its own streams while the sequential tail and the bitstream writer of the previous call on the same hip_stream are
 * still running (consecutive calls of equal n_frames of up to 256 frames; up to three calls are then in flight); results are identical, and the output of a call is complete in stream order
 * on hip_stream as before.  The promise covers the OUTPUT buffer as well: it must be free to be written when the call is made - not still being read by work
 * queued earlier on hip_stream - because the bitstream writers of consecutive calls may run beside each other (large frames, short calls), each into the
 * buffer of its own call.  Streaming servers that fill their PCM ring ahead of the encode calls and drain their output ring behind them are the use. */
LC3_Error lc3plus_enc_batch_set_input_ready(lc3plus_batch* batch, int ready);

/* Kernel-only timing of the last encode() call in milliseconds (HIP events on the launch stream). */
float lc3plus_enc_batch_last_kernel_ms(lc3plus_batch* batch);

/* Per channel-frame status of the last encode() call (device-pointer calls and host calls short enough for one run): conditions the
 * reference only asserts on - bit 0: side information + range-coder bits exceed the frame (R/ari_codec.c:777), bit 1: a quantised
 * line outside int16 outside the high-resolution mode (R/quantize_spec.c:50).  status: host array [n_streams * channels][n_frames];
 * returns the number of entries written (0 before the first call), negative on error.  Waits for the call to finish. */
int lc3plus_enc_batch_last_status(lc3plus_batch* batch, uint8_t* status, int max_entries);

/* Diagnostics: the per channel-frame records the kernels of the pipelined path hand to each other, of the last encode() call that took that path
 * (calls of more than 8 frames - more than 5 under the input-ready promise - that are neither traced nor run with LC3PLUS_ENC_FUSED / _NO_SPLIT; 0 words when
 * the last call did not take it): host array [n_streams * channels][n_frames][lc3plus_enc_batch_record_words()] of 32-bit words
 * - 16 scale factors, 16 quantised scale factors, 7 SNS indices, bandwidth index, attack-detector words, 4 LTPF words, 20 TNS words (filters,
 * orders, bits, coefficient indices), gain floor, all-zero flag, bandwidth behind the controller, and gain index / gain / bit count / last
 * non-zero line of the first quantisation (layout: FR_* in audio_codec_amd/csrc/lc3_plan.h).  The stage-level parity tests compare them with the
 * reference restatement's trace of the same frames.  Returns the words written, negative on error.  Waits for the call to finish. */
int lc3plus_enc_batch_last_records(lc3plus_batch* batch, float* records, int max_words);
int lc3plus_enc_batch_record_words(void);

/* ---- batched decoder: n_streams independent decoder instances, one wavefront per channel-stream; same
 * conventions as the encoder batch.  num_bytes[n_streams] = bytes per stream-frame (all channels; may be NULL
 * and set later per stream).  R/dec_lc3_fl.c:134-163 is what one (stream, frame) does. ---- */
typedef struct lc3plus_dec_batch lc3plus_dec_batch;
LC3_Error lc3plus_dec_batch_create(lc3plus_dec_batch** batch, int n_streams, int samplerate, int channels,
                                   float frame_ms, int hrmode, const int* num_bytes, int device);
LC3_Error lc3plus_dec_batch_destroy(lc3plus_dec_batch* batch);
int       lc3plus_dec_batch_output_samples(const lc3plus_dec_batch* batch);
int       lc3plus_dec_batch_delay(const lc3plus_dec_batch* batch);
int       lc3plus_dec_batch_num_bytes(const lc3plus_dec_batch* batch, int stream);
LC3_Error lc3plus_dec_batch_set_num_bytes(lc3plus_dec_batch* batch, int stream, int num_bytes);
/*   frames : [n_streams][n_frames][in_stride] bytes (payload = num_bytes(stream) bytes at the start of each slot)
 *   bfi    : host pointer, [n_streams][n_frames] bad-frame flags (1 = conceal) or NULL
 *   pcm    : [n_streams][n_frames][channels][output_samples], int16_t (bps 16) or int32_t (24/32)
 *   status : host pointer or NULL, [n_streams][n_frames]: 1 where the frame was concealed (LC3_DECODE_ERROR) */
LC3_Error lc3plus_dec_batch_decode(lc3plus_dec_batch* batch, const void* frames, int frames_on_device, int in_stride,
                                   const uint8_t* bfi, int n_frames, void* pcm, int pcm_on_device, int bps,
                                   uint8_t* status, void* hip_stream, int sync);
float     lc3plus_dec_batch_last_kernel_ms(lc3plus_dec_batch* batch);
/* checkpoint / resume of the decoders' cross-frame state (overlap-add memory, last good spectrum, LTPF histories, concealment words), as
 * for the encoder batch; the frame sizes are configuration (lc3plus_dec_batch_set_num_bytes), not state */
size_t    lc3plus_dec_batch_state_size(const lc3plus_dec_batch* batch);
LC3_Error lc3plus_dec_batch_get_state(lc3plus_dec_batch* batch, void* state, size_t size);
LC3_Error lc3plus_dec_batch_set_state(lc3plus_dec_batch* batch, const void* state, size_t size);
/* The decoder's counterpart of lc3plus_enc_batch_set_input_ready: with ready != 0 the caller guarantees that the frames passed to every
 * following decode() call with device pointers (no bad-frame flags, no status) are COMPLETE in device memory when the call is made.  The
 * bitstream parser of a call - stateless - then runs on a stream of the batch beside the transform and synthesis of the call before; results
 * are identical, and the PCM of a call is complete in stream order on hip_stream as before.  Off by default. */
LC3_Error lc3plus_dec_batch_set_input_ready(lc3plus_dec_batch* batch, int ready);

/* lc3plus_enc_* spellings of the single-stream API (north-star wording); thin aliases. */
LC3_Error lc3plus_enc_init(LC3_Enc* e, int samplerate, int channels);
LC3_Error lc3plus_enc_set_frame_ms(LC3_Enc* e, float frame_ms);
LC3_Error lc3plus_enc_set_hrmode(LC3_Enc* e, int hrmode);
LC3_Error lc3plus_enc_set_bitrate(LC3_Enc* e, int bitrate);
LC3_Error lc3plus_enc16(LC3_Enc* e, int16_t** input_samples, void* output_bytes, int* num_bytes);
int       lc3plus_enc_get_size(int samplerate, int channels);

#ifdef __cplusplus
}
#endif
#endif
