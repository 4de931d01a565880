/* lc3.h -- drop-in C ABI of the MI355X LC3plus encode engine.
 *
 * Every entry point below replaces the identically named function of the ETSI TS 103 634 V1.2.1
 * floating-point reference (R = LC3plus_ETSI_src_v17171_20200723/src/floating_point): same symbol,
 * same argument meaning, same LC3_Error values (R/lc3.h:53-75), same ownership rules, so a caller such
 * as R/codec_exe.c links against liblc3plus_hip.so instead of the reference objects.  Behind the ABI
 * every frame is encoded by hand-written gfx950 HIP kernels - a pipeline of kernels, each with the unit of work its dependences allow (one frame per
 * lane, four frames or one channel-stream per wavefront: DESIGN.md section 3); calls of a few frames and this single-stream API run one kernel with one
 * channel-stream per wavefront.  There is no CPU fallback: if no MI355X / HIP runtime is available the first lc3_enc_* call that needs the device
 * returns LC3_ERROR and prints a diagnostic.
 *
 * The float decoder (lc3_dec_*, R/lc3.h:318-406) is exported too, for every operating point of the reference.
 * Error protection (lc3_*_set_ep_*) is not part of the float reference build and has no entry points.
 */
#ifndef LC3PLUS_HIP_LC3_H
#define LC3PLUS_HIP_LC3_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LC3_VERSION_INT(major, minor, micro) (((major) << 16) | ((minor) << 8) | (micro))
#define LC3_VERSION LC3_VERSION_INT(1, 4, 10)        /* R/lc3.h:36-39 */
#define LC3_MAX_CHANNELS 16
#define LC3_MAX_SAMPLES 960
#define LC3_MAX_BYTES 1250

typedef enum {                                       /* values identical to R/lc3.h:53-75 */
    LC3_OK = 0, LC3_ERROR = 1, LC3_DECODE_ERROR = 2, LC3_NULL_ERROR = 3, LC3_SAMPLERATE_ERROR = 4,
    LC3_CHANNELS_ERROR = 5, LC3_BITRATE_ERROR = 6, LC3_NUMBYTES_ERROR = 7, LC3_EPMODE_ERROR = 8,
    LC3_FRAMEMS_ERROR = 9, LC3_ALIGN_ERROR = 10, LC3_HRMODE_ERROR = 11, LC3_BITRATE_UNSET_ERROR = 12,
    LC3_BITRATE_SET_ERROR = 13, LC3_HRMODE_BW_ERROR = 14, LC3_PLCMODE_ERROR = 15, LC3_EPMR_ERROR = 16,
    LC3_WARNING = 17, LC3_BW_WARNING = 18
} LC3_Error;

typedef struct LC3_Enc LC3_Enc;                      /* opaque, caller-allocated: R/lc3.h:113 */

int lc3_version(void);                                                   /* R/lc3.h:120, R/lc3.c:27 */
int lc3_channels_supported(int channels);                                /* R/lc3.h:127, R/lc3.c:32 */
int lc3_samplerate_supported(int samplerate);                            /* R/lc3.h:134, R/lc3.c:37 */

int       lc3_enc_get_size(int samplerate, int channels);                /* R/lc3.h:201, R/lc3.c:111 */
LC3_Error lc3_enc_init(LC3_Enc* encoder, int samplerate, int channels);  /* R/lc3.h:163, R/lc3.c:102 */
LC3_Error lc3_enc_set_frame_ms(LC3_Enc* encoder, float frame_ms);        /* R/lc3.h:255, R/lc3.c:165 */
LC3_Error lc3_enc_set_hrmode(LC3_Enc* encoder, int hrmode);              /* R/lc3.h:265, R/lc3.c:177 */
LC3_Error lc3_enc_set_bitrate(LC3_Enc* encoder, int bitrate);            /* R/lc3.h:236, R/lc3.c:149 */
LC3_Error lc3_enc_set_bandwidth(LC3_Enc* encoder, int bandwidth);        /* R/lc3.h:274, R/lc3.c:187 */
int       lc3_enc_get_input_samples(const LC3_Enc* encoder);             /* R/lc3.h:208, R/lc3.c:118 */
int       lc3_enc_get_num_bytes(const LC3_Enc* encoder);                 /* R/lc3.h:245, R/lc3.c:124 */
int       lc3_enc_get_real_bitrate(const LC3_Enc* encoder);              /* R/lc3.h:216, R/lc3.c:131 */
int       lc3_enc_get_delay(const LC3_Enc* encoder);                     /* R/lc3.h:281, R/lc3.c:159 */

/* One frame: input_samples[ch] -> lc3_enc_get_input_samples() samples (planar), output_bytes receives
 * lc3_enc_get_num_bytes() bytes (channel payloads concatenated).  R/lc3.h:182-191, R/lc3.c:210-234. */
LC3_Error lc3_enc_fl(LC3_Enc* encoder, void** input_samples, int bitdepth, void* output_bytes, int* num_bytes);
LC3_Error lc3_enc16(LC3_Enc* encoder, int16_t** input_samples, void* output_bytes, int* num_bytes);
LC3_Error lc3_enc24(LC3_Enc* encoder, int32_t** input_samples, void* output_bytes, int* num_bytes);
LC3_Error lc3_enc32(LC3_Enc* encoder, int32_t** input_samples, void* output_bytes, int* num_bytes);

/* Releases device resources; lc3_enc_free_memory additionally free()s `encoder` exactly like the
 * reference does (R/lc3.c:301-309), so the block must come from malloc().  R/lc3.h:288,295. */
LC3_Error lc3_enc_free_memory(LC3_Enc* encoder);
LC3_Error lc3_free_encoder_structs(LC3_Enc* encoder);

/* ---- decoder ---- */
typedef struct LC3_Dec LC3_Dec;                      /* opaque, caller-allocated: R/lc3.h:116 */
typedef enum { LC3_PLC_STANDARD = 0, LC3_PLC_ADVANCED = 1 } LC3_PlcMode;   /* R/lc3.h:107-111; only STANDARD is accepted (R/lc3.c:64-72) */

int       lc3_dec_get_size(int samplerate, int channels);                 /* R/lc3.h:359, R/lc3.c:247 */
LC3_Error lc3_dec_init(LC3_Dec* decoder, int samplerate, int channels, LC3_PlcMode plc_mode);   /* R/lc3.h:318, R/lc3.c:238 */
LC3_Error lc3_dec_set_frame_ms(LC3_Dec* decoder, float frame_ms);         /* R/lc3.h:368, R/lc3.c:254 */
LC3_Error lc3_dec_set_hrmode(LC3_Dec* decoder, int hrmode);               /* R/lc3.h:392, R/lc3.c:347 */
int       lc3_dec_get_output_samples(const LC3_Dec* decoder);             /* R/lc3.h:375, R/lc3.c:265 */
int       lc3_dec_get_delay(const LC3_Dec* decoder);                      /* R/lc3.h:382, R/lc3.c:271 */

/* One frame: input_bytes holds num_bytes bytes (channel payloads concatenated, split as R/dec_lc3_fl.c:148),
 * output_samples[ch] receives lc3_dec_get_output_samples() samples of bps 16 (int16_t) or 24/32 (int32_t).
 * bfi_ext = 1 (or num_bytes = 0) conceals the frame.  Returns LC3_DECODE_ERROR when the frame was concealed
 * (lost or found corrupt), exactly as R/lc3.c:277-282 -> R/dec_lc3_fl.c:134-163. */
LC3_Error lc3_dec_fl(LC3_Dec* decoder, void* input_bytes, int num_bytes, void** output_samples, int bps, int bfi_ext);
LC3_Error lc3_dec16(LC3_Dec* decoder, void* input_bytes, int num_bytes, int16_t** output_samples, int bfi_ext);
LC3_Error lc3_dec24(LC3_Dec* decoder, void* input_bytes, int num_bytes, int32_t** output_samples, int bfi_ext);
LC3_Error lc3_dec32(LC3_Dec* decoder, void* input_bytes, int num_bytes, int32_t** output_samples, int bfi_ext);

LC3_Error lc3_dec_free_memory(LC3_Dec* decoder);                          /* R/lc3.h:399, R/lc3.c:311 */
LC3_Error lc3_free_decoder_structs(LC3_Dec* decoder);                     /* R/lc3.h:406, R/lc3.c:334 */

#ifdef __cplusplus
}
#endif
#endif
