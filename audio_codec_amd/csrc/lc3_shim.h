/* lc3_shim.h -- the thin C-ABI between the host C code (lc3_host.c) and the HIP side (lc3_kernels.hip). */
#ifndef LC3_SHIM_H
#define LC3_SHIM_H
#include <stdint.h>
#include <stddef.h>
#include "lc3_plan.h"

/* per channel-frame intermediates (debug / stage-level parity tests); same field order as oracle/lc3_oracle.h:lc3o_trace */
typedef struct {
    float spec_mdct[960]; float s12k8[129]; int T0; float normcorr; int ltpf_param[3]; int ltpf_bits; int attack;
    float ener[64]; int bw_idx; float scf[16]; int scf_idx[7]; float scf_q[16]; float spec_shaped[960];
    int tns_nfilt, tns_order[2], tns_rc_idx[16], tns_bits; float spec_tns[960];
    int target_bits_quant; float gain0; int gg_idx0, gg_min; int nbits0; float gain; int gg_idx, gain_change;
    int nbits, nbits2, lastnz, lsb_mode; int xq[960]; int fac_ns; int n_res_bits; int bp_side, mask_side;
} lc3d_trace;

/* per decoded channel-frame intermediates; same layout as oracle/lc3_oracle.h: lc3o_dec_trace */
typedef struct {
    int bfi, bw_idx, lastnz, lsb_mode, gg_idx, fac_ns, nfilt, tns_order[2], tns_idx[16], scf_idx[7], ltpf[3], nf_seed, zero_frame, nres;
    int xq[960]; float scf_q[16]; float q_gain[960]; float q_tns[960]; float q_shaped[960]; float x_imdct[960]; float x_out[960];
} lc3d_dec_trace;


#ifdef __cplusplus
extern "C" {
#endif
int   lc3hip_dec_create(void** ctx, const lc3d_plan* plan, int n_streams, int device);
int   lc3hip_dec_upload_chans(void* ctx, const lc3d_dchan* chans, int first, int count);
int   lc3hip_dec_decode(void* ctx, const void* frames, int frames_on_device, int in_stride, const uint8_t* bfi_flags_host, int n_frames,
                        void* pcm, int pcm_on_device, int bps, uint8_t* status_host, void* hip_stream, int sync, void* trace_host);
float lc3hip_dec_last_ms(void* ctx);
int   lc3hip_dec_destroy(void* ctx);
int   lc3hip_create(void** ctx, const lc3d_plan* plan, int n_streams, int device);
int   lc3hip_reset_state(void* ctx, const float* init_state_one);
int   lc3hip_upload_chans(void* ctx, const lc3d_chan* chans, int first, int count);
int   lc3hip_encode(void* ctx, const void* pcm, int pcm_on_device, int bitdepth, int n_frames, void* out, int out_stride,
                    int out_on_device, void* hip_stream, int sync, void* trace_host);
float lc3hip_last_ms(void* ctx);
size_t lc3hip_state_bytes(void* ctx);                             /* checkpoint / resume of the per-stream state (include/lc3plus_batch.h) */
int   lc3hip_get_state(void* ctx, void* host, size_t bytes);
int   lc3hip_set_state(void* ctx, const void* host, size_t bytes);
size_t lc3hip_dec_state_bytes(void* ctx);
int   lc3hip_dec_get_state(void* ctx, void* host, size_t bytes);
int   lc3hip_dec_set_state(void* ctx, const void* host, size_t bytes);
int   lc3hip_dec_set_input_ready(void* ctx, int ready);          /* see lc3plus_dec_batch_set_input_ready (include/lc3plus_batch.h) */
int   lc3hip_set_input_ready(void* ctx, int ready);              /* see lc3plus_enc_batch_set_input_ready (include/lc3plus_batch.h) */
int   lc3hip_last_status(void* ctx, uint8_t* status_host, int n);        /* LC3D_ENC_ST_* bits per channel-frame of the last call; returns the count copied */
int   lc3hip_last_records(void* ctx, float* rec_host, int max_words);   /* the per-frame records of the last pipelined call [channel-stream][frame][FR_WORDS]; returns the words copied (0: the last call did not take that path) */
int   lc3hip_destroy(void* ctx);
int   lc3hip_test_fastmath(int kind, const float* x_host, float* y_host, long long n);   /* test hook: lc3_fastmath.h on the device over an array (0 log2, 1 log10, 2 2^x) */
#ifdef __cplusplus
}
#endif
#endif
