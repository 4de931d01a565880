/* lc3_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement of the ETSI TS 103 634 (LC3plus V1.4.10) floating-point ENCODER, written
 * from scratch with the reference's arithmetic order and C type promotions so that on x86-64
 * (gcc -O2 -ffp-contract=off, glibc libm) its bitstreams are byte-identical to the reference's
 * (R = /root/reference/LC3plus_ETSI_src_v17171_20200723/src/floating_point).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * Parity pinning: tests/test_oracle_vs_ref.py (against oracle/_ref, the unmodified reference built
 * by oracle/Makefile) and tests/golden/ (vectors generated from oracle/_ref, committed).
 */
#ifndef LC3_ORACLE_H
#define LC3_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LC3O_MAX_N 960
#define LC3O_MAX_CH 2
#define LC3O_RESB 640          /* bytes of residual bits kept by the decoder restatement (R/defines.h:118-119 rounds 5000 bits up) */

/* error codes follow R/lc3.h:53-75 */
enum { LC3O_OK = 0, LC3O_ERROR = 1, LC3O_DECODE_ERROR = 2, LC3O_NUMBYTES_ERROR = 7, LC3O_NULL_ERROR = 3, LC3O_SAMPLERATE_ERROR = 4, LC3O_CHANNELS_ERROR = 5,
       LC3O_BITRATE_ERROR = 6, LC3O_FRAMEMS_ERROR = 9, LC3O_HRMODE_ERROR = 11, LC3O_BITRATE_SET_ERROR = 13,
       LC3O_HRMODE_BW_ERROR = 14, LC3O_BW_WARNING = 18, LC3O_UNSUPPORTED = 100 };

/* Intermediate values of one channel-frame, for stage-by-stage kernel debugging. */
typedef struct {
    float spec_mdct[LC3O_MAX_N];    /* after MDCT (R/mdct.c:103) */
    float s12k8[129];               /* resampler output (R/resamp12k8.c) */
    int   T0; float normcorr;       /* OLPA */
    int   ltpf_param[3]; int ltpf_bits;
    int   attack;
    float ener[64];                 /* per-band energy before SNS modifies it */
    int   bw_idx;
    float scf[16]; int scf_idx[7]; float scf_q[16];
    float spec_shaped[LC3O_MAX_N];  /* after SNS shaping */
    int   tns_nfilt, tns_order[2], tns_rc_idx[16], tns_bits;
    float spec_tns[LC3O_MAX_N];     /* after TNS */
    int   target_bits_quant;
    float gain0; int gg_idx0, gg_min;   /* estimate */
    int   nbits0;                   /* first quantisation */
    float gain; int gg_idx, gain_change;
    int   nbits, nbits2, lastnz, lsb_mode;
    int   xq[LC3O_MAX_N];
    int   fac_ns;
    int   n_res_bits;
    int   bp_side, mask_side;       /* after side info */
} lc3o_trace;

typedef struct lc3o_enc lc3o_enc;

int  lc3o_enc_sizeof(void);
int  lc3o_enc_init(lc3o_enc* e, int samplerate, int channels);            /* R/lc3.c:102 */
int  lc3o_enc_set_frame_ms(lc3o_enc* e, float frame_ms);                  /* R/lc3.c:165 */
int  lc3o_enc_set_hrmode(lc3o_enc* e, int hrmode);                        /* R/lc3.c:177 */
int  lc3o_enc_set_bitrate(lc3o_enc* e, int bitrate);                      /* R/lc3.c:149 */
int  lc3o_enc_set_bandwidth(lc3o_enc* e, int bandwidth);                  /* R/lc3.c:187 */
int  lc3o_enc_get_input_samples(const lc3o_enc* e);
int  lc3o_enc_get_num_bytes(const lc3o_enc* e);
int  lc3o_enc_get_real_bitrate(const lc3o_enc* e);
int  lc3o_enc_get_delay(const lc3o_enc* e);
/* planar input, input[ch] -> frame_length samples; bitdepth 16 (int16_t*), 24 or 32 (int32_t*) */
int  lc3o_enc_frame(lc3o_enc* e, void** input, int bitdepth, uint8_t* out, int* num_bytes);
void lc3o_enc_set_trace(lc3o_enc* e, lc3o_trace* tr /* array of `channels` traces or NULL */);
void lc3o_enc_free(lc3o_enc* e);

/* Convenience for tests / CPU baseline: B independent MONO streams, T frames each.
 * pcm[B][T][N] int16, out[B][T][stride]; per-stream bitrate; returns 0 or an error code. */
/* decoder restatement (lc3_oracle_dec.inc): R/lc3.h:318-399.  Caller provides lc3o_dec_sizeof() bytes. */
typedef struct lc3o_dec lc3o_dec;
/* intermediate values of one decoded channel-frame (same layout as the device decoder's lc3d_dec_trace) */
typedef struct {
    int bfi, bw_idx, lastnz, lsb_mode, gg_idx, fac_ns, nfilt, tns_order[2], tns_idx[16], scf_idx[7], ltpf[3], nf_seed, zero_frame, nres;
    int xq[LC3O_MAX_N];
    float scf_q[16];
    float q_gain[LC3O_MAX_N];       /* after residual decoding, noise filling and the global gain */
    float q_tns[LC3O_MAX_N];        /* after the TNS synthesis filter */
    float q_shaped[LC3O_MAX_N];     /* after SNS shaping (the IMDCT input; the concealed spectrum on a lost frame) */
    float x_imdct[LC3O_MAX_N];      /* time signal before the LTPF */
    float x_out[LC3O_MAX_N];        /* after the LTPF */
} lc3o_dec_trace;
void lc3o_dec_set_trace(lc3o_dec* d, lc3o_dec_trace* tr /* [channels] or NULL */);
int  lc3o_dec_sizeof(void);
int  lc3o_dec_init(lc3o_dec* d, int samplerate, int channels);
int  lc3o_dec_set_frame_ms(lc3o_dec* d, float frame_ms);
int  lc3o_dec_set_hrmode(lc3o_dec* d, int hrmode);
int  lc3o_dec_get_output_samples(const lc3o_dec* d);
int  lc3o_dec_frame(lc3o_dec* d, const uint8_t* input, int num_bytes, void** output, int bps, int bfi_ext);   /* 0, LC3O_DECODE_ERROR (concealed) or an error */
int  lc3o_dft(float* x, int n);      /* test hook: forward complex DFT of length n in place (interleaved re, im); 0 = no kernel */
int  lc3o_encode_batch16_ch(int samplerate, float frame_ms, int hrmode, int channels, int B, int T, const int* bitrate,
                            const int16_t* pcm, uint8_t* out, int stride);
int  lc3o_encode_batch16_bw(int samplerate, float frame_ms, int hrmode, int B, int T, const int* bitrate, const int* bw_plan /* [B][T], 0 = keep; may be NULL */,
                            const int16_t* pcm, uint8_t* out, int stride);
int  lc3o_encode_batch16(int samplerate, float frame_ms, int hrmode, int B, int T, const int* bitrate,
                         const int16_t* pcm, uint8_t* out, int stride);

#ifdef __cplusplus
}
#endif
#endif
