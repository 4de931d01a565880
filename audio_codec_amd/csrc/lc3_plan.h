/* lc3_plan.h -- structures shared by the host C code (lc3_host.c) and the HIP kernels (lc3_kernels.hip).
 *
 * lc3d_plan : everything that is constant for a batch (geometry derived as in R/setup_enc_lc3.c:73-193 and
 *             the init-time tables the reference builds with libm at start-up: DCT-IV twiddles R/dct4.c:51-63,
 *             DCT-II(16) twiddles R/dct4.c:43-45, IDCT-II cosines R/sns_quantize_scf.c:30, SNS pre-emphasis
 *             R/sns_compute_scf.c:91, global-gain powers R/estimate_global_gain.c:136 / R/adjust_global_gain.c:47).
 *             Built on the host with the host libm -- exactly where the reference evaluates them -- and uploaded once.
 * lc3d_chan : per channel-stream bitrate-derived values (R/setup_enc_lc3.c:196-375).
 * State     : per channel-stream cross-frame state (R/setup_enc_lc3.h:17-62), LC3D_STATE_WORDS 32-bit words in HBM.
 */
#ifndef LC3_PLAN_H
#define LC3_PLAN_H
#include <stdint.h>

#define LC3D_MAX_N 960          /* largest frame length (96 kHz / 10 ms); the kernels come in two LDS layouts, see lc3_kernels.hip */
#define LC3D_PFA_STRIDE 160     /* longest prime-factor DFT (32 kHz / 10 ms: 160 = 32 x 5) */
#define LC3D_GAIN_TAB 512       /* gain index k = ind + gg_off in [-256, 255] -> tab[k + 256] */

typedef struct {
    int32_t fs, fs_idx, dms, hrmode, N, ylen, la, nbands, bw_bits, fft_len, channels;
    int32_t rs_mem_in_len, rs_stride, len12, n12, ltpf_mem_len;
    int32_t att_nblocks, att_hang, bw_cls, win_off, band_off, tilt;
    float   att_damping, sns_damping, rs_scale, frame_ms, dct4_norm;
    /* float constants the reference obtains from powf()/sqrtf() at run time with constant arguments */
    float   c_1em5_a;           /* powf(10.0,-5.0)  R/olpa.c:112 */
    float   c_1em5_b;           /* powf(10,-5)      R/ltpf_coder.c:97 */
    float   c_1em4;             /* powf(10.0,-4.0)  R/sns_compute_scf.c:101 */
    float   c_2m32, c_2m31, c_2m24, c_2p15, c_2p100;
    float   c_sqrt2;            /* sqrtf(2) */
    float   c_idct_n1, c_idct_n2;   /* R/sns_quantize_scf.c:24-25 */
    float   c_thr7_up;          /* smallest float >= (7.0)*(28.0/20.0):  (double)t < thr  <=>  t < c_thr7_up  (R/estimate_global_gain.c:106) */
    float   c_thr50_dn;         /* largest float <= (50.0)*(28.0/20.0): (double)t > thr  <=>  t > c_thr50_dn (R/estimate_global_gain.c:111) */
    int32_t pfa_nst;            /* prime-factor DFT stages (2 or 3) for N/2 in {10,20,30,40,80,120,160}; 0: N/2 = 240 or 60 use their own kernels */
    int32_t pfa_rad[3];         /* stage radices (leaf DFT lengths), in execution order */
    float   pad0;
    float   tw1[LC3D_MAX_N], tw2[LC3D_MAX_N];      /* N/2 complex (re,im) pairs each */
    float   dct2_tw[32];
    float   sns_preemph[64];
    float   gain_est[LC3D_GAIN_TAB];               /* powf(10, (k)/28.0)  (double division) */
    float   gain_adj[LC3D_GAIN_TAB];               /* powf(10, (float)k/28) (float division) */
    float   rs_taps[240];                          /* 12.8 kHz resampler low-pass, phase-major: [start][m] = lp[239 - start - m*stride] (R/resamp12k8.c:48-57) */
    double  idct_cos[256];
    /* quantiser bit estimate (R/quantize_spec.c:60-170), derived from the arithmetic-coder tables so that a 2-tuple costs three loads:
     * per context t in [0, 1024): the probability-model index of each of the four escape classes in one word; the cost of the
     * escape symbols as cumulative sums over the classes; and the bit-cost table itself behind the same base pointer */
    uint16_t q_esc[1024][4];    /* e0, e0 + e1, e0 + e1 + e2, e3   with e_j = ac_bits[ctx_lut[t + 1024 j] * 17 + 16] (3 x 20480 < 2^16) */
    uint32_t q_lut4[1024];      /* ctx_lut[t] | ctx_lut[t + 1024] << 8 | ctx_lut[t + 2048] << 16 | ctx_lut[t + 3072] << 24 */
    uint16_t q_bits[1088];      /* = lc3t_ac_bits */
    uint8_t band_of_bin[LC3D_MAX_N];
    uint8_t pfa_src[3 * LC3D_PFA_STRIDE];   /* prime-factor DFT: gather maps of up to three stages; for N/2 = 60: [0..59] = (45k+16l)%60 */
    uint8_t pfa_dst[LC3D_PFA_STRIDE];       /* scatter of the last stage; for N/2 = 60: (15k+4l)%60 */
} lc3d_plan;
#define LC3D_PLAN_HEAD_WORDS 45     /* the scalar head (up to and including pad0) that every wave copies into LDS */

typedef struct {
    int32_t nbytes, total_bits, target_bits_init, lpc_weighting, ltpf_enable, gg_off, attack_handling, reg_bits;
    int32_t out_off;            /* byte offset of this channel's payload inside the stream-frame */
    int32_t bandwidth, bw_cut_bin, bw_index;
    int32_t reset_attack;       /* set by a bitrate change that disables attack handling (R/setup_enc_lc3.c:297-308) */
    int32_t pad[3];
} lc3d_chan;

/* ---- state layout (32-bit words), parametrised by the kernel layout's MDCT-memory slot (300 standard, 600 large) ---- */
#define LC3D_MEMCAP_STD 300
#define LC3D_MEMCAP_BIG 600                       /* 96 kHz / 10 ms: N - la_zeros = 960 - 360 */
#define LC3D_LAYOUT_BIG(N, la) ((N) > 480 || (N) - (la) > LC3D_MEMCAP_STD)
#define LC3D_ST_XPREV   0                         /* MDCT / resampler memory: tail of the previous frame, right-aligned in the slot */
#define LC3D_H12_KEEP   384                       /* last 384 samples of the HP-filtered 12.8 kHz stream */
#define LC3D_H6_KEEP    194                       /* last 194 samples of the 6.4 kHz stream */
#define LC3D_ST_H12(mc)  (LC3D_ST_XPREV + (mc))
#define LC3D_ST_H6(mc)   (LC3D_ST_H12(mc) + LC3D_H12_KEEP)
#define LC3D_ST_SCAL(mc) (LC3D_ST_H6(mc) + LC3D_H6_KEEP + 2)   /* 16 float scalars (kernel fsc[0..15]) then 16 int scalars (isc[0..15]) */
#define LC3D_S_OLPA_PITCH_WORD(mc) (LC3D_ST_SCAL(mc) + 16 + 0) /* isc[I_OLPA_PITCH]: initial value 17 (R/setup_enc_lc3.c:178) */
#define LC3D_STATE_WORDS(mc) ((mc) + 660)
#define LC3D_STATE_WORDS_MAX LC3D_STATE_WORDS(LC3D_MEMCAP_BIG)

/* hand-over from the frame-parallel front of the encoder (lc3_enc_front_kernel: MDCT, band energies, bandwidth, SNS scale factors) through
 * the one-frame-per-lane vector quantiser (lc3_enc_snsvq_kernel) to the sequential kernel: per channel-frame a record of FR_WORDS words
 * and the MDCT spectrum row of N floats */
#define FR_SCF   0                       /* 16 floats: scale factors (R/sns_compute_scf.c) */
#define FR_SCFQ  16                      /* 16 floats: quantised scale factors (R/sns_quantize_scf.c) */
#define FR_IDX   32                      /* 7 ints: the SNS indices */
#define FR_BW    39                      /* int: bandwidth index (R/detect_cutoff_warped.c) */
#define FR_ATT   40                      /* attack detector (R/attack_detector.c): 4 block energies, [44..45] the filter memory after this frame, [46] the flag (lc3_enc_attack_kernel) */
#define FR_ATTM  44
#define FR_ATTFLAG 46
#define FR_LTPF  48                      /* 4 ints from lc3_enc_pitch_kernel: LTPF flag, active, pitch index, side bits (R/ltpf_coder.c:245-254) */
#define FR_TNS   52                      /* 20 ints from lc3_enc_shape_kernel: filters, orders (2), bits, coefficient indices (16) = isc[I_TNS_NF ...] (R/tns_coder.c) */
#define FR_GGMIN 72                      /* float: smallest gain index of the frame, before the offset (R/estimate_global_gain.c:72-77) */
#define FR_XZERO 73                      /* int: the shaped spectrum is all zero (:65-70) */
#define FR_BWC   74                      /* int: bandwidth index behind the bandwidth controller (R/cutoff_bandwidth.c) */
#define FR_RATE  76                      /* 4 words from lc3_enc_rate_kernel: gain index (int), gain (float), bits of the first quantisation, its lastnz */
#define FR_WORDS 80
/* spectrum row of the pipelined encoder path, per channel-frame: [0, ylen) the MDCT spectrum (lc3_enc_front_kernel), shaped and TNS-filtered in
 * place by lc3_enc_shape_kernel, which appends the ylen / 4 log energies of the gain estimate; rows are SROW words apart */
#define LC3D_SROW(ylen) (((ylen) + ((ylen) >> 2) + 15) & ~15)
/* Storage: a row is SROW / 16 chunks of 16 floats, and chunk c of row (cs, t) lies at ((cs * NCH + c) * RT + t) * 16 (RT rows per channel-stream):
 * the same chunk of consecutive frames of a stream is adjacent in memory.  The one-frame-per-lane kernels give 64 consecutive frames to the
 * 64 lanes of a wave, so each of their 64-byte-per-lane accesses is one 4 KB run; a wave that owns one frame reads 64-byte segments. */
#define LC3D_ROW_BASE(rows, cs, t, RT, srow) ((rows) + (((size_t)(cs) * ((srow) >> 4)) * (size_t)(RT) + (size_t)(t)) * 16)
#define LC3D_ROW_OFF(k, RT) (((((size_t)((k) >> 4)) * (size_t)(RT)) << 4) + (size_t)((k) & 15))

/* per channel-frame status bits of the encoder: conditions the reference only asserts on (SURVEY 5 "failure detection") */
#define LC3D_ENC_ST_BIT_BUDGET  1        /* side information + range-coder bits exceed the frame (R/ari_codec.c:777) */
#define LC3D_ENC_ST_QUANT_RANGE 2        /* a quantised line outside int16 without the high-resolution mode (R/quantize_spec.c:50) */

/* ---- decoder (lc3_dec_kernels.inc) ---- */
#define DEC_LY 864                       /* LTPF output history: ceil(228 * 48000 / 12800) + 6 = 861 */
#define DEC_LX 16                        /* LTPF input history (tilt filter length - 1 <= 10) */
typedef struct {                         /* per channel-stream decoder configuration, R/setup_dec_lc3.c:188-299 */
    int32_t nbytes, lpc_weighting, gg_off, N_red_tns, fs_red_tns, ltpf_beta_idx; float ltpf_beta; int32_t in_off;
} lc3d_dchan;

/* decoder state words per channel-stream */
#define DST_IMEM   0                                   /* 600: IMDCT overlap memory (300 used by the standard layout) */
#define DST_QPREV  600                                 /* 960: last good spectrum (concealment) */
#define DST_LY     1560                                /* 864: LTPF output history */
#define DST_LX     (1560 + DEC_LY)                     /* 16 : LTPF input history */
#define DST_SCAL   (1560 + DEC_LY + DEC_LX)            /* 16 scalars */
#define DST_WORDS  (1560 + DEC_LY + DEC_LX + 16)
/* hand-over between the two encoder kernels (lc3_encode_kernel -> lc3_enc_pack_kernel, lc3_enc_pack.inc), per channel-frame in HBM */
#define PK_RES 64                        /* [0..55] the encoder's isc[] scalars; [64..223] residual bits, LSB first (the LSB-mode list in LSB mode) */
#define PK_XQ  224                       /* quantised spectrum up to lastnz: one word per 2-tuple (int16 pairs), or int32 lines in high-resolution mode */
#define PK_STRIDE(N, hr) (PK_XQ + ((hr) ? ((N) > 480 ? 960 : 480) : ((N) > 480 ? 480 : 240)))
/* hand-over between the two decoder kernels (lc3_dec_parse.inc -> lc3_dec_kernels.inc), both in HBM */
#define PR_WORDS 112                     /* per channel-frame record: isc[0..38] of the decoder (side information), [39] = bfi after parsing, [48..111] = the 64 SNS band gains */
#define PR_GAINS 48
#define PR_PLC 40                        /* lost frames: [40] nbLostCmpt, [41] cumulative attenuation (float), [42] first seed, [43] last good frame of the launch or -1 */
#define OV_ROW_STD 480                    /* transformed frame in HBM: the N samples of the time-domain aliasing buffer (R/imdct.c:34-44), standard layout */
#define OV_ROW_BIG 960                    /* large layout */
#define PR_BFI 39
#define WS_ROW(N) ((N) > 480 ? 960 : 480)  /* per channel-frame spectrum row (words) */
enum { DS_PITCH_INT = 0, DS_PITCH_FR, DS_BETA_IDX, DS_PARAM0, DS_PARAM1, DS_PARAM2, DS_GAIN /* float */, DS_NBLOST, DS_CUM_ALPHA /* float */, DS_PLC_SEED,
       DS_PREV_BFI, DS_PREVPREV_BFI };


#endif
