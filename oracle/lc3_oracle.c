/* lc3_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See lc3_oracle.h.
 *
 * CPU restatement of the ETSI LC3plus floating-point encoder.  Each function cites the reference
 * lines it follows (R = /root/reference/LC3plus_ETSI_src_v17171_20200723/src/floating_point).
 * Arithmetic order and C promotions (float vs. double sub-expressions) mirror the reference exactly;
 * control/data structure is ours.
 *
 * Math modes:  default            -> glibc float libm (powf/log2f/log10f), byte-identical to oracle/_ref
 *              -DLC3O_PORTABLE_MATH -> (float)f((double)x) versions, which is what the HIP kernels use
 *                                      (device double libm rounded to float); see DESIGN.md "libm boundary".
 */
#include <assert.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include "lc3_oracle.h"
#include "lc3_tables.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#if defined(LC3O_PORTABLE_MATH) && LC3O_PORTABLE_MATH
static inline float m_log2f(float x) { return (float)log2((double)x); }
static inline float m_log10f(float x) { return (float)log10((double)x); }
static inline float m_powf(float x, float y) { return (float)pow((double)x, (double)y); }
#else
static inline float m_log2f(float x) { return log2f(x); }
static inline float m_log10f(float x) { return log10f(x); }
static inline float m_powf(float x, float y) { return powf(x, y); }
#endif

#define IMIN(a, b) ((a) < (b) ? (a) : (b))
#define IMAX(a, b) ((a) > (b) ? (a) : (b))

#include "lc3_oracle_fft.inc"

/* ------------------------------------------------------------------------------------------------ */
/* state                                                                                             */
/* ------------------------------------------------------------------------------------------------ */
typedef struct { float r, i; } cpx;

typedef struct {                       /* per channel: R/setup_enc_lc3.h:17-62 */
    /* bitrate-derived */
    int nbytes, total_bits, target_bits_init, target_bits_ari, lpc_weighting, ltpf_enable, gg_off,
        attack_handling, reg_bits;
    /* cross-frame state */
    float mdct_mem[LC3O_MAX_N];
    float rs_mem_in[120], rs_mem_out[24], rs_mem_50[2];
    float olpa_mem12[3], olpa_mem6[64 + 114 + 16];
    int   olpa_pitch;
    float ltpf_mem_x[232 + 128 + 1 + 32];
    float ltpf_nc1, ltpf_nc2, ltpf_pitch; int ltpf_on;
    float att_mem[2], att_acc; int att_pos, att_flag;
    float tbits_off; int mem_target_bits, mem_spec_bits;
} chan_t;

struct lc3o_enc {
    int fs, fs_in, fs_idx, channels, dms, hrmode, br_set, bitrate;
    float frame_ms;
    int N, ylen, la, nbands, bw_bits, tilt, rs_mem_in_len, ltpf_mem_len;
    int bandwidth, bw_cut_bin, bw_index;
    float sns_damping, att_damping; int att_nblocks, att_hang;
    const float* win; const uint16_t* bands; const uint16_t* cut_bins;
    /* plans */
    cpx tw1[LC3O_MAX_N / 2], tw2[LC3O_MAX_N / 2]; float dct4_norm;
    cpx dct2_tw[16];
    double idct_cos[16][16];
    float sns_preemph[64];
    int fft_kind;    /* N/2 when a DFT kernel of that length is restated (lc3_oracle_fft.inc dft_any), 0 = unsupported */
    lc3o_trace* trace;
    chan_t ch[LC3O_MAX_CH];
};

int lc3o_enc_sizeof(void) { return (int)sizeof(lc3o_enc); }
void lc3o_enc_set_trace(lc3o_enc* e, lc3o_trace* tr) { e->trace = tr; }
void lc3o_enc_free(lc3o_enc* e) { (void)e; }

static inline cpx cexpi_f(float x) { cpx c = {cosf(x), sinf(x)}; return c; }          /* R/util.h:109 */
static inline cpx cmul_f(cpx a, cpx b) { cpx c = {a.r * b.r - a.i * b.i, a.i * b.r + a.r * b.i}; return c; } /* R/util.h:104 */

static const lc3t_cfg_t* find_cfg(int fs_idx, int dms, int hr)
{
    for (int i = 0; i < LC3T_NCFG; i++)
        if (lc3t_cfg[i].valid && lc3t_cfg[i].fs_idx == fs_idx && lc3t_cfg[i].dms == dms && lc3t_cfg[i].hr == hr) return &lc3t_cfg[i];
    return NULL;
}

/* R/setup_enc_lc3.c:73-193 (set_enc_frame_params) + R/mdct.c:72-92 + R/dct4.c:51-63 plan setup */
static void frame_params(lc3o_enc* e)
{
    e->N = e->fs / 100;
    if (e->hrmode == 1) { e->ylen = e->N; e->sns_damping = 0.6; }
    else { e->ylen = IMIN(400, e->N); e->sns_damping = 0.85; }
    e->ltpf_mem_len = 232;
    if (e->fs_idx == 5) e->hrmode = 1;                         /* quirk kept: SURVEY 9 */
    e->bw_bits = e->hrmode ? 0 : lc3t_bw_bits[e->fs_idx];
    int cls = e->dms == 25 ? 0 : e->dms == 50 ? 1 : 2;
    e->cut_bins = &lc3t_bw_bins[cls * 6];
    if (e->dms == 100) { e->att_nblocks = 4; e->att_damping = 0.5; e->att_hang = 2; }
    if (e->dms == 25) { e->N >>= 2; e->ylen /= 4; e->ltpf_mem_len = 232 + 32; }
    if (e->dms == 50) { e->N >>= 1; e->ylen /= 2; }
    const lc3t_cfg_t* c = find_cfg(e->fs_idx, e->dms, e->hrmode);
    e->win = NULL; e->bands = NULL; e->nbands = 64; e->la = 0; e->fft_kind = 0;
    if (c) { e->win = &lc3t_win_pool[c->win_off]; e->bands = &lc3t_band_pool[c->band_off]; e->nbands = c->nbands; e->la = c->la_zeros; }
    for (int ch = 0; ch < e->channels; ch++) {
        e->ch[ch].olpa_pitch = 17;
        memset(e->ch[ch].mdct_mem, 0, sizeof e->ch[ch].mdct_mem);   /* mdct_free + mdct_init: calloc'd memory */
    }
    /* DCT-IV twiddles: R/dct4.c:51-63 */
    int len = e->N;
    for (int i = 0; i < len / 2; i++) {
        e->tw1[i] = cexpi_f(-M_PI * (i + 0.25) / len);
        e->tw2[i] = cexpi_f(-M_PI * i / len);
    }
    e->dct4_norm = 1.0 / sqrtf(len / 2);                       /* R/dct4.c:82 */
    { const int h = len / 2; e->fft_kind = (h == 10 || h == 20 || h == 30 || h == 40 || h == 60 || h == 80 || h == 120 || h == 160 || h == 240 || h == 480) ? h : 0; }
    /* DCT-II(16) post-twiddle: R/dct4.c:43-45 */
    for (int i = 0; i < 16; i++) {
        cpx s = {2 / sqrtf(2 * 16), 0};
        e->dct2_tw[i] = cmul_f(cexpi_f(-M_PI * i / (2 * 16)), s);
    }
    /* IDCT-II cosine table in double: R/sns_quantize_scf.c:30 */
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++)
        e->idct_cos[i][j] = cos(M_PI / (2.0 * (float)16) * (2.0 * ((float)i + 1.0) - 1.0) * ((float)j));
    /* SNS pre-emphasis: R/sns_compute_scf.c:91 */
    for (int i = 0; i < 64; i++)
        e->sns_preemph[i] = powf(10.0, (float)i * (float)e->tilt / ((float)64 - 1.0) / 10.0);
}

int lc3o_enc_init(lc3o_enc* e, int samplerate, int channels)    /* R/lc3.c:102-109, R/setup_enc_lc3.c:31-70 */
{
    if (!e) return LC3O_NULL_ERROR;
    switch (samplerate) { case 8000: case 16000: case 24000: case 32000: case 44100: case 48000: case 96000: break;
                          default: return LC3O_SAMPLERATE_ERROR; }
    if (channels < 1 || channels > LC3O_MAX_CH) return LC3O_CHANNELS_ERROR;
    memset(e, 0, sizeof *e);
    e->fs = samplerate == 44100 ? 48000 : samplerate; e->fs_in = samplerate;
    e->fs_idx = e->fs / 10000; if (e->fs_idx > 4) e->fs_idx = 5;
    e->dms = 100; e->frame_ms = 10; e->channels = channels;
    e->rs_mem_in_len = 2 * 8 * e->fs / 12800;
    static const int tilts[6] = {14, 18, 22, 26, 30, 34};
    e->tilt = tilts[e->fs_idx];
    frame_params(e);
    return LC3O_OK;
}

int lc3o_enc_set_frame_ms(lc3o_enc* e, float frame_ms)        /* R/lc3.c:165-174 */
{
    if (!e) return LC3O_NULL_ERROR;
    int d = (int)ceil(frame_ms * 10);
    if (d != 25 && d != 50 && d != 100) return LC3O_FRAMEMS_ERROR;
    if (e->br_set) return LC3O_BITRATE_SET_ERROR;
    e->dms = (int)(frame_ms * 10); e->frame_ms = frame_ms;
    frame_params(e);
    return LC3O_OK;
}

int lc3o_enc_set_hrmode(lc3o_enc* e, int hrmode)              /* R/lc3.c:177-184 */
{
    if (!e) return LC3O_NULL_ERROR;
    if (e->fs_in < 48000 && hrmode != 0) return LC3O_SAMPLERATE_ERROR;
    e->hrmode = hrmode > 0;
    frame_params(e);
    return LC3O_OK;
}

int lc3o_enc_set_bitrate(lc3o_enc* e, int bitrate)            /* R/lc3.c:149-157, R/setup_enc_lc3.c:196-375 */
{
    if (!e) return LC3O_NULL_ERROR;
    if (bitrate <= 0) return LC3O_BITRATE_ERROR;
    if (e->fs_idx == 5 && e->hrmode == 0) return LC3O_HRMODE_ERROR;
    int minBR = 0, maxBR = 0;
    if (e->hrmode) {
        switch (e->dms) {
        case 25: maxBR = 672000; minBR = e->fs == 48000 ? 172800 : e->fs == 96000 ? 198400 : -1; break;
        case 50: maxBR = 600000; minBR = e->fs == 48000 ? 148800 : e->fs == 96000 ? 174400 : -1; break;
        case 100: maxBR = 500000; minBR = e->fs == 48000 ? 124800 : e->fs == 96000 ? 149600 : -1; break;
        default: return LC3O_HRMODE_ERROR;
        }
        if (minBR < 0) return LC3O_HRMODE_ERROR;
    } else {
        minBR = 20 * 8 * (1000 / e->frame_ms) * (e->fs_in == 44100 ? 441. / 480 : 1);
        maxBR = 400 * 8 * (1000 / e->frame_ms) * (e->fs_in == 44100 ? 441. / 480 : 1);
    }
    minBR *= e->channels; maxBR *= e->channels;
    if (bitrate < minBR || bitrate > maxBR) return LC3O_BITRATE_ERROR;
    e->br_set = 1;
    int totalBytes = bitrate * e->N / (8 * e->fs_in);
    for (int ch = 0; ch < e->channels; ch++) {
        chan_t* s = &e->ch[ch];
        s->nbytes = totalBytes / e->channels + (ch < (totalBytes % e->channels));
        s->total_bits = s->nbytes << 3;
        s->target_bits_init = s->total_bits - 38 - 8 - 3 - e->bw_bits - ceil(log2f(e->N / 2)) - 2 - 1;
        if (s->total_bits > 1280) s->target_bits_init -= 1;
        if (s->total_bits > 2560) s->target_bits_init -= 1;
        if (e->hrmode) s->target_bits_init -= 1;
        s->target_bits_ari = s->total_bits;
        s->lpc_weighting = s->total_bits < 480;
        if (e->frame_ms == 5) s->lpc_weighting = s->total_bits < 240;
        if (e->frame_ms == 2.5) s->lpc_weighting = s->total_bits < 120;
        s->gg_off = -(IMIN(115, s->total_bits / (10 * (e->fs_idx + 1))) + 105 + 5 * (e->fs_idx + 1));
        if (e->frame_ms == 10 && ((e->fs_in >= 44100 && s->nbytes >= 100) || (e->fs_in == 32000 && s->nbytes >= 81)) &&
            s->nbytes < 340 && e->hrmode == 0) {
            s->attack_handling = 1;
        } else {
            s->attack_handling = 0; s->att_mem[0] = s->att_mem[1] = 0; s->att_flag = 0; s->att_pos = 0; s->att_acc = 0;
        }
        int bitsTmp = s->total_bits;
        if (e->frame_ms == 2.5) bitsTmp = bitsTmp * 4.0 * (1.0 - 0.4);
        if (e->frame_ms == 5) bitsTmp = bitsTmp * 2 - 160;
        s->ltpf_enable = bitsTmp < 640 + (e->fs_idx - 1) * 80;
        if (e->hrmode) s->ltpf_enable = 0;
        if (e->hrmode && e->fs_idx >= 4) {
            int real_rate = s->nbytes * 8000 / e->frame_ms;
            s->reg_bits = real_rate / 12500;
            if (e->fs_idx == 5) { if (e->frame_ms == 10) s->reg_bits += 2; if (e->frame_ms == 2.5) s->reg_bits -= 6; }
            else { if (e->frame_ms == 2.5) s->reg_bits -= 6; if (e->frame_ms == 10) s->reg_bits += 5; }
        } else s->reg_bits = -1;
    }
    e->bitrate = bitrate;
    return LC3O_OK;
}

int lc3o_enc_set_bandwidth(lc3o_enc* e, int bandwidth)        /* R/lc3.c:187-208 */
{
    if (!e) return LC3O_NULL_ERROR;
    if (e->hrmode == 1) return LC3O_HRMODE_BW_ERROR;
    int eff = e->fs_in;
    if (e->bandwidth != bandwidth) {
        if (e->fs_in > 40000) eff = 40000;
        if (bandwidth * 2 > eff) return LC3O_BW_WARNING;
        e->bandwidth = bandwidth;
        e->bw_cut_bin = (bandwidth * e->dms) / 5000;
        e->bw_index = IMAX(0, (bandwidth / 4000) - 1);
    }
    return LC3O_OK;
}

int lc3o_enc_get_input_samples(const lc3o_enc* e) { return e ? e->N : 0; }
int lc3o_enc_get_num_bytes(const lc3o_enc* e) { return e ? e->ch[0].nbytes * e->channels : 0; }
int lc3o_enc_get_delay(const lc3o_enc* e) { return e ? e->N - 2 * e->la : 0; }
int lc3o_enc_get_real_bitrate(const lc3o_enc* e)              /* R/lc3.c:131-147 */
{
    if (!e) return 0;
    if (!e->br_set) return 12;
    int tot = 0;
    for (int ch = 0; ch < e->channels; ch++) tot += e->ch[ch].nbytes;
    int br = (tot * 80000) / e->dms;
    if (e->fs_in == 44100) { int rem = br % 480; br = ((br - rem) / 480) * 441 + (rem * 441) / 480; }
    return br;
}

/* ------------------------------------------------------------------------------------------------ */
/* stages                                                                                            */
/* ------------------------------------------------------------------------------------------------ */

/* R/mdct.c:103-124 + R/dct4.c:75-95 */
static void stage_mdct(const lc3o_enc* e, chan_t* s, const float* in, float* out)
{
    const int N = e->N, la = e->la, ml = N - la, h = N / 2;
    float tmp[2 * LC3O_MAX_N];
    memcpy(tmp, s->mdct_mem, sizeof(float) * ml);
    memcpy(tmp + ml, in, sizeof(float) * N);
    memset(tmp + 2 * N - la, 0, sizeof(float) * la);
    memcpy(s->mdct_mem, tmp + N, sizeof(float) * ml);
    for (int i = 0; i < 2 * N; i++) tmp[i] *= e->win[i];
    float fold[LC3O_MAX_N];
    for (int i = 0; i < h; i++) {
        fold[i] = -tmp[3 * h - i - 1] - tmp[3 * h + i];
        fold[h + i] = tmp[i] - tmp[2 * h - i - 1];
    }
    /* DCT-IV through an N/2 complex DFT */
    float z[LC3O_MAX_N], scratch[LC3O_MAX_N];
    for (int i = 0; i < h; i++) {
        cpx a = {fold[2 * i], fold[N - 2 * i - 1]};
        cpx c = cmul_f(a, e->tw1[i]);
        z[2 * i] = c.r; z[2 * i + 1] = c.i;
    }
    if (!dft_any(z, h, scratch)) assert(!"unsupported transform length");
    for (int i = 0; i < h; i++) {
        cpx a = {z[2 * i], z[2 * i + 1]};
        cpx t = cmul_f(a, e->tw2[i]);
        out[2 * i] = t.r * e->dct4_norm;
        out[N - 2 * i - 1] = -t.i * e->dct4_norm;
    }
}

/* R/resamp12k8.c:13-84.  y gets len12+1 samples. */
static int stage_resample(const lc3o_enc* e, chan_t* s, const float* x, float* y)
{
    const int xlen = e->N, mlen = e->rs_mem_in_len;
    const int len12 = e->dms == 25 ? 32 : e->dms == 50 ? 64 : 128;
    const int n12 = xlen * 12800 / e->fs;
    const float sf = lc3t_rs_scale[e->fs_idx];
    const int stride = lc3t_rs_upfac[e->fs_idx];
    float buf[120 + LC3O_MAX_N], down[128] = {0}, bout[24 + 128 + 8];
    memcpy(buf, s->rs_mem_in, sizeof(float) * mlen);
    memcpy(buf + mlen, x, sizeof(float) * xlen);
    memcpy(s->rs_mem_in, buf + xlen, sizeof(float) * mlen);
    for (int n = 0; n < n12; n++) {
        int i = 15 * n, start = (-i) % stride;
        if (start < 0) start += stride;
        float mac = 0;
        for (int j = start; j < 240; j += stride) mac += buf[(i + j) / stride] * sf * lc3t_rs_lp[240 - j - 1];
        down[n] = mac;
    }
    double u11 = s->rs_mem_50[0], u21 = s->rs_mem_50[1];
    for (int i = 0; i < len12; i++) {
        double y1 = (lc3t_hp50_b[0] * down[i] + u11);
        double u1 = (lc3t_hp50_b[1] * down[i] + u21) - lc3t_hp50_a[1] * y1;
        double u2 = lc3t_hp50_b[2] * down[i] - lc3t_hp50_a[2] * y1;
        u11 = u1; u21 = u2;
        down[i] = (float)y1;
    }
    s->rs_mem_50[0] = (float)u11; s->rs_mem_50[1] = (float)u21;
    memcpy(bout, s->rs_mem_out, sizeof(float) * 24);
    memcpy(bout + 24, down, sizeof(float) * len12);
    memcpy(y, bout, sizeof(float) * (len12 + 1));
    memcpy(s->rs_mem_out, bout + n12, sizeof(float) * 24);
    return len12;
}

static int argmax_first(const float* v, int n, float init)   /* R/olpa.c:33-50 (init=v[0]) / R/ltpf_coder.c:15-32 (init=0) */
{
    int best = 0; float m = init;
    for (int i = 0; i < n; i++) if (v[i] > m) { m = v[i]; best = i; }
    return best;
}

/* R/olpa.c:52-143 */
static void stage_olpa(const lc3o_enc* e, chan_t* s, const float* s12, int len, int* T0_out, float* nc_out)
{
    float buf[64 + 114 + 16 + 8] = {0}, filt[128 + 3] = {0}, d6[64] = {0}, R0[98], R[98];
    int mem_len = 114, len2 = len / 2, acf = len2;
    if (e->dms == 25) { mem_len += 16; acf += 16; }
    float in12[128 + 3];
    memcpy(in12, s->olpa_mem12, sizeof(float) * 3);
    memcpy(in12 + 3, s12, sizeof(float) * len);
    memcpy(s->olpa_mem12, in12 + len, sizeof(float) * 3);
    for (int i = 0; i < len + 3; i++) {                      /* filter_olpa R/olpa.c:16-31 */
        float sum = 0;
        for (int j = 0; j < 5 && j <= i; j++) sum += lc3t_olpa_dec[j] * in12[i - j];
        filt[i] = sum;
    }
    for (int i = 4, j = 0; i < len + 3; i += 2) d6[j++] = filt[i];
    float* s6 = buf + mem_len;
    memcpy(buf, s->olpa_mem6, sizeof(float) * mem_len);
    memcpy(s6, d6, sizeof(float) * len2);
    memcpy(s->olpa_mem6, buf + len2, sizeof(float) * mem_len);
    if (e->dms == 25) s6 -= 16;
    for (int lag = 17; lag <= 114; lag++) {
        float sum = 0;
        for (int j = 0; j < acf; j++) sum += s6[j] * s6[j - lag];
        R0[lag - 17] = sum;
    }
    memcpy(R, R0, sizeof R);
    for (int i = 0; i < 98; i++) R0[i] = R0[i] * lc3t_olpa_w[i];
    int T0 = argmax_first(R0, 98, R0[0]) + 17;
    float s0 = 0, s1 = 0, s2 = 0;
    for (int i = 0; i < acf; i++) { s0 += s6[i] * s6[i - T0]; s1 += s6[i - T0] * s6[i - T0]; s2 += s6[i] * s6[i]; }
    s1 = s1 * s2;
    s1 = sqrtf(s1) + powf(10.0, -5.0);
    float nc = s0 / s1;
    nc = 0 > nc ? 0 : nc;
    int lo = IMAX(17, s->olpa_pitch - 4), hi = IMIN(114, s->olpa_pitch + 4);
    int T02 = argmax_first(&R[lo - 17], hi - lo + 1, R[lo - 17]) + lo;
    if (T02 != T0) {
        s0 = s1 = s2 = 0;
        for (int i = 0; i < acf; i++) { s0 += s6[i] * s6[i - T02]; s1 += s6[i - T02] * s6[i - T02]; s2 += s6[i] * s6[i]; }
        s1 = s1 * s2;
        s1 = sqrtf(s1) + powf(10.0, -5.0);
        float nc2 = s0 / s1;
        nc2 = 0 > nc2 ? 0 : nc2;
        if (nc2 > (nc * 0.85)) { T0 = T02; nc = nc2; }
    }
    s->olpa_pitch = T0;
    *T0_out = T0 * 2.0;
    *nc_out = nc;
}

/* R/ltpf_coder.c:34-263 */
static void stage_ltpf(const lc3o_enc* e, chan_t* s, const float* xin, int xLen, int pitch_ol, float ol_nc, int* param, int* bits)
{
    float buffer[232 + 128 + 1 + 32 + 8] = {0}, cor[64] = {0}, cor_up[160] = {0}, cor_int[64] = {0};
    float cur[256], pred[256];
    const int memLen = e->ltpf_mem_len, N = xLen - 1;
    float* x = buffer + memLen;
    memcpy(buffer, s->ltpf_mem_x, sizeof(float) * memLen);
    memcpy(x, xin, sizeof(float) * xLen);
    memcpy(s->ltpf_mem_x, buffer + N, sizeof(float) * (xLen + memLen - N));
    int active = 0, pitch_index = 0, gain = 0;
    float norm_corr = 0, pitch = 0;
    if (ol_nc > 0.6) {
        int t0_min = IMAX(pitch_ol - 4, 32), t0_max = IMIN(pitch_ol + 4, 228), acf = N;
        if (e->dms == 25) { acf = 2 * N; x = x - N; }
        int t_min = t0_min - 4, t_max = t0_max + 4;
        float sum1 = 0, sum2 = 0;
        for (int j = 0; j < acf; j++) { sum1 += x[j] * x[j]; sum2 += x[j - t_min] * x[j - t_min]; }
        for (int i = t_min; i <= t_max; i++) {
            float sum = 0;
            for (int j = 0; j < acf; j++) sum += x[j] * x[j - i];
            if (i > t_min) sum2 = sum2 + x[-i] * x[-i] - x[acf - 1 - (i - 1)] * x[acf - 1 - (i - 1)];
            float sum3 = sqrtf(sum1 * sum2) + powf(10, -5);
            float nc = sum / sum3;
            nc = 0 > nc ? 0 : nc;
            cor[i - t_min] = nc;
        }
        int t1 = argmax_first(cor + 4, t_max - t_min - 8 + 1, 0) + t0_min;
        int pitch_int, pitch_fr;
        if (t1 >= 157) { pitch_int = t1; pitch_fr = 0; }
        else {
            for (int i = 0, j = 0; i < 4 * (t_max - t_min) + 1; i += 4) cor_up[i] = cor[j++];
            for (int i = 0; i < 4 * (t0_max - t0_min + 1); i++) {
                float sum = 0;
                for (int k = 0; k < 32; k++) sum += cor_up[i + k] * lc3t_ltpf_int4[k];
                cor_int[i] = sum;
            }
            int step = t1 >= 127 ? 2 : 1;
            int mid = 4 * (t1 - t0_min) + 1, up = 4 - step, down = t1 == t0_min ? 0 : 4 - step;
            float sel[16]; int n = 0;
            for (int i = mid - down - 1; i <= mid + up; i += step) sel[n++] = cor_int[i];
            int k = argmax_first(sel, ((mid + up) - (mid - down)) / step + 1, 0);
            pitch_fr = k * step - down;
            if (pitch_fr >= 0) pitch_int = t1; else { pitch_int = t1 - 1; pitch_fr = 4 + pitch_fr; }
        }
        if (pitch_int < 127) pitch_index = pitch_int * 4 + pitch_fr - 128;
        else if (pitch_int < 157) pitch_index = pitch_int * 2 + (pitch_fr / 2) - 254 + 380;
        else pitch_index = pitch_int - 157 + 380 + 60;
        pitch = (float)pitch_int + (float)pitch_fr / 4.0;
        const float* f0 = &lc3t_ltpf_frac[0]; const float* fp = &lc3t_ltpf_frac[4 * pitch_fr];
        for (int n = 0; n < acf; n++) {
            cur[n] = x[n + 1] * f0[0] + x[n] * f0[1] + x[n - 1] * f0[2];
            pred[n] = x[n - pitch_int + 1] * fp[0] + x[n - pitch_int] * fp[1] + x[n - pitch_int - 1] * fp[2] + x[n - pitch_int - 2] * fp[3];
        }
        float a = 0, b = 0, c = 0;
        for (int i = 0; i < acf; i++) a += cur[i] * pred[i];
        for (int i = 0; i < acf; i++) b += cur[i] * cur[i];
        for (int i = 0; i < acf; i++) c += pred[i] * pred[i];
        b = sqrtf(b * c) + powf(10, -5);
        norm_corr = a / b;
        { float lo = -1 > norm_corr ? -1 : norm_corr; norm_corr = 1 < lo ? 1 : lo; }
        if (norm_corr < 0) norm_corr = 0;
        if (s->ltpf_enable == 1) {
            if ((s->ltpf_on == 0 && (e->dms == 100 || s->ltpf_nc2 > 0.94) && s->ltpf_nc1 > 0.94 && norm_corr > 0.94) ||
                (s->ltpf_on == 1 && norm_corr > 0.9) ||
                (s->ltpf_on == 1 && fabsf(pitch - s->ltpf_pitch) < 2 && (norm_corr - s->ltpf_nc1) > -0.1 && norm_corr > 0.84))
                active = 1;
        }
        gain = 4;
    } else { gain = 0; norm_corr = ol_nc; pitch = 0; }
    if (gain > 0) { param[0] = 1; param[1] = active; param[2] = pitch_index; *bits = 11; }
    else { param[0] = param[1] = param[2] = 0; *bits = 1; }
    if (e->dms < 100) s->ltpf_nc2 = s->ltpf_nc1;
    s->ltpf_nc1 = norm_corr; s->ltpf_on = active; s->ltpf_pitch = pitch;
}

/* R/attack_detector.c:13-104 */
static void stage_attack(const lc3o_enc* e, chan_t* s, const float* in)
{
    if (!s->attack_handling) return;
    float tmp[162] = {0}, fsig[160] = {0}, nrg[4] = {0};
    float* p = tmp + 2; float mval = 0;
    const int n16 = e->att_nblocks * 40, N = e->N;
    int j = 0;
    if (e->fs == 96000) { for (int i = 0; i < N; i += 6) p[j++] = in[i] + in[i + 1] + in[i + 2] + in[i + 3] + in[i + 4] + in[i + 5]; mval = 1e-5; }
    else if (e->fs == 48000) { for (int i = 0; i < N; i += 3) p[j++] = (in[i] + in[i + 1] + in[i + 2]); }
    else if (e->fs == 32000) { for (int i = 0; i < N; i += 2) p[j++] = (in[i] + in[i + 1]); }
    else if (e->fs == 24000) { for (int i = 0; i < N; i += 3) p[j++] = (in[i] + (in[i + 1] + in[i + 2]) / 2.0); }
    p[-2] = s->att_mem[0]; p[-1] = s->att_mem[1];
    s->att_mem[0] = p[n16 - 2]; s->att_mem[1] = p[n16 - 1];
    for (int i = 159; i >= 0; i--) {
        float t = 0;
        t += p[i] * 0.375; t += p[i - 1] * (-0.5); t += p[i - 2] * (0.125);
        fsig[i] = t;
    }
    for (int b = 0; b < e->att_nblocks; b++) {
        float sum = 0;
        for (int k = 0; k < 40; k++) sum += fsig[k + b * 40] * fsig[k + b * 40];
        nrg[b] = sum;
    }
    s->att_flag = 0; int pos = -1;
    for (int b = 0; b < e->att_nblocks; b++) {
        float t = nrg[b] / 8.5;
        if (t > (s->att_acc > mval ? s->att_acc : mval)) { s->att_flag = 1; pos = b + 1; }
        s->att_acc = nrg[b] > 0.25 * s->att_acc ? nrg[b] : 0.25 * s->att_acc;
    }
    if (s->att_pos > e->att_hang) s->att_flag = 1;
    s->att_pos = pos;
}

/* R/per_band_energy.c:13-30 */
static void stage_band_energy(const lc3o_enc* e, const float* d, float* en)
{
    for (int b = 0; b < e->nbands; b++) {
        float sum = 0;
        for (int j = e->bands[b]; j < e->bands[b + 1]; j++) sum += d[j] * d[j];
        en[b] = sum / (float)(e->bands[b + 1] - e->bands[b]);
    }
}

/* R/detect_cutoff_warped.c:13-83 */
static int stage_bw_detect(const lc3o_enc* e, const float* en)
{
    const int cls = e->dms == 25 ? 0 : e->dms == 50 ? 1 : 2, f = e->fs_idx;
    const uint8_t* st = &lc3t_bw_start[(cls * 4 + f - 1) * 4]; const uint8_t* sp = &lc3t_bw_stop[(cls * 4 + f - 1) * 4];
    int counter = f;
    float sum = 0;
    for (int i = st[counter - 1]; i <= sp[counter - 1]; i++) sum += en[i];
    float mean = sum / (sp[counter - 1] - st[counter - 1] + 1);
    while (mean < lc3t_bw_quiet_thr[counter - 1]) {
        counter--;
        if (counter == 0) break;
        sum = 0;
        for (int i = st[counter - 1]; i <= sp[counter - 1]; i++) sum += en[i];
        mean = sum / (sp[counter - 1] - st[counter - 1] + 1);
    }
    int bw = counter;
    if (bw < f) {
        float thr = (float)lc3t_bw_brick_thr[counter];
        int stop = st[counter], dist = lc3t_bw_brick_dist[counter], brick = 0;
        for (int i = stop; i >= stop - dist; i--) {
            float ediff = 10.0 * m_log10f(en[i - dist + 1] + FLT_EPSILON) - 10.0 * m_log10f(en[i + 1] + FLT_EPSILON);
            if (ediff > thr) { brick = 1; break; }
        }
        if (!brick) bw = f;
    }
    return bw;
}

/* R/sns_compute_scf.c:13-176 -- modifies x (band energies) in place */
static void stage_sns_scf(const lc3o_enc* e, float* x, float* gains, int smooth)
{
    int nb = e->nbands;
    float tmp[64] = {0};
    if (nb < 64) {
        int d = 64 - nb;
        if (d < nb) {
            for (int i = 0, j = 0; i < 2 * d; i += 2, j++) { tmp[i] = x[j]; tmp[i + 1] = x[j]; }
            memcpy(&tmp[2 * d], &x[d], sizeof(float) * (64 - 2 * d));
        } else if (ceil(64.0 / (float)nb) == 4) {
            float ratio = fabsf((float)(1.0 - 32.0 / (float)nb));
            int n4 = round(ratio * nb), n2 = nb - n4, map[64], j = 0;
            for (int i = 1; i <= n4; i++) { map[j] = map[j + 1] = map[j + 2] = map[j + 3] = i; j += 4; }
            for (int i = n4 + 1; i <= n4 + n2; i++) { map[j] = map[j + 1] = i; j += 2; }
            for (int i = 0; i < 64; i++) tmp[i] = x[map[i] - 1];
        } else assert(!"unsupported band count");
        memcpy(x, tmp, sizeof(float) * 64);
        nb = 64;
    }
    float xm[64], xp[64];
    xm[0] = x[0]; memcpy(&xm[1], &x[0], sizeof(float) * 63);
    memcpy(&xp[0], &x[1], sizeof(float) * 63); xp[63] = x[63];
    for (int i = 0; i < 64; i++) x[i] = 0.5 * x[i] + 0.25 * xm[i] + 0.25 * xp[i];
    for (int i = 0; i < 64; i++) x[i] = x[i] * e->sns_preemph[i];
    float sum = 0;
    for (int i = 0; i < 64; i++) sum += x[i];
    float mean = sum / (float)64;
    float nf = mean * powf(10.0, -40.0 / 10.0);
    { float fl = powf(2.0, -32.0); nf = nf > fl ? nf : fl; }
    for (int i = 0; i < 64; i++) if (x[i] < nf) x[i] = nf;
    float xl[64], xl4[16];
    for (int i = 0; i < 64; i++) xl[i] = m_log2f(x[i]) / 2.0;
    static const float W[6] = {1.0 / 12.0, 2.0 / 12.0, 3.0 / 12.0, 3.0 / 12.0, 2.0 / 12.0, 1.0 / 12.0};
    for (int n = 0; n < 16; n++) {
        float t[6];
        if (n == 0) { t[0] = xl[0]; memcpy(&t[1], &xl[0], sizeof(float) * 5); }
        else if (n == 15) { memcpy(t, &xl[59], sizeof(float) * 5); t[5] = xl[63]; }
        else memcpy(t, &xl[n * 4 - 1], sizeof(float) * 6);
        sum = 0;
        for (int i = 0; i < 6; i++) sum += t[i] * W[i];
        xl4[n] = sum;
    }
    sum = 0;
    for (int i = 0; i < 16; i++) sum += xl4[i];
    mean = sum / ((float)nb / 4.0);
    for (int i = 0; i < 16; i++) gains[i] = e->sns_damping * (xl4[i] - mean);
    if (smooth) {
        float g[16];
        g[0] = (gains[0] + gains[1] + gains[2]) / 3.0;
        g[1] = (gains[0] + gains[1] + gains[2] + gains[3]) / 4.0;
        for (int i = 2; i < 14; i++) g[i] = (gains[i - 2] + gains[i - 1] + gains[i] + gains[i + 1] + gains[i + 2]) / 5.0;
        g[14] = (gains[12] + gains[13] + gains[14] + gains[15]) / 4.0;
        g[15] = (gains[13] + gains[14] + gains[15]) / 3.0;
        sum = 0;
        for (int i = 0; i < 16; i++) sum += g[i];
        mean = sum / (float)16;
        for (int i = 0; i < 16; i++) gains[i] = e->att_damping * (g[i] - mean);
    }
}

/* R/sns_quantize_scf.c:43-136.  y has 17 slots (the reference's zero-input branch writes y[dim]). */
static void pvq_search(const float* x_in, int dim, int pulses, int* y, float* y_norm)
{
    float xabs[16]; int sgn[16];
    float xsum = 0, yy = 0, xy = 0;
    const float eps = powf(2, -24);
    if (pulses == 0) return;
    for (int i = 0; i < dim; i++) xabs[i] = fabs(x_in[i]);
    for (int i = 0; i < dim; i++) sgn[i] = x_in[i] >= 0 ? 1 : -1;
    for (int i = 0; i < dim; i++) xsum += xabs[i];
    if (xsum > eps) {
        int tot = 0;
        float proj = (pulses - 1) / xsum;
        for (int i = 0; i < dim; i++) {
            y[i] = floor(xabs[i] * proj);
            tot += y[i];
            yy = yy + y[i] * y[i];
            xy = xy + xabs[i] * y[i];
        }
        yy = yy * 0.5;
        while (tot < pulses) {
            int imax = 0; float cnum = -powf(2, 15), cden = 0;
            yy = yy + 0.5;
            for (int i = 0; i < dim; i++) {
                float a = xy + xabs[i]; a = a * a;
                float b = yy + y[i];
                if (a * cden > b * cnum) { cnum = a; cden = b; imax = i; }
            }
            xy = xy + xabs[imax]; yy = yy + y[imax]; y[imax] = y[imax] + 1; tot++;
        }
        yy = yy * 2.0;
    } else {
        if (dim > 1) { y[0] = pulses / 2; y[dim] = -(pulses - pulses / 2); yy = y[0] * y[0] + y[dim] * y[dim]; }
        else { y[1] = pulses; yy = pulses * pulses; }
    }
    float g = 1.0 * 1.0 / sqrtf(yy);
    for (int i = 0; i < dim; i++) { y[i] = y[i] * sgn[i]; y_norm[i] = y[i] * g; }
}

/* R/sns_quantize_scf.c:138-163 */
static void mpvq_index(const int* pulses, int len, int* ls, int* idx)
{
    int k = 0; *ls = -1; *idx = 0;
    for (int pos = len - 1; pos >= 0; pos--) {
        if (*ls >= 0 && pulses[pos] != 0) *idx = 2 * (*idx) + *ls;
        if (pulses[pos] > 0) *ls = 0;
        if (pulses[pos] < 0) *ls = 1;
        *idx = *idx + (int)lc3t_mpvq_offs[(len - pos - 1) * 11 + k];
        k += abs(pulses[pos]);
    }
}

/* R/sns_quantize_scf.c:19-41 */
static void idct2_16(const lc3o_enc* e, const float* in, float* out)
{
    float n1 = sqrtf(2.0 / (float)16), n2 = 1.0 / (sqrtf(2.0));
    for (int i = 0; i < 16; i++) {
        float sum = 0;
        for (int j = 0; j < 16; j++) {
            float t = in[j] * e->idct_cos[i][j];
            if (j == 0) t *= n2;
            sum += t;
        }
        out[i] = n1 * sum;
    }
}

/* R/dct4.c:28-48 with the 16-point DFT */
static void dct2_16(const lc3o_enc* e, const float* in, float* out)
{
    float z[32];
    for (int i = 0; i < 8; i++) {
        z[2 * i] = in[2 * i]; z[2 * i + 1] = 0;
        z[2 * (15 - i)] = in[2 * i + 1]; z[2 * (15 - i) + 1] = 0;
    }
    dft16(z);
    for (int i = 0; i < 16; i++) out[i] = z[2 * i] * e->dct2_tw[i].r - z[2 * i + 1] * e->dct2_tw[i].i;
    out[0] /= sqrtf(2);
}

/* R/sns_quantize_scf.c:165-430 */
static void stage_sns_vq(const lc3o_enc* e, const float* env, int* index, float* envq)
{
    float st1[16], tgt_pre[16], tgt[16];
    int idx = 0;
    for (int sec = 0; sec < 2; sec++) {
        const float* cb = sec ? lc3t_sns_hf : lc3t_sns_lf;
        float best = powf(2, 100);
        for (int c = 0; c < 32; c++) {
            float sum = 0;
            for (int i = 0; i < 8; i++) sum += (env[8 * sec + i] - cb[c * 8 + i]) * (env[8 * sec + i] - cb[c * 8 + i]);
            if (sum < best) { best = sum; idx = c; }
        }
        index[sec] = idx;
        for (int i = 0; i < 8; i++) st1[8 * sec + i] = cb[idx * 8 + i];
    }
    for (int i = 0; i < 16; i++) tgt_pre[i] = env[i] - st1[i];
    dct2_16(e, tgt_pre, tgt);

    int pA[17] = {0}, pB[17] = {0}, yC[17] = {0}, pulses[16] = {0};
    float nA[16] = {0}, nB[16] = {0}, yCn[16], normZero[16] = {0}, v[6][16];
    pvq_search(tgt, 10, 10, pA, nA);
    pvq_search(&tgt[10], 6, 1, pB, nB);
    memcpy(yC, pA, sizeof(int) * 10); memcpy(&yC[10], pB, sizeof(int) * 6);
    float sum = 0;
    for (int i = 0; i < 16; i++) sum += yC[i] * yC[i];
    float gf = 1.0 / sqrtf(sum);
    for (int i = 0; i < 16; i++) yCn[i] = yC[i] * gf;
    memcpy(normZero, nA, sizeof(float) * 10);
    for (int i = 0; i < 16; i++) { v[0][i] = lc3t_sns_gain_reg[0] * yCn[i]; v[1][i] = lc3t_sns_gain_reg[1] * yCn[i]; }
    for (int i = 0; i < 16; i++) for (int k = 0; k < 4; k++) v[2 + k][i] = lc3t_sns_gain_reg_lf[k] * normZero[i];
    float min_err = powf(2, 15);
    for (int i = 0; i < 6; i++) {
        sum = 0;
        for (int j = 0; j < 16; j++) sum += (tgt[j] - v[i][j]) * (tgt[j] - v[i][j]);
        if (sum < min_err) { min_err = sum; idx = i; }
    }
    for (int i = 0; i < 16; i++) yCn[i] = v[idx][i] / lc3t_sns_gain_q[idx];
    float glob = lc3t_sns_gain_q[idx];
    float split[16], st2[16] = {0};
    idct2_16(e, yCn, split);
    sum = 0;
    for (int i = 0; i < 16; i++) { float d = tgt_pre[i] - glob * split[i]; sum += d * d; }
    float e_split = sum, e_sofar = powf(2, 15);
    if (e_split < e_sofar) {
        if (idx <= 1) { index[2] = 0; index[3] = idx; memcpy(pulses, yC, sizeof pulses); }
        else { index[2] = 1; index[3] = idx - 2; memcpy(pulses, pA, sizeof(int) * 10); }
        for (int i = 0; i < 16; i++) st2[i] = glob * split[i];
        e_sofar = e_split;
    }
    /* outlier near: K=8 over all 16; outlier far: K=6 */
    for (int mode = 2; mode <= 3; mode++) {
        float pre[16] = {0}, sub[16];
        memset(yC, 0, sizeof yC);   /* note: the reference reuses yC without clearing; every slot < dim is overwritten */
        pvq_search(tgt, 16, mode == 2 ? 8 : 6, yC, pre);
        idct2_16(e, pre, sub);
        const float* gt = mode == 2 ? lc3t_sns_gain_near : lc3t_sns_gain_far;
        int ng = mode == 2 ? 4 : 8;
        min_err = powf(2, 15);
        for (int i = 0; i < ng; i++) {
            float g = gt[i];
            sum = 0;
            for (int j = 0; j < 16; j++) sum += (tgt_pre[j] - g * sub[j]) * (tgt_pre[j] - g * sub[j]);
            if (sum < min_err) { idx = i; min_err = sum; glob = g; }
        }
        if (min_err < e_sofar) {
            index[2] = mode; index[3] = idx;
            for (int i = 0; i < 16; i++) st2[i] = glob * sub[i];
            memcpy(pulses, yC, sizeof pulses);
            e_sofar = min_err;
        }
    }
    if (index[2] < 2) mpvq_index(pulses, 10, &index[4], &index[5]); else mpvq_index(pulses, 16, &index[4], &index[5]);
    if (index[2] == 0) { int a, b; mpvq_index(&pulses[10], 6, &a, &b); index[6] = b * 2 + a; }
    else if (index[2] == 2) index[6] = -1; else index[6] = -2;
    for (int i = 0; i < 16; i++) envq[i] = st1[i] + st2[i];
}

/* R/sns_interpolate_scf.c:13-89 (encoder side) */
static void stage_sns_interp(const lc3o_enc* e, const float* g, float* gi)
{
    float tmp[80] = {0};
    gi[0] = g[0]; gi[1] = g[0];
    for (int n = 0; n <= 14; n++) {
        gi[n * 4 + 2] = g[n] + (g[n + 1] - g[n]) / 8.0;
        gi[n * 4 + 3] = g[n] + 3.0 * (g[n + 1] - g[n]) / 8.0;
        gi[n * 4 + 4] = g[n] + 5.0 * (g[n + 1] - g[n]) / 8.0;
        gi[n * 4 + 5] = g[n] + 7.0 * (g[n + 1] - g[n]) / 8.0;
    }
    gi[62] = g[15] + (g[15] - g[14]) / 8.0;
    gi[63] = g[15] + 3.0 * (g[15] - g[14]) / 8.0;
    int nb = e->nbands;
    if (nb < 64) {
        int d = 64 - nb;
        if (d < 32) {
            for (int n = 0, i = 0; n < 2 * d; n += 2, i++) tmp[i] = (gi[n] + gi[n + 1]) / 2.0;
            for (int n = 1; n < d; n++) gi[n] = gi[2 * n];
            for (int n = 2 * d; n < 64; n++) gi[n - d] = gi[n];
            memcpy(gi, tmp, sizeof(float) * d);
        } else if (ceil(64.0 / (float)nb) == 4) {
            float ratio = fabsf((float)(1.0 - 32.0 / (float)nb));
            int n4 = round(ratio * nb);
            for (int i = 0; i < n4; i++) tmp[i] = (gi[4 * i] + gi[4 * i + 1] + gi[4 * i + 2] + gi[4 * i + 3]) / 4.0;
            for (int i = 0; i < nb - n4; i++) tmp[n4 + i] = (gi[4 * n4 + 2 * i] + gi[4 * n4 + 2 * i + 1]) / 2.0;
            memcpy(gi, tmp, sizeof(float) * nb);
        } else assert(!"unsupported band count");
    }
    for (int n = 0; n < nb; n++) gi[n] = -gi[n];
    for (int n = 0; n < nb; n++) gi[n] = m_powf(2, gi[n]);
}

/* R/tns_coder.c:41-89 */
static void levinson(const float* r, float* a, float* rc, float* err, int len)
{
    float buf[16];
    float g = r[1] / r[0];
    a[0] = g;
    float v = (1.0 - g * g) * r[0];
    rc[0] = -g;
    for (int t = 1; t < len; t++) {
        memset(buf, 0, sizeof(float) * (len + 1));
        float sum = 0;
        for (int i = 1; i <= t; i++) sum += a[i - 1] * r[i];
        g = (r[t + 1] - sum) / v;
        for (int i = t - 1, j = 1; i >= 0; i--, j++) buf[j] = a[j - 1] - g * a[i];
        memcpy(&a[1], &buf[1], sizeof(float) * len);
        a[0] = g;
        v = v * (1 - g * g);
        rc[t] = -g;
    }
    a[0] = 1;
    for (int i = len - 1, j = 1; i >= 0; i--, j++) buf[j] = -a[i];
    memcpy(&a[1], &buf[1], sizeof(float) * (len - 1));
    a[len] = rc[len - 1];
    *err = v;
}

/* R/tns_coder.c:91-155 (poly2rc with levdown); only reached with LPC weighting (low bit rates) */
static void poly_to_rc(float* a, float* out, int len)
{
    const int len0 = len;
    float buf[9] = {0};
    memset(out, 0, sizeof(float) * (len - 1));
    { float a0 = a[0]; for (int i = 0; i < len; i++) { a[i] = a[i] / a0; a0 = a[0]; } }
    out[len - 1] = a[len - 1];
    for (int k = len - 2; k >= 0; k--) {
        /* levdown */
        float t0[8] = {0}, t2[8] = {0};
        memcpy(t0, &a[1], sizeof(float) * (len - 1));
        int l = len - 1;
        float knxt = t0[l - 1];
        l = l - 1;
        for (int i = l - 1, j = 0; i >= 0; i--, j++) t2[j] = knxt * t0[i];
        buf[0] = 1;
        for (int i = 0; i < l; i++) buf[i + 1] = (t0[i] - t2[i]) / (1.0 - (fabsf(knxt)) * (fabsf(knxt)));
        len = l + 1;
        out[k] = buf[len - 1];
        memcpy(a, buf, sizeof(float) * len);
    }
    for (int i = 0; i < len0 - 1; i++) out[i] = out[i + 1];
}

/* R/tns_coder.c:170-362 */
static void stage_tns(const lc3o_enc* e, const chan_t* s, float* x, int bw_idx, int bw_bin, int* order_out, int* rc_idx, int* nfilt_out, int* bits_out)
{
    int fs = e->fs, N = e->N, nBits = s->total_bits, dms = e->dms;
    int numfilters = (fs >= 32000 && dms >= 50) ? 2 : 1;
    int start[2] = {0}, stop[2] = {0};
    if (N > 40 * ((float)(dms) / 10.0)) { N = 40 * ((float)(dms) / 10.0); fs = 40000; }
    start[0] = (600 * N * 2 / fs) + 1;
    if (numfilters == 1) stop[0] = N; else { start[1] = N / 2 + 1; stop[0] = N / 2; stop[1] = N; }
    int maxOrder = dms == 100 ? 8 : 4; float nSub = dms == 100 ? 3.0 : 2.0;
    float minPGfac = 0.85, maxPG = 2, minPG = 1.5;
    const uint16_t* obits = &lc3t_tns_order_bits[8];
    if ((dms >= 50 && nBits >= 48 * ((float)dms / 10.0)) || dms == 25) { maxPG = minPG; obits = &lc3t_tns_order_bits[0]; }
    if (bw_idx >= 3 && numfilters == 2) { start[1] = bw_bin / 2 + 1; stop[0] = bw_bin / 2; stop[1] = bw_bin; }
    else { numfilters = 1; stop[0] = bw_bin; }
    int bits = 0;
    float st[9] = {0}, rc[8] = {0};
    for (int f = 0; f < numfilters; f++) {
        float r[9] = {0}, a[16] = {0}, rcu[16] = {0}, err = 0;
        float sublen = ((float)stop[f] + 1.0 - (float)start[f]) / nSub;
        for (int sub = 1; sub <= nSub; sub++) {
            int lo = floor(sublen * (sub - 1)) + start[f] - 1;
            int hi = floor(sublen * sub) + start[f] - 1;
            float sum = 0;
            for (int i = lo; i < hi; i++) sum += x[i] * x[i];
            if (sum == 0) { memset(r, 0, sizeof r); r[0] = 1; break; }
            int n = hi - lo;
            for (int k = 0; k <= maxOrder; k++) {           /* xcorr R/tns_coder.c:18-39: zero-padded terms add +-0 */
                float acc = 0;
                for (int i = 0; i < n; i++) acc += x[lo + i] * (i >= k ? x[lo + i - k] : 0.0f);
                r[k] = r[k] + acc / sum;
            }
        }
        for (int i = 0; i <= maxOrder; i++) r[i] = r[i] * lc3t_tns_lagwin[i];
        levinson(r, a, rcu, &err, maxOrder);
        float predGain = r[0] / err;
        int tns = predGain > minPG;
        bits++;
        int ord = 0, idx_tmp[8] = {0};
        if (tns) {
            if (predGain < maxPG) {
                float alpha = (maxPG - predGain) * (minPGfac - 1.0) / (maxPG - minPG) + 1.0;
                for (int i = 0; i <= maxOrder; i++) a[i] = a[i] * m_powf(alpha, i);
                poly_to_rc(a, rcu, maxOrder + 1);
            }
            for (int i = 0; i < maxOrder; i++) {
                int ret = 0;
                for (int q = 0; q < 17; q++) if (rcu[i] <= lc3t_tns_rc_thr[q + 1] && rcu[i] > lc3t_tns_rc_thr[q]) ret = q;
                idx_tmp[i] = ret;
            }
            for (int i = 0; i < maxOrder; i++) { rc[i] = lc3t_tns_rc_pts[idx_tmp[i]]; if (rc[i] != 0) ord = i + 1; }
            /* ord == 0 would index order_tmp[-1] in the reference (undefined); treat as filter off */
            if (ord == 0) tns = 0;
        }
        order_out[f] = 0;
        if (tns) {
            order_out[f] = ord;
            int tmp = obits[ord - 1];
            for (int i = 0; i < ord; i++) tmp += lc3t_tns_coef_bits[i * 17 + idx_tmp[i]];
            bits = bits + ceil((float)tmp / 2048.0);
            for (int i = 0; i < ord; i++) rc_idx[f * 8 + i] = idx_tmp[i];
            for (int i = start[f]; i <= stop[f]; i++) {
                float sv = x[i - 1], save = sv;
                for (int j = 0; j < ord - 1; j++) {
                    float t = rc[j] * sv + st[j];
                    sv += rc[j] * st[j];
                    st[j] = save; save = t;
                }
                sv += rc[ord - 1] * st[ord - 1];
                st[ord - 1] = save;
                x[i - 1] = sv;
            }
        }
    }
    *nfilt_out = numfilters; *bits_out = bits;
}

/* R/estimate_global_gain.c:30-137 */
static void stage_gain_estimate(const lc3o_enc* e, chan_t* s, const float* x, int nbitsSQ, float* gain, int* qgain, int* qmin)
{
    const int lg = e->ylen, off = s->gg_off;
    float en[LC3O_MAX_N / 4] = {0}, reg_val = 0, ind = 0, ind_min = 0;
    if (s->mem_target_bits < 0) s->tbits_off = 0;
    else {
        float v = s->tbits_off + s->mem_target_bits - s->mem_spec_bits;
        v = -40 > v ? -40 : v; v = 40 < v ? 40 : v;
        s->tbits_off = 0.8 * s->tbits_off + 0.2 * v;
    }
    s->mem_target_bits = nbitsSQ;
    nbitsSQ = nbitsSQ + round(s->tbits_off);
    float x_max = 0;
    for (int i = 0; i < lg; i++) { float t = fabsf(x[i]); if (t > x_max) x_max = t; }
    if (e->hrmode && s->reg_bits > 0) {
        float M0 = 1e-5, M1 = 1e-5, thresh = 2 * e->frame_ms;
        for (int i = 0; i < lg; i++) { M0 += fabs(x[i]); M1 += i * fabs(x[i]); }
        float q = M1 / M0;
        float rB = 8 * (1 - (q < thresh ? q : thresh) / thresh);
        reg_val = x_max * m_powf(2, -s->reg_bits - rB);
    }
    if (x_max == 0) { ind_min = off; ind = 0; s->mem_target_bits = -1; }
    else {
        float g_min = e->hrmode == 1 ? x_max / (32768 * 256 - 2) : x_max / (32768 - 0.375);
        ind_min = ceil(28.0 * m_log10f(g_min));
        for (int i = 0, j = 0; i < lg; i += 4, j++) {
            float t = x[i] * x[i];
            t += x[i + 1] * x[i + 1]; t += x[i + 2] * x[i + 2]; t += x[i + 3] * x[i + 3];
            en[j] = (28.0 / 20.0) * (7 + 10.0 * m_log10f(t + reg_val + powf(2, -31)));
        }
        float target = (28.0 / 20.0) * (1.4) * nbitsSQ, fac = 256;
        int offset = 255 + off;
        for (int it = 0; it < 8; it++) {
            fac = fac * 0.5; offset = offset - fac;
            float ener = 0; int iszero = 1;
            for (int j = lg / 4 - 1; j >= 0; j--) {
                float t = en[j] - offset;
                if (t < (7.0) * (28.0 / 20.0)) { if (iszero == 0) ener = ener + (2.7) * (28.0 / 20.0); }
                else {
                    if (t > (50.0) * (28.0 / 20.0)) ener = ener + 2.0 * t - (50.0) * (28.0 / 20.0);
                    else ener = ener + t;
                    iszero = 0;
                }
            }
            if (ener > target && iszero == 0) offset = offset + fac;
        }
        if (offset < ind_min) s->mem_target_bits = -1;
        ind = (ind_min > offset ? ind_min : offset) - off;
    }
    *qmin = ind_min; *qgain = ind;
    *gain = powf(10.0, ((ind + off) / 28.0));   /* product: host-side libm table, both math modes */
}

/* R/quantize_spec.c:26-197 */
static void stage_quantize(const lc3o_enc* e, const chan_t* s, const float* x, float gain, int* xq, int* nbits_o, int* nbits2_o,
                           int* lastnz_o, int* cdata, int* lsb_o, int mode, int target)
{
    const int nt = e->ylen, fs = e->fs, tb = s->total_bits;
    const float offs = e->hrmode ? 0.5 : 0.375;
    int rate = 0, lastnz = 1, lastnz2, nbits = 0, nbits2 = 0, nlsb = 0, c = 0;
    for (int i = 0; i < nt; i++) {
        int sg = x[i] > 0 ? 1 : x[i] < 0 ? -1 : 0;
        xq[i] = trunc(x[i] / gain + offs * sg);
    }
    if ((fs < 48000 && tb > 320 + (fs / 8000 - 2) * 160) || (fs == 48000 && tb > 800)) rate = 512;
    if (mode == 0 && ((fs < 48000 && tb >= 640 + (fs / 8000 - 2) * 160) || (fs == 48000 && tb >= 1120))) mode = 1;
    for (int i = nt - 2; i >= 2; i -= 2) if (xq[i + 1] != 0 || xq[i] != 0) { lastnz = i + 1; break; }
    lastnz2 = mode < 0 ? lastnz + 1 : 2;
    for (int k = 0; k < lastnz; k += 2) {
        int t = c + rate;
        if (k > nt / 2) t += 256;
        cdata[0] = t;
        int a = abs(xq[k]), b = abs(xq[k + 1]), m = IMAX(a, b);
        int maxlev = m == 0 ? -1 : (int)floor(m_log2f(IMAX(m, 3))) - 1;
        cdata[1] = maxlev;
        if (mode <= 0) { nbits += IMIN(a, 1) * 2048; nbits += IMIN(b, 1) * 2048; }
        int lev = 0;
        while (IMAX(a, b) >= 4) {
            int pki = lc3t_ac_ctx_lut[t + lev * 1024];
            nbits += lc3t_ac_bits[pki * 17 + 16];
            if (lev == 0 && mode > 0) nlsb += 2; else nbits += 2 * 2048;
            a >>= 1; b >>= 1; lev = IMIN(lev + 1, 3);
        }
        int pki = lc3t_ac_ctx_lut[t + lev * 1024], sym = a + 4 * b;
        cdata[2] = sym; cdata += 3;
        nbits += lc3t_ac_bits[pki * 17 + sym];
        if (mode > 0) {
            int am = abs(xq[k]), bm = abs(xq[k + 1]);
            if (lev > 0) {
                am >>= 1; bm >>= 1;
                if (am == 0 && xq[k] != 0) nlsb++;
                if (bm == 0 && xq[k + 1] != 0) nlsb++;
            }
            nbits += IMIN(am, 1) * 2048; nbits += IMIN(bm, 1) * 2048;
        }
        if (mode >= 0 && (abs(xq[k]) != 0 || abs(xq[k + 1]) != 0) && nbits <= target * 2048) { lastnz2 = k + 2; nbits2 = nbits; }
        lev = lev - 1;
        if (lev <= 0) t = 1 + (a + b) * (lev + 2); else t = 13 + lev;
        c = (c & 15) * 16 + t;
    }
    nbits = ceil((float)nbits / 2048.0);
    if (mode >= 0) nbits2 = ceil((float)nbits2 / 2048.0); else nbits2 = nbits;
    if (mode > 0) { nbits += nlsb; nbits2 += nlsb; }
    for (int i = lastnz2; i <= lastnz; i++) xq[i] = 0;
    *lsb_o = (mode > 0 && nbits > target) ? 1 : 0;
    *lastnz_o = lastnz2; *nbits_o = nbits; *nbits2_o = nbits2;
}

/* R/adjust_global_gain.c:13-50 */
static void stage_gain_adjust(const lc3o_enc* e, const chan_t* s, int* gg, int gg_min, float* gain, int target, int nBits, int* change)
{
    const int f = e->fs_idx, off = s->gg_off;
    float delta;
    if (nBits < lc3t_gg_p1[f]) delta = (nBits + 48.0) / 16.0;
    else if (nBits < lc3t_gg_p2[f]) delta = (nBits + lc3t_gg_d[f]) * lc3t_gg_c[f];
    else if (nBits < lc3t_gg_p3[f]) delta = nBits / 48.0;
    else delta = lc3t_gg_p3[f] / 48.0;
    delta = round(delta);
    int delta2 = delta + 2;
    *change = 0;
    if (*gg == 255 && nBits > target) *change = 1;
    if ((*gg < 255 && nBits > target) || (*gg > 0 && nBits < target - delta2)) {
        if (nBits < target - delta2) *gg = *gg - 1;
        else if (*gg == 254 || nBits < target + delta) *gg = *gg + 1;
        else *gg = *gg + 2;
        *gg = IMAX(*gg, gg_min - off);
        *gain = powf(10, (float)(*gg + off) / 28);       /* product: host-side libm table */
        *change = 1;
    }
}

/* R/noise_factor.c:13-108 */
static int stage_noise_factor(const lc3o_enc* e, const chan_t* s, const float* x, const int* xq, float gg, int bw_bin)
{
    int zl[LC3O_MAX_N], nz = 0, sumz = 0;
    const int width = e->dms == 100 ? 8 : 4, first = e->dms == 100 ? 24 : e->dms == 50 ? 12 : 6;
    for (int k = first; k < bw_bin; k++) {
        int allz = 1, lo = k - (width - 2) / 2, hi = IMIN(bw_bin - 1, k + (width - 2) / 2);
        for (int i = lo; i <= hi; i++) if (xq[i] != 0) allz = 0;
        if (allz) zl[nz++] = k + 1;
    }
    for (int i = 0; i < nz; i++) sumz += zl[i];
    float fac = 0, mean = 0;
    if (sumz > 0) { for (int j = 0; j < nz; j++) mean += fabsf(x[zl[j] - 1] / gg); fac = mean / nz; }
    if (s->nbytes <= 20 && e->dms == 100 && nz > 0) {
        int m = sumz / nz, j = 0, k = 0;
        float m1 = 0, m2 = 0;
        for (int i = 0; i < nz; i++) { if (zl[i] <= m) { m1 += fabsf(x[zl[i] - 1]) / gg; j++; } }
        for (int i = 0; i < nz; i++) { if (zl[i] > m) { m2 += fabsf(x[zl[i] - 1]) / gg; k++; } }
        float n1 = m1 / j, n2 = m2 / k;
        fac = n1 < n2 ? n1 : n2;
    }
    float idx = round(8 - 16 * fac);
    { float t = idx > 0 ? idx : 0; idx = t < 7 ? t : 7; }
    return (int)idx;
}

/* R/residual_coding.c:13-75 */
static int stage_residual(const lc3o_enc* e, float* x, const int* xq, float gain, int targetBits, int nBits, uint8_t* res)
{
    int nzi[LC3O_MAX_N], nnz = 0, n = 0, iter = 0;
    const int iter_max = e->hrmode ? 20 : 1;
    int m = targetBits - nBits + 4;
    if (e->hrmode) m += 10;
    float offset = .25;
    memset(res, 0, 625);
    for (int k = 0; k < e->ylen; k++) if (xq[k]) nzi[nnz++] = k;
    while (iter < iter_max && n < m) {
        for (int k = 0; k < nnz && n < m; k++, n++) {
            int id = nzi[k];
            if (x[id] >= (float)xq[id] * gain) { res[n >> 3] |= 1 << (n & 7); x[id] -= gain * offset; }
            else { res[n >> 3] &= ~(1 << (n & 7)); x[id] += gain * offset; }
        }
        iter++; offset *= .5;
    }
    return n;
}

/* ---- bitstream: R/enc_entropy.c:13-115 and R/ari_codec.c:511-800 ---- */
typedef struct { uint8_t* p; int bp_side, mask_side; int bp, low, range, cache, carry, carry_count; } bitw_t;

static void put_bit_back(bitw_t* w, int bit)                     /* R/enc_entropy.c:101-115 */
{
    if (bit == 0) w->p[w->bp_side] &= (255 - w->mask_side); else w->p[w->bp_side] |= w->mask_side;
    if (w->mask_side == 128) { w->mask_side = 1; w->bp_side--; } else w->mask_side *= 2;
}
static void put_uint_back(bitw_t* w, int val, int nbits) { for (int k = 0; k < nbits; k++) { put_bit_back(w, val & 1); val = val / 2; } }

static void stage_side_info(const lc3o_enc* e, const chan_t* s, bitw_t* w, int bw_idx, int lastnz, int lsb, int gg, int nfilt,
                            const int* tns_order, const int* ltpf, const int* scf, int fac_ns)
{
    static const int gain_msb_bits[4] = {1, 1, 2, 2}, gain_lsb_bits[4] = {0, 1, 0, 1};
    w->bp_side = s->nbytes - 1; w->mask_side = 1;
    if (e->bw_bits > 0) put_uint_back(w, bw_idx, e->bw_bits);
    put_uint_back(w, lastnz / 2 - 1, (int)ceil(log2f(e->ylen / 2)));
    put_bit_back(w, lsb);
    put_uint_back(w, gg, 8);
    for (int i = 0; i < nfilt; i++) put_bit_back(w, IMIN(1, tns_order[i]));
    put_bit_back(w, ltpf[0]);
    put_uint_back(w, scf[0], 5); put_uint_back(w, scf[1], 5);
    int sub_msb = scf[2] / 2, sub_lsb = scf[2] & 1;
    put_bit_back(w, sub_msb);
    int g_msb = scf[3] >> gain_lsb_bits[scf[2]], g_lsb = scf[3] & 1;
    put_uint_back(w, g_msb, gain_msb_bits[scf[2]]);
    put_bit_back(w, scf[4]);
    if (sub_msb == 0) {
        int t = sub_lsb == 0 ? scf[6] + 2 : g_lsb;
        t = t * 2390004 + scf[5];
        put_uint_back(w, t, 25);
    } else {
        int t = scf[5];
        if (sub_lsb != 0) t = 2 * t + g_lsb + 15158272;
        put_uint_back(w, t, 24);
    }
    if (ltpf[0] == 1) { put_uint_back(w, ltpf[1], 1); put_uint_back(w, ltpf[2], 9); }
    put_uint_back(w, fac_ns, 3);
}

static void ac_shift(bitw_t* w)                                 /* R/ari_codec.c:531-553 */
{
    if (w->low < 16711680 || w->carry == 1) {
        if (w->cache >= 0) { w->p[w->bp] = w->cache + w->carry; w->bp++; }
        while (w->carry_count > 0) { w->p[w->bp] = (w->carry + 255) & 255; w->bp++; w->carry_count--; }
        w->cache = w->low >> 16; w->carry = 0;
    } else w->carry_count++;
    w->low = (int)(((unsigned)w->low << 8) & 0xFFFFFFu);       /* unsigned: the reference's int shift overflows for low >= 2^23 */
}
static void ac_encode(bitw_t* w, int freq, int cum)             /* R/ari_codec.c:511-529 */
{
    int r = w->range >> 10;
    w->low += r * cum;
    if ((w->low >> 24) == 1) w->carry = 1;
    w->low &= 0xFFFFFF;
    w->range = r * freq;
    while (w->range < 65536) { w->range <<= 8; ac_shift(w); }
}
static int flog2_of(int v) { return (int)floor(log2f(v)); }     /* R/ari_codec.c:577,765 use floor(log2f(int)) */
static void ac_finish(bitw_t* w)                                /* R/ari_codec.c:573-647 */
{
    int bits = 24 - flog2_of(w->range);
    int mask = 0xFFFFFF >> bits, val = w->low + mask, over1 = val >> 24;
    val &= 0xFFFFFF;
    int high = w->low + w->range, over2 = high >> 24;
    high &= 0xFFFFFF;
    val &= (0xFFFFFF - mask);
    if (over1 == over2) {
        if (val + mask >= high) { bits++; mask >>= 1; val = ((w->low + mask) & 0xFFFFFF) & (0xFFFFFF - mask); }
        if (val < w->low) w->carry = 1;
    }
    w->low = val;
    int b = bits;
    if (bits > 8) { for (; b >= 1; b -= 8) ac_shift(w); } else ac_shift(w);
    bits = b; if (bits < 0) bits += 8;
    int last, nb = bits;
    if (w->carry_count > 0) {
        w->p[w->bp++] = w->cache;
        for (int c = w->carry_count; c >= 2; c--) w->p[w->bp++] = 255;
        last = 255 << (bits - 8);
    } else last = w->cache;
    for (int k = 0, m = 128; k < nb; k++, m >>= 1) {           /* write_uint_forward: ORs the top bits, bp not advanced */
        if ((last & m) == 0) w->p[w->bp] &= (255 - m); else w->p[w->bp] |= m;
    }
}

static void stage_ari(const lc3o_enc* e, const chan_t* s, bitw_t* w, const int* x, const int* tns_order, int nfilt, const int* tns_idx,
                      int lastnz, const int* cdata, const uint8_t* res, int nres, int lsbMode)
{
    int lsbs[LC3O_MAX_N * 2], nl = 0, lsb1 = 0, lsb2 = 0;
    w->bp = 0; w->low = 0; w->range = 0xFFFFFF; w->cache = -1; w->carry = 0; w->carry_count = 0;
    for (int i = 0; i < nfilt; i++) if (tns_order[i] > 0) {
        const uint16_t* oc = &lc3t_tns_order_cum[s->lpc_weighting * 9];
        ac_encode(w, oc[tns_order[i]] - oc[tns_order[i] - 1], oc[tns_order[i] - 1]);
        for (int j = 0; j < tns_order[i]; j++) {
            const uint16_t* cc = &lc3t_tns_coef_cum[j * 18]; int id = tns_idx[i * 8 + j];
            ac_encode(w, cc[id + 1] - cc[id], cc[id]);
        }
    }
    for (int k = 0; k < lastnz; k += 2, cdata += 3) {
        for (int lev = 0; lev < cdata[1]; lev++) {
            int pki = lc3t_ac_ctx_lut[cdata[0] + IMIN(lev, 3) * 1024];
            const uint16_t* cf = &lc3t_ac_cum[pki * 18];
            ac_encode(w, cf[17] - cf[16], cf[16]);
            int b1 = (abs(x[k]) >> lev) & 1, b2 = (abs(x[k + 1]) >> lev) & 1;
            if (lsbMode == 1 && lev == 0) { lsb1 = b1; lsb2 = b2; }
            else { put_bit_back(w, b1); put_bit_back(w, b2); }
        }
        int pki = lc3t_ac_ctx_lut[cdata[0] + IMIN(IMAX(cdata[1], 0), 3) * 1024];
        const uint16_t* cf = &lc3t_ac_cum[pki * 18];
        ac_encode(w, cf[cdata[2] + 1] - cf[cdata[2]], cf[cdata[2]]);
        int a = abs(x[k]), b = abs(x[k + 1]);
        if (lsbMode == 1 && cdata[1] > 0) {
            a >>= 1; lsbs[nl++] = lsb1;
            if (a == 0 && x[k] != 0) lsbs[nl++] = x[k] < 0;
            b >>= 1; lsbs[nl++] = lsb2;
            if (b == 0 && x[k + 1] != 0) lsbs[nl++] = x[k + 1] < 0;
        }
        if (a != 0) put_bit_back(w, x[k] < 0);
        if (b != 0) put_bit_back(w, x[k + 1] < 0);
    }
    int total = s->target_bits_ari;
    int nbits_side = total - (8 * (w->bp_side + 1) + 8 - flog2_of(w->mask_side));
    int nbits_ari = (w->bp + 1) * 8 + 25 - flog2_of(w->range);
    if (w->cache >= 0) nbits_ari += 8;
    if (w->carry_count > 0) nbits_ari += w->carry_count * 8;
    int nres_enc = total - (nbits_side + nbits_ari);
    assert(nres_enc >= 0);
    if (lsbMode == 0) {
        nres_enc = IMIN(nres_enc, nres);
        for (int k = 0; k < nres_enc; k++) put_bit_back(w, (res[k >> 3] >> (k & 7)) & 1);
    } else {
        nres_enc = IMIN(nres_enc, nl);
        for (int k = 0; k < nres_enc; k++) put_bit_back(w, lsbs[k]);
    }
    ac_finish(w);
    (void)e;
}

/* ------------------------------------------------------------------------------------------------ */
/* frame driver: R/enc_lc3_fl.c:13-160                                                               */
/* ------------------------------------------------------------------------------------------------ */
static void encode_channel(lc3o_enc* e, int chn, const void* pcm, int bps, uint8_t* bytes)
{
    chan_t* s = &e->ch[chn];
    lc3o_trace* tr = e->trace ? &e->trace[chn] : NULL;
    const int N = e->N;
    float sin_[LC3O_MAX_N], d[LC3O_MAX_N] = {0}, s12[129 + 8] = {0}, ener[64] = {0}, scf[16], scfq[16], gi[64];
    int q[LC3O_MAX_N] = {0}, tns_idx[16] = {0}, tns_order[2] = {0}, scf_idx[7] = {0}, ltpf[3] = {0};
    static const int zero3[3] = {0, 0, 0}; (void)zero3;
    int cdata[3 * LC3O_MAX_N]; uint8_t res[625];

    memset(bytes, 0, s->nbytes);
    if (bps == 24) for (int i = 0; i < N; i++) sin_[i] = (float)(((const int32_t*)pcm)[i] / powf(2, 8));
    else if (bps == 32) for (int i = 0; i < N; i++) sin_[i] = (float)(((const int32_t*)pcm)[i] / powf(2, 16));
    else for (int i = 0; i < N; i++) sin_[i] = (float)((const int16_t*)pcm)[i];

    stage_mdct(e, s, sin_, d);
    if (tr) memcpy(tr->spec_mdct, d, sizeof(float) * N);
    int len12 = stage_resample(e, s, sin_, s12);
    if (tr) memcpy(tr->s12k8, s12, sizeof(float) * (len12 + 1));
    int T0 = 0, ltpf_bits = 0; float nc = 0;
    stage_olpa(e, s, s12, len12, &T0, &nc);
    stage_ltpf(e, s, s12, len12 + 1, T0, nc, ltpf, &ltpf_bits);
    if (tr) { tr->T0 = T0; tr->normcorr = nc; memcpy(tr->ltpf_param, ltpf, sizeof ltpf); tr->ltpf_bits = ltpf_bits; }
    stage_attack(e, s, sin_);
    if (tr) tr->attack = s->att_flag;
    stage_band_energy(e, d, ener);
    if (tr) memcpy(tr->ener, ener, sizeof ener);
    int bw = (e->fs_idx > 0 && e->hrmode == 0) ? stage_bw_detect(e, ener) : e->fs_idx;
    stage_sns_scf(e, ener, scf, s->att_flag);
    stage_sns_vq(e, scf, scf_idx, scfq);
    stage_sns_interp(e, scfq, gi);
    for (int b = 0, j = 0; b < e->nbands; b++) for (; j < e->bands[b + 1]; j++) d[j] = d[j] * gi[b];   /* R/mdct_shaping.c */
    if (tr) { memcpy(tr->scf, scf, sizeof scf); memcpy(tr->scf_idx, scf_idx, sizeof scf_idx); memcpy(tr->scf_q, scfq, sizeof scfq);
              memcpy(tr->spec_shaped, d, sizeof(float) * N); }
    if (e->bandwidth) {                                                                              /* R/cutoff_bandwidth.c */
        int bin = e->bw_cut_bin;
        if (e->ylen > bin) {
            for (int i = -1; i < 3; i++) d[bin + i] = d[bin + i] * powf(2, -(i + 2));
            for (int i = bin + 3; i < e->ylen; i++) d[i] = 0;
        }
        bw = IMIN(bw, e->bw_index);
    }
    if (tr) tr->bw_idx = bw;
    int nfilt = 0, tns_bits = 0;
    stage_tns(e, s, d, bw, e->cut_bins[bw], tns_order, tns_idx, &nfilt, &tns_bits);
    if (tr) { tr->tns_nfilt = nfilt; memcpy(tr->tns_order, tns_order, sizeof tns_order); memcpy(tr->tns_rc_idx, tns_idx, sizeof tns_idx);
              tr->tns_bits = tns_bits; memcpy(tr->spec_tns, d, sizeof(float) * N); }
    int tbq = s->target_bits_init - (tns_bits + ltpf_bits);
    float gain = 0; int gg = 0, ggmin = 0, nbits = 0, nbits2 = 0, lastnz = 0, lsb = 0, change = 0;
    stage_gain_estimate(e, s, d, tbq, &gain, &gg, &ggmin);
    if (tr) { tr->target_bits_quant = tbq; tr->gain0 = gain; tr->gg_idx0 = gg; tr->gg_min = ggmin; }
    stage_quantize(e, s, d, gain, q, &nbits, &nbits2, &lastnz, cdata, &lsb, -1, tbq);
    s->mem_spec_bits = nbits;
    if (tr) tr->nbits0 = nbits;
    stage_gain_adjust(e, s, &gg, ggmin, &gain, tbq, nbits, &change);
    if (change) stage_quantize(e, s, d, gain, q, &nbits, &nbits2, &lastnz, cdata, &lsb, 0, tbq);
    int fac_ns = stage_noise_factor(e, s, d, q, gain, e->cut_bins[bw]);
    int nres = 0;
    if (lsb == 0) nres = stage_residual(e, d, q, gain, tbq, nbits2, res);
    if (tr) { tr->gain = gain; tr->gg_idx = gg; tr->gain_change = change; tr->nbits = nbits; tr->nbits2 = nbits2; tr->lastnz = lastnz;
              tr->lsb_mode = lsb; memcpy(tr->xq, q, sizeof(int) * N); tr->fac_ns = fac_ns; tr->n_res_bits = nres; }
    bitw_t w; memset(&w, 0, sizeof w); w.p = bytes;
    stage_side_info(e, s, &w, bw, lastnz, lsb, gg, nfilt, tns_order, ltpf, scf_idx, fac_ns);
    if (tr) { tr->bp_side = w.bp_side; tr->mask_side = w.mask_side; }
    stage_ari(e, s, &w, q, tns_order, nfilt, tns_idx, lastnz, cdata, res, nres, lsb);
}

int lc3o_enc_frame(lc3o_enc* e, void** input, int bitdepth, uint8_t* out, int* num_bytes)   /* R/lc3.c:226-234, R/enc_lc3_fl.c:162-174 */
{
    if (!e || !input || !out || !num_bytes) return LC3O_NULL_ERROR;
    for (int ch = 0; ch < e->channels; ch++) if (!input[ch]) return LC3O_NULL_ERROR;
    if (bitdepth != 16 && bitdepth != 24 && bitdepth != 32) return LC3O_ERROR;
    if (!e->win || !e->fft_kind) return LC3O_UNSUPPORTED;
    int total = 0;
    for (int ch = 0; ch < e->channels; ch++) {
        encode_channel(e, ch, input[ch], bitdepth, out);
        out += e->ch[ch].nbytes; total += e->ch[ch].nbytes;
    }
    *num_bytes = total;
    return LC3O_OK;
}

/* B independent streams of `channels` channels each, T frames, PCM [B][T][channels][N] int16 -> out [B][T][stride] (the channel
 * payloads of a frame concatenated, R/enc_lc3_fl.c:167-171).  Test-side convenience: one C call instead of B*T ctypes calls. */
int lc3o_encode_batch16_ch(int samplerate, float frame_ms, int hrmode, int channels, int B, int T, const int* bitrate,
                           const int16_t* pcm, uint8_t* out, int stride)
{
    lc3o_enc* e = (lc3o_enc*)malloc(sizeof *e);
    int rc = e ? LC3O_OK : LC3O_ERROR;
    for (int b = 0; b < B && rc == LC3O_OK; b++) {
        rc = lc3o_enc_init(e, samplerate, channels);
        if (!rc) rc = lc3o_enc_set_frame_ms(e, frame_ms);
        if (!rc) rc = lc3o_enc_set_hrmode(e, hrmode);
        if (!rc) rc = lc3o_enc_set_bitrate(e, bitrate[b]);
        if (rc) break;
        const int N = e->N;
        for (int t = 0; t < T && !rc; t++) {
            void* in[2];
            for (int c = 0; c < channels; c++) in[c] = (void*)(pcm + (((size_t)b * T + t) * channels + c) * N);
            int nb = 0;
            rc = lc3o_enc_frame(e, in, 16, out + ((size_t)b * T + t) * stride, &nb);
        }
    }
    free(e);
    return rc;
}
/* the same for mono streams with a bandwidth plan [B][T]: before frame t of stream b the bandwidth is set to plan[b*T + t] when that is not 0
 * (the switching file of R/codec_exe.c:338-352 as an array; a value the API refuses - R/lc3.c:197 - is ignored like there) */
int lc3o_encode_batch16_bw(int samplerate, float frame_ms, int hrmode, int B, int T, const int* bitrate, const int* bw_plan,
                           const int16_t* pcm, uint8_t* out, int stride)
{
    lc3o_enc* e = (lc3o_enc*)malloc(sizeof *e);
    int rc = e ? LC3O_OK : LC3O_ERROR;
    for (int b = 0; b < B && rc == LC3O_OK; b++) {
        rc = lc3o_enc_init(e, samplerate, 1);
        if (!rc) rc = lc3o_enc_set_frame_ms(e, frame_ms);
        if (!rc) rc = lc3o_enc_set_hrmode(e, hrmode);
        if (!rc) rc = lc3o_enc_set_bitrate(e, bitrate[b]);
        if (rc) break;
        const int N = e->N;
        for (int t = 0; t < T && !rc; t++) {
            if (bw_plan && bw_plan[(size_t)b * T + t]) (void)lc3o_enc_set_bandwidth(e, bw_plan[(size_t)b * T + t]);
            void* in[1] = {(void*)(pcm + ((size_t)b * T + t) * N)};
            int nb = 0;
            rc = lc3o_enc_frame(e, in, 16, out + ((size_t)b * T + t) * stride, &nb);
        }
    }
    free(e);
    return rc;
}
int lc3o_encode_batch16(int samplerate, float frame_ms, int hrmode, int B, int T, const int* bitrate,
                        const int16_t* pcm, uint8_t* out, int stride)
{
    return lc3o_encode_batch16_ch(samplerate, frame_ms, hrmode, 1, B, T, bitrate, pcm, out, stride);
}

/* test hook: the restated forward DFT on its own (tests/test_oracle_vs_ref.py pins it against the reference's LC3_iisfft_apply) */
int lc3o_dft(float* x, int n) { float scratch[2 * LC3O_MAX_N]; return dft_any(x, n, scratch); }

#include "lc3_oracle_dec.inc"
