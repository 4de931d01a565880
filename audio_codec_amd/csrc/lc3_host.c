/* lc3_host.c -- host side (plain C) of the MI355X LC3plus encode engine.
 *
 * Exports the reference's encoder ABI (include/lc3.h: lc3_enc_*, same semantics as R/lc3.c:102-309 with
 * R = LC3plus_ETSI_src_v17171_20200723/src/floating_point) plus the batched extension (include/lc3plus_batch.h).
 * All signal processing happens in the HIP kernels (lc3_kernels.hip) reached through lc3_shim.h; this file only
 * derives configuration (R/setup_enc_lc3.c), builds the init-time tables with the host libm exactly where the
 * reference evaluates them, and moves buffers.  There is no CPU encode path: without a HIP device every encode
 * call fails with LC3_ERROR.
 */
#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/lc3.h"
#include "../../include/lc3plus_batch.h"
#include "lc3_tables.h"
#include "lc3_shim.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#define MAX_CH 2                    /* R/defines.h:121 */
#define IMIN(a, b) ((a) < (b) ? (a) : (b))
#define IMAX(a, b) ((a) > (b) ? (a) : (b))

/* ------------------------------------------------------------------------------------------------ */
/* configuration shared by the single-stream and the batch API                                       */
/* ------------------------------------------------------------------------------------------------ */
typedef struct {
    int fs, fs_in, fs_idx, channels, dms, hrmode;
    float frame_ms;
    int N, ylen, la, nbands, bw_bits, tilt;
    int att_nblocks, att_hang; float att_damping, sns_damping;
    const lc3t_cfg_t* tab;          /* NULL when the frame size has no window/band table */
} geom_t;

static int samplerate_ok(int sr)
{
    switch (sr) { case 8000: case 16000: case 24000: case 32000: case 44100: case 48000: case 96000: return 1; default: return 0; }
}

static void geom_init(geom_t* g, int samplerate, int channels)            /* R/setup_enc_lc3.c:31-70 */
{
    static const int tilts[6] = {14, 18, 22, 26, 30, 34};
    memset(g, 0, sizeof *g);
    g->fs = samplerate == 44100 ? 48000 : samplerate; g->fs_in = samplerate;
    g->fs_idx = g->fs / 10000; if (g->fs_idx > 4) g->fs_idx = 5;
    g->channels = channels; g->dms = 100; g->frame_ms = 10;
    g->tilt = tilts[g->fs_idx];
}

static void geom_update_ex(geom_t* g, int decoder)
{
    if (decoder && g->fs_idx == 5) g->hrmode = 1;                        /* the decoder forces hrmode first: R/setup_dec_lc3.c:76-80 */
    g->N = g->fs / 100;
    if (g->hrmode == 1) { g->ylen = g->N; g->sns_damping = 0.6; }
    else { g->ylen = IMIN(400, g->N); g->sns_damping = 0.85; }
    if (g->fs_idx == 5) g->hrmode = 1;                                    /* reference order kept */
    g->bw_bits = g->hrmode ? 0 : lc3t_bw_bits[g->fs_idx];
    if (g->dms == 100) { g->att_nblocks = 4; g->att_damping = 0.5; g->att_hang = 2; }
    if (g->dms == 25) { g->N >>= 2; g->ylen /= 4; }
    if (g->dms == 50) { g->N >>= 1; g->ylen /= 2; }
    g->tab = NULL; g->nbands = 64; g->la = 0;
    for (int i = 0; i < LC3T_NCFG; i++)
        if (lc3t_cfg[i].valid && lc3t_cfg[i].fs_idx == g->fs_idx && lc3t_cfg[i].dms == g->dms && lc3t_cfg[i].hr == g->hrmode) g->tab = &lc3t_cfg[i];
    if (g->tab) { g->nbands = g->tab->nbands; g->la = g->tab->la_zeros; }
}
static void geom_update(geom_t* g) { geom_update_ex(g, 0); }              /* R/setup_enc_lc3.c:73-193 */

/* the kernels are built for frame lengths up to LC3D_MAX_N = 960 whose N/2-point DFT has a restated kernel (480 = 15x32, 240 = 15x16,
 * 60 = 4x15, and the prime-factor lengths 10 ... 160) and an MDCT overlap memory of at most 600 samples: every operating point of
 * the reference.  N > 480 or a memory > 300 run in the large-layout kernel (lc3_kernels.hip) */
static int fft_supported(int len);
static int geom_supported(const geom_t* g) { return g->tab && g->N <= LC3D_MAX_N && fft_supported(g->N / 2) && g->N - g->la <= LC3D_MEMCAP_BIG; }

/* R/setup_enc_lc3.c:196-375: bitrate -> per-channel budgets.  Returns an LC3_Error. */
static LC3_Error derive_bitrate(const geom_t* g, int bitrate, lc3d_chan* ch /* [channels] */)
{
    int minBR = 0, maxBR = 0;
    if (g->hrmode) {
        switch (g->dms) {
        case 25: maxBR = 672000; minBR = g->fs == 48000 ? 172800 : g->fs == 96000 ? 198400 : -1; break;
        case 50: maxBR = 600000; minBR = g->fs == 48000 ? 148800 : g->fs == 96000 ? 174400 : -1; break;
        case 100: maxBR = 500000; minBR = g->fs == 48000 ? 124800 : g->fs == 96000 ? 149600 : -1; break;
        default: return LC3_HRMODE_ERROR;
        }
        if (minBR < 0) return LC3_HRMODE_ERROR;
    } else {
        minBR = 20 * 8 * (1000 / g->frame_ms) * (g->fs_in == 44100 ? 441. / 480 : 1);
        maxBR = 400 * 8 * (1000 / g->frame_ms) * (g->fs_in == 44100 ? 441. / 480 : 1);
    }
    minBR *= g->channels; maxBR *= g->channels;
    if (bitrate < minBR || bitrate > maxBR) return LC3_BITRATE_ERROR;
    const int totalBytes = bitrate * g->N / (8 * g->fs_in);
    int off = 0;
    for (int c = 0; c < g->channels; c++) {
        lc3d_chan* s = &ch[c];
        const int was_attack = s->attack_handling;
        s->nbytes = totalBytes / g->channels + (c < (totalBytes % g->channels));
        s->out_off = off; off += s->nbytes;
        s->total_bits = s->nbytes << 3;
        s->target_bits_init = s->total_bits - 38 - 8 - 3 - g->bw_bits - (int)ceil(log2f(g->N / 2)) - 2 - 1;
        if (s->total_bits > 1280) s->target_bits_init -= 1;
        if (s->total_bits > 2560) s->target_bits_init -= 1;
        if (g->hrmode) s->target_bits_init -= 1;
        s->lpc_weighting = s->total_bits < 480;
        if (g->frame_ms == 5) s->lpc_weighting = s->total_bits < 240;
        if (g->frame_ms == 2.5) s->lpc_weighting = s->total_bits < 120;
        s->gg_off = -(IMIN(115, s->total_bits / (10 * (g->fs_idx + 1))) + 105 + 5 * (g->fs_idx + 1));
        if (g->frame_ms == 10 && ((g->fs_in >= 44100 && s->nbytes >= 100) || (g->fs_in == 32000 && s->nbytes >= 81)) &&
            s->nbytes < 340 && g->hrmode == 0) s->attack_handling = 1;
        else { s->attack_handling = 0; s->reset_attack = 1; }
        (void)was_attack;
        int bitsTmp = s->total_bits;
        if (g->frame_ms == 2.5) bitsTmp = bitsTmp * 4.0 * (1.0 - 0.4);
        if (g->frame_ms == 5) bitsTmp = bitsTmp * 2 - 160;
        s->ltpf_enable = bitsTmp < 640 + (g->fs_idx - 1) * 80;
        if (g->hrmode) s->ltpf_enable = 0;
        if (g->hrmode && g->fs_idx >= 4) {
            int real_rate = s->nbytes * 8000 / g->frame_ms;
            s->reg_bits = real_rate / 12500;
            if (g->fs_idx == 5) { if (g->frame_ms == 10) s->reg_bits += 2; if (g->frame_ms == 2.5) s->reg_bits -= 6; }
            else { if (g->frame_ms == 2.5) s->reg_bits -= 6; if (g->frame_ms == 10) s->reg_bits += 5; }
        } else s->reg_bits = -1;
    }
    return LC3_OK;
}

/* ---- prime-factor index plan for the 120-point DFT (index logic of R/fft/fft_generic.h:634-699).
 * The recursion is run on slot labels instead of samples: every leaf DFT records where its inputs live in the
 * previous stage's output buffer; its outputs get consecutive slots.  The last stage's scatter is folded in. ---- */
static int pfa_inverse(int a, int b)
{
    int b0 = b, x0 = 0, x1 = 1;
    if (b == 1) return 1;
    while (a > 1) { int q = a / b, t = b; b = a % b; a = t; t = x0; x0 = x1 - q * x0; x1 = t; }
    if (x1 < 0) x1 += b0;
    return x1;
}
typedef struct { uint8_t* src[3]; int count[3]; int leaf[3]; } pfa_rec;
static void pfa_leaf(int* x, int n, pfa_rec* r)
{
    int st = n == r->leaf[0] ? 0 : n == r->leaf[1] ? 1 : 2;
    for (int j = 0; j < n; j++) { r->src[st][r->count[st]] = (uint8_t)x[j]; x[j] = r->count[st]; r->count[st]++; }
}
static void pfa_label(int* x, int length, int* scratch, int nfac, const int* fac, pfa_rec* r)
{
    if (nfac <= 1) { pfa_leaf(x, length, r); return; }
    int* tmp = scratch;
    const int n2 = fac[0], n1 = length / n2, incr = n1 * pfa_inverse(n1, n2);
    int idx = 0, cnt = 0;
    for (int i = 0; i < n1; i++) {
        for (int ii = 0; ii < n2 - 1; ii++) { tmp[cnt++] = x[idx]; idx += incr; if (idx > length) idx -= length; }
        tmp[cnt++] = x[idx]; idx++;
    }
    for (cnt = 0; cnt < length; cnt += n2) pfa_leaf(tmp + cnt, n2, r);
    for (cnt = 0; cnt < n1; cnt++) for (int i = 0; i < n2; i++) x[cnt + i * n1] = tmp[cnt * n2 + i];
    for (cnt = 0; cnt < length; cnt += n1) pfa_label(x + cnt, n1, tmp, nfac - 1, fac + 1, r);
    cnt = 0;
    for (int i = 0; i < n2; i++) {
        idx = i * n1;
        for (int ii = 0; ii < n1; ii++) { tmp[idx] = x[cnt++]; idx += n2; if (idx > length) idx -= length; }
    }
    memcpy(x, tmp, sizeof(int) * length);
}
/* factors: the prime powers of the length in increasing prime order (R/fft/fft_generic.h:89-145 factorize) */
static int pfa_plan(lc3d_plan* p, int len)
{
    static const struct { int len, n, f[3]; } tab[] = {{10, 2, {2, 5, 0}}, {20, 2, {4, 5, 0}}, {30, 3, {2, 3, 5}}, {40, 2, {8, 5, 0}},
                                                        {80, 2, {16, 5, 0}}, {120, 3, {8, 3, 5}}, {160, 2, {32, 5, 0}}};
    if (len == 60) {                                /* 4 x 15 Good-Thomas with the reference's index tables (R/fft/fft_60_128.h:18-23) */
        for (int k = 0; k < 4; k++) for (int l = 0; l < 15; l++) { p->pfa_src[k + 4 * l] = (uint8_t)((45 * k + 16 * l) % 60); p->pfa_dst[k + 4 * l] = (uint8_t)((15 * k + 4 * l) % 60); }
        p->pfa_nst = 0;
        return 1;
    }
    for (unsigned t = 0; t < sizeof tab / sizeof tab[0]; t++) if (tab[t].len == len) {
        int x[LC3D_PFA_STRIDE], scratch[2 * LC3D_PFA_STRIDE];
        pfa_rec r; memset(&r, 0, sizeof r);
        for (int k = 0; k < 3; k++) { r.src[k] = p->pfa_src + LC3D_PFA_STRIDE * k; r.leaf[k] = tab[t].f[k]; p->pfa_rad[k] = tab[t].f[k]; }
        p->pfa_nst = tab[t].n;
        for (int i = 0; i < len; i++) x[i] = i;
        pfa_label(x, len, scratch, tab[t].n, tab[t].f, &r);
        /* x[i] = slot of the last stage holding output bin i  ->  pfa_dst[slot] = i */
        for (int i = 0; i < len; i++) p->pfa_dst[x[i]] = (uint8_t)i;
        return 1;
    }
    return 0;
}
static int fft_supported(int len) { return len == 480 || len == 240 || len == 60 || len == 10 || len == 20 || len == 30 || len == 40 || len == 80 || len == 120 || len == 160; }

/* R/util.h:109 cexpi with the reference's float argument conversion */
static void cexpi_f(float x, float* re, float* im) { *re = cosf(x); *im = sinf(x); }

static void build_plan(const geom_t* g, lc3d_plan* p)
{
    memset(p, 0, sizeof *p);
    p->fs = g->fs; p->fs_idx = g->fs_idx; p->dms = g->dms; p->hrmode = g->hrmode; p->N = g->N; p->ylen = g->ylen; p->la = g->la;
    p->nbands = g->nbands; p->bw_bits = g->bw_bits; p->fft_len = g->N / 2; p->channels = g->channels;
    p->rs_mem_in_len = 2 * 8 * g->fs / 12800;                              /* R/setup_enc_lc3.c:48 */
    p->rs_stride = lc3t_rs_upfac[g->fs_idx]; p->rs_scale = lc3t_rs_scale[g->fs_idx];
    p->len12 = g->dms == 25 ? 32 : g->dms == 50 ? 64 : 128; p->n12 = g->N * 12800 / g->fs;
    {   /* polyphase view of the resampler filter: output n uses phase start = (stride - 15n % stride) % stride, taps lp[239 - start - m*stride] */
        const int st = p->rs_stride, T = 240 / st;
        for (int ph = 0; ph < st; ph++) for (int m = 0; m < T; m++) p->rs_taps[ph * T + m] = lc3t_rs_lp[239 - ph - m * st];
    }
    p->ltpf_mem_len = g->dms == 25 ? 232 + 32 : 232;
    p->att_nblocks = g->att_nblocks; p->att_hang = g->att_hang; p->att_damping = g->att_damping; p->sns_damping = g->sns_damping;
    p->bw_cls = g->dms == 25 ? 0 : g->dms == 50 ? 1 : 2;
    p->win_off = g->tab->win_off; p->band_off = g->tab->band_off; p->tilt = g->tilt; p->frame_ms = g->frame_ms;
    const int len = g->N;
    for (int i = 0; i < len / 2; i++) {                                   /* R/dct4.c:58-61 */
        cexpi_f(-M_PI * (i + 0.25) / len, &p->tw1[2 * i], &p->tw1[2 * i + 1]);
        cexpi_f(-M_PI * i / len, &p->tw2[2 * i], &p->tw2[2 * i + 1]);
    }
    p->dct4_norm = 1.0 / sqrtf(len / 2);                                   /* R/dct4.c:82 */
    for (int i = 0; i < 16; i++) {                                         /* R/dct4.c:43-45 */
        float cr, ci, sr = 2 / sqrtf(2 * 16), si = 0;
        cexpi_f(-M_PI * i / (2 * 16), &cr, &ci);
        p->dct2_tw[2 * i] = cr * sr - ci * si;
        p->dct2_tw[2 * i + 1] = ci * sr + cr * si;
    }
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++)              /* R/sns_quantize_scf.c:30 */
        p->idct_cos[i * 16 + j] = cos(M_PI / (2.0 * (float)16) * (2.0 * ((float)i + 1.0) - 1.0) * ((float)j));
    p->c_idct_n1 = sqrtf(2.0 / (float)16); p->c_idct_n2 = 1.0 / (sqrtf(2.0));
    for (int i = 0; i < 64; i++)                                           /* R/sns_compute_scf.c:91 */
        p->sns_preemph[i] = powf(10.0, (float)i * (float)g->tilt / ((float)64 - 1.0) / 10.0);
    for (int k = -256; k < 256; k++) {
        float ind = (float)k;                                              /* R/estimate_global_gain.c:136: (ind + off) is a float holding k */
        p->gain_est[k + 256] = powf(10.0, (ind / 28.0));
        p->gain_adj[k + 256] = powf(10, (float)(k) / 28);                  /* R/adjust_global_gain.c:47 */
    }
    p->c_1em5_a = powf(10.0, -5.0); p->c_1em5_b = powf(10, -5); p->c_1em4 = powf(10.0, -40.0 / 10.0);
    p->c_2m32 = powf(2.0, -32.0); p->c_2m31 = powf(2, -31); p->c_2m24 = powf(2, -24); p->c_2p15 = powf(2, 15); p->c_2p100 = powf(2, 100);
    p->c_sqrt2 = sqrtf(2);
    {   /* float thresholds equivalent to the reference's double comparisons against float operands */
        const double t7 = (7.0) * (28.0 / 20.0), t50 = (50.0) * (28.0 / 20.0);
        float f = (float)t7; if ((double)f < t7) f = nextafterf(f, INFINITY); p->c_thr7_up = f;
        f = (float)t50; if ((double)f > t50) f = nextafterf(f, -INFINITY); p->c_thr50_dn = f;
    }
    for (int t = 0; t < 1024; t++) {
        unsigned w = 0, e[4];
        for (int j = 0; j < 4; j++) { const unsigned pk = lc3t_ac_ctx_lut[t + 1024 * j]; w |= pk << (8 * j); e[j] = lc3t_ac_bits[pk * 17 + 16]; }
        p->q_lut4[t] = w;
        p->q_esc[t][0] = (uint16_t)e[0]; p->q_esc[t][1] = (uint16_t)(e[0] + e[1]); p->q_esc[t][2] = (uint16_t)(e[0] + e[1] + e[2]); p->q_esc[t][3] = (uint16_t)e[3];
    }
    memcpy(p->q_bits, lc3t_ac_bits, sizeof p->q_bits);
    memset(p->band_of_bin, 255, sizeof p->band_of_bin);
    const uint16_t* be = &lc3t_band_pool[g->tab->band_off];
    for (int b = 0; b < g->nbands; b++) for (int j = be[b]; j < be[b + 1] && j < LC3D_MAX_N; j++) p->band_of_bin[j] = (uint8_t)b;
    if (g->N / 2 != 240 && g->N / 2 != 480) pfa_plan(p, g->N / 2);
}

static void init_state(float* st, int memcap)                             /* zeroed EncSetup + olpa_mem_pitch = 17 (R/setup_enc_lc3.c:178) */
{
    memset(st, 0, sizeof(float) * LC3D_STATE_WORDS(memcap));
    ((int*)st)[LC3D_S_OLPA_PITCH_WORD(memcap)] = 17;
}

/* ------------------------------------------------------------------------------------------------ */
/* batch object                                                                                      */
/* ------------------------------------------------------------------------------------------------ */
struct lc3plus_batch {
    geom_t g; int n_streams; int stride;
    lc3d_chan* chans;               /* [n_streams * channels] host mirror */
    int* bitrates;
    void* dev;
};

static LC3_Error batch_upload(lc3plus_batch* b, int first_stream, int count)
{
    const int C = b->g.channels;
    if (lc3hip_upload_chans(b->dev, b->chans + (size_t)first_stream * C, first_stream * C, count * C)) return LC3_ERROR;
    return LC3_OK;
}

static void batch_restride(lc3plus_batch* b)
{
    int s = 0;
    for (int i = 0; i < b->n_streams; i++) { int n = 0; for (int c = 0; c < b->g.channels; c++) n += b->chans[i * b->g.channels + c].nbytes; s = IMAX(s, n); }
    b->stride = s;
}

LC3_Error lc3plus_enc_batch_create(lc3plus_batch** out, int n_streams, int samplerate, int channels, float frame_ms, int hrmode,
                                   const int* bitrates, int device)
{
    if (!out || !bitrates) return LC3_NULL_ERROR;
    *out = NULL;
    if (n_streams <= 0) return LC3_ERROR;
    if (!samplerate_ok(samplerate)) return LC3_SAMPLERATE_ERROR;
    if (channels < 1 || channels > MAX_CH) return LC3_CHANNELS_ERROR;
    { int d = (int)ceil(frame_ms * 10); if (d != 25 && d != 50 && d != 100) return LC3_FRAMEMS_ERROR; }
    if (samplerate < 48000 && hrmode != 0) return LC3_SAMPLERATE_ERROR;
    lc3plus_batch* b = (lc3plus_batch*)calloc(1, sizeof *b);
    if (!b) return LC3_ERROR;
    geom_init(&b->g, samplerate, channels);
    b->g.dms = (int)(frame_ms * 10); b->g.frame_ms = frame_ms; b->g.hrmode = hrmode > 0;
    geom_update(&b->g);
    if (!geom_supported(&b->g)) {
        fprintf(stderr, "lc3plus_hip: %d Hz / %.1f ms%s is not built into the gfx950 kernels yet\n", samplerate, frame_ms, hrmode ? " hr" : "");
        free(b); return LC3_ERROR;
    }
    b->n_streams = n_streams;
    b->chans = (lc3d_chan*)calloc((size_t)n_streams * channels, sizeof(lc3d_chan));
    b->bitrates = (int*)calloc(n_streams, sizeof(int));
    if (!b->chans || !b->bitrates) { free(b->chans); free(b->bitrates); free(b); return LC3_ERROR; }
    for (int i = 0; i < n_streams; i++) {
        LC3_Error e = derive_bitrate(&b->g, bitrates[i], b->chans + (size_t)i * channels);
        if (e) { free(b->chans); free(b->bitrates); free(b); return e; }
        for (int c = 0; c < channels; c++) b->chans[i * channels + c].reset_attack = 0;
        b->bitrates[i] = bitrates[i];
    }
    batch_restride(b);
    lc3d_plan* plan = (lc3d_plan*)malloc(sizeof *plan);
    float st[LC3D_STATE_WORDS_MAX];
    build_plan(&b->g, plan);
    init_state(st, LC3D_LAYOUT_BIG(b->g.N, b->g.la) ? LC3D_MEMCAP_BIG : LC3D_MEMCAP_STD);
    int rc = lc3hip_create(&b->dev, plan, n_streams, device);
    free(plan);
    if (!rc) rc = lc3hip_reset_state(b->dev, st);
    if (!rc) rc = batch_upload(b, 0, n_streams) != LC3_OK;
    if (rc) { if (b->dev) lc3hip_destroy(b->dev); free(b->chans); free(b->bitrates); free(b); return LC3_ERROR; }
    *out = b;
    return LC3_OK;
}

LC3_Error lc3plus_enc_batch_destroy(lc3plus_batch* b)
{
    if (!b) return LC3_NULL_ERROR;
    lc3hip_destroy(b->dev);
    free(b->chans); free(b->bitrates); free(b);
    return LC3_OK;
}

int lc3plus_enc_batch_input_samples(const lc3plus_batch* b) { return b ? b->g.N : 0; }
int lc3plus_enc_batch_stride(const lc3plus_batch* b) { return b ? b->stride : 0; }
int lc3plus_enc_batch_num_bytes(const lc3plus_batch* b, int stream)
{
    if (!b || stream < 0 || stream >= b->n_streams) return 0;
    int n = 0;
    for (int c = 0; c < b->g.channels; c++) n += b->chans[stream * b->g.channels + c].nbytes;
    return n;
}

LC3_Error lc3plus_enc_batch_set_bitrate(lc3plus_batch* b, int stream, int bitrate)
{
    if (!b) return LC3_NULL_ERROR;
    if (stream < 0 || stream >= b->n_streams) return LC3_ERROR;
    if (bitrate <= 0) return LC3_BITRATE_ERROR;
    lc3d_chan tmp[MAX_CH];
    memcpy(tmp, b->chans + (size_t)stream * b->g.channels, sizeof(lc3d_chan) * b->g.channels);
    for (int c = 0; c < b->g.channels; c++) tmp[c].reset_attack = 0;
    LC3_Error e = derive_bitrate(&b->g, bitrate, tmp);
    if (e) return e;
    memcpy(b->chans + (size_t)stream * b->g.channels, tmp, sizeof(lc3d_chan) * b->g.channels);
    b->bitrates[stream] = bitrate;
    batch_restride(b);
    return batch_upload(b, stream, 1);
}

LC3_Error lc3plus_enc_batch_set_bandwidth(lc3plus_batch* b, int stream, int bandwidth)   /* R/lc3.c:187-208 */
{
    if (!b) return LC3_NULL_ERROR;
    if (stream < 0 || stream >= b->n_streams) return LC3_ERROR;
    if (b->g.hrmode == 1) return LC3_HRMODE_BW_ERROR;
    lc3d_chan* ch = b->chans + (size_t)stream * b->g.channels;
    int eff = b->g.fs_in;
    if (ch[0].bandwidth != bandwidth) {
        if (b->g.fs_in > 40000) eff = 40000;
        if (bandwidth * 2 > eff) return LC3_BW_WARNING;
        for (int c = 0; c < b->g.channels; c++) {
            ch[c].bandwidth = bandwidth;
            ch[c].bw_cut_bin = (bandwidth * b->g.dms) / 5000;
            ch[c].bw_index = IMAX(0, (bandwidth / 4000) - 1);
        }
        return batch_upload(b, stream, 1);
    }
    return LC3_OK;
}

static LC3_Error batch_encode(lc3plus_batch* b, const void* pcm, int pcm_on_device, int bitdepth, int n_frames, void* out, int out_stride,
                              int out_on_device, void* hip_stream, int sync, void* trace)
{
    if (!b || !pcm || !out) return LC3_NULL_ERROR;
    if (bitdepth != 16 && bitdepth != 24 && bitdepth != 32) return LC3_ERROR;
    if (n_frames <= 0 || out_stride < b->stride) return LC3_ERROR;
    if (lc3hip_encode(b->dev, pcm, pcm_on_device, bitdepth, n_frames, out, out_stride, out_on_device, hip_stream, sync, trace)) return LC3_ERROR;
    /* one-shot attack-state reset requests have been consumed by this launch */
    int dirty = 0;
    for (int i = 0; i < b->n_streams * b->g.channels; i++) if (b->chans[i].reset_attack) { b->chans[i].reset_attack = 0; dirty = 1; }
    if (dirty) return batch_upload(b, 0, b->n_streams);
    return LC3_OK;
}

LC3_Error lc3plus_enc_batch_encode(lc3plus_batch* b, const void* pcm, int pcm_on_device, int bitdepth, int n_frames, void* out,
                                   int out_stride, int out_on_device, void* hip_stream, int sync)
{
    return batch_encode(b, pcm, pcm_on_device, bitdepth, n_frames, out, out_stride, out_on_device, hip_stream, sync, NULL);
}

/* debug / stage-parity entry point used by tests: additionally returns one lc3d_trace per channel-frame */
LC3_Error lc3plus_enc_batch_encode_traced(lc3plus_batch* b, const void* pcm, int bitdepth, int n_frames, void* out, int out_stride, void* traces)
{
    return batch_encode(b, pcm, 0, bitdepth, n_frames, out, out_stride, 0, NULL, 1, traces);
}
int lc3plus_trace_sizeof(void) { return (int)sizeof(lc3d_trace); }

float lc3plus_enc_batch_last_kernel_ms(lc3plus_batch* b) { return b ? lc3hip_last_ms(b->dev) : 0.0f; }
int lc3plus_enc_batch_last_status(lc3plus_batch* b, uint8_t* status, int max_entries)
{
    if (!b || !status || max_entries < 0) return -1;
    return lc3hip_last_status(b->dev, status, max_entries);
}
int lc3plus_enc_batch_last_records(lc3plus_batch* b, float* records, int max_words)
{
    if (!b || !records || max_words < 0) return -1;
    return lc3hip_last_records(b->dev, records, max_words);
}
int lc3plus_enc_batch_record_words(void) { return FR_WORDS; }

size_t lc3plus_enc_batch_state_size(const lc3plus_batch* b) { return b ? lc3hip_state_bytes(b->dev) : 0; }
LC3_Error lc3plus_enc_batch_get_state(lc3plus_batch* b, void* state, size_t size)
{
    if (!b || !state) return LC3_NULL_ERROR;
    return lc3hip_get_state(b->dev, state, size) ? LC3_ERROR : LC3_OK;
}
LC3_Error lc3plus_enc_batch_set_state(lc3plus_batch* b, const void* state, size_t size)
{
    if (!b || !state) return LC3_NULL_ERROR;
    return lc3hip_set_state(b->dev, state, size) ? LC3_ERROR : LC3_OK;
}

LC3_Error lc3plus_enc_batch_set_input_ready(lc3plus_batch* b, int ready)
{
    if (!b) return LC3_NULL_ERROR;
    return lc3hip_set_input_ready(b->dev, ready) ? LC3_ERROR : LC3_OK;
}

/* ------------------------------------------------------------------------------------------------ */
/* single-stream drop-in API (R/lc3.h:163-295)                                                       */
/* ------------------------------------------------------------------------------------------------ */
struct LC3_Enc {
    /* R/codec_exe.c is not a pure client of the opaque API: compiled against the reference's own headers it reads encoder->bitrate
     * (:298-299), ->epmode (:310-312) and ->bandwidth (:320) directly, i.e. the words at byte offsets 48, 64 and 144 of the
     * reference's struct (R/setup_enc_lc3.h:65-104 on LP64: five pointers, then ints).  The three fields sit at those offsets here
     * so that the unmodified CLI relinks against this library (`make relink`); the words between them are unused. */
    void* ref_ptr[5];               /*   0 ..  39 */
    int ref_w40[2];                 /*  40 ..  47 */
    int bitrate;                    /*  48 */
    int ref_w52[3];                 /*  52 ..  63 */
    int epmode;                     /*  64 */
    int ref_w68[19];                /*  68 .. 143 */
    int bandwidth;                  /* 144 */
    int ref_w148[7];                /* 148 .. 175: the rest of the reference struct's 176 bytes */
    int lc3_br_set, channels, samplerate, hrmode; float frame_ms;
    geom_t g;
    lc3d_chan ch[MAX_CH];
    lc3plus_batch* batch;           /* batch of one stream, created lazily at the first encode */
    int16_t* stage16; int32_t* stage32; uint8_t* stage_out;
    unsigned magic;
};
_Static_assert(offsetof(struct LC3_Enc, bitrate) == 48 && offsetof(struct LC3_Enc, epmode) == 64 && offsetof(struct LC3_Enc, bandwidth) == 144,
               "fields R/codec_exe.c reads must sit at the reference's offsets");
#define ENC_MAGIC 0x4C433350u

int lc3_version(void) { return LC3_VERSION; }
int lc3_channels_supported(int channels) { return channels >= 1 && channels <= MAX_CH; }
int lc3_samplerate_supported(int samplerate) { return samplerate_ok(samplerate); }

int lc3_enc_get_size(int samplerate, int channels)
{
    if (!lc3_samplerate_supported(samplerate) || !lc3_channels_supported(channels)) return 0;
    return (int)sizeof(struct LC3_Enc);
}

static void enc_drop_device(LC3_Enc* e)
{
    if (e->batch) { lc3plus_enc_batch_destroy(e->batch); e->batch = NULL; }
    free(e->stage16); free(e->stage32); free(e->stage_out); e->stage16 = NULL; e->stage32 = NULL; e->stage_out = NULL;
}

LC3_Error lc3_enc_init(LC3_Enc* e, int samplerate, int channels)
{
    if (e == NULL) return LC3_NULL_ERROR;
    if ((uintptr_t)e % 4 != 0) return LC3_ALIGN_ERROR;
    if (!lc3_samplerate_supported(samplerate)) return LC3_SAMPLERATE_ERROR;
    if (!lc3_channels_supported(channels)) return LC3_CHANNELS_ERROR;
    memset(e, 0, sizeof *e);
    e->magic = ENC_MAGIC; e->channels = channels; e->samplerate = samplerate; e->frame_ms = 10;
    geom_init(&e->g, samplerate, channels);
    geom_update(&e->g);
    return LC3_OK;
}

LC3_Error lc3_enc_set_frame_ms(LC3_Enc* e, float frame_ms)
{
    if (e == NULL) return LC3_NULL_ERROR;
    { int d = (int)ceil(frame_ms * 10); if (d != 25 && d != 50 && d != 100) return LC3_FRAMEMS_ERROR; }
    if (e->lc3_br_set) return LC3_BITRATE_SET_ERROR;
    e->g.dms = (int)(frame_ms * 10); e->g.frame_ms = frame_ms; e->frame_ms = frame_ms;
    geom_update(&e->g);
    enc_drop_device(e);
    return LC3_OK;
}

LC3_Error lc3_enc_set_hrmode(LC3_Enc* e, int hrmode)
{
    if (e == NULL) return LC3_NULL_ERROR;
    if (e->g.fs_in < 48000 && hrmode != 0) return LC3_SAMPLERATE_ERROR;
    e->g.hrmode = hrmode > 0; e->hrmode = e->g.hrmode;
    geom_update(&e->g);
    enc_drop_device(e);
    return LC3_OK;
}

LC3_Error lc3_enc_set_bitrate(LC3_Enc* e, int bitrate)
{
    if (e == NULL) return LC3_NULL_ERROR;
    if (bitrate <= 0) return LC3_BITRATE_ERROR;
    if (e->g.fs_idx == 5 && e->g.hrmode == 0) return LC3_HRMODE_ERROR;
    lc3d_chan tmp[MAX_CH];
    memcpy(tmp, e->ch, sizeof tmp);
    LC3_Error err = derive_bitrate(&e->g, bitrate, tmp);
    if (err) return err;
    memcpy(e->ch, tmp, sizeof tmp);
    e->lc3_br_set = 1; e->bitrate = bitrate;
    if (e->batch) return lc3plus_enc_batch_set_bitrate(e->batch, 0, bitrate);
    return LC3_OK;
}

LC3_Error lc3_enc_set_bandwidth(LC3_Enc* e, int bandwidth)
{
    if (e == NULL) return LC3_NULL_ERROR;
    if (e->g.hrmode == 1) return LC3_HRMODE_BW_ERROR;
    int eff = e->g.fs_in;
    if (e->bandwidth != bandwidth) {
        if (e->g.fs_in > 40000) eff = 40000;
        if (bandwidth * 2 > eff) return LC3_BW_WARNING;
        e->bandwidth = bandwidth;
        for (int c = 0; c < e->channels; c++) {
            e->ch[c].bandwidth = bandwidth; e->ch[c].bw_cut_bin = (bandwidth * e->g.dms) / 5000; e->ch[c].bw_index = IMAX(0, (bandwidth / 4000) - 1);
        }
        if (e->batch) return lc3plus_enc_batch_set_bandwidth(e->batch, 0, bandwidth);
    }
    return LC3_OK;
}

int lc3_enc_get_input_samples(const LC3_Enc* e) { return e ? e->g.N : 0; }
int lc3_enc_get_num_bytes(const LC3_Enc* e) { return e ? e->ch[0].nbytes * e->channels : 0; }   /* R/lc3.c:124-129 (sic) */
int lc3_enc_get_delay(const LC3_Enc* e) { return e ? e->g.N - 2 * e->g.la : 0; }
int lc3_enc_get_real_bitrate(const LC3_Enc* e)
{
    if (e == NULL) return 0;
    if (!e->lc3_br_set) return LC3_BITRATE_UNSET_ERROR;
    int tot = 0;
    for (int c = 0; c < e->channels; c++) tot += e->ch[c].nbytes;
    int br = (tot * 80000) / e->g.dms;
    if (e->g.fs_in == 44100) { int rem = br % 480; br = ((br - rem) / 480) * 441 + (rem * 441) / 480; }
    return br;
}

LC3_Error lc3_enc_fl(LC3_Enc* e, void** input_samples, int bitdepth, void* output_bytes, int* num_bytes)
{
    if (!e || !input_samples || !output_bytes || !num_bytes) return LC3_NULL_ERROR;
    for (int c = 0; c < e->channels; c++) if (input_samples[c] == NULL) return LC3_NULL_ERROR;
    if (bitdepth != 16 && bitdepth != 24 && bitdepth != 32) return LC3_ERROR;
    if (!e->lc3_br_set) return LC3_BITRATE_UNSET_ERROR;
    const int N = e->g.N, C = e->channels;
    if (!e->batch) {
        LC3_Error err = lc3plus_enc_batch_create(&e->batch, 1, e->g.fs_in, C, e->g.frame_ms, e->g.hrmode, &e->bitrate, -1);
        if (err) return err == LC3_BITRATE_ERROR ? err : LC3_ERROR;
        if (e->bandwidth) lc3plus_enc_batch_set_bandwidth(e->batch, 0, e->bandwidth);
        e->stage16 = (int16_t*)malloc(sizeof(int16_t) * C * LC3D_MAX_N);
        e->stage32 = (int32_t*)malloc(sizeof(int32_t) * C * LC3D_MAX_N);
        e->stage_out = (uint8_t*)malloc(LC3_MAX_BYTES);
        if (!e->stage16 || !e->stage32 || !e->stage_out) { enc_drop_device(e); return LC3_ERROR; }
    }
    const void* pcm;
    if (bitdepth == 16) { for (int c = 0; c < C; c++) memcpy(e->stage16 + c * N, input_samples[c], sizeof(int16_t) * N); pcm = e->stage16; }
    else { for (int c = 0; c < C; c++) memcpy(e->stage32 + c * N, input_samples[c], sizeof(int32_t) * N); pcm = e->stage32; }
    const int nb = lc3plus_enc_batch_num_bytes(e->batch, 0);
    LC3_Error err = lc3plus_enc_batch_encode(e->batch, pcm, 0, bitdepth, 1, output_bytes, nb, 0, NULL, 1);
    if (err) return err;
    *num_bytes = nb;
    return LC3_OK;
}
LC3_Error lc3_enc16(LC3_Enc* e, int16_t** in, void* out, int* nb) { return lc3_enc_fl(e, (void**)in, 16, out, nb); }
LC3_Error lc3_enc24(LC3_Enc* e, int32_t** in, void* out, int* nb) { return lc3_enc_fl(e, (void**)in, 24, out, nb); }
LC3_Error lc3_enc32(LC3_Enc* e, int32_t** in, void* out, int* nb) { return lc3_enc_fl(e, (void**)in, 32, out, nb); }

LC3_Error lc3_free_encoder_structs(LC3_Enc* e)
{
    if (!e) return LC3_NULL_ERROR;
    if (e->magic == ENC_MAGIC) enc_drop_device(e);
    return LC3_OK;
}
LC3_Error lc3_enc_free_memory(LC3_Enc* e)
{
    if (!e) return LC3_NULL_ERROR;
    lc3_free_encoder_structs(e);
    free(e);
    return LC3_OK;
}

/* lc3plus_enc_* aliases (north-star wording) */
LC3_Error lc3plus_enc_init(LC3_Enc* e, int sr, int ch) { return lc3_enc_init(e, sr, ch); }
LC3_Error lc3plus_enc_set_frame_ms(LC3_Enc* e, float ms) { return lc3_enc_set_frame_ms(e, ms); }
LC3_Error lc3plus_enc_set_hrmode(LC3_Enc* e, int hr) { return lc3_enc_set_hrmode(e, hr); }
LC3_Error lc3plus_enc_set_bitrate(LC3_Enc* e, int br) { return lc3_enc_set_bitrate(e, br); }
LC3_Error lc3plus_enc16(LC3_Enc* e, int16_t** in, void* out, int* nb) { return lc3_enc16(e, in, out, nb); }
int lc3plus_enc_get_size(int sr, int ch) { return lc3_enc_get_size(sr, ch); }


/* ================================================================================================ */
/* decoder: lc3_dec_* drop-in API and the batched form                          */
/* ================================================================================================ */
/* same operating points as the encoder: standard and large kernel layout */
static int dec_geom_supported(const geom_t* g) { return geom_supported(g); }

/* R/setup_dec_lc3.c:203-299 (update_dec_bitrate) for one channel */
static LC3_Error derive_dchan(const geom_t* g, int nbytes, lc3d_dchan* d)
{
    int min_b = 20, max_b = 400;                                          /* R/defines.h MIN_NBYTES / MAX_NBYTES */
    if (g->hrmode) {
        switch (g->dms) {
        case 25:  max_b = 210; if (g->fs == 48000) min_b = 54;  else if (g->fs == 96000) min_b = 62;  else return LC3_HRMODE_ERROR; break;
        case 50:  max_b = 375; if (g->fs == 48000) min_b = 93;  else if (g->fs == 96000) min_b = 109; else return LC3_HRMODE_ERROR; break;
        case 100: max_b = 625; if (g->fs == 48000) min_b = 156; else if (g->fs == 96000) min_b = 187; else return LC3_HRMODE_ERROR; break;
        default: return LC3_HRMODE_ERROR;
        }
    }
    if (nbytes < min_b || nbytes > max_b) return LC3_NUMBYTES_ERROR;
    const int total_bits = nbytes << 3;
    d->nbytes = nbytes;
    d->lpc_weighting = total_bits < 480;
    d->gg_off = -(IMIN(115, total_bits / (10 * (g->fs_idx + 1))) + 105 + 5 * (g->fs_idx + 1));
    int tb = total_bits;
    if (g->dms == 25) { d->lpc_weighting = total_bits < 120; tb = (int)(total_bits * 4.0 * (1.0 - 0.4)); }
    if (g->dms == 50) { d->lpc_weighting = total_bits < 240; tb = total_bits * 2 - 160; }
    if (g->N > 40 * ((float)g->dms / 10.0)) { d->N_red_tns = (int)(40 * ((float)g->dms / 10.0)); d->fs_red_tns = 40000; }
    else { d->N_red_tns = g->N; d->fs_red_tns = g->fs; }
    const int k = (g->fs_idx - 1) * 80;
    if (tb < 400 + k)      { d->ltpf_beta = 0.4f;  d->ltpf_beta_idx = 0; }
    else if (tb < 480 + k) { d->ltpf_beta = 0.35f; d->ltpf_beta_idx = 1; }
    else if (tb < 560 + k) { d->ltpf_beta = 0.3f;  d->ltpf_beta_idx = 2; }
    else if (tb < 640 + k) { d->ltpf_beta = 0.25f; d->ltpf_beta_idx = 3; }
    else                   { d->ltpf_beta = 0;     d->ltpf_beta_idx = -1; }
    if (g->hrmode == 1) { d->ltpf_beta = 0; d->ltpf_beta_idx = -1; }
    return LC3_OK;
}

/* split of a stream-frame of num_bytes over the channels (R/dec_lc3_fl.c:148) and the payload offsets */
static LC3_Error derive_dstream(const geom_t* g, int num_bytes, lc3d_dchan* d /* [channels] */)
{
    int off = 0;
    for (int c = 0; c < g->channels; c++) {
        LC3_Error e = derive_dchan(g, num_bytes / g->channels + (c < (num_bytes % g->channels)), d + c);
        if (e) return e;
        d[c].in_off = off; off += d[c].nbytes;
    }
    return LC3_OK;
}

struct lc3plus_dec_batch {
    geom_t g; int n_streams;
    lc3d_dchan* chans;               /* [n_streams * channels] host mirror */
    void* dev;
};

LC3_Error lc3plus_dec_batch_create(lc3plus_dec_batch** out, int n_streams, int samplerate, int channels, float frame_ms, int hrmode,
                                   const int* num_bytes, int device)
{
    if (!out) return LC3_NULL_ERROR;
    *out = NULL;
    if (n_streams <= 0) return LC3_ERROR;
    if (!samplerate_ok(samplerate)) return LC3_SAMPLERATE_ERROR;
    if (channels < 1 || channels > MAX_CH) return LC3_CHANNELS_ERROR;
    { int d = (int)ceil(frame_ms * 10); if (d != 25 && d != 50 && d != 100) return LC3_FRAMEMS_ERROR; }
    lc3plus_dec_batch* b = (lc3plus_dec_batch*)calloc(1, sizeof *b);
    if (!b) return LC3_ERROR;
    geom_init(&b->g, samplerate, channels);
    if (b->g.fs_idx < 4 && hrmode != 0) { free(b); return LC3_SAMPLERATE_ERROR; }          /* R/lc3.c:350 */
    if (b->g.fs_idx == 5 && hrmode == 0) { free(b); return LC3_HRMODE_ERROR; }             /* R/lc3.c:351 */
    b->g.dms = (int)(frame_ms * 10); b->g.frame_ms = frame_ms; b->g.hrmode = hrmode > 0;
    geom_update_ex(&b->g, 1);
    if (!dec_geom_supported(&b->g)) {
        fprintf(stderr, "lc3plus_hip: decoding %d Hz / %.1f ms%s is not built into the gfx950 kernels yet\n", samplerate, frame_ms, hrmode ? " hr" : "");
        free(b); return LC3_ERROR;
    }
    b->n_streams = n_streams;
    b->chans = (lc3d_dchan*)calloc((size_t)n_streams * channels, sizeof(lc3d_dchan));
    if (!b->chans) { free(b); return LC3_ERROR; }
    if (num_bytes)
        for (int i = 0; i < n_streams; i++) {
            LC3_Error e = derive_dstream(&b->g, num_bytes[i], b->chans + (size_t)i * channels);
            if (e) { free(b->chans); free(b); return e; }
        }
    lc3d_plan* plan = (lc3d_plan*)malloc(sizeof *plan);
    if (!plan) { free(b->chans); free(b); return LC3_ERROR; }
    build_plan(&b->g, plan);
    int rc = lc3hip_dec_create(&b->dev, plan, n_streams, device);
    free(plan);
    if (!rc) rc = lc3hip_dec_upload_chans(b->dev, b->chans, 0, n_streams * channels);
    if (rc) { if (b->dev) lc3hip_dec_destroy(b->dev); free(b->chans); free(b); return LC3_ERROR; }
    *out = b;
    return LC3_OK;
}

size_t lc3plus_dec_batch_state_size(const lc3plus_dec_batch* b) { return b ? lc3hip_dec_state_bytes(b->dev) : 0; }
LC3_Error lc3plus_dec_batch_get_state(lc3plus_dec_batch* b, void* state, size_t size)
{
    if (!b || !state) return LC3_NULL_ERROR;
    return lc3hip_dec_get_state(b->dev, state, size) ? LC3_ERROR : LC3_OK;
}
LC3_Error lc3plus_dec_batch_set_state(lc3plus_dec_batch* b, const void* state, size_t size)
{
    if (!b || !state) return LC3_NULL_ERROR;
    return lc3hip_dec_set_state(b->dev, state, size) ? LC3_ERROR : LC3_OK;
}

LC3_Error lc3plus_dec_batch_destroy(lc3plus_dec_batch* b)
{
    if (!b) return LC3_NULL_ERROR;
    lc3hip_dec_destroy(b->dev);
    free(b->chans); free(b);
    return LC3_OK;
}

int lc3plus_dec_batch_output_samples(const lc3plus_dec_batch* b) { return b ? b->g.N : 0; }
int lc3plus_dec_batch_delay(const lc3plus_dec_batch* b) { return b ? b->g.N - 2 * b->g.la : 0; }
int lc3plus_dec_batch_num_bytes(const lc3plus_dec_batch* b, int stream)
{
    if (!b || stream < 0 || stream >= b->n_streams) return 0;
    int n = 0;
    for (int c = 0; c < b->g.channels; c++) n += b->chans[stream * b->g.channels + c].nbytes;
    return n;
}

/* frame size change of one stream between decode() calls: what R/dec_lc3_fl.c:149-155 does when num_bytes changes */
LC3_Error lc3plus_dec_batch_set_num_bytes(lc3plus_dec_batch* b, int stream, int num_bytes)
{
    if (!b) return LC3_NULL_ERROR;
    if (stream < 0 || stream >= b->n_streams) return LC3_ERROR;
    lc3d_dchan tmp[MAX_CH];
    memset(tmp, 0, sizeof tmp);
    LC3_Error e = derive_dstream(&b->g, num_bytes, tmp);
    if (e) return e;
    memcpy(b->chans + (size_t)stream * b->g.channels, tmp, sizeof(lc3d_dchan) * b->g.channels);
    return lc3hip_dec_upload_chans(b->dev, tmp, stream * b->g.channels, b->g.channels) ? LC3_ERROR : LC3_OK;
}

static LC3_Error dec_batch_decode(lc3plus_dec_batch* b, const void* frames, int frames_on_device, int in_stride, const uint8_t* bfi, int n_frames,
                                  void* pcm, int pcm_on_device, int bps, uint8_t* status, void* hip_stream, int sync, void* traces)
{
    if (!b || !frames || !pcm) return LC3_NULL_ERROR;
    if (bps != 16 && bps != 24 && bps != 32) return LC3_ERROR;
    if (n_frames <= 0) return LC3_ERROR;
    for (int i = 0; i < b->n_streams; i++) if (lc3plus_dec_batch_num_bytes(b, i) > in_stride) return LC3_NUMBYTES_ERROR;
    return lc3hip_dec_decode(b->dev, frames, frames_on_device, in_stride, bfi, n_frames, pcm, pcm_on_device, bps, status, hip_stream, sync, traces)
               ? LC3_ERROR : LC3_OK;
}
LC3_Error lc3plus_dec_batch_decode(lc3plus_dec_batch* b, const void* frames, int frames_on_device, int in_stride, const uint8_t* bfi, int n_frames,
                                   void* pcm, int pcm_on_device, int bps, uint8_t* status, void* hip_stream, int sync)
{
    return dec_batch_decode(b, frames, frames_on_device, in_stride, bfi, n_frames, pcm, pcm_on_device, bps, status, hip_stream, sync, NULL);
}
/* test hook: per channel-stream per frame stage traces (lc3d_dec_trace), host pointers only */
LC3_Error lc3plus_dec_batch_decode_traced(lc3plus_dec_batch* b, const void* frames, int in_stride, const uint8_t* bfi, int n_frames, void* pcm, int bps,
                                          uint8_t* status, void* traces)
{
    return dec_batch_decode(b, frames, 0, in_stride, bfi, n_frames, pcm, 0, bps, status, NULL, 1, traces);
}
int lc3plus_dec_trace_sizeof(void) { return (int)sizeof(lc3d_dec_trace); }
float lc3plus_dec_batch_last_kernel_ms(lc3plus_dec_batch* b) { return b ? lc3hip_dec_last_ms(b->dev) : 0.0f; }
LC3_Error lc3plus_dec_batch_set_input_ready(lc3plus_dec_batch* b, int ready)
{
    if (!b) return LC3_NULL_ERROR;
    return lc3hip_dec_set_input_ready(b->dev, ready) ? LC3_ERROR : LC3_OK;
}

/* ---- single-stream drop-in API (R/lc3.h:318-406) ---- */
struct LC3_Dec {
    int channels, samplerate, plc_mode; float frame_ms;
    geom_t g;
    int last_size[MAX_CH];           /* R/setup_dec_lc3.h last_size: bytes of the channel's last good frame */
    lc3d_dchan ch[MAX_CH];
    lc3plus_dec_batch* batch;        /* batch of one stream, created lazily at the first decode */
    uint8_t* stage_in; void* stage_pcm;
    unsigned magic;
};
#define DEC_MAGIC 0x4C433344u

int lc3_dec_get_size(int samplerate, int channels)
{
    if (!lc3_samplerate_supported(samplerate) || !lc3_channels_supported(channels)) return 0;
    return (int)sizeof(struct LC3_Dec);
}

static void dec_drop_device(LC3_Dec* d)
{
    if (d->batch) { lc3plus_dec_batch_destroy(d->batch); d->batch = NULL; }
    free(d->stage_in); free(d->stage_pcm); d->stage_in = NULL; d->stage_pcm = NULL;
}

LC3_Error lc3_dec_init(LC3_Dec* d, int samplerate, int channels, LC3_PlcMode plc_mode)
{
    if (d == NULL) return LC3_NULL_ERROR;
    if (!lc3_samplerate_supported(samplerate)) return LC3_SAMPLERATE_ERROR;
    if (!lc3_channels_supported(channels)) return LC3_CHANNELS_ERROR;
    if ((int)plc_mode != LC3_PLC_STANDARD) return LC3_PLCMODE_ERROR;      /* R/lc3.c:64-72 */
    memset(d, 0, sizeof *d);
    d->magic = DEC_MAGIC; d->channels = channels; d->samplerate = samplerate; d->frame_ms = 10; d->plc_mode = (int)plc_mode;
    geom_init(&d->g, samplerate, channels);
    geom_update_ex(&d->g, 1);
    return LC3_OK;
}

/* Changing the frame size or hrmode restarts the stream with fresh memories (the reference keeps the old buffers,
 * R/setup_dec_lc3.c:73-199, whose contents belong to the other frame size; no caller in the reference does this mid-stream) */
LC3_Error lc3_dec_set_frame_ms(LC3_Dec* d, float frame_ms)
{
    if (d == NULL) return LC3_NULL_ERROR;
    { int k = (int)ceil(frame_ms * 10); if (k != 25 && k != 50 && k != 100) return LC3_FRAMEMS_ERROR; }
    d->g.dms = (int)(frame_ms * 10); d->g.frame_ms = frame_ms; d->frame_ms = frame_ms;
    geom_update_ex(&d->g, 1);
    dec_drop_device(d);
    memset(d->last_size, 0, sizeof d->last_size); memset(d->ch, 0, sizeof d->ch);
    return LC3_OK;
}

LC3_Error lc3_dec_set_hrmode(LC3_Dec* d, int hrmode)
{
    if (d == NULL) return LC3_NULL_ERROR;
    if (d->g.fs_idx < 4 && hrmode != 0) return LC3_SAMPLERATE_ERROR;
    if (d->g.fs_idx == 5 && hrmode == 0) return LC3_HRMODE_ERROR;
    d->g.hrmode = hrmode > 0;
    geom_update_ex(&d->g, 1);
    dec_drop_device(d);
    memset(d->last_size, 0, sizeof d->last_size); memset(d->ch, 0, sizeof d->ch);
    return LC3_OK;
}

int lc3_dec_get_output_samples(const LC3_Dec* d) { return d ? d->g.N : 0; }
int lc3_dec_get_delay(const LC3_Dec* d) { return d ? d->g.N - 2 * d->g.la : 0; }

LC3_Error lc3_dec_fl(LC3_Dec* d, void* input_bytes, int num_bytes, void** output_samples, int bps, int bfi_ext)
{
    if (!d || !input_bytes || !output_samples) return LC3_NULL_ERROR;
    for (int c = 0; c < d->channels; c++) if (output_samples[c] == NULL) return LC3_NULL_ERROR;
    if (bps != 16 && bps != 24 && bps != 32) return LC3_ERROR;
    if (num_bytes < 0 || num_bytes > LC3_MAX_BYTES) return LC3_NUMBYTES_ERROR;
    const int N = d->g.N, C = d->channels;
    if (!d->batch) {
        LC3_Error err = lc3plus_dec_batch_create(&d->batch, 1, d->g.fs_in, C, d->g.frame_ms, d->g.hrmode, NULL, -1);
        if (err) return err;
        d->stage_in = (uint8_t*)malloc(LC3_MAX_BYTES);
        d->stage_pcm = malloc(sizeof(int32_t) * C * LC3D_MAX_N);
        if (!d->stage_in || !d->stage_pcm) { dec_drop_device(d); return LC3_ERROR; }
    }
    int bfi = bfi_ext;
    if (bfi == 0) bfi = !num_bytes;                                        /* R/dec_lc3_fl.c:140-143 */
    if (bfi != 1) {
        /* R/dec_lc3_fl.c:146-155.  (The reference skips the update of a later channel when an earlier channel of the
         * SAME frame turns out corrupt; that needs the decode result and is not reproduced: INTEGRATION.md.) */
        int changed = 0, off = 0;
        lc3d_dchan tmp[MAX_CH];
        memcpy(tmp, d->ch, sizeof tmp);
        for (int c = 0; c < C; c++) {
            const int nb2 = num_bytes / C + (c < (num_bytes % C));
            if (nb2 != d->last_size[c]) {
                LC3_Error e = derive_dchan(&d->g, nb2, &tmp[c]);
                if (e) return e;
                changed = 1;
            }
            tmp[c].in_off = off; off += tmp[c].nbytes;
        }
        if (changed || memcmp(tmp, d->ch, sizeof tmp)) {
            memcpy(d->ch, tmp, sizeof tmp);
            for (int c = 0; c < C; c++) d->last_size[c] = d->ch[c].nbytes;
            memcpy(d->batch->chans, d->ch, sizeof(lc3d_dchan) * C);
            if (lc3hip_dec_upload_chans(d->batch->dev, d->ch, 0, C)) return LC3_ERROR;
        }
    }
    const int stride = IMAX(num_bytes, lc3plus_dec_batch_num_bytes(d->batch, 0));
    memset(d->stage_in, 0, LC3_MAX_BYTES);
    memcpy(d->stage_in, input_bytes, num_bytes);
    uint8_t flag = (uint8_t)(bfi == 1), status = 0;
    LC3_Error err = lc3plus_dec_batch_decode(d->batch, d->stage_in, 0, IMAX(stride, 1), &flag, 1, d->stage_pcm, 0, bps, &status, NULL, 1);
    if (err) return err;
    const size_t ss = bps == 16 ? 2 : 4;
    for (int c = 0; c < C; c++) memcpy(output_samples[c], (char*)d->stage_pcm + ss * c * N, ss * N);
    return status ? LC3_DECODE_ERROR : LC3_OK;
}
LC3_Error lc3_dec16(LC3_Dec* d, void* in, int nb, int16_t** out, int bfi_ext) { return lc3_dec_fl(d, in, nb, (void**)out, 16, bfi_ext); }
LC3_Error lc3_dec24(LC3_Dec* d, void* in, int nb, int32_t** out, int bfi_ext) { return lc3_dec_fl(d, in, nb, (void**)out, 24, bfi_ext); }
LC3_Error lc3_dec32(LC3_Dec* d, void* in, int nb, int32_t** out, int bfi_ext) { return lc3_dec_fl(d, in, nb, (void**)out, 32, bfi_ext); }

LC3_Error lc3_free_decoder_structs(LC3_Dec* d)
{
    if (!d) return LC3_NULL_ERROR;
    if (d->magic == DEC_MAGIC) dec_drop_device(d);
    return LC3_OK;
}
LC3_Error lc3_dec_free_memory(LC3_Dec* d)
{
    if (!d) return LC3_NULL_ERROR;
    lc3_free_decoder_structs(d);
    free(d);
    return LC3_OK;
}
