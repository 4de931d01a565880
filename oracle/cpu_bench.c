/* oracle/cpu_bench.c -- TEST / MEASUREMENT INFRASTRUCTURE ONLY (never linked into the product library).
 *
 * The CPU baseline of bench.py (SURVEY 8(d) "CPU baseline"): the ETSI floating-point reference encoder / decoder
 * (oracle/_ref/liblc3_etsi_fl.so, compiled from the sources where they lie; kind "reference": oracle/_ref/cpu_bench_ref) or,
 * where that did not travel, the C restatement (oracle/liblc3_oracle.so; kind "port": oracle/cpu_bench_port, -DCPU_BENCH_PORT),
 * driven from C over the SAME PCM (or bitstreams) the GPU ran on: bench.py writes a bounded sample of its device buffers to a
 * file and this driver reads it.  Streams are independent (R/enc_lc3_fl.c:167-171), so P threads take contiguous blocks of
 * streams; the figure is frames / wall time of the threaded region (threads created -> last one joined), with no per-frame
 * Python or ctypes in the way.  The calls are the reference's own API in the order R/codec_exe.c:171-199,369-381 makes them.
 *
 * usage: cpu_bench enc|dec FS FRAME_MS HRMODE CHANNELS BITRATE_OR_NBYTES N_STREAMS N_FRAMES THREADS IN_FILE [OUT_FILE]
 *   enc: IN_FILE = int16 [stream][frame][channel][N];  BITRATE = per stream, all channels together
 *   dec: IN_FILE = uint8 [stream][frame][NBYTES]
 *   OUT_FILE (optional): the encoded frames [stream][frame][nbytes] / decoded PCM, for a cross-check against the GPU output
 * environment (enc only, tools/ref_soak.py): LC3_BENCH_BW_PLAN = file of int32 [stream][frame]; before frame t of stream s the bandwidth is set to
 *   plan[s][t] when that is not 0 - the switching file of R/codec_exe.c:338-352 as an array (a refused value is ignored, as there)
 * prints one line: "frames seconds threads"
 */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#ifdef CPU_BENCH_PORT
#include "lc3_oracle.h"
#define ENC_SIZE(fs, ch) lc3o_enc_sizeof()
#define DEC_SIZE(fs, ch) lc3o_dec_sizeof()
#define ENC_INIT(h, fs, ch) lc3o_enc_init((lc3o_enc*)(h), fs, ch)
#define DEC_INIT(h, fs, ch) lc3o_dec_init((lc3o_dec*)(h), fs, ch)
#define ENC_SET_MS(h, ms) lc3o_enc_set_frame_ms((lc3o_enc*)(h), ms)
#define DEC_SET_MS(h, ms) lc3o_dec_set_frame_ms((lc3o_dec*)(h), ms)
#define ENC_SET_HR(h, v) lc3o_enc_set_hrmode((lc3o_enc*)(h), v)
#define DEC_SET_HR(h, v) lc3o_dec_set_hrmode((lc3o_dec*)(h), v)
#define ENC_SET_BR(h, v) lc3o_enc_set_bitrate((lc3o_enc*)(h), v)
#define ENC_SET_BW(h, v) lc3o_enc_set_bandwidth((lc3o_enc*)(h), v)
#define ENC_NB(h) lc3o_enc_get_num_bytes((lc3o_enc*)(h))
#define ENC_N(h) lc3o_enc_get_input_samples((lc3o_enc*)(h))
#define DEC_N(h) lc3o_dec_get_output_samples((lc3o_dec*)(h))
#define ENC_FRAME(h, in, out, nb) lc3o_enc_frame((lc3o_enc*)(h), in, 16, out, nb)
#define DEC_FRAME(h, in, nb, out) lc3o_dec_frame((lc3o_dec*)(h), in, nb, out, 16, 0)
#define ENC_FREE(h) lc3o_enc_free((lc3o_enc*)(h))
#define DEC_FREE(h) ((void)0)
#else
/* the prototypes a C client of R/lc3.h:120-406 sees (the header itself stays where it lies) */
int lc3_enc_get_size(int samplerate, int channels);
int lc3_enc_init(void* e, int samplerate, int channels);
int lc3_enc_set_frame_ms(void* e, float ms);
int lc3_enc_set_hrmode(void* e, int hr);
int lc3_enc_set_bitrate(void* e, int br);
int lc3_enc_set_bandwidth(void* e, int bw);
int lc3_enc_get_num_bytes(const void* e);
int lc3_enc_get_input_samples(const void* e);
int lc3_enc_fl(void* e, void** in, int bitdepth, void* out, int* nb);
int lc3_free_encoder_structs(void* e);
int lc3_dec_get_size(int samplerate, int channels, int plc);
int lc3_dec_init(void* d, int samplerate, int channels, int plc);
int lc3_dec_set_frame_ms(void* d, float ms);
int lc3_dec_set_hrmode(void* d, int hr);
int lc3_dec_get_output_samples(const void* d);
int lc3_dec_fl(void* d, void* in, int nb, void** out, int bps, int bfi);
int lc3_free_decoder_structs(void* d);
#define ENC_SIZE(fs, ch) lc3_enc_get_size(fs, ch)
#define DEC_SIZE(fs, ch) lc3_dec_get_size(fs, ch, 0)
#define ENC_INIT(h, fs, ch) lc3_enc_init(h, fs, ch)
#define DEC_INIT(h, fs, ch) lc3_dec_init(h, fs, ch, 0)
#define ENC_SET_MS(h, ms) lc3_enc_set_frame_ms(h, ms)
#define DEC_SET_MS(h, ms) lc3_dec_set_frame_ms(h, ms)
#define ENC_SET_HR(h, v) lc3_enc_set_hrmode(h, v)
#define DEC_SET_HR(h, v) lc3_dec_set_hrmode(h, v)
#define ENC_SET_BR(h, v) lc3_enc_set_bitrate(h, v)
#define ENC_SET_BW(h, v) lc3_enc_set_bandwidth(h, v)
#define ENC_NB(h) lc3_enc_get_num_bytes(h)
#define ENC_N(h) lc3_enc_get_input_samples(h)
#define DEC_N(h) lc3_dec_get_output_samples(h)
#define ENC_FRAME(h, in, out, nb) lc3_enc_fl(h, in, 16, out, nb)
#define DEC_FRAME(h, in, nb, out) lc3_dec_fl(h, (void*)(in), nb, out, 16, 0)
#define ENC_FREE(h) lc3_free_encoder_structs(h)
#define DEC_FREE(h) lc3_free_decoder_structs(h)
#endif

static struct {
    int dec, fs, hr, ch, rate, S, T;
    float ms;
    const uint8_t* in; uint8_t* out; size_t out_unit;
    const int32_t* bw_plan;
} G;

typedef struct { int first, last, rc; long long sink; } job_t;

static void* work(void* arg)
{
    job_t* j = (job_t*)arg;
    uint8_t frame[2 * 1250];
    int16_t pcm_out[2][960];
    for (int s = j->first; s < j->last && !j->rc; s++) {
        const int size = G.dec ? DEC_SIZE(G.fs, G.ch) : ENC_SIZE(G.fs, G.ch);
        void* h = calloc(1, (size_t)size + 64);
        int rc = G.dec ? DEC_INIT(h, G.fs, G.ch) : ENC_INIT(h, G.fs, G.ch);
        if (!rc) rc = G.dec ? DEC_SET_MS(h, G.ms) : ENC_SET_MS(h, G.ms);
        if (!rc) rc = G.dec ? DEC_SET_HR(h, G.hr) : ENC_SET_HR(h, G.hr);
        if (!rc && !G.dec) rc = ENC_SET_BR(h, G.rate);
        if (rc) { j->rc = rc; free(h); break; }
        const int N = G.dec ? DEC_N(h) : ENC_N(h);
        for (int t = 0; t < G.T; t++) {
            if (!G.dec) {
                if (G.bw_plan && G.bw_plan[(size_t)s * G.T + t]) (void)ENC_SET_BW(h, G.bw_plan[(size_t)s * G.T + t]);
                const int16_t* p = (const int16_t*)G.in + ((size_t)s * G.T + t) * G.ch * N;
                void* in[2] = {(void*)p, (void*)(p + N)};
                int nb = ENC_NB(h);
                rc = ENC_FRAME(h, in, frame, &nb);
                j->sink += frame[3];
                if (G.out) memcpy(G.out + ((size_t)s * G.T + t) * G.out_unit, frame, (size_t)nb);
            } else {
                void* o[2] = {pcm_out[0], pcm_out[1]};
                rc = DEC_FRAME(h, G.in + ((size_t)s * G.T + t) * G.rate, G.rate, o);
                if (rc == 2) rc = 0;                       /* LC3_DECODE_ERROR: frame concealed, not a failure */
                j->sink += pcm_out[0][5];
                if (G.out) for (int c = 0; c < G.ch; c++) memcpy(G.out + (((size_t)s * G.T + t) * G.ch + c) * (size_t)N * 2, pcm_out[c], (size_t)N * 2);
            }
            if (rc) { j->rc = rc; break; }
        }
        if (G.dec) DEC_FREE(h); else ENC_FREE(h);
        free(h);
    }
    return 0;
}

int main(int argc, char** argv)
{
    if (argc != 11 && argc != 12) {
        fprintf(stderr, "usage: cpu_bench enc|dec FS FRAME_MS HRMODE CHANNELS BITRATE_OR_NBYTES N_STREAMS N_FRAMES THREADS IN_FILE [OUT_FILE]\n");
        return 2;
    }
    G.dec = strcmp(argv[1], "dec") == 0;
    G.fs = atoi(argv[2]); G.ms = (float)atof(argv[3]); G.hr = atoi(argv[4]); G.ch = atoi(argv[5]); G.rate = atoi(argv[6]);
    G.S = atoi(argv[7]); G.T = atoi(argv[8]);
    int P = atoi(argv[9]);
    if (P < 1) P = 1;
    if (P > G.S) P = G.S;
    FILE* f = fopen(argv[10], "rb");
    if (!f) { perror(argv[10]); return 2; }
    fseek(f, 0, SEEK_END); const long len = ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t* buf = (uint8_t*)malloc((size_t)len);
    if (!buf || fread(buf, 1, (size_t)len, f) != (size_t)len) { fprintf(stderr, "cpu_bench: cannot read %s\n", argv[10]); return 2; }
    fclose(f);
    G.in = buf;
    const int Nio = (int)((G.fs == 44100 ? 48000 : G.fs) * G.ms / 1000.0f + 0.5f);
    const size_t need = G.dec ? (size_t)G.S * G.T * G.rate : (size_t)G.S * G.T * G.ch * Nio * 2;
    if ((size_t)len < need) { fprintf(stderr, "cpu_bench: %s holds %ld bytes, %zu needed\n", argv[10], len, need); return 2; }
    size_t out_len = 0;
    if (argc == 12) {
        if (!G.dec) {       /* bytes per stream-frame: ask an encoder instance */
            void* h = calloc(1, (size_t)ENC_SIZE(G.fs, G.ch) + 64);
            if (ENC_INIT(h, G.fs, G.ch) || ENC_SET_MS(h, G.ms) || ENC_SET_HR(h, G.hr) || ENC_SET_BR(h, G.rate)) { fprintf(stderr, "cpu_bench: configuration rejected\n"); return 2; }
            G.out_unit = (size_t)ENC_NB(h);
            ENC_FREE(h); free(h);
            out_len = (size_t)G.S * G.T * G.out_unit;
        } else out_len = (size_t)G.S * G.T * G.ch * Nio * 2;
        G.out = (uint8_t*)calloc(1, out_len);
    }
    if (!G.dec && getenv("LC3_BENCH_BW_PLAN")) {
        FILE* bf = fopen(getenv("LC3_BENCH_BW_PLAN"), "rb");
        const size_t nw = (size_t)G.S * G.T;
        int32_t* plan = (int32_t*)malloc(nw * sizeof(int32_t));
        if (!bf || !plan || fread(plan, sizeof(int32_t), nw, bf) != nw) { fprintf(stderr, "cpu_bench: cannot read the bandwidth plan\n"); return 2; }
        fclose(bf);
        G.bw_plan = plan;
    }
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)P);
    job_t* jobs = (job_t*)calloc((size_t)P, sizeof(job_t));
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int i = 0; i < P; i++) {
        jobs[i].first = (int)((long long)G.S * i / P); jobs[i].last = (int)((long long)G.S * (i + 1) / P);
        if (pthread_create(&th[i], 0, work, &jobs[i])) { fprintf(stderr, "cpu_bench: pthread_create failed\n"); return 2; }
    }
    int rc = 0; long long sink = 0;
    for (int i = 0; i < P; i++) { pthread_join(th[i], 0); rc |= jobs[i].rc; sink += jobs[i].sink; }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (rc) { fprintf(stderr, "cpu_bench: codec error %d\n", rc); return 1; }
    if (G.out) { FILE* o = fopen(argv[11], "wb"); if (!o || fwrite(G.out, 1, out_len, o) != out_len) { perror(argv[11]); return 2; } fclose(o); }
    const double sec = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    printf("%lld %.6f %d %lld\n", (long long)G.S * G.T * G.ch, sec, P, sink & 1);
    return 0;
}
