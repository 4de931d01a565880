/* lc3plus_enc_cli.c -- WAV -> .lc3plus / G.192 encoder front end on top of the C ABI (include/lc3.h, lc3plus_batch.h).
 *
 * Mirrors the ENCODE mode of the ETSI command line tool (R/codec_exe.c, R = LC3plus_ETSI_src_v17171_20200723/src/floating_point):
 * same positional arguments and the options that matter for encoding, same frame loop semantics (last frame zero padded,
 * R/codec_exe.c:329-338), same bitstream container (R/codec_exe.c:636-668 header, :737-749 frames, :705-735 G.192), so its
 * output files are byte-comparable with `LC3plus -E ...` of the reference.  A file is ONE stream, i.e. strictly sequential
 * in time: the whole file is pushed through lc3plus_enc_batch_encode() in chunks of frames so that the encoder state stays
 * on the GPU between frames of a chunk.
 *
 *   lc3plus_enc_cli [-E] [-q] [-frame_ms 2.5|5|10] [-hrmode] [-bandwidth HZ|FILE] [-swf FILE] [-formatG192] [-cfgG192 FILE]
 *                   in.wav out.lc3plus BITRATE|FILE
 *
 * Switching files (R/codec_exe.c:296-326, loopy_read64 :858-866) hold one int64 per frame and wrap around: the bitrate per channel
 * (-swf FILE, or a file name in place of BITRATE) and the audio bandwidth in Hz (-bandwidth FILE).  Frames with equal settings are
 * still pushed through the GPU in runs; a change of setting ends the run.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "lc3.h"
#include "lc3plus_batch.h"

static void die(const char* msg) { fprintf(stderr, "lc3plus_enc_cli: %s\n", msg); exit(1); }

static int64_t loopy_read64(FILE* f)               /* R/codec_exe.c:858-866 */
{
    int64_t tmp = 0;
    if (fread(&tmp, sizeof tmp, 1, f) != 1) { fseek(f, 0, SEEK_SET); if (fread(&tmp, sizeof tmp, 1, f) != 1) die("empty switching file"); }
    return tmp;
}

typedef struct { int rate, channels, bits; uint32_t frames; uint8_t* data; } wav_t;

static uint32_t rd32(const uint8_t* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

static void read_wav(const char* path, wav_t* w)
{
    FILE* f = fopen(path, "rb");
    if (!f) die("Error opening wav file!");
    fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t* buf = (uint8_t*)malloc(sz);
    if (!buf || fread(buf, 1, sz, f) != (size_t)sz) die("short read");
    fclose(f);
    if (sz < 44 || memcmp(buf, "RIFF", 4) || memcmp(buf + 8, "WAVE", 4)) die("not a RIFF/WAVE file");
    long pos = 12; int have_fmt = 0;
    memset(w, 0, sizeof *w);
    while (pos + 8 <= sz) {
        uint32_t len = rd32(buf + pos + 4);
        if (!memcmp(buf + pos, "fmt ", 4)) {
            if (rd16(buf + pos + 8) != 1 && rd16(buf + pos + 8) != 0xFFFE) die("only PCM wav is supported");
            w->channels = rd16(buf + pos + 10); w->rate = (int)rd32(buf + pos + 12); w->bits = rd16(buf + pos + 22); have_fmt = 1;
        } else if (!memcmp(buf + pos, "data", 4)) {
            if (!have_fmt) die("data chunk before fmt chunk");
            if (pos + 8 + (long)len > sz) len = (uint32_t)(sz - pos - 8);
            w->frames = len / (uint32_t)(w->channels * w->bits / 8);
            w->data = buf + pos + 8;
            return;
        }
        pos += 8 + len + (len & 1);
    }
    die("no data chunk");
}

int main(int ac, char** av)
{
    float frame_ms = 10; int hrmode = 0, g192 = 0, bandwidth = 0, quiet = 0; const char* cfg = NULL;
    const char* swf = NULL; const char* bwf = NULL;
    int i = 1;
    for (; i < ac && av[i][0] == '-'; i++) {
        if (!strcmp(av[i], "-E")) continue;
        else if (!strcmp(av[i], "-q")) quiet = 1;
        else if (!strcmp(av[i], "-frame_ms") && i + 1 < ac) frame_ms = (float)atof(av[++i]);
        else if (!strcmp(av[i], "-hrmode")) hrmode = 1;
        else if (!strcmp(av[i], "-bandwidth") && i + 1 < ac) { bandwidth = atoi(av[++i]); if (bandwidth == 0) bwf = av[i]; }
        else if (!strcmp(av[i], "-swf") && i + 1 < ac) swf = av[++i];
        else if (!strcmp(av[i], "-formatG192")) g192 = 1;
        else if (!strcmp(av[i], "-cfgG192") && i + 1 < ac) cfg = av[++i];
        else die("unknown option (encode-only front end)");
    }
    if (ac - i != 3) die("usage: lc3plus_enc_cli [options] in.wav out.lc3plus bitrate");
    const char* in = av[i]; const char* outp = av[i + 1]; int bitrate = atoi(av[i + 2]);
    if (bitrate == 0) { bitrate = 64000; swf = av[i + 2]; }          /* a file name in place of the bitrate (R/codec_exe.c:598-604) */
    FILE* fswf = swf ? fopen(swf, "rb") : NULL; FILE* fbwf = bwf ? fopen(bwf, "rb") : NULL;
    if (swf && !fswf) die("Error opening bitrate switching file!");
    if (bwf && !fbwf) die("Error opening bandwidth switching file!");

    wav_t w; read_wav(in, &w);
    if (w.bits != 16 && w.bits != 24 && w.bits != 32) die("unsupported sample width");

    /* the single-stream API validates the configuration exactly like the reference CLI does (R/codec_exe.c:177-199) */
    LC3_Enc* enc = (LC3_Enc*)malloc(lc3_enc_get_size(w.rate, w.channels) > 0 ? lc3_enc_get_size(w.rate, w.channels) : 1);
    LC3_Error err = lc3_enc_init(enc, w.rate, w.channels);
    if (!err) err = lc3_enc_set_frame_ms(enc, frame_ms);
    if (!err) err = lc3_enc_set_hrmode(enc, hrmode);
    if (!err) err = lc3_enc_set_bitrate(enc, bitrate);
    if (err) { fprintf(stderr, "lc3plus_enc_cli: configuration error %d\n", (int)err); return 1; }
    const int N = lc3_enc_get_input_samples(enc), C = w.channels; int nbytes = lc3_enc_get_num_bytes(enc);
    if (!quiet) printf("Sample rate: %d  Channels: %d  Frame length: %d  Target bitrate: %d  Real bitrate: %d  Bytes/frame: %d\n",
                       w.rate, C, N, bitrate, lc3_enc_get_real_bitrate(enc), nbytes);
    lc3_enc_free_memory(enc);

    lc3plus_batch* b = NULL;
    err = lc3plus_enc_batch_create(&b, 1, w.rate, C, frame_ms, hrmode, &bitrate, -1);
    if (err) { fprintf(stderr, "lc3plus_enc_cli: cannot create the GPU encoder (LC3_Error %d)\n", (int)err); return 1; }
    nbytes = lc3plus_enc_batch_num_bytes(b, 0);
    if (bandwidth) { err = lc3plus_enc_batch_set_bandwidth(b, 0, bandwidth); if (err && err < LC3_WARNING) die("bandwidth error"); }
    if (fbwf) bandwidth = 0;

    FILE* fo = fopen(outp, "wb");
    if (!fo) die("Error creating bitstream file!");
    {   /* container header: R/codec_exe.c:651-661 (20 bytes; goes to the .cfg side file for G.192) */
        uint16_t header[10] = {0xcc1c, 20, (uint16_t)(w.rate / 100), (uint16_t)(bitrate / 100), (uint16_t)C, (uint16_t)(frame_ms * 100), 0,
                               (uint16_t)w.frames, (uint16_t)(w.frames >> 16), (uint16_t)hrmode};
        FILE* fh = fo;
        if (g192) {
            char* name = NULL;
            if (!cfg) { name = (char*)malloc(strlen(outp) + 5); sprintf(name, "%s.cfg", outp); cfg = name; }
            fh = fopen(cfg, "wb");
            if (!fh) die("Error opening G192 configuration-file!");
            free(name);
        }
        fwrite(header, sizeof header, 1, fh);
        if (g192) fclose(fh);
    }

    const uint32_t total_frames = (w.frames + (uint32_t)N - 1) / (uint32_t)N;
    const int CH = 256;                                   /* frames per launch */
    const int wide = w.bits != 16;
    void* pcm = calloc((size_t)CH * C * N, wide ? 4 : 2);
    uint8_t* out = (uint8_t*)malloc((size_t)CH * LC3_MAX_BYTES * 2);
    const int bps = w.bits / 8;
    int cur_bitrate = bitrate, cur_bw = bandwidth;
    uint32_t f0 = 0;
    while (f0 < total_frames) {
        /* settings of frame f0 (the reference applies them before reading the frame, R/codec_exe.c:296-326) */
        int64_t nbr = fswf ? loopy_read64(fswf) * C : cur_bitrate, nbw = fbwf ? loopy_read64(fbwf) : cur_bw;
        if ((int)nbr != cur_bitrate) {
            err = lc3plus_enc_batch_set_bitrate(b, 0, (int)nbr); if (err) { fprintf(stderr, "lc3plus_enc_cli: bitrate switch failed (LC3_Error %d)\n", (int)err); return 1; }
            cur_bitrate = (int)nbr; nbytes = lc3plus_enc_batch_num_bytes(b, 0);
        }
        if ((int)nbw != cur_bw) {
            err = lc3plus_enc_batch_set_bandwidth(b, 0, (int)nbw); if (err && err < LC3_WARNING) die("bandwidth error");
            cur_bw = (int)nbw;
        }
        /* extend the run while the switching files keep the settings */
        int T = 1;
        while (T < CH && f0 + T < total_frames) {
            if (fswf || fbwf) {
                const long ps = fswf ? ftell(fswf) : 0, pb = fbwf ? ftell(fbwf) : 0;
                const int64_t br2 = fswf ? loopy_read64(fswf) * C : cur_bitrate, bw2 = fbwf ? loopy_read64(fbwf) : cur_bw;
                if ((int)br2 != cur_bitrate || (int)bw2 != cur_bw) { if (fswf) fseek(fswf, ps, SEEK_SET); if (fbwf) fseek(fbwf, pb, SEEK_SET); break; }
            }
            T++;
        }
        memset(pcm, 0, (size_t)T * C * N * (wide ? 4 : 2));
        for (int t = 0; t < T; t++) for (int n = 0; n < N; n++) {                 /* de-interleave into [frame][channel][N] */
            const uint64_t s = (uint64_t)(f0 + t) * N + n;
            if (s >= w.frames) break;
            for (int c = 0; c < C; c++) {
                const uint8_t* p = w.data + (s * C + c) * bps;
                const size_t o = ((size_t)t * C + c) * N + n;
                if (w.bits == 16) ((int16_t*)pcm)[o] = (int16_t)rd16(p);
                else if (w.bits == 24) ((int32_t*)pcm)[o] = ((int32_t)((uint32_t)p[0] << 8 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 24)) >> 8;
                else ((int32_t*)pcm)[o] = ((int32_t)rd32(p)) >> 8;   /* the reference reader narrows 32-bit WAV to 24 bit (R/tinywavein_c.h:528-533) and still calls lc3_enc32 */
            }
        }
        err = lc3plus_enc_batch_encode(b, pcm, 0, w.bits, T, out, nbytes, 0, NULL, 1);
        if (err) { fprintf(stderr, "lc3plus_enc_cli: encode failed (LC3_Error %d)\n", (int)err); return 1; }
        for (int t = 0; t < T; t++) {
            const uint8_t* fr = out + (size_t)t * nbytes;
            if (g192) {                                   /* R/codec_exe.c:705-735 */
                const uint16_t sync = 0x6B21, nbits = (uint16_t)(nbytes * 8);
                fwrite(&sync, 2, 1, fo); fwrite(&nbits, 2, 1, fo);
                for (int k = 0; k < nbytes; k++) for (int bit = 0; bit < 8; bit++) {
                    const int16_t v = (fr[k] >> bit) & 1 ? 0x0081 : 0x007F;
                    fwrite(&v, 2, 1, fo);
                }
            } else {                                      /* R/codec_exe.c:742-748 */
                const uint16_t nb = (uint16_t)nbytes;
                fwrite(&nb, 2, 1, fo);
                fwrite(fr, 1, nbytes, fo);
            }
        }
        f0 += (uint32_t)T;
        if (!quiet) { printf("\rProcessing frame %u", f0); fflush(stdout); }
    }
    if (!quiet) puts("\nProcessing done!");
    fclose(fo);
    lc3plus_enc_batch_destroy(b);
    free(pcm); free(out);
    return 0;
}
