/* lc3plus_dec_cli.c -- .lc3plus / G.192 -> WAV decoder front end on top of the C ABI (include/lc3.h, lc3plus_batch.h).
 *
 * Mirrors the DECODE mode of the ETSI command line tool (R/codec_exe.c, R = LC3plus_ETSI_src_v17171_20200723/src/floating_point):
 * same container reader (R/codec_exe.c:670-703 header, :751-815 frames), same options that matter for decoding, same frame loop
 * (error pattern file :397-399, delay compensation :230,:433-435, zero padding of a short tail :447-450) and the same WAV writer
 * conventions (R/tinywaveout_c.h: 44-byte header, 16 / 24 / 32 bit little-endian PCM), so its output files are byte-comparable
 * with `LC3plus -D ...` of the reference.  A file is ONE stream: it is pushed through lc3plus_dec_batch_decode() in runs of frames
 * of equal size so that the decoder memories stay on the GPU between the frames of a run.
 *
 *   lc3plus_dec_cli [-D] [-q] [-bps 16|24|32] [-dc 0|1|2] [-epf FILE] [-edf FILE] [-formatG192] [-cfgG192 FILE] in.lc3plus out.wav
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "lc3.h"
#include "lc3plus_batch.h"

static void die(const char* msg) { fprintf(stderr, "lc3plus_dec_cli: %s\n", msg); exit(1); }

static int16_t loopy_read16(FILE* f)               /* R/codec_exe.c:818-856, including the G.192 style frame-erasure files */
{
    static int first = -1;
    int16_t tmp = 0;
    if (fread(&tmp, sizeof tmp, 1, f) != 1) { fseek(f, 0, SEEK_SET); if (fread(&tmp, sizeof tmp, 1, f) != 1) die("empty error pattern file"); }
    if (first < 0) first = (uint16_t)tmp;
    if ((first == 0x6B20 || first == 0x6B21) && ((uint16_t)tmp == 0x6B20 || (uint16_t)tmp == 0x6B21)) tmp = (int16_t)(0x6B21 - (uint16_t)tmp);
    return tmp;
}

typedef struct { uint8_t* data; int nbytes; int bfi; } frame_t;

int main(int ac, char** av)
{
    int quiet = 0, bps = 16, dc = 1, g192 = 0;
    const char *epf = NULL, *edf = NULL, *cfg = NULL;
    int i = 1;
    for (; i < ac && av[i][0] == '-'; i++) {
        if (!strcmp(av[i], "-D")) continue;
        else if (!strcmp(av[i], "-q")) quiet = 1;
        else if (!strcmp(av[i], "-bps") && i + 1 < ac) { bps = atoi(av[++i]); if (bps != 16 && bps != 24 && bps != 32) die("Only 16, 24 or 32 bits per sample are supported!"); }
        else if (!strcmp(av[i], "-dc") && i + 1 < ac) { dc = atoi(av[++i]); if (dc < 0 || dc > 2) die("dc musst be 0, 1 or 2!"); }
        else if (!strcmp(av[i], "-epf") && i + 1 < ac) epf = av[++i];
        else if (!strcmp(av[i], "-edf") && i + 1 < ac) edf = av[++i];
        else if (!strcmp(av[i], "-formatG192")) g192 = 1;
        else if (!strcmp(av[i], "-cfgG192") && i + 1 < ac) cfg = av[++i];
        else die("unknown option (decode-only front end)");
    }
    if (ac - i != 2) die("usage: lc3plus_dec_cli [options] in.lc3plus out.wav");
    const char* in = av[i]; const char* outp = av[i + 1];

    /* ---- container header R/codec_exe.c:670-703 ---- */
    FILE* fi = fopen(in, "rb");
    if (!fi) die("Error opening bitstream file!");
    uint16_t header[10] = {0};
    {
        FILE* fh = fi;
        if (g192) {
            char* name = NULL;
            if (!cfg) { name = (char*)malloc(strlen(in) + 5); sprintf(name, "%s.cfg", in); cfg = name; }
            fh = fopen(cfg, "rb");
            if (!fh) die("Error opening G192 configuration-file!");
            free(name);
        }
        if (fread(header, sizeof header, 1, fh) != 1 && !g192) die("short bitstream header");
        if (header[1] < 18) die("not an LC3plus bitstream (header size)");
        fseek(fh, header[1], SEEK_SET);
        if (g192) fclose(fh);
    }
    const int rate = header[2] * 100, C = header[4], hrmode = header[1] > 18 ? header[9] : 0;
    const float frame_ms = (float)(header[5] / 100.0);
    uint32_t n_file = (uint32_t)header[7] | ((uint32_t)header[8] << 16);
    if (header[6] != 0) die("error protected streams are not supported by the float codec");

    /* the single-stream API validates the configuration exactly like the reference CLI does (R/codec_exe.c:216-231) */
    const int dsz = lc3_dec_get_size(rate, C);
    LC3_Dec* d = (LC3_Dec*)malloc(dsz > 0 ? dsz : 1);
    LC3_Error err = lc3_dec_init(d, rate, C, LC3_PLC_STANDARD);
    if (!err) err = lc3_dec_set_hrmode(d, hrmode);
    if (!err) err = lc3_dec_set_frame_ms(d, frame_ms);
    if (err) { fprintf(stderr, "lc3plus_dec_cli: configuration error %d\n", (int)err); return 1; }
    const int N = lc3_dec_get_output_samples(d);
    int delay = dc ? lc3_dec_get_delay(d) / dc : 0;
    lc3_dec_free_memory(d);
    if (!quiet) printf("Sample rate: %d  Channels: %d  Frame length: %d  Signal length: %u  Output format: %d bit\n", rate, C, N, n_file, bps);

    /* ---- all frames of the file R/codec_exe.c:751-815 ---- */
    FILE* fep = epf ? fopen(epf, "rb") : NULL;
    if (epf && !fep) die("Error opening error pattern file!");
    size_t cap = 1024, nf = 0;
    frame_t* fr = (frame_t*)malloc(cap * sizeof *fr);
    for (;;) {
        uint8_t bytes[LC3_MAX_BYTES * 2]; int nb = 0, bfi = 0;
        if (g192) {
            int16_t ind = 0, bit = 0; uint16_t nbits = 0;
            if (fread(&ind, 2, 1, fi) != 1) break;
            if (fread(&nbits, 2, 1, fi) != 1) nbits = 0;
            if ((uint16_t)ind != 0x6B21 && (uint16_t)ind != 0x6B20 && (uint16_t)ind != 0x6B22)
                die("Wrong G192 format detected in bitstream file! The sync word could not be recognized!");
            nb = (int16_t)(nbits / 8);
            for (int k = 0; k < nb && k < (int)sizeof bytes; k++) {
                int byte = 0;
                for (int j = 0; j < 8; j++) { if (fread(&bit, 2, 1, fi) != 1) bit = 0; if ((uint16_t)bit == 0x0081) byte |= 1 << j; }
                bytes[k] = (uint8_t)byte;
            }
            if ((uint16_t)ind == 0x6B20) { nb = 0; bfi = 1; } else if ((uint16_t)ind == 0x6B22) bfi = 3;
        } else {
            uint16_t n16 = 0;
            if (fread(&n16, 2, 1, fi) != 1) break;
            nb = n16;
            for (int k = 0; k < nb && k < (int)sizeof bytes; k++) bytes[k] = (uint8_t)getc(fi);
        }
        if (fep && loopy_read16(fep)) nb = 0;                               /* R/codec_exe.c:397-399 */
        if (nb > LC3_MAX_BYTES) die("frame too large");
        if (nf == cap) { cap *= 2; fr = (frame_t*)realloc(fr, cap * sizeof *fr); if (!fr) die("out of memory"); }
        fr[nf].nbytes = nb; fr[nf].bfi = bfi;
        fr[nf].data = (uint8_t*)malloc(nb > 0 ? nb : 1);
        memcpy(fr[nf].data, bytes, nb);
        nf++;
    }
    fclose(fi);

    /* ---- output WAV R/tinywaveout_c.h:316-375: sizes are patched at the end ---- */
    FILE* fo = fopen(outp, "wb+");
    if (!fo) die("Error creating wav file!");
    {
        const uint32_t ba = (uint32_t)C * (uint32_t)(bps >> 3), bytes_s = (uint32_t)rate * ba, m1 = 0xffffffffu, m2 = 0xffffffffu - 36u, sixteen = 16;
        const uint16_t tag = 1, ch16 = (uint16_t)C, ba16 = (uint16_t)ba, bps16 = (uint16_t)bps; const uint32_t sr = (uint32_t)rate;
        fwrite("RIFF", 1, 4, fo); fwrite(&m1, 4, 1, fo); fwrite("WAVE", 1, 4, fo);
        fwrite("fmt ", 1, 4, fo); fwrite(&sixteen, 4, 1, fo); fwrite(&tag, 2, 1, fo); fwrite(&ch16, 2, 1, fo); fwrite(&sr, 4, 1, fo);
        fwrite(&bytes_s, 4, 1, fo); fwrite(&ba16, 2, 1, fo); fwrite(&bps16, 2, 1, fo);
        fwrite("data", 1, 4, fo); fwrite(&m2, 4, 1, fo);
    }
    FILE* fed = edf ? fopen(edf, "wb") : NULL;
    if (edf && !fed) die("Error creating error detection file!");

    /* ---- decode in runs of equal frame size ---- */
    lc3plus_dec_batch* b = NULL;
    err = lc3plus_dec_batch_create(&b, 1, rate, C, frame_ms, hrmode, NULL, -1);
    if (err) { fprintf(stderr, "lc3plus_dec_cli: cannot create the GPU decoder (LC3_Error %d)\n", (int)err); return 1; }
    const int CH = 256;
    uint8_t* in_buf = (uint8_t*)calloc((size_t)CH, LC3_MAX_BYTES);
    uint8_t* flags = (uint8_t*)malloc(CH); uint8_t* status = (uint8_t*)malloc(CH);
    void* pcm = malloc((size_t)CH * C * N * 4);
    uint32_t data_bytes = 0, clipped = 0;
    int cur = 0;                                                            /* bytes per frame the decoder is configured for */
    size_t f0 = 0;
    while (f0 < nf) {
        /* R/dec_lc3_fl.c:140-155: a lost frame (bfi_ext = 1 or num_bytes = 0) keeps the configuration, a good one of another size changes it */
        int lost0 = fr[f0].bfi == 1 || (fr[f0].bfi == 0 && fr[f0].nbytes == 0);
        if (!lost0 && fr[f0].nbytes != cur) {
            err = lc3plus_dec_batch_set_num_bytes(b, 0, fr[f0].nbytes);
            if (err) { fprintf(stderr, "lc3plus_dec_cli: frame size %d rejected (LC3_Error %d)\n", fr[f0].nbytes, (int)err); return 1; }
            cur = fr[f0].nbytes;
        }
        int T = 0;
        const int stride = cur > 0 ? cur : 1;
        while (T < CH && f0 + T < nf) {
            const frame_t* f = &fr[f0 + T];
            const int lost = f->bfi == 1 || (f->bfi == 0 && f->nbytes == 0);
            if (!lost && f->nbytes != cur) break;
            memset(in_buf + (size_t)T * stride, 0, stride);
            if (!lost) memcpy(in_buf + (size_t)T * stride, f->data, f->nbytes);
            flags[T] = lost ? 1 : (uint8_t)f->bfi;
            T++;
        }
        err = lc3plus_dec_batch_decode(b, in_buf, 0, stride, flags, T, pcm, 0, bps, status, NULL, 1);
        if (err) { fprintf(stderr, "lc3plus_dec_cli: decode failed (LC3_Error %d)\n", (int)err); return 1; }
        for (int t = 0; t < T; t++) {
            if (fed) { const int16_t e = status[t]; fwrite(&e, 2, 1, fed); }
            /* R/codec_exe.c:430-435: interleave, skip the codec delay at the start, stop at the signal length of the header */
            uint32_t n_out = (uint32_t)(N - delay) < n_file ? (uint32_t)(N - delay) : n_file;
            for (uint32_t n = 0; n < n_out; n++) for (int c = 0; c < C; c++) {
                const size_t o = ((size_t)t * C + c) * N + delay + n;
                if (bps == 16) { const int16_t v = ((const int16_t*)pcm)[o]; fwrite(&v, 2, 1, fo); data_bytes += 2; }
                else if (bps == 24) {                                       /* R/tinywaveout_c.h:403-424 (clip to 24 bit, 3 bytes) */
                    int32_t v = ((const int32_t*)pcm)[o];
                    if (v > 8388607) { v = 8388607; clipped++; } else if (v < -8388608) { v = -8388608; clipped++; }
                    fwrite(&v, 3, 1, fo); data_bytes += 3;
                } else { const int32_t v = ((const int32_t*)pcm)[o]; fwrite(&v, 4, 1, fo); data_bytes += 4; }
            }
            n_file -= (uint32_t)(N - delay);                                /* unsigned wrap-around as in the reference */
            delay = 0;
        }
        f0 += (size_t)T;
        if (!quiet) { printf("\rProcessing frame %zu", f0); fflush(stdout); }
    }
    if (n_file > 0 && n_file < (uint32_t)N) {                               /* R/codec_exe.c:447-450 */
        const int32_t zero = 0;
        for (uint32_t n = 0; n < n_file * (uint32_t)C; n++) { fwrite(&zero, bps >> 3, 1, fo); data_bytes += (uint32_t)(bps >> 3); }
    }
    {   /* R/tinywaveout_c.h:562-586 */
        const uint32_t riff = 36 - 8 + 8 + data_bytes;
        fseek(fo, 4, SEEK_SET); fwrite(&riff, 4, 1, fo);
        fseek(fo, 40, SEEK_SET); fwrite(&data_bytes, 4, 1, fo);
    }
    fclose(fo);
    if (fed) fclose(fed);
    if (fep) fclose(fep);
    if (!quiet) { puts("\nProcessing done!"); printf("%u samples clipped!\n", clipped); }
    lc3plus_dec_batch_destroy(b);
    for (size_t k = 0; k < nf; k++) free(fr[k].data);
    free(fr); free(in_buf); free(flags); free(status); free(pcm);
    return 0;
}
