/* A plain-C libsoxr client, used by tests/test_soxr_abi.py to exercise include/soxr.h + libsoxr.so
 * the way the reference binding does (src/soxr_ext.cpp): create -> process in pieces -> flush ->
 * delete; one-shot; split layout; pull mode; introspection calls.
 *
 *   soxr_client MODE in_rate out_rate channels dtype recipe piece infile outfile
 *     MODE  push | oneshot | pull | info
 *     dtype 0..7 (soxr_datatype_t), piece = frames per soxr_process call
 * Input file: raw samples, interleaved [frame][channel].  Output file: raw samples interleaved.
 * For split datatypes the client de-interleaves / re-interleaves itself (the library sees T**).
 */
#include <soxr.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static size_t esize(int dtype) { return soxr_datatype_size(dtype); }

typedef struct {
    char *data;
    size_t frames, pos, piece;
    unsigned ch;
    size_t es;
} feeder_t;

static size_t feed(void *state, soxr_in_t *data, size_t requested)
{
    feeder_t *f = (feeder_t *)state;
    size_t n = f->frames - f->pos;
    if (n > requested) n = requested;
    if (n > f->piece) n = f->piece;
    *data = f->data + f->pos * f->ch * f->es; /* non-NULL even at the end: 0 frames == end of input */
    f->pos += n;
    return n;
}

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    if (!strcmp(argv[1], "info")) {
        soxr_quality_spec_t q = soxr_quality_spec(SOXR_VHQ, 0);
        soxr_io_spec_t io = soxr_io_spec(SOXR_INT16_I, SOXR_INT16_I);
        printf("version=%s\nvhq_precision=%g\nvhq_passband_end=%.6f\nio_itype=%d\nio_scale=%g\n", soxr_version(),
               q.precision, q.passband_end, (int)io.itype, io.scale);
        return 0;
    }
    if (argc != 10) return 2;
    const char *mode = argv[1];
    double in_rate = atof(argv[2]), out_rate = atof(argv[3]);
    unsigned ch = (unsigned)atoi(argv[4]);
    int dtype = atoi(argv[5]);
    unsigned long recipe = strtoul(argv[6], 0, 10);
    size_t piece = (size_t)atol(argv[7]);
    size_t es = esize(dtype);
    int split = dtype & SOXR_SPLIT;

    FILE *fi = fopen(argv[8], "rb");
    if (!fi) return 3;
    fseek(fi, 0, SEEK_END);
    size_t bytes = (size_t)ftell(fi);
    fseek(fi, 0, SEEK_SET);
    size_t frames = bytes / (es * ch);
    char *x = (char *)malloc(bytes + 16);
    if (fread(x, 1, bytes, fi) != bytes) return 3;
    fclose(fi);

    size_t olen = (size_t)((double)frames * out_rate / in_rate) + 1016, opos = 0; /* slack for the pull loop */
    char *y = (char *)calloc(olen * ch, es);

    /* split layout: planar copies + pointer arrays */
    char *xs = 0, *ys = 0;
    void **xin = 0, **yout = 0;
    if (split) {
        xs = (char *)malloc(bytes + 16);
        ys = (char *)calloc(olen * ch, es);
        xin = (void **)malloc(sizeof(void *) * ch);
        yout = (void **)malloc(sizeof(void *) * ch);
        for (size_t f = 0; f < frames; ++f)
            for (unsigned c = 0; c < ch; ++c) memcpy(xs + (c * frames + f) * es, x + (f * ch + c) * es, es);
    }

    soxr_error_t err = 0;
    soxr_io_spec_t io = soxr_io_spec((soxr_datatype_t)dtype, (soxr_datatype_t)dtype);
    soxr_quality_spec_t q = soxr_quality_spec(recipe, 0);
    size_t clips = 0;
    double delay_mid = -1;

    if (!strcmp(mode, "oneshot")) {
        size_t idone = 0, odone = 0;
        err = soxr_oneshot(in_rate, out_rate, ch, x, frames, &idone, y, olen, &odone, &io, &q, NULL);
        if (!err && idone != frames) err = "idone != ilen";
        opos = odone;
    } else {
        soxr_t s = soxr_create(in_rate, out_rate, ch, &err, &io, &q, NULL);
        if (!s || err) { fprintf(stderr, "soxr_create: %s\n", err ? err : "?"); return 4; }
        if (!strcmp(mode, "pull")) {
            feeder_t f = {x, frames, 0, piece, ch, es};
            err = soxr_set_input_fn(s, feed, &f, piece);
            while (!err) {
                size_t want = 1000, got = soxr_output(s, y + opos * ch * es, want);
                opos += got;
                if (got < want) break;
                if (opos + want > olen) { err = "output overrun"; break; }
            }
            if (!err) err = soxr_error(s);
        } else { /* push: the reference's csoxr_divide_proc / csoxr_split_ch loop */
            for (size_t idx = 0; idx <= frames && !err; idx += piece) {
                int last = idx + piece >= frames;
                size_t n = last ? frames - idx : piece, odone = 0;
                const void *in;
                void *out;
                if (split) {
                    for (unsigned c = 0; c < ch; ++c) {
                        xin[c] = xs + (c * frames + idx) * es;
                        yout[c] = ys + (c * olen + opos) * es;
                    }
                    in = xin; out = yout;
                } else {
                    in = x + idx * ch * es; out = y + opos * ch * es;
                }
                err = soxr_process(s, in, n, NULL, out, olen - opos, &odone);
                opos += odone;
                if (delay_mid < 0) delay_mid = soxr_delay(s);
                if (last) break;
            }
            while (!err) { /* flush until dry (src/soxr_ext.cpp:109-127) */
                size_t odone = 0;
                void *out = y + opos * ch * es;
                if (split) {
                    for (unsigned c = 0; c < ch; ++c) yout[c] = ys + (c * olen + opos) * es;
                    out = yout;
                }
                err = soxr_process(s, NULL, 0, NULL, out, olen - opos, &odone);
                opos += odone;
                if (!odone) break;
            }
        }
        clips = *soxr_num_clips(s);
        printf("engine=%s\ndelay_after_first=%g\ndelay_end=%g\n", soxr_engine(s), delay_mid, soxr_delay(s));
        if (!err) err = soxr_clear(s);
        soxr_delete(s);
    }
    if (err) { fprintf(stderr, "error: %s\n", err); return 5; }
    if (split)
        for (size_t f = 0; f < opos; ++f)
            for (unsigned c = 0; c < ch; ++c) memcpy(y + (f * ch + c) * es, ys + (c * olen + f) * es, es);
    FILE *fo = fopen(argv[9], "wb");
    if (!fo) return 3;
    fwrite(y, es * ch, opos, fo);
    fclose(fo);
    printf("frames_out=%zu\nclips=%zu\n", opos, clips);
    return 0;
}
