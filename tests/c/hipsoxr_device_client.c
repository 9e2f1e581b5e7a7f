/* A plain-C client of the NATIVE ABI's device-chunk entry (include/hipsoxr.h: hipsoxr_stream_process_device), used by
 * tests/test_gpu_device_client.py: no Python, no torch — HIP runtime calls for memory and a stream, the library for the
 * rest.  The call pattern is the reference binding's push loop + flush (src/soxr_ext.cpp:129-187, :109-127) with device
 * pointers: upload the whole input once, then per piece  process_device(d_in + pos, n, d_out + done, room)  on one HIP
 * stream, flush with in == NULL until 0 frames come back, ONE synchronisation at the end, download.
 *
 *   hipsoxr_device_client in_rate out_rate channels elem recipe piece vr infile outfile
 *     elem 0..3 (hipsoxr_elem_t: f32 f64 i32 i16), piece = frames per call, vr = 0 | 1 (1: the io ratio is halved with a
 *     300-frame slew after the third call)
 */
#include <hip/hip_runtime_api.h>
#include <hipsoxr.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 3; } } while (0)
#define SOXR(x) do { hipsoxr_error_t e_ = (x); if (e_) { fprintf(stderr, "%s: %s\n", #x, e_); return 4; } } while (0)

int main(int argc, char **argv)
{
    if (argc < 10) return 2;
    const double in_rate = atof(argv[1]), out_rate = atof(argv[2]);
    const unsigned ch = (unsigned)atoi(argv[3]);
    const int elem = atoi(argv[4]);
    const unsigned long recipe = (unsigned long)atoi(argv[5]);
    const size_t piece = (size_t)atol(argv[6]);
    const int vr = atoi(argv[7]);
    static const size_t es_of[4] = {4, 8, 4, 2};
    const size_t es = es_of[elem & 3], frame = es * ch;

    FILE *f = fopen(argv[8], "rb");
    if (!f) return 5;
    fseek(f, 0, SEEK_END);
    const size_t bytes = (size_t)ftell(f), frames = bytes / frame;
    fseek(f, 0, SEEK_SET);
    char *h_in = (char *)malloc(bytes ? bytes : 1);
    if (fread(h_in, 1, bytes, f) != bytes) return 5;
    fclose(f);

    hipStream_t st;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const size_t cap = (size_t)((double)frames * (vr ? 2.2 : 1.0) * out_rate / in_rate) + 4096;
    char *d_in = NULL, *d_out = NULL;
    CHECK(hipMalloc((void **)&d_in, bytes ? bytes : 16));
    CHECK(hipMalloc((void **)&d_out, cap * frame));
    CHECK(hipMemcpyAsync(d_in, h_in, bytes, hipMemcpyHostToDevice, st));

    hipsoxr_stream_t *s = NULL;
    SOXR(hipsoxr_stream_create(in_rate, out_rate, ch, (hipsoxr_datatype_t)elem, recipe, vr ? HIPSOXR_VR : 0, &s));
    size_t pos = 0, done = 0, calls = 0;
    while (pos < frames) {
        const size_t n = frames - pos < piece ? frames - pos : piece;
        size_t od = 0;
        if (vr && calls == 3) SOXR(hipsoxr_stream_set_io_ratio(s, in_rate / out_rate / 2., 300));
        SOXR(hipsoxr_stream_process_device(s, d_in + pos * frame, n, d_out + done * frame, cap - done, &od, st));
        pos += n; done += od; ++calls;
    }
    for (;;) { /* flush */
        size_t od = 0;
        SOXR(hipsoxr_stream_process_device(s, NULL, 0, d_out + done * frame, cap - done, &od, st));
        if (!od) break;
        done += od;
    }
    char *h_out = (char *)malloc(done * frame + 1);
    CHECK(hipMemcpyAsync(h_out, d_out, done * frame, hipMemcpyDeviceToHost, st));
    CHECK(hipStreamSynchronize(st)); /* the only wait of the run */
    printf("frames_out=%zu\ncalls=%zu\ndelay_end=%.6f\nengine=%s\n", done, calls, hipsoxr_stream_delay(s), hipsoxr_stream_engine(s));
    hipsoxr_stream_delete(s);
    f = fopen(argv[9], "wb");
    if (!f || fwrite(h_out, 1, done * frame, f) != done * frame) return 6;
    fclose(f);
    CHECK(hipFree(d_in)); CHECK(hipFree(d_out)); CHECK(hipStreamDestroy(st));
    free(h_in); free(h_out);
    return 0;
}
