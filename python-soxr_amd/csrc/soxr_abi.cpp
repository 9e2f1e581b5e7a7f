// soxr_abi.cpp — the libsoxr-named C ABI (include/soxr.h) on top of the native one
// (include/hipsoxr.h).  SURVEY.md §8(b)(i) / §8(f)-4: what `USE_SYSTEM_LIBSOXR=ON` links
// (reference CMakeLists.txt:83-93); every function below names the reference call site it serves.
//
// The shim holds no arithmetic: it translates spec structures to the (datatype, recipe, flags)
// triple of hipsoxr_stream_create and forwards.  Restrictions are errors, not silent fallbacks.
#include "../../include/soxr.h"

#include <cmath>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/hipsoxr.h"
#include "plan.h"

struct soxr {
    hipsoxr_stream_t *h = nullptr;
    size_t clips = 0;              // soxr_num_clips returns a pointer (src/soxr_ext.cpp:190 dereferences it)
    soxr_error_t err = nullptr;
    soxr_input_fn_t fn = nullptr;  // pull mode
    void *fn_state = nullptr;
    size_t max_ilen = 0;
    unsigned channels = 0;
    int dtype = 0;                 // soxr_datatype_t
    double io_ratio = 1.;          // input frames per output frame (pull mode: how much input an output request needs)
    bool input_ended = false;
};

namespace {

size_t sample_size(int dtype) { return (size_t)soxr_datatype_size(dtype); }

// `frames` further into an output (or input) buffer of the handle's layout
struct Cursor {
    std::vector<void *> split; // storage for the per-channel pointer array of split layouts
    void *at(const soxr *p, void *base, size_t frames)
    {
        if (!(p->dtype & SOXR_SPLIT)) return (char *)base + frames * p->channels * sample_size(p->dtype);
        split.resize(p->channels);
        void *const *chan = (void *const *)base;
        for (unsigned c = 0; c < p->channels; ++c) split[c] = (char *)chan[c] + frames * sample_size(p->dtype);
        return split.data();
    }
};

// quality spec -> recipe index understood by plan.cpp.  precision decides; LQ and MQ share 16 bits
// and differ by their pass band (1385/2048 vs the formula).
soxr_error_t recipe_of(const soxr_quality_spec_t *q, unsigned long *recipe)
{
    if (!q) { *recipe = HIPSOXR_HQ; return nullptr; }
    if (q->phase_response != 50.) return "hipsoxr: only linear phase response is implemented";
    const double b = q->precision;
    if (!(b >= 0.) || b > 32.) return "invalid precision";
    if (b == 0.) *recipe = HIPSOXR_QQ;
    else if (b <= 16.) *recipe = q->passband_end < .8 ? HIPSOXR_LQ : HIPSOXR_MQ;
    else *recipe = (unsigned long)std::lround(std::ceil((b - 4.) / 4.));
    return nullptr;
}

soxr_error_t translate(const soxr_io_spec_t *io, const soxr_quality_spec_t *q, int *dtype, unsigned long *recipe,
                       unsigned long *flags)
{
    *dtype = SOXR_FLOAT32_I;
    *flags = 0;
    if (io) {
        if (io->itype != io->otype) return "hipsoxr: input and output sample types must be equal";
        if ((unsigned)io->itype > 7u) return "invalid io datatype(s)";
        if (io->scale != 1. && io->scale != 0.) return "hipsoxr: io scale other than 1 is not implemented";
        *dtype = io->itype;
        if (io->flags & SOXR_NO_DITHER) *flags |= HIPSOXR_NO_DITHER;
    }
    if (soxr_error_t e = recipe_of(q, recipe)) return e;
    if (q && (q->flags & SOXR_VR)) *flags |= HIPSOXR_VR;
    return nullptr;
}

} // namespace

extern "C" {

// reference: src/csoxr_version.cpp:6-8
char const *soxr_version(void) { return "libsoxr-compatible hipsoxr-" HIPSOXR_VERSION_STRING " (gfx950)"; }

// reference: src/soxr_ext.cpp:74, :228, :303, :376
soxr_quality_spec_t soxr_quality_spec(unsigned long recipe, unsigned long flags)
{
    soxr_quality_spec_t s;
    std::memset(&s, 0, sizeof s);
    hipsoxr::QualitySpec q;
    if (hipsoxr::quality_spec(recipe & 0xf, &q)) { // invalid recipe: soxr_create will refuse it
        s.precision = -1.;
        s.phase_response = 50.;
        return s;
    }
    s.precision = q.bits;
    s.passband_end = q.passband_end;
    s.stopband_begin = q.stopband_begin;
    const unsigned long phase = recipe & 0x30;
    s.phase_response = phase == SOXR_LINEAR_PHASE ? 50. : phase == SOXR_INTERMEDIATE_PHASE ? 25. : 0.;
    s.flags = flags;
    return s;
}

// reference: src/soxr_ext.cpp:73, :227, :302, :375
soxr_io_spec_t soxr_io_spec(soxr_datatype_t itype, soxr_datatype_t otype)
{
    soxr_io_spec_t s;
    std::memset(&s, 0, sizeof s);
    s.itype = itype;
    s.otype = otype;
    s.scale = 1.;
    return s;
}

soxr_runtime_spec_t soxr_runtime_spec(unsigned num_threads)
{
    soxr_runtime_spec_t s;
    std::memset(&s, 0, sizeof s);
    s.log2_min_dft_size = 10;
    s.log2_large_dft_size = 17;
    s.coef_size_kbytes = 400;
    s.num_threads = num_threads;
    return s;
}

// reference: src/soxr_ext.cpp:76-78, :230-232, :305-307
soxr_t soxr_create(double input_rate, double output_rate, unsigned num_channels, soxr_error_t *error,
                   soxr_io_spec_t const *io_spec, soxr_quality_spec_t const *quality_spec,
                   soxr_runtime_spec_t const *)
{
    int dtype;
    unsigned long recipe, flags;
    soxr_error_t e = translate(io_spec, quality_spec, &dtype, &recipe, &flags);
    soxr *p = nullptr;
    if (!e) {
        p = new (std::nothrow) soxr();
        if (!p) e = "malloc failed";
    }
    if (!e) {
        p->channels = num_channels;
        p->dtype = dtype;
        p->io_ratio = input_rate / output_rate;
        e = hipsoxr_stream_create(input_rate, output_rate, num_channels, (hipsoxr_datatype_t)dtype, recipe, flags,
                                  &p->h);
        if (e) { delete p; p = nullptr; }
    }
    if (error) *error = e;
    return p;
}

// reference: src/soxr_ext.cpp:118-121, :163-166, :245-248, :253-256, :328-331, :339-342
soxr_error_t soxr_process(soxr_t p, soxr_in_t in, size_t ilen, size_t *idone, soxr_out_t out, size_t olen,
                          size_t *odone)
{
    if (!p) return "null pointer";
    // libsoxr's end-of-input convention for callers that pass a length with the flush: ilen = ~ilen
    // (a "negative" size_t) means "these are the last ilen frames"
    bool last = false;
    if ((ptrdiff_t)ilen < 0) { ilen = ~ilen; last = true; }
    size_t od = 0;
    soxr_error_t e = hipsoxr_stream_process(p->h, in, ilen, out, olen, &od);
    if (!e && last && in) { // input consumed: now flush into what is left of the output buffer
        size_t od2 = 0;
        Cursor cur;
        e = hipsoxr_stream_process(p->h, nullptr, 0, cur.at(p, out, od), olen - od, &od2);
        od += od2;
    }
    if (idone) *idone = e ? 0 : ilen; // everything handed over is consumed (the stream keeps it on the device)
    if (odone) *odone = od;
    // (the clip counter lives on the device: it is fetched when soxr_num_clips() asks for it, not after
    // every chunk — that would be a device-to-host copy and a stream synchronise per call)
    p->err = e;
    return e;
}

soxr_error_t soxr_set_input_fn(soxr_t p, soxr_input_fn_t fn, void *state, size_t max_ilen)
{
    if (!p) return "null pointer";
    p->fn = fn;
    p->fn_state = state;
    p->max_ilen = max_ilen ? max_ilen : (size_t)-1;
    return nullptr;
}

size_t soxr_output(soxr_t p, soxr_out_t out, size_t olen)
{
    if (!p || !p->fn) return 0;
    size_t total = 0;
    Cursor cur;
    while (total < olen && !p->err) {
        size_t od = 0;
        void *o = cur.at(p, out, total);
        // first whatever is already producible (in != NULL with ilen == 0 drains; NULL flushes)
        p->err = hipsoxr_stream_process(p->h, p->input_ended ? nullptr : (const void *)o, 0, o, olen - total, &od);
        if (p->err) break;
        total += od;
        if (od || total == olen) continue;
        if (p->input_ended) break; // flushed dry
        soxr_in_t in = nullptr;
        // ask for what the missing output needs, as libsoxr does (never an unbounded request: callbacks
        // that fill a fixed buffer rely on it), capped by max_ilen
        size_t want = (size_t)((double)(olen - total) * p->io_ratio) + 2;
        if (want > p->max_ilen) want = p->max_ilen;
        const size_t ilen = p->fn(p->fn_state, &in, want);
        if (!in) { p->err = "input function reported failure"; break; }
        if (ilen == 0) { p->input_ended = true; continue; }
        p->err = hipsoxr_stream_process(p->h, in, ilen, o, olen - total, &od);
        total += od;
    }
    return total;
}

soxr_error_t soxr_error(soxr_t p) { return p ? p->err : "null pointer"; }

// reference: src/soxr_ext.cpp:190
size_t *soxr_num_clips(soxr_t p)
{
    static size_t zero = 0;
    if (!p) return &zero;
    p->clips = hipsoxr_stream_num_clips(p->h);
    return &p->clips;
}

// reference: src/soxr_ext.cpp:157, :191
double soxr_delay(soxr_t p) { return p ? hipsoxr_stream_delay(p->h) : 0.; }

// reference: src/soxr_ext.cpp:192
char const *soxr_engine(soxr_t p) { return p ? hipsoxr_stream_engine(p->h) : ""; }

// reference: src/soxr_ext.cpp:195
soxr_error_t soxr_clear(soxr_t p)
{
    if (!p) return "null pointer";
    p->clips = 0;
    p->err = nullptr;
    p->input_ended = false;
    return hipsoxr_stream_clear(p->h);
}

// reference: src/soxr_ext.cpp:86, :260, :346
void soxr_delete(soxr_t p)
{
    if (!p) return;
    hipsoxr_stream_delete(p->h);
    delete p;
}

// reference: src/soxr_ext.cpp:201
soxr_error_t soxr_set_io_ratio(soxr_t p, double io_ratio, size_t slew_len)
{
    if (!p) return "null pointer";
    if (io_ratio > p->io_ratio) p->io_ratio = io_ratio; // (pull-mode requests are sized for the largest ratio seen)
    return hipsoxr_stream_set_io_ratio(p->h, io_ratio, slew_len);
}

// reference: src/soxr_ext.cpp:385-389
soxr_error_t soxr_oneshot(double input_rate, double output_rate, unsigned num_channels, soxr_in_t in, size_t ilen,
                          size_t *idone, soxr_out_t out, size_t olen, size_t *odone, soxr_io_spec_t const *io_spec,
                          soxr_quality_spec_t const *quality_spec, soxr_runtime_spec_t const *)
{
    int dtype;
    unsigned long recipe, flags;
    if (soxr_error_t e = translate(io_spec, quality_spec, &dtype, &recipe, &flags)) return e;
    size_t od = 0;
    soxr_error_t e = hipsoxr_oneshot(input_rate, output_rate, num_channels, in, ilen, out, olen, &od,
                                     (hipsoxr_datatype_t)dtype, recipe, flags);
    if (idone) *idone = e ? 0 : ilen;
    if (odone) *odone = od;
    return e;
}

} // extern "C"
