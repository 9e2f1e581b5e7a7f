/*
 * soxr.h — libsoxr-compatible C ABI, served by the MI355X (gfx950) engine of hipsoxr.
 *
 * Purpose: python-soxr can be built against a system libsoxr
 * (reference CMakeLists.txt:29, :83-93 — `find_library(SOXR_LIBRARY NAMES soxr)`,
 * `find_path(SOXR_INCLUDE_DIR soxr.h)`; CI job .github/workflows/run-test.yml:44-59).  Pointing that
 * build at this header and at `libsoxr.so` from this repository makes the reference's own binding
 * (src/soxr_ext.cpp) run its hot path on the GPU with no source change.
 *
 * This file is written from libsoxr's documented public interface — the names, argument order and
 * structure fields that the reference's call sites pin (src/soxr_ext.cpp:32-46, :72-78, :118-121,
 * :163-166, :190-204, :227-232, :245-256, :385-389, :447-451; src/csoxr_version.cpp:6-8).  libsoxr's
 * own header is not part of the reference checkout; layouts marked (*) below follow the published
 * 0.1.3 API and are the one thing that cannot be cross-checked inside this repository.
 *
 * Coverage: every entry the reference binding links, plus the pull-style pair
 * soxr_set_input_fn / soxr_output and soxr_error.  Restrictions are reported as errors, never
 * silently ignored: itype must equal otype, io scale must be 1, phase response must be linear,
 * custom pass/stop-band edges are replaced by the recipe of the requested precision.
 */
#ifndef SOXR_COMPAT_FOR_HIPSOXR_H
#define SOXR_COMPAT_FOR_HIPSOXR_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SOXR __attribute__((visibility("default")))

typedef struct soxr *soxr_t;               /* opaque converter handle                        */
typedef char const *soxr_error_t;          /* 0 == success; static text otherwise            */
typedef void const *soxr_in_t;             /* T const* (interleaved) or T const* const* (split) */
typedef void *soxr_out_t;                  /* T* or T* const*                                */
typedef void *soxr_buf_t;
typedef void const *soxr_cbuf_t;
typedef soxr_buf_t const *soxr_bufs_t;
typedef soxr_cbuf_t const *soxr_cbufs_t;

/* same numbering as hipsoxr_datatype_t (reference: src/soxr_ext.cpp:35-46) */
typedef enum {
    SOXR_FLOAT32, SOXR_FLOAT64, SOXR_INT32, SOXR_INT16, SOXR_SPLIT = 4,
    SOXR_FLOAT32_I = SOXR_FLOAT32, SOXR_FLOAT64_I, SOXR_INT32_I, SOXR_INT16_I,
    SOXR_FLOAT32_S = SOXR_SPLIT, SOXR_FLOAT64_S, SOXR_INT32_S, SOXR_INT16_S
} soxr_datatype_t;

#define soxr_datatype_size(x) ((unsigned char const *)"\4\10\4\2")[(x) & 3]

typedef struct soxr_io_spec { /* (*) */
    soxr_datatype_t itype;   /* input  sample type                                  */
    soxr_datatype_t otype;   /* output sample type (must equal itype here)          */
    double scale;            /* linear gain applied while resampling (must be 1)    */
    void *e;                 /* reserved                                            */
    unsigned long flags;     /* SOXR_TPDF | SOXR_NO_DITHER                          */
} soxr_io_spec_t;
#define SOXR_TPDF 0u
#define SOXR_NO_DITHER 8u

typedef struct soxr_quality_spec { /* (*) */
    double precision;        /* conversion precision in bits: 0 (cubic), 16, 20, 24, 28, 32   */
    double phase_response;   /* 0 = minimum ... 50 = linear ... 100 = maximum (50 only here)  */
    double passband_end;     /* 0 dB point, fraction of Nyquist                               */
    double stopband_begin;   /* aliasing/imaging control, fraction of Nyquist                 */
    void *e;                 /* reserved                                                      */
    unsigned long flags;
} soxr_quality_spec_t;
#define SOXR_ROLLOFF_SMALL 0u
#define SOXR_ROLLOFF_MEDIUM 1u
#define SOXR_ROLLOFF_NONE 2u
#define SOXR_HI_PREC_CLOCK 8u
#define SOXR_DOUBLE_PRECISION 16u
#define SOXR_VR 32u

typedef struct soxr_runtime_spec { /* (*) accepted and ignored: there is no CPU engine to tune */
    unsigned log2_min_dft_size;
    unsigned log2_large_dft_size;
    unsigned coef_size_kbytes;
    unsigned num_threads;
    void *e;
    unsigned long flags;
} soxr_runtime_spec_t;

/* quality recipes (reference: src/soxr_ext.cpp:447-451) */
#define SOXR_QQ 0
#define SOXR_LQ 1
#define SOXR_MQ 2
#define SOXR_HQ SOXR_20_BITQ
#define SOXR_VHQ SOXR_28_BITQ
#define SOXR_16_BITQ 3
#define SOXR_20_BITQ 4
#define SOXR_24_BITQ 5
#define SOXR_28_BITQ 6
#define SOXR_32_BITQ 7
#define SOXR_LINEAR_PHASE 0x00
#define SOXR_INTERMEDIATE_PHASE 0x10
#define SOXR_MINIMUM_PHASE 0x30
#define SOXR_STEEP_FILTER 0x40

SOXR char const *soxr_version(void);

SOXR soxr_quality_spec_t soxr_quality_spec(unsigned long recipe, unsigned long flags);
SOXR soxr_io_spec_t soxr_io_spec(soxr_datatype_t itype, soxr_datatype_t otype);
SOXR soxr_runtime_spec_t soxr_runtime_spec(unsigned num_threads);

SOXR soxr_t soxr_create(double input_rate, double output_rate, unsigned num_channels, soxr_error_t *error,
                        soxr_io_spec_t const *io_spec, soxr_quality_spec_t const *quality_spec,
                        soxr_runtime_spec_t const *runtime_spec);

/* Push-style conversion.  in == NULL: end of input (keep calling until *odone == 0).
 * idone may be NULL; all of ilen is always consumed. */
SOXR soxr_error_t soxr_process(soxr_t resampler, soxr_in_t in, size_t ilen, size_t *idone, soxr_out_t out,
                               size_t olen, size_t *odone);

/* Pull-style conversion. */
typedef size_t (*soxr_input_fn_t)(void *input_fn_state, soxr_in_t *data, size_t requested_len);
SOXR soxr_error_t soxr_set_input_fn(soxr_t resampler, soxr_input_fn_t fn, void *input_fn_state, size_t max_ilen);
SOXR size_t soxr_output(soxr_t resampler, soxr_out_t data, size_t olen);

SOXR soxr_error_t soxr_error(soxr_t);
SOXR size_t *soxr_num_clips(soxr_t);
SOXR double soxr_delay(soxr_t);
SOXR char const *soxr_engine(soxr_t);
SOXR soxr_error_t soxr_clear(soxr_t);
SOXR void soxr_delete(soxr_t);
SOXR soxr_error_t soxr_set_io_ratio(soxr_t, double io_ratio, size_t slew_len);

SOXR soxr_error_t soxr_oneshot(double input_rate, double output_rate, unsigned num_channels, soxr_in_t in,
                               size_t ilen, size_t *idone, soxr_out_t out, size_t olen, size_t *odone,
                               soxr_io_spec_t const *io_spec, soxr_quality_spec_t const *quality_spec,
                               soxr_runtime_spec_t const *runtime_spec);

#undef SOXR
#ifdef __cplusplus
}
#endif
#endif
