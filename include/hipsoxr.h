/*
 * hipsoxr.h — native C ABI of the MI355X (gfx950) sample-rate converter.
 *
 * This is the drop-in boundary for the ONE hot path this repository accelerates:
 * libsoxr's constant-rate `soxr_process` underneath python-soxr's
 * `soxr.resample` / `ResampleStream.resample_chunk`.
 *
 * Every entry point names the reference interface it replaces (paths are relative to the
 * reference checkout, dofuuz/python-soxr):
 *
 *   hipsoxr_stream_create   <- soxr_create        call sites src/soxr_ext.cpp:76-78, :230-232, :305-307
 *   hipsoxr_stream_process  <- soxr_process       call sites src/soxr_ext.cpp:118-121, :163-166, :245-248,
 *                                                            :253-256, :328-331, :339-342
 *   hipsoxr_oneshot         <- soxr_oneshot       call site  src/soxr_ext.cpp:385-389
 *   hipsoxr_stream_delete   <- soxr_delete        src/soxr_ext.cpp:86, :260, :346
 *   hipsoxr_stream_clear    <- soxr_clear         src/soxr_ext.cpp:195
 *   hipsoxr_stream_delay    <- soxr_delay         src/soxr_ext.cpp:157, :191
 *   hipsoxr_stream_num_clips<- soxr_num_clips     src/soxr_ext.cpp:190
 *   hipsoxr_stream_engine   <- soxr_engine        src/soxr_ext.cpp:192
 *   hipsoxr_stream_set_io_ratio <- soxr_set_io_ratio src/soxr_ext.cpp:201
 *   hipsoxr_version         <- soxr_version       src/csoxr_version.cpp:6-8
 *   datatype / recipe constants <- soxr_datatype_t, SOXR_QQ..SOXR_VHQ  src/soxr_ext.cpp:32-46, :447-451
 *
 * Additions that have no counterpart in the reference (it has no GPU, no batch axis):
 *   hipsoxr_plan_*          the shared, immutable filter bank (what soxr_create designs per handle),
 *                           so that many streams / ranks share one bank (and RCCL can broadcast it);
 *   hipsoxr_run_device      device-pointer, batched, stateless launch of the hot path — the
 *                           entry the benchmark times (host-pointer soxr_process is PCIe-bound).
 *
 * Conventions (same as libsoxr): errors are static `const char *` strings, NULL == success;
 * handles are owned by the caller; input buffers are borrowed for the duration of the call and
 * fully consumed; output buffers are caller-allocated.  Distinct handles may be used concurrently
 * from different threads; one handle must not be.
 *
 * Plain C, no torch / HIP types in any signature (streams are passed as `void *` = hipStream_t).
 */
#ifndef HIPSOXR_H
#define HIPSOXR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HIPSOXR_API __attribute__((visibility("default")))

typedef const char *hipsoxr_error_t; /* NULL == success (reference: src/soxr_ext.cpp:80-82) */

/* Same numbering as libsoxr's soxr_datatype_t (reference: src/soxr_ext.cpp:35-46). */
typedef enum {
    HIPSOXR_FLOAT32_I = 0, /* interleaved [frame][channel] */
    HIPSOXR_FLOAT64_I = 1,
    HIPSOXR_INT32_I = 2,
    HIPSOXR_INT16_I = 3,
    HIPSOXR_FLOAT32_S = 4, /* split: one contiguous buffer per channel */
    HIPSOXR_FLOAT64_S = 5,
    HIPSOXR_INT32_S = 6,
    HIPSOXR_INT16_S = 7
} hipsoxr_datatype_t;

/* Quality recipes — libsoxr values (reference: src/soxr_ext.cpp:447-451). */
#define HIPSOXR_QQ 0UL
#define HIPSOXR_LQ 1UL
#define HIPSOXR_MQ 2UL
#define HIPSOXR_HQ 4UL
#define HIPSOXR_VHQ 6UL

/* Stream flags. */
#define HIPSOXR_VR 32UL        /* quality-spec flag: variable rate (reference: src/soxr_ext.cpp:74) */
#define HIPSOXR_NO_DITHER 8UL  /* io-spec flag: disable int16 TPDF dither */
#define HIPSOXR_DEFER 64UL     /* (extension) deferred output: a process call enqueues its work and returns the
                                  PREVIOUS call's frames — no GPU round trip inside the call.  Same concatenated
                                  output; frames surface one call later (the reference's contract allows any
                                  per-call count, README.md:77-78).  Constant-rate interleaved streams. */
#define HIPSOXR_RESIDENT 128UL /* (extension) synchronous small chunks (results <= 2048 frames, ring in pinned host
                                  memory) are served by a kernel that STAYS on the GPU between calls and is fed
                                  through a mailbox in pinned memory: no HIP call per chunk, ~3x lower latency
                                  per call.  Same output, bit for bit.  The kernel leaves by itself after
                                  HIPSOXR_RESIDENT_IDLE_US (default 1000) without a call; until then device-wide
                                  synchronisations elsewhere in the process wait for it.  Interleaved streams,
                                  constant or variable rate (a variable-rate message carries its own Q64.64
                                  clock), without HIPSOXR_DEFER.  Also: environment HIPSOXR_RESIDENT. */
#define HIPSOXR_AUTO_RESIDENT 256UL /* (extension, opt-in; also environment HIPSOXR_AUTO_RESIDENT) the stream turns the
                                  resident path on BY ITSELF once it has been fed 16 small chunks back to back (each
                                  within 500 us of the one before) and off again at the first call that breaks the run;
                                  such instances may hold at most an eighth of the chip.  Not a default: while a
                                  resident kernel spins, hipDeviceSynchronize / hipFree anywhere in the process wait
                                  until it leaves (up to the idle time, longer if another thread keeps feeding it). */

/* Element types used by device jobs (layout is given by strides, not by the type). */
typedef enum { HIPSOXR_F32 = 0, HIPSOXR_F64 = 1, HIPSOXR_I32 = 2, HIPSOXR_I16 = 3 } hipsoxr_elem_t;

/* Kernel selector for hipsoxr_run_device. */
typedef enum {
    HIPSOXR_KERNEL_AUTO = 0,   /* fastest admissible engine, including the FFT engine for large float32 jobs */
    HIPSOXR_KERNEL_GATHER = 1, /* one lane per output sample, operands gathered through L1/L2 */
    HIPSOXR_KERNEL_TILE = 2,   /* period-tiled, best variant for the engine (MFMA f32, else VALU) */
    HIPSOXR_KERNEL_TILE_VALU = 3, /* period-tiled: input slab in LDS, coefficients on the scalar path */
    HIPSOXR_KERNEL_TILE_MFMA = 4, /* period-tiled on the f32-input matrix pipe (f32 engine only) */
    HIPSOXR_KERNEL_FFT = 5,    /* frequency-domain overlap-save engine (whole-signal float32 jobs; 1e-6-class,
                                  not bit-identical to the canonical order) */
    HIPSOXR_KERNEL_EXACT = 6,  /* AUTO restricted to the canonical-order kernels (bit-exact invariances) */
    HIPSOXR_KERNEL_WAVE_DOT = 7, /* reference point only: one wavefront per output sample + shuffle reduction
                                  (the shape BASELINE.json's north star describes); 64-way tree order, so
                                  1e-6-class like FFT, never chosen automatically */
    HIPSOXR_KERNEL_FFT_F64 = 8  /* the frequency-domain engine computing in float64 whatever the I/O type: float32
                                  jobs at the width libsoxr's VHQ recipe itself computes in (its float64 engine;
                                  reference src/soxr_ext.cpp:74,228 pass the recipe through unchanged) — results
                                  differ from the float64 direct form by the float32 OUTPUT rounding only
                                  (~3e-8 relative RMS).  Unit-stride columns of the tabled ratios. */
} hipsoxr_kernel_t;

typedef struct hipsoxr_plan hipsoxr_plan_t;     /* immutable: ratio + polyphase bank (host + device) */
typedef struct hipsoxr_stream hipsoxr_stream_t; /* stateful converter: the `soxr_t` counterpart */

/* ---- library ---------------------------------------------------------------------------- */
#define HIPSOXR_VERSION_STRING "0.5.0" /* one number for hipsoxr_version() and the libsoxr-named soxr_version() */
HIPSOXR_API const char *hipsoxr_version(void);
HIPSOXR_API int hipsoxr_device_count(void); /* 0 when no HIP device is visible (never throws) */

/* ---- plan: what soxr_create designs (filter bank for in_rate -> out_rate at a recipe) ---- */
typedef struct {
    double in_rate, out_rate;
    unsigned long recipe;
    int64_t L, M;          /* out/in = L/M in lowest terms: L phases, phase step M              */
    int32_t taps;          /* taps per phase (even, multiple of 8)                               */
    int32_t interpolated;  /* 0: exact rational bank [L][taps]; P > 0: interpolated-phase plan
                              (ratios whose exact bank would exceed 2^22 entries): the bank is the
                              cubic coefficient table [P][taps][4]                               */
    double precision_bits; /* 0 (QQ), 16, 20, 28                                                 */
    double passband_end;   /* fraction of the lower rate's Nyquist                                */
    double stopband_begin;
    double att_db;         /* design stop-band attenuation                                        */
    double kaiser_beta;
    uint64_t bank_elems;   /* L * taps, or P * taps * 4                                           */
} hipsoxr_plan_info_t;

HIPSOXR_API hipsoxr_error_t hipsoxr_plan_create(double in_rate, double out_rate,
                                                unsigned long recipe, hipsoxr_plan_t **out);
/* The plan a variable-rate stream created with the same arguments uses (always an interpolated-phase
 * table, designed for the largest io ratio in_rate/out_rate; reference: src/soxr_ext.cpp:74).  Host only. */
HIPSOXR_API hipsoxr_error_t hipsoxr_plan_create_vr(double in_rate, double out_rate,
                                                   unsigned long recipe, hipsoxr_plan_t **out);
HIPSOXR_API void hipsoxr_plan_delete(hipsoxr_plan_t *);
HIPSOXR_API hipsoxr_error_t hipsoxr_plan_info(const hipsoxr_plan_t *, hipsoxr_plan_info_t *info);
/* Copy the float64 bank (phase-major [L][taps], or [P][taps][4]) into dst (n = bank_elems doubles). */
HIPSOXR_API hipsoxr_error_t hipsoxr_plan_get_bank(const hipsoxr_plan_t *, double *dst, size_t n);
/* Replace the bank (e.g. with the one RCCL-broadcast from rank 0); device tables are rebuilt. */
HIPSOXR_API hipsoxr_error_t hipsoxr_plan_set_bank(hipsoxr_plan_t *, const double *src, size_t n);
/* Multi-GPU: broadcast the bank from rank `root` to every rank of an RCCL communicator (`nccl_comm` is an
 * ncclComm_t; one process per GPU, each calls this with its own plan created from the same arguments;
 * `my_rank` is the caller's rank in the communicator).  Ranks other than root install what they receive
 * (device tables rebuild on next use).  This is the only collective of the path: clips shard across
 * ranks with no data-path exchange (BASELINE.json north_star: "RCCL broadcast of the shared filter bank
 * over xGMI").  RCCL is looked up in the running process (e.g. PyTorch's copy), else dlopen'ed. */
HIPSOXR_API hipsoxr_error_t hipsoxr_plan_broadcast(hipsoxr_plan_t *, void *nccl_comm, int root, int my_rank,
                                                   void *hip_stream);
/* Total output frames for `in_len` input frames: floor(in_len*L/M + 1/2). */
HIPSOXR_API uint64_t hipsoxr_plan_out_len(const hipsoxr_plan_t *, uint64_t in_len);

/* ---- device job: the hot path on device-resident buffers (batched, stateless) ------------- */
typedef struct {
    const void *in;  /* device pointer */
    void *out;       /* device pointer */
    int32_t elem;    /* hipsoxr_elem_t of in and out */
    int32_t kernel;  /* hipsoxr_kernel_t */
    uint32_t n_clips, n_channels;
    /* strides in ELEMENTS: address(clip, frame, ch) = base + clip*cs + frame*fs + ch*chs */
    int64_t in_clip_stride, in_frame_stride, in_chan_stride;
    int64_t out_clip_stride, out_frame_stride, out_chan_stride;
    int64_t in_abs0;   /* absolute stream index of in[frame 0] (0 for a whole signal)            */
    int64_t in_frames; /* frames present at `in`; the signal is zero outside [in_abs0, +frames)  */
    int64_t out_k0;    /* absolute index of the first output frame to produce                    */
    int64_t out_frames;/* output frames to produce (per clip)                                    */
    uint64_t *clip_counter; /* device counter of saturated integer outputs, or NULL            */
    uint32_t dither;        /* 1: TPDF dither on int16 output                                    */
    uint32_t dither_seed;
    /* Ragged batches — independent clips of unequal length in ONE job (what a corpus is: the reference takes any
     * length per call, src/soxr/__init__.py:182-231).  NULL: every clip has in_frames / out_frames and starts at
     * clip * clip_stride.  Else n_clips rows of four int64 { in_offset, in_frames, out_offset, out_frames }, offsets in
     * ELEMENTS from `in` / `out` (clip c, frame f, channel ch is at in + in_offset[c] + f*in_frame_stride +
     * ch*in_chan_stride; the clip strides are ignored), out_frames[c] <= hipsoxr_plan_out_len(in_frames[c]).
     * clip_table is in HOST memory and is what the library validates (counts against the job and the plan; the
     * caller guarantees that offset + extent stays inside its buffers — the library does not know their sizes).
     * clip_table_dev is optional: NULL = the library uploads the host table itself, in stream order, on every call
     * (~10 us); a device copy supplied by the caller (a prepared job launched many times) is used as it is and must
     * equal the host table — it is not checked.  in_frames / out_frames of the job are then the LARGEST per-clip
     * values.  Whole signals only (in_abs0 == 0, out_k0 == 0).  Unit-stride float columns go out as one launch of
     * the frequency-domain engine (AUTO: from 2^13 output samples in all, as for equal-length jobs); every other
     * case is served clip by clip, same results. */
    const int64_t *clip_table;
    const int64_t *clip_table_dev;
} hipsoxr_job_t;
/* ZERO-INITIALISE the struct (memset / = {0}) before filling it: fields are only ever APPENDED, a zero field always
 * means "feature not used", and hipsoxr_version() changes when one is added (0.1: up to dither_seed; 0.3: clip_table,
 * clip_table_dev).  A client compiled against an older header must not be run against a newer struct-consuming
 * library without recompiling — check the version string at load time as soxr_amd/_native.py does. */

/* Enqueue the job on `hip_stream` (a hipStream_t; NULL = default stream). Asynchronous. */
HIPSOXR_API hipsoxr_error_t hipsoxr_run_device(hipsoxr_plan_t *, const hipsoxr_job_t *job,
                                               void *hip_stream);

/* ---- stream: the soxr_t counterpart (host pointers, state carried across calls) ----------- */
HIPSOXR_API hipsoxr_error_t hipsoxr_stream_create(double in_rate, double out_rate,
                                                  unsigned num_channels,
                                                  hipsoxr_datatype_t io_type, /* itype == otype */
                                                  unsigned long recipe, unsigned long flags,
                                                  hipsoxr_stream_t **out);
/* Share an existing plan (no filter design); the plan must outlive the stream. */
HIPSOXR_API hipsoxr_error_t hipsoxr_stream_create_with_plan(hipsoxr_plan_t *, unsigned num_channels,
                                                            hipsoxr_datatype_t io_type,
                                                            unsigned long flags,
                                                            hipsoxr_stream_t **out);
/* soxr_process semantics: `in` = T const* (interleaved) or T const* const* (split); in == NULL
 * marks end of input (flush; call until *odone == 0); ilen == 0 with in != NULL drains only.
 * All of `ilen` is always consumed.  At most `olen` frames are written; *odone = frames written. */
HIPSOXR_API hipsoxr_error_t hipsoxr_stream_process(hipsoxr_stream_t *, const void *in, size_t ilen,
                                                   void *out, size_t olen, size_t *odone);
/* The same call on DEVICE buffers (round 4; version 0.4.1; round 5: small constant-rate calls are ONE dispatch — the kernel appends the chunk): `in` / `out` are device pointers (interleaved streams only),
 * the chunk is appended to the stream's ring by a device-to-device copy and every output the input so far determines is
 * written straight into `out`, all enqueued on `hip_stream` (a hipStream_t; NULL = default stream) — asynchronous, no
 * copy over PCIe, no host synchronisation; *odone is known at once (it is a function of the counts alone).  The caller
 * keeps `in` valid until the copy has run and orders its own use of `out` on `hip_stream`.  Same contract otherwise
 * (in == NULL: end of input; variable-rate streams and hipsoxr_stream_set_io_ratio included); frames and call
 * boundaries are those of hipsoxr_stream_process on the same input.  What CSoxr::process (src/soxr_ext.cpp:129-187)
 * would be for a caller whose audio already lives in HBM.  Not for streams created with HIPSOXR_DEFER / HIPSOXR_RESIDENT
 * flags or the split layout. */
HIPSOXR_API hipsoxr_error_t hipsoxr_stream_process_device(hipsoxr_stream_t *, const void *in, size_t ilen,
                                                          void *out, size_t olen, size_t *odone, void *hip_stream);
/* Many INDEPENDENT streams in one call (round 5; version 0.5.0): handles[i] gets chunk ins[i] / ilens[i] and writes up to
 * olens[i] frames to outs[i], exactly as n calls of hipsoxr_stream_process_device in index order would — same frames,
 * same counters, bit for bit — but streams that share a plan (rates, quality), element type, channel count and device are
 * served by ONE kernel launch on `hip_stream`, whatever their phases and pending counts (a service with N live callers
 * at 10 ms chunks: one dispatch per tick instead of 2 N).  Streams the shared launch does not take (variable rate, end
 * of input, 4096 or more outputs due, mixed plans) are processed one by one inside the same call.  What N threads around
 * CSoxr::process (reference src/soxr_ext.cpp:129-187, tests/bench.py:71-88) would be for callers whose audio lives in HBM. */
HIPSOXR_API hipsoxr_error_t hipsoxr_streams_process_device(hipsoxr_stream_t *const *handles, size_t n, const void *const *ins,
                                                           const size_t *ilens, void *const *outs, const size_t *olens,
                                                           size_t *odones, void *hip_stream);
HIPSOXR_API void hipsoxr_stream_delete(hipsoxr_stream_t *);
HIPSOXR_API hipsoxr_error_t hipsoxr_stream_clear(hipsoxr_stream_t *);
HIPSOXR_API double hipsoxr_stream_delay(hipsoxr_stream_t *);
HIPSOXR_API size_t hipsoxr_stream_num_clips(hipsoxr_stream_t *);
HIPSOXR_API const char *hipsoxr_stream_engine(hipsoxr_stream_t *);
HIPSOXR_API hipsoxr_error_t hipsoxr_stream_set_io_ratio(hipsoxr_stream_t *, double io_ratio,
                                                        size_t slew_len);
HIPSOXR_API hipsoxr_plan_t *hipsoxr_stream_plan(hipsoxr_stream_t *);
/* int16 output is TPDF-dithered with a counter-based hash of (seed, channel, absolute output index):
 * deterministic and chunk-invariant.  libsoxr seeds its dither randomly per handle (the reference xfails
 * exact equality of int16 results for that reason, tests/test_resample.py:208-211); here the seed is 0
 * unless set, so equal inputs give equal outputs — and two streams with the same seed dither alike.
 * Give concurrent streams distinct seeds to decorrelate them. */
HIPSOXR_API hipsoxr_error_t hipsoxr_stream_set_dither_seed(hipsoxr_stream_t *, uint32_t seed);

/* ---- one-shot (create + process all + flush + delete), host pointers ---------------------- */
HIPSOXR_API hipsoxr_error_t hipsoxr_oneshot(double in_rate, double out_rate, unsigned num_channels,
                                            const void *in, size_t ilen, void *out, size_t olen,
                                            size_t *odone, hipsoxr_datatype_t io_type,
                                            unsigned long recipe, unsigned long flags);

#ifdef __cplusplus
}
#endif
#endif /* HIPSOXR_H */
