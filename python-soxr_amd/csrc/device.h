// device.h — interface between the host engine (engine.cpp) and the HIP side (kernels.hip).
#pragma once
#include "../../include/hipsoxr.h"
#include "plan.h"

namespace hipsoxr {

// Build (once per precision) the device tables of a plan.  prec: 0 = f32 engine, 1 = f64 engine.
const char *device_bank_ensure(Plan *p, int prec);
void device_bank_release(Plan *p);

// Engine precision used for an element type: f32 for f32/i16, f64 for f64/i32.
inline int engine_prec(int elem) { return (elem == HIPSOXR_F32 || elem == HIPSOXR_I16) ? 0 : 1; }
inline size_t elem_size(int elem)
{
    return elem == HIPSOXR_F32 ? 4 : elem == HIPSOXR_F64 ? 8 : elem == HIPSOXR_I32 ? 4 : 2;
}

// Variable-rate launches: input position of local output i is the Q64.64 fixed-point quadratic
//   t(i) = T0 + i*S0 + D*i(i-1)/2     (128-bit two's-complement words, hi:lo)
struct VrPos {
    uint64_t t_hi, t_lo, s_hi, s_lo, d_hi, d_lo;
};

// Enqueue one job (validated by the caller) on `stream`.  vr != nullptr: positions come from *vr
// instead of the plan's rational ratio (interpolated-phase plans only).
// Resident form of the small-launch kernel (kernels.hip, k_chain_resident): the mailbox in pinned host memory
// and what a launch of an instance needs.
static constexpr unsigned kResidentMaxWgs = 64;
// Watchdog: whatever HIPSOXR_RESIDENT_IDLE_US says, an instance that has heard nothing for this long leaves — a kernel that
// spins holds every device-wide synchronisation of the process (torch.cuda.synchronize() included) until it does.
static constexpr int kResidentWatchdogUs = 20000;
struct ResidentBox {
    uint64_t w[16];                 // lines 0-1: host -> device (tagged words, see k_chain_resident): 5 of them for a constant-rate
                                    // message, 14 with a variable-rate clock (three 128-bit numbers in 48-bit pieces); w[15] = leave
    uint32_t exited, pad[15];       // line 1: device -> host: the instance that has left
    uint32_t done[kResidentMaxWgs]; // lines 2..5: device -> host: last message finished, per workgroup
};
struct ResidentCtl { unsigned long long dec; unsigned arrived, pad; };
struct ResidentLaunch {
    ResidentBox *box;      // pinned, device-mapped
    const uint64_t *words; // where the instance reads the host -> device words: box->w, or 64 bytes of device memory
                           // the CPU stores into directly (large-BAR systems: a shorter trip, see engine.cpp)
    ResidentCtl *ctl;      // device memory, zeroed, not used by an earlier instance
    uint32_t base_seq;     // messages taken by earlier instances
    uint32_t epoch;        // number of this instance (never 0)
    int64_t idle_us;       // the instance leaves after this long without a message
    int64_t used_mcu = 0;  // in: milli-CUs already held by this process's resident instances
    int budget_shift = 1;  // in: resident instances may hold (CUs >> budget_shift) of the chip together
    uint32_t n_wgs = 0;    // out: workgroups of the instance
    int64_t max_out = 0;   // out: outputs per column one message may ask for
    uint32_t cost_mcu = 0; // out: CU capacity the instance occupies, n_wgs * 1024 / (workgroups that fit one CU)
    bool over_budget = false; // out: refused because used_mcu + cost_mcu would exceed the budget
};
// Completion words for a small launch: if the job turns out to be ONE launch of the small-launch kernel (k_chain) of at
// most `cap` workgroups, each workgroup stores `seq` into words[w] (pinned host memory) after its results are in host
// memory, and n_wgs says how many words to wait for; otherwise n_wgs = 0 and the caller waits for the stream.
// Polling these costs nothing; an event costs a record call, the command processor's signal after the kernel has
// drained, and the query calls (~4 us per streaming call).
struct ChainDone { uint32_t *words; uint32_t cap, seq; uint32_t n_wgs = 0; };
// res: job.out_frames = the largest message the instance must serve; launches nothing but the resident kernel
const char *launch_job(Plan *p, const hipsoxr_job_t &job, void *stream, const VrPos *vr = nullptr, ResidentLaunch *res = nullptr,
                       ChainDone *cd = nullptr);
// hand message `seq` to the instance: outputs [out_k0, out_k0 + out_frames) from ring frames [in_abs0, in_abs0 + in_frames)
bool resident_post(const Plan &p, volatile uint64_t *words, uint32_t seq, int64_t in_abs0, int64_t in_frames, int64_t out_k0, int64_t out_frames,
                   const VrPos *vr = nullptr);
void resident_leave(volatile uint64_t *words, uint32_t epoch);

int device_count();

// Dynamic LDS above 64 KB needs hipFuncAttributeMaxDynamicSharedMemorySize raised on the function: a driver call, made
// once per (function, device) and high-water mark of the sizes asked for — not inside every launch.
const char *ensure_dyn_lds(const void *fn, size_t bytes);

// Environment switches, read ONCE per process.  The product library reads four names:
//   HIPSOXR_NO_FFT, HIPSOXR_RESIDENT, HIPSOXR_AUTO_RESIDENT, HIPSOXR_RESIDENT_IDLE_US.
// Everything else is an A/B or timing-experiment switch behind the numbers in DESIGN.md / profiles/ and is compiled in
// only with -DHIPSOXR_DEBUG_SWITCHES (build.sh makes that build beside the product: _variants/dbg/libhipsoxr.so, loaded
// through HIPSOXR_LIBRARY by tests/test_gpu_switches.py, tests/test_gpu_launch_forms.py and tools/*.sh).  None of them
// changes results beyond what the selected engine implies (dbg_flags excepted: timing only).
struct Switches {
    // ---- product ----
    bool no_fft = false;          // HIPSOXR_NO_FFT           AUTO never picks the frequency-domain engine (every job bit-exact)
    bool resident = false;        // HIPSOXR_RESIDENT         small-chunk synchronous streams use the resident kernel (as the HIPSOXR_RESIDENT flag)
    bool auto_resident = false;   // HIPSOXR_AUTO_RESIDENT    streams turn the resident path on by themselves after 16 small back-to-back calls
    int resident_idle_us = 1000;  // HIPSOXR_RESIDENT_IDLE_US an idle resident kernel leaves after this long (at most kResidentWatchdogUs = 20 ms)
    // ---- debug builds only (-DHIPSOXR_DEBUG_SWITCHES) ----
    bool fft_no_pair = false;     // HIPSOXR_FFT_NO_PAIR      one block per workgroup instead of the paired kernels
    bool fft_no_chpair = false;   // HIPSOXR_FFT_NO_CHPAIR    pair blocks even for interleaved even-channel data
    bool fft_no_xcd_map = false;  // HIPSOXR_FFT_NO_XCD_MAP   plain (block, column) workgroup ids for interleaved data
    bool fft_large_only = false;  // HIPSOXR_FFT_LARGE_ONLY   never the small-block variant
    bool fft_small_only = false;  // HIPSOXR_FFT_SMALL_ONLY   always the small-block variant
    bool fft_no_tiny = false;     // HIPSOXR_FFT_NO_TINY      never the quarter-size blocks
    bool fft_no_wave = false;     // HIPSOXR_FFT_NO_WAVE      never the one-wave-per-pair kernel (fftwave.hip)
    int dbg_wave_min = 0;         // HIPSOXR_DEBUG_WAVE_MIN   ... from this many block pairs up (default 4096)
    int dbg_wave_slots = 0;       // HIPSOXR_DEBUG_WAVE_SLOTS ... at most this many persistent waves per launch (default: what the chip holds)
    bool no_planes = false;       // HIPSOXR_NO_PLANES        k_tile_mfma instead of k_tile_mfma_p
    bool no_mfma64 = false;       // HIPSOXR_NO_MFMA64        float64 engine (float64 / int32 I/O) on the vector ALU (k_tile) instead of v_mfma_f64
    bool no_host_ring = false;    // HIPSOXR_NO_HOST_RING     small-chunk streams keep their ring in device memory (copy per call)
    bool no_chain = false;        // HIPSOXR_NO_CHAIN         small launches on k_gather / k_interp
    bool no_done_words = false;   // HIPSOXR_NO_DONE_WORDS    streaming calls wait on an event, not on the kernel's completion words
    int direct_max = 0;           // HIPSOXR_DEBUG_DIRECT_MAX  largest result (bytes) a kernel writes straight into pinned host memory (0: engine.cpp's rule)
    bool resident_no_bar = false; // HIPSOXR_RESIDENT_NO_BAR  mailbox words and input stay in pinned host memory even on large-BAR systems
    bool no_xcd_split = false;    // HIPSOXR_NO_XCD_SPLIT     k_tile_mfma_p unit split on grid.z instead of XCD-aware ids
    bool no_tile_split = false;   // HIPSOXR_NO_TILE_SPLIT    k_tile / k_tile_mfma: never spread a slab's row tiles over several workgroups
    bool no_interp_tile = false;  // HIPSOXR_NO_INTERP_TILE   large interpolated launches on k_interp
    bool no_interp_wave = false;  // HIPSOXR_NO_INTERP_WAVE   mid-size interpolated / variable-rate launches on lane-per-output k_interp
    bool no_gather_wave = false;  // HIPSOXR_NO_GATHER_WAVE   exact-bank jobs below the tile kernels' size on k_chain / k_gather / the tile kernels as before
    int dbg_gw_taps = 0;          // HIPSOXR_DEBUG_GW_TAPS    k_gather_wave takes AUTO jobs of up to this many million output x tap products (default 16)
    bool no_two_stage = false;    // HIPSOXR_NO_TWO_STAGE     float device jobs of interpolated-phase plans stay on the exact engine
    // timing experiments on the tile kernels (results may be wrong with dbg_flags != 0)
    int dbg_flags = 0;            // HIPSOXR_DEBUG_FLAGS      1 no staging, 2 no LDS reads, 4 no coefficient loads, 8 no stores
    int dbg_nrt = 0, dbg_nw = 0;  // HIPSOXR_DEBUG_NRT / _NW  tiles / waves per workgroup
    int dbg_split = 0;            // HIPSOXR_DEBUG_SPLIT      grid.z unit split
    int dbg_chain_no = 0;         // HIPSOXR_DEBUG_CHAIN_NO   outputs per workgroup of k_chain (power of two <= 32)
    size_t dbg_lds = 0;           // HIPSOXR_DEBUG_LDS        extra dynamic LDS (occupancy experiments)
    bool dbg_slab32 = false;       // HIPSOXR_DEBUG_SLAB32     k_tile_mfma_p: 32-period slabs whatever the job size (A/B)
    bool no_halves = false;        // HIPSOXR_DEBUG_NO_HALVES  k_tile_mfma: both half-chains of a row tile on one wave for small jobs too (A/B)
    bool dbg_pad = false;          // HIPSOXR_DEBUG_PAD        k_tile_mfma: padded slab rows for odd periods too (round 2; A/B; read when a plan's tables are built)
    bool dbg_slab64 = false;       // HIPSOXR_DEBUG_SLAB64     k_tile_mfma_p: 64-period slabs for small jobs too (round-2 behaviour; A/B)
    int dbg_mfma64_pb = 0;         // HIPSOXR_DEBUG_MFMA64_PB  k_tile_mfma64_p: periods per slab, 16 or 32 (default: 16 below 1536 slabs of 32)
    bool dbg_mfma64_split = false; // HIPSOXR_DEBUG_MFMA64_SPLIT k_tile_mfma64_p: units of 16 periods even where 4 divides the tile count
    size_t dbg_mfma64_lds = 0;    // HIPSOXR_DEBUG_MFMA64_LDS LDS budget (bytes) that picks the float64 MFMA kernel's slab: 64, 32 or 16 periods
    size_t dbg_fft_lds = 0;       // HIPSOXR_DEBUG_FFT_LDS    the same for the frequency-domain kernels
    int dbg_poly_r = 0;           // HIPSOXR_DEBUG_POLY_R     k_poly: force the outputs per thread and tile
    bool no_interp_pair = false;  // HIPSOXR_NO_INTERP_PAIR   k_interp_tile: one output per lane (A/B)
    bool dbg_interp_no_twin = false;     // HIPSOXR_DEBUG_INTERP_NO_TWIN      ... float pairs on ONE copy of the staged span (8-byte reads) always (A/B, tests)
    bool dbg_interp_pair_always = false; // HIPSOXR_DEBUG_INTERP_PAIR_ALWAYS  ... two per lane wherever a pair exists, whatever the cost model says (tests)
    bool poly_no_pair = false;    // HIPSOXR_POLY_NO_PAIR     interleaved channel pairs on k_poly (one channel per pass) instead of k_poly2 (A/B)
    int dbg_tile_form = 0;        // HIPSOXR_DEBUG_TILE_FORM  k_tile_mfma_p: force launch form 1..4 (slab 64 whole / 64 split / 32 whole / 32 split)
    const char *dbg_trace = nullptr; // HIPSOXR_DEBUG_TRACE   path for per-wave s_memtime stamps (k_tile_mfma_p; k_fft_pair2 with -DFFT2_TRACE)
};
const Switches &switches();

// One independent stream's part of a many-streams launch (kernels.hip k_chain_multi; engine.cpp
// hipsoxr_streams_process_device): the small-launch kernel's job description per stream, plus the chunk this call appends
// to the stream's ring — frames the launch reads straight from the caller's buffer while its workgroups copy them into the
// ring for the calls to come (one dispatch per call instead of a copy kernel and a launch).
struct ChainItem {
    const void *ring;    // the stream's device ring: frame 0 of it is absolute input frame in_abs0
    const void *chunk;   // this call's new frames (device memory, the stream's own layout), or nullptr
    void *out;           // where this stream's outputs go
    void *clip_counter;  // device counter of saturated integer outputs (or nullptr)
    int64_t in_abs0;     // absolute index of ring frame 0
    int64_t in_frames;   // ring frames that count for this launch, INCLUDING the chunk's
    int64_t split;       // first ring-relative frame that (still) lives in `chunk` (= in_frames - chunk_frames)
    int64_t chunk_frames;
    int64_t out_k0, out_frames; // outputs [out_k0, out_k0 + out_frames)
    int64_t d0, p0;      // out_k0 * M = L * d0 + p0 (exact-bank plans)
    uint32_t dither_seed, pad;
    // Ring compaction rides on the same launch: when ring_dst != ring the workgroups copy the frames still needed —
    // ring frames [keep_from, split) — to the start of ring_dst and the chunk behind them; the NEXT launch's ring is ring_dst
    // (frame 0 of it = absolute frame in_abs0 + keep_from).  ring_dst == ring: the chunk goes to ring frame `split`.
    void *ring_dst;
    int64_t keep_from;
    int64_t reserved;    // (sizeof = 128: the table is staged 16 bytes per thread)
};
// n_items streams of one plan, element type and channel count in ONE launch of the small-launch kernel (constant rate;
// every stream < 4096 outputs).  items_dev: the same table where the device can read it (pinned, device-mapped host
// memory is fine) — not needed for one item, which travels in the kernel arguments.  *handled = false: not a job for
// this kernel, nothing was launched.
const char *launch_chain_items(Plan *p, int elem, uint32_t n_channels, bool dither, const ChainItem *items, const ChainItem *items_dev,
                               uint32_t n_items, void *stream, bool *handled);

// device-to-device copy as an ordinary kernel launch on `stream` (a stream's chunk appended to its ring: hipMemcpyAsync costs
// ~2x the API time and queues behind a barrier packet)
const char *launch_copy(void *dst, const void *src, size_t bytes, void *stream);

// frequency-domain engine (fft.hip): whole-signal float32 jobs
bool fft_job_eligible(const Plan &p, const hipsoxr_job_t &job);
const char *launch_fft(Plan *p, const hipsoxr_job_t &job, void *stream, bool *handled);
void fft_release(const Plan *p);
// two-stage form for interpolated-phase plans (twostage.hip): FFT stage at 1:2 / 2:1 + a short polyphase stage in LDS
const char *launch_two_stage(Plan *p, const hipsoxr_job_t &job, void *stream, bool *handled);

} // namespace hipsoxr
