// engine.cpp — host engine behind the C ABI (include/hipsoxr.h): plans, device jobs, and the
// stateful stream that mirrors libsoxr's soxr_t as python-soxr drives it
// (reference: CSoxr in src/soxr_ext.cpp:49-205 and the three one-shot drivers :208-402).
//
// State carried across soxr_process calls lives on the device: the not-yet-retired tail of the
// input (a linear staging buffer in the I/O dtype and layout) plus two absolute counters
// (frames received, frames emitted).  Because every output sample is a pure function of absolute
// positions, emitted output is independent of how the input was chunked.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <new>
#include <vector>

#include "device.h"

using namespace hipsoxr;

struct hipsoxr_plan {
    Plan p;
    // plan cache (streams / one-shot calls created from rates): key and use count
    bool cached = false;
    double key_in = 0, key_out = 0;
    unsigned long key_recipe = 0;
    bool key_vr = false;
    int key_device = -1;
    int users = 0;
    uint64_t last_use = 0;
};

namespace hipsoxr {
Plan::~Plan() { twostage_release(this); device_bank_release(this); fft_release(this); }
} // namespace hipsoxr

// Variable-rate state (SOXR_VR streams; reference: src/soxr_ext.cpp:74, :200-204).  Time is kept in
// Q64.64 fixed point.  The current segment starts at output k_s with input position t_s and step
// s0; during the first n_slew outputs the step grows by `delta` per output, afterwards it is s1:
//     t(k_s + n) = t_s + n*s0 + delta*n(n-1)/2                      n <= n_slew
//                = t(k_s + n_slew) + (n - n_slew)*s1                n >  n_slew
// All integer arithmetic: positions are exact, monotonic, and independent of how calls are cut.
typedef unsigned __int128 u128;
typedef __int128 i128;
struct VrState {
    bool on = false;
    double max_io = 0.;  // in_rate/out_rate at creation: the largest io ratio the filter allows
    uint64_t k_s = 0, n_slew = 0;
    i128 t_s = 0, s0 = 0, delta = 0, s1 = 0;

    i128 pos(uint64_t k) const
    {
        const u128 n = k - k_s;
        if (n <= n_slew) return t_s + (i128)n * s0 + delta * (i128)(n * (n - 1) / 2);
        const u128 N = n_slew;
        return t_s + (i128)N * s0 + delta * (i128)(N * (N - 1) / 2) + (i128)(n - N) * s1;
    }
    i128 step(uint64_t k) const
    {
        const u128 n = k - k_s;
        return n < n_slew ? s0 + (i128)n * delta : s1;
    }
};
static inline i128 q64(double x) { return (i128)(u128)std::ldexp(x, 64); } // truncating, exact scaling

struct hipsoxr_stream {
    VrState vr;
    hipsoxr_plan *plan = nullptr;
    bool own_plan = false;
    unsigned ch = 1;
    int elem = HIPSOXR_F32;
    bool split = false;       // layout of the stream's OWN buffers and jobs: one plane per channel
    // A split-layout stream whose first chunk is small runs interleaved inside (host ring, resident kernel, deferred output:
    // everything the small-chunk path has) behind an adapter at the call boundary: split_io says what the caller's pointers are
    bool split_io = false, adapt = false, adapt_decided = false;
    std::vector<char> ad_in, ad_out;
    unsigned long flags = 0;
    bool ended = false;
    uint64_t n_in_total = 0, k_done = 0;
    // device staging of pending input: frames [in_base, in_base + in_fill)
    void *d_in = nullptr, *d_in_alt = nullptr;
    size_t in_cap = 0, alt_cap = 0; // frames
    int64_t in_base = 0;
    size_t in_fill = 0;
    // Small-chunk streams keep the ring in pinned, device-mapped HOST memory: appending is a memcpy (no
    // copy call on the HIP stream at all) and the kernel reads its few hundred window samples over PCIe in
    // the one round trip it makes anyway.  A chunk above kHostRingChunk moves the ring to device memory for good.
    bool ring_on_host = false;
    void *d_out = nullptr;
    size_t out_cap = 0; // frames
    uint64_t *d_clips = nullptr;
    hipStream_t st = nullptr;
    // device chunks (hipsoxr_stream_process_device) run on the CALLER's HIP stream: the last one used, whether work of such
    // calls may still be in flight there, and whether host-pointer calls have used the stream's own since
    hipStream_t ext_st = nullptr;
    bool ext_pending = false, own_used = true; // (true: creation enqueues a memset on the stream's own HIP stream)
    // pinned bounce buffers for small chunks: a pageable hipMemcpyAsync is staged by the runtime
    // with a blocking hand-shake per call (~30 us each way); a memcpy into pinned memory + a true
    // async copy costs a few us
    void *h_in = nullptr, *h_out = nullptr;
    size_t h_in_bytes = 0, h_out_bytes = 0;
    hipEvent_t ev = nullptr; // completion of a call's last operation (see stream_wait)
    uint32_t *h_done = nullptr; // pinned: completion words of small launches (ChainDone, device.h)
    uint32_t done_seq = 0;
    // Deferred output (HIPSOXR_DEFER): a call enqueues its copy and launch and returns the PREVIOUS call's
    // result, which finished long ago — no GPU round trip inside the call (see stream_process_deferred)
    bool defer = false;
    size_t pend_n = 0, pend_off = 0; // frames produced by the launch in flight / already handed out of them
    int pend_slot = 0;
    void *h_res[2] = {nullptr, nullptr}, *h_src[2] = {nullptr, nullptr}; // pinned result / input bounce buffers, alternating
    size_t h_res_bytes[2] = {0, 0}, h_src_bytes[2] = {0, 0};
    hipEvent_t ev_res[2] = {nullptr, nullptr}, ev_src[2] = {nullptr, nullptr};
    unsigned calls = 0;
    // Resident kernel (HIPSOXR_RESIDENT): synchronous small chunks are handed to a kernel that stays on the GPU
    // between calls, through a mailbox in pinned memory — no HIP call per chunk (see resident_emit)
    bool resident = false;
    // Opt-in (flag HIPSOXR_AUTO_RESIDENT / environment HIPSOXR_AUTO_RESIDENT): an interleaved stream that is being fed small
    // chunks back to back turns the resident path on by itself after kAutoResidentRun such calls in a row (each within
    // half the kernel's idle time of the one before: a caller pacing 10 ms chunks in real time gains nothing from a
    // kernel that leaves after 1 ms, and is left alone) and OFF again at the first call that breaks the run (a gap, a
    // large chunk, an instance that idled out).  Not a default: a spinning kernel makes every device-wide
    // synchronisation in the process (hipDeviceSynchronize, hipFree) wait until it leaves — README / INTEGRATION.md.
    // Auto instances may hold an eighth of the chip together, flagged ones half of it.
    bool resident_auto_ok = false, resident_by_auto = false;
    unsigned small_run = 0;
    std::chrono::steady_clock::time_point last_small;
    struct Resident {
        ResidentBox *box = nullptr;  // pinned
        ResidentCtl *ctl = nullptr;  // device: kCtlSlots arbiter words, one per instance
        // Large-BAR systems: the CPU stores straight into device memory.  The host -> device words then live there
        // (round trip 2.4 instead of 3.9 us, tools/ubench/bar_write.hip) and so does a MIRROR of the input ring,
        // which the call extends by its chunk: the kernel reads its span from HBM instead of over PCIe.
        volatile uint64_t *words = nullptr; // box->w or 64 bytes of fine-grained device memory
        void *words_dev = nullptr;
        void *mirror = nullptr;             // fine-grained device memory, as large as the ring
        size_t mirror_bytes = 0, mirror_fill = 0; // capacity; frames of the ring [0, mirror_fill) that are in it
        const void *mirror_of = nullptr; int64_t mirror_base = -1; // the ring (pointer, first frame) it mirrors
        size_t ctl_next = 0;
        uint32_t seq = 0, epoch = 0;
        bool running = false;
        const void *in = nullptr; void *out = nullptr; // what the running instance was launched on
        int64_t max_out = 0;
        unsigned n_wgs = 0;
        uint32_t cost_mcu = 0;       // what the running instance holds of the process-wide budget (milli-CUs)
        unsigned failed = 0;         // launches refused (job not eligible): stop trying
    } res;
    int device = -1;         // the device the HIP stream and every buffer above live on
    uint32_t dither_seed = 0; // int16 TPDF dither: hash(seed, channel, absolute output index); see hipsoxr_stream_set_dither_seed
    char engine_name[32] = {0};
};

// Entry points that touch a stream's resources run with the stream's own device current, whatever
// device the calling thread has selected since (Python finalisers run at arbitrary times; a process
// may drive several GPUs): resources are never used, pooled or freed under another device's context.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int want)
    {
        if (want >= 0 && hipGetDevice(&prev) == hipSuccess && prev != want) switched = hipSetDevice(want) == hipSuccess;
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};

// Wait for everything queued on the stream.  A chunked call is a 20-30 us round trip on the GPU;
// the runtime's blocking hipStreamSynchronize adds an interrupt-and-wake-up latency of the same
// order, so the first ~100 us are spent polling an event instead.
// Resident kernels hold workgroup slots for as long as they live, and one that cannot get its slots keeps its
// host waiting: the process admits at most kResidentBudget resident workgroups at a time (a stream over the
// budget simply takes the ordinary path for that call and tries again later).
// The budget counts CU capacity, not workgroups: an instance of n workgroups whose kernel fits `occ` per CU costs
// n * 1024 / occ milli-CUs (with a large-LDS plan, one workgroup per CU, 64 workgroups are 64 CUs), and the process
// keeps at most half the chip resident, so every admitted instance's workgroups really are on the chip together.
static std::atomic<int64_t> g_resident_mcu{0};
static std::mutex g_resident_mu; // reserving budget = read it, launch, add: one instance at a time
static const unsigned kAutoResidentRun = 16; // small back-to-back synchronous calls before a stream turns resident by itself

// Retire the stream's resident kernel, if one is running: everything else that uses the HIP stream queues
// behind it (and would wait until it leaves by itself, HIPSOXR_RESIDENT_IDLE_US later).
static void resident_stop(hipsoxr_stream *s)
{
    if (!s->res.running) return;
    resident_leave(s->res.words, s->res.epoch);
    (void)hipStreamSynchronize(s->st);
    s->res.running = false;
    g_resident_mcu -= (int64_t)s->res.cost_mcu;
}

static hipError_t stream_wait(hipsoxr_stream *s)
{
    resident_stop(s);
    if (s->ev && hipEventRecord(s->ev, s->st) == hipSuccess) {
        for (int spin = 0; spin < 20000; ++spin) {
            const hipError_t q = hipEventQuery(s->ev);
            if (q == hipSuccess) return hipSuccess;
            if (q != hipErrorNotReady) break;
        }
    }
    return hipStreamSynchronize(s->st);
}
static const size_t kPinnedMax = (size_t)1 << 20; // chunks up to 1 MiB go through the bounce buffers
static const size_t kHostRingChunk = (size_t)64 << 10; // chunks up to 64 KiB: ring in pinned host memory (see hipsoxr_stream)

static const char *pinned_ensure(void **buf, size_t *cap, size_t bytes)
{
    if (*cap >= bytes) return nullptr;
    if (*buf) (void)hipHostFree(*buf);
    *buf = nullptr; *cap = 0;
    size_t want = 4096;
    while (want < bytes) want <<= 1;
    if (hipHostMalloc(buf, want, hipHostMallocDefault) != hipSuccess) { *buf = nullptr; return "hipHostMalloc failed"; }
    *cap = want;
    return nullptr;
}

#define HIP_TRY(expr)                                       \
    do {                                                    \
        hipError_t e_ = (expr);                             \
        if (e_ != hipSuccess) return hipGetErrorString(e_); \
    } while (0)

static const char *kNoDevice = "no HIP device available (hipsoxr has no CPU fallback)";

// ------------------------------------------------------------------------------------------------
// Plan cache.  soxr.resample designs a filter per call in the reference (soxr_create inside
// csoxr_divide_proc, src/soxr_ext.cpp:230); here the design, its device tables and the FFT
// geometry are kept and shared by every later stream / one-shot call with the same
// (rates, recipe, device): a repeated call pays for transfers and launches only.
// ------------------------------------------------------------------------------------------------
static std::mutex g_cache_mu;
static std::vector<hipsoxr_plan *> g_cache;
static uint64_t g_cache_clock = 0;
static const size_t kCacheMax = 32;

static hipsoxr_plan *cache_find(double in_rate, double out_rate, unsigned long recipe, bool vr, int dev)
{
    for (hipsoxr_plan *h : g_cache)
        if (h->key_in == in_rate && h->key_out == out_rate && h->key_recipe == recipe && h->key_vr == vr &&
            h->key_device == dev) {
            ++h->users; h->last_use = ++g_cache_clock;
            return h;
        }
    return nullptr;
}

static const char *plan_acquire(double in_rate, double out_rate, unsigned long recipe, bool vr, hipsoxr_plan **out)
{
    int dev = -1;
    (void)hipGetDevice(&dev);
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        if ((*out = cache_find(in_rate, out_rate, recipe, vr, dev))) return nullptr;
    }
    // The design (up to 4M bank entries with a Bessel series each) runs OUTSIDE the cache lock: other
    // threads keep creating and releasing streams meanwhile.  Two threads may design the same plan
    // concurrently; the second to finish finds the first one's entry and drops its own.
    hipsoxr_plan *h = new (std::nothrow) hipsoxr_plan();
    if (!h) return "out of memory";
    if (const char *e = plan_design(in_rate, out_rate, recipe, &h->p, vr)) { delete h; return e; }
    std::lock_guard<std::mutex> lk(g_cache_mu);
    if ((*out = cache_find(in_rate, out_rate, recipe, vr, dev))) { delete h; return nullptr; }
    h->cached = true; h->key_in = in_rate; h->key_out = out_rate; h->key_recipe = recipe; h->key_vr = vr;
    h->key_device = dev; h->users = 1; h->last_use = ++g_cache_clock;
    if (g_cache.size() >= kCacheMax) { // evict the least recently used idle plan
        size_t victim = g_cache.size();
        for (size_t i = 0; i < g_cache.size(); ++i)
            if (g_cache[i]->users == 0 && (victim == g_cache.size() || g_cache[i]->last_use < g_cache[victim]->last_use))
                victim = i;
        if (victim < g_cache.size()) { delete g_cache[victim]; g_cache.erase(g_cache.begin() + (long)victim); }
    }
    g_cache.push_back(h);
    *out = h;
    return nullptr;
}

// ------------------------------------------------------------------------------------------------
// Stream-resource pool: HIP stream + device staging buffers of finished streams, handed to the
// next stream on the same device (hipMalloc / hipFree / hipStreamCreate cost more than resampling a
// short clip).  Capacities are kept in bytes; at most kPoolMax shells, kPoolBytes in total.
// ------------------------------------------------------------------------------------------------
struct StreamShell {
    int device = -1;
    bool ring_on_host = false; // d_in / d_in_alt are pinned host memory (small-chunk streams), not device memory
    hipStream_t st = nullptr;
    void *d_in = nullptr, *d_in_alt = nullptr, *d_out = nullptr;
    size_t in_bytes = 0, alt_bytes = 0, out_bytes = 0;
    uint64_t *d_clips = nullptr;
    void *h_in = nullptr, *h_out = nullptr;
    size_t h_in_bytes = 0, h_out_bytes = 0;
    uint32_t *h_done = nullptr; uint32_t done_seq = 0; // completion words of small launches, and the last number used in them
    hipEvent_t ev = nullptr;
};
static std::mutex g_pool_mu;
static std::vector<StreamShell> g_pool;
static const size_t kPoolMax = 8, kPoolBytes = (size_t)1 << 30;

static void shell_free(StreamShell &sh)
{
    if (sh.d_in) (void)(sh.ring_on_host ? hipHostFree(sh.d_in) : hipFree(sh.d_in));
    if (sh.d_in_alt) (void)(sh.ring_on_host ? hipHostFree(sh.d_in_alt) : hipFree(sh.d_in_alt));
    if (sh.d_out) (void)hipFree(sh.d_out);
    if (sh.d_clips) (void)hipFree(sh.d_clips);
    if (sh.h_in) (void)hipHostFree(sh.h_in);
    if (sh.h_out) (void)hipHostFree(sh.h_out);
    if (sh.h_done) (void)hipHostFree(sh.h_done);
    if (sh.ev) (void)hipEventDestroy(sh.ev);
    if (sh.st) (void)hipStreamDestroy(sh.st);
    sh = StreamShell();
}

static void plan_release(hipsoxr_plan *h)
{
    std::lock_guard<std::mutex> lk(g_cache_mu);
    if (h->users > 0) --h->users;
}

// ------------------------------------------------------------------------------------------------
// library / plan
// ------------------------------------------------------------------------------------------------
extern "C" {

const char *hipsoxr_version(void) { return "hipsoxr-" HIPSOXR_VERSION_STRING " (gfx950)"; }

int hipsoxr_device_count(void) { return device_count(); }

hipsoxr_error_t hipsoxr_plan_create(double in_rate, double out_rate, unsigned long recipe,
                                    hipsoxr_plan_t **out)
{
    if (!out) return "null argument";
    *out = nullptr;
    hipsoxr_plan *h = new (std::nothrow) hipsoxr_plan();
    if (!h) return "out of memory";
    if (const char *e = plan_design(in_rate, out_rate, recipe, &h->p)) {
        delete h;
        return e;
    }
    *out = h;
    return nullptr;
}

hipsoxr_error_t hipsoxr_plan_create_vr(double in_rate, double out_rate, unsigned long recipe,
                                       hipsoxr_plan_t **out)
{
    if (!out) return "null argument";
    *out = nullptr;
    hipsoxr_plan *h = new (std::nothrow) hipsoxr_plan();
    if (!h) return "out of memory";
    if (const char *e = plan_design(in_rate, out_rate, recipe, &h->p, /*force_interp=*/true)) {
        delete h;
        return e;
    }
    *out = h;
    return nullptr;
}

void hipsoxr_plan_delete(hipsoxr_plan_t *h)
{
    if (h && h->cached) return; // a stream's plan (hipsoxr_stream_plan) belongs to the plan cache, not to the caller
    delete h;
}

hipsoxr_error_t hipsoxr_plan_info(const hipsoxr_plan_t *h, hipsoxr_plan_info_t *info)
{
    if (!h || !info) return "null argument";
    const Plan &p = h->p;
    info->in_rate = p.in_rate; info->out_rate = p.out_rate; info->recipe = p.recipe;
    info->L = p.L; info->M = p.M; info->taps = p.T; info->interpolated = p.phases;
    info->precision_bits = p.q.bits; info->passband_end = p.q.passband_end;
    info->stopband_begin = p.q.stopband_begin; info->att_db = p.att_db; info->kaiser_beta = p.beta;
    info->bank_elems = (uint64_t)p.bank.size();
    return nullptr;
}

hipsoxr_error_t hipsoxr_plan_get_bank(const hipsoxr_plan_t *h, double *dst, size_t n)
{
    if (!h || !dst) return "null argument";
    if (n != h->p.bank.size()) return "bank size mismatch";
    std::memcpy(dst, h->p.bank.data(), n * sizeof(double));
    return nullptr;
}

static uint64_t bank_hash(const double *b, size_t n)
{
    uint64_t hsh = 1469598103934665603ULL;
    const unsigned char *p = reinterpret_cast<const unsigned char *>(b);
    for (size_t i = 0; i < n * sizeof(double); ++i) { hsh ^= p[i]; hsh *= 1099511628211ULL; }
    return hsh;
}
// the plan's bank is about to be replaced by `src` (which differs from it): is the plan on a bank of its own afterwards?
static void note_bank_change(Plan &p, const double *src, size_t n)
{
    if (!p.custom_bank && !p.have_designed_hash) { p.designed_hash = bank_hash(p.bank.data(), p.bank.size()); p.have_designed_hash = true; }
    p.custom_bank = !p.have_designed_hash || bank_hash(src, n) != p.designed_hash;
}

hipsoxr_error_t hipsoxr_plan_set_bank(hipsoxr_plan_t *h, const double *src, size_t n)
{
    if (!h || !src) return "null argument";
    if (n != h->p.bank.size()) return "bank size mismatch";
    if (h->cached) return "this plan is shared through the plan cache (it belongs to a stream); create one with hipsoxr_plan_create";
    // The bank every rank designs for itself is deterministic: installing the root's copy of it (the broadcast's usual
    // case) changes nothing and keeps every derived table.  A bank that differs becomes the plan's filter for every
    // engine that reads `bank`; the two-stage form, which samples the analytic prototype, declines the plan until the designed
    // bank is installed again (its hash is kept).
    if (std::memcmp(h->p.bank.data(), src, n * sizeof(double)) == 0) return nullptr;
    device_bank_release(&h->p);
    fft_release(&h->p);
    twostage_release(&h->p);
    note_bank_change(h->p, src, n);
    std::memcpy(h->p.bank.data(), src, n * sizeof(double));
    return nullptr;
}

// The one collective of the multi-GPU path: the shared bank, from `root` to every rank
// of the communicator.  RCCL is resolved at run time from whatever copy the process already holds
// (PyTorch bundles its own librccl: a communicator is only valid inside the library that made it),
// falling back to the system's librccl.so; libhipsoxr.so itself does not link against it.
hipsoxr_error_t hipsoxr_plan_broadcast(hipsoxr_plan_t *h, void *nccl_comm, int root, int my_rank, void *hip_stream)
{
    if (!h || !nccl_comm) return "null argument";
    if (h->cached) return "this plan is shared through the plan cache (it belongs to a stream); create one with hipsoxr_plan_create";
    typedef int (*bcast_fn)(const void *, void *, size_t, int, int, void *, hipStream_t);
    static bcast_fn bcast = nullptr;
    if (!bcast) {
        void *sym = dlsym(RTLD_DEFAULT, "ncclBroadcast");
        if (!sym) {
            void *lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
            if (lib) sym = dlsym(lib, "ncclBroadcast");
        }
        if (!sym) return "RCCL not found (no ncclBroadcast in this process and no librccl.so to load)";
        bcast = (bcast_fn)sym;
    }
    if (device_count() <= 0) return kNoDevice;
    const size_t n = h->p.bank.size();
    hipStream_t st = (hipStream_t)hip_stream;
    double *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, n * sizeof(double)));
    const char *err = nullptr;
    do {
        if (my_rank == root && hipMemcpyAsync(d, h->p.bank.data(), n * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess) {
            err = "hipMemcpy failed"; break;
        }
        if (bcast(d, d, n, /*ncclFloat64*/ 8, root, nccl_comm, st) != 0) { err = "ncclBroadcast failed"; break; }
        if (my_rank != root) {
            std::vector<double> got(n);
            if (hipMemcpyAsync(got.data(), d, n * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipStreamSynchronize(st) != hipSuccess) { err = "hipMemcpy failed"; break; }
            if (std::memcmp(h->p.bank.data(), got.data(), n * sizeof(double)) != 0) { // (see hipsoxr_plan_set_bank)
                device_bank_release(&h->p);
                fft_release(&h->p);
                twostage_release(&h->p);
                note_bank_change(h->p, got.data(), n);
                h->p.bank.swap(got);
            }
        } else if (hipStreamSynchronize(st) != hipSuccess) {
            err = "hip sync failed";
        }
    } while (0);
    (void)hipFree(d);
    return err;
}

uint64_t hipsoxr_plan_out_len(const hipsoxr_plan_t *h, uint64_t in_len)
{
    return h ? plan_out_len(h->p, in_len) : 0;
}

hipsoxr_error_t hipsoxr_run_device(hipsoxr_plan_t *h, const hipsoxr_job_t *job, void *hip_stream)
{
    if (!h || !job) return "null argument";
    if (job->elem < 0 || job->elem > 3) return "invalid element type";
    if (job->out_frames < 0 || job->in_frames < 0 || job->out_k0 < 0) return "invalid job extent";
    if ((job->out_frames > 0 && !job->out) || (job->out_frames > 0 && job->in_frames > 0 && !job->in)) return "null buffer";
    if (device_count() <= 0) return kNoDevice;
    return launch_job(&h->p, *job, hip_stream);
}

} // extern "C"

// ------------------------------------------------------------------------------------------------
// stream internals
// ------------------------------------------------------------------------------------------------
static inline size_t esz(const hipsoxr_stream *s) { return elem_size(s->elem); }

// Number of outputs computable from the first N input frames without zero-extension:
// output k needs inputs up to floor(k*M/L) + T/2.
static uint64_t k_avail(const Plan &p, uint64_t N)
{
    const int64_t H = p.T / 2;
    if ((int64_t)N - 1 - H < 0) return 0;
    unsigned __int128 Q = (unsigned __int128)(N - 1 - (uint64_t)H);
    unsigned __int128 v = ((Q + 1) * (unsigned __int128)p.L - 1) / (unsigned __int128)p.M;
    return (uint64_t)v + 1;
}

// Absolute index of the first input sample the next output (k_done) needs.
static int64_t first_needed(const hipsoxr_stream *s)
{
    const Plan &p = s->plan->p;
    if (s->vr.on) return (int64_t)(s->vr.pos(s->k_done) >> 64) - (p.T / 2 - 1);
    int64_t n0, ph;
    locate(p, (int64_t)s->k_done, &n0, &ph);
    return n0;
}

// Variable rate: number of outputs [0, K) computable from N input frames without zero-extension
// (output k reads up to floor(t(k)) + T/2), or — at end of input — the total K with
// t(k) + step(k)/2 <= N (the constant-rate rule floor(N*L/M + 1/2), restated for a moving step).
static uint64_t vr_k_limit(const hipsoxr_stream *s, bool ended)
{
    const VrState &v = s->vr;
    const int64_t H = s->plan->p.T / 2;
    const i128 N = (i128)s->n_in_total << 64;
    auto ok = [&](uint64_t k) -> bool {
        if (ended) return v.pos(k) + v.step(k) / 2 <= N;
        return (int64_t)(v.pos(k) >> 64) + H <= (int64_t)s->n_in_total - 1;
    };
    uint64_t lo = s->k_done; // invariant: every k < lo is ok (already emitted, or checked)
    if (!ok(lo)) return lo;
    uint64_t span = 1;
    while (ok(lo + span)) { lo += span; span <<= 1; } // t is strictly increasing: exponential + binary search
    uint64_t hi = lo + span;                           // ok(lo), !ok(hi)
    while (hi - lo > 1) {
        const uint64_t mid = lo + (hi - lo) / 2;
        if (ok(mid)) lo = mid; else hi = mid;
    }
    return hi;
}

// Move the still-needed tail of the staged input to the front of the alternate buffer.
static void ring_free(hipsoxr_stream *s, void *p)
{
    if (!p) return;
    if (s->ring_on_host) (void)hipHostFree(p); else (void)hipFree(p);
}

// Host ring: make room for `ilen` more frames (drop what no output needs any more, grow by powers of two).
// Nothing may be reading the ring meanwhile: the stream is drained first (it is idle in synchronous mode).
static const char *host_ring_reserve(hipsoxr_stream *s, size_t ilen)
{
    const size_t frame = (size_t)s->ch * esz(s);
    if (s->d_in && s->in_fill + ilen <= s->in_cap) return nullptr;
    if (!s->res.running) HIP_TRY(stream_wait(s)); // (a resident kernel is idle between calls: it has answered the last one)
    const int64_t n0 = first_needed(s);
    const int64_t keep_from = std::min<int64_t>(std::max<int64_t>(n0, s->in_base), s->in_base + (int64_t)s->in_fill);
    const size_t drop = (size_t)(keep_from - s->in_base), keep = s->in_fill - drop;
    size_t cap = std::max<size_t>(s->in_cap, 1024);
    while (cap < keep + 8 * ilen) cap <<= 1; // compaction every eighth call
    if (cap != s->in_cap || !s->d_in) {
        resident_stop(s); // the ring moves: the next call launches another instance on the new one
        void *nb = nullptr;
        if (hipHostMalloc(&nb, cap * frame, hipHostMallocDefault) != hipSuccess) return "hipHostMalloc failed";
        if (keep) std::memcpy(nb, (char *)s->d_in + drop * frame, keep * frame);
        if (s->d_in) (void)hipHostFree(s->d_in);
        s->d_in = nb; s->in_cap = cap;
    } else if (keep && drop) {
        std::memmove(s->d_in, (char *)s->d_in + drop * frame, keep * frame);
    }
    s->in_base = keep_from;
    s->in_fill = keep;
    return nullptr;
}

// A large chunk arrives on a host-ring stream: move what is pending to device memory and stay there.
static const char *host_ring_to_device(hipsoxr_stream *s)
{
    const size_t frame = (size_t)s->ch * esz(s);
    HIP_TRY(stream_wait(s));
    void *host = s->d_in;
    const size_t fill = s->in_fill, cap = std::max<size_t>(s->in_cap, 1024);
    void *dev = nullptr;
    HIP_TRY(hipMalloc(&dev, cap * frame));
    if (fill) HIP_TRY(hipMemcpy(dev, host, fill * frame, hipMemcpyHostToDevice));
    if (host) (void)hipHostFree(host);
    if (s->d_in_alt) (void)hipHostFree(s->d_in_alt);
    s->ring_on_host = false;
    s->d_in = dev; s->in_cap = cap;
    s->d_in_alt = nullptr; s->alt_cap = 0;
    return nullptr;
}

static const char *stream_compact(hipsoxr_stream *s, size_t want_cap)
{
    const int64_t n0 = first_needed(s);
    int64_t keep_from = std::max<int64_t>(n0, s->in_base);
    keep_from = std::min<int64_t>(keep_from, s->in_base + (int64_t)s->in_fill);
    const size_t drop = (size_t)(keep_from - s->in_base), keep = s->in_fill - drop;
    const size_t cap = std::max(want_cap, s->in_cap);
    if (drop == 0 && cap == s->in_cap) return nullptr;
    if (s->alt_cap < cap || !s->d_in_alt) {
        if (s->d_in_alt) HIP_TRY(hipFree(s->d_in_alt));
        s->d_in_alt = nullptr;
        HIP_TRY(hipMalloc(&s->d_in_alt, cap * s->ch * esz(s)));
        s->alt_cap = cap;
    }
    if (keep) {
        if (!s->split) {
            HIP_TRY(hipMemcpyAsync(s->d_in_alt, (char *)s->d_in + drop * s->ch * esz(s),
                                   keep * s->ch * esz(s), hipMemcpyDeviceToDevice, s->st));
        } else {
            HIP_TRY(hipMemcpy2DAsync(s->d_in_alt, s->alt_cap * esz(s), (char *)s->d_in + drop * esz(s),
                                     s->in_cap * esz(s), keep * esz(s), s->ch,
                                     hipMemcpyDeviceToDevice, s->st));
        }
    }
    std::swap(s->d_in, s->d_in_alt);
    std::swap(s->in_cap, s->alt_cap);
    s->in_base = keep_from;
    s->in_fill = keep;
    return nullptr;
}

static const char *stream_append(hipsoxr_stream *s, const void *in, size_t ilen)
{
    const size_t bytes_in = ilen * s->ch * esz(s);
    if (!s->split && !s->ring_on_host && s->n_in_total == 0 && bytes_in <= kHostRingChunk && !switches().no_host_ring) {
        // first chunk of a stream, and a small one: ring in pinned host memory.  A DEVICE ring inherited from the
        // pool (the stream before this one fed large chunks) is given up for it: keeping it cost every later
        // small-chunk stream of the process the host ring and with it the resident kernel — 441-frame calls 23 us
        // instead of 15 (round 3; finished streams hand their ring, of either kind, back to the pool)
        if (s->d_in) { (void)hipFree(s->d_in); s->d_in = nullptr; s->in_cap = 0; }
        if (s->d_in_alt) { (void)hipFree(s->d_in_alt); s->d_in_alt = nullptr; s->alt_cap = 0; }
        s->ring_on_host = true;
    }
    if (s->ring_on_host) {
        if (bytes_in > kHostRingChunk) {
            if (const char *e = host_ring_to_device(s)) return e;
        } else {
            if (const char *e = host_ring_reserve(s, ilen)) return e;
            std::memcpy((char *)s->d_in + s->in_fill * s->ch * esz(s), in, bytes_in);
            s->in_fill += ilen;
            s->n_in_total += ilen;
            return nullptr;
        }
    }
    if (s->in_fill + ilen > s->in_cap) {
        // retire consumed input first; grow (power of two) only if that is not enough
        const int64_t n0 = first_needed(s);
        int64_t keep_from = std::min<int64_t>(std::max<int64_t>(n0, s->in_base),
                                              s->in_base + (int64_t)s->in_fill);
        size_t keep = s->in_fill - (size_t)(keep_from - s->in_base);
        // room for the history plus four chunks of this size: compaction (a device-to-device copy)
        // then runs every fourth call instead of every call
        size_t need = keep + 4 * ilen, cap = std::max<size_t>(s->in_cap, 1024);
        if (need > ((size_t)1 << 24)) need = keep + ilen;
        while (cap < need) cap <<= 1;
        if (const char *e = stream_compact(s, cap)) return e;
    }
    const size_t chunk_bytes = ilen * s->ch * esz(s);
    const bool bounce = chunk_bytes <= kPinnedMax && !pinned_ensure(&s->h_in, &s->h_in_bytes, chunk_bytes);
    if (!s->split) {
        const void *src = in;
        if (bounce) { std::memcpy(s->h_in, in, chunk_bytes); src = s->h_in; }
        HIP_TRY(hipMemcpyAsync((char *)s->d_in + s->in_fill * s->ch * esz(s), src, chunk_bytes,
                               hipMemcpyHostToDevice, s->st));
    } else {
        const void *const *chans = (const void *const *)in;
        for (unsigned c = 0; c < s->ch; ++c) {
            const void *src = chans[c];
            if (bounce) {
                src = (char *)s->h_in + (size_t)c * ilen * esz(s);
                std::memcpy((void *)src, chans[c], ilen * esz(s));
            }
            HIP_TRY(hipMemcpyAsync((char *)s->d_in + ((size_t)c * s->in_cap + s->in_fill) * esz(s), src,
                                   ilen * esz(s), hipMemcpyHostToDevice, s->st));
        }
    }
    s->in_fill += ilen;
    s->n_in_total += ilen;
    return nullptr;
}

// Emit up to olen frames (host destination).  `out_off` = frame offset into the caller's buffers.
static const char *stream_emit_once(hipsoxr_stream *s, void *out, size_t olen, size_t *odone);

// Variable-rate streams evaluate one position law per launch, so a call that crosses the end of a
// slew is served by two launches; everything else is a single one.
static const char *stream_emit(hipsoxr_stream *s, void *out, size_t olen, size_t *odone)
{
    if (!s->vr.on) return stream_emit_once(s, out, olen, odone);
    size_t total = 0;
    std::vector<void *> chans(s->split ? s->ch : 0);
    for (int pass = 0; pass < 4 && total < olen; ++pass) {
        void *o = out;
        if (s->split) {
            for (unsigned c = 0; c < s->ch; ++c) chans[c] = (char *)((void *const *)out)[c] + total * esz(s);
            o = chans.data();
        } else {
            o = (char *)out + total * s->ch * esz(s);
        }
        size_t got = 0;
        if (const char *e = stream_emit_once(s, o, olen - total, &got)) return e;
        total += got;
        if (!got) break;
        // another pass only if this one stopped at the end of a slew (an empty pass costs a wait on the stream)
        if (!(s->vr.n_slew && s->k_done == s->vr.k_s + s->vr.n_slew)) break;
    }
    *odone = total;
    return nullptr;
}

// Synchronous small chunk through the resident kernel: post the call's numbers, spin on the answer.
// Returns nullptr with *served = false when this call has to take the ordinary path.
static const size_t kCtlSlots = 1024;
static const uint32_t kDoneWords = 64; // (larger launches: an event — 200 workgroups reporting one by one took 10 us longer than it)
static const char *resident_emit(hipsoxr_stream *s, const hipsoxr_job_t &j, bool *served, const VrPos *vp = nullptr)
{
    hipsoxr_stream::Resident &r = s->res;
    *served = false;
    if (r.failed >= 2) return nullptr;
    if (!r.box) {
        if (hipHostMalloc((void **)&r.box, sizeof(ResidentBox), hipHostMallocDefault) != hipSuccess) { r.box = nullptr; r.failed = 2; return nullptr; }
        std::memset(r.box, 0, sizeof(ResidentBox));
        if (hipMalloc((void **)&r.ctl, kCtlSlots * sizeof(ResidentCtl)) != hipSuccess) { r.ctl = nullptr; r.failed = 2; return nullptr; }
        HIP_TRY(hipMemsetAsync(r.ctl, 0, kCtlSlots * sizeof(ResidentCtl), s->st));
        r.ctl_next = 0;
        r.words = r.box->w;
        hipDeviceProp_t pr;
        if (!switches().resident_no_bar && hipGetDeviceProperties(&pr, s->device) == hipSuccess && pr.isLargeBar &&
            hipExtMallocWithFlags(&r.words_dev, 128, hipDeviceMallocFinegrained) == hipSuccess) {
            r.words = (volatile uint64_t *)r.words_dev;
            for (int i = 0; i < 16; ++i) r.words[i] = 0;
            __builtin_ia32_sfence();
        }
    }
    hipsoxr_job_t jr = j;
    if (r.words_dev) { // bring the mirror of the ring up to date (appended frames only, unless the ring was compacted or moved)
        const size_t frame = (size_t)s->ch * esz(s), need = s->in_cap * frame;
        if (r.mirror_bytes < need || !r.mirror) {
            resident_stop(s);
            if (r.mirror) (void)hipFree(r.mirror);
            r.mirror = nullptr; r.mirror_bytes = 0;
            if (hipExtMallocWithFlags(&r.mirror, need, hipDeviceMallocFinegrained) != hipSuccess) { r.mirror = nullptr; r.failed = 2; return nullptr; }
            r.mirror_bytes = need; r.mirror_of = nullptr;
        }
        if (r.mirror_of != s->d_in || r.mirror_base != s->in_base || r.mirror_fill > s->in_fill) { r.mirror_fill = 0; r.mirror_of = s->d_in; r.mirror_base = s->in_base; }
        if (s->in_fill > r.mirror_fill)
            std::memcpy((char *)r.mirror + r.mirror_fill * frame, (const char *)s->d_in + r.mirror_fill * frame, (s->in_fill - r.mirror_fill) * frame);
        r.mirror_fill = s->in_fill;
        jr.in = r.mirror;
    }
    if (r.running && (r.in != jr.in || r.out != j.out || j.out_frames > r.max_out)) resident_stop(s);
    auto launch = [&](uint32_t base_seq) -> const char * {
        if (r.ctl_next == kCtlSlots) { // every arbiter word has been used: wipe them (ordered behind the last instance)
            HIP_TRY(hipMemsetAsync(r.ctl, 0, kCtlSlots * sizeof(ResidentCtl), s->st));
            r.ctl_next = 0;
        }
        ResidentLaunch rl;
        rl.box = r.box; rl.words = (const uint64_t *)r.words; rl.ctl = r.ctl + r.ctl_next++; rl.base_seq = base_seq;
        if (++r.epoch == 0) ++r.epoch;
        rl.epoch = r.epoch; rl.idle_us = std::min(kResidentWatchdogUs, std::max(50, switches().resident_idle_us));
        hipsoxr_job_t cap = jr; // room for chunks a quarter longer than this one
        cap.out_frames = std::max<int64_t>(64, j.out_frames + j.out_frames / 4 + 2);
        std::lock_guard<std::mutex> reserve(g_resident_mu);
        rl.used_mcu = g_resident_mcu.load(); // (the launcher knows occupancy and CU count: it refuses what does not fit)
        rl.budget_shift = s->resident_by_auto ? 3 : 1; // the share of the chip resident instances may hold: 1/8 (auto) or 1/2
        if (const char *e = launch_job(&s->plan->p, cap, s->st, vp, &rl)) { // (vp: only says "variable rate" here — every message carries its own clock)
            if (rl.over_budget) return ""; // over the budget: the ordinary path, this time
            (void)e; // not a job the resident form serves: the ordinary path does
            ++r.failed;
            return "";
        }
        g_resident_mcu += (int64_t)rl.cost_mcu;
        r.running = true; r.in = jr.in; r.out = j.out; r.max_out = rl.max_out; r.n_wgs = rl.n_wgs; r.cost_mcu = rl.cost_mcu; r.failed = 0;
        return nullptr;
    };
    if (!r.running) {
        if (const char *e = launch(r.seq)) return *e ? e : nullptr;
        if (j.out_frames > r.max_out) { resident_stop(s); ++r.failed; return nullptr; }
    }
    const uint32_t seq = r.seq + 1;
    if (!resident_post(s->plan->p, r.words, seq, j.in_abs0, j.in_frames, j.out_k0, j.out_frames, vp)) { resident_stop(s); return nullptr; }
    r.seq = seq;
    volatile uint32_t *done = r.box->done, *exited = &r.box->exited;
    const auto t0 = std::chrono::steady_clock::now();
    unsigned next = 0; // workgroups [0, next) have answered
    for (uint64_t spin = 0;; ++spin) {
        while (next < r.n_wgs && done[next] == seq) ++next;
        if (next == r.n_wgs) break;
        if (*exited == r.epoch) {
            // the instance left (idle for too long) before it saw the message, which is still in the box: the next
            // instance takes it.  (An instance answers a message completely or not at all: k_chain_resident.)
            (void)hipStreamSynchronize(s->st);
            r.running = false;
            g_resident_mcu -= (int64_t)r.cost_mcu;
            while (next < r.n_wgs && done[next] == seq) ++next;
            if (next == r.n_wgs) break;
            // (refused — over the budget, say: nobody has touched the message, the ordinary path serves this call;
            //  a later instance starts behind it, at r.seq, and ignores the stale words)
            if (const char *e = launch(seq - 1)) return *e ? e : nullptr;
            next = 0;
        }
        __builtin_ia32_pause();
        if ((spin & 0xfff) == 0xfff && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(10)) {
            // Retire the instance (leave word, wait for it to drain, give its budget back) and stop using the resident
            // path on this stream: later calls must not post to a dead instance or hold capacity it no longer uses.
            resident_stop(s);
            r.failed = 2;
            return "resident kernel does not answer";
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
#ifdef HIPSOXR_RES_TRACE
    if ((r.seq & 1023) == 1000) {
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        std::fprintf(stderr, "resident msg %u: host post->done %.2f us; workgroup 0: idle+poll %.2f us, body %.2f, fence %.2f\n",
                     r.seq, us, r.box->pad[0] * 0.01, r.box->pad[1] * 0.01, r.box->pad[2] * 0.01);
        std::fprintf(stderr, "   workgroup 0 body: positions %.2f us, (span %.2f) staged %.2f, chains %.2f, stored %.2f\n", r.box->pad[5] * 0.01, r.box->pad[9] * 0.01, r.box->pad[6] * 0.01,
                     r.box->pad[7] * 0.01, r.box->pad[8] * 0.01);
    }
#endif
    *served = true;
    return nullptr;
}

static const char *stream_emit_once(hipsoxr_stream *s, void *out, size_t olen, size_t *odone)
{
    const Plan &p = s->plan->p;
    VrState &v = s->vr;
    if (v.on && v.n_slew && s->k_done >= v.k_s + v.n_slew) { // slew finished: renormalise to a constant segment
        const uint64_t k1 = v.k_s + v.n_slew;
        v.t_s = v.pos(k1); v.k_s = k1; v.s0 = v.s1; v.delta = 0; v.n_slew = 0;
    }
    const uint64_t k_end = v.on ? vr_k_limit(s, s->ended)
                                : s->ended ? plan_out_len(p, s->n_in_total) : k_avail(p, s->n_in_total);
    size_t n = 0;
    if (k_end > s->k_done) n = (size_t)std::min<uint64_t>(k_end - s->k_done, olen);
    // one launch evaluates one quadratic: stop at the end of a slew (stream_emit comes back for the rest)
    if (v.on && v.n_slew && s->k_done + n > v.k_s + v.n_slew) n = (size_t)(v.k_s + v.n_slew - s->k_done);
    *odone = n;
    if (n == 0) {
        // the caller's input buffer is borrowed only for the call; a host ring took its copy with memcpy, nothing is queued
        if (!s->res.running && !s->ring_on_host) HIP_TRY(stream_wait(s));
        return nullptr;
    }
    if (n > s->out_cap) {
        size_t cap = 1024;
        while (cap < n) cap <<= 1;
        if (s->d_out) HIP_TRY(hipFree(s->d_out));
        s->d_out = nullptr;
        HIP_TRY(hipMalloc(&s->d_out, cap * s->ch * esz(s)));
        s->out_cap = cap;
    }
    // Small results are written by the kernel straight into pinned host memory (it is mapped into
    // the device's address space): no device-to-host copy call, just the completion wait.
    const size_t out_bytes = n * s->ch * esz(s);
    if (s->res.running && out_bytes > s->h_out_bytes) resident_stop(s); // (the result buffer is about to move)
    // ... up to 1 MiB where the output is unit-stride (mono, split layouts); interleaved multi-channel results only up to
    // 64 KiB: every 4-byte sample then lands in host memory as a write of its own, and how many of them the fabric merges
    // depends on which workgroups happen to store side by side — 44.1k -> 48k stereo, 100 000 frames: 348 us against 121
    // through device memory and a copy; 16k -> 48k 8 channels, 10 000 frames: 366 against 109 (tools/direct_ab.py).  The
    // copy path costs ~17 us more where the direct writes did merge (48k -> 44.1k stereo, 20 000 frames: 52 -> 70 us).
    const size_t direct_max = switches().direct_max > 0 ? (size_t)switches().direct_max
                              : (s->ch == 1 || s->split) ? kPinnedMax : kHostRingChunk;
    const bool direct = out_bytes <= direct_max && !pinned_ensure(&s->h_out, &s->h_out_bytes, out_bytes);
    hipsoxr_job_t j;
    std::memset(&j, 0, sizeof j);
    j.in = s->d_in; j.out = direct ? s->h_out : s->d_out; j.elem = s->elem;
    j.kernel = HIPSOXR_KERNEL_EXACT; // bit-exact chunk invariance
    j.n_clips = 1; j.n_channels = s->ch;
    if (!s->split) {
        j.in_frame_stride = s->ch; j.in_chan_stride = 1;
        j.out_frame_stride = s->ch; j.out_chan_stride = 1;
    } else {
        j.in_frame_stride = 1; j.in_chan_stride = (int64_t)s->in_cap;
        j.out_frame_stride = 1; j.out_chan_stride = direct ? (int64_t)n : (int64_t)s->out_cap;
    }
    j.in_abs0 = s->in_base; j.in_frames = (int64_t)s->in_fill;
    j.out_k0 = (int64_t)s->k_done; j.out_frames = (int64_t)n;
    j.clip_counter = s->d_clips;
    j.dither = (s->elem == HIPSOXR_I16 && !(s->flags & HIPSOXR_NO_DITHER)) ? 1u : 0u;
    j.dither_seed = s->dither_seed;
    if (s->in_fill == 0) { // nothing staged yet (e.g. flush of an empty stream): any valid pointer
        j.in = s->d_out;
    }
    ChainDone cd;
    cd.words = nullptr; cd.cap = 0; cd.seq = 0;
    bool served = false;
    const bool small_call = s->ring_on_host && direct && !s->split && s->in_fill > 0 && n <= 2048;
    VrPos vp = {0, 0, 0, 0, 0, 0};
    if (v.on) { // this launch's (or message's) clock: position and step at the first output, step increment while a slew lasts
        const i128 T0 = v.pos(s->k_done), S0 = v.step(s->k_done), D = s->k_done < v.k_s + v.n_slew ? v.delta : 0;
        vp = VrPos{(uint64_t)((u128)T0 >> 64), (uint64_t)(u128)T0, (uint64_t)((u128)S0 >> 64), (uint64_t)(u128)S0,
                   (uint64_t)((u128)D >> 64), (uint64_t)(u128)D};
    }
    if (!s->resident && s->resident_auto_ok) {
        const auto now = std::chrono::steady_clock::now();
        const auto gap = std::chrono::microseconds(std::min(kResidentWatchdogUs, std::max(50, switches().resident_idle_us)) / 2);
        s->small_run = (small_call && (s->small_run == 0 || now - s->last_small < gap)) ? s->small_run + 1 : 0;
        s->last_small = now;
        if (s->small_run >= kAutoResidentRun) s->resident = s->resident_by_auto = true;
    } else if (s->resident_by_auto) {
        // a stream that turned resident by itself drops back the moment the run breaks: a call that is not small, one
        // that comes after a gap (its instance has idled out or is about to), or an instance that left on its own
        const auto now = std::chrono::steady_clock::now();
        const auto gap = std::chrono::microseconds(std::min(kResidentWatchdogUs, std::max(50, switches().resident_idle_us)) / 2);
        if (!small_call || now - s->last_small >= gap || (s->res.running && s->res.box->exited == s->res.epoch)) {
            resident_stop(s);
            s->resident = s->resident_by_auto = false;
            s->small_run = small_call ? 1 : 0;
        }
        s->last_small = now;
    }
    if (s->resident && small_call) {
        if (const char *e = resident_emit(s, j, &served, v.on ? &vp : nullptr)) return e;
    }
    if (served) {
        std::memcpy(out, s->h_out, out_bytes);
        s->k_done += n;
        return nullptr;
    }
    resident_stop(s);
    if (direct && !s->h_done && !switches().no_done_words &&
        hipHostMalloc((void **)&s->h_done, kDoneWords * sizeof(uint32_t), hipHostMallocDefault) == hipSuccess)
        std::memset(s->h_done, 0, kDoneWords * sizeof(uint32_t));
    cd.words = s->h_done; cd.cap = s->h_done && direct ? kDoneWords : 0; cd.seq = ++s->done_seq;
    if (v.on) {
        if (const char *e = launch_job(&s->plan->p, j, s->st, &vp, nullptr, cd.cap ? &cd : nullptr)) return e;
    } else {
        if (const char *e = launch_job(&s->plan->p, j, s->st, nullptr, nullptr, cd.cap ? &cd : nullptr)) return e;
    }
    if (direct) {
        bool seen = false;
        if (cd.n_wgs) { // the kernel reports by itself (ChainDone): no event
            volatile uint32_t *w = cd.words;
            unsigned next = 0;
            for (uint64_t spin = 0; spin < (1ULL << 22); ++spin) { // (~10 ms: then the ordinary wait, which also reports errors)
                while (next < cd.n_wgs && w[next] == cd.seq) ++next;
                if (next == cd.n_wgs) { seen = true; break; }
                __builtin_ia32_pause();
            }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
        }
        if (!seen) HIP_TRY(stream_wait(s));
        if (!s->split) {
            std::memcpy(out, s->h_out, out_bytes);
        } else {
            void *const *chans = (void *const *)out;
            for (unsigned c = 0; c < s->ch; ++c)
                std::memcpy(chans[c], (char *)s->h_out + (size_t)c * n * esz(s), n * esz(s));
        }
    } else if (!s->split) {
        HIP_TRY(hipMemcpyAsync(out, s->d_out, out_bytes, hipMemcpyDeviceToHost, s->st));
        HIP_TRY(stream_wait(s));
    } else {
        void *const *chans = (void *const *)out;
        for (unsigned c = 0; c < s->ch; ++c)
            HIP_TRY(hipMemcpyAsync(chans[c], (char *)s->d_out + (size_t)c * s->out_cap * esz(s), n * esz(s),
                                   hipMemcpyDeviceToHost, s->st));
        HIP_TRY(stream_wait(s));
    }
    s->k_done += n;
    return nullptr;
}

// Deferred output.  The reference's contract lets a call return any number of frames, including none
// (README.md:77-78: "[0, 0, 0, 186, 186, 166, ...]"; CSoxr::process returns whatever soxr_process produced,
// src/soxr_ext.cpp:162-187): libsoxr itself buffers internally.  With HIPSOXR_DEFER a call
//   1. hands out what the PREVIOUS call's launch produced (complete by now: one event query),
//   2. copies its input to the device ring (pinned bounce buffer, asynchronous copy),
//   3. launches the kernel for everything computable now, writing into the other pinned result buffer,
// and returns without waiting for 2 and 3.  The concatenated output is bit-identical to the synchronous
// mode; only the call on which a frame appears moves (one call later), and delay() counts it as pending.
// Constant-rate interleaved streams with small chunks (results <= 1 MiB); anything else runs synchronously.
// Hand out (up to olen frames of) what the last deferred launch produced, waiting for it if need be.
static const char *deferred_take(hipsoxr_stream *s, void *out, size_t olen, size_t *got)
{
    *got = 0;
    if (s->pend_off >= s->pend_n) return nullptr;
    const size_t frame = (size_t)s->ch * esz(s);
    if (s->ev_res[s->pend_slot]) HIP_TRY(hipEventSynchronize(s->ev_res[s->pend_slot]));
    else HIP_TRY(hipStreamSynchronize(s->st));
    const size_t take = std::min(s->pend_n - s->pend_off, olen);
    std::memcpy(out, (char *)s->h_res[s->pend_slot] + s->pend_off * frame, take * frame);
    s->pend_off += take;
    *got = take;
    return nullptr;
}

static const char *stream_process_deferred(hipsoxr_stream *s, const void *in, size_t ilen, void *out, size_t olen,
                                           size_t *odone)
{
    const Plan &p = s->plan->p;
    const size_t frame = (size_t)s->ch * esz(s);
    size_t got = 0;
    // 1. the previous launch's result
    if (const char *e = deferred_take(s, out, olen, &got)) return e;
    // 2. this call's input.  Device ring: through one of two pinned bounce buffers, alternating; the copy
    //    issued from this one two calls ago may still sit in the stream behind a long kernel (no launch, hence
    //    no completion observed, while the caller drains a backlog): its own event says when it is free.
    if (in && ilen) {
        if (s->ended) return "Input after last input";
        const int b = (int)(s->calls & 1);
        const size_t bytes = ilen * frame;
        if (s->ring_on_host || ((!s->split || !s->d_in) && s->n_in_total == 0 && bytes <= kHostRingChunk && !switches().no_host_ring)) {
            if (const char *e = stream_append(s, in, ilen)) return e; // host ring: a memcpy, nothing in flight to protect
        } else if (bytes <= kPinnedMax && (!s->ev_src[b] || hipEventSynchronize(s->ev_src[b]) == hipSuccess) &&
                   !pinned_ensure(&s->h_src[b], &s->h_src_bytes[b], bytes)) {
            if (s->in_fill + ilen > s->in_cap) { // retire / grow the ring exactly as stream_append does
                void *keep_h = s->h_in; size_t keep_b = s->h_in_bytes;
                s->h_in = s->h_src[b]; s->h_in_bytes = s->h_src_bytes[b];
                const char *e = stream_append(s, in, ilen);
                s->h_src[b] = s->h_in; s->h_src_bytes[b] = s->h_in_bytes;
                s->h_in = keep_h; s->h_in_bytes = keep_b;
                if (e) return e;
            } else {
                std::memcpy(s->h_src[b], in, bytes);
                HIP_TRY(hipMemcpyAsync((char *)s->d_in + s->in_fill * frame, s->h_src[b], bytes, hipMemcpyHostToDevice, s->st));
                s->in_fill += ilen;
                s->n_in_total += ilen;
            }
            if (!s->ev_src[b] && hipEventCreateWithFlags(&s->ev_src[b], hipEventDisableTiming) != hipSuccess) s->ev_src[b] = nullptr;
            if (s->ev_src[b]) HIP_TRY(hipEventRecord(s->ev_src[b], s->st));
            else HIP_TRY(hipStreamSynchronize(s->st));
        } else {
            if (const char *e = stream_append(s, in, ilen)) return e;
            HIP_TRY(stream_wait(s)); // large chunk straight from the caller's (borrowed) buffer
        }
        ++s->calls;
    }
    // 3. launch for what is computable now, once the previous result is fully handed out
    if (s->pend_off == s->pend_n) {
        s->pend_n = s->pend_off = 0;
        const uint64_t k_end = k_avail(p, s->n_in_total);
        size_t n = k_end > s->k_done ? (size_t)(k_end - s->k_done) : 0;
        if (n * frame > kPinnedMax) {
            // More than one pinned result buffer holds (chunks whose output exceeds 1 MiB, or a backlog left by a
            // caller who drained slowly): one capped launch per call would fall behind by the excess on every call
            // and the ring would grow without bound.  This call runs synchronously instead — everything computable
            // now, as far as the caller's buffer reaches; the rest stays pending exactly as in synchronous mode.
            size_t more = 0;
            if (got < olen)
                if (const char *e = stream_emit(s, (char *)out + got * frame, olen - got, &more)) return e;
            *odone = got + more;
            return nullptr;
        }
        const int slot = s->pend_slot ^ 1;
        if (n && !pinned_ensure(&s->h_res[slot], &s->h_res_bytes[slot], n * frame)) {
            if (!s->ev_res[slot] && hipEventCreateWithFlags(&s->ev_res[slot], hipEventDisableTiming) != hipSuccess)
                s->ev_res[slot] = nullptr;
            hipsoxr_job_t j;
            std::memset(&j, 0, sizeof j);
            j.in = s->d_in; j.out = s->h_res[slot]; j.elem = s->elem;
            j.kernel = HIPSOXR_KERNEL_EXACT;
            j.n_clips = 1; j.n_channels = s->ch;
            j.in_frame_stride = s->ch; j.in_chan_stride = 1;
            j.out_frame_stride = s->ch; j.out_chan_stride = 1;
            j.in_abs0 = s->in_base; j.in_frames = (int64_t)s->in_fill;
            j.out_k0 = (int64_t)s->k_done; j.out_frames = (int64_t)n;
            j.clip_counter = s->d_clips;
            j.dither = (s->elem == HIPSOXR_I16 && !(s->flags & HIPSOXR_NO_DITHER)) ? 1u : 0u;
            j.dither_seed = s->dither_seed;
            if (const char *e = launch_job(&s->plan->p, j, s->st)) return e;
            if (s->ev_res[slot]) HIP_TRY(hipEventRecord(s->ev_res[slot], s->st));
            s->k_done += n;
            s->pend_n = n; s->pend_off = 0; s->pend_slot = slot;
        }
    }
    *odone = got;
    return nullptr;
}

static const char *stream_new(hipsoxr_plan *plan, bool own, unsigned ch, hipsoxr_datatype_t io,
                              unsigned long flags, hipsoxr_stream **out)
{
    if (ch < 1) return "invalid channel count";
    if ((int)io < 0 || (int)io > 7) return "invalid io datatype";
    if ((flags & HIPSOXR_VR) && !plan->p.phases) return "variable-rate streams need an interpolated-phase plan";
    if (device_count() <= 0) return kNoDevice;
    hipsoxr_stream *s = new (std::nothrow) hipsoxr_stream();
    if (!s) return "out of memory";
    s->plan = plan; s->own_plan = own; s->ch = ch;
    s->elem = (int)io & 3; s->split = s->split_io = ((int)io & 4) != 0; s->flags = flags;
    s->defer = (flags & HIPSOXR_DEFER) && !(flags & HIPSOXR_VR) && !s->split;
    s->resident = ((flags & HIPSOXR_RESIDENT) || switches().resident) && !s->defer && !s->split;
    s->resident_auto_ok = !s->resident && !s->defer && !s->split && ((flags & HIPSOXR_AUTO_RESIDENT) || switches().auto_resident);
    if (flags & HIPSOXR_VR) {
        const double io0 = plan->p.in_rate / plan->p.out_rate;
        if (!(io0 > 9.5367431640625e-07) || !(io0 < 1048576.)) { delete s; return "io ratio out of range for variable rate"; }
        s->vr.on = true; s->vr.max_io = io0;
        s->vr.s0 = s->vr.s1 = q64(io0);
    }
    const char *err = nullptr;
    do {
        int dev = -1;
        (void)hipGetDevice(&dev);
        s->device = dev;
        {
            std::lock_guard<std::mutex> lk(g_pool_mu);
            for (size_t i = g_pool.size(); i-- > 0;) // newest first: the shell a loop of same-sized calls just gave back has its
                if (g_pool[i].device == dev) {       // pinned buffers at that size already (oldest-first cycled through all eight, growing each)
                    StreamShell sh = g_pool[i];
                    g_pool.erase(g_pool.begin() + (long)i);
                    const size_t frame = (size_t)ch * esz(s);
                    s->st = sh.st; s->d_clips = sh.d_clips; s->ev = sh.ev;
                    s->h_in = sh.h_in; s->h_in_bytes = sh.h_in_bytes; s->h_out = sh.h_out; s->h_out_bytes = sh.h_out_bytes;
                    s->h_done = sh.h_done; s->done_seq = sh.done_seq;
                    s->ring_on_host = sh.ring_on_host && !s->split;
                    if (sh.ring_on_host && s->split) { // (split layouts keep the ring on the device)
                        if (sh.d_in) (void)hipHostFree(sh.d_in);
                        if (sh.d_in_alt) (void)hipHostFree(sh.d_in_alt);
                        sh.d_in = sh.d_in_alt = nullptr; sh.in_bytes = sh.alt_bytes = 0;
                    }
                    s->d_in = sh.d_in; s->in_cap = sh.in_bytes / frame;
                    s->d_in_alt = sh.d_in_alt; s->alt_cap = sh.alt_bytes / frame;
                    s->d_out = sh.d_out; s->out_cap = sh.out_bytes / frame;
                    if (!s->in_cap && s->d_in) { ring_free(s, s->d_in); s->d_in = nullptr; }
                    if (!s->alt_cap && s->d_in_alt) { ring_free(s, s->d_in_alt); s->d_in_alt = nullptr; }
                    if (!s->out_cap && s->d_out) { (void)hipFree(s->d_out); s->d_out = nullptr; }
                    break;
                }
        }
        if (!s->st && hipStreamCreateWithFlags(&s->st, hipStreamNonBlocking) != hipSuccess) {
            err = "hipStreamCreate failed"; break;
        }
        if (!s->ev && hipEventCreateWithFlags(&s->ev, hipEventDisableTiming) != hipSuccess) s->ev = nullptr;
        if (!s->d_clips && hipMalloc((void **)&s->d_clips, sizeof(uint64_t)) != hipSuccess) {
            err = "hipMalloc failed"; break;
        }
        if (hipMemsetAsync(s->d_clips, 0, sizeof(uint64_t), s->st) != hipSuccess) {
            err = "hipMemset failed"; break;
        }
        err = device_bank_ensure(&plan->p, engine_prec(s->elem));
    } while (0);
    std::snprintf(s->engine_name, sizeof s->engine_name, "hip-gfx950-%s",
                  engine_prec(s->elem) == 0 ? "f32" : "f64");
    if (err) {
        hipsoxr_stream_delete(s);
        return err;
    }
    *out = s;
    return nullptr;
}

extern "C" {

hipsoxr_error_t hipsoxr_stream_create(double in_rate, double out_rate, unsigned num_channels,
                                      hipsoxr_datatype_t io_type, unsigned long recipe,
                                      unsigned long flags, hipsoxr_stream_t **out)
{
    if (!out) return "null argument";
    *out = nullptr;
    hipsoxr_plan_t *plan = nullptr;
    // variable rate: positions are not tied to L/M, always the interpolated-phase table
    if (const char *e = plan_acquire(in_rate, out_rate, recipe, (flags & HIPSOXR_VR) != 0, &plan)) return e;
    if (const char *e = stream_new(plan, false, num_channels, io_type, flags, out)) {
        plan_release(plan);
        return e;
    }
    (*out)->own_plan = true; // "own" = holds a reference of the cached plan
    return nullptr;
}

hipsoxr_error_t hipsoxr_stream_create_with_plan(hipsoxr_plan_t *plan, unsigned num_channels,
                                                hipsoxr_datatype_t io_type, unsigned long flags,
                                                hipsoxr_stream_t **out)
{
    if (!out || !plan) return "null argument";
    *out = nullptr;
    return stream_new(plan, false, num_channels, io_type, flags, out);
}

// Device calls of this stream may still be in flight on the caller's HIP stream: wait for them (before a host-pointer
// call, clear, delete or a counter read touches what they use).
static void ext_sync(hipsoxr_stream *s)
{
    if (s->ext_pending && s->ext_st) (void)hipStreamSynchronize(s->ext_st);
    s->ext_pending = false;
}

void hipsoxr_stream_delete(hipsoxr_stream_t *s)
{
    if (!s) return;
    DeviceGuard guard(s->device);
    ext_sync(s);
    resident_stop(s);
    if (s->st) (void)hipStreamSynchronize(s->st);
    if (s->res.box) (void)hipHostFree(s->res.box);
    if (s->res.words_dev) (void)hipFree(s->res.words_dev);
    if (s->res.mirror) (void)hipFree(s->res.mirror);
    if (s->res.ctl) (void)hipFree(s->res.ctl);
    for (int i = 0; i < 2; ++i) {
        if (s->h_res[i]) (void)hipHostFree(s->h_res[i]);
        if (s->h_src[i]) (void)hipHostFree(s->h_src[i]);
        if (s->ev_res[i]) (void)hipEventDestroy(s->ev_res[i]);
        if (s->ev_src[i]) (void)hipEventDestroy(s->ev_src[i]);
    }
    StreamShell sh;
    sh.device = s->device; // where the resources were created, not whatever device is current now
    sh.ring_on_host = s->ring_on_host;
    const size_t frame = (size_t)s->ch * esz(s);
    sh.st = s->st; sh.d_clips = s->d_clips; sh.ev = s->ev;
    sh.h_in = s->h_in; sh.h_in_bytes = s->h_in_bytes; sh.h_out = s->h_out; sh.h_out_bytes = s->h_out_bytes;
    sh.h_done = s->h_done; sh.done_seq = s->done_seq;
    sh.d_in = s->d_in; sh.in_bytes = s->d_in ? s->in_cap * frame : 0;
    sh.d_in_alt = s->d_in_alt; sh.alt_bytes = s->d_in_alt ? s->alt_cap * frame : 0;
    sh.d_out = s->d_out; sh.out_bytes = s->d_out ? s->out_cap * frame : 0;
    bool pooled = false;
    if (sh.st && sh.d_clips) {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        size_t total = sh.in_bytes + sh.alt_bytes + sh.out_bytes;
        for (const StreamShell &o : g_pool) total += o.in_bytes + o.alt_bytes + o.out_bytes;
        if (g_pool.size() < kPoolMax && total <= kPoolBytes) { g_pool.push_back(sh); pooled = true; }
    }
    if (!pooled) shell_free(sh);
    if (s->own_plan) plan_release(s->plan);
    delete s;
}

static const char *stream_process_inner(hipsoxr_stream *s, const void *in, size_t ilen, void *out, size_t olen, size_t *odone);

// ---- device chunks (hipsoxr_stream_process_device) -----------------------------------------------------------------
// Same counters, ring, clock and launch as the host path; the chunk arrives by a device-to-device copy, the outputs go
// straight to the caller's device buffer, and everything is enqueued on the caller's HIP stream (for the duration of
// the call it stands in for the stream's own).  Nothing waits.
static const char *device_emit_once(hipsoxr_stream *s, void *d_out, size_t olen, size_t *odone)
{
    const Plan &p = s->plan->p;
    VrState &v = s->vr;
    if (v.on && v.n_slew && s->k_done >= v.k_s + v.n_slew) { // slew finished: renormalise to a constant segment
        const uint64_t k1 = v.k_s + v.n_slew;
        v.t_s = v.pos(k1); v.k_s = k1; v.s0 = v.s1; v.delta = 0; v.n_slew = 0;
    }
    const uint64_t k_end = v.on ? vr_k_limit(s, s->ended)
                                : s->ended ? plan_out_len(p, s->n_in_total) : k_avail(p, s->n_in_total);
    size_t n = 0;
    if (k_end > s->k_done) n = (size_t)std::min<uint64_t>(k_end - s->k_done, olen);
    if (v.on && v.n_slew && s->k_done + n > v.k_s + v.n_slew) n = (size_t)(v.k_s + v.n_slew - s->k_done); // one quadratic per launch
    *odone = n;
    if (n == 0) return nullptr;
    hipsoxr_job_t j;
    std::memset(&j, 0, sizeof j);
    j.in = s->in_fill ? s->d_in : d_out; j.out = d_out; j.elem = s->elem;
    j.kernel = HIPSOXR_KERNEL_EXACT;
    j.n_clips = 1; j.n_channels = s->ch;
    j.in_frame_stride = s->ch; j.in_chan_stride = 1;
    j.out_frame_stride = s->ch; j.out_chan_stride = 1;
    j.in_abs0 = s->in_base; j.in_frames = (int64_t)s->in_fill;
    j.out_k0 = (int64_t)s->k_done; j.out_frames = (int64_t)n;
    j.clip_counter = s->d_clips;
    j.dither = (s->elem == HIPSOXR_I16 && !(s->flags & HIPSOXR_NO_DITHER)) ? 1u : 0u;
    j.dither_seed = s->dither_seed;
    if (v.on) {
        const i128 T0 = v.pos(s->k_done), S0 = v.step(s->k_done), D = s->k_done < v.k_s + v.n_slew ? v.delta : 0;
        const VrPos vp = {(uint64_t)((u128)T0 >> 64), (uint64_t)(u128)T0, (uint64_t)((u128)S0 >> 64), (uint64_t)(u128)S0,
                          (uint64_t)((u128)D >> 64), (uint64_t)(u128)D};
        if (const char *e = launch_job(&s->plan->p, j, s->st, &vp, nullptr, nullptr)) return e;
    } else {
        if (const char *e = launch_job(&s->plan->p, j, s->st, nullptr, nullptr, nullptr)) return e;
    }
    s->k_done += n;
    return nullptr;
}

// Room in the device ring for `ilen` more frames (retire consumed input first; grow — power of two, room for four chunks,
// so that compaction runs every fourth call — only if that is not enough).  Enqueues on s->st.
static const char *device_ring_reserve(hipsoxr_stream *s, size_t ilen)
{
    if (s->ring_on_host) { if (const char *e = host_ring_to_device(s)) return e; } // (a ring inherited from a small-chunk stream)
    if (s->in_fill + ilen > s->in_cap) {
        const int64_t n0 = first_needed(s);
        const int64_t keep_from = std::min<int64_t>(std::max<int64_t>(n0, s->in_base), s->in_base + (int64_t)s->in_fill);
        const size_t keep = s->in_fill - (size_t)(keep_from - s->in_base);
        size_t need = keep + 4 * ilen, cap = std::max<size_t>(s->in_cap, 1024);
        if (need > ((size_t)1 << 24)) need = keep + ilen;
        while (cap < need) cap <<= 1;
        if (const char *e = stream_compact(s, cap)) return e;
    }
    return nullptr;
}

// A constant-rate device call as ONE item of the small-launch kernel (round 5): the chunk is read where the caller left it
// and copied into the ring by the launch's own workgroups — and when the ring is full, the frames still needed move to the
// stream's other buffer in the same launch (no copy command, no extra dispatch: with 128 streams in one call a
// hipMemcpyAsync per compacting stream was 70 of the call's 87 us).  Fills `it` and returns true when the call is of that
// kind (constant rate, a chunk to append, every output the input so far determines — what device_emit_once would emit in
// one go — fewer than 4096 of them); the stream's counters are NOT touched (stream_item_commit does that once the launch
// is in).  *err: an allocation failed.
static bool stream_item_prepare(hipsoxr_stream *s, const void *d_in, size_t ilen, void *d_out, size_t olen, ChainItem *it, size_t *n_out,
                                const char **err)
{
    const Plan &p = s->plan->p;
    *err = nullptr;
    if (s->vr.on || s->ended || !d_in || !ilen || !d_out || s->ring_on_host || s->split) return false;
    const uint64_t n_total = s->n_in_total + ilen;
    const uint64_t k_end = k_avail(p, n_total);
    const size_t n = k_end > s->k_done ? (size_t)std::min<uint64_t>(k_end - s->k_done, olen) : 0;
    if (n >= 4096) return false;
    it->ring = s->d_in; it->ring_dst = s->d_in; it->keep_from = 0;
    if (s->in_fill + ilen > s->in_cap || !s->d_in) { // the ring moves: keep [first frame the next output needs, end), into the other buffer
        const int64_t n0 = first_needed(s);
        const int64_t keep_from = std::min<int64_t>(std::max<int64_t>(n0, s->in_base), s->in_base + (int64_t)s->in_fill);
        const size_t keep = s->in_fill - (size_t)(keep_from - s->in_base);
        // room for sixteen SMALL chunks (the move then runs every ~16th call), for four of the larger ones (advisor, round 5: both
        // buffers of a stream end up at this capacity — with thousands of grouped streams the factor is memory)
        size_t need = keep + std::min<size_t>(16 * ilen, std::max<size_t>(4 * ilen, 65536)), cap = std::max<size_t>(s->in_cap, 1024);
        if (need > ((size_t)1 << 24)) need = keep + ilen;
        while (cap < need) cap <<= 1;
        if (s->alt_cap < cap || !s->d_in_alt) {
            if (s->d_in_alt && hipFree(s->d_in_alt) != hipSuccess) { *err = "hipFree failed"; return false; }
            s->d_in_alt = nullptr; s->alt_cap = 0;
            if (hipMalloc(&s->d_in_alt, cap * s->ch * esz(s)) != hipSuccess) { *err = "out of device memory (stream ring)"; return false; }
            s->alt_cap = cap;
        }
        it->ring_dst = s->d_in_alt;
        it->keep_from = keep_from - s->in_base;
        if (!s->d_in) it->ring = s->d_in_alt; // (first call: nothing to keep, and the launch reads only the chunk)
    }
    it->chunk = d_in; it->out = d_out; it->clip_counter = s->d_clips;
    it->in_abs0 = s->in_base; it->in_frames = (int64_t)(s->in_fill + ilen); it->split = (int64_t)s->in_fill; it->chunk_frames = (int64_t)ilen;
    it->out_k0 = (int64_t)s->k_done; it->out_frames = (int64_t)n;
    const __int128 kM = (__int128)it->out_k0 * p.M;
    it->d0 = (int64_t)(kM / p.L); it->p0 = (int64_t)(kM % p.L);
    it->dither_seed = s->dither_seed; it->pad = 0; it->reserved = 0;
    *n_out = n;
    return true;
}
static void stream_item_commit(hipsoxr_stream *s, const ChainItem &it, size_t ilen, size_t n)
{
    if (it.ring_dst != s->d_in) { // the launch moved the ring
        std::swap(s->d_in, s->d_in_alt);
        std::swap(s->in_cap, s->alt_cap);
        s->in_base += it.keep_from;
        s->in_fill -= (size_t)it.keep_from;
    }
    s->in_fill += ilen;
    s->n_in_total += ilen;
    s->k_done += n;
}

static const char *device_process(hipsoxr_stream *s, const void *d_in, size_t ilen, void *d_out, size_t olen, size_t *odone)
{
    const size_t frame = (size_t)s->ch * esz(s);
    if (d_in == nullptr) {
        s->ended = true;
    } else if (ilen > 0) {
        if (s->ended) return "Input after last input";
        if (s->ring_on_host) { if (const char *e = host_ring_to_device(s)) return e; } // (a ring inherited from a small-chunk stream)
        {   // one dispatch: the small-launch kernel appends the chunk (and moves a full ring) itself
            ChainItem it;
            size_t n = 0;
            const char *perr = nullptr;
            if (stream_item_prepare(s, d_in, ilen, d_out, olen, &it, &n, &perr)) {
                bool handled = false;
                const bool dither = s->elem == HIPSOXR_I16 && !(s->flags & HIPSOXR_NO_DITHER);
                if (const char *e = launch_chain_items(&s->plan->p, s->elem, s->ch, dither, &it, nullptr, 1, s->st, &handled)) return e;
                if (handled) {
                    stream_item_commit(s, it, ilen, n);
                    *odone = n;
                    return nullptr;
                }
            }
            if (perr) return perr;
        }
        if (const char *e = device_ring_reserve(s, ilen)) return e;
        if (const char *e = launch_copy((char *)s->d_in + s->in_fill * frame, d_in, ilen * frame, s->st)) return e;
        s->in_fill += ilen;
        s->n_in_total += ilen;
    }
    size_t total = 0;
    while (d_out && total < olen) {
        size_t got = 0;
        if (const char *e = device_emit_once(s, (char *)d_out + total * frame, olen - total, &got)) return e;
        total += got;
        if (!got) break;
        if (!(s->vr.on && s->vr.n_slew && s->k_done == s->vr.k_s + s->vr.n_slew)) break; // another pass only at the end of a slew
    }
    *odone = total;
    return nullptr;
}

hipsoxr_error_t hipsoxr_stream_process_device(hipsoxr_stream_t *s, const void *in, size_t ilen, void *out, size_t olen,
                                              size_t *odone, void *hip_stream)
{
    if (!s || !odone) return "null argument";
    *odone = 0;
    if (s->split || s->split_io || s->defer || (s->flags & (HIPSOXR_RESIDENT | HIPSOXR_AUTO_RESIDENT)))
        return "device chunks: interleaved streams without the deferred / resident flags only";
    DeviceGuard guard(s->device);
    resident_stop(s);
    hipStream_t own = s->st, user = (hipStream_t)hip_stream;
    // Ordering against the stream's OWN HIP stream, where host-pointer calls and the pool put their work: what such calls
    // left there comes first (one event, only when there was such a call since the last device call); the other way
    // round — device calls still in flight when a host-pointer call, clear or delete arrives — `ext_sync` waits.
    if (user != own && s->own_used && s->ev) {
        HIP_TRY(hipEventRecord(s->ev, own));
        HIP_TRY(hipStreamWaitEvent(user, s->ev, 0));
    }
    s->own_used = false;
    if (s->ext_st && s->ext_st != user) HIP_TRY(hipStreamSynchronize(s->ext_st)); // (a different caller stream than last time)
    s->ext_st = user; s->ext_pending = true;
    s->st = user;
    const char *err = device_process(s, in, ilen, out, olen, odone);
    s->st = own;
    return err;
}

// ---- many independent streams, one launch (round 5) -------------------------------------------------------------------
// The item table is written into pinned, device-mapped host memory and staged into a device mirror by a copy KERNEL in front
// of the launch (one coalesced sweep over PCIe; letting every workgroup of the launch read its item from host memory cost
// 85 us per call for 128 streams: thousands of 64-byte PCIe reads).  Two halves of kItemsHalf items used in turn; before a
// half is used again the launches that read it are waited for (an event per caller stream, recorded when the half is left).
static constexpr size_t kItemsHalf = 8192;
struct ItemArena {
    ChainItem *host = nullptr, *dev = nullptr, *mirror = nullptr; // pinned table, its device address, the device copy
    size_t pos = 0;
    int half = 0;
    std::vector<hipStream_t> used[2];
    std::vector<hipEvent_t> pending[2];
    std::vector<hipEvent_t> spare;
    std::mutex mu;
};
static ItemArena &item_arena(int device)
{
    static std::mutex mu;
    static std::vector<std::pair<int, ItemArena *>> all;
    std::lock_guard<std::mutex> lk(mu);
    for (auto &e : all)
        if (e.first == device) return *e.second;
    all.push_back({device, new ItemArena});
    return *all.back().second;
}
// room for n items (n <= kItemsHalf); the arena's mutex is held by the caller
static const char *arena_take(ItemArena &ar, size_t n, hipStream_t st, ChainItem **host, ChainItem **dev, ChainItem **mirror)
{
    static_assert(sizeof(ChainItem) % 16 == 0, "the table is staged 16 bytes per thread");
    if (!ar.host) {
        HIP_TRY(hipHostMalloc((void **)&ar.host, 2 * kItemsHalf * sizeof(ChainItem), hipHostMallocMapped));
        HIP_TRY(hipHostGetDevicePointer((void **)&ar.dev, ar.host, 0));
        HIP_TRY(hipMalloc((void **)&ar.mirror, 2 * kItemsHalf * sizeof(ChainItem)));
    }
    if (ar.pos + n > kItemsHalf) { // leave this half: mark what still reads it, take the other one once ITS readers are done
        bool lost = false; // a stream of this half can no longer be asked (its owner destroyed it: C-API callers may; torch never does)
        for (hipStream_t u : ar.used[ar.half]) {
            hipEvent_t ev = nullptr;
            if (!ar.spare.empty()) { ev = ar.spare.back(); ar.spare.pop_back(); }
            else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); lost = true; continue; }
            if (hipEventRecord(ev, u) != hipSuccess) { (void)hipGetLastError(); ar.spare.push_back(ev); lost = true; continue; }
            ar.pending[ar.half].push_back(ev);
        }
        ar.used[ar.half].clear(); // (on every path: a stream that failed once is not asked again)
        ar.half ^= 1; ar.pos = 0;
        hipError_t bad = hipSuccess;
        for (hipEvent_t ev : ar.pending[ar.half]) { // every event goes back to the pool, whatever its wait returns
            const hipError_t e = hipEventSynchronize(ev);
            if (e != hipSuccess) bad = e;
            ar.spare.push_back(ev);
        }
        ar.pending[ar.half].clear();
        if (lost || bad != hipSuccess) { // whoever still reads either half is waited for the blunt way, once
            (void)hipGetLastError();
            HIP_TRY(hipDeviceSynchronize());
        }
    }
    if (std::find(ar.used[ar.half].begin(), ar.used[ar.half].end(), st) == ar.used[ar.half].end()) ar.used[ar.half].push_back(st);
    *host = ar.host + ar.half * kItemsHalf + ar.pos;
    *dev = ar.dev + ar.half * kItemsHalf + ar.pos;
    *mirror = ar.mirror + ar.half * kItemsHalf + ar.pos;
    ar.pos += n;
    return nullptr;
}

hipsoxr_error_t hipsoxr_streams_process_device(hipsoxr_stream_t *const *handles, size_t n, const void *const *ins, const size_t *ilens,
                                               void *const *outs, const size_t *olens, size_t *odones, void *hip_stream)
{
    if (!n) return nullptr;
    if (!handles || !ins || !ilens || !outs || !olens || !odones) return "null argument";
    for (size_t i = 0; i < n; ++i) {
        if (!handles[i]) return "null argument";
        odones[i] = 0;
    }
    hipsoxr_stream *s0 = handles[0];
    // one launch serves streams of ONE plan, element type, channel count and dither setting on one device, constant rate,
    // each with a chunk to append and fewer than 4096 outputs due; anything else goes handle by handle (same results)
    bool together = n > 1 && n <= kItemsHalf;
    for (size_t i = 0; together && i < n; ++i) {
        hipsoxr_stream *s = handles[i];
        together = s->plan == s0->plan && s->elem == s0->elem && s->ch == s0->ch && s->device == s0->device &&
                   ((s->flags ^ s0->flags) & HIPSOXR_NO_DITHER) == 0 && !s->vr.on && !s->split && !s->split_io && !s->defer &&
                   !(s->flags & (HIPSOXR_RESIDENT | HIPSOXR_AUTO_RESIDENT)) && !s->ended && ins[i] && ilens[i] && outs[i];
    }
    if (together) { // (a handle twice in one call: its calls must run in order, one by one)
        std::vector<const hipsoxr_stream *> seen(handles, handles + n);
        std::sort(seen.begin(), seen.end());
        together = std::adjacent_find(seen.begin(), seen.end()) == seen.end();
    }
    if (!together) {
        for (size_t i = 0; i < n; ++i)
            if (const char *e = hipsoxr_stream_process_device(handles[i], ins[i], ilens[i], outs[i], olens[i], &odones[i], hip_stream)) return e;
        return nullptr;
    }
    DeviceGuard guard(s0->device);
    hipStream_t user = (hipStream_t)hip_stream;
    // stream-order bookkeeping of every handle, as in hipsoxr_stream_process_device
    for (size_t i = 0; i < n; ++i) {
        hipsoxr_stream *s = handles[i];
        resident_stop(s);
        if (user != s->st && s->own_used && s->ev) {
            HIP_TRY(hipEventRecord(s->ev, s->st));
            HIP_TRY(hipStreamWaitEvent(user, s->ev, 0));
        }
        s->own_used = false;
        if (s->ext_st && s->ext_st != user) HIP_TRY(hipStreamSynchronize(s->ext_st));
        s->ext_st = user; s->ext_pending = true;
        if (s->ring_on_host) {                          // (a ring inherited from a small-chunk stream)
            hipStream_t own = s->st;
            s->st = user;
            const char *e = host_ring_to_device(s);
            s->st = own;
            if (e) return e;
        }
    }
    ItemArena &ar = item_arena(s0->device);
    std::lock_guard<std::mutex> lk(ar.mu);
    ChainItem *host = nullptr, *dev = nullptr, *mirror = nullptr;
    if (const char *e = arena_take(ar, n, user, &host, &dev, &mirror)) return e;
    std::vector<size_t> nout(n);
    bool all = true;
    const char *perr = nullptr;
    for (size_t i = 0; all && i < n; ++i) all = stream_item_prepare(handles[i], ins[i], ilens[i], outs[i], olens[i], &host[i], &nout[i], &perr);
    if (perr) return perr;
    bool handled = false;
    if (all) {
        const bool dither = s0->elem == HIPSOXR_I16 && !(s0->flags & HIPSOXR_NO_DITHER);
        if (const char *e = launch_copy(mirror, dev, n * sizeof(ChainItem), user)) return e;
        if (const char *e = launch_chain_items(&s0->plan->p, s0->elem, s0->ch, dither, host, mirror, (uint32_t)n, user, &handled)) return e;
    }
    if (!handled) { // (a stream with too many outputs due, or a plan the small-launch kernel does not take)
        for (size_t i = 0; i < n; ++i)
            if (const char *e = hipsoxr_stream_process_device(handles[i], ins[i], ilens[i], outs[i], olens[i], &odones[i], hip_stream)) return e;
        return nullptr;
    }
    for (size_t i = 0; i < n; ++i) {
        stream_item_commit(handles[i], host[i], ilens[i], nout[i]);
        odones[i] = nout[i];
    }
    return nullptr;
}

hipsoxr_error_t hipsoxr_stream_process(hipsoxr_stream_t *s, const void *in, size_t ilen, void *out,
                                       size_t olen, size_t *odone)
{
    if (!s || !odone) return "null argument";
    DeviceGuard guard(s->device);
    *odone = 0;
    ext_sync(s);
    s->own_used = true;
    if (s->split_io && !s->adapt_decided && in && ilen) {
        // first chunk of a split-layout stream: a small one puts the stream on the interleaved small-chunk path for good
        // (its per-channel planes are woven / unwoven at this boundary: a few hundred frames per call); a large one keeps
        // the planar device ring, where every channel moves with one copy
        s->adapt_decided = true;
        if (s->n_in_total == 0 && ilen * s->ch * esz(s) <= kHostRingChunk && !switches().no_host_ring) {
            s->adapt = true; s->split = false;
            s->defer = (s->flags & HIPSOXR_DEFER) && !(s->flags & HIPSOXR_VR);
            s->resident = ((s->flags & HIPSOXR_RESIDENT) || switches().resident) && !s->defer;
            s->resident_auto_ok = !s->resident && !s->defer && ((s->flags & HIPSOXR_AUTO_RESIDENT) || switches().auto_resident);
            if (s->d_in && !s->ring_on_host) { (void)hipFree(s->d_in); s->d_in = nullptr; s->in_cap = 0; } // (a planar ring inherited from the pool)
            if (s->d_in_alt && !s->ring_on_host) { (void)hipFree(s->d_in_alt); s->d_in_alt = nullptr; s->alt_cap = 0; }
            if (s->d_out) { (void)hipFree(s->d_out); s->d_out = nullptr; s->out_cap = 0; }
        }
    }
    if (!s->adapt) return stream_process_inner(s, in, ilen, out, olen, odone);
    if (s->ch == 1) // one plane IS the interleaved signal
        return stream_process_inner(s, in ? ((const void *const *)in)[0] : nullptr, ilen, out ? ((void *const *)out)[0] : nullptr, olen, odone);
    const size_t e = esz(s), ch = s->ch;
    const void *in_i = nullptr;
    if (in) {
        s->ad_in.resize(std::max<size_t>(ilen * ch * e, 1));
        const void *const *planes = (const void *const *)in;
        for (size_t c = 0; c < ch; ++c) {
            const char *src = (const char *)planes[c];
            char *dst = s->ad_in.data() + c * e;
            if (e == 4) for (size_t f = 0; f < ilen; ++f) std::memcpy(dst + f * ch * 4, src + f * 4, 4);
            else if (e == 2) for (size_t f = 0; f < ilen; ++f) std::memcpy(dst + f * ch * 2, src + f * 2, 2);
            else for (size_t f = 0; f < ilen; ++f) std::memcpy(dst + f * ch * 8, src + f * 8, 8);
        }
        in_i = s->ad_in.data();
    }
    void *out_i = nullptr;
    if (out && olen) { s->ad_out.resize(olen * ch * e); out_i = s->ad_out.data(); }
    if (const char *err = stream_process_inner(s, in_i, ilen, out_i, olen, odone)) return err;
    if (out_i && *odone) {
        void *const *planes = (void *const *)out;
        const size_t n = *odone;
        for (size_t c = 0; c < ch; ++c) {
            char *dst = (char *)planes[c];
            const char *src = s->ad_out.data() + c * e;
            if (e == 4) for (size_t f = 0; f < n; ++f) std::memcpy(dst + f * 4, src + f * ch * 4, 4);
            else if (e == 2) for (size_t f = 0; f < n; ++f) std::memcpy(dst + f * 2, src + f * ch * 2, 2);
            else for (size_t f = 0; f < n; ++f) std::memcpy(dst + f * 8, src + f * ch * 8, 8);
        }
    }
    return nullptr;
}

static const char *stream_process_inner(hipsoxr_stream *s, const void *in, size_t ilen, void *out, size_t olen, size_t *odone)
{
    if (s->defer && out && olen) {
        if (in != nullptr) return stream_process_deferred(s, in, ilen, out, olen, odone);
        // End of input: what the last launch produced first, then — in the SAME call — the synchronous flush for
        // everything else (frames appended without a launch included).  No deferred launch is made here, so a
        // return of 0 means the stream is dry, which is what drain loops take it to mean (src/soxr_ext.cpp:114-125).
        size_t got = 0;
        if (const char *e = deferred_take(s, out, olen, &got)) return e;
        if (s->pend_off < s->pend_n || got == olen) { *odone = got; return nullptr; } // (olen > 0: got > 0 here)
        s->pend_n = s->pend_off = 0;
        s->ended = true;
        size_t tail = 0;
        if (const char *e = stream_emit(s, (char *)out + got * s->ch * esz(s), olen - got, &tail)) return e;
        *odone = got + tail;
        return nullptr;
    }
    if (in == nullptr) {
        s->ended = true; // end of input: flush
    } else if (ilen > 0) {
        if (s->ended) return "Input after last input";
        if (const char *e = stream_append(s, in, ilen)) return e;
    }
    if (olen == 0 || out == nullptr) {
        if (in && ilen && !s->res.running && !s->ring_on_host) HIP_TRY(stream_wait(s)); // host buffer is borrowed only for the call
        return nullptr;
    }
    return stream_emit(s, out, olen, odone);
}

hipsoxr_error_t hipsoxr_stream_set_dither_seed(hipsoxr_stream_t *s, uint32_t seed)
{
    if (!s) return "null argument";
    if (s->res.running) { // (the seed is a launch argument of the resident kernel)
        DeviceGuard guard(s->device);
        resident_stop(s);
    }
    s->dither_seed = seed;
    return nullptr;
}

hipsoxr_error_t hipsoxr_stream_clear(hipsoxr_stream_t *s)
{
    if (!s) return "null argument";
    DeviceGuard guard(s->device);
    ext_sync(s);
    resident_stop(s);
    s->res.mirror_of = nullptr; // (the ring starts over at frame 0: nothing in the device mirror is current)
    if (s->st) (void)hipStreamSynchronize(s->st);
    s->ended = false; s->n_in_total = 0; s->k_done = 0; s->in_base = 0; s->in_fill = 0;
    s->pend_n = s->pend_off = 0;
    if (s->vr.on) { // fresh signal at the ratio last requested
        s->vr.k_s = 0; s->vr.t_s = 0; s->vr.s0 = s->vr.s1; s->vr.delta = 0; s->vr.n_slew = 0;
    }
    if (s->elem == HIPSOXR_I16 || s->elem == HIPSOXR_I32) { // (float streams never touch the clip counter)
        HIP_TRY(hipMemsetAsync(s->d_clips, 0, sizeof(uint64_t), s->st));
        HIP_TRY(stream_wait(s));
    }
    return nullptr;
}

double hipsoxr_stream_delay(hipsoxr_stream_t *s)
{
    if (!s) return 0.;
    const Plan &p = s->plan->p;
    double d;
    if (s->vr.on) { // input not yet passed by the output clock, in output samples at the current step
        const double two64 = 18446744073709551616.;
        d = ((double)s->n_in_total - (double)s->vr.pos(s->k_done) / two64) / ((double)s->vr.step(s->k_done) / two64);
    } else {
        d = (double)s->n_in_total * (double)p.L / (double)p.M - (double)(s->k_done - (uint64_t)(s->pend_n - s->pend_off));
    }
    return d > 0. ? d : 0.;
}

size_t hipsoxr_stream_num_clips(hipsoxr_stream_t *s)
{
    if (!s) return 0;
    DeviceGuard guard(s->device);
    uint64_t v = 0;
    ext_sync(s);
    resident_stop(s);
    if (hipMemcpyAsync(&v, s->d_clips, sizeof v, hipMemcpyDeviceToHost, s->st) != hipSuccess) return 0;
    (void)hipStreamSynchronize(s->st);
    return (size_t)v;
}

const char *hipsoxr_stream_engine(hipsoxr_stream_t *s) { return s ? s->engine_name : ""; }

// reference: soxr_set_io_ratio, src/soxr_ext.cpp:200-204.  Takes effect at the next output frame
// (index k_done); with slew_len > 0 the step moves linearly to the new value over that many OUTPUT
// frames, otherwise at once.  The input position is continuous across the change.
hipsoxr_error_t hipsoxr_stream_set_io_ratio(hipsoxr_stream_t *s, double io_ratio, size_t slew_len)
{
    if (!s) return "null argument";
    if (!s->vr.on) return "set_io_ratio needs a stream created with the variable-rate flag (SOXR_VR)";
    if (!(io_ratio > 9.5367431640625e-07) || !(io_ratio < 1048576.)) return "io ratio out of range";
    if (io_ratio > s->vr.max_io * (1. + 1e-12))
        return "io ratio exceeds the maximum given at creation (the filter would alias)";
    if (slew_len > ((size_t)1 << 40)) return "slew length out of range";
    VrState &v = s->vr;
    const i128 t_now = v.pos(s->k_done), s_now = v.step(s->k_done), s_new = q64(io_ratio);
    v.k_s = s->k_done; v.t_s = t_now; v.s1 = s_new;
    if (slew_len > 0) { v.s0 = s_now; v.n_slew = slew_len; v.delta = (s_new - s_now) / (i128)slew_len; }
    else { v.s0 = s_new; v.n_slew = 0; v.delta = 0; }
    return nullptr;
}

hipsoxr_plan_t *hipsoxr_stream_plan(hipsoxr_stream_t *s) { return s ? s->plan : nullptr; }

hipsoxr_error_t hipsoxr_oneshot(double in_rate, double out_rate, unsigned num_channels,
                                const void *in, size_t ilen, void *out, size_t olen, size_t *odone,
                                hipsoxr_datatype_t io_type, unsigned long recipe,
                                unsigned long flags)
{
    if (!odone) return "null argument";
    *odone = 0;
    hipsoxr_stream_t *s = nullptr;
    if (const char *e = hipsoxr_stream_create(in_rate, out_rate, num_channels, io_type, recipe, flags, &s))
        return e;
    const char *err = nullptr;
    if (ilen) err = stream_append(s, in, ilen);
    s->ended = true; // whole signal known: a single launch covers body and tail
    if (!err && olen && out) err = stream_emit(s, out, olen, odone);
    else if (!err) err = (hipStreamSynchronize(s->st) == hipSuccess) ? nullptr : "hip sync failed";
    hipsoxr_stream_delete(s);
    return err;
}

} // extern "C"
