// kernels.hip — the soxr_process hot path on CDNA4 (gfx950): the EXACT (canonical-order) engine.
// Hand-written HIP; the f32 throughput kernels run the canonical fma chains on the f32-input matrix pipe.
//
// Replaces the inner product libsoxr runs inside soxr_process (reference call sites
// src/soxr_ext.cpp:163-166, :245-248, :328-331):
//     y[k] = sum_j bank[(k*M) mod L][j] * x[floor(k*M/L) - (T/2-1) + j]
// followed by the conversion to the I/O type (round-half-even, saturate, clip count, TPDF dither
// for int16).
//
// CANONICAL ARITHMETIC (shared with oracle/soxr_oracle.c *_port, bit for bit):
//     accL = 0; for j = 0 .. T/2-1 ascending : accL = fma(c[j], x[j], accL)
//     accR = 0; for j = T-1 .. T/2 descending: accR = fma(c[j], x[j], accR)
//     y = accL + accR
// in the engine precision Real (float for f32/i16 I/O, double for f64/i32 I/O).  Each output
// sample is one pair of serial FMA chains that depends on nothing but its own taps, so results are
// independent of chunking, tiling, launch geometry and kernel choice (the bit-exact invariances of
// reference tests test_divide_match / test_stream_length).  Zero-padded table entries contribute
// fma(0, x, acc) == acc exactly for finite x.
//
// Kernels (all bit-identical to each other and to the oracle; DESIGN.md §5.1 has the table):
//   k_gather        one lane per output sample; coefficients gathered from the tap-major bank
//                   [T][Lpad], input read straight from global memory.  Universal fallback: every
//                   ratio / layout / length.
//   k_chain         small launches (streaming chunks, < 4096 outputs): a workgroup stages the coefficient
//                   rows and the input span its outputs share into LDS in one round trip, then two waves run
//                   the two canonical half-chains of every output.  Exact, interpolated and variable-rate
//                   plans.  Reports completion through words in pinned host memory (ChainDone) when asked.
//   k_chain_resident  the same body as a RESIDENT kernel: launched once, fed through a mailbox (pinned host
//                   memory, or device memory the CPU stores into on large-BAR systems) — no HIP call per chunk.
//   k_interp(_tile) interpolated-phase plans (arbitrary ratios) and variable rate: per tap a cubic in
//                   the fractional position (Horner FMAs); lane per output (fallback) / outputs sorted by phase
//                   interval, large launches.
//   k_gather_wave   exact-bank launches of 4096 outputs and more that the period tiles do not take (too few periods for a
//                   slab; stream chunks written straight into host memory): a half-chain per QUAD of lanes on the
//                   phase-major bank, the chain by DPP — in place of lane-per-output k_gather.
//   k_interp_wave   the same for launches between 512 outputs and what fills the chip (a stream's 96 000-frame
//                   chunk, 1 s clips), and every large variable-rate launch: a half-chain per QUAD of lanes —
//                   lane k fetches and evaluates tap 4s + k, the chain takes the four coefficients by DPP.
//   k_tile          period-tiled VALU kernel: 64 (32, 16 where LDS demands) periods of one column staged in LDS, a wave = 16
//                   output phases whose coefficients travel on the scalar path (s_load -> SGPR operands
//                   of v_pk_fma_f32).  The f64 engine (float64 / int32 I/O); f32 A/B reference.
//   k_tile_mfma(_p) the same tiling on v_mfma_f32_16x16x4_f32 — on gfx950 the f32-input MFMA IS the
//                   k-ordered fmaf chain of the canonical arithmetic, bit for bit, at the vector ALU's
//                   peak rate, with both operands in VGPRs (prefetchable arbitrarily deep; SGPR-fed VALU
//                   FMAs top out at 47-61 TF here).  _p: k-de-interleaved LDS planes, 4-wave workgroups,
//                   software-pipelined half-chains.  The f32 engine for float32 / int16 I/O.
//   k_wave_dot      reference point only: the shape BASELINE.json's north star describes (one wavefront
//                   per output sample + shuffle reduction): 587 us where k_tile_mfma_p takes 32 —
//                   6 cross-lane steps and two LDS reads per ~4.6 FMAs.  Never chosen automatically.
// The frequency-domain engine (1e-6-class, not bit-identical: whole-signal float32/float64 device jobs)
// lives in fft.hip / fftwave.hip.
// Layout (round 6): this file holds the switches, the conversions, k_gather / k_wave_dot and the whole host side (tables,
// launchers); the kernel families are cut into kernels_interp.h, kernels_chain.h and kernels_tile.h, included below — one
// translation unit still (27 s of the build; the frequency-domain engine's three are the long pole).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <type_traits>
#include <vector>

#include "device.h"

namespace hipsoxr {

const Switches &switches()
{
    static const Switches sw = [] {
        Switches w;
        auto on = [](const char *n) { return getenv(n) != nullptr; };
        auto num = [](const char *n) { const char *v = getenv(n); return v ? atoi(v) : 0; };
        w.no_fft = on("HIPSOXR_NO_FFT"); w.resident = on("HIPSOXR_RESIDENT"); w.auto_resident = on("HIPSOXR_AUTO_RESIDENT");
        if (getenv("HIPSOXR_RESIDENT_IDLE_US")) w.resident_idle_us = num("HIPSOXR_RESIDENT_IDLE_US");
#ifdef HIPSOXR_DEBUG_SWITCHES
        w.fft_no_pair = on("HIPSOXR_FFT_NO_PAIR"); w.fft_no_chpair = on("HIPSOXR_FFT_NO_CHPAIR"); w.fft_no_xcd_map = on("HIPSOXR_FFT_NO_XCD_MAP");
        w.fft_large_only = on("HIPSOXR_FFT_LARGE_ONLY"); w.fft_small_only = on("HIPSOXR_FFT_SMALL_ONLY"); w.fft_no_tiny = on("HIPSOXR_FFT_NO_TINY");
        w.fft_no_wave = on("HIPSOXR_FFT_NO_WAVE"); w.dbg_wave_min = num("HIPSOXR_DEBUG_WAVE_MIN"); w.dbg_wave_slots = num("HIPSOXR_DEBUG_WAVE_SLOTS");
        w.no_planes = on("HIPSOXR_NO_PLANES"); w.no_mfma64 = on("HIPSOXR_NO_MFMA64"); w.no_host_ring = on("HIPSOXR_NO_HOST_RING");
        w.no_chain = on("HIPSOXR_NO_CHAIN"); w.no_done_words = on("HIPSOXR_NO_DONE_WORDS"); w.direct_max = num("HIPSOXR_DEBUG_DIRECT_MAX");
        w.resident_no_bar = on("HIPSOXR_RESIDENT_NO_BAR"); w.no_xcd_split = on("HIPSOXR_NO_XCD_SPLIT"); w.no_tile_split = on("HIPSOXR_NO_TILE_SPLIT");
        w.no_interp_tile = on("HIPSOXR_NO_INTERP_TILE"); w.no_two_stage = on("HIPSOXR_NO_TWO_STAGE");
        w.no_interp_wave = on("HIPSOXR_NO_INTERP_WAVE"); w.no_gather_wave = on("HIPSOXR_NO_GATHER_WAVE"); w.dbg_gw_taps = num("HIPSOXR_DEBUG_GW_TAPS");
        w.dbg_flags = num("HIPSOXR_DEBUG_FLAGS"); w.dbg_nrt = num("HIPSOXR_DEBUG_NRT"); w.dbg_nw = num("HIPSOXR_DEBUG_NW");
        w.dbg_split = num("HIPSOXR_DEBUG_SPLIT"); w.dbg_chain_no = num("HIPSOXR_DEBUG_CHAIN_NO"); w.dbg_lds = (size_t)num("HIPSOXR_DEBUG_LDS");
        w.dbg_slab32 = on("HIPSOXR_DEBUG_SLAB32"); w.no_halves = on("HIPSOXR_DEBUG_NO_HALVES"); w.dbg_pad = on("HIPSOXR_DEBUG_PAD");
        w.dbg_slab64 = on("HIPSOXR_DEBUG_SLAB64"); w.dbg_mfma64_pb = num("HIPSOXR_DEBUG_MFMA64_PB"); w.dbg_mfma64_split = on("HIPSOXR_DEBUG_MFMA64_SPLIT");
        w.dbg_mfma64_lds = (size_t)num("HIPSOXR_DEBUG_MFMA64_LDS"); w.dbg_fft_lds = (size_t)num("HIPSOXR_DEBUG_FFT_LDS");
        w.dbg_tile_form = num("HIPSOXR_DEBUG_TILE_FORM"); w.dbg_poly_r = num("HIPSOXR_DEBUG_POLY_R"); w.no_interp_pair = on("HIPSOXR_NO_INTERP_PAIR"); w.dbg_interp_pair_always = on("HIPSOXR_DEBUG_INTERP_PAIR_ALWAYS"); w.dbg_interp_no_twin = on("HIPSOXR_DEBUG_INTERP_NO_TWIN"); w.poly_no_pair = on("HIPSOXR_POLY_NO_PAIR"); w.dbg_trace = getenv("HIPSOXR_DEBUG_TRACE");
#endif
        return w;
    }();
    return sw;
}

// ---------------------------------------------------------------------------------------------
// conversions
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27; z *= 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return z;
}
// TPDF dither in (-1, 1) LSB: pure function of (seed, channel, absolute output index).
__device__ __forceinline__ float dither_tpdf(uint32_t seed, uint32_t ch, int64_t k)
{
    uint64_t z = mix64((uint64_t)k * 0x9E3779B97F4A7C15ULL + (((uint64_t)ch << 32) | seed));
    int32_t u1 = (int32_t)(z & 0xFFFFFF), u2 = (int32_t)((z >> 24) & 0xFFFFFF);
    return (float)(u1 - u2) * (1.f / 16777216.f);
}

struct OutCtx {
    uint64_t *clip_counter;
    uint32_t dither, seed;
    uint32_t ch0; // channel index of the job's channel 0 in the caller's signal (jobs folded over channel ranges)
};
// set by launch_job while it issues the parts of a job folded over channel ranges: the dither of a
// channel is keyed by its index in the WHOLE signal
static thread_local uint32_t t_ch_base = 0;

// (Plain stores: a lane writes 2-8 bytes, a wave's instruction a part of each line it touches, and the L2's write combining
// is what turns that into whole-line traffic — non-temporal stores cost the 60 s clip 3 %, int32 6 %, 8-channel interleaved
// data 87 %; tools/nt_ab.sh, profiles/r04_cache_policy.txt.  The frequency-domain kernel's staged 16-byte stores are the
// case where they pay: fft.hip, FFT_STORE_AUX.)
template <typename T> __device__ __forceinline__ void put_out(T *p, T v) { *p = v; }
template <typename Real>
__device__ __forceinline__ void store_out(float *p, Real v, const OutCtx &, uint32_t, int64_t)
{
    put_out(p, (float)v);
}
template <typename Real>
__device__ __forceinline__ void store_out(double *p, Real v, const OutCtx &, uint32_t, int64_t)
{
    put_out(p, (double)v);
}
template <typename Real>
__device__ __forceinline__ void store_out(int16_t *p, Real v, const OutCtx &c, uint32_t ch, int64_t k)
{
    float a = (float)v;
    if (c.dither) a = a + dither_tpdf(c.seed, ch + c.ch0, k);
    float r = __builtin_rintf(a);
    bool clip = false;
    if (r > 32767.f) { r = 32767.f; clip = true; }
    else if (r < -32768.f) { r = -32768.f; clip = true; }
    if (clip && c.clip_counter) atomicAdd((unsigned long long *)c.clip_counter, 1ULL);
    put_out(p, (int16_t)r);
}
template <typename Real>
__device__ __forceinline__ void store_out(int32_t *p, Real v, const OutCtx &c, uint32_t, int64_t)
{
    double r = __builtin_rint((double)v);
    bool clip = false;
    if (r > 2147483647.) { r = 2147483647.; clip = true; }
    else if (r < -2147483648.) { r = -2147483648.; clip = true; }
    if (clip && c.clip_counter) atomicAdd((unsigned long long *)c.clip_counter, 1ULL);
    put_out(p, (int32_t)r);
}

__device__ __forceinline__ float fma_r(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_r(double a, double b, double c) { return __builtin_fma(a, b, c); }

// ---------------------------------------------------------------------------------------------
// k_gather
// ---------------------------------------------------------------------------------------------
struct GatherArgs {
    const void *in;
    void *out;
    const void *bank; // tap-major [T][Lpad] Real
    int64_t Lpad, L, M;
    int32_t T;
    uint32_t n_clips, n_channels;
    int64_t ics, ifs, ichs, ocs, ofs, ochs;
    int64_t in_abs0, in_frames;
    int64_t out_k0, out_frames;
    int64_t d0, p0; // out_k0*M = L*d0 + p0
    OutCtx oc;
    int32_t ch_fast; // 1: consecutive threads = consecutive channels of one frame
};

template <typename IO, typename Real>
__global__ void __launch_bounds__(256) k_gather(GatherArgs a)
{
    int64_t idx;
    uint32_t ch, clip;
    if (a.ch_fast) {
        int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
        idx = e / a.n_channels;
        ch = (uint32_t)(e - idx * a.n_channels);
        clip = blockIdx.y;
    } else {
        idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
        ch = blockIdx.y % a.n_channels;
        clip = blockIdx.y / a.n_channels;
    }
    if (idx >= a.out_frames) return;
    // position: (out_k0 + idx)*M = L*d + p
    const int64_t t = a.p0 + idx * a.M;
    const int64_t q = t / a.L;
    const int64_t p = t - q * a.L;
    const int64_t n0 = a.d0 + q - (a.T / 2 - 1);  // absolute index of tap 0's input sample
    const int64_t loc0 = n0 - a.in_abs0;          // its index relative to in[frame 0]
    const IO *xin = (const IO *)a.in + (int64_t)clip * a.ics + (int64_t)ch * a.ichs;
    const Real *c = (const Real *)a.bank + p;
    const int32_t T = a.T, H = T / 2;
    Real accL = 0, accR = 0;
    // The two half-chains are independent: they advance together (each in its own canonical order),
    // eight taps of each per trip, so 32 loads are in flight per lane.  This kernel serves the small
    // launches of streaming calls, where its latency — a serial chain of T dependent FMAs fed by L2
    // loads — is the whole cost (81 us -> ~20 us for T = 736).
    if (loc0 >= 0 && loc0 + T <= a.in_frames) {
        const IO *xp = xin + loc0 * a.ifs;
#pragma unroll 8
        for (int i = 0; i < H; ++i) {
            const int jr = T - 1 - i;
            accL = fma_r(c[(int64_t)i * a.Lpad], (Real)xp[(int64_t)i * a.ifs], accL);
            accR = fma_r(c[(int64_t)jr * a.Lpad], (Real)xp[(int64_t)jr * a.ifs], accR);
        }
    } else {
#pragma unroll 4
        for (int i = 0; i < H; ++i) {
            const int jr = T - 1 - i;
            const int64_t ll = loc0 + i, lr = loc0 + jr;
            const Real xl = (ll >= 0 && ll < a.in_frames) ? (Real)xin[ll * a.ifs] : (Real)0;
            const Real xr = (lr >= 0 && lr < a.in_frames) ? (Real)xin[lr * a.ifs] : (Real)0;
            accL = fma_r(c[(int64_t)i * a.Lpad], xl, accL);
            accR = fma_r(c[(int64_t)jr * a.Lpad], xr, accR);
        }
    }
    IO *yo = (IO *)a.out + (int64_t)clip * a.ocs + idx * a.ofs + (int64_t)ch * a.ochs;
    store_out<Real>(yo, accL + accR, a.oc, ch, a.out_k0 + idx);
}

// ---------------------------------------------------------------------------------------------
// k_wave_dot — the shape BASELINE.json's north star describes, kept as a measured reference point:
// one wavefront per output sample, the taps of the phase spread over the 64 lanes (coefficient row
// and input window both read coalesced), partial dot products combined by a wavefront shuffle
// reduction.  NOT in the canonical order (a 64-way tree instead of two serial chains), so it is
// never chosen automatically and its results are 1e-6-class, not bit-identical.  Measured (DESIGN.md
// §6): far slower than the tiled kernels — six cross-lane steps and two loads per ~4.6 FMAs.
// ---------------------------------------------------------------------------------------------
template <typename IO, typename Real>
__global__ void __launch_bounds__(256) k_wave_dot(GatherArgs a, const Real *__restrict__ phase_major, int32_t per_wave)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t ch = blockIdx.y % a.n_channels, clip = blockIdx.y / a.n_channels;
    const IO *xin = (const IO *)a.in + (int64_t)clip * a.ics + (int64_t)ch * a.ichs;
    IO *yo = (IO *)a.out + (int64_t)clip * a.ocs + (int64_t)ch * a.ochs;
    const int32_t T = a.T, H = T / 2;
    for (int32_t o = 0; o < per_wave; ++o) {
        const int64_t idx = wave * per_wave + o;
        if (idx >= a.out_frames) return; // wave-uniform
        const int64_t t = a.p0 + idx * a.M, q = t / a.L, p = t - q * a.L;
        const int64_t loc0 = a.d0 + q - (H - 1) - a.in_abs0;
        const Real *c = phase_major + p * T;
        Real acc = 0;
        for (int j = lane; j < T; j += 64) {
            const int64_t l = loc0 + j;
            const Real xv = (l >= 0 && l < a.in_frames) ? (Real)xin[l * a.ifs] : (Real)0;
            acc = fma_r(c[j], xv, acc);
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d, 64);
        if (lane == 0) store_out<Real>(yo + idx * a.ofs, acc, a.oc, ch, a.out_k0 + idx);
    }
}

#include "kernels_interp.h" // k_interp, k_gather_wave, k_interp_tile, k_interp_wave
#include "kernels_chain.h"  // k_chain, k_chain_multi, k_chain_resident
#include "kernels_tile.h"   // k_tile, k_tile_mfma, k_tile_mfma_p, k_tile_mfma64_p

// ---------------------------------------------------------------------------------------------
// host side: device tables
// ---------------------------------------------------------------------------------------------
#define HIP_TRY(expr)                                                    \
    do {                                                                 \
        hipError_t e_ = (expr);                                          \
        if (e_ != hipSuccess) return hipGetErrorString(e_);              \
    } while (0)

int device_count()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *ensure_dyn_lds(const void *fn, size_t bytes)
{
    if (bytes <= 64 * 1024) return nullptr;
    static std::mutex mu;
    struct Raised { const void *fn; int dev; size_t bytes; };
    static std::vector<Raised> done; // the limit each (function, device) has been raised to
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    Raised *r = nullptr;
    for (auto &e : done)
        if (e.fn == fn && e.dev == dev) { r = &e; break; }
    if (r && r->bytes >= bytes) return nullptr;
    // (the limit counts against 160 KB together with the kernel's STATIC LDS: raise to what is asked for, not to the maximum)
    HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    if (r) r->bytes = bytes;
    else done.push_back({fn, dev, bytes});
    return nullptr;
}

static inline int32_t floor4(int32_t v) { return v >= 0 ? (v / 4) * 4 : -(((-v) + 3) / 4) * 4; }

struct TileGeom {
    int RT = 16, c = 1;
    int variant = 0; // 0: k_tile (coefficients on the scalar path), 1: k_tile_mfma
    bool aligned = false;
    int32_t n_rt = 0, I_h = 0, pad = 0, i_min = 0, x_count = 0;
    int32_t pb = 64; // periods per slab (k_tile: 32 or 16 when a 64-period slab does not fit LDS — float64, long periods)
    int64_t Lc = 0, Mc = 0;
    size_t lds_bytes = 0;
    int32_t rowR = 0, plane = 0; // variant 2 (k_tile_mfma_p)
    int32_t span = 0;            // variant 2: inputs one period's tiles reach over (i_max - i_min + 1): x_count = (pb - 1) Mc + span
    std::vector<int32_t> e0;
    bool ok = false;
};

static inline int32_t floor16(int32_t v) { return v >= 0 ? (v / 16) * 16 : -(((-v) + 15) / 16) * 16; }

// Geometry + coefficient table of k_tile_mfma_p (f32 engine, Mc % 16 == 0).
// Table: [n_rt][2][n_groups + 4][64 lanes][4 chunks]; lane (row j = l & 15, k = l >> 4), chunk c:
//   left : C'[row][i0L + 16*grp + 4*c + k]        right: C'[row][i1R - (16*grp + 4*c + k)]
// Real = double (k_tile_mfma64_p): the same table in float64, a slab of 32 periods (pb) instead of 64 — 8 bytes per
// sample — and plane rows of R doubles with R == 2 (mod 4): the 16 rows of a ds_read_b128 group then start in 16
// different 16-byte bank groups.
template <typename Real>
static TileGeom build_mfma_planes(const Plan &p, std::vector<Real> *tab)
{
    TileGeom g;
    g.variant = 2;
    g.pb = sizeof(Real) == 4 ? 64 : 32;
    const int64_t L = p.L, M = p.M;
    const int32_t T = p.T, H = T / 2;
    g.RT = 16;
    int c = 1;
    while (L * c < g.RT && c < 64) c *= 2;
    while ((M * c) % 16 != 0 && M * c * 2 <= 512 && c < 64) c *= 2;
    g.c = c; g.Lc = L * c; g.Mc = M * c;
    if (g.Mc % 16 != 0 || g.Mc > 4096 || g.Lc > 16384) return g;
    const int32_t Mc = (int32_t)g.Mc;
    g.n_rt = (int32_t)((g.Lc + 15) / 16);
    auto n_of = [&](int64_t r) { return (int32_t)((r * M) / L) - (H - 1); };
    auto p_of = [&](int64_t r) { return (r * M) % L; };
    std::vector<int32_t> i0L(g.n_rt), i1R(g.n_rt);
    int32_t I_h = 0;
    for (int rt = 0; rt < g.n_rt; ++rt) {
        int64_t r0 = (int64_t)rt * 16, r1 = std::min<int64_t>(r0 + 16, g.Lc) - 1;
        i0L[rt] = floor16(n_of(r0));
        i1R[rt] = floor16(n_of(r1) + T - 1) + 15;
        I_h = std::max(I_h, std::max(n_of(r1) + H - i0L[rt], i1R[rt] - (n_of(r0) + H) + 1));
    }
    I_h = (I_h + 15) / 16 * 16;
    g.I_h = I_h;
    int32_t i_min = INT32_MAX, i_max = INT32_MIN;
    for (int rt = 0; rt < g.n_rt; ++rt) { // +-16: the B operand of one group past either end is read too? no: only A is prefetched
        i_min = std::min(i_min, std::min(i0L[rt], i1R[rt] - I_h + 1));
        i_max = std::max(i_max, std::max(i0L[rt] + I_h - 1, i1R[rt]));
    }
    g.i_min = i_min; // multiples of 16 by construction
    g.span = i_max - i_min + 1;
    g.x_count = (g.pb - 1) * Mc + g.span;
    g.rowR = Mc / 4;
    if (sizeof(Real) == 4) while ((g.rowR % 8) != 4) ++g.rowR; // R/4 odd -> conflict-free ds_read_b128 across the 16 periods
    else while ((g.rowR % 4) != 2) ++g.rowR;                    // float64: R/2 odd
    g.pad = g.rowR - Mc / 4;
    const int32_t rows_total = (g.x_count + Mc - 1) / Mc + 3; // + slack rows: pipelined reads overrun by one group
    g.plane = (rows_total * g.rowR + 63) / 64 * 64;
    g.lds_bytes = ((size_t)g.plane * 4 + g.rowR) * sizeof(Real);
    // Groups a tile's half-chain really needs (bits 24..31 of its e0 word): the table rows are I_h long for every
    // tile — the longest span over all tiles, rounded to 16, from a start rounded down to 16 — but the groups past a
    // tile's own last tap hold only zero coefficients, and fma(0, x, acc) == acc: they are not issued (10-11 of 12
    // groups at 48k -> 44.1k VHQ).
    g.e0.resize((size_t)g.n_rt * 2);
    for (int rt = 0; rt < g.n_rt; ++rt) {
        const int64_t r0 = (int64_t)rt * 16, r1 = std::min<int64_t>(r0 + 16, g.Lc) - 1;
        const int32_t gl = (n_of(r1) + H - i0L[rt] + 15) / 16, gr = (i1R[rt] - (n_of(r0) + H) + 1 + 15) / 16;
        g.e0[rt * 2 + 0] = (i0L[rt] - i_min) | (std::min(gl, I_h / 16) << 24);
        g.e0[rt * 2 + 1] = (i1R[rt] - 15 - i_min) | (std::min(gr, I_h / 16) << 24);
    }
    if (g.lds_bytes > 160 * 1024 || g.x_count + I_h >= (1 << 24) || I_h / 16 > 127) return g;
    g.ok = true;
    if (tab) {
        const int ng = I_h / 16;
        tab->assign((size_t)g.n_rt * 2 * (ng + 4) * 256, (Real)0);
        for (int rt = 0; rt < g.n_rt; ++rt)
            for (int rr = 0; rr < 16; ++rr) {
                int64_t r = (int64_t)rt * 16 + rr;
                if (r >= g.Lc) continue;
                const int32_t nr = n_of(r);
                const double *cp = p.bank.data() + (size_t)(p_of(r) * T);
                for (int ii = 0; ii < I_h; ++ii) {
                    const int grp = ii / 16, cc = (ii % 16) / 4, k = ii % 4, lane = k * 16 + rr;
                    const size_t at = ((size_t)grp * 64 + lane) * 4 + cc;
                    int32_t jl = i0L[rt] + ii - nr;
                    if (jl >= 0 && jl < H) (*tab)[(size_t)(rt * 2 + 0) * (ng + 4) * 256 + at] = (Real)cp[jl];
                    int32_t jr = i1R[rt] - ii - nr;
                    if (jr >= H && jr < T) (*tab)[(size_t)(rt * 2 + 1) * (ng + 4) * 256 + at] = (Real)cp[jr];
                }
            }
    }
    return g;
}

// Tile geometry + (optionally) tables for one precision.
template <typename Real>
static TileGeom build_tile_tables(const Plan &p, std::vector<Real> *tab, int variant = 0)
{
    TileGeom g;
    g.variant = variant;
    const int64_t L = p.L, M = p.M;
    const int32_t T = p.T, H = T / 2;
    g.RT = 16;
    // replicate short periods so that a period holds at least one full tile
    int c = 1;
    while (L * c < g.RT && c < 64) c *= 2;
    // prefer an input period that is a multiple of 4 (b128 LDS reads) when the slab stays small
    if (variant == 0 && (M % 2 == 0 || M * 4 <= 256))
        while ((M * c) % 4 != 0 && M * c * 2 <= 256 && c < 64) c *= 2;
    g.c = c;
    g.Lc = L * c; g.Mc = M * c;
    if (g.Mc > 8192 || g.Lc > 16384) return g; // period too long for an LDS-resident slab
    g.aligned = variant == 0 && (g.Mc % 4 == 0);
    const int32_t Mc = (int32_t)g.Mc;
    if (variant == 1) {
        // k_tile_mfma: a 32-lane half reads x[(16 periods j)*S + (2 inputs k)] with ds_read_b32;
        // conflict-free iff the row stride S = Mc + pad is 2*odd (mod 32).
        g.pad = 0;
        while (((Mc + g.pad) % 4) != 2) ++g.pad;
        // Odd periods (441, 147 ...) run UNPADDED (round 3): a lane's LDS offset is then just its input index — no
        // period-boundary test and no second offset in the chain's inner step (five vector-ALU instructions per chunk
        // fewer, in a loop that is bound by exactly those) — at the price of two-way conflicts on about half the
        // banks of each B read (row stride odd: the sixteen periods start in sixteen different banks, their second
        // input collides with a neighbour's first).
        if (Mc % 2 == 1 && !switches().dbg_pad) g.pad = 0;
    } else if (g.aligned) { // row stride = 4*odd words -> conflict-free ds_read_b128 across lanes
        g.pad = ((Mc / 4) % 2 == 0) ? 4 : 0;
    } else {         // row stride odd -> conflict-free ds_read_b32
        g.pad = (Mc % 2 == 0) ? 1 : 0;
    }
    g.n_rt = (int32_t)((g.Lc + g.RT - 1) / g.RT);
    auto n_of = [&](int64_t r) { return (int32_t)((r * M) / L) - (H - 1); };
    auto p_of = [&](int64_t r) { return (r * M) % L; };
    std::vector<int32_t> i0L(g.n_rt), i1R(g.n_rt);
    int32_t I_h = 0;
    for (int rt = 0; rt < g.n_rt; ++rt) {
        int64_t r0 = (int64_t)rt * g.RT, r1 = std::min<int64_t>(r0 + g.RT, g.Lc) - 1;
        int32_t a0 = n_of(r0), a1 = n_of(r1) + T - 1;
        if (g.aligned) {
            a0 = floor4(a0);
            a1 = floor4(a1) + 3; // smallest value >= a1 that is == 3 (mod 4)
        }
        i0L[rt] = a0; i1R[rt] = a1;
        int32_t IL = n_of(r1) + H - a0;        // inputs a0 .. n_r1+H-1
        int32_t IR = a1 - (n_of(r0) + H) + 1;  // inputs n_r0+H .. a1
        I_h = std::max(I_h, std::max(IL, IR));
    }
    I_h = variant == 1 ? (I_h + 15) / 16 * 16 : (I_h + 3) / 4 * 4; // k_tile_mfma works in groups of 4 chunks
    g.I_h = I_h;
    int32_t i_min = INT32_MAX, i_max = INT32_MIN;
    for (int rt = 0; rt < g.n_rt; ++rt) {
        i_min = std::min(i_min, std::min(i0L[rt], i1R[rt] - I_h + 1));
        i_max = std::max(i_max, std::max(i0L[rt] + I_h - 1, i1R[rt]));
    }
    i_min = floor4(i_min); i_max = floor4(i_max) + 3; // slab = whole quads (vectorised staging)
    g.i_min = i_min;
    g.span = i_max - i_min + 1;
    // periods per slab: 64 (one per lane); the VALU kernel also runs with 32 or 16 (the other lanes idle) when the
    // slab would not fit — float64 at 44.1k -> 16k: 64 x 441 x 8 B = 226 KB — which still beats one lane per output
    // walking T dependent loads by 5x (60 s mono int32: 1015 us on k_gather)
    g.pb = 64;
    for (;;) {
        g.x_count = ((g.pb - 1) * Mc + (i_max - i_min + 1) + 3) / 4 * 4; // whole quads ((pb-1)*Mc may be odd)
        g.lds_bytes = ((size_t)g.x_count + (size_t)g.pad * (g.x_count / Mc + 1) + 8) * sizeof(Real);
        // (variant 1 in float64 — k_tile_mfma<IO, double, NG> — runs NG = pb / 16 groups of 16 periods: 4, 2 or 1)
        // Slab of the float64 MFMA kernel: a 64-period slab only when small (two or more workgroups per CU must fit: a lone
        // workgroup cannot hide its own staging — 48k -> 44.1k int32 60 s: 105 us on 83 KB slabs, 90 us on 42 KB ones),
        // else 32 periods up to 120 KB (44.1k -> 16k: 72 us on 116 KB against 84 us on 59 KB), else 16.
        const size_t limit = (variant == 1 && sizeof(Real) == 8)
                                 ? (g.pb == 64 ? 0 /* (four period groups of float64: 64 accumulator registers — the pipelined chain no longer fits; two groups it is) */
                                               : switches().dbg_mfma64_lds ? switches().dbg_mfma64_lds : 120 * 1024) : 160 * 1024;
        if (g.lds_bytes <= limit || (variant != 0 && sizeof(Real) == 4) || g.pb == 16) break;
        g.pb /= 2;
    }
    g.e0.resize((size_t)g.n_rt * 2);
    for (int rt = 0; rt < g.n_rt; ++rt) {
        g.e0[rt * 2 + 0] = i0L[rt] - i_min;
        g.e0[rt * 2 + 1] = i1R[rt] - 3 - i_min;
    }
    if (g.lds_bytes > 160 * 1024) return g;
    g.ok = true;
    if (tab) {
        const size_t rows = (size_t)I_h + (variant == 1 ? 16 : 0); // prefetch slack (zeros)
        tab->assign((size_t)g.n_rt * 2 * rows * g.RT, (Real)0);
        for (int rt = 0; rt < g.n_rt; ++rt) {
            Real *tl = tab->data() + (size_t)(rt * 2 + 0) * rows * g.RT;
            Real *tr = tab->data() + (size_t)(rt * 2 + 1) * rows * g.RT;
            for (int rr = 0; rr < g.RT; ++rr) {
                int64_t r = (int64_t)rt * g.RT + rr;
                if (r >= g.Lc) continue;
                const int32_t nr = n_of(r);
                const double *cp = p.bank.data() + (size_t)(p_of(r) * T);
                for (int ii = 0; ii < I_h; ++ii) {
                    int32_t jl = i0L[rt] + ii - nr;
                    if (jl >= 0 && jl < H) tl[(size_t)ii * g.RT + rr] = (Real)cp[jl];
                    int32_t jr = i1R[rt] - ii - nr;
                    if (jr >= H && jr < T) tr[(size_t)ii * g.RT + rr] = (Real)cp[jr];
                }
            }
        }
    }
    return g;
}

template <typename Real>
static const char *bank_upload(Plan *p, DeviceBank &d, TileGeom *geom_out, TileGeom *geom_m_out)
{
    const int64_t L = p->L;
    const int32_t T = p->T;
    if (p->phases) { // interpolated-phase plan: the cubic table in the engine precision, nothing else
        std::vector<Real> tb(p->bank.size());
        for (size_t i = 0; i < tb.size(); ++i) tb[i] = (Real)p->bank[i];
        HIP_TRY(hipMalloc(&d.interp_tab, tb.size() * sizeof(Real)));
        HIP_TRY(hipMemcpy(d.interp_tab, tb.data(), tb.size() * sizeof(Real), hipMemcpyHostToDevice));
        return nullptr;
    }
    d.Lpad = (L + 15) / 16 * 16;
    std::vector<Real> tm((size_t)T * d.Lpad, (Real)0);
    for (int64_t ph = 0; ph < L; ++ph)
        for (int j = 0; j < T; ++j) tm[(size_t)j * d.Lpad + ph] = (Real)p->bank[(size_t)(ph * T + j)];
    HIP_TRY(hipMalloc(&d.tap_major, tm.size() * sizeof(Real)));
    HIP_TRY(hipMemcpy(d.tap_major, tm.data(), tm.size() * sizeof(Real), hipMemcpyHostToDevice));

    std::vector<Real> tab;
    TileGeom g = build_tile_tables<Real>(*p, &tab);
    if (g.ok) {
        HIP_TRY(hipMalloc(&d.tile_tab, tab.size() * sizeof(Real)));
        HIP_TRY(hipMemcpy(d.tile_tab, tab.data(), tab.size() * sizeof(Real), hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc((void **)&d.tile_i0, g.e0.size() * sizeof(int32_t)));
        HIP_TRY(hipMemcpy(d.tile_i0, g.e0.data(), g.e0.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        d.RT = g.RT; d.n_rt = g.n_rt; d.I_h = g.I_h;
    }
    *geom_out = g;
    if (sizeof(Real) == 8 && geom_m_out && !switches().no_mfma64) { // float64 engine on v_mfma_f64_16x16x4_f64 (k_tile_mfma<IO, double, NG>)
        std::vector<Real> tabm;
        TileGeom gm = build_mfma_planes<Real>(*p, &tabm); // k_tile_mfma64_p where the period admits planes, else k_tile_mfma<IO, double, NG>
        if (!gm.ok || switches().no_planes) gm = build_tile_tables<Real>(*p, &tabm, 1);
        if (gm.ok) {
            HIP_TRY(hipMalloc(&d.tile_tab_m, tabm.size() * sizeof(Real)));
            HIP_TRY(hipMemcpy(d.tile_tab_m, tabm.data(), tabm.size() * sizeof(Real), hipMemcpyHostToDevice));
            HIP_TRY(hipMalloc((void **)&d.tile_i0_m, gm.e0.size() * sizeof(int32_t)));
            HIP_TRY(hipMemcpy(d.tile_i0_m, gm.e0.data(), gm.e0.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        }
        *geom_m_out = gm;
    }
    if (sizeof(Real) == 4 && geom_m_out) {
        std::vector<float> tabm;
        TileGeom gm = build_mfma_planes<float>(*p, &tabm);
        if (!gm.ok || switches().no_planes) gm = build_tile_tables<float>(*p, &tabm, 1);
        if (gm.ok) {
            HIP_TRY(hipMalloc(&d.tile_tab_m, tabm.size() * sizeof(Real)));
            HIP_TRY(hipMemcpy(d.tile_tab_m, tabm.data(), tabm.size() * sizeof(Real), hipMemcpyHostToDevice));
            HIP_TRY(hipMalloc((void **)&d.tile_i0_m, gm.e0.size() * sizeof(int32_t)));
            HIP_TRY(hipMemcpy(d.tile_i0_m, gm.e0.data(), gm.e0.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        }
        *geom_m_out = gm;
    }
    return nullptr;
}

// geometry is cheap to recompute; keep it beside the bank in a side table keyed by (plan, prec)
static std::mutex g_geom_mu;
static std::vector<std::pair<std::pair<const Plan *, int>, TileGeom>> g_geoms; // key: (plan, prec*2+variant)

static TileGeom *geom_find(const Plan *p, int prec, int variant)
{
    for (auto &e : g_geoms)
        if (e.first.first == p && e.first.second == prec * 2 + variant) return &e.second;
    return nullptr;
}

const char *device_bank_ensure(Plan *p, int prec)
{
    std::lock_guard<std::mutex> lk(p->mu);
    DeviceBank &d = p->dev[prec];
    int cur = -1;
    if (p->device >= 0 && hipGetDevice(&cur) == hipSuccess && cur != p->device)
        return "this plan's device tables live on another device (one plan per device)";
    if (d.ready) return nullptr;
    if (device_count() <= 0) return "no HIP device available (hipsoxr has no CPU fallback)";
    if (p->device < 0 && hipGetDevice(&cur) == hipSuccess) p->device = cur; // tables are built on first use, here
    TileGeom g, gm;
    const char *e = prec == 0 ? bank_upload<float>(p, d, &g, &gm) : bank_upload<double>(p, d, &g, &gm);
    if (e) return e;
    {
        std::lock_guard<std::mutex> lk2(g_geom_mu);
        g_geoms.push_back({{p, prec * 2 + 0}, g});
        g_geoms.push_back({{p, prec * 2 + 1}, gm});
    }
    d.ready = true;
    return nullptr;
}

void device_bank_release(Plan *p)
{
    for (int i = 0; i < 2; ++i) {
        DeviceBank &d = p->dev[i];
        if (d.tap_major) (void)hipFree(d.tap_major);
        if (d.phase_major) (void)hipFree(d.phase_major);
        if (d.interp_tab) (void)hipFree(d.interp_tab);
        if (d.tile_tab) (void)hipFree(d.tile_tab);
        if (d.tile_i0) (void)hipFree(d.tile_i0);
        if (d.tile_tab_m) (void)hipFree(d.tile_tab_m);
        if (d.tile_i0_m) (void)hipFree(d.tile_i0_m);
        d = DeviceBank();
    }
    std::lock_guard<std::mutex> lk2(g_geom_mu);
    for (size_t i = 0; i < g_geoms.size();)
        if (g_geoms[i].first.first == p) g_geoms.erase(g_geoms.begin() + i);
        else ++i;
}

// ---------------------------------------------------------------------------------------------
// launch
// ---------------------------------------------------------------------------------------------
template <typename IO, typename Real>
static const char *launch_gather(Plan *p, const hipsoxr_job_t &j, hipStream_t st, const VrPos *vr = nullptr, ResidentLaunch *res = nullptr,
                                 ChainDone *cd = nullptr)
{
    if (cd) cd->n_wgs = 0;
    const DeviceBank &d = p->dev[sizeof(Real) == 4 ? 0 : 1];
    // split so that idx*M stays far below 2^63 and grid.x below 2^31
    const int64_t max_chunk = (int64_t)1 << 30;
    for (int64_t done = 0; done < j.out_frames; done += max_chunk) {
        GatherArgs a;
        const int64_t k0 = j.out_k0 + done;
        const int64_t nf = std::min<int64_t>(max_chunk, j.out_frames - done);
        a.in = j.in;
        a.out = (char *)j.out + (size_t)(done * j.out_frame_stride) * sizeof(IO);
        a.bank = d.tap_major; a.Lpad = d.Lpad; a.L = p->L; a.M = p->M; a.T = p->T;
        a.n_clips = j.n_clips; a.n_channels = j.n_channels;
        a.ics = j.in_clip_stride; a.ifs = j.in_frame_stride; a.ichs = j.in_chan_stride;
        a.ocs = j.out_clip_stride; a.ofs = j.out_frame_stride; a.ochs = j.out_chan_stride;
        a.in_abs0 = j.in_abs0; a.in_frames = j.in_frames;
        a.out_k0 = k0; a.out_frames = nf;
        __int128 kM = (__int128)k0 * p->M;
        a.d0 = (int64_t)(kM / p->L); a.p0 = (int64_t)(kM % p->L);
        a.oc.clip_counter = j.clip_counter; a.oc.dither = j.dither; a.oc.seed = j.dither_seed; a.oc.ch0 = t_ch_base;
        a.ch_fast = (j.n_channels > 1 && j.in_chan_stride == 1) ? 1 : 0;
        dim3 grid, block(256);
        if (a.ch_fast) {
            int64_t e = nf * (int64_t)j.n_channels;
            grid = dim3((unsigned)((e + 255) / 256), j.n_clips, 1);
            if (j.n_clips > 65535) return "too many clips for one launch (max 65535)";
        } else {
            uint64_t cols = (uint64_t)j.n_clips * j.n_channels;
            if (cols > 65535) {
                // fold: launch per clip group
                return "too many (clip, channel) columns for one launch (max 65535)";
            }
            grid = dim3((unsigned)((nf + 255) / 256), (unsigned)cols, 1);
        }
        // k_interp_wave (in place of lane-per-output k_interp): needs a (clip, channel) grid dimension and LDS for the
        // span of 32 consecutive outputs at the launch's largest step
        constexpr double kWaveUsPerTap = 1.0e-6; // us per output x tap, 256 CUs
        bool wave_ok = false;
        int64_t wave_span = 0;
        size_t wave_lds = 0;
        if (p->phases && !switches().no_interp_wave && (uint64_t)j.n_clips * j.n_channels <= 65535) {
            double step = (double)p->M / (double)p->L;
            if (vr) {
                const double two64 = 18446744073709551616.;
                const double s0 = (double)vr->s_hi + (double)vr->s_lo / two64;
                const double dd = (double)(int64_t)vr->d_hi + (double)vr->d_lo / two64;
                step = std::max(s0, s0 + dd * (double)(done + nf)) * (1. + 1e-9);
            }
            if (step < 1e6) {
                wave_span = ((int64_t)std::ceil(31. * step) + p->T + 4 + 3) & ~(int64_t)3;
                wave_lds = (size_t)(wave_span + 32) * sizeof(Real);
                wave_ok = wave_lds <= 64 * 1024;
            }
        }
        // exact-bank launches of 4096 outputs and more that come here (launch_typed: periods too few for a slab, or a
        // stream chunk whose result goes straight to host memory): k_gather_wave instead of lane-per-output k_gather
        if (!p->phases && !res && !vr && !switches().no_gather_wave && nf >= 4096 && p->T >= 32 && (uint64_t)j.n_clips * j.n_channels <= 65535) {
            const int64_t gspan = ((int64_t)(31 * ((p->M + p->L - 1) / p->L + 1)) + p->T + 4 + 3) & ~(int64_t)3; // 31 window shifts of at most ceil(M/L) + T
            const size_t glds = (size_t)(gspan + 32) * sizeof(Real);
            if (glds <= 64 * 1024) {
                DeviceBank &dm = p->dev[sizeof(Real) == 4 ? 0 : 1];
                const char *err = nullptr;
                {
                    std::lock_guard<std::mutex> lk(p->mu);
                    if (!dm.phase_major) {
                        std::vector<Real> pm(p->bank.size());
                        for (size_t i = 0; i < pm.size(); ++i) pm[i] = (Real)p->bank[i];
                        if (hipMalloc(&dm.phase_major, pm.size() * sizeof(Real)) != hipSuccess) err = "hipMalloc failed";
                        else if (hipMemcpy(dm.phase_major, pm.data(), pm.size() * sizeof(Real), hipMemcpyHostToDevice) != hipSuccess)
                            err = "hipMemcpy failed";
                    }
                }
                if (err) return err;
                GatherWaveArgs ga;
                ga.g = a; ga.phase_major = dm.phase_major; ga.span_cap = (int32_t)gspan; ga.done_words = nullptr; ga.done_seq = 0;
                const uint64_t wgs = (uint64_t)((nf + 31) / 32) * ((uint64_t)j.n_clips * j.n_channels);
                if (cd && done == 0 && nf == j.out_frames && wgs <= cd->cap) { // the whole job is this launch
                    ga.done_words = cd->words; ga.done_seq = cd->seq;
                    cd->n_wgs = (uint32_t)wgs;
                }
                hipLaunchKernelGGL((k_gather_wave<IO, Real>), dim3((unsigned)((nf + 31) / 32), (unsigned)((uint64_t)j.n_clips * j.n_channels), 1),
                                   dim3(256), glds, st, ga);
                HIP_TRY(hipGetLastError());
                continue;
            }
        }
        // small launches (streaming chunks): the low-latency chain kernel (interpolated plans above 512 outputs: k_interp_wave —
        // 4410-frame variable-rate calls 29.0 -> 27.0 us; 441-frame calls are 2.5 us faster here: 20 short workgroups against 5)
        const bool no_chain = switches().no_chain;
        if (!no_chain && nf < 4096 && (uint64_t)j.n_clips * j.n_channels <= 65535 && !(wave_ok && !res && nf > 512)) {
            // few outputs per workgroup: the staging loop is then two or three trips of 16 loads per
            // thread (its latency is the kernel's latency), and there are enough workgroups anyway
            // (above 512 outputs 32 per workgroup: at most 64 workgroups then read their span over PCIe, poll the mailbox
            //  of the resident form, or report through completion words — 4410-frame chunks 25.3 -> 22.7 us per call)
            int NO = nf <= 512 ? 8 : 32;
            if (switches().dbg_chain_no) NO = switches().dbg_chain_no;
            // LDS: NO coefficient rows of T + V words, the shared input span (T + what NO-1 window shifts of at
            // most ceil(M/L) + 1 samples add; variable rate: the plan's ratio is the largest step), bookkeeping
            const int64_t shift = (p->M + p->L - 1) / p->L + 2;
            auto chain_lds = [&](int no, int32_t *span_cap) {
                const int64_t sc = (int64_t)p->T + (int64_t)no * shift + 4;
                *span_cap = (int32_t)sc;
                return (size_t)no * (p->T + 16 / sizeof(Real)) * sizeof(Real) + (size_t)((sc + 3) & ~3) * sizeof(Real) + (size_t)no * 16;
            };
            int32_t span_cap = 0;
            while (NO > 2 && chain_lds(NO, &span_cap) > 150 * 1024) NO /= 2;
            const size_t lds = chain_lds(NO, &span_cap);
            if (lds <= 150 * 1024 && shift < (1 << 20)) {
                ChainArgs ca;
                std::memset(&ca, 0, sizeof ca);
                ca.ia.g = a; ca.NO = NO; ca.span_cap = span_cap;
                const char *err = nullptr;
                void (*ck)(ChainArgs) = nullptr;
                if (p->phases) {
                    ca.ia.tab = d.interp_tab; ca.ia.P = p->phases;
                    while ((1 << ca.ia.lgP) < ca.ia.P) ++ca.ia.lgP;
                    if (vr) {
                        if ((1 << ca.ia.lgP) != ca.ia.P) return "variable-rate needs a power-of-two phase count";
                        typedef unsigned __int128 u128;
                        const u128 T0 = ((u128)vr->t_hi << 64) | vr->t_lo, S0 = ((u128)vr->s_hi << 64) | vr->s_lo,
                                   D = ((u128)vr->d_hi << 64) | vr->d_lo;
                        const u128 n = (u128)(uint64_t)done, m = n * (n - 1) / 2;
                        const u128 T1 = T0 + n * S0 + D * (done ? m : 0), S1 = S0 + D * n;
                        ca.ia.t_hi = (uint64_t)(T1 >> 64); ca.ia.t_lo = (uint64_t)T1;
                        ca.ia.s_hi = (uint64_t)(S1 >> 64); ca.ia.s_lo = (uint64_t)S1;
                        ca.ia.d_hi = vr->d_hi; ca.ia.d_lo = vr->d_lo;
                        ck = k_chain<IO, Real, 2>;
                    } else {
                        ck = k_chain<IO, Real, 1>;
                    }
                } else {
                    DeviceBank &dm = p->dev[sizeof(Real) == 4 ? 0 : 1];
                    {
                        std::lock_guard<std::mutex> lk(p->mu);
                        if (!dm.phase_major) {
                            std::vector<Real> pm(p->bank.size());
                            for (size_t i = 0; i < pm.size(); ++i) pm[i] = (Real)p->bank[i];
                            if (hipMalloc(&dm.phase_major, pm.size() * sizeof(Real)) != hipSuccess) err = "hipMalloc failed";
                            else if (hipMemcpy(dm.phase_major, pm.data(), pm.size() * sizeof(Real), hipMemcpyHostToDevice) != hipSuccess)
                                err = "hipMemcpy failed";
                        }
                    }
                    if (err) return err;
                    ca.phase_major = dm.phase_major;
                    ck = k_chain<IO, Real, 0>;
                }
                if (res) { // the resident form: same staging, same chains, fed by messages (k_chain_resident)
                    if (p->L >= (1 << 24) && !vr) return "resident kernel: ratio numerator too large";
                    void (*rk)(ResidentArgs) = vr ? k_chain_resident<IO, Real, 2> : p->phases ? k_chain_resident<IO, Real, 1> : k_chain_resident<IO, Real, 0>;
                    const unsigned gx = (unsigned)((nf + NO - 1) / NO), gy = (unsigned)((uint64_t)j.n_clips * j.n_channels);
                    // every workgroup must be on the chip at once (they wait for each other): a quarter of the slots at most
                    int occ = 0, dev = 0, cus = 0;
                    if (const char *e = ensure_dyn_lds((const void *)rk, lds)) return e;
                    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)rk, 256, lds));
                    HIP_TRY(hipGetDevice(&dev));
                    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
                    if (occ < 1 || (int64_t)gx * gy > (int64_t)occ * cus / 4 || (int64_t)gx * gy > (int64_t)kResidentMaxWgs) return "resident kernel: message too large";
                    // process-wide budget in CU capacity (engine.cpp): all resident instances together hold at most half the chip
                    res->cost_mcu = (uint32_t)(((int64_t)gx * gy * 1024 + occ - 1) / occ);
                    if (res->used_mcu + (int64_t)res->cost_mcu > ((int64_t)cus * 1024) >> res->budget_shift) { res->over_budget = true; return "resident kernel: over the budget"; }
                    ResidentArgs ra;
                    std::memset(&ra, 0, sizeof ra);
                    ra.ca = ca; ra.box = res->box; ra.words = res->words; ra.ctl = res->ctl; ra.base_seq = res->base_seq; ra.epoch = res->epoch;
                    ra.idle_ticks = res->idle_us * 100; // wall_clock64: 100 MHz
                    ra.n_wgs = gx * gy;
                    res->n_wgs = ra.n_wgs; res->max_out = (int64_t)gx * NO;
                    hipLaunchKernelGGL(rk, dim3(gx, gy, 1), dim3(256), lds, st, ra);
                    HIP_TRY(hipGetLastError());
                    return nullptr;
                }
                if (const char *e = ensure_dyn_lds((const void *)ck, lds)) return e;
                {
                    const uint64_t wgs = (uint64_t)((nf + NO - 1) / NO) * ((uint64_t)j.n_clips * j.n_channels);
                    if (cd && done == 0 && nf == j.out_frames && wgs <= cd->cap) { // the whole job is this launch
                        ca.done_words = cd->words; ca.done_seq = cd->seq;
                        cd->n_wgs = (uint32_t)wgs;
                    }
                }
                hipLaunchKernelGGL(ck, dim3((unsigned)((nf + NO - 1) / NO), (unsigned)((uint64_t)j.n_clips * j.n_channels), 1),
                                   dim3(256), lds, st, ca);
                HIP_TRY(hipGetLastError());
                continue;
            }
        }
        if (res) return "resident kernel: unavailable for this job";
        if (p->phases) {
            InterpArgs ia;
            std::memset(&ia, 0, sizeof ia);
            ia.g = a; ia.tab = d.interp_tab; ia.P = p->phases;
            while ((1 << ia.lgP) < ia.P) ++ia.lgP;
            // throughput kernel for large launches: KO outputs per workgroup, as many as LDS allows
            // (input span + 2 bytes of bookkeeping per output), at least ~32 outputs per interval
            const bool no_itile = switches().no_interp_tile;
            int64_t KO = 0, span_cap = 0;
            int pair_mode = 0;          // k_interp_tile: 0 one output per lane, 1 channel pairs, 2 the column's two halves
            bool twin = false;          // ... pairs with the span staged twice (float: 16-byte aligned reads)
            int64_t split_h = 0, nf_t = nf; // (outputs the tiles are counted over: member 1's)
            // position of this launch's first output on the variable-rate clock: (T0, S0) advanced by `done` outputs
            auto vr_advance = [&](InterpArgs &x) {
                typedef unsigned __int128 u128;
                const u128 T0 = ((u128)vr->t_hi << 64) | vr->t_lo, S0 = ((u128)vr->s_hi << 64) | vr->s_lo,
                           D = ((u128)vr->d_hi << 64) | vr->d_lo;
                const u128 n = (u128)(uint64_t)done, m = n * (n - 1) / 2;
                const u128 T1 = T0 + n * S0 + D * (done ? m : 0), S1 = S0 + D * n;
                x.t_hi = (uint64_t)(T1 >> 64); x.t_lo = (uint64_t)T1;
                x.s_hi = (uint64_t)(S1 >> 64); x.s_lo = (uint64_t)S1;
                x.d_hi = vr->d_hi; x.d_lo = vr->d_lo;
            };
            if (vr && (1 << ia.lgP) != ia.P) return "variable-rate needs a power-of-two phase count";
            if (!no_itile && nf >= 4096 && (uint64_t)j.n_clips * j.n_channels <= 65535) {
                double step = (double)p->M / (double)p->L; // input samples per output
                if (vr) {
                    const double two64 = 18446744073709551616.;
                    const double s0 = (double)vr->s_hi + (double)vr->s_lo / two64;
                    const double dd = (double)(int64_t)vr->d_hi + (double)vr->d_lo / two64;
                    step = std::max(s0, s0 + dd * (double)(done + nf)) * (1. + 1e-9);
                }
                // a bucket (outputs of one interval) is served 64 at a time: aim at a mean of 60 per
                // interval (30, 15 when LDS cannot hold that many outputs and their input span)
                // (round 3: among the sizes that fit, the one that leaves the fewest workgroup-layers x outputs per
                //  workgroup on the 256 CUs — 48000 -> 44101 stereo 60 s: 60 per interval are 346 workgroups, two layers
                //  of which the second is a third full; 41 per interval are 506)
                // two outputs per lane (InterpTileArgs): neighbouring channels of an even channel count, else — constant rate —
                // the column's own second half, split h periods of L outputs in when that half has >= 0.7 of the first's outputs
                // — taken when the launch's workgroup layers x outputs per workgroup come out cheaper than with one output
                // per lane (a pair workgroup takes ~1.7x a single one: 60 s stereo 393 -> 348 us, 8 channels 1622 -> 1120,
                // mono 232 -> 190; a 10 s stereo job has too few workgroups to halve them)
                constexpr double kPairWg = 1.7, kTwinWg = 1.4; // (... 1.4 with the span staged twice for 16-byte reads: stereo 348 -> 296, mono 190 -> 145)
                int cand_mode = 0;
                int64_t cand_h = 0, cand_nf = nf;
                if (!switches().no_interp_pair) {
                    if (j.n_channels % 2 == 0) cand_mode = 1;
                    else if (!vr) {
                        const int64_t h = (nf + 2 * p->L - 1) / (2 * p->L), n1 = h * p->L;
                        if (h >= 1 && n1 < nf && 10 * (nf - n1) >= 7 * n1 && h * p->M < ((int64_t)1 << 40)) { cand_mode = 2; cand_h = h; cand_nf = n1; }
                    }
                }
                double best_cost = 1e300, cols_ = (double)j.n_clips * j.n_channels;
                for (int mode : {0, cand_mode}) {
                    if (mode == 0 && cand_mode && switches().dbg_interp_pair_always) continue;
                    const int nm = mode ? 2 : 1;
                    const int64_t nft = mode == 2 ? cand_nf : nf;
                    const double cols_m = (double)j.n_clips * j.n_channels / (mode == 1 ? 2 : 1);
                    for (int tw = 0; tw <= (mode && sizeof(Real) == 4 && !switches().dbg_interp_no_twin ? 1 : 0); ++tw) // float pairs: one or two copies of the span
                        for (int per = 64; per >= 15; --per) {
                            const int64_t k = (int64_t)per * p->phases;
                            if (k > 16384 || k > nft) continue;
                            const int64_t sc = (int64_t)std::ceil((double)k * step) + p->T + 8;
                            const int64_t bytes = (tw ? 2 * (sc + 4) : sc) * nm * (int64_t)sizeof(Real) + k * 2 + (2 * p->phases + 2) * 4 + 64;
                            if (bytes > 150 * 1024) continue;
                            const double wgs_ = std::ceil((double)nft / (double)k) * cols_m;
                            const double cost = std::ceil(wgs_ / 256.) * (double)k * (per >= 30 ? 1. : 30. / per) * (tw ? kTwinWg : mode ? kPairWg : 1.); // (thin buckets: idle lanes)
                            if (cost < best_cost) { best_cost = cost; KO = k; span_cap = sc; pair_mode = mode; twin = tw != 0; split_h = mode == 2 ? cand_h : 0; nf_t = nft; cols_ = cols_m; }
                        }
                    if (!cand_mode) break;
                }
                // ... which pays off once the launch fills the chip.  A workgroup of it is long (KO outputs x T taps one
                // interval at a time: ~130 us at VHQ, 1.5 ms with the variable-rate clock), so a launch of a few of them
                // loses to lane-per-output k_interp, whose time grows with the work instead (measured, us per output x tap:
                // k_interp 2.5e-6; a k_interp_tile workgroup 6.3e-5 constant rate, 2.9e-4 variable rate; 256 CUs):
                // 96 000-frame variable-rate chunk 1.5 ms -> 0.1 ms on k_interp; 10 s stereo constant rate 134 us on
                // the tile kernel (503 on k_interp); 1 s stereo 75 us on k_interp (127 on the tile kernel).
                if (KO) {
                    const double wgs = (double)((nf_t + KO - 1) / KO) * cols_;
                    const double t_tile = std::ceil(wgs / 256.) * (double)KO * p->T * (vr ? 2.9e-4 : 6.3e-5) * (twin ? kTwinWg : pair_mode ? kPairWg : 1.);
                    const double t_lane = (wave_ok ? kWaveUsPerTap : 2.5e-6) * (double)nf * ((double)j.n_clips * j.n_channels) * p->T;
                    if (t_lane < t_tile) KO = 0;
                }
            }
            if (!KO && wave_ok) { // a half-chain per quad of lanes (k_interp_wave)
                InterpWaveArgs wa;
                wa.ia = ia; wa.span_cap = (int32_t)wave_span; wa.done_words = nullptr; wa.done_seq = 0;
                if (vr) vr_advance(wa.ia);
                {
                    const uint64_t wgs = (uint64_t)((nf + 31) / 32) * ((uint64_t)j.n_clips * j.n_channels);
                    if (cd && done == 0 && nf == j.out_frames && wgs <= cd->cap) { // the whole job is this launch
                        wa.done_words = cd->words; wa.done_seq = cd->seq;
                        cd->n_wgs = (uint32_t)wgs;
                    }
                }
                void (*wk)(InterpWaveArgs) = vr ? k_interp_wave<IO, Real, true> : k_interp_wave<IO, Real, false>;
                hipLaunchKernelGGL(wk, dim3((unsigned)((nf + 31) / 32), (unsigned)((uint64_t)j.n_clips * j.n_channels), 1), dim3(256), wave_lds, st, wa);
                HIP_TRY(hipGetLastError());
                continue;
            }
            if (KO) {
                InterpTileArgs ta;
                ta.ia = ia; ta.KO = (int32_t)KO; ta.span_cap = (int32_t)((span_cap + 1) / 2 * 2);
                if (vr) { // positions relative to the first output of this chunk
                    typedef unsigned __int128 u128;
                    const u128 T0 = ((u128)vr->t_hi << 64) | vr->t_lo, S0 = ((u128)vr->s_hi << 64) | vr->s_lo,
                               D = ((u128)vr->d_hi << 64) | vr->d_lo;
                    const u128 n = (u128)(uint64_t)done, m = n * (n - 1) / 2;
                    const u128 T1 = T0 + n * S0 + D * (done ? m : 0), S1 = S0 + D * n;
                    ta.ia.t_hi = (uint64_t)(T1 >> 64); ta.ia.t_lo = (uint64_t)T1;
                    ta.ia.s_hi = (uint64_t)(S1 >> 64); ta.ia.s_lo = (uint64_t)S1;
                    ta.ia.d_hi = vr->d_hi; ta.ia.d_lo = vr->d_lo;
                }
                ta.cols_per_clip = pair_mode == 1 ? j.n_channels / 2 : j.n_channels; ta.ch_step = pair_mode == 1 ? 2 : 1;
                ta.m2_in = ta.m2_out = ta.m2_l = ta.m2_k = 0; ta.m2_n = nf; ta.m2_dch = 0;
                if (pair_mode == 1) { ta.m2_in = j.in_chan_stride; ta.m2_out = j.out_chan_stride; ta.m2_dch = 1; }
                if (pair_mode == 2) {
                    ta.m2_l = split_h * p->M; ta.m2_k = split_h * p->L; ta.m2_in = ta.m2_l * j.in_frame_stride; ta.m2_out = ta.m2_k * j.out_frame_stride;
                    ta.m2_n = nf - nf_t; ta.ia.g.out_frames = nf_t;
                }
                const size_t lds = (size_t)(twin ? 2 * (ta.span_cap + 2) : ta.span_cap) * (pair_mode ? 2 : 1) * sizeof(Real) + (size_t)KO * 2 + (size_t)(2 * p->phases + 2) * 4 + 64;
                const dim3 tgrid((unsigned)((nf_t + KO - 1) / KO), (unsigned)((uint64_t)j.n_clips * ta.cols_per_clip), 1);
                void (*tk)(InterpTileArgs) = pair_mode ? (vr ? k_interp_tile<IO, Real, true, true> : k_interp_tile<IO, Real, false, true>)
                                                       : (vr ? k_interp_tile<IO, Real, true, false> : k_interp_tile<IO, Real, false, false>);
                if constexpr (sizeof(Real) == 4)
                    if (twin) tk = vr ? k_interp_tile<IO, Real, true, true, true> : k_interp_tile<IO, Real, false, true, true>;
                if (const char *e = ensure_dyn_lds((const void *)tk, lds)) return e;
                hipLaunchKernelGGL(tk, tgrid, dim3(1024), lds, st, ta);
                HIP_TRY(hipGetLastError());
                continue;
            }
            if (vr) {
                if ((1 << ia.lgP) != ia.P) return "variable-rate needs a power-of-two phase count";
                // position of the first output of this launch: advance (T0, S0) by `done` outputs
                typedef unsigned __int128 u128;
                const u128 T0 = ((u128)vr->t_hi << 64) | vr->t_lo, S0 = ((u128)vr->s_hi << 64) | vr->s_lo,
                           D = ((u128)vr->d_hi << 64) | vr->d_lo;
                const u128 n = (u128)(uint64_t)done, m = n * (n - 1) / 2;
                const u128 T1 = T0 + n * S0 + D * (done ? m : 0), S1 = S0 + D * n;
                ia.t_hi = (uint64_t)(T1 >> 64); ia.t_lo = (uint64_t)T1;
                ia.s_hi = (uint64_t)(S1 >> 64); ia.s_lo = (uint64_t)S1;
                ia.d_hi = vr->d_hi; ia.d_lo = vr->d_lo;
                hipLaunchKernelGGL((k_interp<IO, Real, true>), grid, block, 0, st, ia);
            } else {
                hipLaunchKernelGGL((k_interp<IO, Real, false>), grid, block, 0, st, ia);
            }
        } else {
            hipLaunchKernelGGL((k_gather<IO, Real>), grid, block, 0, st, a);
        }
        HIP_TRY(hipGetLastError());
    }
    return nullptr;
}

template <typename IO, typename Real>
static const char *launch_tile(Plan *p, const hipsoxr_job_t &j, hipStream_t st, const TileGeom &g_in)
{
    const DeviceBank &d = p->dev[sizeof(Real) == 4 ? 0 : 1];
    // float64 planar kernel: a job of few 32-period slabs runs on 16-period ones (k_tile_mfma64_p<.., PB>) — same
    // tables, half the slab: the geometry's LDS figures are re-derived here
    TileGeom g = g_in;
    int f64_pb = 32;
    if (sizeof(Real) == 8 && g.variant == 2) {
        const int64_t slabs32 = ((j.out_k0 + j.out_frames - 1) / g.Lc - j.out_k0 / g.Lc + 32) / 32 * (int64_t)j.n_clips * j.n_channels;
        if ((slabs32 < 6 * 256 || switches().dbg_mfma64_pb == 16) && switches().dbg_mfma64_pb != 32) {
            f64_pb = 16;
            g.pb = 16;
            g.x_count = (g.pb - 1) * (int32_t)g.Mc + g.span;
            const int32_t rows_total = (g.x_count + (int32_t)g.Mc - 1) / (int32_t)g.Mc + 3;
            g.plane = (rows_total * g.rowR + 63) / 64 * 64;
            g.lds_bytes = ((size_t)g.plane * 4 + g.rowR) * sizeof(Real);
        }
    }
    // float32 planar kernel: slab size and unit split by job size.  A slab of 64 periods (41 KB of LDS, three workgroups
    // per CU) has 2 n_rt units (row tile x 32 periods), one of 32 periods (20 KB, seven per CU) n_rt; either runs as ONE
    // workgroup (four waves, the units dealt round-robin) or SPLIT over ceil(units / 4) workgroups of one unit per wave,
    // each staging the slab for itself.  What a job of few slabs costs is decided by how many workgroups deep the CUs
    // are stacked ("layers": the dispatcher fills 256 CUs evenly only in whole layers) times what one workgroup does
    // serially, plus staging; the constants are fitted to tools/slab_ab.sh sweeps (10 .. 6016 slabs, 48k -> 44.1k VHQ,
    // profiles/r03_ab_experiments.txt), in units of one unit's MFMA time:
    //     cost = c0 + layers x (units per wave) x k;   (pb, one unit per wave): c0, k = 32: 1.15, 1.153 | 64: 1.25, 1.41
    //                                                  (pb, several)          :         32: 2.35, 0.958 | 64: 3.32, 1.052
    // e.g. 47 slabs (a 10 s clip): 64/split (235 workgroups, one layer); 20: 32/split (120 workgroups staging half as
    // much); 376: 32/whole (752 workgroups, 3 layers of 3 units: 35 us where round 2's 64/4 took 51); from 512 slabs of
    // 64 on the whole-slab form is the rule again (12 waves per CU stream coefficients for 20 units each).
    int f32_split = 0;
    if (sizeof(Real) == 4 && g.variant == 2 && !switches().dbg_slab64) {
        const int64_t periods = (j.out_k0 + j.out_frames - 1) / g.Lc - j.out_k0 / g.Lc + 1, cols_ = (int64_t)j.n_clips * j.n_channels;
        const int64_t slabs64 = (periods + 63) / 64 * cols_, slabs32 = (periods + 31) / 32 * cols_;
        int best_pb = 64, best_split = 1;
        if (slabs64 < 2048 || switches().dbg_slab32) {
            double best = 1e300;
            for (int pb = switches().dbg_slab32 ? 32 : 64; pb >= 32; pb -= 32) {
                const int units = (pb / 32) * g.n_rt, full = (units + 3) / 4;
                for (int split : {1, full}) {
                    const int upw = (units + 4 * split - 1) / (4 * split);
                    const double wgs_ = (double)((pb == 64 ? slabs64 : slabs32) * split);
                    double layers = std::ceil(wgs_ / 256.);
                    // (a partly filled last layer of multi-unit workgroups costs less than a full one: half-way;
                    //  64-period slabs split into single units, three per CU: between 1.5 and 3 x 256 workgroups the
                    //  dispatcher stacks them three deep on the CUs it has started on — refit after the round-3 kernels)
                    if (upw > 1) layers = 0.5 * (layers + wgs_ / 256.);
                    else if (pb == 64 && wgs_ > 384. && wgs_ <= 768.) layers = 3.;
                    const double c0 = pb == 32 ? (upw == 1 ? 1.15 : 2.35) : (upw == 1 ? 1.25 : 3.32);
                    const double k = pb == 32 ? (upw == 1 ? 1.153 : 0.958) : (upw == 1 ? 1.41 : 1.052);
                    const double cost = c0 + layers * upw * k;
                    if (cost < best) { best = cost; best_pb = pb; best_split = split; }
                }
            }
        }
        // HIPSOXR_DEBUG_TILE_FORM (debug builds): 1 = 64 periods whole, 2 = 64 split, 3 = 32 whole, 4 = 32 split — what
        // tests/test_gpu_launch_forms.py::test_chosen_form_is_near_the_best compares the rule above against.
        // (Round 4 also built a fifth form — 512 workgroups each WALKING an equal share of a column's units, slab after
        //  slab — on the theory that 282 slabs on 256 CUs lose a fifth to layer quantisation.  They do not any more: the
        //  split forms already give every SIMD its 6-7 units, all resident at once; walk 33.7 us vs 28.9 (32 split) on the
        //  60 s clip, never ahead at any of eight sizes — profiles/r04_ab_experiments.txt §6.  Removed.)
        const int force = switches().dbg_tile_form;
        if (force >= 1 && force <= 4) {
            best_pb = force <= 2 ? 64 : 32;
            const int units = (best_pb / 32) * g.n_rt;
            best_split = (force & 1) ? 1 : (units + 3) / 4;
        }
        f32_split = best_split;
        if (best_pb == 32) {
            g.pb = 32;
            g.x_count = (g.pb - 1) * (int32_t)g.Mc + g.span;
            const int32_t rows_total = (g.x_count + (int32_t)g.Mc - 1) / (int32_t)g.Mc + 3;
            g.plane = (rows_total * g.rowR + 63) / 64 * 64;
            g.lds_bytes = ((size_t)g.plane * 4 + g.rowR) * sizeof(Real);
        }
    }
    // float32 MFMA kernel in its general form (k_tile_mfma: input periods that are no multiple of 16, e.g. 44.1k -> 16k):
    // a job of few 64-period slabs — a 96 000-frame stream chunk is four — runs on 16-period ones: four times as many
    // workgroups, each staging a quarter and walking a chain a quarter as long (one wave does a row tile x ALL the
    // slab's periods, and its ~880 k-steps cost the same whether they feed four MFMAs or one: the chain is bound by its
    // per-step address arithmetic).  96 000-frame chunk, int16 44.1k -> 16k: kernel 53.6 -> 26.5 us, the stream call 108 -> 81 us.
    bool v1_small = false; // the small-job form of the general-period kernel (16-period slabs, half-chains on two waves)
    if (g.variant == 1 && g.pb > 16 && !switches().dbg_slab64) { // (float64 too: k_tile_mfma<IO, double, 1>)
        const int64_t periods = (j.out_k0 + j.out_frames - 1) / g.Lc - j.out_k0 / g.Lc + 1;
        // (up to 96 slabs of 64 periods: 8 x 96 workgroups of 10 waves are what the chip holds at once — tools/slab16_ab.sh:
        //  50 slabs 54 -> 33 us, 100 slabs 65 -> 63, 127 slabs 66 -> 76)
        const int64_t s64 = (periods + 63) / 64 * (int64_t)j.n_clips * j.n_channels;
        v1_small = s64 <= 96 || switches().dbg_slab32;
        // Beyond that: 16-period slabs WITHOUT the half-chain split where four times as many, four times shorter
        // workgroups fill the chip's layers better than 64-period ones — a layer of 256 workgroups of the 16-period
        // form costs 0.276 of a 64-period layer (not 0.25), a last 64-period layer that is at most half full 0.82
        // (tools/slab16_ab.sh: 127 slabs 45 -> 33 us, 300: 108 -> 77, 800: 213 -> 190; 250 and 500 stay)
        bool v1_mid = false;
        if (!v1_small && s64 < 4096 && !switches().no_halves) {
            // (in layers of the plan's own slab size — 64 periods, or 32 where a float64 slab of 64 does not fit LDS,
            //  whose layer a 16-period one costs 0.53 of)
            const double s0 = (double)((periods + g.pb - 1) / g.pb * (int64_t)j.n_clips * j.n_channels);
            const double l0 = std::ceil(s0 / 256.), frac = s0 / 256. - (l0 - 1.);
            const double est0 = (l0 - 1.) + (frac <= 0.5 ? 0.82 : 1.0);
            const double est16 = 0.04 + (g.pb == 64 ? 0.276 : 0.53) * std::ceil((double)((periods + 15) / 16 * (int64_t)j.n_clips * j.n_channels) / 256.);
            v1_mid = est16 < est0;
        }
        if (v1_small || v1_mid) {
            g.pb = 16;
            g.x_count = ((g.pb - 1) * (int32_t)g.Mc + g.span + 3) / 4 * 4;
            g.lds_bytes = ((size_t)g.x_count + (size_t)g.pad * (g.x_count / g.Mc + 1) + 8) * sizeof(Real);
        }
    }
    TileArgs a;
    a.in = j.in; a.out = j.out;
    a.tab = g.variant >= 1 ? d.tile_tab_m : d.tile_tab;
    a.e0 = g.variant >= 1 ? d.tile_i0_m : d.tile_i0;
    a.Lc = g.Lc; a.Mc = g.Mc; a.n_rt = g.n_rt; a.I_h = g.I_h;
    a.pad = g.pad; a.i_min = g.i_min; a.x_count = g.x_count; a.pb = g.pb;
    a.n_clips = j.n_clips; a.n_channels = j.n_channels;
    a.ics = j.in_clip_stride; a.ifs = j.in_frame_stride; a.ichs = j.in_chan_stride;
    a.ocs = j.out_clip_stride; a.ofs = j.out_frame_stride; a.ochs = j.out_chan_stride;
    a.in_abs0 = j.in_abs0; a.in_frames = j.in_frames;
    a.out_k0 = j.out_k0; a.out_frames = j.out_frames;
    a.oc.clip_counter = j.clip_counter; a.oc.dither = j.dither; a.oc.seed = j.dither_seed; a.oc.ch0 = t_ch_base;
    // periods touched: floor(k0/Lc) .. floor((k0+n-1)/Lc)
    const int64_t b_lo = j.out_k0 / g.Lc, b_hi = (j.out_k0 + j.out_frames - 1) / g.Lc;
    a.b_first = b_lo;
    const int64_t n_blocks = (b_hi - b_lo + g.pb) / g.pb;
    const uint64_t cols = (uint64_t)j.n_clips * j.n_channels;
    if (cols > 65535) return "too many (clip, channel) columns for one launch (max 65535)";
    if (n_blocks > 2147483647LL) return "job too long for one launch";
    // waves per workgroup: one tile per wave when n_rt <= 16, else the even split with most waves
    int nw = g.n_rt;
    if (g.n_rt > 16) {
        int best = 16, best_waste = 1 << 30;
        for (int w = 16; w >= 8; --w) {
            int rounds = (g.n_rt + w - 1) / w, waste = rounds * w - g.n_rt;
            if (waste < best_waste) { best_waste = waste; best = w; }
        }
        nw = best;
    }
    if (g.variant == 2) nw = 4; // k_tile_mfma_p / k_tile_mfma64_p: one wave per SIMD, the slab's units dealt round-robin
    // (f32: tile x half of 64 periods; float64: tile x all periods of the slab — or tile x 16 periods, HIPSOXR_DEBUG_MFMA64_SPLIT)
    const int units_per_slab = sizeof(Real) == 4 ? (g.pb / 32) * g.n_rt : (f64_pb == 32 && switches().dbg_mfma64_split) ? 2 * g.n_rt : g.n_rt;
    if (switches().dbg_nrt) { a.n_rt = switches().dbg_nrt; nw = a.n_rt; }
    if (switches().dbg_nw) nw = switches().dbg_nw;
    a.n_waves = nw;
    {
        a.dbg = switches().dbg_flags;
    }
    // (HIPSOXR_DEBUG_* are timing experiments only; results may be wrong when they are set)
    void (*kern)(TileArgs) = g.aligned ? k_tile<IO, Real, 16, true> : k_tile<IO, Real, 16, false>;
    if constexpr (sizeof(Real) == 4) {
        if (g.variant == 1) kern = g.pb == 64 ? k_tile_mfma<IO, float, 4> : g.pb == 32 ? k_tile_mfma<IO, float, 2> : k_tile_mfma<IO, float, 1>;
        if (g.variant == 2) kern = k_tile_mfma_p<IO>;
    } else {
        if (g.variant == 1) kern = g.pb == 32 ? k_tile_mfma<IO, double, 2> : k_tile_mfma<IO, double, 1>; // (pb = 64 is never chosen for float64: build_tile_tables)
        if (g.variant == 2) kern = f64_pb == 16 ? k_tile_mfma64_p<IO, 1, 16> : switches().dbg_mfma64_split ? k_tile_mfma64_p<IO, 1, 32> : k_tile_mfma64_p<IO, 2, 32>;
    }
    a.rowR = g.rowR; a.plane = g.plane;
    a.halves = 0; a.scratch_off = 0;
    dim3 grid((unsigned)n_blocks, (unsigned)cols, 1), block(64 * nw);
    if (g.variant == 2) {
        // few slabs (e.g. one 60 s mono clip = 282): spread each slab's 2*n_rt units over up to
        // ceil(2*n_rt/4) workgroups so that every CU gets an equal share (3 resident per CU)
        const int64_t wgs = n_blocks * (int64_t)cols;
        // (from two workgroups per CU on, splitting only adds staging: measured 80 vs 92 us on a 60 s stereo clip)
        int split = wgs >= 512 ? 1 : (int)std::min<int64_t>((units_per_slab + 3) / 4, (2 * 3 * 256) / std::max<int64_t>(wgs, 1));
        if (f32_split) split = f32_split; // (float32: chosen with the slab size above)
        if (switches().dbg_split) split = switches().dbg_split;
        grid.z = (unsigned)std::max(1, split);
        a.xz = 0; a.nx = (int32_t)n_blocks;
        if (grid.z > 1 && !switches().no_xcd_split && (n_blocks + 7) / 8 * 8 * (int64_t)grid.z < 2147483647LL) {
            a.xz = (int32_t)grid.z; // XCD-aware 1-D ids instead of the z dimension
            grid.x = (unsigned)((n_blocks + 7) / 8 * 8 * (int64_t)grid.z);
            grid.z = 1;
        }
    } else {
        a.xz = 0; a.nx = (int32_t)n_blocks;
        // few slabs (one column of a stream chunk: 96 000 frames at 44.1k -> 16k are 4 slabs on 256 CUs): the
        // row tiles of a slab go to several workgroups of fewer computing waves, each staging the slab for itself
        const int64_t wgs = n_blocks * (int64_t)cols;
        int split = 1;
        // (as many workgroups as fill the chip once: every one of them stages the whole slab)
        if (wgs < 128 && g.n_rt > 1 && !switches().dbg_nw && !switches().no_tile_split) split = (int)std::min<int64_t>(g.n_rt, 256 / wgs);
        if (switches().dbg_split) split = std::min(switches().dbg_split, g.n_rt);
        if (split > 1) {
            const int per_wg = std::min(16, (g.n_rt + split - 1) / split); // row tiles (= computing waves) per workgroup: a block holds 16 waves
            nw = per_wg; a.n_waves = nw;
            block = dim3((unsigned)std::max(256, 64 * per_wg)); // (at least four waves stage the slab)
            grid.z = (unsigned)((g.n_rt + per_wg - 1) / per_wg);
        }
        // small float32 jobs on 16-period slabs: a row tile's two half-chains on two waves (k_tile_mfma, a.halves)
        if (g.variant == 1 && g.pb == 16 && g_in.pb != 16 && v1_small && !switches().dbg_nw && !switches().dbg_nrt && !switches().no_halves) {
            const int want = (int)grid.z > 1 ? nw : g.n_rt, parts = (want + 7) / 8;
            const int per_wg = (want + parts - 1) / parts; // row tiles per workgroup (at most 8: two waves each), evenly
            nw = 2 * per_wg; a.n_waves = nw; a.halves = 1;
            block = dim3((unsigned)std::max(256, 64 * nw));
            grid.z = (unsigned)((g.n_rt + per_wg - 1) / per_wg);
            a.scratch_off = (int32_t)((g.lds_bytes / sizeof(Real) + 63) / 64 * 64);
            g.lds_bytes = ((size_t)a.scratch_off + (size_t)per_wg * (g.pb / 16) * 4 * 64) * sizeof(Real);
        }
    }
    size_t lds_bytes = g.lds_bytes;
    lds_bytes = std::max<size_t>(lds_bytes, switches().dbg_lds); // occupancy experiments
    if (const char *e = ensure_dyn_lds((const void *)kern, lds_bytes)) return e;
    a.trace = nullptr;
    const char *trace_path = switches().dbg_trace;
    size_t trace_n = 0;
    if (trace_path && g.variant == 2) {
        trace_n = (size_t)grid.x * cols * grid.z * 4 * 16;
        HIP_TRY(hipMalloc((void **)&a.trace, trace_n * 8));
        HIP_TRY(hipMemset(a.trace, 0, trace_n * 8));
    }
    hipLaunchKernelGGL(kern, grid, block, lds_bytes, st, a);
    HIP_TRY(hipGetLastError());
    if (a.trace) { // debugging aid only: synchronous dump of the per-wave time stamps
        std::vector<unsigned long long> h(trace_n);
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipMemcpy(h.data(), a.trace, trace_n * 8, hipMemcpyDeviceToHost));
        if (FILE *f = fopen(trace_path, "wb")) { fwrite(h.data(), 8, trace_n, f); fclose(f); }
        (void)hipFree(a.trace);
    }
    return nullptr;
}

template <typename IO, typename Real>
static const char *launch_wave_dot(Plan *p, const hipsoxr_job_t &j, hipStream_t st)
{
    DeviceBank &d = p->dev[sizeof(Real) == 4 ? 0 : 1];
    if (p->phases) return "wave-dot kernel needs an exact-bank plan";
    {
        std::lock_guard<std::mutex> lk(p->mu);
        if (!d.phase_major) {
            std::vector<Real> pm(p->bank.size());
            for (size_t i = 0; i < pm.size(); ++i) pm[i] = (Real)p->bank[i];
            HIP_TRY(hipMalloc(&d.phase_major, pm.size() * sizeof(Real)));
            HIP_TRY(hipMemcpy(d.phase_major, pm.data(), pm.size() * sizeof(Real), hipMemcpyHostToDevice));
        }
    }
    const uint64_t cols = (uint64_t)j.n_clips * j.n_channels;
    if (cols > 65535) return "too many (clip, channel) columns for one launch (max 65535)";
    if (j.out_frames > ((int64_t)1 << 30)) return "job too long for the wave-dot kernel";
    GatherArgs a;
    a.in = j.in; a.out = j.out; a.bank = nullptr; a.Lpad = 0; a.L = p->L; a.M = p->M; a.T = p->T;
    a.n_clips = j.n_clips; a.n_channels = j.n_channels;
    a.ics = j.in_clip_stride; a.ifs = j.in_frame_stride; a.ichs = j.in_chan_stride;
    a.ocs = j.out_clip_stride; a.ofs = j.out_frame_stride; a.ochs = j.out_chan_stride;
    a.in_abs0 = j.in_abs0; a.in_frames = j.in_frames; a.out_k0 = j.out_k0; a.out_frames = j.out_frames;
    __int128 kM = (__int128)j.out_k0 * p->M;
    a.d0 = (int64_t)(kM / p->L); a.p0 = (int64_t)(kM % p->L);
    a.oc.clip_counter = j.clip_counter; a.oc.dither = j.dither; a.oc.seed = j.dither_seed; a.oc.ch0 = t_ch_base;
    a.ch_fast = 0;
    const int32_t per_wave = 16;
    const int64_t waves = (j.out_frames + per_wave - 1) / per_wave;
    hipLaunchKernelGGL((k_wave_dot<IO, Real>), dim3((unsigned)((waves + 3) / 4), (unsigned)cols, 1), dim3(256), 0, st, a,
                       (const Real *)d.phase_major, per_wave);
    HIP_TRY(hipGetLastError());
    return nullptr;
}

static constexpr double kGatherWaveTaps = 16e6;
template <typename IO, typename Real>
static const char *launch_typed(Plan *p, const hipsoxr_job_t &j, hipStream_t st, const VrPos *vr, ResidentLaunch *res = nullptr,
                                ChainDone *cd = nullptr)
{
    if (res) return launch_gather<IO, Real>(p, j, st, vr, res);
    const int prec = sizeof(Real) == 4 ? 0 : 1;
    TileGeom gv, gm; // VALU-tile and MFMA-tile geometries (f32: planes / k_tile_mfma; f64: k_tile_mfma<IO, double, NG>)
    {
        std::lock_guard<std::mutex> lk(g_geom_mu);
        if (TileGeom *gp = geom_find(p, prec, 0)) gv = *gp;
        if (TileGeom *gp = geom_find(p, prec, 1)) gm = *gp;
    }
    int kernel = j.kernel;
    if (kernel == HIPSOXR_KERNEL_WAVE_DOT) return vr ? "wave-dot kernel does not do variable rate" : launch_wave_dot<IO, Real>(p, j, st);
    if (kernel == HIPSOXR_KERNEL_EXACT || kernel == HIPSOXR_KERNEL_FFT) kernel = HIPSOXR_KERNEL_AUTO;
    if (p->phases) { // interpolated-phase plan: one kernel (k_interp, dispatched by launch_gather)
        if (kernel != HIPSOXR_KERNEL_AUTO && kernel != HIPSOXR_KERNEL_GATHER)
            return "tile kernel unavailable for this plan";
        return launch_gather<IO, Real>(p, j, st, vr, nullptr, cd);
    }
    if (vr) return "variable-rate needs an interpolated-phase plan";
    if (kernel == HIPSOXR_KERNEL_TILE_VALU && !gv.ok) return "tile kernel unavailable for this plan";
    if (kernel == HIPSOXR_KERNEL_TILE_MFMA && !gm.ok) return "tile kernel unavailable for this plan";
    if (kernel == HIPSOXR_KERNEL_TILE) {
        if (!gv.ok && !gm.ok) return "tile kernel unavailable for this plan";
        kernel = gm.ok ? HIPSOXR_KERNEL_TILE_MFMA : HIPSOXR_KERNEL_TILE_VALU;
    }
    if (kernel == HIPSOXR_KERNEL_AUTO) {
        // a tile kernel pays off once a job spans a few thousand outputs per column
        const TileGeom &g = gm.ok ? gm : gv;
        bool big = g.ok && j.out_frames >= 16 * g.Lc && j.out_frames >= 4096;
        // ... except for a stream chunk whose result the kernel writes straight into pinned host memory (`cd`: engine.cpp's
        // direct path) while it is far too small to fill the chip with slabs.  As a kernel k_gather_wave is the slower one
        // even there (96 000 frames at 44.1k -> 16k: 17.5 against 15.6 us; back to back on device buffers 15.0 against 9.6),
        // but the CALL is shorter with it — its outputs leave as runs of 16 neighbouring samples, the tiles' as one sample
        // per lane of a row tile: 20 000-frame int16 calls 35 against 41 us, 48 000-frame 43.5 against 45.6, 96 000-frame
        // the same (interleaved A/B on one box, tools/gw_time.sh).  Up to kGatherWaveTaps output x tap products.
        if (big && cd && !switches().no_gather_wave && p->T >= 32 &&
            (double)j.out_frames * j.n_clips * j.n_channels * p->T < (switches().dbg_gw_taps ? switches().dbg_gw_taps * 1e6 : kGatherWaveTaps))
            big = false;
        kernel = !big ? HIPSOXR_KERNEL_GATHER : gm.ok ? HIPSOXR_KERNEL_TILE_MFMA : HIPSOXR_KERNEL_TILE_VALU;
    }
    if (kernel == HIPSOXR_KERNEL_TILE_MFMA) return launch_tile<IO, Real>(p, j, st, gm);
    if (kernel == HIPSOXR_KERNEL_TILE_VALU) return launch_tile<IO, Real>(p, j, st, gv);
    return launch_gather<IO, Real>(p, j, st, nullptr, nullptr, cd);
}

bool resident_post(const Plan &p, volatile uint64_t *w, uint32_t seq, int64_t in_abs0, int64_t in_frames, int64_t out_k0, int64_t out_frames,
                   const VrPos *vr)
{
    const __int128 kM = vr ? (__int128)0 : (__int128)out_k0 * p.M; // (variable rate: positions come from the message's own clock)
    const int64_t d0 = (int64_t)(kM / p.L), p0 = (int64_t)(kM % p.L);
    const uint64_t lim = 1ULL << 48;
    if ((uint64_t)in_abs0 >= lim || (uint64_t)out_k0 >= lim || (uint64_t)d0 >= lim || (uint64_t)in_frames >= (1u << 24) ||
        (uint64_t)p0 >= (1u << 24) || (uint64_t)out_frames >= lim)
        return false;
    const uint64_t tag = (uint64_t)(seq & 0xffffu) << 48;
    // the words validate themselves (k_chain_resident): no order is needed among them — they may sit in
    // write-combining device memory — only everything the message refers to must have left before them
    __builtin_ia32_sfence();
    w[0] = tag | (uint64_t)in_abs0;
    w[1] = tag | (uint64_t)out_k0;
    w[2] = tag | (uint64_t)d0;
    w[3] = tag | ((uint64_t)in_frames << 24) | (uint64_t)p0;
    w[4] = tag | (uint64_t)out_frames;
    if (vr) { // the variable-rate clock of this message: position, step, step increment (Q64.64), 48 + 48 + 32 bits each
        auto put = [&](int i, uint64_t hi, uint64_t lo) {
            w[i] = tag | (lo & kResidentMask48);
            w[i + 1] = tag | ((lo >> 48) | ((hi & 0xffffffffULL) << 16));
            w[i + 2] = tag | (hi >> 32);
        };
        put(5, vr->t_hi, vr->t_lo); put(8, vr->s_hi, vr->s_lo); put(11, vr->d_hi, vr->d_lo);
    }
    __builtin_ia32_sfence();
    return true;
}
void resident_leave(volatile uint64_t *w, uint32_t epoch)
{
    w[15] = (uint64_t)epoch;
    __builtin_ia32_sfence();
}

// a chunk appended to a stream's device ring (engine.cpp device_process): 16 bytes per thread where both ends allow it
__global__ void __launch_bounds__(256) k_copy16(uint4 *__restrict__ dst, const uint4 *__restrict__ src, size_t n16)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}
__global__ void __launch_bounds__(256) k_copy2(uint16_t *__restrict__ dst, const uint16_t *__restrict__ src, size_t n2)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n2) dst[i] = src[i];
}
const char *launch_copy(void *dst, const void *src, size_t bytes, void *stream)
{
    if (!bytes) return nullptr;
    hipStream_t st = (hipStream_t)stream;
    if ((((uintptr_t)dst | (uintptr_t)src | bytes) & 15) == 0) {
        const size_t n = bytes / 16;
        hipLaunchKernelGGL(k_copy16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (uint4 *)dst, (const uint4 *)src, n);
    } else if ((((uintptr_t)dst | (uintptr_t)src | bytes) & 1) == 0) { // (frames are at least two bytes)
        const size_t n = bytes / 2;
        hipLaunchKernelGGL(k_copy2, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (uint16_t *)dst, (const uint16_t *)src, n);
    } else {
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st));
        return nullptr;
    }
    HIP_TRY(hipGetLastError());
    return nullptr;
}

template <typename IO, typename Real>
static const char *launch_chain_items_typed(Plan *p, uint32_t nch, bool dither, const ChainItem *items, const ChainItem *items_dev, uint32_t n_items,
                                            hipStream_t st, bool *handled)
{
    int64_t max_out = 0;
    for (uint32_t i = 0; i < n_items; ++i) max_out = std::max(max_out, items[i].out_frames);
    if (max_out >= 4096 || (uint64_t)n_items * nch > 65535 || switches().no_chain) return nullptr;
    const DeviceBank &d = p->dev[sizeof(Real) == 4 ? 0 : 1];
    // geometry of k_chain as launch_gather sets it up: few outputs per workgroup for short chunks, LDS = NO coefficient rows + the span
    int NO = max_out <= 512 ? 8 : 32;
    if (switches().dbg_chain_no) NO = switches().dbg_chain_no;
    const int64_t shift = (p->M + p->L - 1) / p->L + 2;
    auto chain_lds = [&](int no, int32_t *span_cap) {
        const int64_t sc = (int64_t)p->T + (int64_t)no * shift + 4;
        *span_cap = (int32_t)sc;
        return (size_t)no * (p->T + 16 / sizeof(Real)) * sizeof(Real) + (size_t)((sc + 3) & ~3) * sizeof(Real) + (size_t)no * 16;
    };
    int32_t span_cap = 0;
    while (NO > 2 && chain_lds(NO, &span_cap) > 150 * 1024) NO /= 2;
    const size_t lds = chain_lds(NO, &span_cap);
    if (lds > 150 * 1024 || shift >= (1 << 20)) return nullptr;
    ChainMultiArgs m;
    std::memset(&m, 0, sizeof m);
    GatherArgs &a = m.ca.ia.g;
    a.bank = d.tap_major; a.Lpad = d.Lpad; a.L = p->L; a.M = p->M; a.T = p->T;
    a.n_clips = 1; a.n_channels = nch;
    a.ics = 0; a.ifs = nch; a.ichs = 1; a.ocs = 0; a.ofs = nch; a.ochs = 1; // a stream's own layout: interleaved frames
    a.oc.dither = dither ? 1u : 0u; a.oc.ch0 = 0;
    a.ch_fast = 0;
    m.ca.NO = NO; m.ca.span_cap = span_cap;
    m.n_channels = nch;
    void (*ck)(ChainMultiArgs) = nullptr;
    if (p->phases) {
        m.ca.ia.tab = d.interp_tab; m.ca.ia.P = p->phases;
        while ((1 << m.ca.ia.lgP) < m.ca.ia.P) ++m.ca.ia.lgP;
        ck = k_chain_multi<IO, Real, 1>;
    } else {
        DeviceBank &dm = p->dev[sizeof(Real) == 4 ? 0 : 1];
        const char *err = nullptr;
        {
            std::lock_guard<std::mutex> lk(p->mu);
            if (!dm.phase_major) {
                std::vector<Real> pm(p->bank.size());
                for (size_t i = 0; i < pm.size(); ++i) pm[i] = (Real)p->bank[i];
                if (hipMalloc(&dm.phase_major, pm.size() * sizeof(Real)) != hipSuccess) err = "hipMalloc failed";
                else if (hipMemcpy(dm.phase_major, pm.data(), pm.size() * sizeof(Real), hipMemcpyHostToDevice) != hipSuccess) err = "hipMemcpy failed";
            }
        }
        if (err) return err;
        m.ca.phase_major = dm.phase_major;
        ck = k_chain_multi<IO, Real, 0>;
    }
    if (n_items == 1) m.one = items[0];
    else if (!items_dev) return "internal: a many-streams launch needs a device-readable item table";
    else m.items = items_dev;
    if (const char *e = ensure_dyn_lds((const void *)ck, lds)) return e;
    const unsigned gx = (unsigned)std::max<int64_t>(1, (max_out + NO - 1) / NO);
    hipLaunchKernelGGL(ck, dim3(gx, (unsigned)(n_items * nch), 1), dim3(256), lds, st, m);
    HIP_TRY(hipGetLastError());
    *handled = true;
    return nullptr;
}

const char *launch_chain_items(Plan *p, int elem, uint32_t n_channels, bool dither, const ChainItem *items, const ChainItem *items_dev,
                               uint32_t n_items, void *stream, bool *handled)
{
    *handled = false;
    if (!n_items || !n_channels) return nullptr;
    if (const char *e = device_bank_ensure(p, engine_prec(elem))) return e;
    hipStream_t st = (hipStream_t)stream;
    switch (elem) {
    case HIPSOXR_F32: return launch_chain_items_typed<float, float>(p, n_channels, false, items, items_dev, n_items, st, handled);
    case HIPSOXR_F64: return launch_chain_items_typed<double, double>(p, n_channels, false, items, items_dev, n_items, st, handled);
    case HIPSOXR_I32: return launch_chain_items_typed<int32_t, double>(p, n_channels, false, items, items_dev, n_items, st, handled);
    case HIPSOXR_I16: return launch_chain_items_typed<int16_t, float>(p, n_channels, dither, items, items_dev, n_items, st, handled);
    }
    return "unknown element type";
}

const char *launch_job(Plan *p, const hipsoxr_job_t &j, void *stream, const VrPos *vr, ResidentLaunch *res, ChainDone *cd)
{
    if (cd) cd->n_wgs = 0;
    if (j.out_frames <= 0 || j.n_clips == 0 || j.n_channels == 0) return res ? "resident kernel: empty job" : nullptr;
    if (res && (uint64_t)j.n_clips * j.n_channels > 65535) return "resident kernel: too many columns";
    // Ragged batch (hipsoxr_job_t::clip_table): one launch of the frequency-domain engine when it can take the job
    // (the kernel reads its clip's row), else clip by clip through the ordinary path — clips are independent, so the
    // results are the same either way; bit-exact engines stay bit-exact.
    if (j.clip_table) {
        if (vr || res) return "ragged batches: constant-rate device jobs only";
        if (j.in_abs0 != 0 || j.out_k0 != 0) return "ragged batches: whole signals only (in_abs0 == 0, out_k0 == 0)";
        int64_t total_out = 0;
        for (uint32_t c = 0; c < j.n_clips; ++c) {
            const int64_t *r = j.clip_table + 4 * (size_t)c;
            if (r[0] < 0 || r[2] < 0 || r[1] < 0 || r[3] < 0 || r[1] > j.in_frames || r[3] > j.out_frames || (uint64_t)r[3] > plan_out_len(*p, (uint64_t)r[1]))
                return "ragged batches: a clip's offsets or frame counts are negative, exceed the job's, or exceed the plan's output length";
            total_out += r[3];
        }
        // AUTO takes the 1e-6-class engine under the same rule as for equal-length jobs (>= 2^13 outputs in all): engine
        // choice — and with it bit-exactness — does not depend on whether a table is present
        const bool want_fft = j.kernel == HIPSOXR_KERNEL_FFT || j.kernel == HIPSOXR_KERNEL_FFT_F64;
        const bool big = total_out * (int64_t)j.n_channels >= (1 << 13);
        if ((want_fft || (j.kernel == HIPSOXR_KERNEL_AUTO && big && !switches().no_fft)) &&
            (uint64_t)j.n_clips * j.n_channels <= 65535 && fft_job_eligible(*p, j)) {
            if (const char *e = device_bank_ensure(p, engine_prec(j.elem))) return e;
            // The kernel reads the DEVICE copy of the table.  Without one (clip_table_dev == NULL) the host table — the
            // one validated above — is uploaded here, in stream order (stream-ordered allocation: the buffer lives until
            // the launch behind it has run).  A caller-supplied device copy is trusted to equal the host table.
            hipsoxr_job_t jj = j;
            void *tmp = nullptr;
            if (!jj.clip_table_dev) {
                const size_t bytes = (size_t)j.n_clips * 4 * sizeof(int64_t);
                if (hipMallocAsync(&tmp, bytes, (hipStream_t)stream) != hipSuccess) return "ragged batches: no device memory for the clip table";
                if (hipMemcpyAsync(tmp, j.clip_table, bytes, hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess) {
                    (void)hipFreeAsync(tmp, (hipStream_t)stream);
                    return "ragged batches: clip table upload failed";
                }
                jj.clip_table_dev = (const int64_t *)tmp;
            }
            bool handled = false;
            const char *e = launch_fft(p, jj, stream, &handled);
            if (tmp) (void)hipFreeAsync(tmp, (hipStream_t)stream);
            if (e) return e;
            if (handled) return nullptr;
        }
        if (j.kernel == HIPSOXR_KERNEL_FFT || j.kernel == HIPSOXR_KERNEL_FFT_F64) return "FFT engine unavailable for this ragged job (unit-stride float columns of a tabled ratio)";
        const size_t es = elem_size(j.elem);
        for (uint32_t c = 0; c < j.n_clips; ++c) {
            const int64_t *r = j.clip_table + 4 * (size_t)c;
            if (r[3] == 0) continue;
            hipsoxr_job_t one = j;
            one.clip_table = one.clip_table_dev = nullptr;
            one.n_clips = 1;
            one.in = (const char *)j.in + r[0] * (int64_t)es;
            one.out = (char *)j.out + r[2] * (int64_t)es;
            one.in_frames = r[1]; one.out_frames = r[3];
            if (const char *e = launch_job(p, one, stream)) return e;
        }
        return nullptr;
    }
    // Kernels index (clip, channel) columns through grid.y (<= 65535).  Wider jobs — the Python surface
    // admits 65536 channels like the reference, src/soxr/__init__.py:22 — are folded into several
    // launches over channel (or clip) ranges; columns are independent, so the result is the same.
    if ((uint64_t)j.n_clips * j.n_channels > 65535) {
        const size_t es = elem_size(j.elem);
        hipsoxr_job_t part = j;
        if (j.n_channels > 1) {
            const uint32_t step = j.n_clips > 65535 ? 1 : 65535 / j.n_clips;
            if (j.n_clips > 65535) { // both wide: one clip range at a time, channels folded below it
                for (uint32_t c0 = 0; c0 < j.n_clips; c0 += 65535) {
                    part = j;
                    part.n_clips = std::min<uint32_t>(65535, j.n_clips - c0);
                    part.in = (const char *)j.in + (int64_t)c0 * j.in_clip_stride * (int64_t)es;
                    part.out = (char *)j.out + (int64_t)c0 * j.out_clip_stride * (int64_t)es;
                    if (const char *e = launch_job(p, part, stream, vr)) return e;
                }
                return nullptr;
            }
            for (uint32_t h0 = 0; h0 < j.n_channels; h0 += step) {
                part = j;
                part.n_channels = std::min<uint32_t>(step, j.n_channels - h0);
                part.in = (const char *)j.in + (int64_t)h0 * j.in_chan_stride * (int64_t)es;
                part.out = (char *)j.out + (int64_t)h0 * j.out_chan_stride * (int64_t)es;
                const uint32_t saved = t_ch_base;
                t_ch_base = saved + h0;
                const char *e = launch_job(p, part, stream, vr);
                t_ch_base = saved;
                if (e) return e;
            }
            return nullptr;
        }
        for (uint32_t c0 = 0; c0 < j.n_clips; c0 += 65535) {
            part = j;
            part.n_clips = std::min<uint32_t>(65535, j.n_clips - c0);
            part.in = (const char *)j.in + (int64_t)c0 * j.in_clip_stride * (int64_t)es;
            part.out = (char *)j.out + (int64_t)c0 * j.out_clip_stride * (int64_t)es;
            if (const char *e = launch_job(p, part, stream, vr)) return e;
        }
        return nullptr;
    }
    const int prec = engine_prec(j.elem);
    if (const char *e = device_bank_ensure(p, prec)) return e;
    // Frequency-domain engine: explicit request, or AUTO for large whole-signal float32 jobs.
    // It is NOT bit-identical to the canonical order (about 2e-7 relative RMS), so it is never chosen
    // for HIPSOXR_KERNEL_EXACT — which is what the stream / one-shot host entry points pass.
    // HIPSOXR_KERNEL_FFT_F64: the same engine with float64 arithmetic whatever the I/O type (float32 jobs at the width
    // libsoxr's VHQ recipe computes in; float64 jobs run it anyway).
    const bool want_fft = j.kernel == HIPSOXR_KERNEL_FFT || j.kernel == HIPSOXR_KERNEL_FFT_F64;
    if (want_fft && (vr || res)) return "FFT engine: whole-signal device jobs only";
    // Ratios without an exact bank (interpolated-phase plans): the two-stage form — FFT engine at 1:2 / 2:1 plus a short
    // polyphase stage — for whole-signal float jobs (1e-6 class, like the FFT engine itself; twostage.hip)
    if (!vr && !res && p->phases && (want_fft || (j.kernel == HIPSOXR_KERNEL_AUTO && !switches().no_fft)) &&
        (j.elem == HIPSOXR_F32 || j.elem == HIPSOXR_F64) && !switches().no_two_stage) {
        bool handled = false;
        if (const char *e = launch_two_stage(p, j, stream, &handled)) return e;
        if (handled) return nullptr;
        if (want_fft) return "FFT engine unavailable for this plan (the two-stage form serves HQ / VHQ ratios down to 4:1, whole signals of >= ~5000 frames)";
    }
    if (!vr && !res && (want_fft || j.kernel == HIPSOXR_KERNEL_AUTO)) {
        const bool no_fft = switches().no_fft;
        const bool eligible = fft_job_eligible(*p, j);
        const bool big = (int64_t)j.out_frames * j.n_clips * j.n_channels >= (1 << 13); // even one block pair beats the tiled exact kernels (7 vs 10 us)
        if (want_fft && !eligible)
            return "FFT engine needs a whole-signal float32 or float64 job (in_abs0 == 0, out_k0 == 0) on an HQ/VHQ exact-ratio plan";
        if (eligible && (want_fft || (big && !no_fft))) {
            bool handled = false;
            if (const char *e = launch_fft(p, j, stream, &handled)) return e;
            if (handled) return nullptr;
            if (want_fft) return j.kernel == HIPSOXR_KERNEL_FFT_F64 ? "FFT engine (float64 arithmetic) unavailable for this plan or layout (unit-stride columns of a tabled ratio)"
                                                                     : "FFT engine unavailable for this plan";
        }
    }
    hipStream_t st = (hipStream_t)stream;
    switch (j.elem) {
    case HIPSOXR_F32: return launch_typed<float, float>(p, j, st, vr, res, cd);
    case HIPSOXR_F64: return launch_typed<double, double>(p, j, st, vr, res, cd);
    case HIPSOXR_I32: return launch_typed<int32_t, double>(p, j, st, vr, res, cd);
    case HIPSOXR_I16: return launch_typed<int16_t, float>(p, j, st, vr, res, cd);
    }
    return "invalid element type";
}

} // namespace hipsoxr
