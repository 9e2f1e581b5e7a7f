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
// lives in fft.hip.
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

// ---------------------------------------------------------------------------------------------
// k_interp — interpolated-phase plans (ratios without a small rational form; plan.cpp)
// ---------------------------------------------------------------------------------------------
// One lane per output sample, as k_gather, but the coefficient of tap j is evaluated from the cubic
// of the output's phase interval:  c = fma(fma(fma(a3, x, a2), x, a1), x, a0), one 16/32-byte load
// per tap.  Interval and residual come from exact integer arithmetic on (k*M) mod L, so the result
// is again a pure function of the absolute output index (chunk / launch invariant) and equals
// oracle_interp_port_* bit for bit.
template <typename Real, int N> struct VecN;
template <> struct VecN<float, 4> { typedef float4 type; };
template <> struct VecN<double, 2> { typedef double2 type; };
template <typename Real> struct Vec4;
template <> struct Vec4<float> { typedef float4 type; };
template <> struct Vec4<double> { typedef double4 type; };

//
// VR = true (variable-rate streams, engine.cpp): the position of local output i is the Q64.64
// fixed-point quadratic  t(i) = T0 + i*S0 + D*i(i-1)/2  (constant step: D = 0; linear slew of the
// step: D != 0), evaluated in 128-bit integers — again exact and launch-invariant.  P is a power
// of two, so interval and residual are bit fields of the fraction.
struct InterpArgs {
    GatherArgs g;   // bank/Lpad unused
    const void *tab; // [P][T] of Vec4<Real>
    int32_t P, lgP;
    uint64_t t_hi, t_lo, s_hi, s_lo, d_hi, d_lo; // VR: T0, S0, D (two's complement), Q64.64
};

template <typename IO, typename Real, bool VR>
__global__ void __launch_bounds__(256) k_interp(InterpArgs ia)
{
    typedef typename Vec4<Real>::type V4;
    const GatherArgs &a = ia.g;
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
    constexpr int SH = sizeof(Real) == 4 ? 24 : 32;
    const int32_t T = a.T, H = T / 2;
    uint64_t iv, xq;
    int64_t n0;
    if (VR) {
        typedef unsigned __int128 u128;
        const u128 T0 = ((u128)ia.t_hi << 64) | ia.t_lo, S0 = ((u128)ia.s_hi << 64) | ia.s_lo,
                   D = ((u128)ia.d_hi << 64) | ia.d_lo;
        const uint64_t i = (uint64_t)idx, m = i * (i - 1) / 2; // i = 0: 0 * (2^64-1) / 2 ... handled below
        const u128 tt = T0 + (u128)i * S0 + D * (u128)(i ? m : 0); // modular arithmetic == signed D
        const uint64_t frac = (uint64_t)tt;
        n0 = (int64_t)(uint64_t)(tt >> 64) - (H - 1);
        iv = ia.lgP ? frac >> (64 - ia.lgP) : 0;
        xq = (frac << ia.lgP) >> (64 - SH);
    } else {
        const int64_t t = a.p0 + idx * a.M;
        const int64_t q = t / a.L;
        const uint64_t r = (uint64_t)(t - q * a.L);
        const uint64_t tp = r * (uint64_t)ia.P, rem = tp % (uint64_t)a.L;
        iv = tp / (uint64_t)a.L;
        xq = (rem << SH) / (uint64_t)a.L;
        n0 = a.d0 + q - (H - 1);
    }
    const Real xx = (Real)xq * (Real)(1. / (double)(1ULL << SH));
    const int64_t loc0 = n0 - a.in_abs0;
    const IO *xin = (const IO *)a.in + (int64_t)clip * a.ics + (int64_t)ch * a.ichs;
    const V4 *c = (const V4 *)ia.tab + (int64_t)iv * T;
    Real accL = 0, accR = 0;
    auto coef = [&](int j) -> Real {
        const V4 v = c[j];
        return fma_r(fma_r(fma_r(v.w, xx, v.z), xx, v.y), xx, v.x);
    };
    if (loc0 >= 0 && loc0 + T <= a.in_frames) {
        const IO *xp = xin + loc0 * a.ifs;
        for (int j = 0; j < H; ++j) accL = fma_r(coef(j), (Real)xp[(int64_t)j * a.ifs], accL);
        for (int j = T - 1; j >= H; --j) accR = fma_r(coef(j), (Real)xp[(int64_t)j * a.ifs], accR);
    } else {
        for (int j = 0; j < H; ++j) {
            int64_t l = loc0 + j;
            Real xv = (l >= 0 && l < a.in_frames) ? (Real)xin[l * a.ifs] : (Real)0;
            accL = fma_r(coef(j), xv, accL);
        }
        for (int j = T - 1; j >= H; --j) {
            int64_t l = loc0 + j;
            Real xv = (l >= 0 && l < a.in_frames) ? (Real)xin[l * a.ifs] : (Real)0;
            accR = fma_r(coef(j), xv, accR);
        }
    }
    IO *yo = (IO *)a.out + (int64_t)clip * a.ocs + idx * a.ofs + (int64_t)ch * a.ochs;
    store_out<Real>(yo, accL + accR, a.oc, ch, a.out_k0 + idx);
}

// the value lane K of the quad holds, in all four of its lanes (DPP quad_perm)
template <int K> __device__ __forceinline__ float quad_bcast_f(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), K * 0x55, 0xf, 0xf, true));
}
template <int K> __device__ __forceinline__ double quad_bcast_f(double v)
{
    const uint64_t u = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)u, K * 0x55, 0xf, 0xf, true);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)(u >> 32), K * 0x55, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}

// ---------------------------------------------------------------------------------------------
// k_gather_wave — exact-bank jobs too small to fill the chip with period tiles (a stream's 96 000-frame chunk, 1 s clips)
// ---------------------------------------------------------------------------------------------
// The tile kernels share a phase's coefficients among the periods of a slab: 96 000 frames at 44.1k -> 16k are 218 periods
// = 14 sixteen-period slabs, 14 workgroups on 256 CUs (15.7 us).  k_interp_wave's shape needs no sharing to fill the
// chip: a half-chain per QUAD of lanes, here with the phase's own row of the phase-major bank [L][T] — lane k holds taps
// 16 s + 4 k .. + 3 of step s as one 16-byte (float) / 32-byte (double) load, a quad reads 64 / 128 contiguous bytes
// per step, requested 8 steps ahead; the chain takes the sixteen coefficients in canonical order by DPP.  Same
// arithmetic per output as k_gather and the tile kernels: bit-identical.
struct GatherWaveArgs {
    GatherArgs g;            // .bank unused
    const void *phase_major; // [L][T] Real
    int32_t span_cap;        // staged samples per workgroup (>= 31 window shifts + T)
    uint32_t *done_words;    // (optional) completion words, as ChainArgs::done_words
    uint32_t done_seq;
};

template <typename IO, typename Real>
__global__ void __launch_bounds__(256) k_gather_wave(GatherWaveArgs wa)
{
    typedef typename Vec4<Real>::type V4;
    constexpr int U = 8; // steps (of sixteen taps) requested ahead
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ int64_t s_loc[4];
    const GatherArgs &a = wa.g;
    const int lane = threadIdx.x & 63, k = lane & 3, quad = lane >> 2;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), half = wave & 1, grp = wave >> 1;
    Real *xs = reinterpret_cast<Real *>(smem_raw);
    Real *accx = xs + wa.span_cap; // [32]
    const int32_t T = a.T, H = T / 2, NS = (H + 15) / 16; // T is a multiple of 8: H of 4
    const uint32_t col = blockIdx.y;
    const uint32_t ch = __builtin_amdgcn_readfirstlane(col % a.n_channels), clip = __builtin_amdgcn_readfirstlane(col / a.n_channels);
    const IO *xin = (const IO *)a.in + (int64_t)clip * a.ics + (int64_t)ch * a.ichs;

    const int64_t o = (int64_t)blockIdx.x * 32 + grp * 16 + quad;
    const int64_t oc = o < a.out_frames ? o : a.out_frames - 1;
    const int64_t t = a.p0 + oc * a.M, q = t / a.L, ph = t - q * a.L; // (out_k0 + o) * M = L * (d0 + q) + ph
    const int64_t loc0 = a.d0 + q - (H - 1) - a.in_abs0;
    if (half == 0 && (lane == 0 || lane == 63)) s_loc[grp * 2 + (lane ? 1 : 0)] = loc0;
    __syncthreads();
    const int64_t base = s_loc[0];
    int32_t span = (int32_t)(s_loc[3] - base) + T;
    if (span > wa.span_cap) span = wa.span_cap; // (never: the host sized span_cap from M / L)
    const int32_t rel = (int32_t)(loc0 - base);
    for (int m = (int)threadIdx.x; m < span; m += 256) {
        const int64_t l = base + m;
        xs[m] = (l >= 0 && l < a.in_frames) ? (Real)xin[l * a.ifs] : (Real)0;
    }
    __syncthreads();

    const Real *row = (const Real *)wa.phase_major + ph * T;
    Real acc = 0;
    auto run = [&](auto half_c) {
        constexpr bool HALF = decltype(half_c)::value;
        // step s: taps 16 s .. 16 s + 15 of the first half-chain (upwards), T-1-16 s .. T-16-16 s of the second
        // (downwards); lane k holds four of them, ascending in memory either way.  Loads are unconditional and clamped
        // to the last step (a load under a condition is waited for at once: k_interp_wave).  A last step of fewer than
        // sixteen taps reads past the half-chain, inside the row (T >= 32).
        auto at = [&](int s_) {
            const Real *p4 = row + (HALF ? T - 16 * (s_ + 1) : 16 * s_) + 4 * k;
            return *reinterpret_cast<const V4 *>(p4);
        };
        const Real *xp = xs + rel + (HALF ? T - 1 : 0);
        V4 r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) r[u] = at(u < NS ? u : NS - 1);
        auto four = [&](const V4 c, const Real *xq, auto lane_c) { // the four taps lane M holds, in chain order
            constexpr int M = decltype(lane_c)::value;
            if (!HALF) {
                acc = fma_r(quad_bcast_f<M>(c.x), xq[4 * M + 0], acc);
                acc = fma_r(quad_bcast_f<M>(c.y), xq[4 * M + 1], acc);
                acc = fma_r(quad_bcast_f<M>(c.z), xq[4 * M + 2], acc);
                acc = fma_r(quad_bcast_f<M>(c.w), xq[4 * M + 3], acc);
            } else { // xq points at the step's HIGHEST tap; lane M's taps sit 15 - 4 M - e below it
                acc = fma_r(quad_bcast_f<M>(c.w), xq[-(12 - 4 * M) - 0], acc);
                acc = fma_r(quad_bcast_f<M>(c.z), xq[-(12 - 4 * M) - 1], acc);
                acc = fma_r(quad_bcast_f<M>(c.y), xq[-(12 - 4 * M) - 2], acc);
                acc = fma_r(quad_bcast_f<M>(c.x), xq[-(12 - 4 * M) - 3], acc);
            }
        };
        auto chain = [&](const V4 c, int s_, int taps) { // taps: 16, or what is left of the half-chain in its last step
            const Real *xq = HALF ? xp - 16 * s_ : xp + 16 * s_;
            if (!HALF) {
                four(c, xq, std::integral_constant<int, 0>());
                if (taps > 4) four(c, xq, std::integral_constant<int, 1>());
                if (taps > 8) four(c, xq, std::integral_constant<int, 2>());
                if (taps > 12) four(c, xq, std::integral_constant<int, 3>());
            } else {
                four(c, xq, std::integral_constant<int, 3>());
                if (taps > 4) four(c, xq, std::integral_constant<int, 2>());
                if (taps > 8) four(c, xq, std::integral_constant<int, 1>());
                if (taps > 12) four(c, xq, std::integral_constant<int, 0>());
            }
        };
        const int full = H / 16; // steps of sixteen taps; a shorter last one follows when H is not a multiple of 16
        int s0 = 0;
        for (; s0 + 2 * U <= full; s0 += U) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const V4 v = r[u];
                r[u] = at(s0 + u + U);
                chain(v, s0 + u, 16);
            }
        }
        for (; s0 + U <= full; s0 += U) { // (the look-ahead reaches the end: clamped)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const V4 v = r[u];
                const int sn = s0 + u + U;
                r[u] = at(sn < NS ? sn : NS - 1);
                chain(v, s0 + u, 16);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { // fewer than U steps left, already requested; the last may be short
            const int s_ = s0 + u;
            if (s_ < full) chain(r[u], s_, 16);
            else if (s_ < NS) chain(r[u], s_, H - 16 * full);
        }
    };
    if (half) run(std::integral_constant<bool, true>());
    else run(std::integral_constant<bool, false>());
    if (half && k == 0) accx[grp * 16 + quad] = acc;
    __syncthreads();
    if (!half && k == 0 && o < a.out_frames) {
        IO *yo = (IO *)a.out + (int64_t)clip * a.ocs + o * a.ofs + (int64_t)ch * a.ochs;
        store_out<Real>(yo, acc + accx[grp * 16 + quad], a.oc, ch, a.out_k0 + o);
    }
    if (wa.done_words) {
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0)
            __hip_atomic_store(&wa.done_words[blockIdx.y * gridDim.x + blockIdx.x], wa.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---------------------------------------------------------------------------------------------
// k_interp_tile — throughput kernel for interpolated-phase plans and variable-rate launches
// ---------------------------------------------------------------------------------------------
// k_interp is bound by the texture-address path: the 64 lanes of a wave sit in 64 different phase
// intervals, so every tap fetches 64 different 16-byte cubic records (1.9 Gsamples/s at VHQ).
// Here a workgroup takes KO consecutive outputs of one column, stages their input span in LDS, and
// SORTS the outputs by phase interval (counting sort in LDS).  A wave then processes outputs of ONE
// interval at a time: the interval's cubic records are wave-uniform (one broadcast load per tap
// instead of 64 scattered ones), each lane reads its own input window from LDS.
// Same canonical arithmetic per output as k_interp / the oracle, so results stay bit-identical.
struct InterpTileArgs {
    InterpArgs ia;
    int32_t KO;        // outputs per workgroup
    int32_t span_cap;  // staged input samples (>= span of any workgroup)
    // PAIR instances: a lane carries TWO outputs that share position, interval and cubic argument — the neighbouring channel
    // (ch + 1), or the same column h periods of L outputs further on (output k + h L sits exactly h M input samples behind
    // output k with the same remainder) — so the interval's records stream through the scalar cache once for both and the
    // cubic per tap is evaluated once; each member's own FMA chain is untouched (bit-identical results).
    uint32_t cols_per_clip, ch_step; // column -> (clip, first channel): col / cols_per_clip, (col % cols_per_clip) * ch_step
    int64_t m2_in, m2_out;           // member 2: element offsets of its input frame l / output k from member 1's
    int64_t m2_l, m2_k, m2_n;        // ... its input frame = l + m2_l, its output index = k + m2_k, and how many outputs it has
    int32_t m2_dch;                  // ... its channel = ch + m2_dch (dither / clip-counter context)
};

template <typename Real> struct InterpPos { int64_t n0; uint32_t iv; uint64_t xq; };

template <typename Real, bool VR>
__device__ __forceinline__ InterpPos<Real> interp_locate(const InterpArgs &ia, int64_t idx)
{
    constexpr int SH = sizeof(Real) == 4 ? 24 : 32;
    const GatherArgs &a = ia.g;
    InterpPos<Real> r;
    const int32_t H = a.T / 2;
    if (VR) {
        typedef unsigned __int128 u128;
        const u128 T0 = ((u128)ia.t_hi << 64) | ia.t_lo, S0 = ((u128)ia.s_hi << 64) | ia.s_lo,
                   D = ((u128)ia.d_hi << 64) | ia.d_lo;
        const uint64_t i = (uint64_t)idx, m = i * (i - 1) / 2;
        const u128 tt = T0 + (u128)i * S0 + D * (u128)(i ? m : 0);
        const uint64_t frac = (uint64_t)tt;
        r.n0 = (int64_t)(uint64_t)(tt >> 64) - (H - 1);
        r.iv = ia.lgP ? (uint32_t)(frac >> (64 - ia.lgP)) : 0u;
        r.xq = (frac << ia.lgP) >> (64 - SH);
    } else {
        const int64_t t = a.p0 + idx * a.M;
        const int64_t q = t / a.L;
        const uint64_t rr = (uint64_t)(t - q * a.L);
        const uint64_t tp = rr * (uint64_t)ia.P, rem = tp % (uint64_t)a.L;
        r.iv = (uint32_t)(tp / (uint64_t)a.L);
        r.xq = (rem << SH) / (uint64_t)a.L;
        r.n0 = a.d0 + q - (H - 1);
    }
    return r;
}

// floor(t / L) and t mod L for 0 <= t < 2^51, 0 < L < 2^31, through one double-precision multiply
// and a +-1 correction (exact: the estimate is off by at most one).  Integer division proper costs
// ~80 VALU instructions on this hardware and the tile kernel needs three per output.
__device__ __forceinline__ uint64_t divmod_small(uint64_t t, uint32_t L, double invL, uint32_t *rem)
{
    uint64_t q = (uint64_t)((double)t * invL);
    int64_t r = (int64_t)(t - q * (uint64_t)L);
    if (r < 0) { --q; r += L; }
    else if (r >= (int64_t)L) { ++q; r -= L; }
    *rem = (uint32_t)r;
    return q;
}

// interp_locate for local output i of a workgroup whose first output sits at (q_base, r_base):
// (k_base + i) * M = L * (q_base + q) + r  with  r_base + i*M = L*q + r,  i*M < 2^45.
template <typename Real>
__device__ __forceinline__ InterpPos<Real> interp_locate_local(const InterpArgs &ia, int64_t n0_base, uint32_t r_base,
                                                               double invL, int32_t i)
{
    constexpr int SH = sizeof(Real) == 4 ? 24 : 32;
    const uint32_t L = (uint32_t)ia.g.L;
    InterpPos<Real> p;
    uint32_t r, rem;
    const uint64_t q = divmod_small((uint64_t)r_base + (uint64_t)i * (uint64_t)ia.g.M, L, invL, &r);
    p.n0 = n0_base + (int64_t)q;
    p.iv = (uint32_t)divmod_small((uint64_t)r * (uint64_t)ia.P, L, invL, &rem);
    // floor(rem * 2^SH / L) by long division in two digits of SH/2 bits (each dividend < 2^47)
    uint32_t rem2;
    const uint64_t hi = divmod_small((uint64_t)rem << (SH / 2), L, invL, &rem2);
    const uint64_t lo = divmod_small((uint64_t)rem2 << (SH / 2), L, invL, &rem);
    p.xq = (hi << (SH / 2)) | lo;
    return p;
}

// ---------------------------------------------------------------------------------------------
// k_interp_wave — mid-size interpolated-phase and variable-rate launches (a stream's 96 000-frame chunk)
// ---------------------------------------------------------------------------------------------
// k_interp gives every output one lane: 64 lanes in 64 different phase intervals fetch 64 different 16-byte cubic
// records per tap (each pulling a 128-byte line through the texture path for 16 bytes of use), a chunk of 35 000
// outputs is one wave per SIMD at best, and every wave walks its T taps through ~T/8 serialised round trips to the L2:
// 227 us for 34 830 outputs x 736 taps (44.1k -> 16k VHQ, variable rate).  Here a half-chain — the canonical order has
// exactly two per output — is a QUAD of lanes:
//   * lane k of the quad fetches the record of tap 4s + k of step s and evaluates its cubic: a quad reads 64 contiguous
//     bytes of its row per step (128 in float64), a wave 16 such runs — no over-fetch, no transposition, a quarter of
//     the cubic arithmetic per lane, and eight times the waves of k_interp (16 half-chains per wave instead of 64
//     outputs), each a quarter as long;
//   * the chain itself — the only serial part — takes the four coefficients in order out of the quad's lanes by DPP
//     (`quad_perm` broadcast, folded into v_fmac_f32_dpp where the compiler can): acc = fma(c_k, x, acc), k = 0..3,
//     computed by all four lanes alike;
//   * records are requested U steps ahead (a register is refilled as soon as its cubic is taken);
//   * the input span of a workgroup's 32 consecutive outputs (<= 31 steps + T samples) is staged once in LDS, converted,
//     zero-extended; a quad's four samples per step are one broadcast LDS read.
// Per output the arithmetic is k_interp's to the letter (cubic by three fma, then the chain fma, accL + accR), so results
// are bit-identical to it and to the oracle, however the outputs spread over the phase intervals (a constant step of
// exactly 2.0 puts every output in ONE interval, a generic step in all of them).
struct InterpWaveArgs {
    InterpArgs ia;
    int32_t span_cap; // staged samples per workgroup (>= 31 steps + T)
    uint32_t *done_words; // (optional, pinned host memory) completion words, as ChainArgs::done_words
    uint32_t done_seq;
};

template <typename IO, typename Real, bool VR>
__global__ void __launch_bounds__(256) k_interp_wave(InterpWaveArgs wa)
{
    typedef typename Vec4<Real>::type V4;
    constexpr int U = 8; // steps (of four taps) requested ahead
    constexpr int SH = sizeof(Real) == 4 ? 24 : 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ int64_t s_loc[4]; // first / last window start of the two output groups
    const InterpArgs &ia = wa.ia;
    const GatherArgs &a = ia.g;
    const int lane = threadIdx.x & 63, k = lane & 3, quad = lane >> 2;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), half = wave & 1, grp = wave >> 1;
    Real *xs = reinterpret_cast<Real *>(smem_raw);
    Real *accx = xs + wa.span_cap; // [32] the second half-chains' sums
    const int32_t T = a.T, H = T / 2, NS = H / 4; // T is a multiple of 8
    const uint32_t col = blockIdx.y;
    const uint32_t ch = __builtin_amdgcn_readfirstlane(col % a.n_channels), clip = __builtin_amdgcn_readfirstlane(col / a.n_channels);
    const IO *xin = (const IO *)a.in + (int64_t)clip * a.ics + (int64_t)ch * a.ichs;

    const int64_t o = (int64_t)blockIdx.x * 32 + grp * 16 + quad;
    const int64_t oc = o < a.out_frames ? o : a.out_frames - 1; // (quads past the end repeat the last output and store nothing)
    const InterpPos<Real> pos = interp_locate<Real, VR>(ia, oc);
    const Real xx = (Real)pos.xq * (Real)(1. / (double)(1ULL << SH));
    const int64_t loc0 = pos.n0 - a.in_abs0;
    // positions grow with the output index: the first quad of group 0 holds the span's first sample, the last quad of group 1 its last window
    if (half == 0 && (lane == 0 || lane == 63)) s_loc[grp * 2 + (lane ? 1 : 0)] = loc0;
    __syncthreads();
    const int64_t base = s_loc[0];
    int32_t span = (int32_t)(s_loc[3] - base) + T;
    if (span > wa.span_cap) span = wa.span_cap; // (never: the host sized span_cap from the launch's largest step)
    const int32_t rel = (int32_t)(loc0 - base);
    for (int m = (int)threadIdx.x; m < span; m += 256) {
        const int64_t l = base + m;
        xs[m] = (l >= 0 && l < a.in_frames) ? (Real)xin[l * a.ifs] : (Real)0;
    }
    __syncthreads();

    const unsigned char *row = (const unsigned char *)ia.tab + (size_t)pos.iv * (size_t)T * sizeof(V4);
    Real acc = 0;
    auto run = [&](auto half_c) {
        constexpr bool HALF = decltype(half_c)::value;
        // step s: taps 4s .. 4s+3 of the first half-chain (upwards), T-1-4s .. T-4-4s of the second (downwards);
        // lane k holds tap 4s + k / T-4-4s + k — ascending in memory either way
        const V4 *rp = reinterpret_cast<const V4 *>(row) + (HALF ? T - 4 + k : k);
        const Real *xp = xs + rel + (HALF ? T - 1 : 0);
        // (every load below is unconditional — a load under a condition merges with the register's old value, and the
        //  copy that merge needs waits for the load at once: 380 cycles per step, measured — so indices are clamped
        //  to the last step instead, and the loop is cut where the look-ahead reaches the end)
        auto at = [&](int s_) { return rp[HALF ? -4 * s_ : 4 * s_]; };
        V4 r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) r[u] = at(u < NS ? u : NS - 1);
        auto chain = [&](const V4 v, int s_) {
            const Real c = fma_r(fma_r(fma_r(v.w, xx, v.z), xx, v.y), xx, v.x);
            const Real *xq = HALF ? xp - 4 * s_ : xp + 4 * s_;
            if (!HALF) {
                acc = fma_r(quad_bcast_f<0>(c), xq[0], acc);
                acc = fma_r(quad_bcast_f<1>(c), xq[1], acc);
                acc = fma_r(quad_bcast_f<2>(c), xq[2], acc);
                acc = fma_r(quad_bcast_f<3>(c), xq[3], acc);
            } else {
                acc = fma_r(quad_bcast_f<3>(c), xq[0], acc);
                acc = fma_r(quad_bcast_f<2>(c), xq[-1], acc);
                acc = fma_r(quad_bcast_f<1>(c), xq[-2], acc);
                acc = fma_r(quad_bcast_f<0>(c), xq[-3], acc);
            }
        };
        int s0 = 0;
        for (; s0 + 2 * U <= NS; s0 += U) { // the look-ahead stays inside the half-chain: immediate offsets
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const V4 v = r[u];
                r[u] = at(s0 + u + U);
                chain(v, s0 + u);
            }
        }
        if (s0 + U <= NS) { // the last full group: its look-ahead is the tail (clamped)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const V4 v = r[u];
                const int sn = s0 + u + U;
                r[u] = at(sn < NS ? sn : NS - 1);
                chain(v, s0 + u);
            }
            s0 += U;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) // the tail: fewer than U steps, already requested
            if (s0 + u < NS) chain(r[u], s0 + u);
    };
    if (half) run(std::integral_constant<bool, true>());
    else run(std::integral_constant<bool, false>());
    if (half && k == 0) accx[grp * 16 + quad] = acc;
    __syncthreads();
    if (!half && k == 0 && o < a.out_frames) {
        IO *yo = (IO *)a.out + (int64_t)clip * a.ocs + o * a.ofs + (int64_t)ch * a.ochs;
        store_out<Real>(yo, acc + accx[grp * 16 + quad], a.oc, ch, a.out_k0 + o);
    }
    if (wa.done_words) {
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0)
            __hip_atomic_store(&wa.done_words[blockIdx.y * gridDim.x + blockIdx.x], wa.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

template <typename IO, typename Real, bool VR, bool PAIR, bool TWIN = false>
__global__ void __launch_bounds__(1024) k_interp_tile(InterpTileArgs ta)
{
    static_assert(!TWIN || (PAIR && sizeof(Real) == 4), "TWIN: float pairs only");
    constexpr int NM = PAIR ? 2 : 1; // members per lane; the staged span is [sample][member]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const InterpArgs &ia = ta.ia;
    const GatherArgs &a = ia.g;
    constexpr int SH = sizeof(Real) == 4 ? 24 : 32;
    const int32_t KO = ta.KO, P = ia.P, T = a.T, H = T / 2;
    // (round 3: no per-output record in LDS any more — 8 of the 10 bytes of bookkeeping per output; an output's position
    //  is located again where it is needed, ~50 vector-ALU instructions against the ~1500 of its taps.  LDS then holds
    //  twice the outputs per workgroup, a bucket — the outputs of one phase interval, served 64 at a time — 60 instead
    //  of 30.  48000 -> 44101: mono 60 s 254 -> 242 us, 200 000 frames 152 -> 114; stereo 60 s stays at 402.  The kernel issues
    //  one vector-ALU instruction per 7 cycles per SIMD, and it is not the LDS: with every lane reading lane 0's window — no
    //  bank conflict left — it takes 382 us.  Time goes with the NUMBER OF GROUPS, whatever the occupancy (30 outputs per
    //  interval, three workgroups per CU: 628 us; 15: 1013): a group of <= 64 outputs walks its interval's whole row of
    //  cubic records, 4.8 KB, through the scalar cache, which it misses — 423 MB per launch, ~6 bytes per cycle per scalar
    //  cache.  Coefficient delivery is the bound; requesting a block ahead (one block is all the SGPRs hold) was slower.)
    Real *xs = reinterpret_cast<Real *>(smem_raw);                       // [span_cap][NM]
    // float pairs: the span TWICE, the second copy one sample further on — a lane reads the copy in which its window starts
    // 16-byte aligned, two taps x two members per ds_read_b128 (256 B/clk) instead of one tap per half of a ds_read2_b64 (128)
    // (TWIN; where two copies leave too few outputs per workgroup — long steps, long filters — the pair runs on one)
    constexpr int NCOPY = TWIN ? 2 : 1;
    Real *xsB = xs + (size_t)(ta.span_cap + 2) * NM; // (span_cap is even: 16-byte aligned)
    uint16_t *order = reinterpret_cast<uint16_t *>(xs + (size_t)(NCOPY == 2 ? 2 * (ta.span_cap + 2) : ta.span_cap) * NM); // [KO]  outputs sorted by interval
    uint32_t *off = reinterpret_cast<uint32_t *>(order + ((KO + 1) & ~1)); // [P + 1] bucket offsets
    uint32_t *cur = off + (P + 1);                                       // [P]     scatter cursors

    const uint32_t col = blockIdx.y;
    // run-time division goes through the vector ALU; readfirstlane keeps the results (and every
    // address derived from them) on the scalar side
    const uint32_t ch = __builtin_amdgcn_readfirstlane((col % ta.cols_per_clip) * ta.ch_step), clip = __builtin_amdgcn_readfirstlane(col / ta.cols_per_clip);
    const int64_t o_base = (int64_t)blockIdx.x * KO;
    const int32_t n_here = (int32_t)((a.out_frames - o_base) < KO ? (a.out_frames - o_base) : KO);
    const IO *xin = (const IO *)a.in + (int64_t)clip * a.ics + (int64_t)ch * a.ichs;
    IO *yo = (IO *)a.out + (int64_t)clip * a.ocs + (int64_t)ch * a.ochs;

    // span of inputs this workgroup needs (positions are monotonic in the output index)
    const int64_t n_first = interp_locate<Real, VR>(ia, o_base).n0;
    const int64_t n_end = interp_locate<Real, VR>(ia, o_base + n_here - 1).n0 + T;
    const int32_t span = (int32_t)(n_end - n_first);
    // rational mode: position of the workgroup's first output, then cheap local arithmetic
    uint32_t r_base = 0;
    double invL = 0.;
    if (!VR) {
        const int64_t t = a.p0 + o_base * a.M;
        r_base = (uint32_t)(t - (t / a.L) * a.L);
        invL = 1. / (double)a.L;
    }
    auto locate = [&](int i) -> InterpPos<Real> {
        if (VR) return interp_locate<Real, VR>(ia, o_base + i);
        return interp_locate_local<Real>(ia, n_first, r_base, invL, i);
    };

    for (int i = threadIdx.x; i <= 2 * P; i += blockDim.x) off[i] = 0; // off[0..P] and cur[0..P-1] are contiguous
    __syncthreads();
    // 1. locate every output once; histogram of intervals
    for (int i = threadIdx.x; i < n_here; i += blockDim.x) {
        const InterpPos<Real> r = locate(i);
        atomicAdd(&off[r.iv + 1], 1u);
    }
    // 2. stage the input span (zero outside the signal), converted to the engine precision
    for (int m = threadIdx.x; m < span; m += blockDim.x) {
        const int64_t l = n_first + m - a.in_abs0;
        const Real v1 = (l >= 0 && l < a.in_frames) ? (Real)xin[l * a.ifs] : (Real)0;
        xs[NM * m] = v1;
        if constexpr (PAIR) {
            const Real v2 = (l + ta.m2_l >= 0 && l + ta.m2_l < a.in_frames) ? (Real)xin[ta.m2_in + l * a.ifs] : (Real)0;
            xs[NM * m + 1] = v2;
            if constexpr (NCOPY == 2) { xsB[NM * (m + 1)] = v1; xsB[NM * (m + 1) + 1] = v2; }
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) { // inclusive scan of off[1..P] (P <= 256 = 64 lanes x 4) by the first wave
        const int l = threadIdx.x;
        uint32_t v[4], sum = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = (4 * l + e < P) ? off[1 + 4 * l + e] : 0u; sum += v[e]; }
        uint32_t incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d, 64);
            if (l >= d) incl += up;
        }
        uint32_t run = incl - sum;
#pragma unroll
        for (int e = 0; e < 4; ++e) { run += v[e]; if (4 * l + e < P) off[1 + 4 * l + e] = run; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_here; i += blockDim.x) {
        const uint32_t iv = locate(i).iv;
        order[off[iv] + atomicAdd(&cur[iv], 1u)] = (uint16_t)i;
    }
    __syncthreads();

    // 3. one interval at a time per wave.  The interval's cubic records are the same for all 64
    //    lanes, so they must not go through the vector memory path (a lane-uniform
    //    global_load_dwordx4 still costs 64 x 16 bytes of texture-address bandwidth: measured
    //    TA-bound at 640 us) — they are read four taps at a time with one scalar s_load_dwordx16
    //    (wave-uniform pointer in the constant address space) and used as SGPR operands.
    const int lane = threadIdx.x & 63;
    const int n_waves = blockDim.x >> 6;
    typedef Real RealX16 __attribute__((ext_vector_type(16)));
    typedef const __attribute__((address_space(4))) RealX16 *CPtr16;
    for (int iv_ = threadIdx.x >> 6; iv_ < P; iv_ += n_waves) {
        const int iv = __builtin_amdgcn_readfirstlane(iv_);
        const uint32_t b0 = __builtin_amdgcn_readfirstlane(off[iv]), b1 = __builtin_amdgcn_readfirstlane(off[iv + 1]);
        CPtr16 row = (CPtr16)((const Real *)ia.tab + (size_t)iv * T * 4); // row[b] = taps 4b .. 4b+3
        for (uint32_t g = b0; g < b1; g += 64) {
            // the 64 outputs of this group, sorted by index across the lanes (bitonic, in registers):
            // consecutive lanes then read input windows a near-constant distance apart, which keeps
            // the per-tap ds_read_b32 spread over the LDS banks (the counting sort scatters within a
            // bucket in arrival order)
            uint32_t key = g + lane < b1 ? order[g + lane] : 0xFFFFu;
#pragma unroll
            for (int k = 2; k <= 64; k <<= 1)
#pragma unroll
                for (int jj = k >> 1; jj > 0; jj >>= 1) {
                    const uint32_t other = __shfl_xor(key, jj, 64);
                    const bool take_min = ((lane & k) == 0) == ((lane & jj) == 0);
                    key = take_min ? (key < other ? key : other) : (key > other ? key : other);
                }
            const bool live = key != 0xFFFFu;
            const int i = live ? (int)key : (int)order[b0];
            const InterpPos<Real> rc = locate(i);
            const uint32_t m_first = (uint32_t)(rc.n0 - n_first);
            const Real *xl = (NCOPY == 2 && (m_first & 1)) ? xsB + NM * (m_first + 1) : xs + NM * m_first;
            const Real xx = (Real)(uint32_t)rc.xq * (Real)(1. / (double)(1ULL << SH));
            Real accL = 0, accR = 0, accL2 = 0, accR2 = 0;
            // (one tap: the canonical cubic, then each member's own chain FMA)
            // (PAIR: the two members' samples in ONE 8- / 16-byte LDS read — separate 4-byte reads at a stride of two words
            //  would use every other bank)
            typedef Real RealX2 __attribute__((ext_vector_type(2)));
#define HIPSOXR_ITILE_TAP(c0, c1, c2, c3, t, L, L2)                                   \
    {                                                                                \
        const Real cj = fma_r(fma_r(fma_r(c3, xx, c2), xx, c1), xx, c0);             \
        if constexpr (NCOPY == 2) {                                                  \
            L = fma_r(cj, xq[(t) >> 1][2 * ((t) & 1)], L);                           \
            L2 = fma_r(cj, xq[(t) >> 1][2 * ((t) & 1) + 1], L2);                     \
        } else if constexpr (PAIR) {                                                 \
            const RealX2 xv = reinterpret_cast<const RealX2 *>(x4)[t];               \
            L = fma_r(cj, xv.x, L);                                                  \
            L2 = fma_r(cj, xv.y, L2);                                                \
        } else                                                                       \
            L = fma_r(cj, x4[t], L);                                                 \
    }
            typedef Real RealX4 __attribute__((ext_vector_type(4)));
#define HIPSOXR_ITILE_QUADS                                                                                              \
    RealX4 xq[2];                                                                                                        \
    if constexpr (NCOPY == 2) {                                                                                          \
        xq[0] = *reinterpret_cast<const RealX4 *>(__builtin_assume_aligned(x4, 16));                                     \
        xq[1] = *reinterpret_cast<const RealX4 *>(__builtin_assume_aligned(x4 + 4, 16));                                 \
    }                                                                                                                    \
    (void)xq;
#pragma unroll 2
            for (int b = 0; b < H / 4; ++b) { // T is a multiple of 8: H is a multiple of 4
                const RealX16 c = row[b];
                const Real *x4 = xl + NM * 4 * b;
                HIPSOXR_ITILE_QUADS
                HIPSOXR_ITILE_TAP(c[0], c[1], c[2], c[3], 0, accL, accL2)
                HIPSOXR_ITILE_TAP(c[4], c[5], c[6], c[7], 1, accL, accL2)
                HIPSOXR_ITILE_TAP(c[8], c[9], c[10], c[11], 2, accL, accL2)
                HIPSOXR_ITILE_TAP(c[12], c[13], c[14], c[15], 3, accL, accL2)
            }
#pragma unroll 2
            for (int b = T / 4 - 1; b >= H / 4; --b) { // descending taps
                const RealX16 c = row[b];
                const Real *x4 = xl + NM * 4 * b;
                HIPSOXR_ITILE_QUADS
                HIPSOXR_ITILE_TAP(c[12], c[13], c[14], c[15], 3, accR, accR2)
                HIPSOXR_ITILE_TAP(c[8], c[9], c[10], c[11], 2, accR, accR2)
                HIPSOXR_ITILE_TAP(c[4], c[5], c[6], c[7], 1, accR, accR2)
                HIPSOXR_ITILE_TAP(c[0], c[1], c[2], c[3], 0, accR, accR2)
            }
#undef HIPSOXR_ITILE_TAP
#undef HIPSOXR_ITILE_QUADS
            if (live) {
                const int64_t idx = o_base + i;
                store_out<Real>(yo + idx * a.ofs, accL + accR, a.oc, ch, a.out_k0 + idx);
                if constexpr (PAIR)
                    if (idx < ta.m2_n) store_out<Real>(yo + ta.m2_out + idx * a.ofs, accL2 + accR2, a.oc, ch + ta.m2_dch, a.out_k0 + idx + ta.m2_k);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_chain — low-latency kernel for SMALL launches (streaming chunks: tens to a few thousand outputs)
// ---------------------------------------------------------------------------------------------
// k_gather's cost on a small launch is pure latency: every lane walks T taps, each a pair of L2 loads
// feeding a dependent FMA (81 us for T = 736, whatever the chunk size).  Here a workgroup of 256
// threads takes NO consecutive outputs: ALL threads first stage the operands into LDS — the NO
// coefficient rows, and ONCE the input span the NO windows share (consecutive windows are shifted by
// M/L samples: 8 windows of 736 taps are 756 distinct samples, not 5888) — with every load of the
// workgroup in flight at once (one round trip for T <= 768, not T); then 2*NO lanes run the canonical
// half-chains out of LDS (lane o: left half of output o, lane NO+o: right half), four taps per
// 16-byte coefficient read, and the two halves are added.  Same arithmetic, bit for bit.
// The input may be pinned host memory (small-chunk streams keep their ring there, engine.cpp): the span
// is then the only thing that crosses PCIe, once.
// MODE 0: exact bank (phase-major [L][T]); 1: interpolated-phase plan; 2: variable rate.  In the
// interpolated modes the staging thread evaluates the tap's cubic (the canonical Horner FMAs).
struct ChainArgs {
    InterpArgs ia;           // .g: job geometry; .tab/.P/...: interpolated plans
    const void *phase_major; // exact plans: [L][T] Real
    int32_t NO;              // outputs per workgroup (power of two, <= 32)
    int32_t span_cap;        // LDS room for the shared input span, in samples
    uint32_t *done_words;    // (optional, pinned host memory) workgroup w stores done_seq into done_words[w] once its
    uint32_t done_seq;       //  results are in host memory: the host polls these instead of an event (ChainDone)
};

// what changes from one launch (or one message to the resident form, below) to the next
struct ChainMsg { int64_t in_abs0, in_frames, out_k0, out_frames, d0, p0; uint64_t t_hi, t_lo, s_hi, s_lo, d_hi, d_lo; /* MODE 2: the Q64.64 clock of this launch / message */ };

#ifndef HIPSOXR_RPW
#define HIPSOXR_RPW 2
#endif
struct NoHook { __device__ __forceinline__ void operator()() const {} };
// `pre` runs in every thread before the barrier in front of the output stores; outputs are withheld if *veto has its
// top bit set after that barrier (the resident form's arbiter, see k_chain_resident)
// chunk / split: input frames from ring-relative index `split` on are read from `chunk` (a stream's new frames, not yet in its
// ring: k_chain_multi) instead of the ring
template <typename IO, typename Real, int MODE, typename Pre = NoHook>
__device__ __forceinline__ void chain_body(const ChainArgs &ca, const ChainMsg &m, const uint32_t bx, const uint32_t by,
                                           unsigned char *smem_raw, uint32_t *trace = nullptr, Pre pre = Pre(),
                                           const unsigned long long *veto = nullptr, const void *chunk = nullptr, const int64_t split = 0)
{
#ifdef HIPSOXR_RES_TRACE
    const long long tb0 = wall_clock64();
#define HIPSOXR_CB_STAMP(k) do { if (trace && threadIdx.x == 0) trace[k] = (uint32_t)(wall_clock64() - tb0); } while (0)
#else
#define HIPSOXR_CB_STAMP(k) do { } while (0)
#endif
    InterpArgs ia = ca.ia;
    GatherArgs &a = ia.g;
    a.in_abs0 = m.in_abs0; a.in_frames = m.in_frames; a.out_k0 = m.out_k0; a.out_frames = m.out_frames; a.d0 = m.d0; a.p0 = m.p0;
    if (MODE == 2) { ia.t_hi = m.t_hi; ia.t_lo = m.t_lo; ia.s_hi = m.s_hi; ia.s_lo = m.s_lo; ia.d_hi = m.d_hi; ia.d_lo = m.d_lo; }
    constexpr int SH = sizeof(Real) == 4 ? 24 : 32;
    constexpr int V = 16 / (int)sizeof(Real);      // taps per 16-byte coefficient read: 4 (f32) or 2 (f64)
    const int32_t T = a.T, H = T / 2, NO = ca.NO, RS = T + V; // RS: row stride (rows 16-byte aligned, banks rotate by V per row)
    Real *cs = reinterpret_cast<Real *>(smem_raw); // [NO][RS] coefficients, row-major
    Real *xs = cs + (size_t)NO * RS;               // [span_cap] the input span shared by the NO windows
    int64_t *n0s = reinterpret_cast<int64_t *>(xs + ((ca.span_cap + 3) & ~3)); // [NO] first-tap input index (relative to in[0])
    uint64_t *aux = reinterpret_cast<uint64_t *>(n0s + NO);                    // [NO] phase (MODE 0) or iv<<32 | xq (MODE 1, 2)

    const uint32_t ch = by % a.n_channels, clip = by / a.n_channels;
    const int64_t o_base = (int64_t)bx * NO;
    const IO *xin = (const IO *)a.in + (int64_t)clip * a.ics + (int64_t)ch * a.ichs;
    const IO *xchunk = chunk ? (const IO *)chunk + (int64_t)ch * a.ichs - split * a.ifs : nullptr; // (indexed like the ring)
    auto sample_at = [&](int64_t l) -> const IO * { return (xchunk && l >= split) ? xchunk + l * a.ifs : xin + l * a.ifs; };
    typedef typename Vec4<Real>::type V4;

    if ((int)threadIdx.x < NO) { // one thread per output: where it sits
        const int64_t idx = o_base + threadIdx.x < a.out_frames ? o_base + threadIdx.x : a.out_frames - 1;
        if (MODE == 0) {
            const int64_t t = a.p0 + idx * a.M;
            int64_t q;
            uint32_t rem;
            if (a.L < (1LL << 31) && t < (1LL << 51)) { // (integer division proper: ~1 us of this kernel's latency)
                q = (int64_t)divmod_small((uint64_t)t, (uint32_t)a.L, 1. / (double)a.L, &rem);
            } else {
                q = t / a.L;
                rem = (uint32_t)(t - q * a.L); // (banks are L*T coefficients: L < 2^32)
            }
            n0s[threadIdx.x] = a.d0 + q - (H - 1) - a.in_abs0;
            aux[threadIdx.x] = (uint64_t)rem;
        } else {
            const InterpPos<Real> r = interp_locate<Real, MODE == 2>(ia, idx);
            n0s[threadIdx.x] = r.n0 - a.in_abs0;
            aux[threadIdx.x] = ((uint64_t)r.iv << 32) | (uint64_t)(uint32_t)r.xq; // xq < 2^32
        }
    }
    __syncthreads();
    HIPSOXR_CB_STAMP(0);
    // ---- stage.  Wave w takes coefficient rows w, w+4, ...; a lane takes taps lane, lane+64, ... of a row (no
    //      run-time division in the index arithmetic: that alone was a quarter of this kernel), EPT taps per
    //      trip, loads first, RPW rows at a time.  The input span (SPT samples per thread) is requested AFTER the
    //      first trip's coefficients and stored after them: when the ring lives in host memory its loads are a
    //      PCIe round trip (2-3.5 us), and loads return in order — requested first, they held every coefficient
    //      behind them (5.0-5.7 us for the whole staging; this way 3.4-4.8 us).
    const int64_t nfirst = n0s[0];
    const int32_t span = (int32_t)(n0s[NO - 1] - nfirst) + T; // windows are ordered: n0 is non-decreasing in o
    constexpr int SPT = 4;
    IO xv[SPT];
    bool span_loaded = false, span_stored = false;
    auto load_span = [&]() {
        if (span_loaded) return;
        span_loaded = true;
        // (unconditional loads from clamped addresses, zeroed afterwards: a load under a per-lane condition is a branch
        //  and a conservative wait each, and the compiler then serialises what should be one round trip)
#pragma unroll
        for (int u = 0; u < SPT; ++u) xv[u] = 0;
        if (a.in_frames > 0) {
#pragma unroll
            for (int u = 0; u < SPT; ++u) {
                const int64_t l = nfirst + (int32_t)threadIdx.x + u * 256;
                const int64_t lc = l < 0 ? 0 : l >= a.in_frames ? a.in_frames - 1 : l;
                const IO v = *sample_at(lc);
                xv[u] = (l == lc) ? v : (IO)0;
            }
        }
    };
    auto store_span = [&]() {
        if (span_stored) return;
        span_stored = true;
#pragma unroll
        for (int u = 0; u < SPT; ++u)
            if ((int32_t)threadIdx.x + u * 256 < span) xs[threadIdx.x + u * 256] = (Real)xv[u];
        for (int sidx = threadIdx.x + SPT * 256; sidx < span; sidx += 256) { // (spans beyond 1024 samples: very long filters)
            const int64_t l = nfirst + sidx;
            xs[sidx] = (l >= 0 && l < a.in_frames) ? (Real)*sample_at(l) : (Real)0;
        }
    };
    constexpr int EPT = MODE == 0 ? 12 : 8, RPW = HIPSOXR_RPW;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int o0 = wave; o0 < NO; o0 += 4 * RPW) {
        uint64_t au[RPW];
#pragma unroll
        for (int r = 0; r < RPW; ++r) au[r] = aux[o0 + 4 * r < NO ? o0 + 4 * r : o0];
        for (int j0 = lane; j0 < T; j0 += 64 * EPT) {
            Real cv[RPW][EPT];
            V4 pv[RPW][MODE == 0 ? 1 : EPT];
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const Real *crow = MODE == 0 ? (const Real *)ca.phase_major + au[r] * (uint64_t)T : nullptr;
                const V4 *prow = MODE == 0 ? nullptr : (const V4 *)ia.tab + (size_t)(au[r] >> 32) * T;
#pragma unroll
                for (int u = 0; u < EPT; ++u) { // (unconditional, clamped: see load_span)
                    const int j = j0 + u * 64, jc = j < T ? j : T - 1;
                    if (MODE == 0) cv[r][u] = crow[jc];
                    else pv[r][u] = prow[jc];
                }
            }
            load_span();
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
#pragma unroll
                for (int u = 0; u < EPT; ++u) {
                    const int j = j0 + u * 64;
                    if (j < T && o0 + 4 * r < NO) {
                        if (MODE != 0) {
                            const Real xx = (Real)(uint32_t)au[r] * (Real)(1. / (double)(1ULL << SH));
                            cv[r][u] = fma_r(fma_r(fma_r(pv[r][u].w, xx, pv[r][u].z), xx, pv[r][u].y), xx, pv[r][u].x);
                        }
                        cs[(size_t)(o0 + 4 * r) * RS + j] = cv[r][u];
                    }
                }
            }
        }
    }
    load_span(); // (waves without a row)
    store_span();
    __syncthreads();
    HIPSOXR_CB_STAMP(1);
    // ---- the half-chains: wave 0 the left halves (ascending), wave 1 the right halves (descending) — one
    //      instruction stream per wave; taps in blocks of UNR*V with every LDS read of a block issued before
    //      its FMAs (the chain is a dependent sequence: what can be hidden is the read latency)
    Real *red = reinterpret_cast<Real *>(n0s); // (positions are consumed: the right halves' sums go here)
    const int64_t my_n0 = n0s[lane < NO ? lane : 0];
    __syncthreads();
    Real acc = 0;
    if (wave < 2 && lane < NO) {
        const int o = lane;
        const Real *row = cs + (size_t)o * RS;
        const Real *xw = xs + (my_n0 - nfirst); // this output's window inside the shared span
        // (blocks of UNR*V taps, all LDS reads of a block in front of its FMAs.  Reading block k+1 during the FMAs
        //  of block k — ping-pong registers — came out slower, 3.5-4.8 vs 2.6 us for 368 taps: a wave can wait on
        //  at most 15 outstanding LDS reads, and the compiler's schedule of the two-block body was worse)
        constexpr int UNR = 8;
        typedef typename VecN<Real, V>::type CV;
        if (wave == 0) {
            int i = 0;
            for (; i + UNR * V <= H; i += UNR * V) {
                Real c[UNR * V], x[UNR * V];
#pragma unroll
                for (int u = 0; u < UNR; ++u) *reinterpret_cast<CV *>(c + u * V) = *reinterpret_cast<const CV *>(row + i + u * V);
#pragma unroll
                for (int v = 0; v < UNR * V; ++v) x[v] = xw[i + v];
#pragma unroll
                for (int v = 0; v < UNR * V; ++v) acc = fma_r(c[v], x[v], acc);
            }
            for (; i < H; i += V) { // taps i .. i+V-1, ascending
                Real c[V];
                *reinterpret_cast<CV *>(c) = *reinterpret_cast<const CV *>(row + i);
#pragma unroll
                for (int v = 0; v < V; ++v) acc = fma_r(c[v], xw[i + v], acc);
            }
        } else {
            int i = T - UNR * V;
            for (; i >= H; i -= UNR * V) { // taps i+UNR*V-1 .. i, descending
                Real c[UNR * V], x[UNR * V];
#pragma unroll
                for (int u = 0; u < UNR; ++u) *reinterpret_cast<CV *>(c + u * V) = *reinterpret_cast<const CV *>(row + i + u * V);
#pragma unroll
                for (int v = 0; v < UNR * V; ++v) x[v] = xw[i + v];
#pragma unroll
                for (int v = UNR * V - 1; v >= 0; --v) acc = fma_r(c[v], x[v], acc);
            }
            for (i += (UNR - 1) * V; i >= H; i -= V) { // taps i+V-1 .. i, descending
                Real c[V];
                *reinterpret_cast<CV *>(c) = *reinterpret_cast<const CV *>(row + i);
#pragma unroll
                for (int v = V - 1; v >= 0; --v) acc = fma_r(c[v], xw[i + v], acc);
            }
            red[o] = acc;
        }
    }
    pre();
    __syncthreads();
    HIPSOXR_CB_STAMP(2);
    if (veto && (*veto >> 63)) return;
    if (wave == 0 && lane < NO) {
        const int o = lane;
        const Real accR = red[o];
        const int64_t idx = o_base + o;
        if (idx < a.out_frames) {
            IO *yo = (IO *)a.out + (int64_t)clip * a.ocs + idx * a.ofs + (int64_t)ch * a.ochs;
            store_out<Real>(yo, acc + accR, a.oc, ch, a.out_k0 + idx);
        }
    }
    HIPSOXR_CB_STAMP(3);
#undef HIPSOXR_CB_STAMP
}

template <typename IO, typename Real, int MODE>
__global__ void __launch_bounds__(256) k_chain(ChainArgs ca)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const GatherArgs &g = ca.ia.g;
    const ChainMsg m = {g.in_abs0, g.in_frames, g.out_k0, g.out_frames, g.d0, g.p0, ca.ia.t_hi, ca.ia.t_lo, ca.ia.s_hi, ca.ia.s_lo, ca.ia.d_hi, ca.ia.d_lo};
    chain_body<IO, Real, MODE>(ca, m, blockIdx.x, blockIdx.y, smem_raw);
    if (ca.done_words) {
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0)
            __hip_atomic_store(&ca.done_words[blockIdx.y * gridDim.x + blockIdx.x], ca.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---------------------------------------------------------------------------------------------
// k_chain_multi — k_chain over MANY INDEPENDENT STREAMS in one launch (round 5): grid.y = stream x channel, every stream
// with its own ring, output buffer, counters and phase (ChainItem).  A stream's new chunk is read where the caller left it
// and copied into the stream's ring by the same workgroups (share by share: nobody in this launch reads the ring region
// they write), so a device-chunk stream call is ONE dispatch; N callers' chunks are one dispatch too.
// ---------------------------------------------------------------------------------------------
struct ChainMultiArgs {
    ChainArgs ca;            // what the streams share: plan tables, geometry of a column, NO, LDS layout
    const ChainItem *items;  // device-readable table, or nullptr: the one item below
    ChainItem one;
    uint32_t n_channels;
};
template <typename IO, typename Real, int MODE>
__global__ void __launch_bounds__(256) k_chain_multi(ChainMultiArgs m)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const uint32_t nch = m.n_channels;
    const uint32_t item_i = __builtin_amdgcn_readfirstlane(blockIdx.y / nch), ch = __builtin_amdgcn_readfirstlane(blockIdx.y % nch);
    const ChainItem it = m.items ? m.items[item_i] : m.one;
    const uint32_t NO = (uint32_t)m.ca.NO;
    const uint32_t nx = (uint32_t)((it.out_frames + NO - 1) / NO), nxc = nx ? nx : 1; // (a stream without outputs still appends its chunk)
    if (blockIdx.x >= nxc) return;
    if (it.chunk && it.chunk_frames > 0) { // this workgroup's share of [the frames the ring keeps, when it moves] + the chunk -> ring_dst
        const bool moving = it.ring_dst != it.ring;
        const size_t keep_n = moving ? (size_t)(it.split - it.keep_from) * nch : 0;
        const size_t total = keep_n + (size_t)it.chunk_frames * nch, W = (size_t)nxc * nch, per = (total + W - 1) / W;
        const size_t lo = ((size_t)ch * nxc + blockIdx.x) * per, hi = lo + per < total ? lo + per : total;
        IO *dst = (IO *)it.ring_dst;
        const IO *old = (const IO *)it.ring + (size_t)it.keep_from * nch, *src = (const IO *)it.chunk;
        const size_t chunk_at = moving ? keep_n : (size_t)it.split * nch;
        for (size_t e = lo + threadIdx.x; e < hi; e += 256) {
            if (e < keep_n) dst[e] = old[e];
            else dst[chunk_at + (e - keep_n)] = src[e - keep_n];
        }
    }
    if (blockIdx.x >= nx) return;
    ChainArgs ca = m.ca;
    GatherArgs &g = ca.ia.g;
    g.in = it.ring; g.out = it.out; g.n_clips = 1;
    g.oc.clip_counter = (uint64_t *)it.clip_counter; g.oc.seed = it.dither_seed;
    const ChainMsg msg = {it.in_abs0, it.in_frames, it.out_k0, it.out_frames, it.d0, it.p0, 0, 0, 0, 0, 0, 0};
    chain_body<IO, Real, MODE>(ca, msg, blockIdx.x, ch, smem_raw, nullptr, NoHook(), nullptr, it.chunk, it.split);
}

// ---------------------------------------------------------------------------------------------
// k_chain_resident — k_chain as a RESIDENT consumer: launched once, fed by messages
// ---------------------------------------------------------------------------------------------
// A synchronous streaming call on a small chunk costs ~31 us, of which the arithmetic is ~2: the rest is one
// kernel launch (API ~7 us, dispatch ~4 us), the completion event and its polling.  Here the kernel stays on
// the GPU between calls and the host talks to it through two cache lines of pinned, device-mapped host
// memory (ResidentBox) — no HIP call per chunk at all (tools/ubench/mailbox.hip: 3.9 us for the bare round
// trip host -> kernel -> host, 5.6 us with 1 KiB read from pinned memory on the way):
//   host -> device  w[0..4]: the ChainMsg of the call, each 8-byte word carrying the message number in its top
//                   16 bits (an 8-byte read is atomic whatever the load is split into: a word is either this
//                   message's or stale, and the message is taken once all five carry the expected number);
//                   w[5]: "instance e, leave" (between messages only);
//   device -> host  done = number of the last message whose output is complete in pinned memory; exited = e.
// Every workgroup polls the box itself (one wave, s_sleep between reads) and owns the same NO outputs of
// every message as in k_chain (workgroups past the end of a short message just report in).  The input ring,
// the result buffer and the plan are launch arguments: when one of them moves, the host retires the instance
// and launches another.
// Leaving.  The kernel must not outlive its host's interest (a device-wide synchronisation elsewhere in the
// process waits for it), so an instance that hears nothing for idle_ticks leaves by itself — and all its
// workgroups must take the SAME decision about every message, or a message would be half computed (and its
// clipped samples counted twice when the next instance repeats it).  One word of device memory per instance
// (ctl->dec = number of messages accepted, top bit = sealed) arbitrates: a workgroup that sees message n+1
// does CAS(n -> n+1), one that has waited too long does CAS(n -> n|SEAL); whichever CAS lands first decides
// for everybody (a workgroup whose seal fails because n+1 was accepted goes back for the message, one whose
// accept fails because the instance was sealed AT n leaves; an accept that finds (n+1)|SEAL was merely late — the
// message had been accepted before the seal — and is answered like any other).  The host, waiting for `done`, sees `exited` instead
// and launches the next instance, which finds the message still in the box.
// ---------------------------------------------------------------------------------------------
// (ResidentBox, ResidentCtl: device.h)
struct ResidentArgs {
    ChainArgs ca;
    ResidentBox *box;
    const uint64_t *words; // host -> device words (box->w, or device memory the CPU stores into)
    ResidentCtl *ctl;
    uint32_t base_seq; // messages taken by earlier instances
    uint32_t epoch;    // this instance
    int64_t idle_ticks; // of wall_clock64 (100 MHz)
    uint32_t n_wgs;
};
static constexpr unsigned long long kResidentSeal = 1ULL << 63;
static constexpr uint64_t kResidentMask48 = (1ULL << 48) - 1;

template <typename IO, typename Real, int MODE>
__global__ void __launch_bounds__(256) k_chain_resident(ResidentArgs ra)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ uint64_t s_w[16];
    constexpr int NW = MODE == 2 ? 14 : 5; // message words (the variable-rate clock rides in words 5..13); word 15 = leave
    __shared__ unsigned long long s_old;
    __shared__ int s_state;
    const uint32_t wg = blockIdx.y * gridDim.x + blockIdx.x;
    unsigned long long n = 0; // messages this instance has completed
    long long t_idle = wall_clock64();
    for (;;) {
        if (threadIdx.x < 64) { // one wave polls
            const int lane = threadIdx.x;
            const uint64_t want = (uint64_t)((ra.base_seq + (uint32_t)n + 1u) & 0xffffu);
            int state;
            uint64_t v = 0;
            for (;;) {
                if (lane < 16) v = __hip_atomic_load(&ra.words[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                const bool ok = lane >= NW || (v >> 48) == want;
                const uint64_t leave = __shfl(v, 15, 64);
                if (__all(ok)) { state = 1; break; }
                if (leave == (uint64_t)ra.epoch) { state = 2; break; }
                if (wall_clock64() - t_idle > ra.idle_ticks) { state = 3; break; }
                __builtin_amdgcn_s_sleep(8);
            }
            if (lane < NW) s_w[lane] = v & kResidentMask48;
            if (lane == 0) {
                // too long without a message: try to seal the instance (the arbiter, see above)
                if (state == 3) s_old = atomicCAS(&ra.ctl->dec, n, n | kResidentSeal);
                s_state = state;
            }
        }
        __syncthreads();
        const int state = s_state;
#ifdef HIPSOXR_RES_TRACE
        const long long tr0 = wall_clock64();
#endif
        if (state == 2) break;                                   // told to leave
        if (state == 3) {
            const unsigned long long old = s_old;
            if (old == n || (old & kResidentSeal)) break;        // sealed: everybody leaves after message n
            __syncthreads();                                     // message n+1 was accepted by somebody: it is in the box
            continue;
        }
        // message n+1 is here: accept it.  The arbiter's round trip (~1 us) runs behind the body's own loads:
        // one lane of the last wave asks now and publishes the answer in front of the body's last barrier.
        unsigned long long old = 0;
        const bool asker = threadIdx.x == 192;
        if (asker) old = atomicCAS(&ra.ctl->dec, n, n + 1);
        // (old == (n+1)|SEAL: the others accepted message n+1, finished it and sealed after idling before this
        //  workgroup's CAS arrived — "accepted, then sealed", not a veto: the message is partly answered already and
        //  this workgroup owes its share; it stores, and leaves on its next poll.  Only a seal AT n withholds.)
        auto publish = [&]() { if (asker) s_old = old == ((n + 1) | kResidentSeal) ? n + 1 : old; };
        ChainMsg m;
        {
            auto uni = [](uint64_t x) {
                return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(x >> 32)) << 32) |
                       (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)x);
            };
            const uint64_t w3 = uni(s_w[3]), w4 = uni(s_w[4]);
            m.in_abs0 = (int64_t)uni(s_w[0]); m.out_k0 = (int64_t)uni(s_w[1]); m.d0 = (int64_t)uni(s_w[2]);
            m.p0 = (int64_t)(w3 & 0xffffffu); m.in_frames = (int64_t)(w3 >> 24);
            m.out_frames = (int64_t)w4;
            m.t_hi = m.t_lo = m.s_hi = m.s_lo = m.d_hi = m.d_lo = 0;
            if (MODE == 2) { // three 128-bit numbers, each as 48 + 48 + 32 bits (low piece first)
                auto u128 = [&](int i, uint64_t &hi, uint64_t &lo) {
                    const uint64_t a0 = uni(s_w[i]), a1 = uni(s_w[i + 1]), a2 = uni(s_w[i + 2]);
                    lo = a0 | (a1 << 48);
                    hi = (a1 >> 16) | (a2 << 32);
                };
                u128(5, m.t_hi, m.t_lo); u128(8, m.s_hi, m.s_lo); u128(11, m.d_hi, m.d_lo);
            }
        }
        if ((int64_t)blockIdx.x * ra.ca.NO < m.out_frames) {
            chain_body<IO, Real, MODE>(ra.ca, m, blockIdx.x, blockIdx.y, smem_raw,
#ifdef HIPSOXR_RES_TRACE
                                       wg == 0 ? ra.box->pad + 5 : nullptr,
#else
                                       nullptr,
#endif
                                       publish, &s_old);
        } else {
            publish();
            __syncthreads();
        }
        if (s_old & kResidentSeal) break;                        // sealed before this workgroup saw the message: nothing was stored
#ifdef HIPSOXR_RES_TRACE
        const long long tr1 = wall_clock64();
#endif
        // this workgroup's results are in host memory before its word says so (the host waits for every word:
        // no arrival counter, no device-wide atomic on the way out)
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) {
#ifdef HIPSOXR_RES_TRACE
            if (wg == 0) { ra.box->pad[0] = (uint32_t)(tr0 - t_idle); ra.box->pad[1] = (uint32_t)(tr1 - tr0); ra.box->pad[2] = (uint32_t)(wall_clock64() - tr1); }
#endif
            __hip_atomic_store(&ra.box->done[wg], ra.base_seq + (uint32_t)n + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        ++n;
        t_idle = wall_clock64();
        __syncthreads(); // (s_w / s_state are rewritten by the next poll)
    }
    if (threadIdx.x == 0 && wg == 0) __hip_atomic_store(&ra.box->exited, ra.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---------------------------------------------------------------------------------------------
// k_tile
// ---------------------------------------------------------------------------------------------
// Geometry (host-built, see build_tile_tables): the plan's period may be replicated c times so
// that Lc = c*L >= RT; "period" below means the replicated period (Lc outputs <- Mc inputs).
//   tile rt covers outputs r = rt*RT .. rt*RT+RT-1 of a period; for row r
//       n_r = floor(r*M/L) - (T/2-1)   (first input, relative to the period's first input)
//       p_r = (r*M) mod L              (phase)
//   left  half-chain: inputs i = eL0 + ii            (ascending),  table L[ii][rr]
//   right half-chain: inputs i = eR0 + 3 - ii        (descending), table R[ii][rr]
//   (e-coordinates are relative to i_min, the first input sample kept in LDS.)
struct TileArgs {
    const void *in;
    void *out;
    const void *tab;     // [n_rt][2][I_h][RT] Real, constant address space
    const int32_t *e0;   // [n_rt][2]  (eL0, eR0)
    int64_t Lc, Mc;      // replicated period
    int32_t n_rt, I_h, n_waves;
    int32_t rowR, plane; // k_tile_mfma_p: plane row stride and plane stride (words)
    unsigned long long *trace; // HIPSOXR_DEBUG_TRACE: per-wave s_memtime stamps [block][wave][16]
    int32_t dbg; // timing ablations only (HIPSOXR_DEBUG_FLAGS): 1 no staging loads, 2 no LDS reads, 4 no coefficient loads, 8 no stores
    int32_t pad, i_min, x_count; // LDS row padding; first staged input; samples staged per tile
    int32_t pb;                  // k_tile: periods per slab (64, or fewer with the upper lanes idle)
    uint32_t n_clips, n_channels;
    int64_t ics, ifs, ichs, ocs, ofs, ochs;
    int64_t in_abs0, in_frames;
    int64_t out_k0, out_frames;
    int64_t b_first;     // absolute (replicated) period index handled by lane 0 of block x = 0
    OutCtx oc;
    // k_tile_mfma_p with a unit split Z > 1: XCD-aware ids.  The Z workgroups of a slab stage the
    // same input; consecutive ids go to different XCDs (private L2s), so they are laid out as
    // id = 8*(chunk*Z + z) + xcd  <->  slab = 8*chunk + xcd: same XCD, adjacent in dispatch order.
    int32_t xz, nx;      // Z (0: plain 3-D grid), number of slabs
    int32_t halves, scratch_off; // k_tile_mfma: a row tile's two half-chains on two waves (sum through LDS at scratch_off, in elements)
};

// Stage the input slab of one workgroup: samples [bw*Mc + i_min, +x_count) of column (clip, ch)
// into LDS as Real, row-padded (address n + pad*(n/Mc)), zero outside the signal.  x_count and
// i_min are multiples of 4 (host geometry).  Each thread first ISSUES up to UNR independent
// 4-sample loads (16-byte global loads when the source is contiguous and aligned), then converts
// and writes them, so that the HBM latency is paid once per batch rather than once per sample.
template <typename IO, typename Real, bool ALIGNED>
__device__ __forceinline__ void stage_slab(const TileArgs &a, Real *xs, uint32_t clip, uint32_t ch,
                                           int64_t bw)
{
    typedef IO IO4 __attribute__((ext_vector_type(4)));
    constexpr int UNR = 4;
    const int32_t Mc = (int32_t)a.Mc, pad = a.pad;
    const IO *xin = (const IO *)a.in + (int64_t)clip * a.ics + (int64_t)ch * a.ichs;
    const int64_t loc_base = bw * a.Mc + a.i_min - a.in_abs0;
    const bool vec = a.ifs == 1 && ((loc_base & 3) == 0) &&
                     ((reinterpret_cast<uintptr_t>(xin) & (4 * sizeof(IO) - 1)) == 0);
    const int32_t n4 = a.x_count >> 2;
    const int32_t stride = (int32_t)blockDim.x;
    for (int32_t q0 = threadIdx.x; q0 < n4; q0 += stride * UNR) {
        IO4 v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int32_t q = q0 + u * stride;
            v[u] = (IO4){0, 0, 0, 0};
            if (q < n4) {
                const int64_t l = loc_base + ((int64_t)q << 2);
                if (vec && l >= 0 && l + 3 < a.in_frames) {
                    v[u] = *reinterpret_cast<const IO4 *>(xin + l);
                } else {
                    if (l >= 0 && l < a.in_frames) v[u].x = xin[l * a.ifs];
                    if (l + 1 >= 0 && l + 1 < a.in_frames) v[u].y = xin[(l + 1) * a.ifs];
                    if (l + 2 >= 0 && l + 2 < a.in_frames) v[u].z = xin[(l + 2) * a.ifs];
                    if (l + 3 >= 0 && l + 3 < a.in_frames) v[u].w = xin[(l + 3) * a.ifs];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int32_t q = q0 + u * stride;
            if (q < n4) {
                const int32_t n = q << 2, row = n / Mc, rem = n - row * Mc;
                Real *dst = xs + n + pad * row;
                if (ALIGNED) { // Mc % 4 == 0 and pad % 4 == 0: the quad never straddles a row
                    typedef Real R4 __attribute__((ext_vector_type(4)));
                    R4 o = {(Real)v[u].x, (Real)v[u].y, (Real)v[u].z, (Real)v[u].w};
                    *reinterpret_cast<R4 *>(__builtin_assume_aligned(dst, 4 * sizeof(Real))) = o;
                } else {
                    dst[0] = (Real)v[u].x;
                    dst[1 + (rem + 1 >= Mc ? pad : 0)] = (Real)v[u].y;
                    dst[2 + (rem + 2 >= Mc ? pad : 0)] = (Real)v[u].z;
                    dst[3 + (rem + 3 >= Mc ? pad : 0)] = (Real)v[u].w;
                }
            }
        }
    }
}

// 4 consecutive staged samples of this lane's row.  The aligned form is one ds_read_b128
// (conflict-free: the row stride is 4*odd words).
template <typename Real> struct Quad { Real v[4]; };
__device__ __forceinline__ Quad<float> lds_quad_aligned(const float *p)
{
    const float4 t = *reinterpret_cast<const float4 *>(__builtin_assume_aligned(p, 16));
    return Quad<float>{{t.x, t.y, t.z, t.w}};
}
__device__ __forceinline__ Quad<double> lds_quad_aligned(const double *p)
{
    const double2 a = *reinterpret_cast<const double2 *>(__builtin_assume_aligned(p, 16));
    const double2 b = *reinterpret_cast<const double2 *>(__builtin_assume_aligned(p + 2, 16));
    return Quad<double>{{a.x, a.y, b.x, b.y}};
}

template <typename IO, typename Real, int RT, bool ALIGNED>
__global__ void __launch_bounds__(1024) k_tile(TileArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Real *xs = reinterpret_cast<Real *>(smem_raw);

    const uint32_t col = blockIdx.y;
    // run-time division goes through the vector ALU; readfirstlane keeps the results (and every
    // address derived from them) on the scalar side
    const uint32_t ch = __builtin_amdgcn_readfirstlane(col % a.n_channels), clip = __builtin_amdgcn_readfirstlane(col / a.n_channels);
    const int32_t pb = a.pb; // periods per slab: 64, or fewer (the lanes above compute a copy of the last row and store nothing)
    const int64_t bw = a.b_first + (int64_t)blockIdx.x * pb; // first period of this workgroup
    const int32_t Mc = (int32_t)a.Mc, pad = a.pad;

    stage_slab<IO, Real, ALIGNED>(a, xs, clip, ch, bw);
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const bool live = lane < pb;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_waves = a.n_waves; // waves that compute (all of them, or the first few of a split slab's workgroup: launch_tile)
    const Real *xl = xs + (live ? lane : pb - 1) * (Mc + pad);
    const int64_t b = bw + lane; // this lane's period
    typedef const __attribute__((address_space(4))) Real *CPtr;

    // whole workgroup inside the requested output range? (uniform) -> stores need no per-sample test
    const bool interior = pb == 64 && bw * a.Lc >= a.out_k0 && (bw + 64) * a.Lc <= a.out_k0 + a.out_frames;
    IO *const yo = (IO *)a.out + (int64_t)clip * a.ocs + (int64_t)ch * a.ochs +
                   (b * a.Lc - a.out_k0) * a.ofs; // this lane's period start (may be out of range)

    // (few slabs: the row tiles of a slab are spread over gridDim.z workgroups, each staging the slab — launch_tile)
    for (int rt_ = wave < n_waves ? wave + n_waves * (int)blockIdx.z : a.n_rt; rt_ < a.n_rt; rt_ += n_waves * (int)gridDim.z) {
        // keep the tile index (and everything derived from it) provably wave-uniform: the
        // coefficient loads below must be scalar (s_load), not per-lane
        const int rt = __builtin_amdgcn_readfirstlane(rt_);
        const int32_t eL0 = __builtin_amdgcn_readfirstlane(a.e0[rt * 2 + 0]);
        const int32_t eR0 = __builtin_amdgcn_readfirstlane(a.e0[rt * 2 + 1]);
        CPtr tL = (CPtr)((const Real *)a.tab + (size_t)(rt * 2 + 0) * a.I_h * RT);
        CPtr tR = (CPtr)((const Real *)a.tab + (size_t)(rt * 2 + 1) * a.I_h * RT);
        Real accL[RT], accR[RT];
#pragma unroll
        for (int rr = 0; rr < RT; ++rr) { accL[rr] = 0; accR[rr] = 0; }

        // left half: ascending inputs
        {
            int32_t e = eL0, padoff = pad * (e / Mc), next = (e / Mc + 1) * Mc;
            for (int32_t q = 0; q < a.I_h; q += 4) {
                Quad<Real> x;
                if (ALIGNED) {
                    x = lds_quad_aligned(xl + e + padoff);
                } else {
                    // a chunk may straddle row-padding points: resolve each sample separately
                    const int32_t e1 = e + 1, e2 = e + 2, e3 = e + 3;
                    if (pad) {
                        x.v[0] = xl[e + pad * (e / Mc)];
                        x.v[1] = xl[e1 + pad * (e1 / Mc)];
                        x.v[2] = xl[e2 + pad * (e2 / Mc)];
                        x.v[3] = xl[e3 + pad * (e3 / Mc)];
                    } else {
                        x.v[0] = xl[e]; x.v[1] = xl[e1]; x.v[2] = xl[e2]; x.v[3] = xl[e3];
                    }
                }
                CPtr t = tL + (size_t)q * RT;
#pragma unroll
                for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                    for (int rr = 0; rr < RT; ++rr) accL[rr] = fma_r(t[ii * RT + rr], x.v[ii], accL[rr]);
                e += 4;
                if (e >= next) { padoff += pad; next += Mc; }
            }
        }
        // right half: descending inputs (chunk = 4 ascending addresses consumed high to low)
        {
            int32_t e = eR0, padoff = pad * (e / Mc), lo = (e / Mc) * Mc;
            for (int32_t q = 0; q < a.I_h; q += 4) {
                Quad<Real> x;
                if (ALIGNED) {
                    x = lds_quad_aligned(xl + e + padoff);
                } else {
                    const int32_t e1 = e + 1, e2 = e + 2, e3 = e + 3;
                    if (pad) {
                        x.v[0] = xl[e + pad * (e / Mc)];
                        x.v[1] = xl[e1 + pad * (e1 / Mc)];
                        x.v[2] = xl[e2 + pad * (e2 / Mc)];
                        x.v[3] = xl[e3 + pad * (e3 / Mc)];
                    } else {
                        x.v[0] = xl[e]; x.v[1] = xl[e1]; x.v[2] = xl[e2]; x.v[3] = xl[e3];
                    }
                }
                CPtr t = tR + (size_t)q * RT;
#pragma unroll
                for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                    for (int rr = 0; rr < RT; ++rr) accR[rr] = fma_r(t[ii * RT + rr], x.v[3 - ii], accR[rr]);
                e -= 4;
                if (e < lo) { padoff -= pad; lo -= Mc; }
            }
        }
        // store: output k = b*Lc + rt*RT + rr
        const int32_t r0 = rt * RT;
        IO *const yt = yo + (int64_t)r0 * a.ofs;
        if (interior && r0 + RT <= a.Lc) {
#pragma unroll
            for (int rr = 0; rr < RT; ++rr)
                store_out<Real>(yt + rr * a.ofs, accL[rr] + accR[rr], a.oc, ch, b * a.Lc + r0 + rr);
        } else {
#pragma unroll
            for (int rr = 0; rr < RT; ++rr) {
                const int64_t k = b * a.Lc + r0 + rr, idx = k - a.out_k0;
                if (live && r0 + rr < a.Lc && idx >= 0 && idx < a.out_frames)
                    store_out<Real>(yt + rr * a.ofs, accL[rr] + accR[rr], a.oc, ch, k);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_tile_mfma — f32 engine.  Same tiling as k_tile (64 periods x 16 output phases per wavefront),
// executed on the f32-input matrix pipe: one v_mfma_f32_16x16x4_f32 adds, for 16 phases x 16
// periods, the contributions of 4 consecutive input samples,
//     D[r][j] += sum_{k=0..3} C'[r][e+k] * x[period j][e+k],
// evaluated by the hardware as the k-ordered chain fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0, C))))
// with one rounding per product (MI355X guide §3 "FP32-input MFMA": bit-for-bit an fmaf chain) —
// i.e. exactly the canonical order.  The right half-chain maps k to DESCENDING input index.
// It is used because this FIR is FMA-bound (592 flop per 8.35 algorithmic bytes, 3.6x the ridge):
// both operands are per-lane VGPRs (coefficients: one coalesced 256-byte global load per chunk;
// samples: four conflict-free ds_read_b32), so nothing has to squeeze through the SGPR file, and
// the f32 MFMA rate equals the f32 VALU rate (64 FLOP/clk/SIMD) while leaving the VALU free for
// addressing.  It is NOT a reshaping into a dense GEMM for low-precision throughput: same flops,
// same f32 arithmetic, same results.
// Operand layouts (16x16x4): A lane l = C'[row l&15][k = l>>4]; B lane l = x[period l&15][k = l>>4];
// D lane l, reg v = D[row 4*(l>>4)+v][period l&15].
// ---------------------------------------------------------------------------------------------
// Real = double (round 3): the float64 engine (float64 / int32 I/O) on v_mfma_f64_16x16x4_f64.  The hardware evaluates it
// as the same k-ordered fma chain, one rounding per product — bitwise equal to std::fma chains on 51 200 random elements
// of 8 chained instructions (tools/ubench/mfma_f64_order.hip; the descending chain, pairwise sums and fma trees all
// differ) — so the canonical order holds and the oracle's port_f64 is reproduced bit for bit.  Two differences from
// the f32 form: the accumulator layout (lane l, register v = row (l >> 4) + 4 v, MI355X guide §3, where the f32 form
// has row 4 (l >> 4) + v) and the slab (8 bytes per sample: NG = 4, 2 or 1 groups of 16 periods, whatever fits LDS).
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <typename Real> struct MfmaOf;
template <> struct MfmaOf<float> {
    typedef f32x4 Acc;
    static __device__ __forceinline__ Acc mac(float a, float b, Acc c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row(int kq, int v) { return 4 * kq + v; }
};
template <> struct MfmaOf<double> {
    typedef f64x4 Acc;
    static __device__ __forceinline__ Acc mac(double a, double b, Acc c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row(int kq, int v) { return kq + 4 * v; }
};

template <typename IO, typename Real = float, int NG = 4>
__global__ void __launch_bounds__(1024) k_tile_mfma(TileArgs a)
{
    typedef typename MfmaOf<Real>::Acc Acc;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Real *xs = reinterpret_cast<Real *>(smem_raw);

    const uint32_t col = blockIdx.y;
    // run-time division goes through the vector ALU; readfirstlane keeps the results (and every
    // address derived from them) on the scalar side
    const uint32_t ch = __builtin_amdgcn_readfirstlane(col % a.n_channels), clip = __builtin_amdgcn_readfirstlane(col / a.n_channels);
    const int64_t bw = a.b_first + (int64_t)blockIdx.x * (16 * NG);
    const int32_t Mc = (int32_t)a.Mc, pad = a.pad, S = Mc + pad;

    if (!(a.dbg & 1)) stage_slab<IO, Real, false>(a, xs, clip, ch, bw);
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int kq = lane >> 4, j = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_waves = a.n_waves;
    const int32_t n_chunks = a.I_h >> 2;
    const Real *xrow = xs + j * S; // period j of group 0; group g adds 16*g*S

    const bool interior = bw * a.Lc >= a.out_k0 && (bw + 16 * NG) * a.Lc <= a.out_k0 + a.out_frames;
    IO *const ybase = (IO *)a.out + (int64_t)clip * a.ocs + (int64_t)ch * a.ochs;

    // Small jobs (a.halves, round 3): a row tile's left and right half-chains run on TWO waves — a chain of ~440 k-steps
    // is bound by its per-step address arithmetic, whatever the number of MFMAs it feeds, and the two halves are
    // independent until their sum — and meet through LDS: wave 2p writes its accumulators, the workgroup synchronises,
    // wave 2p + 1 adds its own (left + right, as ever) and stores.  Every wave then runs the same number of rounds.
    const bool halves = a.halves != 0;
    const int units = halves ? n_waves >> 1 : n_waves;           // row tiles per round of this workgroup
    const int pw = halves ? wave >> 1 : wave, side = halves ? wave & 1 : 2; // side 0: left half, 1: right half, 2: both
    const int stride_rt = units * (int)gridDim.z;
    const int rounds = halves ? (a.n_rt + stride_rt - 1) / stride_rt : 0;
    Real *const scratch = xs + a.scratch_off;
    int round = 0;
    for (int rt_ = wave < n_waves ? pw + units * (int)blockIdx.z : a.n_rt; halves ? round < rounds : rt_ < a.n_rt; rt_ += stride_rt, ++round) { // (gridDim.z: see k_tile)
        const bool active = rt_ < a.n_rt;
        const int rt = __builtin_amdgcn_readfirstlane(active ? rt_ : 0);
        const int32_t eL0 = __builtin_amdgcn_readfirstlane(a.e0[rt * 2 + 0]);
        const int32_t eR0 = __builtin_amdgcn_readfirstlane(a.e0[rt * 2 + 1]);
        const size_t half_stride = (size_t)(a.I_h + 16) * 16; // + 4 chunks of prefetch slack
        const Real *tL = (const Real *)a.tab + (size_t)(rt * 2 + 0) * half_stride; // (wave-uniform: lanes add their column in the load)
        const Real *tR = (const Real *)a.tab + (size_t)(rt * 2 + 1) * half_stride;
        Acc accL[NG], accR[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) { accL[g] = (Acc){0, 0, 0, 0}; accR[g] = (Acc){0, 0, 0, 0}; }

        // The coefficient operand of the next group of G chunks is fetched into registers while
        // the current group's 4*G MFMAs run (one VGPR per chunk).  The prefetch pointer is made
        // opaque so that the compiler cannot fold the software pipeline back into load-then-use.
        constexpr int G = 4; // n_chunks is a multiple of G (host geometry); tables carry G chunks of slack
        // One half-chain, software-pipelined one GROUP (four chunks) ahead for both operands (round 3): the B values of
        // group q + 1 (G x NG ds_read_b32) and the A values of group q + 1 (G loads) are issued before group q's MFMAs.
        // Before, every MFMA waited for its own LDS read (ds_read; s_waitcnt lgkmcnt(0); v_mfma — four LDS round trips
        // per group), and the timing-ablation switches sat inside the loop as branches.
        auto chains = [&](auto pad0_tag) {
        constexpr bool PAD0 = decltype(pad0_tag)::value; // unpadded slab: offset == input index
        auto half = [&](auto right_tag, Acc (&acc)[NG], const Real *tab_half, int32_t e_first) {
            constexpr bool RIGHT = decltype(right_tag)::value;
            // (coefficients through a buffer descriptor — scalar offsets, no vector address arithmetic: see mfma_half_chain)
            const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc((void *)tab_half, 0, 0x40000000, 0x00020000);
            const int lane_bytes = lane * (int)sizeof(Real);
            auto tab_at = [&](int32_t idx) -> Real {
                if constexpr (sizeof(Real) == 4) return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(trs, lane_bytes, idx * 4, 0));
                else return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(trs, lane_bytes, idx * 8, 0));
            };
            // left: lane k handles input e = eL0 + 4q + k (ascending); right: chunk q covers inputs [eR0 - 4q, eR0 - 4q + 3]
            // and lane k takes the (3-k)-th of them, so that k = 0 is the highest index (descending order)
            int32_t e = e_first;
            int32_t off = PAD0 ? e : e + pad * (e / Mc);
            int32_t edge = PAD0 ? 0 : RIGHT ? (e / Mc) * Mc : (e / Mc + 1) * Mc; // next period boundary in e's direction
            auto load_b1 = [&](Real (&b)[NG]) { // one chunk's B values, then on to the next chunk
                const Real *px = xrow + off;
#pragma unroll
                for (int g = 0; g < NG; ++g) b[g] = px[16 * g * S];
                if (!RIGHT) { off += 4; if (!PAD0) { e += 4; if (e >= edge) { off += pad; edge += Mc; } } }
                else { off -= 4; if (!PAD0) { e -= 4; if (e < edge) { off -= pad; edge -= Mc; } } }
            };
            auto load_b = [&](auto &b) {
#pragma unroll
                for (int u = 0; u < G; ++u) load_b1(b[u]);
            };
            int32_t poff = 0; // element offset of the group being prefetched (wave-uniform)
            // (float64 with four period groups: 64 registers of accumulators leave no room for groups of B values in the
            //  128 a 16-wave workgroup may use — there a chunk's B values are loaded in front of its own MFMAs, as before)
            constexpr bool AHEAD = sizeof(Real) * NG <= 16;
            Real ac[G], an[G], bc[AHEAD ? G : 1][NG], bn[AHEAD ? G : 1][NG];
#pragma unroll
            for (int u = 0; u < G; ++u) ac[u] = tab_at(u * 64);
            if constexpr (AHEAD) load_b(bc);
            for (int32_t q = 0; q < n_chunks; q += G) {
                poff += G * 64;
                asm volatile("" : "+s"(poff)); // opaque: keeps the software pipeline from being re-rolled
#pragma unroll
                for (int u = 0; u < G; ++u) an[u] = tab_at(poff + u * 64); // (tables carry G chunks of slack)
                if constexpr (AHEAD) { if (q + G < n_chunks) load_b(bn); } // (the slab carries none: no B read past the chain's last group)
                __builtin_amdgcn_sched_barrier(0); // the prefetches are issued BEFORE this group's MFMAs
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    if constexpr (!AHEAD) load_b1(bc[0]);
#pragma unroll
                    for (int g = 0; g < NG; ++g) acc[g] = MfmaOf<Real>::mac(ac[u], bc[AHEAD ? u : 0][g], acc[g]);
                }
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    ac[u] = an[u];
                    if constexpr (AHEAD) {
#pragma unroll
                        for (int g = 0; g < NG; ++g) bc[u][g] = bn[u][g];
                    }
                }
            }
        };
        if (active && side != 1) half(std::false_type{}, accL, tL, eL0 + kq);
        if (active && side != 0) half(std::true_type{}, accR, tR, eR0 + 3 - kq);
        };
        if (pad == 0) chains(std::true_type{}); else chains(std::false_type{});
        if (halves) { // the left half's accumulators to the wave that holds the right half
            if (round) __syncthreads(); // (the scratch of the round before has been read)
            if (active && side == 0) {
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int v = 0; v < 4; ++v) scratch[((pw * NG + g) * 4 + v) * 64 + lane] = accL[g][v];
            }
            __syncthreads();
            if (!active || side == 0) continue;
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int v = 0; v < 4; ++v) accL[g][v] = scratch[((pw * NG + g) * 4 + v) * 64 + lane];
        }
        // lane holds rows rt*16 + row(kq, v) (v = 0..3; f32: 4 kq + v, f64: kq + 4 v) of periods bw + 16g + j
        const int32_t rbase = rt * 16;
        if ((a.dbg & 8) && accL[0][0] != (Real)12345) continue;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int64_t b = bw + 16 * g + j;
            const int64_t kb = b * a.Lc + rbase;
            IO *const yt = ybase + (kb - a.out_k0) * a.ofs;
            if (interior && rbase + 16 <= a.Lc) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int r = MfmaOf<Real>::row(kq, v);
                    store_out<Real>(yt + r * a.ofs, accL[g][v] + accR[g][v], a.oc, ch, kb + r);
                }
            } else {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int r = MfmaOf<Real>::row(kq, v);
                    const int64_t idx = kb + r - a.out_k0;
                    if (rbase + r < a.Lc && idx >= 0 && idx < a.out_frames)
                        store_out<Real>(yt + r * a.ofs, accL[g][v] + accR[g][v], a.oc, ch, kb + r);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_tile_mfma_p — the fast form of k_tile_mfma for input periods that are a multiple of 16
// samples (48k->44.1k: Mc = 160).  Measured on MI355X (tools/ubench/mfma_rate.hip): the f32 MFMA
// pipe sustains 145-154 TFLOP/s on its own but loses ~4 cycles per VALU instruction issued
// beside it, so the inner loop must contain (almost) nothing but MFMAs.  Therefore:
//   * the slab is stored K-DE-INTERLEAVED in four LDS planes (plane k holds the samples whose
//     offset is == k mod 4), so ONE ds_read_b128 hands lane (j, k) its B operands for FOUR
//     consecutive chunks; plane row stride R = Mc/4 + padR with R/4 odd and plane stride a
//     multiple of 64 words makes every 16-lane read group conflict-free;
//   * the A operands of four chunks arrive with ONE coalesced global_load_dwordx4 per lane,
//     prefetched one group (16 MFMAs) ahead;
//   * all offsets inside the loop are wave-uniform scalars: one v_add per 16 MFMAs.
// Groups of 16 inputs are aligned to 16 (never straddle a slab row).  Same canonical arithmetic.
// ---------------------------------------------------------------------------------------------
// One half-chain of a work unit (16 phases x 32 periods), software-pipelined inside the wave:
// the B operands (two ds_read_b128) of group g+1 and the A operand (one global_load_dwordx4) of
// group g+2 are in flight while the 8 MFMAs of group g issue, so that a single wave per SIMD keeps
// the matrix pipe busy.  The loop is unrolled by two groups with ping-pong registers (no copies).
// RIGHT = false: ascending groups, chunk c uses component c; true: descending, component 3-c.
// (round 3: the table is read through a buffer descriptor — `buffer_load_dwordx4 v, v_lane16, s[rsrc], s_offset offen`:
//  wave-uniform base in the descriptor, the lane's 16-byte column as the one vector offset, the group as a SCALAR offset —
//  so that an A load costs no vector-ALU instruction; as a per-lane pointer plus scalar offset every load came with a
//  64-bit v_lshl_add, and beside a busy matrix pipe each vector-ALU instruction costs ~4 pipe cycles.  Plain pointer
//  arithmetic does not get there: base + lane offset is hoisted out of the loop as one 64-bit per-lane pointer.)
template <bool RIGHT>
__device__ __forceinline__ void mfma_half_chain(f32x4 (&acc)[2], const char *tb, uint32_t lane16, const float *xb,
                                                int32_t e0, int32_t n_groups, int32_t Mc, int32_t R,
                                                int32_t padR)
{
    const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc((void *)tb, 0, 0x40000000, 0x00020000);
    auto t_at = [&](int32_t idx) { // element idx of the lane's column
        // (bit_cast of the builtin's own result: assigning it to an ext_vector_type of unsigned first silently yields
        //  four copies of its first element with this compiler)
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(trs, (int)lane16, idx * 16, 0));
    };
    int32_t rem = e0 % Mc, fo = (e0 / Mc) * R + (rem >> 2); // wave-uniform plane offset of the next B read
    auto ldb = [&](float4 &b0, float4 &b1) {
        const float *px = xb + fo;
        b0 = *reinterpret_cast<const float4 *>(__builtin_assume_aligned(px, 16));
        b1 = *reinterpret_cast<const float4 *>(__builtin_assume_aligned(px + 16 * R, 16));
        if (!RIGHT) { fo += 4; rem += 16; if (rem == Mc) { rem = 0; fo += padR; } }
        else { fo -= 4; rem -= 16; if (rem < 0) { rem += Mc; fo -= padR; } }
    };
#define HIPSOXR_MFMA8(AV, B0, B1)                                                              \
    if (!RIGHT) {                                                                               \
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV.x, B0.x, acc[0], 0, 0, 0);             \
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV.x, B1.x, acc[1], 0, 0, 0);             \
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV.y, B0.y, acc[0], 0, 0, 0);             \
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV.y, B1.y, acc[1], 0, 0, 0);             \
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV.z, B0.z, acc[0], 0, 0, 0);             \
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV.z, B1.z, acc[1], 0, 0, 0);             \
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV.w, B0.w, acc[0], 0, 0, 0);             \
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV.w, B1.w, acc[1], 0, 0, 0);             \
    } else {                                                                                    \
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV.x, B0.w, acc[0], 0, 0, 0);             \
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV.x, B1.w, acc[1], 0, 0, 0);             \
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV.y, B0.z, acc[0], 0, 0, 0);             \
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV.y, B1.z, acc[1], 0, 0, 0);             \
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV.z, B0.y, acc[0], 0, 0, 0);             \
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV.z, B1.y, acc[1], 0, 0, 0);             \
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV.w, B0.x, acc[0], 0, 0, 0);             \
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV.w, B1.x, acc[1], 0, 0, 0);             \
    }
    // A operands: ring of 4 registers, each reloaded for group g+4 right after group g's MFMAs
    // (3 groups = 24 MFMAs = 768 pipe cycles ahead of use: an L2 round trip).  B operands: one
    // group ahead (LDS latency).  sched_barrier(0) pins "loads first, then this group's MFMAs".
    float4 a0 = t_at(0), a1 = t_at(64), a2 = t_at(128), a3 = t_at(192);
    float4 bE0, bE1, bO0, bO1;
    ldb(bE0, bE1);                  // B of group 0
    int32_t poff = 192;             // table offset (float4) of the newest A in flight
    int32_t grp = 0;
#define HIPSOXR_STEP(AR, BC0, BC1, BN0, BN1)                                                   \
    ldb(BN0, BN1);                  /* B of the next group */                                   \
    __builtin_amdgcn_sched_barrier(0);                                                          \
    HIPSOXR_MFMA8(AR, BC0, BC1)                                                                 \
    poff += 64;                                                                                 \
    asm volatile("" : "+s"(poff)); /* opaque: the pipeline must not be re-rolled */             \
    AR = t_at(poff);                /* A of group +4 */
    for (; grp + 3 < n_groups; grp += 4) {
        HIPSOXR_STEP(a0, bE0, bE1, bO0, bO1)
        HIPSOXR_STEP(a1, bO0, bO1, bE0, bE1)
        HIPSOXR_STEP(a2, bE0, bE1, bO0, bO1)
        HIPSOXR_STEP(a3, bO0, bO1, bE0, bE1)
    }
    // 0..3 remaining groups (their A operands are already in a0..a2, B of the first in bE)
    if (grp < n_groups) {
        ldb(bO0, bO1);
        __builtin_amdgcn_sched_barrier(0);
        HIPSOXR_MFMA8(a0, bE0, bE1)
        if (grp + 1 < n_groups) {
            ldb(bE0, bE1);
            __builtin_amdgcn_sched_barrier(0);
            HIPSOXR_MFMA8(a1, bO0, bO1)
            if (grp + 2 < n_groups) {
                __builtin_amdgcn_sched_barrier(0);
                HIPSOXR_MFMA8(a2, bE0, bE1)
            }
        }
    }
#undef HIPSOXR_STEP
#undef HIPSOXR_MFMA8
}

// Stage one slab of k_tile_mfma_p: sample n -> plane (n & 3), index (n / Mc) * R + (n % Mc) / 4.
// The CU's matrix pipes are saturated by other waves while this runs, and every ordinary VALU
// instruction queues behind 32-cycle MFMA issues, so the code is VALU-lean: interior slabs (the
// common case) take a path with no bounds tests, no division (the (row, column) of a thread's next
// quad advances incrementally) and 32-bit offsets from a wave-uniform base; loads are issued in
// batches of UNR before any is consumed.
template <typename IO, typename Real = float>
__device__ __forceinline__ void stage_planes(const TileArgs &a, Real *xs, uint32_t clip, uint32_t ch,
                                             int64_t bw)
{
    typedef IO IO4 __attribute__((ext_vector_type(4)));
    constexpr int UNR = 4;
    const int32_t Mc = (int32_t)a.Mc, R = a.rowR, PLANE = a.plane;
    const IO *xin = (const IO *)a.in + (int64_t)clip * a.ics + (int64_t)ch * a.ichs;
    const int64_t loc_base = bw * a.Mc + a.i_min - a.in_abs0;
    const int32_t n4 = a.x_count >> 2, stride = (int32_t)blockDim.x, Mq = Mc >> 2;
    const bool fast = a.ifs == 1 && ((loc_base & 3) == 0) && loc_base >= 0 &&
                      loc_base + a.x_count <= a.in_frames &&
                      ((reinterpret_cast<uintptr_t>(xin) & (4 * sizeof(IO) - 1)) == 0);
    if (fast) {
        const IO4 *src = reinterpret_cast<const IO4 *>(xin + loc_base); // wave-uniform base
        const int32_t drow = stride / Mq, dcol = stride - drow * Mq;     // uniform step of (row, col)
        int32_t q = threadIdx.x, row = q / Mq, colq = q - row * Mq;
        while (q < n4) {
            IO4 v[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u)
                if (q + u * stride < n4) v[u] = src[q + u * stride];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                if (q + u * stride < n4) {
                    const int32_t m = row * R + colq;
                    xs[m] = (Real)v[u].x;
                    xs[m + PLANE] = (Real)v[u].y;
                    xs[m + 2 * PLANE] = (Real)v[u].z;
                    xs[m + 3 * PLANE] = (Real)v[u].w;
                }
                row += drow; colq += dcol;
                if (colq >= Mq) { colq -= Mq; ++row; }
            }
            q += stride * UNR;
        }
    } else {
        for (int32_t q0 = threadIdx.x; q0 < n4; q0 += stride) {
            const int64_t l = loc_base + ((int64_t)q0 << 2);
            IO4 v = (IO4){0, 0, 0, 0};
            if (l >= 0 && l < a.in_frames) v.x = xin[l * a.ifs];
            if (l + 1 >= 0 && l + 1 < a.in_frames) v.y = xin[(l + 1) * a.ifs];
            if (l + 2 >= 0 && l + 2 < a.in_frames) v.z = xin[(l + 2) * a.ifs];
            if (l + 3 >= 0 && l + 3 < a.in_frames) v.w = xin[(l + 3) * a.ifs];
            const int32_t row = q0 / Mq, m = row * R + (q0 - row * Mq);
            xs[m] = (Real)v.x;
            xs[m + PLANE] = (Real)v.y;
            xs[m + 2 * PLANE] = (Real)v.z;
            xs[m + 3 * PLANE] = (Real)v.w;
        }
    }
}

template <typename IO>
__global__ void __launch_bounds__(1024, 2) k_tile_mfma_p(TileArgs a)
{
    typedef float Real;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int32_t Mc = (int32_t)a.Mc, R = a.rowR, PLANE = a.plane, padR = R - Mc / 4;
    Real *xs = reinterpret_cast<Real *>(smem_raw) + R; // one row of slack below (pipelined reads run one group past the end)

    const uint32_t col = blockIdx.y;
    // run-time division goes through the vector ALU; readfirstlane keeps the results (and every
    // address derived from them) on the scalar side
    const uint32_t ch = __builtin_amdgcn_readfirstlane(col % a.n_channels), clip = __builtin_amdgcn_readfirstlane(col / a.n_channels);
    uint32_t bxi = blockIdx.x, bz = blockIdx.z, nz = gridDim.z;
    if (a.xz) {
        const uint32_t slot = blockIdx.x >> 3;
        nz = (uint32_t)a.xz;
        bz = __builtin_amdgcn_readfirstlane(slot % nz);
        bxi = __builtin_amdgcn_readfirstlane((slot / nz) * 8 + (blockIdx.x & 7u));
        if (bxi >= (uint32_t)a.nx) return; // grid.x is padded to a multiple of 8 slabs
    }
    const int64_t bw = a.b_first + (int64_t)bxi * a.pb; // slabs of 64 periods; of 32 for jobs of few slabs (launch_tile)
    const int64_t k_end = a.out_k0 + a.out_frames;
    unsigned long long *tr = a.trace ? a.trace + ((size_t)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6)) * 16 : nullptr;
    int tri = 0;
#define HIPSOXR_STAMP() do { if (tr && (threadIdx.x & 63) == 0 && tri < 16) tr[tri] = __builtin_amdgcn_s_memtime(); ++tri; } while (0)
    HIPSOXR_STAMP();

    stage_planes<IO>(a, xs, clip, ch, bw);
    HIPSOXR_STAMP();
    __syncthreads();
    HIPSOXR_STAMP();

    const int lane = threadIdx.x & 63;
    const int kq = lane >> 4, j = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_waves = a.n_waves;
    const int32_t n_groups = a.I_h >> 4;
    const size_t half_stride = (size_t)(n_groups + 4) * 64; // float4 per half table (+4 groups of prefetch slack)
    const Real *xL = xs + kq * PLANE + j * R;        // left : lane k reads plane k
    const Real *xR = xs + (3 - kq) * PLANE + j * R;  // right: lane k reads plane 3-k
    IO *const ybase = (IO *)a.out + (int64_t)clip * a.ocs + (int64_t)ch * a.ochs;
    const bool interior = bw * a.Lc >= a.out_k0 && (bw + a.pb) * a.Lc <= k_end;
    const int hp = a.pb >> 5; // units per row tile: halves of a 64-period slab, or the one 32-period slab
    // (rotating which waves take the odd units of a split slab with the slab index changes nothing: measured)

    // Work unit = (tile, half of the 64 periods).  A workgroup runs 4 waves — exactly one per SIMD,
    // because 10-wave workgroups land 3/3/2/2 on the SIMDs and leave 17 % of the matrix pipe idle
    // (tools/ubench/mfma_loop.hip) — and its 2*n_rt equal units are dealt round-robin.
    // Small jobs additionally split a slab's units over gridDim.z workgroups (each stages the slab).
    for (int u_ = wave + n_waves * (int)bz; u_ < hp * a.n_rt; u_ += n_waves * (int)nz) {
        const int unit = __builtin_amdgcn_readfirstlane(u_);
        const int rt = hp == 2 ? unit >> 1 : unit, ph = hp == 2 ? unit & 1 : 0; // periods 32*ph .. 32*ph + 31
        const int32_t wL = __builtin_amdgcn_readfirstlane(a.e0[rt * 2 + 0]), wR = __builtin_amdgcn_readfirstlane(a.e0[rt * 2 + 1]);
        const int32_t eL0 = wL & 0xffffff, eR0 = wR & 0xffffff; // multiples of 16
        const int32_t gL = (a.dbg & 16) ? n_groups : wL >> 24, gR = (a.dbg & 16) ? n_groups : wR >> 24; // groups this tile's half-chains need (build_mfma_planes; HIPSOXR_DEBUG_FLAGS 16: all of them)
        const char *tL = (const char *)a.tab + (size_t)(rt * 2 + 0) * half_stride * 16; // (wave-uniform; lanes add lane * 16)
        const char *tR = tL + half_stride * 16;
        const uint32_t lane16 = (uint32_t)lane * 16;
        f32x4 accL[2], accR[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) { accL[g] = (f32x4){0, 0, 0, 0}; accR[g] = (f32x4){0, 0, 0, 0}; }

        mfma_half_chain<false>(accL, tL, lane16, xL + ph * 32 * R, eL0, gL, Mc, R, padR);
        mfma_half_chain<true>(accR, tR, lane16, xR + ph * 32 * R, eR0, gR, Mc, R, padR);
        HIPSOXR_STAMP();

        const int32_t r0 = rt * 16 + 4 * kq;
        if (interior && rt * 16 + 16 <= a.Lc && a.ofs == 1) {
            // whole unit in range, unit stride: 32-bit offsets from the slab's first output
            IO *const yw = ybase + (bw * a.Lc - a.out_k0);           // wave-uniform
            const int32_t o0 = (32 * ph + j) * (int32_t)a.Lc + r0;   // this lane, group 0
            const int64_t kw = bw * a.Lc;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int32_t o = o0 + 16 * g * (int32_t)a.Lc;
#pragma unroll
                for (int vv = 0; vv < 4; ++vv)
                    store_out<Real>(yw + o + vv, accL[g][vv] + accR[g][vv], a.oc, ch, kw + o + vv);
            }
        } else {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int64_t b = bw + 32 * ph + 16 * g + j;
                const int64_t k0 = b * a.Lc + r0;
                IO *const yt = ybase + (k0 - a.out_k0) * a.ofs;
#pragma unroll
                for (int vv = 0; vv < 4; ++vv) {
                    const int64_t idx = k0 + vv - a.out_k0;
                    if (r0 + vv < a.Lc && idx >= 0 && idx < a.out_frames)
                        store_out<Real>(yt + vv * a.ofs, accL[g][vv] + accR[g][vv], a.oc, ch, k0 + vv);
                }
            }
        }
    }
    tri = 15;
    HIPSOXR_STAMP();
#undef HIPSOXR_STAMP
}

// ---------------------------------------------------------------------------------------------
// k_tile_mfma64_p — the float64 engine (float64 / int32 I/O) in the planar form (round 3), for input periods that are
// a multiple of 16.  Same idea as k_tile_mfma_p: the slab k-de-interleaved into four LDS planes, so that every offset
// inside a half-chain is a wave-uniform scalar and the vector ALU — which costs the matrix pipe ~4 cycles per
// instruction while it runs beside it — does nothing but issue MFMAs: k_tile_mfma<IO, double, NG> spends ten VALU
// instructions per v_mfma_f64 on per-lane index bookkeeping (rocprofv3: 11.3 M VALU against 1.08 M MFMA per launch)
// and reaches 18 TFLOP/s of the 78 the pipe sustains (tools/ubench/mfma_f64_rate.hip).  What differs from the f32 form:
//   * v_mfma_f64_16x16x4_f64 takes 64 cycles, twice the f32 form: a slab is 32 periods (8 bytes per sample: 51 KB at
//     48k -> 44.1k, three workgroups per CU), a work unit is one row tile across all 32 periods (2 accumulators);
//   * a 16-byte access carries TWO samples: a group of 16 inputs is two ds_read_b128 per 16 periods and two
//     global_load_dwordx4 of coefficients per lane, both one group (8 MFMAs = 512 pipe cycles) ahead of use;
//   * the accumulator layout is row (lane >> 4) + 4 v (MfmaOf<double>::row).
// Canonical order as everywhere: groups ascending (left) / descending (right), chunks and k inside them likewise.
// ---------------------------------------------------------------------------------------------
// NG = 2: one unit = a row tile across the slab's 32 periods (two accumulators share every coefficient load);
// NG = 1: a unit is a row tile across 16 periods — twice as many, half as long: the four waves of a workgroup then
// share 2 n_rt units evenly where n_rt is not a multiple of four (147 phases = 10 tiles: 3/3/2/2 -> 5/5/5/5).
template <bool RIGHT, int NG>
__device__ __forceinline__ void mfma64_half_chain(f64x4 (&acc)[NG], const double *tbase, uint32_t lane_bytes, const double *xb, int32_t e0,
                                                  int32_t n_groups, int32_t Mc, int32_t R, int32_t padR)
{
    typedef double d2 __attribute__((ext_vector_type(2)));
    // (coefficients through a buffer descriptor, the group as a scalar offset: see mfma_half_chain)
    const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc((void *)tbase, 0, 0x40000000, 0x00020000);
    int32_t rem = e0 % Mc, fo = (e0 / Mc) * R + (rem >> 2); // wave-uniform plane offset of the next B read
    auto ldb = [&](d2 (&b)[2 * NG]) { // period j (and j + 16), four consecutive chunk columns each
        const double *px = xb + fo;
        b[0] = *reinterpret_cast<const d2 *>(__builtin_assume_aligned(px, 16));
        b[1] = *reinterpret_cast<const d2 *>(__builtin_assume_aligned(px + 2, 16));
        if (NG == 2) {
            b[2 * (NG - 1)] = *reinterpret_cast<const d2 *>(__builtin_assume_aligned(px + 16 * R, 16));
            b[2 * (NG - 1) + 1] = *reinterpret_cast<const d2 *>(__builtin_assume_aligned(px + 16 * R + 2, 16));
        }
        if (!RIGHT) { fo += 4; rem += 16; if (rem == Mc) { rem = 0; fo += padR; } }
        else { fo -= 4; rem -= 16; if (rem < 0) { rem += Mc; fo -= padR; } }
    };
    auto lda = [&](d2 (&av)[2], int32_t off) { // this lane's coefficients of the group's four chunks
        av[0] = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(trs, (int)lane_bytes, off * 8, 0));
        av[1] = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(trs, (int)lane_bytes, off * 8 + 16, 0));
    };
    d2 ac[2], an[2], bc[2 * NG], bn[2 * NG];
    lda(ac, 0);
    ldb(bc);
    int32_t poff = 0; // table offset (doubles) of the group whose coefficients are in flight (wave-uniform)
    for (int32_t grp = 0; grp < n_groups; ++grp) {
        poff += 256;
        asm volatile("" : "+s"(poff)); // opaque: keeps the software pipeline from being re-rolled
        lda(an, poff);                 // (the table carries four groups of slack)
        ldb(bn);                       // (the slab carries a row of slack at either end)
        __builtin_amdgcn_sched_barrier(0); // next group's operands are requested BEFORE this group's MFMAs
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const double av = c < 2 ? ac[0][c] : ac[1][c - 2];
            const int m = RIGHT ? 3 - c : c; // right half-chain: chunk c is plane column 3 - c (descending input index)
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const double bv = m < 2 ? bc[2 * g][m] : bc[2 * g + 1][m - 2];
                acc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[g], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) ac[i] = an[i];
#pragma unroll
        for (int i = 0; i < 2 * NG; ++i) bc[i] = bn[i];
    }
}

// PB = periods per slab: 32, or 16 for jobs of few slabs (half the LDS, twice the workgroups: 563 slabs of 32 periods on
// 256 CUs leave a fifth of them with three workgroups and the rest with two — the launch waits for the fifth).
template <typename IO, int NG, int PB>
__global__ void __launch_bounds__(640) k_tile_mfma64_p(TileArgs a)
{
    typedef double Real;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int32_t Mc = (int32_t)a.Mc, R = a.rowR, PLANE = a.plane, padR = R - Mc / 4;
    Real *xs = reinterpret_cast<Real *>(smem_raw) + R; // one row of slack below (pipelined reads run one group past the end)

    const uint32_t col = blockIdx.y;
    const uint32_t ch = __builtin_amdgcn_readfirstlane(col % a.n_channels), clip = __builtin_amdgcn_readfirstlane(col / a.n_channels);
    uint32_t bxi = blockIdx.x, bz = blockIdx.z, nz = gridDim.z;
    if (a.xz) { // XCD-aware ids of a unit split (see k_tile_mfma_p)
        const uint32_t slot = blockIdx.x >> 3;
        nz = (uint32_t)a.xz;
        bz = __builtin_amdgcn_readfirstlane(slot % nz);
        bxi = __builtin_amdgcn_readfirstlane((slot / nz) * 8 + (blockIdx.x & 7u));
        if (bxi >= (uint32_t)a.nx) return; // grid.x is padded to a multiple of 8 slabs
    }
    const int64_t bw = a.b_first + (int64_t)bxi * PB;
    const int64_t k_end = a.out_k0 + a.out_frames;

    stage_planes<IO, Real>(a, xs, clip, ch, bw);
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int kq = lane >> 4, j = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_waves = a.n_waves;
    const int32_t n_groups = a.I_h >> 4;
    const size_t half_stride = (size_t)(n_groups + 4) * 256; // doubles per half table (+4 groups of prefetch slack)
    const Real *xL = xs + kq * PLANE + j * R;        // left : lane k reads plane k
    const Real *xR = xs + (3 - kq) * PLANE + j * R;  // right: lane k reads plane 3-k
    IO *const ybase = (IO *)a.out + (int64_t)clip * a.ocs + (int64_t)ch * a.ochs;
    const bool interior = bw * a.Lc >= a.out_k0 && (bw + PB) * a.Lc <= k_end;

    constexpr int UPT = PB / (16 * NG); // units per row tile
    for (int u_ = wave + n_waves * (int)bz; u_ < UPT * a.n_rt; u_ += n_waves * (int)nz) { // unit = row tile x 16 NG periods
        const int unit = __builtin_amdgcn_readfirstlane(u_);
        const int rt = unit / UPT, ph = unit % UPT; // periods 16 ph ..
        const int32_t wL = __builtin_amdgcn_readfirstlane(a.e0[rt * 2 + 0]), wR = __builtin_amdgcn_readfirstlane(a.e0[rt * 2 + 1]);
        const int32_t eL0 = wL & 0xffffff, eR0 = wR & 0xffffff; // multiples of 16
        const int32_t gL = wL >> 24, gR = wR >> 24;             // groups this tile's half-chains need (build_mfma_planes)
        const Real *tL = (const Real *)a.tab + (size_t)(rt * 2 + 0) * half_stride; // (wave-uniform; a lane's column starts lane * 32 bytes in)
        const Real *tR = (const Real *)a.tab + (size_t)(rt * 2 + 1) * half_stride;
        f64x4 accL[NG], accR[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) { accL[g] = (f64x4){0, 0, 0, 0}; accR[g] = (f64x4){0, 0, 0, 0}; }
        mfma64_half_chain<false, NG>(accL, tL, (uint32_t)lane * 32, xL + ph * 16 * R, eL0, gL, Mc, R, padR);
        mfma64_half_chain<true, NG>(accR, tR, (uint32_t)lane * 32, xR + ph * 16 * R, eR0, gR, Mc, R, padR);

        const int32_t rbase = rt * 16; // this lane: rows rbase + kq + 4 v, periods bw + 16 g + j
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int64_t b = bw + 16 * (g + ph) + j;
            const int64_t kb = b * a.Lc + rbase;
            IO *const yt = ybase + (kb - a.out_k0) * a.ofs;
            if (interior && rbase + 16 <= a.Lc) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int r = kq + 4 * v;
                    store_out<Real>(yt + r * a.ofs, accL[g][v] + accR[g][v], a.oc, ch, kb + r);
                }
            } else {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int r = kq + 4 * v;
                    const int64_t idx = kb + r - a.out_k0;
                    if (rbase + r < a.Lc && idx >= 0 && idx < a.out_frames)
                        store_out<Real>(yt + r * a.ofs, accL[g][v] + accR[g][v], a.oc, ch, kb + r);
                }
            }
        }
    }
}

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
