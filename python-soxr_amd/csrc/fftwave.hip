// fftwave.hip — frequency-domain engine, throughput form (round 6): ONE WAVE per pair of blocks, two register passes per
// transform.
//
// k_fft_pair2 (fft.hip) runs a 5120 -> 4704-point block pair on a 384-thread workgroup: three Stockham passes per
// transform, five workgroup barriers, a quarter of the threads idle in any one pass, 40 KB of LDS.  Its waves spend 18 %
// of their life issuing vector instructions and most of the rest parked (profiles/NOTES_r05.md §1).  Here the block is
// 24 periods — 3840 -> 3528 points at 48k -> 44.1k — because both lengths then split into two factors of about a wave:
//
//   3840 = 60 * 64:  pass 0 = radix 60 on 64 lanes,  pass 1 = radix 64 on 60 lanes
//   3528 = 56 * 63:  pass 0 = radix 56 on 63 lanes,  pass 1 = radix 63 on 56 lanes
//
// One wave owns the pair: NO workgroup barrier anywhere (an LDS write followed by an LDS read of the same wave is ordered
// by the hardware), every lane busy in every pass, one inter-pass twiddle stage per transform instead of two (the
// twiddles inside the radix-60/56/63 butterflies do not exist — prime-factor maps — and those inside radix 64 are
// literals), three LDS exchanges per pair instead of five.  The exchanges move the real parts, then the imaginary parts,
// through ONE buffer of N floats: 15.6 KB per wave, so a CU holds eight waves (two per SIMD, 256 registers each) — what a
// lone wave cannot do (one vector instruction per 4.4-4.8 cycles, tools/ubench/valu_issue.hip) two waves per SIMD can.
// Kept outputs: 22 of 24 periods (k_fft_pair2: 30 of 32).  5300 vector instructions per pair of 3234-output blocks (4540 with the
// packed FMAs of fft_dev.h) against k_fft_pair2's 8770 per pair of 4410.
// Measured (profiles/NOTES_r06.md §1-§2, the 128 x 10 s batch, one box): 44.1k -> 48k 116.6-116.9 us against 124.9-127.0 for
// k_fft_pair2 — taken, from 8192 pairs up (four full rounds of the 2048 wave slots; below that the partly filled last round
// costs more than the form gains, and a single pair's latency is 39 k cycles on one wave); 48k -> 44.1k 117.0 against 114.0 —
// NOT taken (`min_pairs` in the table below; HIPSOXR_DEBUG_WAVE_MIN in the debug-switch build forces either, which is how
// tests/test_gpu_fft_wave.py covers both).  Float32 unit-stride columns (mono / planar / batches, ragged included).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "device.h"
#include "fft_dev.h"

#ifdef FFT2_TRACE
#include <cstdio>
#include <vector>
#endif

namespace hipsoxr {

// -DFFT2_TRACE: per-wave s_memtime stamps (tools/trace_wave.py): 0 start | 1 loads + forward pass 0 | 2 exchange | 3 twiddles +
// forward pass 1 | 4 spectrum exchange | 5 inverse pass 0 | 6 exchange | 7 twiddles + inverse pass 1 | 8 run a stored |
// 9 run b stored | 13 HW_ID | 14 XCC_ID | 15 end
#ifdef FFT2_TRACE
#define WSTAMP() do { __builtin_amdgcn_sched_barrier(0); if (g_tr && lane == 0) g_tr[g_tri] = __builtin_amdgcn_s_memtime(); ++g_tri; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
// (phase boundary: the scheduler may not carry loads of a later phase — filter gains, twiddle bases, the next item's input —
//  up into a butterfly whose 128 data registers plus temporaries already fill the budget)
#define WSTAMP() __builtin_amdgcn_sched_barrier(0)
#endif

// (N, pass-0 radix, pass-1 radix, exchange stride): pass 0 runs on N / R0 = R1 lanes, pass 1 on R0 lanes.  S = odd >= R0:
// lane j writes its R0 outputs at j * S + q (lanes S dwords apart: conflict-free for odd S), lane k reads t * S + k.
template <int N> struct WaveSched;
template <> struct WaveSched<3840> { static constexpr int R0 = 60, R1 = 64, S = 61; };
template <> struct WaveSched<3528> { static constexpr int R0 = 56, R1 = 63, S = 57; };

template <int NA_, int NB_, int V0_, int HOP_, int LEADIN_, int HOPIN_> struct WaveSpec {
    static constexpr int NA = NA_, NB = NB_, V0 = V0_, HOP = HOP_;
    static constexpr int HOPIN = HOPIN_;   // input samples between the two blocks of a pair (hop_periods * M)
    static constexpr int LEADIN = LEADIN_; // input samples of the lead periods: what a column's first block reads before the column's start
    typedef WaveSched<NA_> A;
    typedef WaveSched<NB_> B;
    static constexpr int cmax(int a, int b) { return a > b ? a : b; }
    // floats of LDS: the two exchange buffers, the spectrum (+ one row of slack for the idle lanes' reads), the staged run
    static constexpr int LDSN = cmax(cmax(64 * A::S, 64 * B::S), cmax(cmax(NA, NB) + 64, HOP + 8));
};

// the wave's LDS traffic is ordered by the hardware; what must not happen is the COMPILER moving a read above a write
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The lane index again, opaque to the compiler: inside the persistent loop every address derived from the plain index is
// loop-invariant, gets hoisted, and ~50 hoisted registers beside a radix-64 butterfly spill; derived from this they are
// recomputed (one or two instructions each) where they are used and die there.
__device__ __forceinline__ int opaque(int lane)
{
    asm volatile("" : "+v"(lane));
    return lane;
}

// pass 0 -> pass 1 through LDS, real parts then imaginary parts: u[q] (q < R0) of lane j is output q of butterfly j;
// lane k takes v[t] = output k of butterfly t (t < R1).  All 64 lanes move data (rows / columns past the live ones hold
// garbage that nobody uses; the buffer has 64 rows).
template <int R0, int R1, int S, typename C> __device__ __forceinline__ void wave_exchange(float *lds, int row, int lane, const C *u, C *v)
{
    float *const wp = lds + row * S; // row = the butterfly this lane ran
    const float *const rp = lds + lane;
#pragma unroll
    for (int q = 0; q < R0; ++q) wp[q] = u[q].x;
    wave_sync();
#pragma unroll
    for (int t = 0; t < R1; ++t) v[t].x = rp[t * S];
    wave_sync();
#pragma unroll
    for (int q = 0; q < R0; ++q) wp[q] = u[q].y;
    wave_sync();
#pragma unroll
    for (int t = 0; t < R1; ++t) v[t].y = rp[t * S];
    wave_sync();
}

// v[t] *= W^(k t), t < R: six table entries W[k 2^i] (exactly rounded), every power the product of the entries of its
// set bits — W^(k t) = W^(k (t & (t - 1))) * W^(k lowbit(t)) — so an element costs one product to form and one to apply,
// at most five roundings deep, and only the chain of a power's ancestors is live at any time.
template <int R, typename C> __device__ __forceinline__ void wave_twiddle(C *v, const C *W, int k)
{
#ifdef WAVE_NO_TW // (timing experiment: no inter-pass twiddles; results are wrong)
    return;
#endif
    constexpr int NBITS = R <= 2 ? 1 : R <= 4 ? 2 : R <= 8 ? 3 : R <= 16 ? 4 : R <= 32 ? 5 : R <= 64 ? 6 : 7;
    C base[NBITS];
#pragma unroll
    for (int i = 0; i < NBITS; ++i) base[i] = W[k << i];
    C pw[R];
#pragma unroll
    for (int t = 1; t < R; ++t) {
        const int lb = t & -t, i = __builtin_ctz(t);
        pw[t] = t == lb ? base[i] : cmul(pw[t - lb], base[i]);
        v[t] = cmul(v[t], pw[t]);
    }
}

// One block's kept run — outputs [V0, V0 + HOP) of the block, `val(s)` = this lane's output n = lane + R0 s — through LDS
// to memory as whole 16-byte granules (non-temporal), index-shifted so that LDS and memory agree on 16-byte phase; the
// descriptor ends with the run (or the column), the hardware range check drops the rest.  The first granule's leading
// elements belong to the previous run: that granule goes element by element.
template <typename Spec, typename Val>
__device__ __forceinline__ void wave_store_run(float *lds, int lane, Val val, float *yrun, int valid)
{
    constexpr int R0 = Spec::B::R0, R1 = Spec::B::R1, V0 = Spec::V0, V1 = Spec::V0 + Spec::HOP;
    const int sh = (int)((reinterpret_cast<uintptr_t>(yrun) / 4) & 3);
    if (lane < R0) {
        float *const sp = lds + (lane + sh); // staged index of output n: n - V0 + sh
#pragma unroll
        for (int s = 0; s < R1; ++s) {
            const int lo = R0 * s, hi = lo + R0 - 1; // the outputs this s can be, over the lanes
            if (hi < V0 || lo >= V1) continue;
            if (lo >= V0 && hi < V1) sp[lo - V0] = val(s);
            else if (lane + lo >= V0 && lane + lo < V1) sp[lo - V0] = val(s);
        }
    }
    wave_sync();
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr((void *)(yrun - sh)), 0,
                                                                         __builtin_amdgcn_readfirstlane((valid + sh) * 4), 0x00020000);
    constexpr int QMAX = (Spec::HOP + 3 + 3) / 4;
#pragma unroll
    for (int it = 0; it < (QMAX + 63) / 64; ++it) {
        const int q = lane + it * 64;
        const float4 v = *reinterpret_cast<const float4 *>(lds + 4 * (q < Spec::LDSN / 4 ? q : Spec::LDSN / 4 - 1));
        if (q == 0 && sh != 0) {
            const float *e = reinterpret_cast<const float *>(&v);
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c >= sh && c - sh < valid) yrun[c - sh] = e[c];
        } else {
#ifdef WAVE_NO_STORE // (timing experiment: no output traffic beyond one granule per wave and run)
            if (q == 1)
#endif
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u_t, v), ro, q * 16, 0, FFT_STORE_AUX);
        }
    }
    wave_sync();
}

#ifndef WAVE_WPE
#define WAVE_WPE 2, 2
#endif
#ifndef WAVE_XCD_MAP
#define WAVE_XCD_MAP 1
#endif
#ifndef WAVE_PERSIST // 1: as many waves as the chip holds, each walking items and prefetching the next one's input; 0: a wave per item
#define WAVE_PERSIST 0 // (measured: 140 against 117 us — a wave's loads are issued IN ORDER with its arithmetic, and with the memory
#endif                 //  system saturated 60 loads in a row hold the wave's butterflies up for as long as the queues take them)
// One work item = a pair of blocks of one column: item = col * pairs + bx -> blocks 2 bx, 2 bx + 1.  All of it wave-uniform.
struct WaveItem {
    const float *src; // first sample the pair may read: sample max(ina, in_lo) of the column
    int32_t nbytes;   // bytes from there to the end of the column (buffer range: loads past it return 0)
    int32_t zlo;      // leading samples of the first block that lie before the column's start (first pair of a column)
    float *ya;        // run a's first element
    int64_t remain;   // outputs from there to the end of the column
};

// PERSISTENT waves: the launch has as many waves as the chip holds (8 per CU), wave w walks items w, w + W, w + 2 W ... and
// requests the NEXT item's 2 x 60 input loads as soon as the registers that receive them are free — behind the last
// butterfly of the current item — so that HBM latency hides behind the staging and the stores of the current item instead of
// heading every item (a lone-item wave: 39 k cycles of which 7 k wait for the input; chip-wide the launches' compute-only
// time and memory-only time ADD when only two waves per SIMD can overlap them: profiles/NOTES_r06.md).
template <typename Spec>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVE_WPE))) k_fft_wave(FftArgs a)
{
    typedef float2 C;
    typedef typename Spec::A SA;
    typedef typename Spec::B SB;
    constexpr int NA = Spec::NA, NB = Spec::NB, L0A = SA::R1;
    __shared__ __attribute__((aligned(16))) float lds[Spec::LDSN];
    const int lane = (int)threadIdx.x;
    const uint32_t pairs = (uint32_t)a.pairs_per_col, total = pairs * a.n_clips * a.n_channels, W = gridDim.x;

    // XCD-aware ids: consecutive workgroup ids go round-robin to the 8 XCDs, each with an L2 of its own, while neighbouring
    // pairs of a column share 8 % of their input: id v -> item (v & 7) * per_xcd + (v >> 3), so an XCD walks a contiguous run.
    const uint32_t per_xcd = (total + 7) / 8, vtotal = 8 * per_xcd;
    auto setup = [&](uint32_t vid, WaveItem &w) -> bool {
        const uint32_t item = WAVE_XCD_MAP ? (vid & 7) * per_xcd + (vid >> 3) : vid;
        if (item >= total) return false;
        const uint32_t col = __builtin_amdgcn_readfirstlane(item / pairs), bx = item - col * pairs;
        const uint32_t clip = __builtin_amdgcn_readfirstlane(col / a.n_channels), ch = col - clip * a.n_channels;
        const int64_t pa = 2 * (int64_t)bx * a.hop_periods - a.lead_periods; // first period of the pair's first block; the second starts hop_periods later
        const int64_t ina = pa * a.M, outa = pa * a.L;
        int64_t clip_in = (int64_t)clip * a.ics, clip_out = (int64_t)clip * a.ocs, in_frames = a.in_frames, out_frames = a.out_frames;
        if (a.clip_tab) { // ragged batch: this clip's own place and length
            const int64_t *row = a.clip_tab + 4 * (size_t)clip;
            clip_in = row[0]; in_frames = row[1]; clip_out = row[2]; out_frames = row[3];
        }
        if (outa + Spec::V0 >= out_frames) return false; // beyond its clip's last pair
        const int64_t zlo = ina < a.in_lo ? a.in_lo - ina : 0, left = (in_frames - (ina + zlo)) * 4;
        w.src = (const float *)a.in + clip_in + (int64_t)ch * a.ichs + (ina + zlo);
        w.nbytes = (int32_t)(left < 0 ? 0 : left > 0x40000000 ? 0x40000000 : left);
        w.zlo = (int32_t)zlo;
        w.ya = (float *)a.out + clip_out + (int64_t)ch * a.ochs + (outa + Spec::V0); // outa + V0 >= 0
        w.remain = out_frames - (outa + Spec::V0);
        return true;
    };
    // forward pass 0's inputs: butterfly (column) j takes x[j + L0 t], t < R0 (z = x_a + i x_b), straight from HBM through a
    // descriptor whose base is the first block's sample 0 — for a column's first pair that lies zlo samples BEFORE the
    // column (never dereferenced there: only the first loads of block a can fall in front of the column's start, and those clamp
    // their index and select 0) — so every offset is a compile-time constant in the instruction; past the end of the column the
    // hardware's range check returns zeros.
    // L0 = 64: 8-byte loads.  Lane l loads x[2 l + 128 t'], x[2 l + 1 + 128 t']: columns 2 l and 2 l + 1 at t = 2 t' for l < 32,
    // the same two columns at t = 2 t' + 1 for lane l + 32.  One v_permlane32_swap per load hands each half-wave what the other
    // holds of its column: lane l < 32 ends up with column 2 l, lane l + 32 with column 2 l + 1, both at t = 2 t' and 2 t' + 1.
    // Half the vector-memory instructions — and a wave can have only 63 of them in flight (`vmcnt`): 120 dword loads in a row
    // stall their own issue for a whole memory latency, 60 do not.  Parts [T0, T1) of the R0 inputs per call: the next item's
    // loads are spread over the tail of the current one (see the loop).
    auto issue_loads = [&](const WaveItem &w, unsigned *raw, auto T0c, auto T1c) {
        constexpr int T0 = decltype(T0c)::value, T1 = decltype(T1c)::value;
        const int zlo = __builtin_amdgcn_readfirstlane(w.zlo), lane = opaque((int)threadIdx.x);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr((void *)(w.src - zlo)), 0,
                                                                             __builtin_amdgcn_readfirstlane(w.nbytes + zlo * 4), 0x00020000);
        auto load_front = [&](int idx) -> unsigned { // block a, where a sample may lie in front of the column
            const unsigned x = __builtin_amdgcn_raw_buffer_load_b32(rs, (idx < zlo ? zlo : idx) * 4, 0, FFT_LOAD_AUX);
            return idx < zlo ? 0u : x;
        };
#ifdef WAVE_NO_LOAD // (timing experiment: no input traffic)
#pragma unroll
        for (int t = T0; t < T1; ++t) { raw[2 * t] = __builtin_bit_cast(unsigned, (float)(lane + t) * 1e-3f); raw[2 * t + 1] = __builtin_bit_cast(unsigned, (float)(lane - t) * 1e-3f); }
        return;
#endif
        if constexpr (L0A == 64) { // raw[4 t' ..] = the pair of block a, the pair of block b
            static_assert(SA::R0 % 2 == 0 && T0 % 2 == 0 && T1 % 2 == 0 && Spec::HOPIN % 2 == 0, "pairs of inputs");
#pragma unroll
            for (int tp = T0 / 2; tp < T1 / 2; ++tp) {
                const v2u_t pb = __builtin_amdgcn_raw_buffer_load_b64(rs, lane * 8, (128 * tp + Spec::HOPIN) * 4, FFT_LOAD_AUX);
                if (128 * tp >= Spec::LEADIN) {
                    const v2u_t pa = __builtin_amdgcn_raw_buffer_load_b64(rs, lane * 8, 128 * tp * 4, FFT_LOAD_AUX);
                    raw[4 * tp] = pa.x; raw[4 * tp + 1] = pa.y;
                } else {
                    raw[4 * tp] = load_front(2 * lane + 128 * tp); raw[4 * tp + 1] = load_front(2 * lane + 1 + 128 * tp);
                }
                raw[4 * tp + 2] = pb.x; raw[4 * tp + 3] = pb.y;
            }
        } else { // raw[2 t] = block a, raw[2 t + 1] = block b
#pragma unroll
            for (int t = T0; t < T1; ++t) {
                raw[2 * t] = t * L0A >= Spec::LEADIN ? __builtin_amdgcn_raw_buffer_load_b32(rs, lane * 4, t * L0A * 4, FFT_LOAD_AUX) : load_front(lane + t * L0A);
                raw[2 * t + 1] = __builtin_amdgcn_raw_buffer_load_b32(rs, lane * 4, (t * L0A + Spec::HOPIN) * 4, FFT_LOAD_AUX);
            }
        }
    };
    // ... and what turns the loaded words into the butterfly's inputs — the swaps — runs at the head of the item that uses them:
    // behind the loads themselves it would make the wave wait for data it has only just asked for.
    auto unpack = [&](const unsigned *raw, C *u) {
        if constexpr (L0A == 64) {
#pragma unroll
            for (int tp = 0; tp < SA::R0 / 2; ++tp) {
                const v2u_t sa = __builtin_amdgcn_permlane32_swap(raw[4 * tp], raw[4 * tp + 1], false, false);
                const v2u_t sb = __builtin_amdgcn_permlane32_swap(raw[4 * tp + 2], raw[4 * tp + 3], false, false);
                // (by value: __builtin_bit_cast applied to the vector-element expression `sa.y` itself reads element 0)
                const unsigned ax = sa.x, ay = sa.y, bx = sb.x, by = sb.y;
                u[2 * tp] = C(__builtin_bit_cast(float, ax), __builtin_bit_cast(float, bx));
                u[2 * tp + 1] = C(__builtin_bit_cast(float, ay), __builtin_bit_cast(float, by));
            }
        } else {
#pragma unroll
            for (int t = 0; t < SA::R0; ++t) u[t] = C(__builtin_bit_cast(float, raw[2 * t]), __builtin_bit_cast(float, raw[2 * t + 1]));
        }
    };
    // (which column a lane holds after the loads)
    auto column = [](int lane) -> int { return L0A == 64 ? ((lane & 31) << 1) | (lane >> 5) : lane; };
    constexpr std::integral_constant<int, 0> kT0{};
    constexpr std::integral_constant<int, SA::R0 / 4 * 2> kTm{};
    constexpr std::integral_constant<int, SA::R0> kT1{};

    uint32_t item = blockIdx.x;
    WaveItem cur;
    bool ok = false;
    while (item < vtotal && !(ok = setup(item, cur))) item += W;
    if (!ok) return;
    unsigned raw[2 * SA::R0];
    issue_loads(cur, raw, kT0, kT1);
    const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr((void *)a.Hr), 0, (NB / 2 + 1) * 4, 0x00020000);

    for (;;) {
#ifdef FFT2_TRACE
        unsigned long long *g_tr = a.trace ? a.trace + (size_t)item * 16 : nullptr;
        int g_tri = 0;
        WSTAMP();
#endif
        C u[SA::R0];
        unpack(raw, u);
        dft_r<SA::R0, -1>(u);
        WSTAMP();

        // ---- forward pass 1: lane k takes output k of every butterfly, times W^(k t); radix R1 -> X[k + R0 s] ------
        C v[SA::R1];
        { const int ln = opaque(lane); wave_exchange<SA::R0, SA::R1, SA::S>(lds, column(ln), ln, u, v); }
        WSTAMP();
        { const int ln = opaque(lane); wave_twiddle<SA::R1>(v, a.WA2, ln < SA::R0 ? ln : SA::R0 - 1); }
        dft_r<SA::R1, -1>(v);
        WSTAMP();

        // ---- spectrum: bin n of the output grid <- bin n (non-negative frequencies, n <= NB/2) or n + NA - NB (negative
        //      ones) of the input grid, times the real gain of |frequency|.  Lane j of inverse pass 0 takes bins j + L0B t.
        //      Registers decide the order: the 128 of X leave through LDS first (real parts, then imaginary parts), and only
        //      when the last of them has been issued are the 56-60 gains requested — into the registers X just left, their
        //      trip to L2 behind the last LDS reads.  (Gains held beside X, on either side of the exchange, spill.) ----
        constexpr int L0B = SB::R1, QH = NB < NA ? NB / 2 : NA / 2 - 1; // lanes of inverse pass 0; the last |frequency| that passes
        const int lsp = opaque(lane), jr = lsp < L0B ? lsp : L0B - 1;
        auto bin_used = [](int s) -> bool { // (bins no output bin reads — the truncated middle of the spectrum — are not written)
            const int lo = SA::R0 * s, hi = lo + SA::R0 - 1;
            return !(lo > QH && hi < NA - QH);
        };
        C y[SB::R0];
        int sidx[SB::R0]; // (compile-time offsets from one or two lane-dependent bases: folded by the compiler)
#pragma unroll
        for (int t = 0; t < SB::R0; ++t) {
            const int lo = t * L0B, hi = lo + L0B - 1, n = jr + lo; // bin n of the output grid
            const bool neg = hi <= NB / 2 ? false : lo > NB / 2 ? true : n > NB / 2; // (compile-time for all but the t that straddles NB/2)
            sidx[t] = neg ? n + (NA - NB) : n;
            if (NA < NB) { // zero extension: bins past the input's band read anything (their gain is 0)
                const int q = neg ? NB - n : n;
                sidx[t] = q <= QH ? (neg ? NA - q : q) : 0;
            }
        }
        if (lsp < SA::R0) {
#pragma unroll
            for (int s = 0; s < SA::R1; ++s)
                if (bin_used(s)) lds[lsp + SA::R0 * s] = v[s].x;
        }
        wave_sync();
#pragma unroll
        for (int t = 0; t < SB::R0; ++t) y[t].x = lds[sidx[t]];
        wave_sync();
        if (lsp < SA::R0) {
#pragma unroll
            for (int s = 0; s < SA::R1; ++s)
                if (bin_used(s)) lds[lsp + SA::R0 * s] = v[s].y;
        }
        wave_sync();
        __builtin_amdgcn_sched_barrier(0);
        float h[SB::R0];
        {
            const int lh = opaque(lane), jh = lh < L0B ? lh : L0B - 1;
#pragma unroll
            for (int t = 0; t < SB::R0; ++t) {
                const int lo = t * L0B, hi = lo + L0B - 1, n = jh + lo;
                // |frequency| q = n or NB - n; past QH (up-sampling: the input's band ends below the output's) the table is not
                // asked: gain 0.  The descriptor covers Hr[0 .. NB/2].
                if (hi <= NB / 2) h[t] = (NA >= NB || hi <= QH) ? buf_load_real<float>(rh, jh * 4, lo * 4) : lo > QH ? 0.f : (n <= QH ? buf_load_real<float>(rh, jh * 4, lo * 4) : 0.f);
                else if (lo > NB / 2) h[t] = (NA >= NB || NB - lo <= QH) ? buf_load_real<float>(rh, (L0B - jh) * 4, (NB - lo - L0B) * 4)
                                             : NB - hi > QH ? 0.f : (NB - n <= QH ? buf_load_real<float>(rh, (L0B - jh) * 4, (NB - lo - L0B) * 4) : 0.f);
                else { const int q = n > NB / 2 ? NB - n : n; h[t] = (NA >= NB || q <= QH) ? buf_load_real<float>(rh, q * 4, 0) : 0.f; }
            }
        }
#pragma unroll
        for (int t = 0; t < SB::R0; ++t) y[t].y = lds[sidx[t]];
        wave_sync();
#pragma unroll
        for (int t = 0; t < SB::R0; ++t) y[t] = C(y[t].x * h[t], y[t].y * h[t]);
        WSTAMP();

        // ---- inverse: pass 0 (radix R0 on L0B lanes), exchange, twiddle, pass 1 -> y[k + R0 s] on lanes k < R0 ------
        dft_r<SB::R0, +1>(y);
        WSTAMP();
        C z[SB::R1];
        { const int ln = opaque(lane); wave_exchange<SB::R0, SB::R1, SB::S>(lds, ln, ln, y, z); }
        WSTAMP();
        { const int ln = opaque(lane); wave_twiddle<SB::R1>(z, a.WB2, ln < SB::R0 ? ln : SB::R0 - 1); }
        dft_r<SB::R1, +1>(z);
        WSTAMP();

        // ---- the next item's inputs are requested here: raw's registers are free, the staging and the stores below cover the trip
        WaveItem nxt;
        bool has = false;
        uint32_t ni = item + W;
#if WAVE_PERSIST
        while (ni < vtotal && !(has = setup(ni, nxt))) ni += W;
        // (no next item: the same loads over an EMPTY range — zeros by the range check, no memory traffic.  A branch around the
        //  loads would merge "raw as it was" with "raw reloaded" behind it, and the register allocator then keeps the 120 dead
        //  values of the current item alive — and spilled — all the way from the first exchange)
        if (!has) { nxt.src = cur.src; nxt.nbytes = 0; nxt.zlo = 0; }
        issue_loads(nxt, raw, kT0, kTm); // first half: behind the last butterfly
#endif
        // ---- the two kept runs: block a = real parts, block b = imaginary parts, HOP outputs further on ----------------
        const int32_t valid_a = (int32_t)(cur.remain > Spec::HOP ? Spec::HOP : cur.remain);
        const int64_t remain_b = cur.remain - Spec::HOP;
        const int32_t valid_b = (int32_t)(remain_b < 0 ? 0 : remain_b > Spec::HOP ? Spec::HOP : remain_b);
        wave_store_run<Spec>(lds, opaque(lane), [&](int s) -> float { return z[s].x; }, cur.ya, valid_a);
        WSTAMP();
#if WAVE_PERSIST
        issue_loads(nxt, raw, kTm, kT1); // second half: behind run a's stores
#endif
        if (valid_b > 0) wave_store_run<Spec>(lds, opaque(lane), [&](int s) -> float { return z[s].y; }, cur.ya + Spec::HOP, valid_b);
        WSTAMP();
#ifdef FFT2_TRACE
        if (g_tr && lane == 0) { // where the wave ran: HW_ID (wave / simd / cu / sh / se fields) and the XCC id
            g_tr[13] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
            g_tr[14] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
            g_tr[15] = __builtin_amdgcn_s_memtime();
        }
#endif
        if (!has) break;
        cur = nxt;
        item = ni;
    }
}

// (L, M, periods per block, first kept output, kept outputs per block) -> kernel.  The geometry is the plan's: the host
// (fft.hip launch_fft) builds it with the forced block size and takes this kernel only when v0 / hop_out come out as here.
typedef WaveSpec<3840, 3528, 147, 3234, 160, 3520> Wave_160_147; // 48k -> 44.1k
typedef WaveSpec<3528, 3840, 160, 3520, 147, 3234> Wave_147_160; // 44.1k -> 48k
template __global__ void k_fft_wave<Wave_160_147>(FftArgs);
template __global__ void k_fft_wave<Wave_147_160>(FftArgs);

bool fft_wave_pick(int64_t L, int64_t M, FftWaveKernel *out)
{
    static const FftWaveKernel tab[] = {
        {147, 160, 24, 147, 3234, 22, 0x7fffffff, (const void *)k_fft_wave<Wave_160_147>}, // (114 us where k_fft_pair2 takes 111: not taken)
        {160, 147, 24, 160, 3520, 22, 8192, (const void *)k_fft_wave<Wave_147_160>},       // (117 against 125-127 us on the 128 x 10 s batch)
    };
    for (const FftWaveKernel &e : tab)
        if (e.L == L && e.M == M) { *out = e; return true; }
    return false;
}

const char *fft_wave_launch(const FftWaveKernel &k, const FftArgs &a_in, unsigned pairs, unsigned cols, void *stream)
{
    FftArgs a = a_in;
    a.pairs_per_col = pairs;
    const uint64_t total = (uint64_t)pairs * cols;
    if (total > 0x7fffffffull) return "job too long for one launch";
    // as many waves as the chip holds: 8 per CU (two per SIMD at 256 registers; 15.6 KB of LDS each)
    static int slots = 0;
    if (!slots) {
        int dev = 0, cus = 0, per_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return "device query";
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k.kern, 64, 0) != hipSuccess || per_cu < 1) per_cu = 8;
        slots = cus * per_cu;
    }
    const uint64_t vtotal = WAVE_XCD_MAP ? (total + 7) / 8 * 8 : total;
    unsigned waves = (unsigned)std::min<uint64_t>(vtotal, WAVE_PERSIST ? (uint64_t)(switches().dbg_wave_slots ? switches().dbg_wave_slots : slots) : vtotal);
    // (every wave the same number of items where that costs no more than a few idle slots: 8832 items on 2048 slots are 5 rounds
    //  for 640 waves and 4 for the rest; on 1767 waves every wave walks exactly 5)
    const uint64_t rounds = (vtotal + waves - 1) / waves;
    waves = (unsigned)((vtotal + rounds - 1) / rounds);
#ifdef FFT2_TRACE
    const size_t trace_n = (size_t)vtotal * 16;
    if (switches().dbg_trace) {
        if (hipMalloc((void **)&a.trace, trace_n * 8) != hipSuccess || hipMemset(a.trace, 0, trace_n * 8) != hipSuccess) return "trace buffer";
    }
#endif
    hipLaunchKernelGGL(reinterpret_cast<void (*)(FftArgs)>(const_cast<void *>(k.kern)), dim3(waves, 1, 1), dim3(64), 0, (hipStream_t)stream, a);
    const hipError_t e = hipGetLastError();
#ifdef FFT2_TRACE
    if (a.trace) { // debugging aid only: synchronous dump of the per-wave time stamps
        std::vector<unsigned long long> h(trace_n);
        (void)hipStreamSynchronize((hipStream_t)stream);
        (void)hipMemcpy(h.data(), a.trace, trace_n * 8, hipMemcpyDeviceToHost);
        if (FILE *f = fopen(switches().dbg_trace, "wb")) { fwrite(h.data(), 8, trace_n, f); fclose(f); }
        (void)hipFree(a.trace);
    }
#endif
    return e == hipSuccess ? nullptr : hipGetErrorString(e);
}

} // namespace hipsoxr
