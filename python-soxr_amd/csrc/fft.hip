// fft.hip — frequency-domain engine for the f32 path: rational overlap-save resampling.
//
// Same filter, different evaluation.  The direct-form kernels (kernels.hip) spend 2*T flops per
// output (592 at VHQ 48k->44.1k), which makes the path FMA-bound at <= 28 % of the HBM roofline.
// This engine evaluates the SAME prototype filter g (the plan's bank) in the frequency domain:
//
//   block of N_in = M*k input samples  --real FFT-->  X[0..N_in/2]
//   Y[q] = X[q] * H[q]  for q <= min(N_in, N_out)/2, else 0      (H = DTFT of g at the bin
//   frequencies; truncating/zero-extending the spectrum IS the rate change: bins of both grids
//   are f_in/N_in = f_out/N_out apart)
//   Y  --inverse real FFT of size N_out = L*k-->  L*k output samples
//
// with overlap-save: blocks start on period boundaries (input index multiple of M <-> output
// index multiple of L), overlap by more than the filter length, and only the outputs whose whole
// filter support lies inside the block are kept.  ~70-80 flop per output instead of 592.
// What is neglected is the aliasing of g's stop band (<= -176 dB for VHQ): measured against the
// direct form 2.5e-10 relative RMS in float64, 1.5e-7 to 2.2e-7 in float32 (FFT rounding) — inside the 1e-6
// bar, but NOT bit-identical to the canonical order, so this engine is used only where no
// bit-exact contract exists: whole-signal float32 and float64 device jobs (hipsoxr_run_device).  The host
// surface (soxr.resample / ResampleStream) and integer I/O stay on the exact engine.
//
//
// Kernels:
//   k_fft_block     general path, any 7-smooth plan: one workgroup per block, everything in LDS: load -> mixed-radix
//                   Stockham FFT (radices 16/8/4/2/3/5/7, twiddles from L2-resident tables) -> real-FFT untangling
//                   * H -> inverse real-FFT tangling -> Stockham inverse FFT -> store the valid outputs.
//   k_fft_pair      paired blocks (two real blocks as one complex signal), compile-time three- or four-pass
//                   schedules for the standard audio ratios; today the low-latency schedule of small jobs.
//   k_fft_pair2     second generation for unit-stride columns (mono, planar, batches): raw buffer loads with the
//                   hardware range check, output runs staged through LDS and stored as 16-byte granules;
//                   float32 and float64 instances.  AUTO's kernel for configs[1] and configs[3].
//   k_fft_strided2  the same for columns with a frame stride: interleaved data paired by channel (CP = true:
//                   one (Real, Real) word per frame; configs[2]) or strided columns paired by block (CP = false).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <vector>

#include "device.h"

#if defined(FFT2_ABL) && (FFT2_ABL & 1) // timing ablation (tools/abl_pair2.sh): no workgroup barriers — results are wrong
#define __syncthreads() ((void)0)
#endif

// Build-time switches of this file (A/B experiments; DESIGN.md §5.2 has the measurements):
//   FFT_BARRIER_LATE      the in-place barrier of a pass behind its butterflies (rounds 1-2) instead of behind its LDS reads
//   FFT_STORE_INTERLEAVE  radix-16 passes issue their LDS stores between the four final radix-4 butterflies (slower: +3 %)
//   FFT_EARLY_TABLES      twiddle / filter table loads one barrier ahead (no gain)
//   FFT_LDS_DMA           k_fft_pair2 (float32): the input blocks land in LDS by `buffer_load_dwordx4 ... lds`, first pass from LDS
//   FFT_DIF               k_fft_pair2 (float32, N_in >= N_out): wave-local schedule, 6 workgroup barriers per pair instead of 11 (slower: see dif_local)
//   FFT_EXPERIMENTS       the looping kernels k_fft_pair2p and k_fft_strided2<.., K > 0> (both slower than what they replace)
#ifndef FFT_BARRIER_LATE
#define FFT_BARRIER_EARLY 1
#endif

namespace hipsoxr {

#define HIP_TRY(expr)                                       \
    do {                                                    \
        hipError_t e_ = (expr);                             \
        if (e_ != hipSuccess) return hipGetErrorString(e_); \
    } while (0)

// ---------------------------------------------------------------------------------------------
// device: complex helpers and small DFTs (SIGN = -1 forward, +1 inverse, unnormalised)
// ---------------------------------------------------------------------------------------------
typedef float2 cf;
typedef double2 cd;
// The complex helpers and butterflies are templates over the complex type C (float2 or double2): the
// float64 instance of the paired kernel (float64 device jobs) shares every line of them.
template <typename C> using real_of = decltype(C().x);
template <typename C> __device__ __forceinline__ C cadd(C a, C b) { return C(a.x + b.x, a.y + b.y); }
template <typename C> __device__ __forceinline__ C csub(C a, C b) { return C(a.x - b.x, a.y - b.y); }
template <typename C> __device__ __forceinline__ C cmul(C a, C b) { return C(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
template <typename C> __device__ __forceinline__ C cconj(C a) { return C(a.x, -a.y); }
// multiply by SIGN * i
template <int SIGN, typename C> __device__ __forceinline__ C cmuli(C a)
{
    return SIGN > 0 ? C(-a.y, a.x) : C(a.y, -a.x);
}

template <int SIGN, typename C> __device__ __forceinline__ void dft2(C &a, C &b)
{
    C t = a; a = cadd(t, b); b = csub(t, b);
}
template <int SIGN, typename C> __device__ __forceinline__ void dft4(C &a0, C &a1, C &a2, C &a3)
{
    C s0 = cadd(a0, a2), d0 = csub(a0, a2), s1 = cadd(a1, a3), d1 = cmuli<SIGN>(csub(a1, a3));
    a0 = cadd(s0, s1); a2 = csub(s0, s1); a1 = cadd(d0, d1); a3 = csub(d0, d1);
}
template <int SIGN, typename C> __device__ __forceinline__ void dft8(C *u)
{
    typedef real_of<C> T;
    const T h = (T)0.70710678118654752440, sg = (T)SIGN;
    // two radix-4 on even/odd, then combine
    C e0 = u[0], e1 = u[2], e2 = u[4], e3 = u[6], o0 = u[1], o1 = u[3], o2 = u[5], o3 = u[7];
    dft4<SIGN>(e0, e1, e2, e3);
    dft4<SIGN>(o0, o1, o2, o3);
    // twiddles w8^m, m = 0..3 : 1, (1 + SIGN i)/sqrt2, SIGN i, (-1 + SIGN i)/sqrt2
    C t1 = C(h * (o1.x - sg * o1.y), h * (o1.y + sg * o1.x));
    C t2 = cmuli<SIGN>(o2);
    C t3 = C(h * (-o3.x - sg * o3.y), h * (-o3.y + sg * o3.x));
    u[0] = cadd(e0, o0); u[4] = csub(e0, o0);
    u[1] = cadd(e1, t1); u[5] = csub(e1, t1);
    u[2] = cadd(e2, t2); u[6] = csub(e2, t2);
    u[3] = cadd(e3, t3); u[7] = csub(e3, t3);
}
template <int SIGN, typename C> __device__ __forceinline__ void dft16(C *u)
{
    typedef real_of<C> T;
    // 4 x 4 decomposition: columns (stride 4), twiddle w16^(a*b), rows
    const T c1 = (T)0.92387953251128675613, s1 = (T)0.38268343236508977173, h = (T)0.70710678118654752440, sg = (T)SIGN;
    C x[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        C v0 = u[a], v1 = u[a + 4], v2 = u[a + 8], v3 = u[a + 12];
        dft4<SIGN>(v0, v1, v2, v3);
        x[a][0] = v0; x[a][1] = v1; x[a][2] = v2; x[a][3] = v3;
    }
    // twiddle x[a][b] *= w16^(a*b), w16 = exp(SIGN * 2 pi i / 16)
    const C w1 = C(c1, sg * s1), w2 = C(h, sg * h), w3 = C(s1, sg * c1);
    const C w4 = C((T)0, sg), w6 = C(-h, sg * h), w9 = C(-c1, -sg * s1);
    x[1][1] = cmul(x[1][1], w1); x[1][2] = cmul(x[1][2], w2); x[1][3] = cmul(x[1][3], w3);
    x[2][1] = cmul(x[2][1], w2); x[2][2] = cmul(x[2][2], w4); x[2][3] = cmul(x[2][3], w6);
    x[3][1] = cmul(x[3][1], w3); x[3][2] = cmul(x[3][2], w6); x[3][3] = cmul(x[3][3], w9);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        C v0 = x[0][b], v1 = x[1][b], v2 = x[2][b], v3 = x[3][b];
        dft4<SIGN>(v0, v1, v2, v3);
        u[b] = v0; u[b + 4] = v1; u[b + 8] = v2; u[b + 12] = v3;
    }
}
// dft16 whose outputs go to `sink(m, X[m])` as soon as each of the four final radix-4 butterflies has produced its four
// — the caller's LDS stores are then issued between the butterflies (a scheduling barrier pins them there) and the
// store path works beside the vector ALU instead of in a burst of sixteen behind it.
template <int SIGN, typename C, typename Sink> __device__ __forceinline__ void dft16_sink(C *u, Sink sink)
{
    typedef real_of<C> T;
    const T c1 = (T)0.92387953251128675613, s1 = (T)0.38268343236508977173, h = (T)0.70710678118654752440, sg = (T)SIGN;
    C x[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        C v0 = u[a], v1 = u[a + 4], v2 = u[a + 8], v3 = u[a + 12];
        dft4<SIGN>(v0, v1, v2, v3);
        x[a][0] = v0; x[a][1] = v1; x[a][2] = v2; x[a][3] = v3;
    }
    const C w1 = C(c1, sg * s1), w2 = C(h, sg * h), w3 = C(s1, sg * c1);
    const C w4 = C((T)0, sg), w6 = C(-h, sg * h), w9 = C(-c1, -sg * s1);
    x[1][1] = cmul(x[1][1], w1); x[1][2] = cmul(x[1][2], w2); x[1][3] = cmul(x[1][3], w3);
    x[2][1] = cmul(x[2][1], w2); x[2][2] = cmul(x[2][2], w4); x[2][3] = cmul(x[2][3], w6);
    x[3][1] = cmul(x[3][1], w3); x[3][2] = cmul(x[3][2], w6); x[3][3] = cmul(x[3][3], w9);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        C v0 = x[0][b], v1 = x[1][b], v2 = x[2][b], v3 = x[3][b];
        dft4<SIGN>(v0, v1, v2, v3);
        sink(b, v0); sink(b + 4, v1); sink(b + 8, v2); sink(b + 12, v3);
        __builtin_amdgcn_sched_barrier(0);
    }
}
// odd prime radix via the conjugate-pair form: X[m], X[R-m] = A_m +- SIGN*i*B_m
template <int R, int SIGN, typename C> __device__ __forceinline__ void dft_odd(C *u)
{
    typedef real_of<C> T;
    constexpr int Hh = (R - 1) / 2;
    constexpr double PI2 = 6.283185307179586476925286766559;
    C s[Hh], d[Hh];
#pragma unroll
    for (int t = 0; t < Hh; ++t) { s[t] = cadd(u[t + 1], u[R - 1 - t]); d[t] = csub(u[t + 1], u[R - 1 - t]); }
    C x0 = u[0];
    C sum = x0;
#pragma unroll
    for (int t = 0; t < Hh; ++t) sum = cadd(sum, s[t]);
    C out[R];
    out[0] = sum;
#pragma unroll
    for (int m = 1; m <= Hh; ++m) {
        C A = x0, B = C((T)0, (T)0);
#pragma unroll
        for (int t = 1; t <= Hh; ++t) {
            const T c = (T)__builtin_cos(PI2 * (double)((m * t) % R) / R);
            const T sn = (T)__builtin_sin(PI2 * (double)((m * t) % R) / R);
            A.x += c * s[t - 1].x; A.y += c * s[t - 1].y;
            B.x += sn * d[t - 1].x; B.y += sn * d[t - 1].y;
        }
        // SIGN*i*B
        C iB = cmuli<SIGN>(B);
        out[m] = cadd(A, iB);
        out[R - m] = csub(A, iB);
    }
#pragma unroll
    for (int m = 0; m < R; ++m) u[m] = out[m];
}
template <int R, int SIGN, typename C> __device__ __forceinline__ void dft_r(C *u);

// Composite radix R1*R2 with coprime factors by the prime-factor (Good-Thomas) index maps: a
// plain R1 x R2 two-dimensional DFT, no internal twiddles; the maps are compile-time constants, so
// they cost register renaming only.
constexpr int inv_mod(int a, int m)
{
    for (int x = 1; x < m; ++x)
        if ((a * x) % m == 1) return x;
    return 1;
}
template <int R1, int R2, int SIGN, typename C> __device__ __forceinline__ void dft_pfa(C *u)
{
    constexpr int N = R1 * R2, e1 = R2 * inv_mod(R2 % R1, R1), e2 = R1 * inv_mod(R1 % R2, R2);
    C x[R2][R1]; // x[n2][n1] = u[(R2 n1 + R1 n2) mod N]
#pragma unroll
    for (int n2 = 0; n2 < R2; ++n2)
#pragma unroll
        for (int n1 = 0; n1 < R1; ++n1) x[n2][n1] = u[(R2 * n1 + R1 * n2) % N];
#pragma unroll
    for (int n2 = 0; n2 < R2; ++n2) dft_r<R1, SIGN>(x[n2]);
#pragma unroll
    for (int k1 = 0; k1 < R1; ++k1) {
        C c[R2];
#pragma unroll
        for (int n2 = 0; n2 < R2; ++n2) c[n2] = x[n2][k1];
        dft_r<R2, SIGN>(c);
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) u[(e1 * k1 + e2 * k2) % N] = c[k2]; // CRT output map
    }
}
template <int R, int SIGN, typename C> __device__ __forceinline__ void dft_r(C *u)
{
    if constexpr (R == 2) dft2<SIGN>(u[0], u[1]);
    else if constexpr (R == 4) dft4<SIGN>(u[0], u[1], u[2], u[3]);
    else if constexpr (R == 8) dft8<SIGN>(u);
    else if constexpr (R == 16) dft16<SIGN>(u);
    else if constexpr (R == 6) dft_pfa<2, 3, SIGN>(u);
    else if constexpr (R == 10) dft_pfa<2, 5, SIGN>(u);
    else if constexpr (R == 12) dft_pfa<4, 3, SIGN>(u);
    else if constexpr (R == 14) dft_pfa<2, 7, SIGN>(u);
    else if constexpr (R == 15) dft_pfa<3, 5, SIGN>(u);
    else if constexpr (R == 20) dft_pfa<4, 5, SIGN>(u);
    else if constexpr (R == 21) dft_pfa<3, 7, SIGN>(u);
    else dft_odd<R, SIGN>(u);
}

// One Stockham pass of a length-N transform, IN PLACE in a single LDS buffer: every thread reads
// the inputs of its butterflies into registers, the workgroup synchronises, then results are
// written to their autosort positions (one buffer instead of two: 20 KB per workgroup, so 7
// workgroups fit a CU and hide each other's barriers and LDS latency).
// Radix R, Ns = product of earlier radices, NB = max butterflies per thread.
// W = table exp(SIGN*2*pi*i*m/N), m = 0..N-1 (global memory, L1/L2 resident); only the t = 1
// twiddle of a butterfly is loaded, its powers are formed in registers.
template <int R, int SIGN, int NB>
__device__ __forceinline__ void fft_pass(cf *buf, int N, int Ns, const cf *W)
{
    const int nb = N / R, wstep = N / (Ns * R);
    cf u[NB][R];
    int dst[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int j = threadIdx.x + i * blockDim.x;
        dst[i] = -1;
        if (j < nb) {
            const int grp = j / Ns, k = j - grp * Ns;
            dst[i] = grp * Ns * R + k;
#pragma unroll
            for (int t = 0; t < R; ++t) u[i][t] = buf[j + t * nb];
            if (Ns > 1) {
                const cf w1 = W[k * wstep];
                cf w = w1;
#pragma unroll
                for (int t = 1; t < R; ++t) {
                    u[i][t] = cmul(u[i][t], w);
                    if (t + 1 < R) w = cmul(w, w1);
                }
            }
            dft_r<R, SIGN>(u[i]);
        }
    }
    __syncthreads(); // all inputs of this pass are in registers
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        if (dst[i] >= 0) {
            cf *o = buf + dst[i];
#pragma unroll
            for (int t = 0; t < R; ++t) o[t * Ns] = u[i][t];
        }
    }
    __syncthreads();
}

// Compile-time specialised pass (N, Ns, R constants): no divisions, fully unrolled.
// `load(n)` supplies element n of the pass input (LDS, or global memory for the first pass) and
// `store(n, v)` consumes element n of the pass output (LDS, or the output signal for the last
// pass), so the first/last passes stream straight from/to HBM without an extra LDS round trip.
// PRE: the butterfly's twiddle was fetched earlier (pre_w1, one butterfly per thread) — for small
// jobs, whose cost is the latency of a single workgroup, every table read that follows a barrier
// is an exposed L2 round trip; fetched at kernel start they all overlap with the input loads.
// EARLY (-DFFT_EARLY_TABLES; measured in round 3 and left off): both table entries of the butterfly's twiddle (w and
// w^4), and the first inverse pass's filter values, fetched one barrier ahead — after the previous pass's LDS stores
// were issued, before the barrier in front of this pass — so that their L2 round trip would run behind the store drain
// and the barrier.  No register is held across a butterfly for it, and it does not pay: the loads' issue delays the
// wave's arrival at the barrier by as much as their latency was hidden behind the LDS reads before (batch 137 vs
// 138 us, 60 s clip 12.2 vs 11.7 us on the same box).
// The thread's index through an opaque move.  In the resident-workgroup kernel (k_fft_pair2p) the item is the body of a
// loop, and everything that depends only on threadIdx.x and the kernel arguments — per-thread table addresses, LDS
// indices and offsets of six passes — is loop-invariant: the compiler hoists it all out and keeps it alive across the
// loop (216 VGPRs and 35 spilled SGPRs where the grid-per-item kernel needs 68; a real function call instead costs
// the calling convention's alternating caller/callee-saved register blocks: highest VGPR 102).  A volatile asm is
// not loop-invariant, so index arithmetic that starts from this value stays where it is written.  (The other half is
// -mllvm -disable-machine-licm: the machine-level pass hoists the materialisation of every literal — butterfly
// constants, scalar offsets — into registers of its own: 112 VGPRs / 17 spilled SGPRs with it, 70 / 0 without.)
// Both only in EXPERIMENT builds — HIPSOXR_EXTRA_FLAGS="-DFFT_EXPERIMENTS -mllvm -disable-machine-licm" build.sh — which
// are also the only builds that contain the two looping kernels (k_fft_pair2p, k_fft_strided2<.., K > 0>): both were
// measured slower than what they replace, and the opaque index costs the plain kernels 5-7 % more VALU instructions.
__device__ __forceinline__ int fft_tid()
{
    int t = (int)threadIdx.x;
#ifdef FFT_EXPERIMENTS // (costs 5-7 % more VALU instructions in the plain kernels: common subexpressions are no longer shared)
    asm volatile("" : "+v"(t));
#endif
    return t;
}
template <typename C> struct TwPre { C w1, w4; };
template <int N, int Ns, int R, int NT, typename C> __device__ __forceinline__ TwPre<C> tw_fetch(const C *W)
{
    constexpr int nb = N / R, wstep = N / (Ns * R);
    static_assert((nb + NT - 1) / NT == 1, "early twiddles: one butterfly per thread");
    const int tid = fft_tid(), j = tid < nb ? tid : 0, k = j % Ns;
    TwPre<C> t;
    t.w1 = W[k * wstep];
    t.w4 = R >= 10 ? W[4 * k * wstep] : t.w1;
    return t;
}
template <int N, int Ns, int R, int SIGN, int NT, bool SYNC_BEFORE_STORE, bool PRE = false, bool EARLY = false, typename C, typename Load, typename Store>
__device__ __forceinline__ void fft_pass_ct(const C *W, Load load, Store store, C pre_w1 = C(), TwPre<C> early = TwPre<C>())
{
    constexpr int nb = N / R, wstep = N / (Ns * R), NB = (nb + NT - 1) / NT;
    static_assert(!PRE || (NB == 1 && R < 10), "prefetched twiddles: one butterfly per thread, radix < 10");
    static_assert(!EARLY || NB == 1, "early twiddles: one butterfly per thread");
    typedef real_of<C> T;
    const int tid = fft_tid();
    C u[NB][R];
#ifdef FFT_BARRIER_EARLY
    // The in-place barrier straight behind the pass's LDS reads instead of behind its butterflies (round 3): what the
    // barrier has to guarantee is that every thread HOLDS its inputs, not that it has finished computing; waves reach
    // it after one LDS round trip instead of after their butterflies, and the skew of the butterflies is absorbed by
    // the barrier in front of the next pass alone.  configs[2] 47.4 -> 45.5 us, batch and clip unchanged, 3 VGPRs fewer.
    C w1s[NB], w4s[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int j = tid + i * NT;
        w1s[i] = C((T)1, (T)0); w4s[i] = w1s[i];
        if (NB * NT == nb || j < nb) {
            const int k = j % Ns;
            if (Ns > 1) w1s[i] = EARLY ? early.w1 : PRE ? pre_w1 : W[k * wstep];
            if (Ns > 1 && R >= 10) w4s[i] = EARLY ? early.w4 : W[4 * k * wstep];
#pragma unroll
            for (int t = 0; t < R; ++t) {
                if constexpr (std::is_invocable_v<Load, int, int>) u[i][t] = load(j + t * nb, t);
                else u[i][t] = load(j + t * nb);
            }
        }
    }
    if (SYNC_BEFORE_STORE) __syncthreads(); // in-place: every input of the pass is in registers
#endif
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int j = tid + i * NT;
        if (NB * NT == nb || j < nb) {
            const int k = j % Ns;
#ifdef FFT_BARRIER_EARLY
            const C w1 = w1s[i];
            (void)k;
#else
            C w1 = C((T)1, (T)0);
            if (Ns > 1) w1 = EARLY ? early.w1 : PRE ? pre_w1 : W[k * wstep]; // issued before the data reads: the latencies overlap
#pragma unroll
            for (int t = 0; t < R; ++t) {
                if constexpr (std::is_invocable_v<Load, int, int>) u[i][t] = load(j + t * nb, t); // t: input slot
                else u[i][t] = load(j + t * nb);
            }
#endif
#if defined(FFT2_ABL) && (FFT2_ABL & 32)
            if (false) {
#else
            if (Ns > 1) {
#endif
                // Powers w^t of the butterfly's twiddle.  Any power formed from ONE rounded table
                // entry inherits t times its phase error ((w(1+e))^t ~ w^t (1+te)), so for the large
                // radices a second entry, w^4, is read and w^(4a+b) = (w^4)^a w^b: the error factor
                // drops from R-1 to <= a+b, for the same number of complex products.
                C pw[R];
                pw[1] = w1;
                if constexpr (R >= 10) {
#ifdef FFT_BARRIER_EARLY
                    const C w4 = w4s[i];
#else
                    const C w4 = EARLY ? early.w4 : W[4 * k * wstep]; // 4*k*wstep < 4N/R <= N
#endif
#pragma unroll
                    for (int t = 2; t < R; ++t) {
                        const int a4 = t / 4, b4 = t % 4;
                        if (a4 == 0) pw[t] = cmul(pw[t - 1], w1);
                        else if (b4 == 0) pw[t] = a4 == 1 ? w4 : (a4 % 2 == 0 ? cmul(pw[t / 2], pw[t / 2]) : cmul(pw[t - 4], w4));
                        else pw[t] = cmul(pw[4 * a4], pw[b4]);
                    }
                } else {
#pragma unroll
                    for (int t = 2; t < R; ++t) pw[t] = (t & 1) ? cmul(pw[t - 1], w1) : cmul(pw[t / 2], pw[t / 2]);
                }
#pragma unroll
                for (int t = 1; t < R; ++t) u[i][t] = cmul(u[i][t], pw[t]);
            }
#if defined(FFT_BARRIER_EARLY) && defined(FFT_STORE_INTERLEAVE)
            if constexpr (R == 16) {
                const int o16 = (j - k) * R + k;
                dft16_sink<SIGN>(u[i], [&](int t, C v) { store(o16 + t * Ns, v); });
                continue; // (stored)
            }
#endif
#if !(defined(FFT2_ABL) && (FFT2_ABL & 16))
            dft_r<R, SIGN>(u[i]);
#endif
        }
    }
#ifndef FFT_BARRIER_EARLY
    if (SYNC_BEFORE_STORE) __syncthreads(); // in-place: every input of the pass is in registers
#endif
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int j = tid + i * NT;
        if (NB * NT == nb || j < nb) {
#if defined(FFT_BARRIER_EARLY) && defined(FFT_STORE_INTERLEAVE)
            if constexpr (R == 16) continue; // (stored from inside the butterfly)
#endif
            const int k = j % Ns, o = (j - k) * R + k;
#pragma unroll
            for (int t = 0; t < R; ++t) store(o + t * Ns, u[i][t]);
        }
    }
}
// Four-pass transform: first pass input from `first_load`, last pass output to `last_store`,
// everything in between in place in `buf`.
template <int N, int SIGN, int NT, int R0, int R1, int R2, int R3, typename Load, typename Store>
__device__ __forceinline__ void fft_ct(cf *buf, const cf *W, Load first_load, Store last_store, bool first_in_lds)
{
    auto lds_load = [&](int n) -> cf { return buf[n]; };
    auto lds_store = [&](int n, cf v) { buf[n] = v; };
    // pass 0 (Ns = 1): when its input is not in LDS nothing has to be protected before storing
    if (first_in_lds) fft_pass_ct<N, 1, R0, SIGN, NT, true>(W, first_load, lds_store);
    else fft_pass_ct<N, 1, R0, SIGN, NT, false>(W, first_load, lds_store);
    __syncthreads();
    fft_pass_ct<N, R0, R1, SIGN, NT, true>(W, lds_load, lds_store);
    __syncthreads();
    fft_pass_ct<N, R0 * R1, R2, SIGN, NT, true>(W, lds_load, lds_store);
    __syncthreads();
    fft_pass_ct<N, R0 * R1 * R2, R3, SIGN, NT, false>(W, lds_load, last_store);
}

// Twiddle of pass (Ns, R) for this thread's (single) butterfly, to be fetched ahead of time.
template <int N, int Ns, int R, int NT> __device__ __forceinline__ cf pass_twiddle(const cf *W)
{
    constexpr int nb = N / R, wstep = N / (Ns * R);
    const int j = threadIdx.x < nb ? threadIdx.x : 0;
    return W[(j % Ns) * wstep];
}
// Four-pass transform whose twiddles (passes 2-4) were fetched by the caller.
template <int N, int SIGN, int NT, int R0, int R1, int R2, int R3, typename Load, typename Store>
__device__ __forceinline__ void fft_ct_pre(cf *buf, const cf *W, Load first_load, Store last_store, bool first_in_lds,
                                           cf w1, cf w2, cf w3)
{
    auto lds_load = [&](int n) -> cf { return buf[n]; };
    auto lds_store = [&](int n, cf v) { buf[n] = v; };
    if (first_in_lds) fft_pass_ct<N, 1, R0, SIGN, NT, true>(W, first_load, lds_store);
    else fft_pass_ct<N, 1, R0, SIGN, NT, false>(W, first_load, lds_store);
    __syncthreads();
    fft_pass_ct<N, R0, R1, SIGN, NT, true, true>(W, lds_load, lds_store, w1);
    __syncthreads();
    fft_pass_ct<N, R0 * R1, R2, SIGN, NT, true, true>(W, lds_load, lds_store, w2);
    __syncthreads();
    fft_pass_ct<N, R0 * R1 * R2, R3, SIGN, NT, false, true>(W, lds_load, last_store, w3);
}

// Three-pass variant (larger radices: fewer LDS round trips and barriers).
// LDS bank conflicts: pass loads are contiguous across lanes (conflict-free); pass stores run in
// groups of Ns consecutive elements, so only the FIRST pass (Ns = 1: lane stride = R0 elements) can
// conflict.  An odd-ish R0 (5, 21: stride 40 / 168 bytes) is conflict-free as it is; for R0 = 16
// (stride 128 bytes = every lane on the same two banks, a 16-way conflict) the buffer between pass
// 1 and pass 2 is kept in a swizzled layout  n -> n ^ ((n >> 4) & 15)  (SWZ).
#ifdef FFT2_TRACE
#define FFT_STAMP() do { if (g_tr && (threadIdx.x & 63) == 0 && g_tri < 16) g_tr[g_tri] = __builtin_amdgcn_s_memtime(); ++g_tri; } while (0)
#define FFT_STAMP_DECL unsigned long long *g_tr, int &g_tri,
#define FFT_STAMP_ARGS g_tr, g_tri,
#else
#define FFT_STAMP() ((void)0)
#define FFT_STAMP_DECL
#define FFT_STAMP_ARGS
#endif
template <int N, int SIGN, int NT, int R0, int R1, int R2, bool SWZ, bool LASTSYNC = false, typename C, typename Load, typename Store>
__device__ __forceinline__ void fft_ct3(FFT_STAMP_DECL C *buf, const C *W, Load first_load, Store last_store, bool first_in_lds)
{
    static_assert(R0 * R1 * R2 == N, "radix schedule");
    auto lds_load = [&](int n) -> C { return buf[n]; };
    auto lds_store = [&](int n, C v) { buf[n] = v; };
    auto swz_load = [&](int n) -> C { return buf[SWZ ? n ^ ((n >> 4) & 15) : n]; };
    auto swz_store = [&](int n, C v) { buf[SWZ ? n ^ ((n >> 4) & 15) : n] = v; };
    if (first_in_lds) fft_pass_ct<N, 1, R0, SIGN, NT, true>(W, first_load, swz_store);
    else fft_pass_ct<N, 1, R0, SIGN, NT, false>(W, first_load, swz_store);
    FFT_STAMP();
#ifdef FFT_EARLY_TABLES
    constexpr bool EARLY = true;
    const TwPre<C> t1 = tw_fetch<N, R0, R1, NT>(W); // (behind this pass's stores, in front of the barrier)
#else
    constexpr bool EARLY = false;
    const TwPre<C> t1 = TwPre<C>();
#endif
    __syncthreads();
    FFT_STAMP();
    fft_pass_ct<N, R0, R1, SIGN, NT, true, false, EARLY>(W, swz_load, lds_store, C(), t1);
    FFT_STAMP();
#ifdef FFT_EARLY_TABLES
    const TwPre<C> t2 = tw_fetch<N, R0 * R1, R2, NT>(W);
#else
    const TwPre<C> t2 = TwPre<C>();
#endif
    __syncthreads();
    FFT_STAMP();
    fft_pass_ct<N, R0 * R1, R2, SIGN, NT, LASTSYNC, false, EARLY>(W, lds_load, last_store, C(), t2);
    FFT_STAMP();
}

struct FftArgs {
    const void *in;
    void *out;
    const float2 *WA, *WB, *P, *Q, *Hs; // twiddles of both transforms, (un)tangling twiddles, filter
    const float2 *WA2, *WB2;            // paired-block kernel: twiddles of the full-length transforms
    const float *Hr;                    // k_fft_pair2: the filter as REAL values (see fft_build)
    const double2 *WA2d, *WB2d;         // k_fft_pair2<double>: the same tables in float64
    const double *Hrd;
    unsigned long long *trace;          // HIPSOXR_DEBUG_TRACE (builds with -DFFT2_TRACE only): per-wave s_memtime stamps [wg][wave][16]
    int32_t A, B;            // complex transform lengths: N_in/2, N_out/2
    int32_t nA, nB;          // number of passes
    int32_t radA[8], radB[8];
    int64_t L, M;
    int32_t lead_periods, hop_periods; // block b covers periods [b*hop - lead, ...): k periods long
    int32_t v0, hop_out;     // first kept local output, outputs kept per block
    uint32_t n_clips, n_channels;
    int64_t ics, ifs, ichs, ocs, ofs, ochs;
    int64_t in_frames, out_frames;
    uint32_t *queue;         // k_fft_pair2p: {items handed out beyond the grid's own, workgroups that have left}, zero between launches
    uint32_t n_items;        // k_fft_pair2p: columns x pairs_per_col
    int32_t stagger;         // k_fft_pair2p: HIPSOXR_DEBUG_STAGGER
    int32_t walk;            // k_fft_strided2<.., K > 0>: consecutive blocks per workgroup
    int64_t n_blocks_col;    // ... blocks per column
    const void *HP;          // k_fft_pair2 -DFFT_DIF: per output-grid bin n: [N_out] H (real), then [N_out] LDS byte offsets of the input-grid bin it takes
    const int64_t *clip_tab; // ragged batch (hipsoxr_job_t::clip_table_dev): [n_clips][4] = in offset, in frames, out offset, out frames; k_fft_pair2 only
    int32_t chpair; // paired kernel: 1 = pair neighbouring channels of interleaved data instead of blocks
    int64_t pairs_per_col; // xcd_map: work items (blocks, or pairs of blocks) per channel unit
    int32_t xcd_map;       // interleaved multi-channel data: XCD-aware workgroup ids (see k_fft_pair)
};

// butterflies per thread are bounded by N/(R*256) rounded up; lengths up to 4096
template <int SIGN>
__device__ __forceinline__ void run_passes(cf *buf, int N, int n_pass, const int32_t *rad, const cf *W)
{
    int Ns = 1;
    for (int p = 0; p < n_pass; ++p) {
        const int R = rad[p];
        const int per = (N / R + (int)blockDim.x - 1) / (int)blockDim.x; // wave-uniform
        // butterflies per thread (256 threads, N <= 4096): R=16: 1, R=8: <=2, R=7: <=3, R=5,4: <=4, R=3: <=6, R=2: <=8
        switch (R) {
        case 16: fft_pass<16, SIGN, 1>(buf, N, Ns, W); break;
        case 8: if (per <= 1) fft_pass<8, SIGN, 1>(buf, N, Ns, W); else fft_pass<8, SIGN, 2>(buf, N, Ns, W); break;
        case 7: if (per <= 2) fft_pass<7, SIGN, 2>(buf, N, Ns, W); else fft_pass<7, SIGN, 3>(buf, N, Ns, W); break;
        case 5: if (per <= 2) fft_pass<5, SIGN, 2>(buf, N, Ns, W); else fft_pass<5, SIGN, 4>(buf, N, Ns, W); break;
        case 4: if (per <= 2) fft_pass<4, SIGN, 2>(buf, N, Ns, W); else fft_pass<4, SIGN, 4>(buf, N, Ns, W); break;
        case 3: if (per <= 4) fft_pass<3, SIGN, 4>(buf, N, Ns, W); else fft_pass<3, SIGN, 6>(buf, N, Ns, W); break;
        default: fft_pass<2, SIGN, 8>(buf, N, Ns, W); break;
        }
        Ns *= R;
    }
}

struct SpecRuntime { static constexpr bool ct = false; static constexpr int NT = 256, A = 0, B = 4096; };
// 48k -> 44.1k family (L = 147, M = 160, k = 32): N_in/2 = 2560 = 5*8*8*8, N_out/2 = 2352 = 3*7*7*16
struct Spec2560x2352 {
    static constexpr bool ct = true;
    static constexpr int A = 2560, B = 2352;
    static constexpr int NT = 256;
    template <typename Ld, typename St> static __device__ __forceinline__ void fwd(cf *b, const cf *W, Ld ld, St st, bool in_lds)
    { fft_ct<2560, -1, NT, 5, 8, 8, 8>(b, W, ld, st, in_lds); }
    template <typename Ld, typename St> static __device__ __forceinline__ void inv(cf *b, const cf *W, Ld ld, St st, bool in_lds)
    { fft_ct<2352, +1, NT, 3, 7, 7, 16>(b, W, ld, st, in_lds); }
};

// same family, half-size blocks (k = 16) for small jobs: N_in/2 = 1280 = 5*8*8*4, N_out/2 = 1176 = 3*7*7*8
struct Spec1280x1176 {
    static constexpr bool ct = true;
    static constexpr int A = 1280, B = 1176;
    static constexpr int NT = 256;
    template <typename Ld, typename St> static __device__ __forceinline__ void fwd(cf *b, const cf *W, Ld ld, St st, bool in_lds)
    { fft_ct<1280, -1, NT, 5, 8, 8, 4>(b, W, ld, st, in_lds); }
    template <typename Ld, typename St> static __device__ __forceinline__ void inv(cf *b, const cf *W, Ld ld, St st, bool in_lds)
    { fft_ct<1176, +1, NT, 3, 7, 7, 8>(b, W, ld, st, in_lds); }
};

template <typename Spec>
__global__ void __launch_bounds__(256, Spec::ct ? 5 : 2) k_fft_block(FftArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int32_t A = a.A, B = a.B;
    cf *cur = reinterpret_cast<cf *>(smem_raw); // single buffer of max(A, B) + 1 complex values

    const uint32_t col = blockIdx.y;
    const uint32_t ch = col % a.n_channels, clip = col / a.n_channels;
    const int64_t blk = blockIdx.x;
    const int64_t p0 = blk * a.hop_periods - a.lead_periods; // first period of the block (may be < 0)
    const int64_t in0 = p0 * a.M, out0 = p0 * a.L;          // absolute indices of local sample 0
    const float *xin = (const float *)a.in + (int64_t)clip * a.ics + (int64_t)ch * a.ichs;

    // ---- load: z[n] = x[2n] + i x[2n+1], zero outside the signal; forward complex FFT of length A.
    //      Interior blocks of the specialised kernel stream the first pass straight from HBM.
    const bool fast = a.ifs == 1 && in0 >= 0 && in0 + 2 * (int64_t)A <= a.in_frames &&
                      (((reinterpret_cast<uintptr_t>(xin) >> 2) + (uint64_t)in0) & 1) == 0;
    auto lds_load = [&](int n) -> cf { return cur[n]; };
    auto lds_store = [&](int n, cf v) { cur[n] = v; };
    if (Spec::ct && fast) {
        if constexpr (Spec::ct) {
            const float2 *src = reinterpret_cast<const float2 *>(xin + in0);
            Spec::fwd(cur, a.WA, [&](int n) -> cf { return src[n]; }, lds_store, false);
        }
    } else {
        for (int n = threadIdx.x; n < A; n += blockDim.x) {
            const int64_t l = in0 + 2 * (int64_t)n;
            float re = (l >= 0 && l < a.in_frames) ? xin[l * a.ifs] : 0.f;
            float im = (l + 1 >= 0 && l + 1 < a.in_frames) ? xin[(l + 1) * a.ifs] : 0.f;
            cur[n] = make_float2(re, im);
        }
        __syncthreads();
        if constexpr (Spec::ct) Spec::fwd(cur, a.WA, lds_load, lds_store, true);
        else run_passes<-1>(cur, A, a.nA, a.radA, a.WA);
    }
    __syncthreads();

    // ---- untangle the real FFT, apply the filter, tangle for the inverse real FFT — in registers:
    //      X[q] = (Z[q] + conj Z[A-q])/2 - i/2 P[q] (Z[q] - conj Z[A-q]),   P[q] = exp(-2 pi i q / N_in)
    //      Y[q] = X[q] Hs[q]  (q <= min(A, B), else 0)
    //      W[q] = (Y[q] + conj Y[B-q]) + i Q[q] (Y[q] - conj Y[B-q]),       Q[q] = exp(+2 pi i q / N_out)
    // thread handles the pair (q, B-q): it needs Z[q], Z[A-q], Z[B-q], Z[A-B+q].
    {
        const int qmax = A < B ? A : B;
        auto spectrum = [&](int q) -> cf { // Y[q]
            if (q > qmax) return make_float2(0.f, 0.f);
            const cf zq = cur[q == A ? 0 : q], zc = cconj(cur[q == 0 ? 0 : A - q]);
            const cf s = cadd(zq, zc), d = cmul(a.P[q], csub(zq, zc));
            const cf x = make_float2(0.5f * (s.x + d.y), 0.5f * (s.y - d.x));
            return cmul(x, a.Hs[q]);
        };
        constexpr int NP = (Spec::B / 2 + 1 + Spec::NT - 1) / Spec::NT; // pairs per thread: B/2 + 1 <= 256 * NP (spec: 1177; generic: B <= 4096)
        cf wq[NP], wr[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int q = threadIdx.x + i * blockDim.x;
            if (q <= B / 2) {
                const cf yq = spectrum(q), yr = spectrum(B - q);
                // W[q] from (Y[q], Y[B-q]);  W[B-q] from (Y[B-q], Y[q])
                cf s = cadd(yq, cconj(yr)), d = cmul(a.Q[q], csub(yq, cconj(yr)));
                wq[i] = make_float2(s.x - d.y, s.y + d.x);
                if (q != 0 && q != B - q) {
                    s = cadd(yr, cconj(yq)); d = cmul(a.Q[B - q], csub(yr, cconj(yq)));
                    wr[i] = make_float2(s.x - d.y, s.y + d.x);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int q = threadIdx.x + i * blockDim.x;
            if (q <= B / 2) {
                cur[q] = wq[i];
                if (q != 0 && q != B - q) cur[B - q] = wr[i];
            }
        }
    }
    __syncthreads();

    // ---- inverse complex FFT of length B (unnormalised; the scale lives in Hs) and store of the
    //      kept outputs: element n of the result holds local outputs 2n (re) and 2n+1 (im); the
    //      specialised kernel writes them from the last pass's registers.
    float *yo = (float *)a.out + (int64_t)clip * a.ocs + (int64_t)ch * a.ochs;
    const int32_t v0 = a.v0, v1 = a.v0 + a.hop_out;
    auto out_store = [&](int n, cf w) {
        const int32_t i0 = 2 * n;
        const int64_t k0 = out0 + i0;
        if (i0 >= v0 && i0 < v1 && k0 >= 0 && k0 < a.out_frames) yo[k0 * a.ofs] = w.x;
        if (i0 + 1 >= v0 && i0 + 1 < v1 && k0 + 1 >= 0 && k0 + 1 < a.out_frames) yo[(k0 + 1) * a.ofs] = w.y;
    };
    if constexpr (Spec::ct) {
        Spec::inv(cur, a.WB, lds_load, out_store, true);
    } else {
        run_passes<+1>(cur, B, a.nB, a.radB, a.WB);
        for (int n = threadIdx.x; n < B; n += blockDim.x) out_store(n, cur[n]);
    }
}

// ---------------------------------------------------------------------------------------------
// Paired-block kernel.  The whole chain  FFT -> multiply by H -> truncate -> inverse FFT  maps real
// signals to real signals and is linear, so it can process TWO real blocks at once as the real and
// imaginary part of one complex signal:  z = x_a + i x_b  ->  y_a + i y_b  (H is Hermitian, the
// truncation symmetric; the Nyquist bin of the output grid sums both aliases so that the operator
// stays exactly real).  That removes the real-FFT untangle/tangle stages altogether — the filter
// multiply rides on the loads of the first inverse pass — and, with radix-16/20/21 butterflies,
// leaves 3 + 3 LDS passes per pair of blocks instead of 4 + 1 + 4 per block.
// One workgroup = blocks (2b, 2b+1) of one column; one LDS buffer of N_in complex values.
// ---------------------------------------------------------------------------------------------
// Schedules: N_in = A0*A1*A2 (forward), N_out = B0*B1*B2 (inverse); *SWZ = swizzled layout after a
// power-of-two first radix (see fft_ct3).  NT >= the largest butterfly count of any pass.
template <int NA_, int NB_, int NT_, int A0, int A1, int A2, bool ASWZ, int B0, int B1, int B2, bool BSWZ>
struct PairSpec {
    static constexpr int NA = NA_, NB = NB_, NT = NT_;
    static constexpr bool prefetch = false;
    static constexpr int RB0 = B0, RA0 = A0, RA1 = A1, RA2 = A2, RB1 = B1, RB2 = B2;
    // the wave-local schedule (k_fft_pair2 under -DFFT_DIF, see dif_local) needs every sub-transform inside one wave
    static constexpr bool dif = (64 / (A1 > A2 ? A1 : A2)) * (NT / 64) >= A0 && (64 / (B1 > B2 ? B1 : B2)) * (NT / 64) >= B0 &&
                                A1 * A2 <= NT && B1 * B2 <= NT;
    struct Tw {};
    template <typename C, typename Ld, typename St> static __device__ __forceinline__ void fwd(FFT_STAMP_DECL C *b, const C *W, Ld ld, St st, bool in_lds, const Tw &)
    { fft_ct3<NA, -1, NT, A0, A1, A2, ASWZ>(FFT_STAMP_ARGS b, W, ld, st, in_lds); }
    template <typename C, typename Ld, typename St> static __device__ __forceinline__ void inv(FFT_STAMP_DECL C *b, const C *W, Ld ld, St st, bool in_lds, const Tw &)
    { fft_ct3<NB, +1, NT, B0, B1, B2, BSWZ>(FFT_STAMP_ARGS b, W, ld, st, in_lds); }
    // last pass stores into LDS in another layout (output staging): all its inputs must be in registers first
    template <typename C, typename Ld, typename St> static __device__ __forceinline__ void inv_staged(FFT_STAMP_DECL C *b, const C *W, Ld ld, St st)
    { fft_ct3<NB, +1, NT, B0, B1, B2, BSWZ, true>(FFT_STAMP_ARGS b, W, ld, st, true); }
};
// Four-pass variant (radices <= 8): more barriers but much shorter butterfly chains per pass —
// the better trade for SMALL jobs, whose cost is the latency of one workgroup, not throughput.
template <int NA_, int NB_, int NT_, int A0, int A1, int A2, int A3, int B0, int B1, int B2, int B3>
struct PairSpec4 {
    static constexpr int NA = NA_, NB = NB_, NT = NT_;
    static constexpr bool prefetch = true;
    static constexpr int RB0 = B0; // radix of the first inverse pass: how many filter values a thread needs
    struct Tw { cf a1, a2, a3, b1, b2, b3; };
    static __device__ __forceinline__ Tw twiddles(const cf *WA, const cf *WB)
    {
        Tw t;
        t.a1 = pass_twiddle<NA, A0, A1, NT>(WA); t.a2 = pass_twiddle<NA, A0 * A1, A2, NT>(WA);
        t.a3 = pass_twiddle<NA, A0 * A1 * A2, A3, NT>(WA);
        t.b1 = pass_twiddle<NB, B0, B1, NT>(WB); t.b2 = pass_twiddle<NB, B0 * B1, B2, NT>(WB);
        t.b3 = pass_twiddle<NB, B0 * B1 * B2, B3, NT>(WB);
        return t;
    }
    template <typename Ld, typename St> static __device__ __forceinline__ void fwd(FFT_STAMP_DECL cf *b, const cf *W, Ld ld, St st, bool in_lds, const Tw &t)
    { fft_ct_pre<NA, -1, NT, A0, A1, A2, A3>(b, W, ld, st, in_lds, t.a1, t.a2, t.a3); }
    template <typename Ld, typename St> static __device__ __forceinline__ void inv(FFT_STAMP_DECL cf *b, const cf *W, Ld ld, St st, bool in_lds, const Tw &t)
    { fft_ct_pre<NB, +1, NT, B0, B1, B2, B3>(b, W, ld, st, in_lds, t.b1, t.b2, t.b3); }
};
typedef PairSpec4<2560, 2352, 512, 5, 8, 8, 8, 6, 7, 7, 8> Pair2560x2352L; // low-latency schedule
// 48k <-> 44.1k (L/M = 147/160 and 160/147): k = 32, and k = 16 for small jobs
// Three-pass schedule of each transform length in use (first radix 21: conflict-free as it is;
// first radix 16: swizzled layout between pass 1 and 2).
template <int N> struct Sched;
#ifdef FFT_EXPERIMENTS
#define HIPSOXR_SCHED_4410(X) X(4410, 15, 14, 21, false) // (first radix 15: 4410 / 15 = 294 divides the 3528-frame hop of the 44.1k -> 16k blocks — k_fft_strided2's walk)
#else
#define HIPSOXR_SCHED_4410(X) X(4410, 21, 14, 15, false) // (the order the product runs: configs[2] 47 us, against 52 us with the radix-15 pass first)
#endif
#define HIPSOXR_SCHED_LIST(X)                                                                                        \
    X(7056, 21, 16, 21, false) X(5376, 21, 16, 16, false) X(5120, 16, 16, 20, true) X(4704, 21, 16, 14, false)       \
    HIPSOXR_SCHED_4410(X) X(4096, 16, 16, 16, true) X(3840, 16, 16, 15, true) X(3584, 14, 16, 16, false)             \
    X(3528, 21, 12, 14, false) X(2688, 21, 16, 8, false) X(2560, 16, 16, 10, true) X(2352, 21, 16, 7, false)         \
    X(2048, 16, 16, 8, true) X(1792, 7, 16, 16, false) X(1024, 16, 8, 8, true) X(1600, 16, 10, 10, true)             \
    X(1280, 5, 16, 16, false) X(1176, 21, 8, 7, false) X(896, 7, 16, 8, false)
#define HIPSOXR_SCHED(N, r0, r1, r2, swz) \
    template <> struct Sched<N> { static constexpr int R0 = r0, R1 = r1, R2 = r2; static constexpr bool SWZ = swz; };
HIPSOXR_SCHED_LIST(HIPSOXR_SCHED)
#undef HIPSOXR_SCHED
static bool sched_of(int n, int *r0, int *r1, int *r2) // the same table at run time (host: fft_build's tables for -DFFT_DIF)
{
    switch (n) {
#define HIPSOXR_SCHED(N, a0, a1, a2, swz) case N: *r0 = a0; *r1 = a1; *r2 = a2; return true;
        HIPSOXR_SCHED_LIST(HIPSOXR_SCHED)
#undef HIPSOXR_SCHED
    default: return false;
    }
}
template <int NA, int NB, int NT>
using PairOf = PairSpec<NA, NB, NT, Sched<NA>::R0, Sched<NA>::R1, Sched<NA>::R2, Sched<NA>::SWZ, Sched<NB>::R0, Sched<NB>::R1,
                        Sched<NB>::R2, Sched<NB>::SWZ>;
typedef PairOf<2560, 2352, 384> Pair2560x2352; // 147/160, small blocks (three-pass)

template <typename Spec>
__global__ void __launch_bounds__(Spec::NT) k_fft_pair(FftArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf *cur = reinterpret_cast<cf *>(smem_raw);
    constexpr int NA = Spec::NA, NB = Spec::NB;
#ifdef FFT2_TRACE
    unsigned long long *g_tr = nullptr;
    int g_tri = 0;
#endif

    // What is paired: two consecutive blocks of one column (planar / mono data), or — for
    // interleaved data with an even channel count (a.chpair) — the same block of two neighbouring
    // channels, whose samples are one aligned float2 in memory: loads and stores then move 8
    // contiguous bytes per lane instead of two 4-byte words with a channel stride between lanes.
    // Grid: block pairs along x, columns along y.  Channel-pair mode is XCD-aware: consecutive
    // workgroup ids are dealt round-robin to the 8 XCDs, each with its own L2, and the channel pairs
    // of one block of frames share every cache line of the interleaved data — so they are given
    // ids that are congruent mod 8 and adjacent in dispatch order (x = 8*(slot) + xcd,
    // slot = chunk*pairs + pair, block = 8*chunk + xcd).  Without this each line is fetched and
    // (partially) written once per channel pair: 2.3x / 4x the algorithmic bytes at 8 channels.
    const bool cp = a.chpair != 0, xm = a.xcd_map != 0; // xcd_map: interleaved data (pairs of channels or single channels)
    const uint32_t cpr = cp ? a.n_channels / 2 : a.n_channels;
    const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    // (integer division by a run-time divisor goes through the vector ALU: tell the compiler that the
    // results are wave-uniform, or every address derived from them lives in VGPRs — +22 registers)
    const uint32_t cu = __builtin_amdgcn_readfirstlane(xm ? slot % cpr : blockIdx.y % cpr); // channel unit: pair (cp) or channel
    const uint32_t ch = cp ? 2 * cu : cu;
    const uint32_t clip = __builtin_amdgcn_readfirstlane(xm ? blockIdx.y : blockIdx.y / cpr);
    const int64_t bx = (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane(xm ? (slot / cpr) * 8 + xcd : blockIdx.x);
    if (xm && bx >= a.pairs_per_col) return; // grid.x is padded to a multiple of 8 work items per channel unit
    const int64_t pa = (cp ? 1 : 2) * bx * a.hop_periods - a.lead_periods; // first period of block a
    const int64_t pb = cp ? pa : pa + a.hop_periods;                                         // ... of block b
    const int64_t ina = pa * a.M, inb = pb * a.M, outa = pa * a.L, outb = pb * a.L;
    const float *xin = (const float *)a.in + (int64_t)clip * a.ics + (int64_t)ch * a.ichs;
    auto lds_store = [&](int n, cf v) { cur[n] = v; };

    // ---- forward: z[n] = x_a[n] + i x_b[n], first pass straight from HBM --------------------------
    // (one instantiation of the transform for interior and edge blocks alike, so that both run the
    // same instruction sequence and round identically)
    // small-job specs: every table value the workgroup will need (twiddles of the six twiddled
    // passes, the filter values of the first inverse pass) is requested now, behind the input loads
    typename Spec::Tw tw;
    cf hpre[Spec::prefetch ? Spec::RB0 : 1];
    if constexpr (Spec::prefetch) {
        tw = Spec::twiddles(a.WA2, a.WB2);
        constexpr int nbB = NB / Spec::RB0;
        const int jb = (int)threadIdx.x < nbB ? (int)threadIdx.x : 0;
#pragma unroll
        for (int t = 0; t < Spec::RB0; ++t) {
            const int n = jb + t * nbB, q = n > NB / 2 ? NB - n : n;
            cf h = a.Hs[q];
            if (n > NB / 2) h.y = -h.y;
            hpre[t] = h;
        }
    }
    const bool interior = ina >= 0 && inb + NA <= a.in_frames && a.ifs < (1 << 16);
    const int32_t ifs32 = (int32_t)a.ifs;
    const bool unit = a.ifs == 1 && !cp;
    const float *xa = xin + ina * a.ifs, *xb = xin + inb * a.ifs;
    Spec::fwd(FFT_STAMP_ARGS cur, a.WA2, [&](int n) -> cf {
#if defined(FFT_ABL) && (FFT_ABL & 2) // timing ablation (tools/fft_ablate.sh): no input loads
        if (interior) return make_float2((float)n * 1e-3f, (float)(n ^ 5) * 1e-3f);
#endif
        if (interior) {
            if (unit) return make_float2(xa[n], xb[n]);           // planar / mono: the common fast path
            if (cp) return *reinterpret_cast<const float2 *>(xa + n * ifs32);
            return make_float2(xa[n * ifs32], xb[n * ifs32]);
        }
        const int64_t la = ina + n, lb = inb + n;
        return make_float2((la >= 0 && la < a.in_frames) ? xin[la * a.ifs] : 0.f,
                           (lb >= 0 && lb < a.in_frames) ? xin[lb * a.ifs + (cp ? 1 : 0)] : 0.f);
    }, lds_store, false, tw);
    __syncthreads();
#if defined(FFT_ABL) && (FFT_ABL & 4) // timing ablation (tools/fft_latency.sh): stop after the forward transform
    if (a.out_frames >= 0) return;
#endif

    // ---- inverse: bin n of the output grid <- bin n (n <= NB/2) or n + NA - NB (negative
    //      frequencies) of the input grid, times H (Hermitian); result n = (y_a[n], y_b[n]) ---------
    float *yo = (float *)a.out + (int64_t)clip * a.ocs + (int64_t)ch * a.ochs;
    const int32_t v0 = a.v0, v1 = a.v0 + a.hop_out;
    auto h_load = [&](int n, int t) -> cf { // t: which of the butterfly's inputs (selects the prefetched H)
        const bool neg = n > NB / 2;
        const int q = neg ? NB - n : n; // |frequency| in bins
        if constexpr (!Spec::prefetch) { // H is real (see fft_build): one table word and a real x complex product per bin
            const float h = a.Hr[q];
            if constexpr (NA >= NB) {
                const cf x = cur[neg ? n + (NA - NB) : n];
                return make_float2(x.x * h, x.y * h);
            } else {
                const bool in_band = q < NA / 2; // the input Nyquist bin itself carries only stop-band energy
                const cf x = cur[in_band ? (neg ? NA - q : q) : 0];
                return in_band ? make_float2(x.x * h, x.y * h) : make_float2(0.f, 0.f);
            }
        } else {
            const cf h = hpre[t];
            if constexpr (NA >= NB) { // down-sampling: the spectrum is truncated
                cf y = cmul(cur[neg ? n + (NA - NB) : n], h);
                if (n == NB / 2) y = cadd(y, cmul(cur[n + (NA - NB)], cconj(h)));
                return y;
            } else {                  // up-sampling: the spectrum is zero-extended
                const bool in_band = q < NA / 2;
                const cf y = cmul(cur[in_band ? (neg ? NA - q : q) : 0], h);
                return in_band ? y : make_float2(0.f, 0.f);
            }
        }
    };
    auto out_store = [&](int n, cf w) {
#if defined(FFT_ABL) && (FFT_ABL & 1) // timing ablation: no output stores
        if (w.x != 1234.5f) return;
#endif
        if (n >= v0 && n < v1) {
            const int64_t ka = outa + n, kb = outb + n;
            if (a.ofs == 1) {
                // streaming (non-temporal) stores: the output is not read again by this launch, and
                // keeping it out of L2's way is worth ~7 % on the batch workload (151 -> 140 us)
                if (ka >= 0 && ka < a.out_frames) __builtin_nontemporal_store(w.x, &yo[ka]);
                if (kb >= 0 && kb < a.out_frames) __builtin_nontemporal_store(w.y, &yo[kb]);
            } else if (cp) { // one aligned float2 per frame
                if (ka >= 0 && ka < a.out_frames) *reinterpret_cast<float2 *>(&yo[ka * a.ofs]) = w;
            } else {
                if (ka >= 0 && ka < a.out_frames) yo[ka * a.ofs] = w.x;
                if (kb >= 0 && kb < a.out_frames) yo[kb * a.ofs] = w.y;
            }
        }
    };
    Spec::inv(FFT_STAMP_ARGS cur, a.WB2, h_load, out_store, true, tw);
}

// ---------------------------------------------------------------------------------------------
// Paired-block kernel, second generation, for unit-stride columns (mono / planar data; batches).
// Same transform chain and schedules as k_fft_pair; what differs is how it touches HBM:
//   * input: raw buffer loads whose descriptor covers [first sample of block a, end of the column):
//     the hardware range check returns 0 past the end of the signal (no per-element bounds code,
//     one path for interior and last pairs), and the per-butterfly offsets t*N/R0 ride in the
//     instruction's scalar offset instead of 64-bit vector address arithmetic;
//   * output: the last inverse pass writes its kept outputs into LDS as the two contiguous runs they
//     are in memory (block a then block b: 2*hop_out consecutive floats of the column), and the
//     workgroup then stores that run with 16-byte-aligned float4 stores — every wave writes 1 KB of
//     whole 16-byte granules instead of 256 unaligned bytes per instruction (k_fft_pair: write traffic
//     1.17x the algorithmic bytes with streaming stores).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void *uniform_ptr(void *p) // the same address, provably wave-uniform (two v_readfirstlane)
{
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    return reinterpret_cast<void *>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32) |
                                    (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v));
}
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
template <typename Real> __device__ __forceinline__ Real buf_load_real(__amdgpu_buffer_rsrc_t r, int voff, int soff);
template <> __device__ __forceinline__ float buf_load_real<float>(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
template <> __device__ __forceinline__ double buf_load_real<double>(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
__device__ __forceinline__ void buf_store_real(float v, __amdgpu_buffer_rsrc_t r, int voff)
{
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, 0, 0);
}
__device__ __forceinline__ void buf_store_real(double v, __amdgpu_buffer_rsrc_t r, int voff)
{
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u_t, v), r, voff, 0, 0);
}
// per-precision views of the kernel arguments
#ifdef FFT_DIF
// ---------------------------------------------------------------------------------------------
// Wave-local schedule (round 3 experiment, -DFFT_DIF).  The three Stockham passes of fft_ct3 are three all-to-all
// exchanges through LDS, each fenced by workgroup barriers (11 per block pair; the batch launch's waves spend 49 % of
// their cycles parked at them).  Decimation in FREQUENCY instead: the first pass (radix R0 over elements S = R1 R2
// apart, one butterfly per thread, its inputs straight from HBM or — inverse — from the spectrum) leaves R0 independent
// S-point transforms, each contiguous in LDS.  A sub-transform is given to GL = max(R1, R2) lanes of ONE wave, so its
// two remaining passes exchange data only between lanes of that wave: no workgroup barrier, only the LDS queue's own
// ordering — waves drift apart and one wave's butterflies overlap another's LDS traffic.  6 barriers per pair.
//   first pass : thread j < S: u[t] = in[j + S t]; DFT_R0; L[k0 S + j] = u[k0]
//   sub-transform k0, lane n < R2: u[t] = Ls[n + R2 t] W_N^((n + R2 t) k0); DFT_R1; u[k1] *= W_S^(n k1); Ls[k1 R2 + n] = u[k1]
//                     lane k1 < R1: u[n] = Ls[k1 R2 + n]; DFT_R2 -> u[k2] = X[R0 (R1 k2 + k1) + k0]
// The forward transform leaves bin (k0, k1, k2) at L[k0 S + ((k1 + k0) % R1) R2 + k2] — rows rotated by k0, so that the
// inverse's first pass (consecutive lanes = consecutive bins = consecutive k0) does not read 16 lanes from one bank —
// and the inverse's first pass finds it through a table (FftArgs::HP: filter value and byte offset per output-grid bin:
// one 8-byte load instead of the |frequency| index arithmetic).
// ---------------------------------------------------------------------------------------------
template <int R, typename C> __device__ __forceinline__ void tw_apply(C *u, C w1, C w4) // u[k] *= w1^k (w4 = w1^4 from the table, see fft_pass_ct)
{
    C pw[R];
    pw[1] = w1;
    if constexpr (R >= 10) {
#pragma unroll
        for (int t = 2; t < R; ++t) {
            const int a4 = t / 4, b4 = t % 4;
            if (a4 == 0) pw[t] = cmul(pw[t - 1], w1);
            else if (b4 == 0) pw[t] = a4 == 1 ? w4 : (a4 % 2 == 0 ? cmul(pw[t / 2], pw[t / 2]) : cmul(pw[t - 4], w4));
            else pw[t] = cmul(pw[4 * a4], pw[b4]);
        }
    } else {
#pragma unroll
        for (int t = 2; t < R; ++t) pw[t] = (t & 1) ? cmul(pw[t - 1], w1) : cmul(pw[t / 2], pw[t / 2]);
    }
#pragma unroll
    for (int t = 1; t < R; ++t) u[t] = cmul(u[t], pw[t]);
}
template <int N, int R0, int S, int SIGN, int NT, bool SYNC, typename C, typename Load>
__device__ __forceinline__ void dif_first(C *L, const C *W, Load load)
{
    static_assert(R0 * S == N && S <= NT, "first pass: one butterfly per thread");
    const int j = fft_tid();
    const bool act = S == NT || j < S;
    C u[R0];
    (void)W;
    if (act) {
#pragma unroll
        for (int t = 0; t < R0; ++t) u[t] = load(j + S * t, t);
    }
    if (SYNC) __syncthreads(); // in place: every thread holds its inputs
    if (act) {
        dft_r<R0, SIGN>(u); // (its twiddles W_N^(j k0) wait for the sub-transform's loads: outputs go straight to LDS)
#pragma unroll
        for (int k = 0; k < R0; ++k) L[k * S + j] = u[k];
    }
}
// the two wave-local passes; `out(k0, k1, k2, value)` takes the results (LASTSYNC: behind a workgroup barrier — they go
// to places other waves still read).  Table values are fetched by dif_local_pre, which the caller runs IN FRONT of the
// workgroup barrier before the sub-transforms (the first pass's registers are free by then: the loads fly while the
// workgroup gathers).  Rows of R2 = 0 (mod 4) points start 8 rows apart in the same bank: such rows swap neighbouring
// columns in their upper half (column c of row r at c ^ ((r >> 3) & 1); dif_col) — two lane bases, no index arithmetic.
template <int R0, int R1, int R2, int NT> struct DifLane {
    static constexpr int GL = R1 > R2 ? R1 : R2, GPW = 64 / GL;
    int li, g;
    bool act;
    __device__ __forceinline__ DifLane()
    {
        const int tid = fft_tid(), lane = tid & 63, wave = tid >> 6;
        const int gl = lane / GL;
        li = lane - gl * GL;
        g = wave * GPW + gl;
        act = gl < GPW && g < R0;
    }
};
template <int R2> __host__ __device__ constexpr bool dif_swz() { return R2 % 4 == 0; }
template <typename C, int R1> struct DifPre { C f[R1]; C w1, w4; };
template <int N, int R0, int R1, int R2, int NT, typename C>
__device__ __forceinline__ void dif_local_pre(const C *W, DifPre<C, R1> &p)
{
    const DifLane<R0, R1, R2, NT> ln;
    if (ln.act && ln.li < R2) {
        // the first pass's twiddle of element n' = li + R2 t of sub-transform g: W_N^(n' g), straight from the table
        // (n' g < S R0 = N: no reduction)
        const C *Wg = W + ln.li * ln.g;
#pragma unroll
        for (int t = 0; t < R1; ++t) p.f[t] = Wg[(R2 * t) * ln.g];
        p.w1 = W[R0 * ln.li];
        p.w4 = R1 >= 10 ? W[4 * R0 * ln.li] : p.w1;
    }
}
template <int N, int R0, int R1, int R2, int SIGN, int NT, bool LASTSYNC, typename C, typename Out>
__device__ __forceinline__ void dif_local(C *L, const DifPre<C, R1> &p, Out out)
{
    constexpr int S = R1 * R2;
    constexpr bool XS = dif_swz<R2>();
    static_assert(R0 * S == N && DifLane<R0, R1, R2, NT>::GPW * (NT / 64) >= R0, "sub-transforms per wave");
    const DifLane<R0, R1, R2, NT> ln;
    const int li = ln.li, g = ln.g;
    C *Ls = L + g * S;
    {
        C u[R1];
        if (ln.act && li < R2) {
#pragma unroll
            for (int t = 0; t < R1; ++t) u[t] = Ls[li + R2 * t]; // (first-pass layout: no swizzle)
#pragma unroll
            for (int t = 0; t < R1; ++t) u[t] = cmul(u[t], p.f[t]);
            dft_r<R1, SIGN>(u);
            tw_apply<R1>(u, p.w1, p.w4);
            C *lo = Ls + li, *hi = Ls + (XS ? li ^ 1 : li);
#pragma unroll
            for (int k = 0; k < R1; ++k) (((k >> 3) & 1) ? hi : lo)[k * R2] = u[k];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {
        C u[R2];
        const bool a3 = ln.act && li < R1;
        if (a3) {
            const int sx = XS ? (li >> 3) & 1 : 0;
            const C *ev = Ls + li * R2 + sx, *od = Ls + li * R2 - sx;
#pragma unroll
            for (int n = 0; n < R2; ++n) u[n] = ((n & 1) ? od : ev)[n];
            dft_r<R2, SIGN>(u);
        }
        if (LASTSYNC) __syncthreads();
        if (a3) {
#pragma unroll
            for (int k = 0; k < R2; ++k) out(g, li, k, u[k]);
        }
    }
}
#endif

template <typename Real> struct PairTabs;
template <> struct PairTabs<float> {
    typedef float2 C; typedef float4 V16;
    static __device__ __forceinline__ const C *wa(const FftArgs &a) { return a.WA2; }
    static __device__ __forceinline__ const C *wb(const FftArgs &a) { return a.WB2; }
    static __device__ __forceinline__ const float *hr(const FftArgs &a) { return a.Hr; }
};
template <> struct PairTabs<double> {
    typedef double2 C; typedef double2 V16;
    static __device__ __forceinline__ const C *wa(const FftArgs &a) { return a.WA2d; }
    static __device__ __forceinline__ const C *wb(const FftArgs &a) { return a.WB2d; }
    static __device__ __forceinline__ const double *hr(const FftArgs &a) { return a.Hrd; }
};

// Real = float: float32 device jobs.  Real = double: float64 device jobs — libsoxr's own VHQ engine is a
// float64 one; the same chain in double2 (LDS 16 bytes per point), results within the
// method's own floor of the float64 direct form (the neglected stop-band aliasing, ~3e-10 for VHQ).
// IO = the signal's element type when it differs from the arithmetic: <double, float> is float32 I/O on float64
// arithmetic — what libsoxr's VHQ recipe itself does for float32 clients (reference src/soxr_ext.cpp:74,228 hand the
// recipe to soxr_quality_spec) — selected by HIPSOXR_KERNEL_FFT_F64.  Loads widen, the staged run and
// the stores are in the I/O type; everything between is the float64 instance.
// One work item = one pair of blocks of one column: item (col, bx).  `staged()` runs in every thread after the run has
// been staged in LDS and in front of the barrier that publishes it (the persistent kernel posts its next item there).
// Returns false, having done nothing, when the item lies beyond its clip (ragged batches).
template <typename Spec, typename Real, typename IO, typename Staged>
__device__ __forceinline__ bool pair2_item(const FftArgs &a, unsigned char *smem_raw, uint32_t col, int64_t bx, Staged staged)
{
    typedef typename PairTabs<Real>::C C;
    typedef typename PairTabs<IO>::V16 V16;
    constexpr int ES = (int)sizeof(IO), EPS = 16 / ES; // element size, elements per 16-byte store
    C *cur = reinterpret_cast<C *>(smem_raw);
    IO *stage = reinterpret_cast<IO *>(smem_raw);
    constexpr int NA = Spec::NA, NB = Spec::NB, NT = Spec::NT, nbA = NA / Spec::RA0;
#ifdef FFT2_TRACE
    unsigned long long *g_tr = a.trace ? a.trace + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (NT / 64) + threadIdx.x / 64) * 16 : nullptr;
    int g_tri = 0;
    FFT_STAMP();
#endif

    const uint32_t ch = __builtin_amdgcn_readfirstlane(col % a.n_channels);
    const uint32_t clip = __builtin_amdgcn_readfirstlane(col / a.n_channels);
    const int64_t pa = 2 * bx * a.hop_periods - a.lead_periods; // first period of block a; block b starts hop_periods later
    const int64_t ina = pa * a.M, outa = pa * a.L;
    const int32_t hop_in = (int32_t)(a.hop_periods * a.M);
    // ragged batch: this clip's own place and length (four scalar loads; the grid spans the longest clip, so a
    // workgroup beyond its clip's last pair has nothing to do)
    int64_t clip_in = (int64_t)clip * a.ics, clip_out = (int64_t)clip * a.ocs, in_frames = a.in_frames, out_frames = a.out_frames;
    if (a.clip_tab) {
        const int64_t *row = a.clip_tab + 4 * (size_t)clip;
        clip_in = row[0]; in_frames = row[1]; clip_out = row[2]; out_frames = row[3];
        if (outa + a.v0 >= out_frames) return false;
    }
    const IO *xin = (const IO *)a.in + clip_in + (int64_t)ch * a.ichs;
#if defined(FFT2_ABL) && (FFT2_ABL & 8)
    auto lds_store = [&](int n, C v) { if (v.x == (Real)1234.5) cur[n] = v; };
#else
    auto lds_store = [&](int n, C v) { cur[n] = v; };
#endif
    typename Spec::Tw tw;

#ifdef FFT_DIF
    // (experiment builds only: in the product the three definitions below stand where the inverse transform starts —
    //  hoisting them costs the plain kernel 2-4 %)
    const int32_t v0 = a.v0, v1 = a.v0 + a.hop_out, hop_out = a.hop_out;
    IO *ybase = (IO *)a.out + clip_out + (int64_t)ch * a.ochs + (outa + v0);
    const int32_t sh = (int32_t)((reinterpret_cast<uintptr_t>(ybase) / ES) & (EPS - 1));
    constexpr bool kDif = Spec::dif && sizeof(Real) == 4 && sizeof(IO) == 4 && NA >= NB; // (NA < NB: bins beyond the input band would have to read zeros)
    if constexpr (kDif) {
        // wave-local schedule (dif_first / dif_local): 6 workgroup barriers per pair instead of 11
        constexpr int A0 = Spec::RA0, A1 = Spec::RA1, A2 = Spec::RA2, B0 = Spec::RB0, B1 = Spec::RB1, B2 = Spec::RB2;
        constexpr int SA = A1 * A2, SB = B1 * B2;
        if (ina >= 0) {
            const int64_t left = (in_frames - ina) * ES;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                uniform_ptr((void *)(xin + ina)), 0, __builtin_amdgcn_readfirstlane((int)(left < 0 ? 0 : left > 0x40000000 ? 0x40000000 : left)), 0x00020000);
            dif_first<NA, A0, SA, -1, NT, false>(cur, PairTabs<Real>::wa(a), [&](int n, int t) -> C {
                const int j4 = (n - t * SA) * ES;
                return C((Real)buf_load_real<IO>(rs, j4, t * SA * ES), (Real)buf_load_real<IO>(rs, j4, (t * SA + hop_in) * ES));
            });
        } else {
            const int64_t inb = ina + hop_in;
            dif_first<NA, A0, SA, -1, NT, false>(cur, PairTabs<Real>::wa(a), [&](int n, int) -> C {
                const int64_t la = ina + n, lb = inb + n;
                return C((la >= 0 && la < in_frames) ? (Real)xin[la] : (Real)0, (lb >= 0 && lb < in_frames) ? (Real)xin[lb] : (Real)0);
            });
        }
        FFT_STAMP();
        {
            DifPre<C, A1> pre;
            dif_local_pre<NA, A0, A1, A2, NT>(PairTabs<Real>::wa(a), pre);
            __syncthreads();
            FFT_STAMP();
            // the filter rides on the forward transform's last stores (20 consecutive values per lane: five 16-byte loads)
            const float *HF = reinterpret_cast<const float *>(a.HP) + NB;
            dif_local<NA, A0, A1, A2, -1, NT, false>(cur, pre, [&](int g, int li, int k, C v) {
                const float h = HF[(g * A1 + li) * A2 + k];
                const int row = (li + g) % A1; // rows rotated by the sub-transform's index (see the inverse's loads)
                const int sx = dif_swz<A2>() ? (row >> 3) & 1 : 0;
                cur[g * SA + row * A2 + ((k & 1) ? k - sx : k + sx)] = C(v.x * h, v.y * h);
            });
        }
        const uint32_t *HPo = reinterpret_cast<const uint32_t *>(a.HP);
        FFT_STAMP();
        __syncthreads();
        FFT_STAMP();
        dif_first<NB, B0, SB, +1, NT, true>(cur, PairTabs<Real>::wb(a), [&](int n, int) -> C {
            return *reinterpret_cast<const C *>(smem_raw + HPo[n]);
        });
        FFT_STAMP();
        DifPre<C, B1> preb;
        dif_local_pre<NB, B0, B1, B2, NT>(PairTabs<Real>::wb(a), preb);
        __syncthreads();
        FFT_STAMP();
        dif_local<NB, B0, B1, B2, +1, NT, true>(cur, preb, [&](int g, int li, int k, C w) {
            const int m = B0 * (B1 * k + li) + g; // local output index
            if (m >= v0 && m < v1) {
                stage[m - v0 + sh] = (IO)w.x;
                stage[m - v0 + sh + hop_out] = (IO)w.y;
            }
        });
        FFT_STAMP();
    } else {
#endif
    // ---- forward: z[n] = x_a[n] + i x_b[n], first pass straight from HBM --------------------------
#ifdef FFT_LDS_DMA
    // Experiment (round 3, measured slower — profiles/r03_ab_experiments.txt): the two blocks land in LDS by DMA
    // (`buffer_load_dwordx4 ... lds`, 1 KB per wave instruction: 40 instead of 160 x 6 vector-memory instructions per
    // pair, no VGPR destinations), x_a in floats [0, NA), x_b in [NA, 2 NA); one barrier; the first pass then reads
    // its operands from LDS and stores in place behind a barrier of its own.
    constexpr bool kDma = sizeof(IO) == 4 && sizeof(Real) == 4 && NA % 256 == 0;
#else
    constexpr bool kDma = false;
#endif
    if (kDma && ina >= 0) {
#ifdef FFT_LDS_DMA
        const int64_t left = (in_frames - ina) * ES;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            uniform_ptr((void *)(xin + ina)), 0, __builtin_amdgcn_readfirstlane((int)(left < 0 ? 0 : left > 0x40000000 ? 0x40000000 : left)), 0x00020000);
        Real *land = reinterpret_cast<Real *>(smem_raw);
        const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
        constexpr int CH = NA / 256, NWV = NT / 64;
#pragma unroll
        for (int r = 0; r < (2 * CH + NWV - 1) / NWV; ++r) {
            const int c = wave + r * NWV; // wave-uniform chunk of 256 floats
            if (c < 2 * CH) {
                const int blk = c >= CH ? 1 : 0, cc = c - blk * CH;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(land + blk * NA + cc * 256), 16,
                                                         lane * 16 + cc * 1024, blk * hop_in * ES, 0, 0);
            }
        }
        __syncthreads();
        FFT_STAMP();
        Spec::fwd(FFT_STAMP_ARGS cur, PairTabs<Real>::wa(a), [&](int n, int) -> C { return C(land[n], land[NA + n]); }, lds_store, true, tw);
#endif
    } else if (ina >= 0) {
        const int64_t left = (in_frames - ina) * ES; // bytes from block a's first sample to the end of the column
        // (descriptor words marked wave-uniform: in the resident-workgroup kernel the item comes out of LDS and the
        //  compiler would otherwise keep the descriptor in VGPRs and wrap every load in a waterfall loop)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            uniform_ptr((void *)(xin + ina)), 0, __builtin_amdgcn_readfirstlane((int)(left < 0 ? 0 : left > 0x40000000 ? 0x40000000 : left)), 0x00020000);
        Spec::fwd(FFT_STAMP_ARGS cur, PairTabs<Real>::wa(a), [&](int n, int t) -> C {
            const int j4 = (n - t * nbA) * ES; // the butterfly's own offset (one VGPR for all t)
#if defined(FFT2_ABL) && (FFT2_ABL & 2)
            return C((Real)(j4 + t) * (Real)1e-4, (Real)(j4 ^ t) * (Real)1e-4);
#endif
            return C((Real)buf_load_real<IO>(rs, j4, t * nbA * ES), (Real)buf_load_real<IO>(rs, j4, (t * nbA + hop_in) * ES));
        }, lds_store, false, tw);
    } else { // the first pair of a column reaches before its start: explicit zero-extension
        const int64_t inb = ina + hop_in;
        Spec::fwd(FFT_STAMP_ARGS cur, PairTabs<Real>::wa(a), [&](int n, int) -> C {
            const int64_t la = ina + n, lb = inb + n;
            return C((la >= 0 && la < in_frames) ? (Real)xin[la] : (Real)0, (lb >= 0 && lb < in_frames) ? (Real)xin[lb] : (Real)0);
        }, lds_store, false, tw);
    }
    // the filter values of the first inverse pass, one barrier early (see TwPre): the butterfly of thread j takes bins
    // j + t * NB/RB0, t = 0 .. RB0-1, at |frequency| q = min(n, NB - n)
    const Real *Hr = PairTabs<Real>::hr(a);
#ifdef FFT_EARLY_TABLES
    constexpr int RB0 = Spec::RB0, nbB = NB / RB0;
    Real hpre[RB0];
    {
        const int tid = fft_tid(), jb = tid < nbB ? tid : 0;
#pragma unroll
        for (int t = 0; t < RB0; ++t) {
            const int n = jb + t * nbB;
            hpre[t] = Hr[n > NB / 2 ? NB - n : n];
        }
    }
#endif
    __syncthreads();
    FFT_STAMP();

    // ---- inverse (see k_fft_pair), last pass into the staging layout -------------------------------
#ifndef FFT_DIF
    const int32_t v0 = a.v0, v1 = a.v0 + a.hop_out, hop_out = a.hop_out;
    IO *ybase = (IO *)a.out + clip_out + (int64_t)ch * a.ochs + (outa + v0); // run[0]; outa + v0 >= 0
    // LDS element index == run index + sh: the 16-byte phases of staging and memory agree
    const int32_t sh = (int32_t)((reinterpret_cast<uintptr_t>(ybase) / ES) & (EPS - 1));
#endif
    auto h_load = [&](int n, int t) -> C { // bin n of the output grid <- bin n or n + NA - NB of the input grid, times (real) H
        const bool neg = n > NB / 2;
        const int q = neg ? NB - n : n; // |frequency| in bins
#ifdef FFT_EARLY_TABLES
        const Real h = hpre[t];
        (void)q;
#else
        const Real h = Hr[q];
        (void)t;
#endif
        if constexpr (NA >= NB) {
            const C x = cur[neg ? n + (NA - NB) : n];
            return C(x.x * h, x.y * h); // (the Nyquist bin's alias term is dropped with Im H: stop band, < -170 dB)
        } else {
            const bool in_band = q < NA / 2;
            const C x = cur[in_band ? (neg ? NA - q : q) : 0];
            return in_band ? C(x.x * h, x.y * h) : C((Real)0, (Real)0);
        }
    };
    Spec::inv_staged(FFT_STAMP_ARGS cur, PairTabs<Real>::wb(a), h_load, [&](int n, C w) {
        if (n >= v0 && n < v1) {
            stage[n - v0 + sh] = (IO)w.x;
            stage[n - v0 + sh + hop_out] = (IO)w.y;
        }
    });
#ifdef FFT_DIF
    } // (!kDif)
#endif
    staged();
    __syncthreads();
    FFT_STAMP();

    // ---- store the run: elements [0, valid) of it exist in the column ----------------------------------
    const int64_t remain = out_frames - (outa + v0);
    const int32_t valid = (int32_t)(remain < 0 ? 0 : remain > 2 * (int64_t)hop_out ? 2 * (int64_t)hop_out : remain);
    // 16-byte buffer stores: the descriptor starts at the 16-byte granule that holds run[0] (sh elements before it)
    // and ends with the run, so the hardware range check drops what lies beyond the column (and the trips past the
    // run: no trip count, no branches — every LDS read and every store of the thread is in flight at once; 11.00 ->
    // 10.85 us on the 60 s clip against per-granule bounds tests and pointer stores, nothing on the batch).  The
    // first granule's sh leading elements belong to the previous run: that one granule goes element by element.
    {
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr((void *)(ybase - sh)), 0,
                                                                             __builtin_amdgcn_readfirstlane((valid + sh) * ES), 0x00020000);
        constexpr int QMAX = (2 * (NB - 1) + EPS - 1 + EPS) / EPS; // 2 hop_out < 2 NB elements, + sh
        constexpr int LQ = (int)((NA > NB ? NA : NB) * sizeof(C) / 16); // 16-byte granules of the LDS buffer
        const int tid_out = fft_tid();
#pragma unroll
        for (int it = 0; it < (QMAX + NT - 1) / NT; ++it) {
            const int q = tid_out + it * NT;
            const V16 v = *reinterpret_cast<const V16 *>(stage + EPS * (q < LQ ? q : LQ - 1));
#if defined(FFT2_ABL) && (FFT2_ABL & 4)
            if (v.x != (IO)1234.5) continue;
#endif
            if (q == 0 && sh != 0) {
                const IO *e = reinterpret_cast<const IO *>(&v);
#pragma unroll
                for (int c = 0; c < EPS; ++c)
                    if (c >= sh && c - sh < valid) ybase[c - sh] = e[c];
            } else {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u_t, v), ro, q * 16, 0, 0);
            }
        }
    }
#ifdef FFT2_TRACE
    if (g_tr && (threadIdx.x & 63) == 0) { // where the wave ran: HW_ID (wave/simd/cu/sh/se fields) and the XCC id
        g_tr[13] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        g_tr[14] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
    g_tri = 15;
    FFT_STAMP();
#endif
    return true;
}

template <typename Spec, typename Real, typename IO = Real>
__global__ void __launch_bounds__(Spec::NT) k_fft_pair2(FftArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    pair2_item<Spec, Real, IO>(a, smem_raw, blockIdx.y, blockIdx.x, [] {});
}

// The same work items served by RESIDENT workgroups (round 3 experiment, opt-in: HIPSOXR_FFT_PERSIST; slower than the
// grid-per-item kernel, see launch_fft).  A per-CU timeline of the grid-per-item kernel on the
// batch workload (tools/trace_pair2.py) showed a CU holding 3.1 of its 4 workgroup slots on average: a freed slot
// waits a median of 1500 cycles, 8700 at the 90th percentile, for the dispatcher's next workgroup, every workgroup
// re-reads its arguments, and the launch ends with a drain of one whole workgroup lifetime.  Here the grid is what the
// chip holds (launcher: LDS-limited workgroups per CU x CUs) and every workgroup pulls items from a queue: its first
// item is its own id, every further one comes from one device-scope atomicAdd, asked for by thread 0 at the START of
// the item it precedes (the round trip hides behind the transforms) and published to the others through the top word
// of the LDS buffer, which is free once the run has been staged.  Items are (column, pair) in column-major order, so a
// workgroup's consecutive items are neighbours in memory more often than not.  The last workgroup out resets the queue
// (the next launch on the same HIP stream finds it zeroed; queues are per stream: launch_fft).
template <typename Spec, typename Real, typename IO = Real>
__global__ void __launch_bounds__(Spec::NT) k_fft_pair2p(FftArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr size_t LDS = (size_t)(Spec::NA > Spec::NB ? Spec::NA : Spec::NB) * sizeof(typename PairTabs<Real>::C);
    uint32_t *top = reinterpret_cast<uint32_t *>(smem_raw) + (LDS / 4 - 2);
    const uint32_t n_items = a.n_items;
    uint32_t item = blockIdx.x;
    // Phase stagger (HIPSOXR_DEBUG_STAGGER, cycles): resident workgroups all start at once and, with items of equal
    // cost, stay in step for the whole launch — every workgroup of a CU loads at the same time, then computes at the same
    // time.  Workgroup b waits (b / CUs) * stagger cycles once, so that a CU's slots run a fraction of an item apart.
    if (a.stagger) {
        const long long until = __builtin_amdgcn_s_memtime() + (long long)(blockIdx.x >> 8) * a.stagger;
        while ((long long)__builtin_amdgcn_s_memtime() < until) __builtin_amdgcn_s_sleep(32);
    }
    while (item < n_items) {
        uint32_t nxt = 0;
        if (threadIdx.x == 0) nxt = gridDim.x + atomicAdd(a.queue, 1u);
        // (the arguments through an opaque pointer to the kernel-argument segment: read inside the loop, they are
        //  forty scalar loads per item; hoisted out of it, forty SGPRs held across it — 101 in all, and at 97-112 SGPRs
        //  the hardware admits one workgroup per CU fewer than the occupancy query answers, MI355X guide)
        typedef const FftArgs __attribute__((address_space(4))) *KArgs; // (typed as constant memory: scalar loads)
        KArgs kc = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kc));
        const FftArgs *ka = (const FftArgs *)kc;
        const uint32_t ppc = (uint32_t)ka->pairs_per_col;
        const uint32_t col = __builtin_amdgcn_readfirstlane(item / ppc), bx = __builtin_amdgcn_readfirstlane(item % ppc);
        const bool did = pair2_item<Spec, Real, IO>(*ka, smem_raw, col, (int64_t)bx, [top, nxt] { if (threadIdx.x == 0) *top = nxt; });
        if (!did) { // (uniform: an item beyond its clip's last pair)
            if (threadIdx.x == 0) *top = nxt;
            __syncthreads();
        }
        item = __builtin_amdgcn_readfirstlane(*top);
        __syncthreads(); // the staged run and the top word have been read: the next item's first pass may store
    }
    if (threadIdx.x == 0 && atomicAdd(a.queue + 1, 1u) == gridDim.x - 1) { // last one out
        a.queue[1] = 0;
        __threadfence();
        a.queue[0] = 0;
    }
}

// ---------------------------------------------------------------------------------------------
// Paired kernel, second generation, CHANNEL-PAIR mode: interleaved data with an even channel count.
// The two real signals of a transform are the same block of two neighbouring channels — one aligned (Real, Real)
// word per frame — and the workgroup ids are XCD-aware, exactly as in k_fft_pair (see there).  What differs is how
// HBM is touched, as in k_fft_pair2: one raw buffer load per element (8 or 16 bytes; descriptor over [first frame
// of the block, end of the column), hardware range check instead of per-element bounds code, the per-butterfly
// offset t*N/R0*frame in the instruction's scalar operand) and one range-checked buffer store per kept output.
// The first-generation kernel spent a third of its time on the address arithmetic and bounds code of these loads:
// configs[2] (8 channels) 61.4 -> see DESIGN.md §6; the same 61 us at 2 channels, where every byte of every line
// fetched is used, which is what ruled the data layout out as the cause.
// ---------------------------------------------------------------------------------------------
template <typename Real> struct CpIo;
template <> struct CpIo<float> {
    static __device__ __forceinline__ float2 load(__amdgpu_buffer_rsrc_t r, int voff, int soff)
    {
        return __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
    }
    static __device__ __forceinline__ void store(float2 v, __amdgpu_buffer_rsrc_t r, int voff)
    {
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u_t, v), r, voff, 0, 0);
    }
};
template <> struct CpIo<double> {
    static __device__ __forceinline__ double2 load(__amdgpu_buffer_rsrc_t r, int voff, int soff)
    {
        return __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    }
    static __device__ __forceinline__ void store(double2 v, __amdgpu_buffer_rsrc_t r, int voff)
    {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u_t, v), r, voff, 0, 0);
    }
};

// CP = true: channel pairs (above).  CP = false: strided columns that cannot be paired by channel (odd channel counts
// of interleaved data, a channel slice with a frame stride): two consecutive blocks of ONE column are paired, as in
// k_fft_pair2, each element a 4/8-byte buffer load or store at the column's frame stride.
// K > 0 (round 3, channel pairs): a workgroup WALKS a.walk consecutive blocks of its channel pair and keeps the K
// butterfly inputs per thread that the next block shares with this one in registers.  Blocks overlap by N - hop input
// frames (882 of 4410 at 44.1k -> 16k: every block re-read 25 % of its input, 1.19x the algorithmic traffic for the
// whole job); when the first-pass butterfly stride N/R0 divides the hop (4410 = 15 * 294, hop 3528 = 12 * 294) the
// next block's inputs t = 0 .. K-1 of thread j ARE this block's inputs R0-K .. R0-1 of the same thread, so the walk
// costs 2 K registers and no LDS.  The launcher checks the geometry; K = 0 is the plain kernel.
template <typename Spec, typename Real, bool CP, int K = 0>
__global__ void __launch_bounds__(Spec::NT) k_fft_strided2(FftArgs a)
{
    typedef typename PairTabs<Real>::C C;
    constexpr int ES = (int)sizeof(Real);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    C *cur = reinterpret_cast<C *>(smem_raw);
    constexpr int NA = Spec::NA, NB = Spec::NB, R0 = Spec::RA0, nbA = NA / R0;
    static_assert(K == 0 || (CP && K < R0), "walking: channel-pair mode");
#ifdef FFT2_TRACE
    unsigned long long *g_tr = nullptr;
    int g_tri = 0;
#endif
    // XCD-aware ids (k_fft_pair): x = 8 * slot + xcd, slot = chunk * units + unit, item = 8 * chunk + xcd;
    // or (a.xcd_map == 0: one column per grid row) items along x, columns along y
    const uint32_t units = CP ? a.n_channels / 2 : a.n_channels;
    const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const bool xm = a.xcd_map != 0;
    const uint32_t cu = __builtin_amdgcn_readfirstlane(xm ? slot % units : blockIdx.y % units);
    const uint32_t clip = __builtin_amdgcn_readfirstlane(xm ? blockIdx.y : blockIdx.y / units);
    const int64_t item = (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane(xm ? (slot / units) * 8 + xcd : blockIdx.x);
    if (item >= a.pairs_per_col) return; // grid.x is padded to a multiple of 8 items per unit
    const uint32_t ch = CP ? 2 * cu : cu;
    const int32_t hop_in = (int32_t)(a.hop_periods * a.M), hop_out = a.hop_out;
    const Real *xin = (const Real *)a.in + (int64_t)clip * a.ics + (int64_t)ch * a.ichs;
    const int32_t ifb = (int32_t)a.ifs * ES, ofb = (int32_t)a.ofs * ES; // bytes per frame (launcher: 2 N * frame < 2^30)
    auto lds_store = [&](int n, C v) { cur[n] = v; };
    typename Spec::Tw tw;
    const Real *Hr = PairTabs<Real>::hr(a);
    const int32_t v0 = a.v0, v1 = a.v0 + hop_out;
    const int walk = K > 0 ? a.walk : 1;
    C keep[K > 0 ? K : 1]; // this block's last K first-pass inputs = the next block's first K
    for (int w = 0; w < walk; ++w) {
    const int64_t bx = K > 0 ? item * walk + w : item;
    if (K > 0 && bx >= a.n_blocks_col) break;
    const int64_t pa = (CP ? 1 : 2) * bx * a.hop_periods - a.lead_periods; // first period of the (first) block
    const int64_t ina = pa * a.M, outa = pa * a.L;
    C nxt[K > 0 ? K : 1];

    // ---- forward: z[n] = x_c[n] + i x_{c+1}[n]  (CP)  or  x_a[n] + i x_b[n]  (two blocks), first pass straight from HBM
    if (ina >= 0) {
        const int64_t left = (a.in_frames - ina) * (int64_t)ifb; // bytes from the block's first frame to the end of the column
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            uniform_ptr((void *)(xin + ina * a.ifs)), 0, __builtin_amdgcn_readfirstlane((int)(left < 0 ? 0 : left > 0x40000000 ? 0x40000000 : left)), 0x00020000);
        const int32_t stepb = nbA * ifb; // one butterfly input further: N/R0 frames
        Spec::fwd(FFT_STAMP_ARGS cur, PairTabs<Real>::wa(a), [&](int n, int t) -> C {
            if constexpr (CP) {
                C v;
                if (K > 0 && t < K && w > 0) v = keep[t < K ? t : 0];
                else v = CpIo<Real>::load(rs, (n - t * nbA) * ifb, t * stepb);
                if (K > 0 && t >= R0 - K) nxt[t >= R0 - K ? t - (R0 - K) : 0] = v;
                return v;
            }
            else return C(buf_load_real<Real>(rs, (n - t * nbA) * ifb, t * stepb), buf_load_real<Real>(rs, (n - t * nbA) * ifb, t * stepb + hop_in * ifb));
        }, lds_store, false, tw);
    } else { // the first block of a column reaches before its start: explicit zero-extension
        Spec::fwd(FFT_STAMP_ARGS cur, PairTabs<Real>::wa(a), [&](int n, int t) -> C {
            const int64_t l = ina + n, lb = l + hop_in;
            if constexpr (CP) {
                C v = C((Real)0, (Real)0);
                if (l >= 0 && l < a.in_frames) v = C(xin[l * a.ifs], xin[l * a.ifs + 1]);
                if (K > 0 && t >= R0 - K) nxt[t >= R0 - K ? t - (R0 - K) : 0] = v;
                return v;
            } else {
                return C((l >= 0 && l < a.in_frames) ? xin[l * a.ifs] : (Real)0, (lb >= 0 && lb < a.in_frames) ? xin[lb * a.ifs] : (Real)0);
            }
        }, lds_store, false, tw);
    }
    __syncthreads();

    // ---- inverse (see k_fft_pair2), outputs straight to HBM -----------------------------------------
    Real *ybase = (Real *)a.out + (int64_t)clip * a.ocs + (int64_t)ch * a.ochs + (outa + v0) * a.ofs; // outa + v0 >= 0
    const int64_t oleft = (a.out_frames - (outa + v0)) * (int64_t)ofb;
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
        uniform_ptr((void *)ybase), 0, __builtin_amdgcn_readfirstlane((int)(oleft < 0 ? 0 : oleft > 0x40000000 ? 0x40000000 : oleft)), 0x00020000);
    auto h_load = [&](int n, int) -> C {
        const bool neg = n > NB / 2;
        const int q = neg ? NB - n : n; // |frequency| in bins
        const Real h = Hr[q];
        if constexpr (NA >= NB) {
            const C x = cur[neg ? n + (NA - NB) : n];
            return C(x.x * h, x.y * h);
        } else {
            const bool in_band = q < NA / 2;
            const C x = cur[in_band ? (neg ? NA - q : q) : 0];
            return in_band ? C(x.x * h, x.y * h) : C((Real)0, (Real)0);
        }
    };
    Spec::inv(FFT_STAMP_ARGS cur, PairTabs<Real>::wb(a), h_load, [&](int n, C wv) {
        if (n >= v0 && n < v1) {
            if constexpr (CP) {
                CpIo<Real>::store(wv, ro, (n - v0) * ofb); // frame outa + n holds (y_c, y_{c+1})
            } else {
                buf_store_real(wv.x, ro, (n - v0) * ofb);             // block a
                buf_store_real(wv.y, ro, (n - v0 + hop_out) * ofb);   // block b: hop_out frames further
            }
        }
    }, true, tw);
    if constexpr (K > 0) {
#pragma unroll
        for (int i = 0; i < K; ++i) keep[i] = nxt[i];
        __syncthreads(); // the last inverse pass has read the buffer: the next block's first pass may store into it
    }
    } // walk
}

// ---------------------------------------------------------------------------------------------
// Three translation units.  Every schedule of the table below is 7 kernels (k_fft_pair2 in float32, float64 and
// float32-on-float64, k_fft_strided2 x 2 in float32 and float64); compiled in one piece they are the build's critical
// path (5 minutes).  build.sh compiles this file three times: -DFFT_PART=0 = everything except the kernels of the
// schedules listed here (declared extern), -DFFT_PART=1 / =2 = the templates above plus exactly the kernels of one of
// the two lists, no host code.  Without FFT_PART: one piece.
// ---------------------------------------------------------------------------------------------
#define HIPSOXR_PART1_SPECS(X) X(4096, 2048, 256) X(2048, 4096, 256) X(2048, 1024, 256) X(1024, 2048, 256) X(5376, 1792, 384) X(1792, 5376, 384) X(5376, 3584, 384) X(3584, 5376, 384) X(2688, 896, 384) X(896, 2688, 384) X(2688, 1792, 384) X(1792, 2688, 384) X(5120, 1280, 320) X(1280, 5120, 320) X(5376, 896, 384) X(896, 5376, 384)
#define HIPSOXR_PART2_SPECS(X) X(7056, 5120, 448) X(5120, 7056, 448) X(4704, 2560, 384) X(2560, 4704, 384) X(5120, 2352, 384) X(2352, 5120, 384) X(7056, 1280, 448) X(1280, 7056, 448) X(5120, 1176, 320) X(1176, 5120, 320) X(3528, 5120, 384) X(5120, 3528, 384) X(4704, 1280, 384) X(1280, 4704, 384) X(3840, 5120, 384) X(5120, 3840, 384)
#define HIPSOXR_INST(NA, NB, NT)                                                                        \
    HIPSOXR_EXTERN template __global__ void k_fft_pair2<PairOf<NA, NB, NT>, float>(FftArgs);             \
    HIPSOXR_EXTERN template __global__ void k_fft_pair2<PairOf<NA, NB, NT>, double>(FftArgs);            \
    HIPSOXR_EXTERN template __global__ void k_fft_pair2<PairOf<NA, NB, NT>, double, float>(FftArgs);     \
    HIPSOXR_EXTERN template __global__ void k_fft_strided2<PairOf<NA, NB, NT>, float, true>(FftArgs);    \
    HIPSOXR_EXTERN template __global__ void k_fft_strided2<PairOf<NA, NB, NT>, double, true>(FftArgs);   \
    HIPSOXR_EXTERN template __global__ void k_fft_strided2<PairOf<NA, NB, NT>, float, false>(FftArgs);   \
    HIPSOXR_EXTERN template __global__ void k_fft_strided2<PairOf<NA, NB, NT>, double, false>(FftArgs);
#if defined(FFT_PART) && FFT_PART == 0
#define HIPSOXR_EXTERN extern
HIPSOXR_PART1_SPECS(HIPSOXR_INST)
HIPSOXR_PART2_SPECS(HIPSOXR_INST)
#elif defined(FFT_PART) && FFT_PART == 1
#define HIPSOXR_EXTERN
HIPSOXR_PART1_SPECS(HIPSOXR_INST)
#elif defined(FFT_PART) && FFT_PART == 2
#define HIPSOXR_EXTERN
HIPSOXR_PART2_SPECS(HIPSOXR_INST)
#endif

#if !defined(FFT_PART) || FFT_PART == 0
// ---------------------------------------------------------------------------------------------
// host: geometry, tables
// ---------------------------------------------------------------------------------------------
struct FftGeom {
    bool ok = false;
    int k = 0;
    int32_t N_in = 0, N_out = 0, A = 0, B = 0;
    int32_t radA[8] = {1, 1, 1, 1, 1, 1, 1, 1}, radB[8] = {1, 1, 1, 1, 1, 1, 1, 1}, nA = 0, nB = 0; // (plain arrays: the cached geometry is copied per launch)
    int32_t lead_periods = 0, hop_periods = 0, v0 = 0, hop_out = 0;
    size_t lds_bytes = 0;
    float2 *dev = nullptr; // [WA: A][WB: B][P: A+1][Q: B][Hs: B+1][WA2: N_in][WB2: N_out][Hr: B+1 floats]
    double2 *devd = nullptr; // float64 instance of the paired kernel: [WA2d: N_in][WB2d: N_out][Hrd: B+1 doubles]
};

static bool factor_radices(int n, std::vector<int> &rad)
{
    rad.clear();
    int twos = 0;
    while (n % 2 == 0) { n /= 2; ++twos; }
    for (int pr : {7, 5, 3})
        while (n % pr == 0) { n /= pr; rad.push_back(pr); }
    if (n != 1) return false;
    while (twos >= 4) { rad.push_back(16); twos -= 4; }
    if (twos == 3) rad.push_back(8);
    else if (twos == 2) rad.push_back(4);
    else if (twos == 1) rad.push_back(2);
    // small radices first keeps the early (small-Ns) passes cheap in LDS bank conflicts
    std::sort(rad.begin(), rad.end());
    return rad.size() <= 8 && !rad.empty();
}

static std::mutex g_fft_mu;
static std::vector<std::pair<std::pair<const Plan *, int>, FftGeom>> g_fft; // key: (plan, geometry variant)

// Work queues of the resident-workgroup kernel: two zeroed words per (device, HIP stream).  Launches on one stream run
// one after the other and the last workgroup of each re-zeroes the words, so one queue per stream is enough; launches
// on different streams may overlap and never share one.  (A stream handle recycled by the runtime finds its words
// zeroed.)  Never freed: 8 bytes per stream the process has launched on.
static std::mutex g_queue_mu;
static std::vector<std::pair<std::pair<int, void *>, uint32_t *>> g_queues;
static const char *fft_queue_for(void *stream, uint32_t **out)
{
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_queue_mu);
    for (auto &e : g_queues)
        if (e.first.first == dev && e.first.second == stream) { *out = e.second; return nullptr; }
    uint32_t *q = nullptr;
    HIP_TRY(hipMalloc((void **)&q, 16));
    HIP_TRY(hipMemset(q, 0, 16));
    g_queues.push_back({{dev, stream}, q});
    *out = q;
    return nullptr;
}

void fft_release(const Plan *p)
{
    std::lock_guard<std::mutex> lk(g_fft_mu);
    for (size_t i = 0; i < g_fft.size();)
        if (g_fft[i].first.first == p) {
            if (g_fft[i].second.dev) (void)hipFree(g_fft[i].second.dev);
            if (g_fft[i].second.devd) (void)hipFree(g_fft[i].second.devd);
            g_fft.erase(g_fft.begin() + i);
        } else ++i;
}

static const char *fft_build(const Plan &p, FftGeom *out, bool small, int force_k = 0)
{
    FftGeom g;
    const int64_t L = p.L, M = p.M;
    const int32_t T = p.T;
    if (p.q.bits == 0.) { *out = g; return nullptr; } // QQ: not worth a transform
    // block of k periods: smallest power-of-two k with <= ~15 % overlap whose two half-lengths are
    // 7-smooth, even, and fit LDS
    // candidates: power-of-two k with 7-smooth even half-lengths; take the largest block whose
    // transforms stay <= 2600 points (one 20 KB LDS buffer, least overlap waste), else the
    // smallest admissible one
    for (int k = force_k ? force_k : 1; k <= (force_k ? force_k : 4096); k *= 2) {
        const int64_t Nin = M * k, Nout = L * k;
        if (Nin % 2 || Nout % 2) continue;
        if (!force_k && Nin < 6 * (int64_t)T) continue;
        if (Nin / 2 > 4096 || Nout / 2 > 4096) break;
        std::vector<int> ra, rb;
        if (!factor_radices((int)(Nin / 2), ra) || !factor_radices((int)(Nout / 2), rb)) continue;
        if (!force_k && g.k && (small || std::max(Nin, Nout) / 2 > 2600)) break;
        g.k = k; g.N_in = (int32_t)Nin; g.N_out = (int32_t)Nout; g.A = g.N_in / 2; g.B = g.N_out / 2;
        g.nA = (int32_t)ra.size(); g.nB = (int32_t)rb.size();
        for (int i = 0; i < 8; ++i) { g.radA[i] = i < g.nA ? ra[i] : 1; g.radB[i] = i < g.nB ? rb[i] : 1; }
    }
    if (!g.k) { *out = g; return nullptr; }
    // outputs whose filter support [n_k, n_k + T) lies inside the block: discard ceil((T/2+2)*L/M)
    // outputs at either end, keep a whole number of periods
    const int64_t disc = ((int64_t)(T / 2 + 2) * L + M - 1) / M;
    g.lead_periods = (int32_t)((disc + L - 1) / L);
    g.hop_periods = (int32_t)((g.N_out - disc - (int64_t)g.lead_periods * L) / L);
    if (g.hop_periods < 1) { *out = g; return nullptr; }
    g.v0 = (int32_t)(g.lead_periods * L);
    g.hop_out = (int32_t)(g.hop_periods * L);
    g.lds_bytes = (size_t)(std::max(g.A, g.B) + 8) * sizeof(float2);
    if (g.lds_bytes > 150 * 1024) { *out = g; return nullptr; }
    // a block must keep a worthwhile share of its outputs (long filters on short blocks do not)
    if (force_k && 2 * (int64_t)g.hop_out < g.N_out) { *out = g; return nullptr; }

    const int A = g.A, B = g.B;
    std::vector<float2> tab((size_t)A + B + (A + 1) + B + (B + 1) + g.N_in + g.N_out + (B + 2) / 2 + 1 + (g.N_out + g.N_in) / 2 + 2 /* HP (-DFFT_DIF) */);
    float2 *WA = tab.data(), *WB = WA + A, *P = WB + B, *Q = P + (A + 1), *Hs = Q + B;
    float2 *WA2 = Hs + (B + 1), *WB2 = WA2 + g.N_in;
    for (int m = 0; m < g.N_in; ++m) WA2[m] = make_float2((float)std::cos(6.283185307179586476925286766559 * m / g.N_in), (float)-std::sin(6.283185307179586476925286766559 * m / g.N_in));
    for (int m = 0; m < g.N_out; ++m) WB2[m] = make_float2((float)std::cos(6.283185307179586476925286766559 * m / g.N_out), (float)std::sin(6.283185307179586476925286766559 * m / g.N_out));
    const double PI2 = 6.283185307179586476925286766559;
    for (int m = 0; m < A; ++m) WA[m] = make_float2((float)std::cos(PI2 * m / A), (float)-std::sin(PI2 * m / A));
    for (int m = 0; m < B; ++m) WB[m] = make_float2((float)std::cos(PI2 * m / B), (float)std::sin(PI2 * m / B));
    for (int q = 0; q <= A; ++q) P[q] = make_float2((float)std::cos(PI2 * q / g.N_in), (float)-std::sin(PI2 * q / g.N_in));
    for (int q = 0; q < B; ++q) Q[q] = make_float2((float)std::cos(PI2 * q / g.N_out), (float)std::sin(PI2 * q / g.N_out));
    // H[q] = sum_p sum_j bank[p][j] exp(-2 pi i q (L*(T/2-1-j) + p) / (L*N_in)), scaled by 1/(N_in*L)
    const double scale = 1.0 / ((double)g.N_in * (double)L);
    const int qmax = std::min(A, B);
    std::vector<double> hr64((size_t)B + 1, 0.); // Re H in float64 (float64 instance of the paired kernel)
    for (int q = 0; q <= B; ++q) {
        if (q > qmax) { Hs[q] = make_float2(0.f, 0.f); continue; }
        double hr = 0., hi = 0.;
        const double wj = PI2 * (double)q / (double)g.N_in; // per tap j the angle grows by +wj
        const double cwj = std::cos(wj), swj = std::sin(wj);
        for (int64_t ph = 0; ph < L; ++ph) {
            // angle for j = 0: -2 pi q (L*(T/2-1) + ph) / (L*N_in)
            const double a0 = -PI2 * (double)q * ((double)(L * (int64_t)(T / 2 - 1) + ph)) / ((double)L * g.N_in);
            double cr = std::cos(a0), ci = std::sin(a0);
            const double *b = p.bank.data() + (size_t)(ph * T);
            for (int j = 0; j < T; ++j) {
                hr += b[j] * cr; hi += b[j] * ci;
                const double nr = cr * cwj - ci * swj; ci = cr * swj + ci * cwj; cr = nr;
            }
        }
        Hs[q] = make_float2((float)(hr * scale), (float)(hi * scale));
        // The prototype is symmetric about the output instant (zero latency) and blocks are cut on period
        // boundaries, so H is real: |Im H| <= 2e-13 |Re H| in the pass band (the one unpaired sample of the
        // even-length support, g[-L T/2], is a window-edge value ~1e-11).  The newer paired kernels use
        // Re H alone: half the table reads and a real x complex product per bin.
        reinterpret_cast<float *>(WB2 + g.N_out)[q] = (float)(hr * scale);
        hr64[q] = hr * scale;
    }
#ifdef FFT_DIF
    { // k_fft_pair2's wave-local schedule: per output-grid bin n the filter value and the LDS byte offset of the input-grid
      // bin it takes, in the layout the forward transform leaves (dif_local: rows rotated by the sub-transform index)
        int r0 = 0, r1 = 0, r2 = 0;
        if (sched_of(g.N_in, &r0, &r1, &r2)) {
            uint32_t *hpo = reinterpret_cast<uint32_t *>(WB2 + g.N_out + (B + 2) / 2 + 1);
            float *hf = reinterpret_cast<float *>(hpo) + g.N_out;
            const float *hrf = reinterpret_cast<const float *>(WB2 + g.N_out);
            const int NA = g.N_in, NB = g.N_out;
            auto nat = [&](int k) { return ((k % r0) * r1 + (k / r0) % r1) * r2 + k / (r0 * r1); }; // (k0, k1, k2) order: what a lane of the last forward pass holds
            for (int k = 0; k < NA; ++k) hf[nat(k)] = 0.f;
            for (int n = 0; n < NB && NA >= NB; ++n) {
                const bool neg = n > NB / 2;
                const int q = neg ? NB - n : n, k = neg ? n + (NA - NB) : n;
                const int k0 = k % r0, k1 = (k / r0) % r1, k2 = k / (r0 * r1);
                hf[nat(k)] = hrf[q];
                const int row = (k1 + k0) % r1, sx = (r2 % 4 == 0) ? (row >> 3) & 1 : 0;
                hpo[n] = (uint32_t)((k0 * r1 * r2 + row * r2 + (k2 ^ sx)) * sizeof(float2));
            }
        }
    }
#endif
    HIP_TRY(hipMalloc((void **)&g.dev, tab.size() * sizeof(float2)));
    HIP_TRY(hipMemcpy(g.dev, tab.data(), tab.size() * sizeof(float2), hipMemcpyHostToDevice));
    { // the float64 instance's tables (small: N_in + N_out + B/2 double2)
        std::vector<double2> td((size_t)g.N_in + g.N_out + (B + 2) / 2 + 1);
        for (int m = 0; m < g.N_in; ++m) td[m] = make_double2(std::cos(PI2 * m / g.N_in), -std::sin(PI2 * m / g.N_in));
        for (int m = 0; m < g.N_out; ++m) td[(size_t)g.N_in + m] = make_double2(std::cos(PI2 * m / g.N_out), std::sin(PI2 * m / g.N_out));
        double *hrd = reinterpret_cast<double *>(td.data() + g.N_in + g.N_out);
        for (int q = 0; q <= B; ++q) hrd[q] = q < (int)hr64.size() ? hr64[q] : 0.;
        HIP_TRY(hipMalloc((void **)&g.devd, td.size() * sizeof(double2)));
        HIP_TRY(hipMemcpy(g.devd, td.data(), td.size() * sizeof(double2), hipMemcpyHostToDevice));
    }
    g.ok = true;
    *out = g;
    return nullptr;
}

// Whole-signal float32 job?  (zero-extended signal starting at absolute index 0, all outputs)
bool fft_job_eligible(const Plan &p, const hipsoxr_job_t &j)
{
    // what the method neglects is the aliasing of the filter's stop band: only recipes whose stop band
    // is far below the 1e-6 bar qualify (HQ 128 dB, VHQ 177 dB; MQ/LQ at 104 dB do not)
    // float64 jobs: the paired kernel has a float64 instance (unit-stride columns, the ratio table below)
    return p.phases == 0 && p.att_db >= 120. && (j.elem == HIPSOXR_F32 || j.elem == HIPSOXR_F64) && j.in_abs0 == 0 &&
           j.out_k0 == 0 && (uint64_t)j.out_frames <= plan_out_len(p, (uint64_t)j.in_frames);
}

// the walking instance of the channel-pair kernel exists where the geometry admits it (k_fft_strided2, K > 0)
template <int NA, int NB, int NT, typename Real> static constexpr void (*walk_kernel())(FftArgs)
{
#ifdef FFT_EXPERIMENTS
    if constexpr (NA == 4410 && NB == 1600) return k_fft_strided2<PairOf<NA, NB, NT>, Real, true, 3>;
#endif
    return nullptr;
}

const char *launch_fft(Plan *p, const hipsoxr_job_t &j, void *stream, bool *handled)
{
    *handled = false;
    // geometry cache key: (plan, variant) with variant 0 = default search, 1 = small-block search,
    // 2 + i = forced k of paired-kernel entry i
    auto get = [&](int variant, int force_k, FftGeom *g) -> const char * {
        std::lock_guard<std::mutex> lk(g_fft_mu);
        for (auto &e : g_fft)
            if (e.first.first == p && e.first.second == variant) { *g = e.second; return nullptr; }
        if (const char *err = fft_build(*p, g, variant == 1, force_k)) return err;
        g_fft.push_back({{p, variant}, *g});
        return nullptr;
    };
    // ---- paired-block kernels: compile-time schedules for the common ratios -------------------
    struct PairEntry {
        int64_t L, M; int k; int small; /* 0: full-size blocks, 1: half-size (small jobs), 2: quarter-size (smaller still) */
        void (*kern)(FftArgs); unsigned nt; void (*kern2)(FftArgs); void (*kern2d)(FftArgs);
        void (*kern2fd)(FftArgs);                    // float32 I/O on float64 arithmetic (HIPSOXR_KERNEL_FFT_F64)
        void (*kern2p)(FftArgs);                     // float32, resident workgroups pulling items from a queue (large jobs)
        void (*kcp)(FftArgs); void (*kcpd)(FftArgs); // channel-pair mode (interleaved data), float32 / float64
        void (*kst)(FftArgs); void (*kstd)(FftArgs); // strided columns, two blocks per transform
        void (*kcpw)(FftArgs); void (*kcpwd)(FftArgs); int walk_k; // channel pairs, walking (k_fft_strided2<.., K>): 44.1k -> 16k only
    };
// (the first-generation kernel is instantiated only where the A/B tools use it — the 44.1k <-> 48k and 44.1k <-> 16k
//  families: HIPSOXR_PAIR_V1; elsewhere a job the second-generation kernels cannot take goes to k_fft_block)
#define HIPSOXR_PAIR_(L, M, k, small, NA, NB, NT, V1, P2P) \
    {L, M, k, small, V1, NT, k_fft_pair2<PairOf<NA, NB, NT>, float>, k_fft_pair2<PairOf<NA, NB, NT>, double>, \
     k_fft_pair2<PairOf<NA, NB, NT>, double, float>, P2P, \
     k_fft_strided2<PairOf<NA, NB, NT>, float, true>, k_fft_strided2<PairOf<NA, NB, NT>, double, true>, \
     k_fft_strided2<PairOf<NA, NB, NT>, float, false>, k_fft_strided2<PairOf<NA, NB, NT>, double, false>, \
     walk_kernel<NA, NB, NT, float>(), walk_kernel<NA, NB, NT, double>(), (NA == 4410 && NB == 1600) ? 3 : 0}
// (the first-generation kernel and the resident-workgroup experiment k_fft_pair2p exist for these families only)
#define HIPSOXR_PAIR(L, M, k, small, NA, NB, NT) HIPSOXR_PAIR_(L, M, k, small, NA, NB, NT, nullptr, nullptr)
#ifdef FFT_EXPERIMENTS
#define HIPSOXR_PAIR_V1(L, M, k, small, NA, NB, NT) HIPSOXR_PAIR_(L, M, k, small, NA, NB, NT, (k_fft_pair<PairOf<NA, NB, NT>>), (k_fft_pair2p<PairOf<NA, NB, NT>, float>))
#else
#define HIPSOXR_PAIR_V1(L, M, k, small, NA, NB, NT) HIPSOXR_PAIR_(L, M, k, small, NA, NB, NT, (k_fft_pair<PairOf<NA, NB, NT>>), nullptr)
#endif
    static const PairEntry pairs[] = {
        // L, M (out/in = L/M), periods per block, small-job variant, N_in, N_out, threads
        HIPSOXR_PAIR_V1(147, 160, 32, false, 5120, 4704, 384), HIPSOXR_PAIR_V1(147, 160, 16, true, 2560, 2352, 384),   // 48k -> 44.1k
        HIPSOXR_PAIR_V1(160, 147, 32, false, 4704, 5120, 384), HIPSOXR_PAIR_V1(160, 147, 16, true, 2352, 2560, 384),   // 44.1k -> 48k
        HIPSOXR_PAIR(147, 160, 8, 2, 1280, 1176, 256), HIPSOXR_PAIR(160, 147, 8, 2, 1176, 1280, 256),           // ... quarter-size blocks: jobs of a few hundred pairs
        HIPSOXR_PAIR_V1(160, 441, 16, false, 7056, 2560, 448), HIPSOXR_PAIR_V1(441, 160, 16, false, 2560, 7056, 448),  // 44.1k <-> 16k
        HIPSOXR_PAIR_V1(160, 441, 10, true, 4410, 1600, 320), HIPSOXR_PAIR_V1(441, 160, 10, true, 1600, 4410, 320),    // ... 35 KB blocks: 4 workgroups per CU
        HIPSOXR_PAIR(1, 2, 2048, false, 4096, 2048, 256), HIPSOXR_PAIR(2, 1, 2048, false, 2048, 4096, 256),      // 2:1, 1:2
        HIPSOXR_PAIR(1, 2, 1024, true, 2048, 1024, 256), HIPSOXR_PAIR(2, 1, 1024, true, 1024, 2048, 256),        // ... half-size blocks: small jobs (10 s mono 7.5 -> 6.6 us), float64
        HIPSOXR_PAIR(1, 3, 1792, false, 5376, 1792, 384), HIPSOXR_PAIR(3, 1, 1792, false, 1792, 5376, 384),      // 48k <-> 16k
        HIPSOXR_PAIR(2, 3, 1792, false, 5376, 3584, 384), HIPSOXR_PAIR(3, 2, 1792, false, 3584, 5376, 384),      // 48k <-> 32k
        HIPSOXR_PAIR(1, 3, 896, true, 2688, 896, 384), HIPSOXR_PAIR(3, 1, 896, true, 896, 2688, 384),            // ... half-size blocks for both:
        HIPSOXR_PAIR(2, 3, 896, true, 2688, 1792, 384), HIPSOXR_PAIR(3, 2, 896, true, 1792, 2688, 384),          //     small jobs, float64
        HIPSOXR_PAIR(1, 4, 1280, false, 5120, 1280, 320), HIPSOXR_PAIR(4, 1, 1280, false, 1280, 5120, 320),      // 4:1, 1:4
        HIPSOXR_PAIR(1, 6, 896, false, 5376, 896, 384), HIPSOXR_PAIR(6, 1, 896, false, 896, 5376, 384),          // 48k <-> 8k
        HIPSOXR_PAIR(320, 441, 16, false, 7056, 5120, 448), HIPSOXR_PAIR(441, 320, 16, false, 5120, 7056, 448),  // 44.1k <-> 32k
        HIPSOXR_PAIR(80, 147, 32, false, 4704, 2560, 384), HIPSOXR_PAIR(147, 80, 32, false, 2560, 4704, 384),    // 88.2k <-> 48k
        HIPSOXR_PAIR(147, 320, 16, false, 5120, 2352, 384), HIPSOXR_PAIR(320, 147, 16, false, 2352, 5120, 384),  // 96k <-> 44.1k
        HIPSOXR_PAIR(80, 441, 16, false, 7056, 1280, 448), HIPSOXR_PAIR(441, 80, 16, false, 1280, 7056, 448),    // 44.1k <-> 8k
        HIPSOXR_PAIR(147, 640, 8, false, 5120, 1176, 320), HIPSOXR_PAIR(640, 147, 8, false, 1176, 5120, 320),    // 192k <-> 44.1k
        HIPSOXR_PAIR(640, 441, 8, false, 3528, 5120, 384), HIPSOXR_PAIR(441, 640, 8, false, 5120, 3528, 384),    // 22.05k <-> 32k, 11.025k <-> 16k
        HIPSOXR_PAIR(40, 147, 32, false, 4704, 1280, 384), HIPSOXR_PAIR(147, 40, 32, false, 1280, 4704, 384),    // 44.1k <-> 12k, 88.2k <-> 24k
        HIPSOXR_PAIR(4, 3, 1280, false, 3840, 5120, 384), HIPSOXR_PAIR(3, 4, 1280, false, 5120, 3840, 384),      // 24k <-> 32k, 12k <-> 16k, 48k <-> 64k
    };
#undef HIPSOXR_PAIR
    const bool no_pair = switches().fft_no_pair;
    const uint64_t cols_p = (uint64_t)j.n_clips * j.n_channels;
    // f64: the ARITHMETIC is float64 (block size, LDS bytes per point, table set) — float64 jobs, and float32 jobs that
    // ask for libsoxr's own VHQ width with HIPSOXR_KERNEL_FFT_F64 (io64 = the signal's elements are 8 bytes)
    const bool io64 = j.elem == HIPSOXR_F64, wide32 = !io64 && j.kernel == HIPSOXR_KERNEL_FFT_F64;
    const bool f64 = io64 || wide32;
    // float64: the second-generation kernels only (unit-stride columns; channel pairs; strided columns) — else the exact engine
    if (f64 && (no_pair || cols_p > 65535)) return nullptr;
    if (!no_pair && cols_p <= 65535) {
        const PairEntry *big = nullptr, *sml = nullptr, *tiny = nullptr;
        int big_i = 0, sml_i = 0, tiny_i = 0;
        for (int i = 0; i < (int)(sizeof pairs / sizeof pairs[0]); ++i)
            if (pairs[i].L == p->L && pairs[i].M == p->M) {
                if (pairs[i].small == 2) { tiny = &pairs[i]; tiny_i = i; }
                else if (pairs[i].small) { sml = &pairs[i]; sml_i = i; }
                else { big = &pairs[i]; big_i = i; }
            }
        if (big) {
            FftGeom g;
            if (const char *err = get(2 + big_i, big->k, &g)) return err;
            const PairEntry *use = g.ok ? big : nullptr;
            if (g.ok && sml) {
                // few work items (one 60 s clip = 300 pairs): half-size blocks give twice as many,
                // shorter workgroups, at the price of more overlap
                const int64_t wgs = ((j.out_frames + g.hop_out - 1) / g.hop_out + 1) / 2 * (int64_t)cols_p;
                // (float64: LDS is 16 bytes per point — the half-size blocks keep four workgroups per CU)
                // (7056-point blocks: 56 KB of LDS, two workgroups per CU — the 35 KB blocks of the k = 10 geometry keep
                //  four and win at every size: 44.1k -> 16k VHQ, 8 x 60 s planar 41 vs 68 us, 80 x 60 s 427 vs 638 us)
                const bool big_lds = big->nt > 384;
                if (((wgs < 480 || big_lds) && !switches().fft_large_only) || switches().fft_small_only || f64) { // measured crossover: ~470 pairs of large blocks
                    FftGeom gs;
                    if (const char *err = get(2 + sml_i, sml->k, &gs)) return err;
                    if (gs.ok) { g = gs; use = sml; }
                }
            }
            // Fewer still (a 60 s clip is 612 pairs of half-size blocks on 2048 workgroup slots): the launch is the
            // latency of one workgroup plus what queues behind it; quarter-size blocks (10 KB of LDS, 32 % overlap)
            // shorten both: 2 s clip 7.4 -> 6.5 us, 10 s HQ 7.6 -> 6.2 us, 30 s 9.0 -> 7.7 us (60 s: 10.85 vs 10.73 us).
            if (use == sml && tiny && !switches().fft_large_only && !switches().fft_no_tiny) {
                const int64_t wgs = ((j.out_frames + g.hop_out - 1) / g.hop_out + 1) / 2 * (int64_t)cols_p;
                // (float32: at 612 pairs — the 60 s clip — the two sizes are within 1 %.  float64: 16 bytes per point, and
                //  the 20 KB blocks win at every size — 60 s mono 27.1 -> 21.6 us, 64 x 10 s 236 -> 202 us, stereo 60 s 47 -> 37 us)
                if (f64 || wgs <= 500) {
                    FftGeom gt;
                    if (const char *err = get(2 + tiny_i, tiny->k, &gt)) return err;
                    if (gt.ok) { g = gt; use = tiny; }
                }
            }
            // Round 1: latency-bound jobs (under ~400 workgroups) ran a four-pass radix <= 8 schedule with prefetched
            // tables on the first-generation kernel (7.1 vs 9.0 us for one workgroup).  Against the second-generation
            // three-pass kernel it no longer wins anywhere (0.5 s .. 20 s clips: equal within 0.2 us; 30 s: 10.6 vs
            // 9.4 us): kept behind HIPSOXR_FFT_SMALL_4PASS for A/B only.
            static const PairEntry low_latency = {147, 160, 16, true, k_fft_pair<Pair2560x2352L>, Pair2560x2352L::NT, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
            if (use == sml && sml && sml->kern == (void (*)(FftArgs))k_fft_pair<Pair2560x2352> && switches().fft_small_4pass) {
                const int64_t wgs = ((j.out_frames + g.hop_out - 1) / g.hop_out + 1) / 2 * (int64_t)cols_p;
                if (wgs < 400 && !f64) use = &low_latency;
            }
            if (use) {
                FftArgs a;
                a.in = j.in; a.out = j.out;
                a.WA = g.dev; a.WB = a.WA + g.A; a.P = a.WB + g.B; a.Q = a.P + (g.A + 1); a.Hs = a.Q + g.B;
                a.WA2 = a.Hs + (g.B + 1); a.WB2 = a.WA2 + g.N_in;
                a.Hr = reinterpret_cast<const float *>(a.WB2 + g.N_out); a.HP = a.WB2 + g.N_out + (g.B + 2) / 2 + 1; a.trace = nullptr;
                a.WA2d = g.devd; a.WB2d = g.devd + g.N_in; a.Hrd = reinterpret_cast<const double *>(g.devd + g.N_in + g.N_out);
                a.A = g.A; a.B = g.B; a.nA = a.nB = 0;
                for (int i = 0; i < 8; ++i) a.radA[i] = a.radB[i] = 1;
                a.L = p->L; a.M = p->M;
                a.lead_periods = g.lead_periods; a.hop_periods = g.hop_periods; a.v0 = g.v0; a.hop_out = g.hop_out;
                a.n_clips = j.n_clips; a.n_channels = j.n_channels;
                a.ics = j.in_clip_stride; a.ifs = j.in_frame_stride; a.ichs = j.in_chan_stride;
                a.ocs = j.out_clip_stride; a.ofs = j.out_frame_stride; a.ochs = j.out_chan_stride;
                a.in_frames = j.in_frames; a.out_frames = j.out_frames;
                a.clip_tab = j.clip_table_dev;
                const int64_t n_blocks = (j.out_frames + g.hop_out - 1) / g.hop_out;
                if (n_blocks > 2147483647LL) return "job too long for one launch";
                // interleaved data with an even channel count: pair channels (aligned float2 per frame)
                const size_t esz = io64 ? sizeof(double) : sizeof(float);
                const bool cp_layout = j.n_channels % 2 == 0 && j.in_chan_stride == 1 && j.out_chan_stride == 1 && !switches().fft_no_chpair;
                // (the first-generation kernel reads the pair through a float2 pointer: every frame 8-byte aligned)
                const bool cp_aligned = j.in_frame_stride % 2 == 0 && j.out_frame_stride % 2 == 0 && j.in_clip_stride % 2 == 0 &&
                                        j.out_clip_stride % 2 == 0 && ((uintptr_t)j.in & (2 * esz - 1)) == 0 && ((uintptr_t)j.out & (2 * esz - 1)) == 0;
                // ... the second-generation channel-pair kernel (buffer loads: element alignment is enough) when a block's
                // byte offsets fit its 32-bit operands; float64 has no first-generation kernel and pairs channels
                // through this one or not at all
                const bool cp2 = !wide32 && (f64 ? use->kcpd : use->kcp) != nullptr && !switches().fft_pair_v1 &&
                                 (int64_t)std::max(g.N_in, g.N_out) * std::max(j.in_frame_stride, j.out_frame_stride) * (int64_t)esz < (1LL << 30);
                // (channel pairing rides on the XCD-aware work-item map: decided together, so that a job without the
                //  map — HIPSOXR_FFT_NO_XCD_MAP, or too many work items — runs unpaired on the strided kernel instead of failing)
                const int64_t cp_items8 = (n_blocks + 7) / 8 * 8;
                const bool cp_map_ok = j.n_channels > 1 && j.n_clips <= 65535 && cp_items8 * (int64_t)(j.n_channels / 2) <= 2147483647LL &&
                                       !switches().fft_no_xcd_map;
                a.chpair = (cp_layout && cp_map_ok && (cp2 || (cp_aligned && !f64 && use->kern))) ? 1 : 0;
                const size_t lds = std::max((size_t)std::max(g.N_in, g.N_out) * (f64 ? sizeof(double2) : sizeof(float2)), switches().dbg_fft_lds);
                if (f64 && lds > 160 * 1024) return nullptr;
                if (lds > 64 * 1024 && use->kern)
                    HIP_TRY(hipFuncSetAttribute((const void *)use->kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                // work items per channel unit: blocks (channel pairs) or pairs of blocks (single channels)
                const int64_t items = a.chpair ? n_blocks : (n_blocks + 1) / 2, items8 = (items + 7) / 8 * 8;
                const int64_t units = a.chpair ? j.n_channels / 2 : j.n_channels;
                a.xcd_map = (j.n_channels > 1 && j.in_chan_stride == 1 && j.out_chan_stride == 1 && j.n_clips <= 65535 &&
                             items8 * units <= 2147483647LL && !switches().fft_no_xcd_map) ? 1 : 0;
                if (a.chpair && !a.xcd_map) return "internal: channel pairing needs the XCD map"; // (cannot happen: cp_map_ok above)
                a.pairs_per_col = items;
                const dim3 grid = a.xcd_map ? dim3((unsigned)(items8 * units), j.n_clips, 1)
                                            : dim3((unsigned)((n_blocks + 1) / 2), (unsigned)cols_p, 1);
                // unit-stride columns (mono / planar): the second-generation kernel (buffer loads, staged aligned stores)
                void (*kern)(FftArgs) = use->kern;
                const bool v2ok = use->kern2 && !a.xcd_map && !a.chpair && j.in_frame_stride == 1 && j.out_frame_stride == 1 &&
                                  2 * (size_t)g.hop_out * esz + 16 <= lds;
                const bool cp2ok = a.chpair && a.xcd_map && cp2;
                // strided columns that are not channel pairs (odd channel counts, channel slices): the strided second-
                // generation kernel when the byte offsets of a pair of blocks fit its 32-bit operands
                const bool st2ok = !wide32 && !a.chpair && !v2ok && (f64 ? use->kstd : use->kst) != nullptr && !switches().fft_pair_v1 &&
                                   2 * (int64_t)std::max(g.N_in, g.N_out) * std::max(j.in_frame_stride, j.out_frame_stride) * (int64_t)esz < (1LL << 30);
                if (f64 && !v2ok && !cp2ok && !st2ok) return nullptr; // (no float64 instance of the first-generation kernel: exact engine)
                a.walk = 1; a.n_blocks_col = n_blocks;
                unsigned launch_grid_x = 0; // (non-zero: the walking kernel's own item count)
                if (cp2ok || st2ok) {
                    kern = cp2ok ? (f64 ? use->kcpd : use->kcp) : (f64 ? use->kstd : use->kst);
                    // HIPSOXR_DEBUG_WALK=W (experiment, measured in round 3 and NOT the default): every workgroup of a channel
                    // pair walks W consecutive blocks and keeps the shared input in registers (k_fft_strided2, K > 0): HBM reads
                    // drop by the re-read share of the overlap, and configs[2] gets SLOWER — 47.0 us plain, 53.7 us at W = 3
                    // (50.1 at 4, 57 at 2 and 6; same box): 81 instead of 61 VGPRs (five waves per SIMD instead of six) and a third
                    // of the workgroups, three times as long.  The launch is not bound by HBM traffic.  DESIGN.md §5.2.
                    void (*kw)(FftArgs) = f64 ? use->kcpwd : use->kcpw;
                    const int radA0 = g.N_in == 4410 ? 15 : 0; // (first radix of the forward schedule the walking instance was built on)
                    int want = switches().dbg_walk ? switches().dbg_walk : 1;
                    if (cp2ok && kw && want > 1 && radA0 && (int64_t)use->walk_k * (g.N_in / radA0) == (int64_t)g.N_in - (int64_t)g.hop_periods * p->M) {
                        kern = kw;
                        a.walk = want;
                        const int64_t witems = (n_blocks + want - 1) / want, witems8 = (witems + 7) / 8 * 8;
                        a.pairs_per_col = witems;
                        launch_grid_x = (unsigned)(witems8 * units);
                    }
                    if (lds > 64 * 1024)
                        HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                }
                if (wide32 && !v2ok) return nullptr; // (float32 on float64 arithmetic: unit-stride columns only)
                dim3 launch_grid = grid;
                if (launch_grid_x) launch_grid.x = launch_grid_x;
                a.queue = nullptr; a.n_items = 0; a.stagger = switches().dbg_stagger;
                if (v2ok && (f64 || !switches().fft_pair_v1)) {
                    kern = io64 ? use->kern2d : wide32 ? use->kern2fd : use->kern2;
                    if (lds > 64 * 1024)
                        HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    // HIPSOXR_FFT_PERSIST (experiment, measured in round 3 and NOT the default): resident workgroups pulling
                    // items from a queue (k_fft_pair2p).  Four workgroups per CU for the whole launch instead of the 3.1 the
                    // dispatcher sustains — and 7 % SLOWER on the batch (140.5 vs 130.8 us per launch, same box): a CU's
                    // throughput does not grow with its fourth resident workgroup.  DESIGN.md §5.2.
                    const int64_t n_items = (int64_t)grid.x * grid.y;
                    if (!f64 && use->kern2p && switches().fft_persist && !switches().dbg_trace && n_items <= 0x7fffffffLL &&
                        (2 * (size_t)g.hop_out + 4) * esz + 24 <= lds) {
                        static std::mutex occ_mu;
                        static std::vector<std::pair<const void *, int>> occ_cache; // (kernel, workgroups per CU) — per process, one device type
                        int per_cu = 0, cus = 0, dev = 0;
                        {
                            std::lock_guard<std::mutex> lk(occ_mu);
                            for (auto &e : occ_cache) if (e.first == (const void *)use->kern2p) per_cu = e.second;
                            if (!per_cu) {
                                if (lds > 64 * 1024)
                                    HIP_TRY(hipFuncSetAttribute((const void *)use->kern2p, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                                HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)use->kern2p, (int)use->nt, lds));
                                if (per_cu < 1) per_cu = 1;
                                occ_cache.push_back({(const void *)use->kern2p, per_cu});
                            }
                        }
                        HIP_TRY(hipGetDevice(&dev));
                        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
                        const int64_t slots = (int64_t)per_cu * cus;
                        if (n_items >= 1) {
                            if (const char *err = fft_queue_for(stream, &a.queue)) return err;
                            a.n_items = (uint32_t)n_items;
                            a.pairs_per_col = grid.x;
                            launch_grid = dim3((unsigned)std::min<int64_t>(n_items, slots), 1, 1);
                            kern = use->kern2p;
                        }
                    }
                }
                // ragged batches: the unit-stride second-generation kernel reads its clip's row; nothing else does
                if (j.clip_table && !(v2ok && (f64 || !switches().fft_pair_v1))) return nullptr;
                if (!kern) return nullptr; // (no first-generation instance of this schedule: the general path takes the job)
#ifdef FFT2_TRACE
                size_t trace_n = 0;
                if (switches().dbg_trace && (kern == use->kern2 || kern == use->kern2d)) {
                    trace_n = (size_t)grid.x * grid.y * (use->nt / 64) * 16;
                    HIP_TRY(hipMalloc((void **)&a.trace, trace_n * 8));
                    HIP_TRY(hipMemset(a.trace, 0, trace_n * 8));
                }
#endif
                hipLaunchKernelGGL(kern, launch_grid, dim3(use->nt), lds, (hipStream_t)stream, a);
                HIP_TRY(hipGetLastError());
#ifdef FFT2_TRACE
                if (a.trace) { // debugging aid only: synchronous dump of the per-wave time stamps
                    std::vector<unsigned long long> h(trace_n);
                    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
                    HIP_TRY(hipMemcpy(h.data(), a.trace, trace_n * 8, hipMemcpyDeviceToHost));
                    if (FILE *f = fopen(switches().dbg_trace, "wb")) { fwrite(h.data(), 8, trace_n, f); fclose(f); }
                    (void)hipFree(a.trace);
                }
#endif
                *handled = true;
                return nullptr;
            }
        }
    }
    // ---- general path: one block per workgroup ---------------------------------------------------
    if (f64 || j.clip_table) return nullptr; // float32 only, no ragged batches
    FftGeom g;
    if (const char *err = get(0, 0, &g)) return err;
    if (g.ok) {
        const int64_t wgs = ((j.out_frames + g.hop_out - 1) / g.hop_out) * (int64_t)j.n_clips * j.n_channels;
        if ((wgs < 8 * 256 && !switches().fft_large_only) || switches().fft_small_only) {
            FftGeom gs;
            if (const char *err = get(1, 0, &gs)) return err;
            if (gs.ok && gs.k < g.k) g = gs;
        }
    }
    if (!g.ok) return nullptr;
    FftArgs a;
    a.in = j.in; a.out = j.out;
    a.WA = g.dev; a.WB = a.WA + g.A; a.P = a.WB + g.B; a.Q = a.P + (g.A + 1); a.Hs = a.Q + g.B;
    a.WA2 = a.Hs + (g.B + 1); a.WB2 = a.WA2 + g.N_in;
    a.Hr = reinterpret_cast<const float *>(a.WB2 + g.N_out); a.HP = a.WB2 + g.N_out + (g.B + 2) / 2 + 1; a.trace = nullptr;
    a.WA2d = a.WB2d = nullptr; a.Hrd = nullptr; a.clip_tab = nullptr; a.queue = nullptr; a.n_items = 0; a.stagger = 0;
    a.walk = 1; a.n_blocks_col = 0;
    a.A = g.A; a.B = g.B; a.nA = g.nA; a.nB = g.nB;
    for (int i = 0; i < 8; ++i) { a.radA[i] = g.radA[i]; a.radB[i] = g.radB[i]; }
    a.L = p->L; a.M = p->M;
    a.lead_periods = g.lead_periods; a.hop_periods = g.hop_periods; a.v0 = g.v0; a.hop_out = g.hop_out;
    a.n_clips = j.n_clips; a.n_channels = j.n_channels;
    a.ics = j.in_clip_stride; a.ifs = j.in_frame_stride; a.ichs = j.in_chan_stride;
    a.ocs = j.out_clip_stride; a.ofs = j.out_frame_stride; a.ochs = j.out_chan_stride;
    a.in_frames = j.in_frames; a.out_frames = j.out_frames;
    const int64_t n_blocks = (j.out_frames + g.hop_out - 1) / g.hop_out;
    const uint64_t cols = (uint64_t)j.n_clips * j.n_channels;
    if (cols > 65535) return "too many (clip, channel) columns for one launch (max 65535)";
    if (n_blocks > 2147483647LL) return "job too long for one launch";
    void (*kern)(FftArgs) = k_fft_block<SpecRuntime>;
    if (g.A == 2560 && g.B == 2352) kern = k_fft_block<Spec2560x2352>; // compile-time radix schedules
    if (g.A == 1280 && g.B == 1176) kern = k_fft_block<Spec1280x1176>; // for the 147/160 family
    if (g.lds_bytes > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)g.lds_bytes));
    const unsigned nt = 256u;
    const size_t dbg_lds = switches().dbg_fft_lds;
    hipLaunchKernelGGL(kern, dim3((unsigned)n_blocks, (unsigned)cols, 1), dim3(nt), std::max(g.lds_bytes, dbg_lds),
                       (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    *handled = true;
    return nullptr;
}

#endif // host part (FFT_PART != 1)

} // namespace hipsoxr
