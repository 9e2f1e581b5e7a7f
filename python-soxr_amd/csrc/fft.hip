// fft.hip — frequency-domain engine: rational overlap-save resampling.
//
// Same filter as the exact engine (kernels.hip), different evaluation.  The direct form spends 2*T flops per output
// (592 at VHQ 48k->44.1k) and is FMA-bound at <= 28 % of the HBM roofline.  Here the plan's own prototype g is applied
// in the frequency domain:
//
//   block of N_in = M*k input samples  --FFT-->  X[q]
//   Y[q] = X[q] * H[q]  for |q| <= min(N_in, N_out)/2, else 0     (H = DTFT of g at the bin frequencies; truncating /
//   zero-extending the spectrum IS the rate change: bins of both grids are f_in/N_in = f_out/N_out apart)
//   Y  --inverse FFT of size N_out = L*k-->  L*k output samples
//
// with overlap-save: blocks start on period boundaries (input index multiple of M <-> output index multiple of L),
// overlap by more than the filter length, and only the outputs whose whole filter support lies inside the block are
// kept.  ~70 flop per output instead of 592.  What is neglected is the aliasing of g's stop band (<= -176 dB for VHQ):
// against the direct form 2.5e-10 relative RMS in float64, ~2e-7 in float32 (FFT rounding) — inside the 1e-6 bar but
// NOT bit-identical to the canonical order, so this engine serves only whole-signal float32 / float64 device jobs
// (hipsoxr_run_device); the host surface (soxr.resample / ResampleStream) and integer I/O stay on the exact engine.
//
// Kernels:
//   k_fft_block     general path, any 7-smooth plan: one workgroup per block, run-time radix schedule, real FFT through
//                   a half-length complex transform (untangle * H * tangle).
//   k_fft_pair2     two real blocks ride as the real and imaginary part of ONE complex signal through
//                   FFT -> *H -> truncate -> inverse FFT (the chain is linear and real-to-real; H is real); compile-time
//                   three-pass schedules for the standard audio ratios; unit-stride columns (mono, planar, batches):
//                   raw buffer loads with the hardware range check, output runs staged through LDS and stored as
//                   16-byte granules.  float32, float64 and float32-on-float64 instances.
//   k_fft_strided2  the same chain for columns with a frame stride: interleaved data paired by channel (CP = true:
//                   one (Real, Real) word per frame) or strided columns paired by block (CP = false).
// Build switches: -DFFT2_TRACE (per-wave s_memtime stamps of k_fft_pair2, tools/trace_pair2.py).  The experiments of
// rounds 1-4 (first-generation kernel, resident workgroups + queue, LDS-DMA staging, wave-local DIF schedule, early
// table loads, interleaved stores, ablation bits; two block pairs per workgroup, hand-packed complex arithmetic, the
// thread-count sweep) are in git history (tags r3-fft-experiments, r4-fft-experiments) and profiles/r03_* / r04_*.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <vector>

#include "device.h"
#ifndef FFT_FORCE_PK
#define FFT_NO_PK // (the packed-FMA forms of fft_dev.h: +8 % on k_fft_pair2 — 124 against 114 us — and -7 % on k_fft_wave, which alone uses them)
#endif
#include "fft_dev.h"

namespace hipsoxr {

#define HIP_TRY(expr)                                       \
    do {                                                    \
        hipError_t e_ = (expr);                             \
        if (e_ != hipSuccess) return hipGetErrorString(e_); \
    } while (0)


// One Stockham pass of a length-N transform with a run-time schedule (k_fft_block), IN PLACE in a single LDS buffer:
// every thread reads the inputs of its butterflies into registers, the workgroup synchronises, then results are written
// to their autosort positions.  Radix R, Ns = product of earlier radices, NB = max butterflies per thread.
// W = table exp(SIGN*2*pi*i*m/N), m = 0..N-1 (L1/L2 resident); only the t = 1 twiddle of a butterfly is loaded, its
// powers are formed in registers.
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
// `load(j, t)` supplies input t of butterfly j — element j + t * N/R of the pass input (LDS, or global memory for the
// first pass) — and `store(o, t, v)` consumes element o + t * Ns of its output (LDS, or the staging layout for the last
// pass), so the first pass streams straight from HBM without an extra LDS round trip.  Loaders and storers get the
// butterfly's own index and the COMPILE-TIME input number apart: everything that depends on t alone (offsets, which side
// of the spectrum a bin is on) folds into immediates, and a thread's addresses are a base register plus a constant.
// The in-place barrier stands straight behind the pass's LDS reads, not behind its butterflies: what it must guarantee
// is that every thread HOLDS its inputs, not that it has finished computing.
// Twiddles: any power formed from ONE rounded table entry inherits t times its phase error ((w(1+e))^t ~ w^t (1+te)),
// so for the large radices a second entry, w^4, is read and w^(4a+b) = (w^4)^a w^b: the error factor drops from R-1 to
// <= a+b for the same number of complex products (engine error 3.5e-7 -> 2.2e-7 relative RMS).
// `tid` = the thread's index among the NT threads that share ONE transform (threadIdx.x; an experiment of round 5 interleaved
// the transforms of four channel pairs across the lanes of one workgroup: profiles/NOTES_r05.md §3).
template <int N, int Ns, int R, int SIGN, int NT, bool SYNC_BEFORE_STORE, typename C, typename Load, typename Store>
__device__ __forceinline__ void fft_pass_ct(const C *W, Load load, Store store, const int tid)
{
    constexpr int nb = N / R, wstep = N / (Ns * R), NB = (nb + NT - 1) / NT;
    typedef real_of<C> T;
    C u[NB][R];
    C w1s[NB], w4s[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int j = tid + i * NT;
        w1s[i] = C((T)1, (T)0); w4s[i] = w1s[i];
        if (NB * NT == nb || j < nb) {
            const int k = j % Ns;
            if (Ns > 1) w1s[i] = W[k * wstep];
            if (Ns > 1 && R >= 10) w4s[i] = W[4 * k * wstep]; // 4*k*wstep < 4N/R <= N
#pragma unroll
            for (int t = 0; t < R; ++t) u[i][t] = load(j, t);
        }
    }
    if (SYNC_BEFORE_STORE) __syncthreads(); // in place: every input of the pass is in registers
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int j = tid + i * NT;
        if (NB * NT == nb || j < nb) {
            const int k = j % Ns, o = (j - k) * R + k;
            if (Ns > 1) {
                C pw[R];
                const C w1 = w1s[i];
                pw[1] = w1;
                if constexpr (R >= 10) {
                    const C w4 = w4s[i];
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
            dft_r<R, SIGN>(u[i]);
#pragma unroll
            for (int t = 0; t < R; ++t) store(o, t, u[i][t]);
        }
    }
}

// Three-pass transform: first pass input from `first_load`, last pass output to `last_store`, in place in between.
// LDS bank conflicts: pass loads are contiguous across lanes (conflict-free); pass stores run in groups of Ns consecutive
// elements, so only the FIRST pass (Ns = 1: lane stride = R0 elements) can conflict.  An odd-ish R0 (5, 21: stride 40 /
// 168 bytes) is conflict-free as it is; for R0 = 16 (stride 128 bytes = every lane on the same two banks) the buffer
// between pass 1 and pass 2 is kept in a swizzled layout  n -> n ^ ((n >> 4) & 15)  (SWZ).  What the swizzle costs in
// address arithmetic (round 5): pass 1 stores element t of butterfly j at 16 j + (t ^ (j & 15)) — one XOR per element;
// pass 2 reads element j + t * N/R1, whose mask ((j >> 4) + t * N/(16 R1)) & 15 takes only 16 / gcd(16, N/(16 R1))
// different values over t (four for 5120 = 16 * 16 * 20): that many base addresses per thread, every read an immediate
// offset from one of them (it was an add, a shift-and-mask and an XOR per element: 65 -> 16 vector instructions).
#ifdef FFT2_TRACE
#define FFT_STAMP() do { if (g_tr && (threadIdx.x & 63) == 0 && g_tri < 16) g_tr[g_tri] = __builtin_amdgcn_s_memtime(); ++g_tri; } while (0)
#define FFT_STAMP_DECL unsigned long long *g_tr, int &g_tri,
#define FFT_STAMP_ARGS g_tr, g_tri,
#else
#define FFT_STAMP() ((void)0)
#define FFT_STAMP_DECL
#define FFT_STAMP_ARGS
#endif
// (last_store_alt / use_alt: a second form of the last pass's consumer behind ONE wave-uniform branch around the whole pass)
template <int N, int SIGN, int NT, int R0, int R1, int R2, bool SWZ, bool LASTSYNC, typename C, typename Load, typename Store, typename StoreAlt>
__device__ __forceinline__ void fft_ct3(FFT_STAMP_DECL C *buf, const C *W, Load first_load, Store last_store, bool first_in_lds, StoreAlt last_store_alt,
                                        bool use_alt, const int tid)
{
    static_assert(R0 * R1 * R2 == N, "radix schedule");
    static_assert(!SWZ || R0 == 16, "the swizzled layout is the radix-16 first pass's");
    constexpr int nb1 = N / R1, nb2 = N / R2;
    auto lds_load2 = [&](int j, int t) -> C { return buf[j + t * nb2]; };
    auto lds_store1 = [&](int o, int t, C v) { buf[o + t * R0] = v; };
    auto swz_load = [&](int j, int t) -> C {
        if constexpr (!SWZ) return buf[j + t * nb1];
        else if constexpr (nb1 % 16 == 0) // the element's row (n >> 4) = (j >> 4) + t * nb1 / 16: its low four bits repeat over t
            return buf[(j & ~15) + t * nb1 + ((j & 15) ^ (((j >> 4) + ((t * (nb1 / 16)) & 15)) & 15))];
        else { const int n = j + t * nb1; return buf[n ^ ((n >> 4) & 15)]; }
    };
    auto swz_store = [&](int o, int t, C v) { // pass 0: Ns = 1, o = R0 j
        // 16 j + (t ^ (j & 15)) = (16 j | (j & 15)) ^ t: one XOR of a per-thread byte offset with a constant per element
        if constexpr (SWZ) *reinterpret_cast<C *>(reinterpret_cast<char *>(buf) + ((unsigned)((o | ((o >> 4) & 15)) * (int)sizeof(C)) ^ (unsigned)(t * (int)sizeof(C)))) = v;
        else buf[o + t] = v;
    };
    // pass 0 (Ns = 1): when its input is not in LDS nothing has to be protected before storing
    if (first_in_lds) fft_pass_ct<N, 1, R0, SIGN, NT, true>(W, first_load, swz_store, tid);
    else fft_pass_ct<N, 1, R0, SIGN, NT, false>(W, first_load, swz_store, tid);
    FFT_STAMP();
    __syncthreads();
    FFT_STAMP();
    fft_pass_ct<N, R0, R1, SIGN, NT, true>(W, swz_load, lds_store1, tid);
    FFT_STAMP();
    __syncthreads();
    FFT_STAMP();
    if (use_alt) fft_pass_ct<N, R0 * R1, R2, SIGN, NT, LASTSYNC>(W, lds_load2, last_store_alt, tid);
    else fft_pass_ct<N, R0 * R1, R2, SIGN, NT, LASTSYNC>(W, lds_load2, last_store, tid);
    FFT_STAMP();
}


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

// ---------------------------------------------------------------------------------------------
// General path: one block per workgroup, run-time radix schedule, any 7-smooth plan (ratios outside the table below).
// ---------------------------------------------------------------------------------------------
#if !defined(FFT_PART) || FFT_PART == 0 // (not a template: one translation unit only)
__global__ void __launch_bounds__(256, 2) k_fft_block(FftArgs a)
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

    // ---- load: z[n] = x[2n] + i x[2n+1], zero outside the signal; forward complex FFT of length A
    for (int n = threadIdx.x; n < A; n += blockDim.x) {
        const int64_t l = in0 + 2 * (int64_t)n;
        float re = (l >= a.in_lo && l < a.in_frames) ? xin[l * a.ifs] : 0.f;
        float im = (l + 1 >= a.in_lo && l + 1 < a.in_frames) ? xin[(l + 1) * a.ifs] : 0.f;
        cur[n] = make_float2(re, im);
    }
    __syncthreads();
    run_passes<-1>(cur, A, a.nA, a.radA, a.WA);
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
        constexpr int NP = (4096 / 2 + 1 + 255) / 256; // pairs per thread: B/2 + 1 <= 256 * NP (B <= 4096)
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

    // ---- inverse complex FFT of length B (unnormalised; the scale lives in Hs) and store of the kept outputs:
    //      element n of the result holds local outputs 2n (re) and 2n+1 (im)
    float *yo = (float *)a.out + (int64_t)clip * a.ocs + (int64_t)ch * a.ochs;
    const int32_t v0 = a.v0, v1 = a.v0 + a.hop_out;
    run_passes<+1>(cur, B, a.nB, a.radB, a.WB);
    for (int n = threadIdx.x; n < B; n += blockDim.x) {
        const cf w = cur[n];
        const int32_t i0 = 2 * n;
        const int64_t k0 = out0 + i0;
        if (i0 >= v0 && i0 < v1 && k0 >= 0 && k0 < a.out_frames) yo[k0 * a.ofs] = w.x;
        if (i0 + 1 >= v0 && i0 + 1 < v1 && k0 + 1 >= 0 && k0 + 1 < a.out_frames) yo[(k0 + 1) * a.ofs] = w.y;
    }
}
#endif

// ---------------------------------------------------------------------------------------------
// Paired-block kernels.  The whole chain  FFT -> multiply by H -> truncate -> inverse FFT  maps real signals to real
// signals and is linear, so it processes TWO real blocks at once as the real and imaginary part of one complex signal:
// z = x_a + i x_b  ->  y_a + i y_b  (H is Hermitian — real, in fact: the prototype is symmetric about the output instant
// and blocks are cut on period boundaries — and the truncation symmetric).  No real-FFT untangle/tangle stages; the
// filter multiply rides on the loads of the first inverse pass; with radix-16/20/21 butterflies 3 + 3 LDS passes per
// pair of blocks.
// Schedules: N_in = A0*A1*A2 (forward), N_out = B0*B1*B2 (inverse); *SWZ = swizzled layout after a power-of-two first
// radix (see fft_ct3).  NT >= the largest butterfly count of any pass.
// ---------------------------------------------------------------------------------------------
template <int NA_, int NB_, int NT_, int A0, int A1, int A2, bool ASWZ, int B0, int B1, int B2, bool BSWZ>
struct PairSpec {
    static constexpr int NA = NA_, NB = NB_, NT = NT_;
    static constexpr int RA0 = A0, RA2 = A2, RB0 = B0, RB2 = B2;
    template <typename C, typename Ld, typename St> static __device__ __forceinline__ void fwd(FFT_STAMP_DECL C *b, const C *W, Ld ld, St st, int tid = (int)threadIdx.x)
    { fft_ct3<NA, -1, NT, A0, A1, A2, ASWZ, false>(FFT_STAMP_ARGS b, W, ld, st, false, st, false, tid); }
    template <typename C, typename Ld, typename St> static __device__ __forceinline__ void inv(FFT_STAMP_DECL C *b, const C *W, Ld ld, St st, int tid = (int)threadIdx.x)
    { fft_ct3<NB, +1, NT, B0, B1, B2, BSWZ, false>(FFT_STAMP_ARGS b, W, ld, st, true, st, false, tid); }
    // last pass stores into LDS in another layout (output staging): all its inputs must be in registers first
    template <typename C, typename Ld, typename St, typename StAlt> static __device__ __forceinline__ void inv_staged(FFT_STAMP_DECL C *b, const C *W, Ld ld, St st, StAlt st_alt, bool use_alt)
    { fft_ct3<NB, +1, NT, B0, B1, B2, BSWZ, true>(FFT_STAMP_ARGS b, W, ld, st, true, st_alt, use_alt, (int)threadIdx.x); }
};
// Three-pass schedule of each transform length in use (first radix 21: conflict-free as it is; first radix 16:
// swizzled layout between pass 1 and 2).  4410 = 21*14*15 is the order the product runs (configs[2] 47 us, against
// 52 us with the radix-15 pass first).
template <int N> struct Sched;
#define HIPSOXR_SCHED_LIST(X)                                                                                        \
    X(7056, 21, 16, 21, false) X(5376, 21, 16, 16, false) X(5120, 16, 16, 20, true) X(4704, 21, 16, 14, false)       \
    X(4410, 21, 14, 15, false) X(4096, 16, 16, 16, true) X(3840, 16, 16, 15, true) X(3584, 14, 16, 16, false)        \
    X(3528, 21, 12, 14, false) X(2688, 21, 16, 8, false) X(2560, 16, 16, 10, true) X(2352, 21, 16, 7, false)         \
    X(2048, 16, 16, 8, true) X(1792, 7, 16, 16, false) X(1024, 16, 8, 8, true) X(1600, 16, 10, 10, true)             \
    X(1280, 5, 16, 16, false) X(1176, 21, 8, 7, false) X(896, 7, 16, 8, false)
#define HIPSOXR_SCHED(N, r0, r1, r2, swz) \
    template <> struct Sched<N> { static constexpr int R0 = r0, R1 = r1, R2 = r2; static constexpr bool SWZ = swz; };
HIPSOXR_SCHED_LIST(HIPSOXR_SCHED)
#undef HIPSOXR_SCHED
template <int NA, int NB, int NT>
using PairOf = PairSpec<NA, NB, NT, Sched<NA>::R0, Sched<NA>::R1, Sched<NA>::R2, Sched<NA>::SWZ, Sched<NB>::R0, Sched<NB>::R1,
                        Sched<NB>::R2, Sched<NB>::SWZ>;


// Input t of butterfly j of the FIRST INVERSE pass: bin n = j + t * NB/RB0 of the output grid <- bin n (non-negative
// frequencies, n <= NB/2) or n + NA - NB (negative ones) of the input grid in `buf`, times the real filter gain of
// |frequency| — read through the descriptor `rh` over Hr[0 .. NB/2] with the t-dependent part of the offset in the scalar
// operand.  Which side of the spectrum a bin lies on is known at compile time for all but the one t that straddles NB/2.
template <typename Spec, typename Real, typename C>
__device__ __forceinline__ C spectrum_load(const C *buf, __amdgpu_buffer_rsrc_t rh, int j, int t)
{
    constexpr int NA = Spec::NA, NB = Spec::NB, nbB = NB / Spec::RB0, RS = (int)sizeof(Real);
    const int lo = t * nbB, hi = lo + nbB - 1; // the bins this input can be, over all butterflies (j < nbB)
    const int n = j + lo;
    if constexpr (NA >= NB) {
        if (hi <= NB / 2) {        // non-negative frequencies
            const Real h = buf_load_real<Real>(rh, j * RS, lo * RS);
            const C x = buf[n];
            return C(x.x * h, x.y * h);
        } else if (lo > NB / 2) {  // negative frequencies: |q| = NB - n = (NB - lo - nbB) + (nbB - j)
            const Real h = buf_load_real<Real>(rh, (nbB - j) * RS, (NB - lo - nbB) * RS);
            const C x = buf[n + (NA - NB)];
            return C(x.x * h, x.y * h);
        } else {                   // the butterfly input that straddles the middle
            const bool neg = n > NB / 2;
            const Real h = buf_load_real<Real>(rh, (neg ? NB - n : n) * RS, 0);
            const C x = buf[neg ? n + (NA - NB) : n];
            return C(x.x * h, x.y * h); // (the Nyquist bin's alias term is dropped with Im H: stop band, < -170 dB)
        }
    } else {
        const bool neg = n > NB / 2;
        const int q = neg ? NB - n : n; // |frequency| in bins
        const bool in_band = q < NA / 2;
        const Real h = buf_load_real<Real>(rh, q * RS, 0);
        const C x = buf[in_band ? (neg ? NA - q : q) : 0];
        return in_band ? C(x.x * h, x.y * h) : C((Real)0, (Real)0);
    }
}

// ---------------------------------------------------------------------------------------------
// k_fft_pair2: unit-stride columns (mono / planar data; batches).  How it touches HBM:
//   * input: raw buffer loads whose descriptor covers [first sample of the item's first block, end of the column): the
//     hardware range check returns 0 past the end of the signal (no per-element bounds code, one path for interior and
//     last items), and the per-butterfly offsets t*N/R0 (+ block * hop) ride in the instruction's scalar offset instead
//     of 64-bit vector address arithmetic;
//   * output: the last inverse pass writes its kept outputs into LDS as the contiguous run they are in memory (block a
//     then block b of the pair: 2 hop_out consecutive elements of the column), index-shifted so that LDS and memory
//     agree on 16-byte phase, and the workgroup then stores the run with 16-byte buffer stores — every wave writes 1 KB
//     of whole 16-byte granules (write traffic = algorithmic bytes).
// Real = float: float32 device jobs.  Real = double: float64 device jobs — libsoxr's own VHQ engine is a float64 one
// (SURVEY.md §0.3); the same chain in double2 (LDS 16 bytes per point), results within the method's own floor of the
// float64 direct form (the neglected stop-band aliasing, ~3e-10 for VHQ).  IO = the signal's element type when it
// differs from the arithmetic: <double, float> is float32 I/O on float64 arithmetic — what libsoxr's VHQ recipe itself
// does for float32 clients (reference src/soxr_ext.cpp:74,228) — selected by HIPSOXR_KERNEL_FFT_F64.
// One work item = a pair of blocks of one column: item (col, bx) = blocks 2 bx, 2 bx + 1.  An item beyond its clip's
// last pair (ragged batches) leaves at once.
// Round 5, instruction diet (tools/isa_stats.py; the launch runs at the board's power cap, so instructions are energy):
// the filter values come through a buffer descriptor too (offsets in the scalar operand; which side of the spectrum a
// bin lies on is decided at compile time for all but the one butterfly input that straddles the middle: 97 -> ~50
// vector instructions in front of the first inverse butterfly), and the staging stores test their range per butterfly
// OUTPUT in the scalar unit — only the two outputs that can straddle an end of the kept run compare per lane.
// ---------------------------------------------------------------------------------------------
template <typename Spec, typename Real, typename IO>
__device__ __forceinline__ void pair2_item(const FftArgs &a, unsigned char *smem_raw, uint32_t col, int64_t bx)
{
    typedef typename PairTabs<Real>::C C;
    typedef typename PairTabs<IO>::V16 V16;
    constexpr int ES = (int)sizeof(IO), EPS = 16 / ES; // element size, elements per 16-byte store
    constexpr int NA = Spec::NA, NB = Spec::NB, NT = Spec::NT, nbA = NA / Spec::RA0;
    constexpr int NsL = NB / Spec::RB2;   // the last inverse pass writes element o + t * NsL, o < NsL
    constexpr int LB = NA > NB ? NA : NB; // complex points of the transform buffer
    C *buf = reinterpret_cast<C *>(smem_raw);
    IO *stage = reinterpret_cast<IO *>(smem_raw);
#ifdef FFT2_TRACE
    unsigned long long *g_tr = a.trace ? a.trace + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (NT / 64) + threadIdx.x / 64) * 16 : nullptr;
    int g_tri = 0;
    FFT_STAMP();
#endif

    const uint32_t ch = __builtin_amdgcn_readfirstlane(col % a.n_channels);
    const uint32_t clip = __builtin_amdgcn_readfirstlane(col / a.n_channels);
    const int64_t pa = 2 * bx * a.hop_periods - a.lead_periods; // first period of the pair's first block; the second starts hop_periods later
    const int64_t ina = pa * a.M, outa = pa * a.L;
    const int32_t hop_in = (int32_t)(a.hop_periods * a.M);
    // ragged batch: this clip's own place and length (four scalar loads; the grid spans the longest clip, so a
    // workgroup beyond its clip's last pair has nothing to do)
    int64_t clip_in = (int64_t)clip * a.ics, clip_out = (int64_t)clip * a.ocs, in_frames = a.in_frames, out_frames = a.out_frames;
    if (a.clip_tab) {
        const int64_t *row = a.clip_tab + 4 * (size_t)clip;
        clip_in = row[0]; in_frames = row[1]; clip_out = row[2]; out_frames = row[3];
    }
    if (outa + a.v0 >= out_frames) return;
    const IO *xin = (const IO *)a.in + clip_in + (int64_t)ch * a.ichs;
    auto last_fwd_store = [&](int o, int t, C v) { buf[o + t * (NA / Spec::RA2)] = v; };

    // ---- forward: z[n] = x_a[n] + i x_b[n], first pass straight from HBM --------------------------
    if (ina >= a.in_lo) {
        const int64_t left = (in_frames - ina) * ES; // bytes from the first block's first sample to the end of the column
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            uniform_ptr((void *)(xin + ina)), 0, __builtin_amdgcn_readfirstlane((int)(left < 0 ? 0 : left > 0x40000000 ? 0x40000000 : left)), 0x00020000);
        Spec::fwd(FFT_STAMP_ARGS buf, PairTabs<Real>::wa(a), [&](int j, int t) -> C { // (the butterfly's own offset: one VGPR for all t)
            return C((Real)buf_load_real<IO>(rs, j * ES, t * nbA * ES), (Real)buf_load_real<IO>(rs, j * ES, (t * nbA + hop_in) * ES));
        }, last_fwd_store);
    } else { // the first item of a column reaches before its start: explicit zero-extension
        Spec::fwd(FFT_STAMP_ARGS buf, PairTabs<Real>::wa(a), [&](int j, int t) -> C {
            const int64_t la = ina + j + t * nbA, lb = la + hop_in;
            return C((la >= a.in_lo && la < in_frames) ? (Real)xin[la] : (Real)0, (lb >= a.in_lo && lb < in_frames) ? (Real)xin[lb] : (Real)0);
        }, last_fwd_store);
    }
    // the filter, |frequency| in bins -> real gain, through a descriptor of its own: Hr[0 .. NB/2]
    const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr((void *)PairTabs<Real>::hr(a)), 0,
                                                                         (NB / 2 + 1) * (int)sizeof(Real), 0x00020000);
    __syncthreads();
    FFT_STAMP();

    // ---- inverse: bin n of the output grid <- bin n or n + NA - NB of the input grid, times (real) H; the last pass
    //      writes into the staging layout.  Input t of butterfly j is bin n = j + t * nbB: bins up to NB/2 are the
    //      non-negative frequencies, the rest the negative ones — for all but one t that is known at compile time.
    const int32_t v0 = a.v0, v1 = a.v0 + a.hop_out, hop_out = a.hop_out;
    IO *ybase = (IO *)a.out + clip_out + (int64_t)ch * a.ochs + (outa + v0); // run[0]; outa + v0 >= 0
    // LDS element index == run index + sh: the 16-byte phases of staging and memory agree
    const int32_t sh = (int32_t)((reinterpret_cast<uintptr_t>(ybase) / ES) & (EPS - 1));
    auto h_load = [&](int j, int t) -> C { return spectrum_load<Spec, Real>(buf, rh, j, t); };
    // Staging stores: output n = o + t * NsL (o < NsL) of both blocks is kept iff v0 <= n < v1.  In every geometry in use
    // the kept run begins inside the first stride of outputs and ends inside the last one (`typical`, wave-uniform):
    // then only outputs t = 0 and t = RB2 - 1 compare per lane, the others are stored as they are.
    const bool typical = v0 >= 0 && v0 <= NsL && v1 >= NB - NsL && v1 <= NB;
    Spec::inv_staged(FFT_STAMP_ARGS buf, PairTabs<Real>::wb(a), h_load,
        [&](int o, int t, C w) { // typical geometry
            IO *const sa = stage + (o - v0 + sh) + t * NsL, *const sb = sa + hop_out;
            if (t == 0) { if (o >= v0) { *sa = (IO)w.x; *sb = (IO)w.y; } }
            else if (t == Spec::RB2 - 1) { if (o < v1 - t * NsL) { *sa = (IO)w.x; *sb = (IO)w.y; } }
            else { *sa = (IO)w.x; *sb = (IO)w.y; }
        },
        [&](int o, int t, C w) { // any geometry
            IO *const sa = stage + (o - v0 + sh) + t * NsL, *const sb = sa + hop_out;
            if ((unsigned)(o + t * NsL - v0) < (unsigned)hop_out) { *sa = (IO)w.x; *sb = (IO)w.y; }
        }, !typical);
    __syncthreads();
    FFT_STAMP();

    // ---- store the run: elements [0, valid) of it exist in the column ----------------------------------
    const int64_t remain = out_frames - (outa + v0);
    const int32_t valid = (int32_t)(remain < 0 ? 0 : remain > 2 * (int64_t)hop_out ? 2 * (int64_t)hop_out : remain);
    // 16-byte buffer stores: the descriptor starts at the 16-byte granule that holds run[0] (sh elements before it)
    // and ends with the run, so the hardware range check drops what lies beyond the column (and the trips past the
    // run: no trip count, no branches — every LDS read and every store of the thread is in flight at once).  The
    // first granule's sh leading elements belong to the previous run: that one granule goes element by element.
    {
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr((void *)(ybase - sh)), 0,
                                                                             __builtin_amdgcn_readfirstlane((valid + sh) * ES), 0x00020000);
        constexpr int QMAX = (2 * (NB - 1) + EPS - 1 + EPS) / EPS; // 2 hop_out < 2 NB elements, + sh
        constexpr int LQ = (int)((size_t)LB * sizeof(C) / 16);     // 16-byte granules of the LDS buffer
        const int tid_out = (int)threadIdx.x;
#pragma unroll
        for (int it = 0; it < (QMAX + NT - 1) / NT; ++it) {
            const int q = tid_out + it * NT;
            const V16 v = *reinterpret_cast<const V16 *>(stage + EPS * (q < LQ ? q : LQ - 1));
            if (q == 0 && sh != 0) {
                const IO *e = reinterpret_cast<const IO *>(&v);
#pragma unroll
                for (int c = 0; c < EPS; ++c)
                    if (c >= sh && c - sh < valid) ybase[c - sh] = e[c];
            } else {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u_t, v), ro, q * 16, 0, FFT_STORE_AUX);
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
}

template <typename Spec, typename Real, typename IO = Real>
__global__ void __launch_bounds__(Spec::NT) k_fft_pair2(FftArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    pair2_item<Spec, Real, IO>(a, smem_raw, blockIdx.y, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// k_fft_strided2: columns with a frame stride.
// CP = true, CHANNEL-PAIR mode: interleaved data with an even channel count.  The two real signals of a transform are
// the same block of two neighbouring channels — one aligned (Real, Real) word per frame: one 8- or 16-byte raw buffer
// load per element (descriptor over [first frame of the block, end of the column), hardware range check instead of
// per-element bounds code, the per-butterfly offset t*N/R0*frame in the instruction's scalar operand) and one
// range-checked buffer store per kept output.
// CP = false: strided columns that cannot be paired by channel (odd channel counts of interleaved data, a channel slice
// with a frame stride): two consecutive blocks of ONE column are paired, as in k_fft_pair2, each element a 4/8-byte
// buffer load or store at the column's frame stride.
// Workgroup ids are XCD-aware for interleaved data (a.xcd_map): consecutive ids are dealt round-robin to the 8 XCDs,
// each with a private L2, while the channel units of one block of frames share every cache line — so they get ids that
// are congruent mod 8 and adjacent in dispatch order (x = 8 * slot + xcd, slot = chunk * units + unit,
// item = xcd * ceil(items / 8) + chunk).  Without it each line is fetched and (partially) written once per channel unit: 2.3x / 4x
// the algorithmic bytes at 8 channels.  (The derived indices need readfirstlane: a run-time integer division goes
// through the vector ALU, and the compiler then keeps every address in VGPRs.)
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

template <typename Spec, typename Real, bool CP>
__global__ void __launch_bounds__(Spec::NT) k_fft_strided2(FftArgs a)
{
    typedef typename PairTabs<Real>::C C;
    constexpr int ES = (int)sizeof(Real);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    C *cur = reinterpret_cast<C *>(smem_raw);
    constexpr int NA = Spec::NA, NB = Spec::NB, R0 = Spec::RA0, nbA = NA / R0;
#ifdef FFT2_TRACE
    unsigned long long *g_tr = a.trace ? a.trace + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (Spec::NT / 64) + threadIdx.x / 64) * 16 : nullptr;
    int g_tri = 0;
    FFT_STAMP();
#endif
    // XCD-aware ids: x = 8 * slot + xcd, slot = chunk * units + unit;
    // or (a.xcd_map == 0: one column per grid row) items along x, columns along y
    const uint32_t units = CP ? a.n_channels / 2 : a.n_channels;
    const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const bool xm = a.xcd_map != 0;
    const uint32_t cu = __builtin_amdgcn_readfirstlane(xm ? slot % units : blockIdx.y % units);
    const uint32_t clip = __builtin_amdgcn_readfirstlane(xm ? blockIdx.y : blockIdx.y / units);
    // ... and each XCD takes a CONTIGUOUS run of the column's blocks (item = xcd * per_xcd + chunk), so that the input two
    // neighbouring blocks share — 17.7 % of a 4410-frame block at 44.1k -> 16k — meets in ONE L2 instead of being fetched
    // from HBM by two (round 4; with item = 8 * chunk + xcd every XCD saw blocks b, b + 8, b + 16 ...: configs[2] HBM
    // traffic 137.3 MB = 1.19x the algorithmic bytes -> 116.0 MB = 1.005x, launch time unchanged: tools/c2_traffic.sh)
    const uint32_t per_xcd = (uint32_t)((a.pairs_per_col + 7) / 8);
    const int64_t bx = (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane(xm ? xcd * per_xcd + slot / units : blockIdx.x);
    if (xm && slot / units >= per_xcd) return;
    if (bx >= a.pairs_per_col) return; // grid.x is padded to a multiple of 8 items per unit
    const uint32_t ch = CP ? 2 * cu : cu;
    const int32_t hop_in = (int32_t)(a.hop_periods * a.M), hop_out = a.hop_out;
    const Real *xin = (const Real *)a.in + (int64_t)clip * a.ics + (int64_t)ch * a.ichs;
    const int32_t ifb = (int32_t)a.ifs * ES, ofb = (int32_t)a.ofs * ES; // bytes per frame (launcher: 2 N * frame < 2^30)
    auto last_fwd_store = [&](int o, int t, C v) { cur[o + t * (NA / Spec::RA2)] = v; };
    const int32_t v0 = a.v0, v1 = a.v0 + hop_out;
    const int64_t pa = (CP ? 1 : 2) * bx * a.hop_periods - a.lead_periods; // first period of the (first) block
    const int64_t ina = pa * a.M, outa = pa * a.L;

    // ---- forward: z[n] = x_c[n] + i x_{c+1}[n]  (CP)  or  x_a[n] + i x_b[n]  (two blocks), first pass straight from HBM
    if (ina >= a.in_lo) {
        const int64_t left = (a.in_frames - ina) * (int64_t)ifb; // bytes from the block's first frame to the end of the column
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            uniform_ptr((void *)(xin + ina * a.ifs)), 0, __builtin_amdgcn_readfirstlane((int)(left < 0 ? 0 : left > 0x40000000 ? 0x40000000 : left)), 0x00020000);
        const int32_t stepb = nbA * ifb; // one butterfly input further: N/R0 frames
        Spec::fwd(FFT_STAMP_ARGS cur, PairTabs<Real>::wa(a), [&](int j, int t) -> C {
            if constexpr (CP) return CpIo<Real>::load(rs, j * ifb, t * stepb);
            else return C(buf_load_real<Real>(rs, j * ifb, t * stepb), buf_load_real<Real>(rs, j * ifb, t * stepb + hop_in * ifb));
        }, last_fwd_store);
    } else { // the first block of a column reaches before its start: explicit zero-extension
        Spec::fwd(FFT_STAMP_ARGS cur, PairTabs<Real>::wa(a), [&](int j, int t) -> C {
            const int64_t l = ina + j + t * nbA, lb = l + hop_in;
            if constexpr (CP) {
                C v = C((Real)0, (Real)0);
                if (l >= a.in_lo && l < a.in_frames) v = C(xin[l * a.ifs], xin[l * a.ifs + 1]);
                return v;
            } else {
                return C((l >= a.in_lo && l < a.in_frames) ? xin[l * a.ifs] : (Real)0, (lb >= a.in_lo && lb < a.in_frames) ? xin[lb * a.ifs] : (Real)0);
            }
        }, last_fwd_store);
    }
    __syncthreads();
    FFT_STAMP();

    // ---- inverse (see k_fft_pair2), outputs straight to HBM -----------------------------------------
    Real *ybase = (Real *)a.out + (int64_t)clip * a.ocs + (int64_t)ch * a.ochs + (outa + v0) * a.ofs; // outa + v0 >= 0
    const int64_t oleft = (a.out_frames - (outa + v0)) * (int64_t)ofb;
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
        uniform_ptr((void *)ybase), 0, __builtin_amdgcn_readfirstlane((int)(oleft < 0 ? 0 : oleft > 0x40000000 ? 0x40000000 : oleft)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr((void *)PairTabs<Real>::hr(a)), 0,
                                                                         (NB / 2 + 1) * (int)sizeof(Real), 0x00020000);
    auto h_load = [&](int j, int t) -> C { return spectrum_load<Spec, Real>(cur, rh, j, t); };
    Spec::inv(FFT_STAMP_ARGS cur, PairTabs<Real>::wb(a), h_load, [&](int o, int t, C wv) {
        const int n = o + t * (NB / Spec::RB2);
        if (n >= v0 && n < v1) {
            if constexpr (CP) {
                CpIo<Real>::store(wv, ro, (n - v0) * ofb); // frame outa + n holds (y_c, y_{c+1})
            } else {
                buf_store_real(wv.x, ro, (n - v0) * ofb);             // block a
                buf_store_real(wv.y, ro, (n - v0 + hop_out) * ofb);   // block b: hop_out frames further
            }
        }
    });
#ifdef FFT2_TRACE
    if (g_tr && (threadIdx.x & 63) == 0) { g_tr[13] = __builtin_amdgcn_s_getreg((31 << 11) | 4); g_tr[14] = __builtin_amdgcn_s_getreg((31 << 11) | 20); }
    g_tri = 15;
    FFT_STAMP();
#endif
}

// ---------------------------------------------------------------------------------------------
// Three translation units.  Every schedule of the table below is 7 kernels (k_fft_pair2 in float32, float64 and
// float32-on-float64, k_fft_strided2 x 2 in float32 and float64); compiled in one piece they are the build's critical
// path.  build.sh compiles this file three times: -DFFT_PART=0 = everything except the kernels of the schedules listed
// here (declared extern), -DFFT_PART=1 / =2 = the templates above plus exactly the kernels of one of the two lists, no
// host code.  Without FFT_PART: one piece.
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
    // block of k periods: candidates are power-of-two k with 7-smooth even half-lengths; take the largest block whose
    // transforms stay <= 2600 points (one 20 KB LDS buffer, least overlap waste), else the smallest admissible one
    for (int k = force_k ? force_k : 1; k <= (force_k ? force_k : 4096); k *= 2) {
        const int64_t Nin = M * k, Nout = L * k;
        if ((Nin % 2 || Nout % 2) && !force_k) continue; // (the half-length tables of k_fft_block; the paired kernels take odd lengths)
        if (!force_k && Nin < 6 * (int64_t)T) continue;
        if (Nin / 2 > 4096 || Nout / 2 > 4096) break;
        std::vector<int> ra, rb;
        if (!factor_radices((int)(Nin / 2), ra) || !factor_radices((int)(Nout / 2), rb)) {
            if (!force_k) continue;
            ra.clear(); rb.clear(); // (k_fft_block's run-time schedule of the half lengths: a forced geometry belongs to a paired kernel, which has its own)
        }
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
    std::vector<float2> tab((size_t)A + B + (A + 1) + B + (B + 1) + g.N_in + g.N_out + (B + 2) / 2 + 1);
    float2 *WA = tab.data(), *WB = WA + A, *P = WB + B, *Q = P + (A + 1), *Hs = Q + B;
    float2 *WA2 = Hs + (B + 1), *WB2 = WA2 + g.N_in;
    const double PI2 = 6.283185307179586476925286766559;
    for (int m = 0; m < g.N_in; ++m) WA2[m] = make_float2((float)std::cos(PI2 * m / g.N_in), (float)-std::sin(PI2 * m / g.N_in));
    for (int m = 0; m < g.N_out; ++m) WB2[m] = make_float2((float)std::cos(PI2 * m / g.N_out), (float)std::sin(PI2 * m / g.N_out));
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
        // even-length support, g[-L T/2], is a window-edge value ~1e-11).  The paired kernels use Re H alone:
        // half the table reads and a real x complex product per bin.
        reinterpret_cast<float *>(WB2 + g.N_out)[q] = (float)(hr * scale);
        hr64[q] = hr * scale;
    }
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

// Whole-signal float32 / float64 job?  (zero-extended signal starting at absolute index 0, all outputs)
bool fft_job_eligible(const Plan &p, const hipsoxr_job_t &j)
{
    // what the method neglects is the aliasing of the filter's stop band: only recipes whose stop band
    // is far below the 1e-6 bar qualify (HQ 128 dB, VHQ 177 dB; MQ/LQ at 104 dB do not)
    return p.phases == 0 && p.att_db >= 120. && (j.elem == HIPSOXR_F32 || j.elem == HIPSOXR_F64) && j.in_abs0 == 0 &&
           j.out_k0 == 0 && (uint64_t)j.out_frames <= plan_out_len(p, (uint64_t)j.in_frames);
}

// (job.in_abs0 != 0 — in[0] is sample in_abs0 of a column that is zero outside [in_abs0, in_abs0 + in_frames) — is served
// for the two-stage form's inner calls; the public paths come here through fft_job_eligible, which wants 0.)
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
        unsigned nt;
        void (*kern2)(FftArgs); void (*kern2d)(FftArgs); // unit-stride columns, float32 / float64
        void (*kern2fd)(FftArgs);                        // float32 I/O on float64 arithmetic (HIPSOXR_KERNEL_FFT_F64)
        void (*kcp)(FftArgs); void (*kcpd)(FftArgs);     // channel-pair mode (interleaved data), float32 / float64
        void (*kst)(FftArgs); void (*kstd)(FftArgs);     // strided columns, two blocks per transform
    };
#define HIPSOXR_PAIR(L, M, k, small, NA, NB, NT) \
    {L, M, k, small, NT, k_fft_pair2<PairOf<NA, NB, NT>, float>, k_fft_pair2<PairOf<NA, NB, NT>, double>, \
     k_fft_pair2<PairOf<NA, NB, NT>, double, float>, \
     k_fft_strided2<PairOf<NA, NB, NT>, float, true>, k_fft_strided2<PairOf<NA, NB, NT>, double, true>, \
     k_fft_strided2<PairOf<NA, NB, NT>, float, false>, k_fft_strided2<PairOf<NA, NB, NT>, double, false>}
    static const PairEntry pairs[] = {
        // L, M (out/in = L/M), periods per block, small-job variant, N_in, N_out, threads
        HIPSOXR_PAIR(147, 160, 32, false, 5120, 4704, 384), HIPSOXR_PAIR(147, 160, 16, true, 2560, 2352, 384),   // 48k -> 44.1k
        HIPSOXR_PAIR(160, 147, 32, false, 4704, 5120, 384), HIPSOXR_PAIR(160, 147, 16, true, 2352, 2560, 384),   // 44.1k -> 48k
        HIPSOXR_PAIR(147, 160, 8, 2, 1280, 1176, 256), HIPSOXR_PAIR(160, 147, 8, 2, 1176, 1280, 256),           // ... quarter-size blocks: jobs of a few hundred pairs
        HIPSOXR_PAIR(160, 441, 16, false, 7056, 2560, 448), HIPSOXR_PAIR(441, 160, 16, false, 2560, 7056, 448),  // 44.1k <-> 16k
        HIPSOXR_PAIR(160, 441, 10, true, 4410, 1600, 320), HIPSOXR_PAIR(441, 160, 10, true, 1600, 4410, 320),    // ... 35 KB blocks: 4 workgroups per CU
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
    // float64: the paired kernels only (unit-stride columns; channel pairs; strided columns) — else the exact engine
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
            if (use) {
                FftArgs a;
                a.in = (const char *)j.in - j.in_abs0 * j.in_frame_stride * (int64_t)(j.elem == HIPSOXR_F64 ? 8 : 4); a.out = j.out; // (sample 0 of the columns)
                auto set_geom = [](FftArgs &a, const FftGeom &g) {
                    a.WA = g.dev; a.WB = a.WA + g.A; a.P = a.WB + g.B; a.Q = a.P + (g.A + 1); a.Hs = a.Q + g.B;
                    a.WA2 = a.Hs + (g.B + 1); a.WB2 = a.WA2 + g.N_in;
                    a.Hr = reinterpret_cast<const float *>(a.WB2 + g.N_out); a.trace = nullptr;
                    a.WA2d = g.devd; a.WB2d = g.devd + g.N_in; a.Hrd = reinterpret_cast<const double *>(g.devd + g.N_in + g.N_out);
                    a.A = g.A; a.B = g.B; a.nA = a.nB = 0;
                    for (int i = 0; i < 8; ++i) a.radA[i] = a.radB[i] = 1;
                    a.lead_periods = g.lead_periods; a.hop_periods = g.hop_periods; a.v0 = g.v0; a.hop_out = g.hop_out;
                };
                set_geom(a, g);
                a.L = p->L; a.M = p->M;
                a.n_clips = j.n_clips; a.n_channels = j.n_channels;
                a.ics = j.in_clip_stride; a.ifs = j.in_frame_stride; a.ichs = j.in_chan_stride;
                a.ocs = j.out_clip_stride; a.ofs = j.out_frame_stride; a.ochs = j.out_chan_stride;
                a.in_lo = j.in_abs0; a.in_frames = j.in_abs0 + j.in_frames; a.out_frames = j.out_frames;
                a.clip_tab = j.clip_table_dev;
                const int64_t n_blocks = (j.out_frames + g.hop_out - 1) / g.hop_out;
                if (n_blocks > 2147483647LL) return "job too long for one launch";
                // interleaved data with an even channel count: pair channels (one (Real, Real) word per frame)
                const size_t esz = io64 ? sizeof(double) : sizeof(float);
                const bool cp_layout = j.n_channels % 2 == 0 && j.in_chan_stride == 1 && j.out_chan_stride == 1 && !switches().fft_no_chpair;
                // ... when a block's byte offsets fit the kernel's 32-bit operands (buffer loads: element alignment is enough)
                const bool cp2 = !wide32 && (int64_t)std::max(g.N_in, g.N_out) * std::max(j.in_frame_stride, j.out_frame_stride) * (int64_t)esz < (1LL << 30);
                // (channel pairing rides on the XCD-aware work-item map: decided together, so that a job without the
                //  map — HIPSOXR_FFT_NO_XCD_MAP, or too many work items — runs unpaired on the strided kernel instead of failing)
                const int64_t cp_items8 = (n_blocks + 7) / 8 * 8;
                const bool cp_map_ok = j.n_channels > 1 && j.n_clips <= 65535 && cp_items8 * (int64_t)(j.n_channels / 2) <= 2147483647LL &&
                                       !switches().fft_no_xcd_map;
                a.chpair = (cp_layout && cp_map_ok && cp2) ? 1 : 0;
                const size_t lds1 = std::max((size_t)std::max(g.N_in, g.N_out) * (f64 ? sizeof(double2) : sizeof(float2)), switches().dbg_fft_lds);
                if (f64 && lds1 > 160 * 1024) return nullptr;
                // work items per channel unit: blocks (channel pairs) or pairs of blocks (single channels)
                const int64_t items = a.chpair ? n_blocks : (n_blocks + 1) / 2, items8 = (items + 7) / 8 * 8;
                const int64_t units = a.chpair ? j.n_channels / 2 : j.n_channels;
                a.xcd_map = (j.n_channels > 1 && j.in_chan_stride == 1 && j.out_chan_stride == 1 && j.n_clips <= 65535 &&
                             items8 * units <= 2147483647LL && !switches().fft_no_xcd_map) ? 1 : 0;
                if (a.chpair && !a.xcd_map) return "internal: channel pairing needs the XCD map"; // (cannot happen: cp_map_ok above)
                a.pairs_per_col = items;
                dim3 grid = a.xcd_map ? dim3((unsigned)(items8 * units), j.n_clips, 1)
                                      : dim3((unsigned)((n_blocks + 1) / 2), (unsigned)cols_p, 1);
                // unit-stride columns (mono / planar): buffer loads, staged aligned stores
                const bool v2ok = !a.xcd_map && !a.chpair && j.in_frame_stride == 1 && j.out_frame_stride == 1 &&
                                  2 * (size_t)g.hop_out * esz + 16 <= lds1;
                const bool cp2ok = a.chpair && a.xcd_map && cp2;
                // strided columns that are not channel pairs (odd channel counts, channel slices): two blocks of one column
                // per transform when the byte offsets of a pair of blocks fit the 32-bit operands
                const bool st2ok = !wide32 && !a.chpair && !v2ok &&
                                   2 * (int64_t)std::max(g.N_in, g.N_out) * std::max(j.in_frame_stride, j.out_frame_stride) * (int64_t)esz < (1LL << 30);
                // Throughput form (fftwave.hip): one wave per pair of 24-period blocks, two register passes per transform —
                // float32 unit-stride columns with enough pairs to fill the chip's 2048 wave slots four times over (below
                // that the last, partly filled round costs more than the form gains; and a single pair's latency is
                // longer than on the 6-wave workgroups of k_fft_pair2).
                FftWaveKernel wk;
                if (v2ok && !f64 && !switches().fft_no_wave && fft_wave_pick(p->L, p->M, &wk)) {
                    FftGeom gw;
                    if (const char *err = get(1000 + wk.k, wk.k, &gw)) return err;
                    const int64_t pairs_w = gw.ok ? ((j.out_frames + gw.hop_out - 1) / gw.hop_out + 1) / 2 : 0;
                    const int64_t wave_min = switches().dbg_wave_min ? switches().dbg_wave_min : wk.min_pairs;
                    if (gw.ok && gw.v0 == wk.v0 && gw.hop_out == wk.hop && gw.hop_periods == wk.hop_periods && pairs_w * (int64_t)cols_p >= wave_min &&
                        pairs_w * (int64_t)cols_p <= 2147483000LL) { // (item numbers are 32-bit; a larger job stays on k_fft_pair2)
                        set_geom(a, gw);
                        if (const char *e = fft_wave_launch(wk, a, (unsigned)pairs_w, (unsigned)cols_p, stream)) return e;
                        *handled = true;
                        return nullptr;
                    }
                }
                if (!v2ok && !cp2ok && !st2ok) return nullptr; // the general path (float32) or the exact engine
                // ragged batches: the unit-stride kernel reads its clip's row; nothing else does
                if (j.clip_table && !v2ok) return nullptr;
                void (*kern)(FftArgs) = nullptr;
                unsigned nt = use->nt;
                size_t lds = lds1;
                if (v2ok) {
                    kern = io64 ? use->kern2d : wide32 ? use->kern2fd : use->kern2;
                } else {
                    kern = cp2ok ? (f64 ? use->kcpd : use->kcp) : (f64 ? use->kstd : use->kst);
                }
                if (const char *e = ensure_dyn_lds((const void *)kern, lds)) return e;
#ifdef FFT2_TRACE
                size_t trace_n = 0;
                if (switches().dbg_trace) {
                    trace_n = (size_t)grid.x * grid.y * (nt / 64) * 16;
                    HIP_TRY(hipMalloc((void **)&a.trace, trace_n * 8));
                    HIP_TRY(hipMemset(a.trace, 0, trace_n * 8));
                }
#endif
                hipLaunchKernelGGL(kern, grid, dim3(nt), lds, (hipStream_t)stream, a);
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
    a.in = (const char *)j.in - j.in_abs0 * j.in_frame_stride * 4; a.out = j.out;
    a.WA = g.dev; a.WB = a.WA + g.A; a.P = a.WB + g.B; a.Q = a.P + (g.A + 1); a.Hs = a.Q + g.B;
    a.WA2 = a.Hs + (g.B + 1); a.WB2 = a.WA2 + g.N_in;
    a.Hr = reinterpret_cast<const float *>(a.WB2 + g.N_out); a.trace = nullptr;
    a.WA2d = a.WB2d = nullptr; a.Hrd = nullptr; a.clip_tab = nullptr;
    a.chpair = 0; a.pairs_per_col = 0; a.xcd_map = 0;
    a.A = g.A; a.B = g.B; a.nA = g.nA; a.nB = g.nB;
    for (int i = 0; i < 8; ++i) { a.radA[i] = g.radA[i]; a.radB[i] = g.radB[i]; }
    a.L = p->L; a.M = p->M;
    a.lead_periods = g.lead_periods; a.hop_periods = g.hop_periods; a.v0 = g.v0; a.hop_out = g.hop_out;
    a.n_clips = j.n_clips; a.n_channels = j.n_channels;
    a.ics = j.in_clip_stride; a.ifs = j.in_frame_stride; a.ichs = j.in_chan_stride;
    a.ocs = j.out_clip_stride; a.ofs = j.out_frame_stride; a.ochs = j.out_chan_stride;
    a.in_lo = j.in_abs0; a.in_frames = j.in_abs0 + j.in_frames; a.out_frames = j.out_frames;
    const int64_t n_blocks = (j.out_frames + g.hop_out - 1) / g.hop_out;
    const uint64_t cols = (uint64_t)j.n_clips * j.n_channels;
    if (cols > 65535) return "too many (clip, channel) columns for one launch (max 65535)";
    if (n_blocks > 2147483647LL) return "job too long for one launch";
    if (const char *e = ensure_dyn_lds((const void *)k_fft_block, std::max(g.lds_bytes, switches().dbg_fft_lds))) return e;
    const size_t dbg_lds = switches().dbg_fft_lds;
    hipLaunchKernelGGL(k_fft_block, dim3((unsigned)n_blocks, (unsigned)cols, 1), dim3(256), std::max(g.lds_bytes, dbg_lds),
                       (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    *handled = true;
    return nullptr;
}

#endif // host part (FFT_PART != 1)
} // namespace hipsoxr
