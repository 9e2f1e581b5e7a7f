// fft_dev.h — device helpers shared by the frequency-domain engine's translation units (fft.hip, fftwave.hip):
// complex arithmetic, small DFTs (radix 2-21 and their prime-factor / Cooley-Tukey composites), the kernel argument
// block and the raw-buffer load / store forms.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <type_traits>

namespace hipsoxr {

// ---------------------------------------------------------------------------------------------
// device: complex helpers and small DFTs (SIGN = -1 forward, +1 inverse, unnormalised).  Templates over the complex
// type C (float2 or double2): the float64 instances share every line.
// ---------------------------------------------------------------------------------------------
typedef float2 cf;
typedef double2 cd;
template <typename C> using real_of = decltype(C().x);
template <typename C> __device__ __forceinline__ C cadd(C a, C b) { return C(a.x + b.x, a.y + b.y); }
template <typename C> __device__ __forceinline__ C csub(C a, C b) { return C(a.x - b.x, a.y - b.y); }
// Packed float32 (round 6).  On gfx950 a scalar v_fma_f32 / v_fmac_f32 issues at HALF the rate of v_add / v_mul (4.0 against
// 2.3 cycles per wave-instruction and SIMD with the pipe full, tools/ubench/valu_issue.hip), while v_pk_fma_f32 does two
// FMAs per lane in 4.4-5.2: the chip's float32 FMA peak exists in the packed form only.  A third of the transforms' vector
// instructions are FMAs — complex products and the odd radices' constant sums — so those are written packed, on the
// (re, im) register pair a complex value already is; swizzles and signs ride in op_sel / neg_lo.  Adds and plain products
// stay scalar: v_pk_add / v_pk_mul take exactly two scalar issues.  (-DFFT_NO_PK: the scalar forms, A/B.)
typedef float v2f_t __attribute__((ext_vector_type(2)));
#ifndef FFT_NO_PK
#define FFT_PK 1
#else
#define FFT_PK 0
#endif
// a * b: (a.x b.x, a.y b.x), then (-a.y b.y, a.x b.y) added
__device__ __forceinline__ float2 pk_cmul(float2 a, float2 b)
{
    const v2f_t av = __builtin_bit_cast(v2f_t, a), bv = __builtin_bit_cast(v2f_t, b);
    v2f_t t, d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(av), "v"(bv));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(d) : "v"(av), "v"(bv), "v"(t));
    return __builtin_bit_cast(float2, d);
}
// ... with a compile-time constant k: the pair lives in two SGPRs (one constant-bus operand, both instructions)
__device__ __forceinline__ float2 pk_cmulk(float2 a, float2 k)
{
    const v2f_t av = __builtin_bit_cast(v2f_t, a), kv = __builtin_bit_cast(v2f_t, k);
    v2f_t t, d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(av), "s"(kv));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(d) : "v"(av), "s"(kv), "v"(t));
    return __builtin_bit_cast(float2, d);
}
// acc + k.x * a (HI = false) or acc + k.y * a (HI = true), k = a pair of real constants in two SGPRs; `first`: no accumulator
template <bool HI> __device__ __forceinline__ float2 pk_axpy(float2 a, float2 k, float2 acc)
{
    const v2f_t av = __builtin_bit_cast(v2f_t, a), kv = __builtin_bit_cast(v2f_t, k), cv = __builtin_bit_cast(v2f_t, acc);
    v2f_t d;
    if (HI) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(av), "s"(kv), "v"(cv));
    else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(av), "s"(kv), "v"(cv));
    return __builtin_bit_cast(float2, d);
}
template <typename C> __device__ __forceinline__ C cmul(C a, C b)
{
    if constexpr (FFT_PK && std::is_same<C, float2>::value) return pk_cmul(a, b);
    else return C(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// k: a compile-time constant.  Explicit fused forms with the NEGATED constant as the multiplier, so that every product is
// a VOP2 instruction with a literal (v_mul / v_fmac): left to itself the compiler turns `x * kx - y * ky` into
// v_fma(x, kx, -t), whose negated addend forces the VOP3 encoding with the constant in an SGPR — half issue rate on
// gfx950 (tools/ubench/valu_ops.hip).
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <typename C> __device__ __forceinline__ C cmulk(C a, C k)
{
    if constexpr (FFT_PK && std::is_same<C, float2>::value) return pk_cmulk(a, k);
    else { const real_of<C> nky = -k.y; return C(fma_(a.y, nky, a.x * k.x), fma_(a.y, k.x, a.x * k.y)); }
}
// k = (h, +-h) (an odd eighth of a turn): two sums and two products
template <typename C> __device__ __forceinline__ C cmul_h(C a, real_of<C> h, bool pos) { return pos ? C(h * (a.x - a.y), h * (a.x + a.y)) : C(h * (a.x + a.y), h * (a.y - a.x)); }
// a + SIGN i b,  a - SIGN i b
template <int SIGN, typename C> __device__ __forceinline__ C cadd_i(C a, C b) { return SIGN > 0 ? C(a.x - b.y, a.y + b.x) : C(a.x + b.y, a.y - b.x); }
template <int SIGN, typename C> __device__ __forceinline__ C csub_i(C a, C b) { return cadd_i<-SIGN>(a, b); }
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
    C s0 = cadd(a0, a2), d0 = csub(a0, a2), s1 = cadd(a1, a3), d1 = csub(a1, a3);
    a0 = cadd(s0, s1); a2 = csub(s0, s1); a1 = cadd_i<SIGN>(d0, d1); a3 = csub_i<SIGN>(d0, d1);
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
    C t3 = C(h * (-o3.x - sg * o3.y), h * (-o3.y + sg * o3.x));
    u[0] = cadd(e0, o0); u[4] = csub(e0, o0);
    u[1] = cadd(e1, t1); u[5] = csub(e1, t1);
    u[2] = cadd_i<SIGN>(e2, o2); u[6] = csub_i<SIGN>(e2, o2);
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
    // (w2 = (h, sg h), w4 = sg i, w6 = (-h, sg h) = sg i w2: the forms without a general complex product)
    const C w1 = C(c1, sg * s1), w3 = C(s1, sg * c1), w9 = C(-c1, -sg * s1);
    x[1][1] = cmulk(x[1][1], w1); x[1][2] = cmul_h(x[1][2], h, SIGN > 0); x[1][3] = cmulk(x[1][3], w3);
    x[2][1] = cmul_h(x[2][1], h, SIGN > 0); x[2][2] = cmuli<SIGN>(x[2][2]); x[2][3] = cmuli<SIGN>(cmul_h(x[2][3], h, SIGN > 0));
    x[3][1] = cmulk(x[3][1], w3); x[3][2] = cmuli<SIGN>(cmul_h(x[3][2], h, SIGN > 0)); x[3][3] = cmulk(x[3][3], w9);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        C v0 = x[0][b], v1 = x[1][b], v2 = x[2][b], v3 = x[3][b];
        dft4<SIGN>(v0, v1, v2, v3);
        u[b] = v0; u[b + 4] = v1; u[b + 8] = v2; u[b + 12] = v3;
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
        // (B starts from its first PRODUCT, not from 0 + product: `fma(sn, d, 0)` cannot be folded — signed zeros — and
        //  becomes a v_fma_f32 with the constant in an SGPR, which issues at half rate on gfx950; a plain product takes
        //  the constant as a literal)
        C A = x0, B;
#pragma unroll
        for (int t = 1; t <= Hh; ++t) {
            const T c = (T)__builtin_cos(PI2 * (double)((m * t) % R) / R);
            const T sn = (T)__builtin_sin(PI2 * (double)((m * t) % R) / R);
            if constexpr (FFT_PK && std::is_same<C, float2>::value) { // both sums as packed FMAs on the pair (c, sn)
                A = pk_axpy<false>(s[t - 1], C(c, sn), A);
                if (t == 1) { B.x = sn * d[0].x; B.y = sn * d[0].y; }
                else B = pk_axpy<true>(d[t - 1], C(c, sn), B);
            } else {
                A.x = fma_(c, s[t - 1].x, A.x); A.y = fma_(c, s[t - 1].y, A.y); // (explicit: `x + y * -k` would be rewritten as x - y * k, a VOP3 form again)
                if (t == 1) { B.x = sn * d[0].x; B.y = sn * d[0].y; }
                else { B.x = fma_(sn, d[t - 1].x, B.x); B.y = fma_(sn, d[t - 1].y, B.y); }
            }
        }
        out[m] = cadd_i<SIGN>(A, B);     // A + SIGN i B
        out[R - m] = csub_i<SIGN>(A, B);
    }
#pragma unroll
    for (int m = 0; m < R; ++m) u[m] = out[m];
}
template <int R, int SIGN, typename C> __device__ __forceinline__ void dft_r(C *u);

// Composite radix R1*R2 with coprime factors by the prime-factor (Good-Thomas) index maps: a plain R1 x R2
// two-dimensional DFT, no internal twiddles; the maps are compile-time constants (register renaming only).
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
// Composite radix R1*R2 by Cooley-Tukey with COMPILE-TIME twiddles (round 6: the radix-64 and radix-9 pieces of the
// one-wave-per-transform kernel, fftwave.hip): input n = R2 n1 + n2, output k = k1 + R1 k2,
// X[k1 + R1 k2] = sum_n2 w_R2^(n2 k2) [ w_N^(n2 k1) sum_n1 x[R2 n1 + n2] w_R1^(n1 k1) ].  Quarter turns are register
// renames; every other twiddle is one cmulk with literal constants.
template <int N, int SIGN, typename C> __device__ __forceinline__ C twk(C a, int e)
{
    typedef real_of<C> T;
    constexpr double PI2 = 6.283185307179586476925286766559;
    if (e == 0) return a;
    if (4 * e == N) return cmuli<SIGN>(a);
    if (2 * e == N) return C(-a.x, -a.y);
    if (4 * e == 3 * N) return cmuli<-SIGN>(a);
    return cmulk(a, C((T)__builtin_cos(PI2 * (double)e / N), (T)(SIGN * __builtin_sin(PI2 * (double)e / N))));
}
template <int R1, int R2, int SIGN, typename C> __device__ __forceinline__ void dft_ct(C *u)
{
    constexpr int N = R1 * R2;
    C x[R2][R1];
#pragma unroll
    for (int n2 = 0; n2 < R2; ++n2) {
#pragma unroll
        for (int n1 = 0; n1 < R1; ++n1) x[n2][n1] = u[R2 * n1 + n2];
        dft_r<R1, SIGN>(x[n2]);
        // (a 64-point butterfly holds 128 data registers: left to itself the scheduler interleaves all eight sub-butterflies
        //  for ILP and their temporaries spill; one sub-butterfly at a time has ILP 8 and needs no more)
        if (N >= 64) __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int k1 = 0; k1 < R1; ++k1) {
        C c[R2];
#pragma unroll
        for (int n2 = 0; n2 < R2; ++n2) c[n2] = twk<N, SIGN>(x[n2][k1], (n2 * k1) % N);
        dft_r<R2, SIGN>(c);
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) u[k1 + R1 * k2] = c[k2];
        if (N >= 64) __builtin_amdgcn_sched_barrier(0);
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
    else if constexpr (R == 9) dft_ct<3, 3, SIGN>(u);
    else if constexpr (R == 56) dft_pfa<8, 7, SIGN>(u);
    else if constexpr (R == 60) dft_pfa<4, 15, SIGN>(u);
    else if constexpr (R == 63) dft_pfa<9, 7, SIGN>(u);
    else if constexpr (R == 64) dft_ct<8, 8, SIGN>(u);
    else dft_odd<R, SIGN>(u);
}

struct FftArgs {
    const void *in;
    void *out;
    const float2 *WA, *WB, *P, *Q, *Hs; // k_fft_block: twiddles of both half-length transforms, (un)tangling twiddles, filter
    const float2 *WA2, *WB2;            // paired kernels: twiddles of the full-length transforms
    const float *Hr;                    // ... the filter as REAL values (see fft_build)
    const double2 *WA2d, *WB2d;         // ... the same tables in float64
    const double *Hrd;
    unsigned long long *trace;          // HIPSOXR_DEBUG_TRACE (builds with -DFFT2_TRACE only): per-wave s_memtime stamps [wg][wave][16]
    int32_t A, B;            // k_fft_block: complex transform lengths N_in/2, N_out/2
    int32_t nA, nB;          // ... number of passes
    int32_t radA[8], radB[8];
    int64_t L, M;
    int32_t lead_periods, hop_periods; // block b covers periods [b*hop - lead, ...): k periods long
    int32_t v0, hop_out;     // first kept local output, outputs kept per block
    uint32_t n_clips, n_channels;
    int64_t ics, ifs, ichs, ocs, ofs, ochs;
    int64_t in_frames, out_frames;
    int64_t in_lo;           // the column holds samples [in_lo, in_frames) (hipsoxr_job_t::in_abs0: `in` points at sample 0, zero outside); 0 for ragged batches
    const int64_t *clip_tab; // ragged batch (hipsoxr_job_t::clip_table_dev): [n_clips][4] = in offset, in frames, out offset, out frames; k_fft_pair2 only
    int32_t chpair;          // k_fft_strided2: 1 = pair neighbouring channels of interleaved data instead of blocks
    int64_t pairs_per_col;   // xcd_map: work items (blocks, or pairs of blocks) per channel unit
    int32_t xcd_map;         // interleaved multi-channel data: XCD-aware workgroup ids (see k_fft_strided2)
};

// fftwave.hip: the one-wave-per-block-pair kernel of a ratio — block size (periods), the geometry it is compiled for,
// the kernel; launched on (pairs of blocks, columns) x 64 threads.
struct FftWaveKernel { int64_t L, M; int k, v0, hop, hop_periods; int64_t min_pairs; const void *kern; }; // min_pairs: the job size it wins from
bool fft_wave_pick(int64_t L, int64_t M, FftWaveKernel *out);
const char *fft_wave_launch(const FftWaveKernel &k, const FftArgs &a, unsigned pairs, unsigned cols, void *stream);

__device__ __forceinline__ void *uniform_ptr(void *p) // the same address, provably wave-uniform (two v_readfirstlane)
{
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    return reinterpret_cast<void *>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32) |
                                    (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v));
}
// Cache policy of the signal's loads and stores (the `aux` operand of the raw buffer builtins: 1 = sc0, 2 = nt, 16 = sc1).
// k_fft_pair2's staged run is written once, in whole 16-byte granules of whole lines, and not read again by the launch:
// NON-TEMPORAL stores keep it out of the L2's and the Infinity Cache's way — and the batch launch, which runs AT the
// board's 1.4 kW power cap (tools/power_probe.py: 1370 W, shader clock throttled from 2.4 to 2.05 GHz), gets 10-14 %
// faster for 12 % less energy per launch (122 -> 109 us on one box, 122 -> 104.6 on another; sc0 0 %, sc1 +4 %,
// sc1 nt -5 %); float64 jobs -10 %, the 60 s clip -4 %.  Only there: stores that write PART of a line per instruction
// want the L2's write combining — k_fft_strided2's 8-byte words at a frame stride +17 % with nt (configs[2] 45.7 -> 53.7
// us), the exact engine's 4-byte stores +3 .. +87 % (tools/nt_ab.sh).  Loads: nt +9 % (neighbouring blocks share their
// overlap through the caches), sc0 / sc1 0 %: default policy.  profiles/r04_cache_policy.txt.
#ifndef FFT_LOAD_AUX
#define FFT_LOAD_AUX 0
#endif
#ifndef FFT_STORE_AUX
#define FFT_STORE_AUX 2
#endif

typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
template <typename Real> __device__ __forceinline__ Real buf_load_real(__amdgpu_buffer_rsrc_t r, int voff, int soff);
template <> __device__ __forceinline__ float buf_load_real<float>(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, FFT_LOAD_AUX));
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

} // namespace hipsoxr
