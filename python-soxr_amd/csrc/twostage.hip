// twostage.hip — arbitrary rate ratios at reduced cost: the frequency-domain engine + a short polyphase stage.
//
// Ratios outside the exact-bank range (random integer or float rates: reference tests/test_random.py:21-25) get an
// interpolated-phase plan, and the canonical-order engine evaluates its 296-tap (VHQ) direct form with a cubic per tap:
// 48000 -> 44101 stereo 60 s took 402 us against 70 us for its rational neighbour.  libsoxr itself serves such ratios at
// near-constant cost with an FFT stage at a fixed ratio plus a short interpolated polyphase stage (SURVEY.md §0.3 / §A.4,
// upstream-unverified).  The MI355X form, for whole-signal float32 / float64 device jobs under HIPSOXR_KERNEL_AUTO /
// _FFT (1e-6 class; the host surface, streams and integer I/O keep the bit-exact k_interp / k_interp_tile):
//
//   up   (out > in):  x --[FFT engine 1:2, the plan's OWN prototype sampled on the half-sample grid]--> u at 2 f_in
//                       --[k_poly: T2-tap interpolated polyphase, transparent on [0, f_in/2], stop from 1.5 f_in]--> y
//   down (out < in):  x --[k_poly: transparent on [0, f_out/2], stop from 1.5 f_out]--> v at 2 f_out
//                       --[FFT engine 2:1, the plan's own prototype re-expressed at 2 f_out]--> y
//
// The sharp filter of the composite IS the plan's prototype h (plan_proto), so the result equals the single-stage
// filter's to the polyphase stage's ripple (designed 6 dB below the recipe) and the FFT engine's own floor — on any input,
// not only band-limited ones: <= 1e-6 of the oracle's float64 direct form (tests/test_gpu_random_rates.py).
// The polyphase stage's cubic table ([P2][T2] records of 16 bytes: 12-50 KB) lives in LDS, which ends the 423 MB stream of
// coefficient records through the scalar cache that bounds k_interp_tile.
// Ends: the intermediate signal is produced a few samples PAST both ends of the job (as far as the second stage reads it;
// hipsoxr_job_t::in_abs0 places it for the FFT engine), so every output, the first and last included, is the composite
// filter's response to the zero-extended signal: two launches in all.  A thread of k_poly keeps its source window in
// registers from one output to the next (see MQ), so the table records are the stage's only per-tap LDS traffic.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <mutex>
#include <vector>

#include "device.h"

namespace hipsoxr {

#define HIP_TRY(expr)                                       \
    do {                                                    \
        hipError_t e_ = (expr);                             \
        if (e_ != hipSuccess) return hipGetErrorString(e_); \
    } while (0)

struct TwoStage {
    bool ok = false, up = false;
    Plan fft;                 // the FFT stage's plan: L/M = 2/1 (up) or 1/2 (down), bank from the owner's prototype
    int32_t T2 = 0, P2 = 0, P2f = 0, row = 0; // polyphase stage: taps, table intervals (float64 / float32 table), records per table row (T2 + 1: bank spreading)
    int64_t Ls = 1, Ms = 1;   // polyphase stage: output k sits at k * Ms / Ls of ITS input samples
    mutable int lane_for[3][16] = {};   // [float32 / float64 / float32 pairs][R]: lane multiplier of k_poly for runs of R outputs per thread (launch_poly; 0 = not simulated yet)
    mutable float conf_for[3][16] = {}; // ... and its LDS conflict cost (1 = every 16-byte read group in one cycle)
    void *tab_f = nullptr, *tab_d = nullptr; // device: [P2f][row] float4 / [P2][row] double4 records (a0..a3 of the cubic in x in [0, 1))
};

void twostage_release(Plan *p)
{
    if (!p->two) return;
    if (p->two->tab_f) (void)hipFree(p->two->tab_f);
    if (p->two->tab_d) (void)hipFree(p->two->tab_d);
    delete p->two; // (the FFT stage's plan releases its own device tables: Plan::~Plan)
    p->two = nullptr;
}

static int64_t gcd64(int64_t a, int64_t b)
{
    while (b) { const int64_t t = a % b; a = b; b = t; }
    return a;
}

// Kaiser-windowed sinc of the polyphase stage in units of ITS input samples: cut-off fc (cycles per sample), half width W
struct PolyProto {
    double fc, W, beta, inv_i0, scale;
    double operator()(double tau) const
    {
        const double u = tau / W;
        double w = 1. - u * u;
        if (w <= 0.) return 0.;
        const double s = tau == 0. ? 2. * fc : std::sin(2. * M_PI * fc * tau) / (M_PI * tau);
        return s * bessel_i0(beta * std::sqrt(w)) * inv_i0 * scale;
    }
};

static const char *twostage_build(Plan *p)
{
    TwoStage *ts = new TwoStage;
    p->two = ts;
    if (!p->phases || p->q.bits < 20. || p->proto_scale == 0. || p->custom_bank) return nullptr; // HQ / VHQ windowed-sinc interpolated plans only
    const double fi = p->in_rate, fo = p->out_rate;
    ts->up = fo > fi;
    const double rho = ts->up ? 1. : fi / (2. * fo); // polyphase stage: input samples per output sample is rho (down) / Ms/Ls (up)
    if (!ts->up && fi / fo > 4.) return nullptr;      // (long polyphase filters: the table would leave LDS)
    // ---- the FFT stage's plan: the owner's prototype H(tau) (tau in the OWNER's input samples) on the 2x grid ----
    Plan &f = ts->fft;
    f.recipe = p->recipe; f.q = p->q; f.att_db = p->att_db; f.beta = p->beta; f.phases = 0;
    if (ts->up) { // u[m] = sum_n x[n] H(m/2 - n): bank[ph][j] = H(ph/2 + T/2 - 1 - j)
        f.in_rate = fi; f.out_rate = 2. * fi; f.L = 2; f.M = 1; f.T = p->T;
        f.bank.assign((size_t)2 * f.T, 0.);
        for (int ph = 0; ph < 2; ++ph)
            for (int j = 0; j < f.T; ++j) f.bank[(size_t)ph * f.T + j] = plan_proto(*p, .5 * ph + (double)(f.T / 2 - 1 - j));
    } else {      // y[k] = sum_m v[m] g(2k - m), g(s) = H(s rho) rho (v at 2 f_out: rho owner-input samples per v sample)
        f.in_rate = 2. * fo; f.out_rate = fo; f.L = 1; f.M = 2;
        const int64_t tb = (int64_t)std::ceil((double)p->T / rho) + 2;
        f.T = (int32_t)((tb + 7) / 8 * 8);
        f.bank.assign((size_t)f.T, 0.);
        for (int j = 0; j < f.T; ++j) f.bank[j] = plan_proto(*p, (double)(f.T / 2 - 1 - j) * rho) * rho;
    }
    // ---- the polyphase stage: transparent over the band the FFT stage passes, stop band where its images begin ----
    // in cycles per sample of ITS input: up: input at 2 f_in: pass 0.25 (= f_in/2), stop 0.75; down: input at f_in: pass
    // f_out/2, stop 1.5 f_out
    const double fpass = ts->up ? .25 : .5 * fo / fi, fstop = ts->up ? .75 : 1.5 * fo / fi;
    const double A = p->att_db + 6.;
    const double n_taps = (A - 7.95) / (2.285 * 2. * M_PI * (fstop - fpass)) + 1.;
    ts->T2 = 0;
    for (int t : {8, 12, 16, 20, 24, 28, 32, 40, 48, 56}) // (HIPSOXR_POLY_TAPS: the kernel's instances)
        if (!ts->T2 && t >= (int)std::ceil(n_taps)) ts->T2 = t;
    if (!ts->T2) return nullptr;
    // intervals of the cubic table: the interpolation error of a Chebyshev cubic over 1/P of a sample of this prototype is
    // about 0.03 / P^4 of full scale: 1.2e-10 at 128 (float64 table), 1.9e-9 at 64 (float32 table: two orders under the
    // float32 engine's floor, half the LDS — which is what lets several workgroups share a CU)
    ts->P2 = 128; ts->P2f = 64;
    ts->row = ts->T2 + 1;
    if ((size_t)ts->P2 * ts->row * 16 > 100 * 1024) return nullptr;
    // output k of the polyphase stage at k * Ms / Ls input samples: up: from 2 f_in to f_out: 2 M / L; down: from f_in to
    // 2 f_out: M / (2 L)
    {
        int64_t a = ts->up ? 2 * p->M : p->M, b = ts->up ? p->L : 2 * p->L;
        const int64_t g = gcd64(a, b);
        ts->Ms = a / g; ts->Ls = b / g;
        if (ts->Ms > (1LL << 31) || ts->Ls > (1LL << 31)) return nullptr;
    }
    PolyProto h;
    h.fc = .5 * (fpass + fstop); h.W = .5 * ts->T2; h.beta = .1102 * (A - 8.7); h.inv_i0 = 1. / bessel_i0(h.beta); h.scale = 1.;
    {
        double sum = 0.;
        for (int64_t m = -32 * ts->T2; m < 32 * ts->T2; ++m) sum += h((double)m / 64.);
        h.scale = 64. / sum;
    }
    // cubic per (interval, tap) through the four Chebyshev nodes of the interval, monomials in x in [0, 1) (plan.cpp)
    double node[4];
    for (int c = 0; c < 4; ++c) node[c] = .5 - .5 * std::cos((double)(2 * c + 1) * M_PI / 8.);
    auto build = [&](int P) {
        std::vector<double> tab((size_t)P * ts->row * 4, 0.);
        for (int i = 0; i < P; ++i)
            for (int j = 0; j < ts->T2; ++j) {
                double v[4];
                for (int c = 0; c < 4; ++c) v[c] = h(((double)i + node[c]) / P + (double)(ts->T2 / 2 - 1 - j));
                const double d01 = (v[1] - v[0]) / (node[1] - node[0]), d12 = (v[2] - v[1]) / (node[2] - node[1]), d23 = (v[3] - v[2]) / (node[3] - node[2]);
                const double d012 = (d12 - d01) / (node[2] - node[0]), d123 = (d23 - d12) / (node[3] - node[1]);
                const double d3 = (d123 - d012) / (node[3] - node[0]);
                double *a = &tab[((size_t)i * ts->row + j) * 4];
                a[3] = d3;
                a[2] = d012 - d3 * (node[0] + node[1] + node[2]);
                a[1] = d01 - d012 * (node[0] + node[1]) + d3 * (node[0] * node[1] + node[0] * node[2] + node[1] * node[2]);
                a[0] = v[0] - d01 * node[0] + d012 * node[0] * node[1] - d3 * node[0] * node[1] * node[2];
            }
        return tab;
    };
    const std::vector<double> tab = build(ts->P2), tab64 = build(ts->P2f);
    std::vector<float> tabf(tab64.size());
    for (size_t i = 0; i < tab64.size(); ++i) tabf[i] = (float)tab64[i];
    HIP_TRY(hipMalloc(&ts->tab_f, tabf.size() * sizeof(float)));
    HIP_TRY(hipMemcpy(ts->tab_f, tabf.data(), tabf.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc(&ts->tab_d, tab.size() * sizeof(double)));
    HIP_TRY(hipMemcpy(ts->tab_d, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
    ts->ok = true;
    return nullptr;
}

// ---------------------------------------------------------------------------------------------
// k_poly: outputs [k_lo, k_lo + n_out) of one column per workgroup tile; output k at src position k * Ms / Ls;
//   y[k] = sum_j c_j(frac) src[floor(pos) - (T/2 - 1) + j],   c_j(f) = cubic of table[floor(f P)][j] at x = f P - floor(f P)
// The table and the tile's source span live in LDS; a thread owns R consecutive outputs (position by one 64-bit division,
// then increments).
// ---------------------------------------------------------------------------------------------
struct PolyArgs {
    const void *src; void *dst; const void *tab;
    int32_t T, P, row, R, span_max, lane_mul, lgP;
    int32_t lg_cg;     // a workgroup takes 2^lg_cg neighbouring channels of a tile one after the other (interleaved data: their parts of a line meet in one L2)
    unsigned long long *trace; // -DPOLY_TRACE builds: per-wave cycle sums [workgroup][wave][8]
    uint64_t step_fx;  // frac(Ms / Ls) in units of 2^-64 (float32 path of k_poly)
    double fx_per_rem; // 2^64 / Ls
    int64_t Ls, Ms, Mq, Mr; // Ms = Mq * Ls + Mr
    int64_t n_lo, n_src, k_lo, n_out; // the source column holds samples [n_lo, n_src) (src points at sample 0; zero outside); outputs [k_lo, k_lo + n_out), k_lo may be < 0
    int64_t scs, sfs, schs, dcs, dfs, dchs;
    uint32_t n_channels;
    // k_poly2<.., IL = false>: member 2 of a split column — element offsets of its sample n / output k from the column's,
    // its sample n = the column's n + m2_shift, and how many outputs it has (member 1 has n_out)
    int64_t m2_src, m2_dst, m2_shift, m2_n_out;
};

// tap counts the polyphase kernel is instantiated for (twostage_build rounds its design up to the next one)
#define HIPSOXR_POLY_TAPS(X) X(8) X(12) X(16) X(20) X(24) X(28) X(32) X(40) X(48) X(56)
#define HIPSOXR_POLY2_TAPS(X) X(8) X(12) X(16) X(20) X(24) X(28) X(32) X(40) // k_poly2 (two windows in registers: longer ones spill)
__device__ __forceinline__ int64_t floor_div(int64_t q, int64_t d) // d > 0
{
    const int64_t n = q / d;
    return n * d > q ? n - 1 : n;
}
template <typename Real> struct Rec4;
template <> struct Rec4<float> { typedef float4 type; };
template <> struct Rec4<double> { typedef double4 type; };

// TT = taps (compile time: the tap loop unrolls and every LDS read of an output is in flight at once).
// MQ = floor(Ms / Ls) when that is 0 or 1 (every ratio the two-stage form takes but down-sampling beyond 2:1): consecutive
// outputs' source windows then differ by MQ or MQ + 1 samples, and the thread keeps its window in REGISTERS, shifted by
// v_cndmask between outputs, with two fresh samples read per output instead of TT — the table records are then the only
// per-tap LDS traffic.  MQ = -1: the window is re-read from LDS for every output.
template <typename Real, int TT, int MQ>
__global__ void __launch_bounds__(256) k_poly(PolyArgs a)
{
    typedef typename Rec4<Real>::type R4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    R4 *tab = reinterpret_cast<R4 *>(smem);
    Real *xs = reinterpret_cast<Real *>(smem + (size_t)a.P * a.row * sizeof(R4));
    Real *ys = xs + a.span_max; // the tile's outputs, staged [R][257] so that they leave as whole lines whatever the lane order
    const int cg = 1 << a.lg_cg;
    const uint32_t col = blockIdx.y << a.lg_cg, ch = col % a.n_channels, clip = col / a.n_channels; // first channel of the group
    const Real *src0 = (const Real *)a.src + (int64_t)clip * a.scs + (int64_t)ch * a.schs;
    Real *dst = (Real *)a.dst + (int64_t)clip * a.dcs + (int64_t)ch * a.dchs;
    const int tid = (int)threadIdx.x;
    {
        const R4 *g = (const R4 *)a.tab;
        for (int i = tid; i < a.P * a.row; i += 256) tab[i] = g[i];
    }
    const int64_t per_tile = 256LL * a.R, n_tiles = (a.n_out + per_tile - 1) / per_tile;
    constexpr int H = TT / 2;
    constexpr int CH = TT % 16 == 0 ? 16 : TT % 12 == 0 ? 12 : TT % 8 == 0 ? 8 : 4; // taps whose LDS reads are in flight together
#ifdef POLY_TRACE
    unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tq = __builtin_amdgcn_s_memtime();
    const unsigned long long tq0 = tq;
#define POLY_STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tr[i] += t_ - tq; tq = t_; } while (0)
#else
#define POLY_STAMP(i) ((void)0)
#endif
    constexpr bool FAST = sizeof(Real) == 4 && MQ >= 0;
    // float32 path: output k sits at k_lo's exact position (one 64-bit division per thread and launch) plus (k - k_lo) steps of
    // Ms / Ls as a 64.64 binary fraction (step_fx is truncated: at most 2^-33 of a sample short after 2^31 outputs); the tile's
    // span and every thread's window come from the same arithmetic
    int64_t nK0 = 0;
    uint64_t phK0 = 0;
    if constexpr (FAST) {
        nK0 = floor_div(a.k_lo * a.Ms, a.Ls);
        phK0 = (uint64_t)((double)(a.k_lo * a.Ms - nK0 * a.Ls) * a.fx_per_rem); // remainder / Ls in units of 2^-64
    }
    auto position = [&](int64_t k, uint64_t &ph) -> int64_t {
        const uint64_t dk = (uint64_t)(k - a.k_lo), lo = dk * a.step_fx;
        ph = phK0 + lo;
        return nK0 + (int64_t)dk * a.Mq + (int64_t)__umul64hi(dk, a.step_fx) + (ph < lo ? 1 : 0);
    };
    const int slot = (tid * a.lane_mul) & 255; // this thread's run of R outputs within a tile
    constexpr int NPF = FAST ? 8 : 0; // float32 path: samples per thread of the NEXT tile's span held in registers
    Real pf[NPF > 0 ? NPF : 1];
    auto fetch = [&](int64_t tile, int c) {
        const Real *src = src0 + (int64_t)c * a.schs;
        const int64_t kA = a.k_lo + tile * per_tile, kEnd = a.k_lo + a.n_out, kB = (kA + per_tile < kEnd ? kA + per_tile : kEnd) - 1;
        uint64_t ph;
        const int64_t nA = position(kA, ph) - (H - 1), nB = position(kB, ph) + H;
#pragma unroll
        for (int q = 0; q < NPF; ++q) {
            const int64_t n = nA + q * 256 + tid;
            pf[q] = (n <= nB && n >= a.n_lo && n < a.n_src) ? src[n * a.sfs] : (Real)0;
        }
    };
    if constexpr (FAST)
        if ((int64_t)blockIdx.x < n_tiles) fetch(blockIdx.x, 0);
    // work items: (tile, channel of the group), channels innermost
    for (int64_t item = 0;; ++item) {
        const int64_t tile = blockIdx.x + (item >> a.lg_cg) * gridDim.x;
        if (tile >= n_tiles) break;
        const int c = (int)(item & (cg - 1));
        const Real *src = src0 + (int64_t)c * a.schs;
        Real *ysc = ys;
        const int64_t kA = a.k_lo + tile * per_tile;
        const int64_t kEnd = a.k_lo + a.n_out, kB = (kA + per_tile < kEnd ? kA + per_tile : kEnd) - 1;
        // (|k| < 2^31 and Ms <= 2^31: the products fit 64 bits — launch_two_stage admits no larger job)
        int64_t nA, nB;
        if constexpr (FAST) {
            uint64_t ph;
            nA = position(kA, ph) - (H - 1); nB = position(kB, ph) + H;
        } else {
            nA = floor_div(kA * a.Ms, a.Ls) - (H - 1); nB = floor_div(kB * a.Ms, a.Ls) + H;
        }
        const int span = (int)(nB - nA + 1);
        if constexpr (FAST) { // the first NPF * 256 samples were fetched while the tile before was being computed
#pragma unroll
            for (int q = 0; q < NPF; ++q)
                if (q * 256 + tid < span) xs[q * 256 + tid] = pf[q];
        }
        for (int i = NPF * 256 + tid; i < span; i += 256) {
            const int64_t n = nA + i;
            xs[i] = (n >= a.n_lo && n < a.n_src) ? src[n * a.sfs] : (Real)0;
        }
        POLY_STAMP(0); // span staged (loads issued and written)
        __syncthreads();
        POLY_STAMP(1); // barrier
        if constexpr (FAST) { // the next item's span: in flight behind this item's arithmetic
            const int64_t tn = blockIdx.x + ((item + 1) >> a.lg_cg) * gridDim.x;
            if (tn < n_tiles) fetch(tn, (int)((item + 1) & (cg - 1)));
        }
        const int64_t k1 = kA + (int64_t)slot * a.R;
        if constexpr (sizeof(Real) == 4 && MQ >= 0) {
            // float32, window in registers: the position is a 64-bit binary fraction stepped by frac(Ms / Ls) 2^64 (its
            // carry moves the window on), the sums over taps are taken per cubic coefficient — S_d = sum_j a_d[j] u[j],
            // two coefficients per v_pk_fma_f32 straight from the 16-byte record, the sample broadcast by op_sel — and
            // y = S0 + x (S1 + x (S2 + x S3)): two packed instructions per tap instead of four scalar ones.
            typedef float v2f __attribute__((ext_vector_type(2)));
            typedef float v4f __attribute__((ext_vector_type(4)));
            if (k1 <= kB) {
                uint64_t phase;
                const int64_t n0 = position(k1, phase);
                const float *w = xs + (n0 - nA - (H - 1));
                v2f u[TT / 2];
#pragma unroll
                for (int j = 0; j < TT / 2; ++j) { u[j].x = w[2 * j]; u[j].y = w[2 * j + 1]; }
                const float *wtop = w + (TT - 2);
                const v4f *tabv = reinterpret_cast<const v4f *>(tab);
                float *yo = ysc + slot; // staged [run index][slot] with rows of 257: conflict-free writes
                const int lg = a.lgP;
                int left = (int)(kB - k1 + 1 < (int64_t)a.R ? kB - k1 + 1 : (int64_t)a.R);
                for (; left > 0; --left) {
                    const uint32_t hi = (uint32_t)(phase >> 32);
                    const uint32_t i = hi >> (32 - lg);
                    const float x = (float)(uint32_t)(hi << lg) * 0x1p-32f;
                    const v4f *row = tabv + i * (uint32_t)a.row;
                    v2f s01e = {0.f, 0.f}, s23e = {0.f, 0.f}, s01o = {0.f, 0.f}, s23o = {0.f, 0.f};
#pragma unroll
                    for (int j0 = 0; j0 < TT; j0 += CH) {
                        v4f c[CH];
#pragma unroll
                        for (int j = 0; j < CH; ++j) c[j] = row[j0 + j];
#pragma unroll
                        for (int j = 0; j < CH; j += 2) {
                            const v2f a01 = __builtin_shufflevector(c[j], c[j], 0, 1), a23 = __builtin_shufflevector(c[j], c[j], 2, 3);
                            const v2f b01 = __builtin_shufflevector(c[j + 1], c[j + 1], 0, 1), b23 = __builtin_shufflevector(c[j + 1], c[j + 1], 2, 3);
                            const v2f up = u[(j0 + j) / 2];
                            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(s01e) : "v"(a01), "v"(up), "v"(s01e));
                            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(s23e) : "v"(a23), "v"(up), "v"(s23e));
                            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(s01o) : "v"(b01), "v"(up), "v"(s01o));
                            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(s23o) : "v"(b23), "v"(up), "v"(s23o));
                        }
                    }
                    const float S0 = s01e.x + s01o.x, S1 = s01e.y + s01o.y, S2 = s23e.x + s23o.x, S3 = s23e.y + s23o.y;
                    *yo = ((S3 * x + S2) * x + S1) * x + S0;
                    yo += 257;
                    const uint64_t next = phase + a.step_fx;
                    const bool adv = next < phase; // the fraction wrapped: one more sample
                    phase = next;
                    wtop += MQ + (adv ? 1 : 0);
                    const float e0 = wtop[0], e1 = wtop[1]; // (the span has 4 spare words behind the last window)
                    float f[TT];
#pragma unroll
                    for (int j = 0; j < TT / 2; ++j) { f[2 * j] = u[j].x; f[2 * j + 1] = u[j].y; }
#pragma unroll
                    for (int j = 0; j < TT - 2; ++j) f[j] = adv ? f[j + MQ + 1] : f[j + MQ];
                    f[TT - 2] = e0; f[TT - 1] = e1;
#pragma unroll
                    for (int j = 0; j < TT / 2; ++j) { u[j].x = f[2 * j]; u[j].y = f[2 * j + 1]; }
                }
            }
        } else
        if (k1 <= kB) {
            int64_t n = floor_div(k1 * a.Ms, a.Ls);
            int64_t rem = k1 * a.Ms - n * a.Ls;
            const double invL = 1. / (double)a.Ls;
            Real u[TT];
            if constexpr (MQ >= 0) {
                const Real *w = xs + (n - nA - (H - 1));
#pragma unroll
                for (int j = 0; j < TT; ++j) u[j] = w[j];
            }
            for (int r = 0; r < a.R && k1 + r <= kB; ++r) {
                const double fP = (double)rem * invL * (double)a.P;
                int i = (int)fP;
                if (i >= a.P) i = a.P - 1;
                const Real x = (Real)(fP - (double)i);
                const R4 *row = tab + (size_t)i * a.row;
                const Real *w = xs + (n - nA - (H - 1));
                Real acc0 = (Real)0, acc1 = (Real)0;
#pragma unroll
                for (int j0 = 0; j0 < TT; j0 += CH) {
                    R4 c[CH];
                    Real v[CH];
#pragma unroll
                    for (int j = 0; j < CH; ++j) {
                        c[j] = row[j0 + j];
                        if constexpr (MQ >= 0) v[j] = u[j0 + j];
                        else v[j] = w[j0 + j];
                    }
#pragma unroll
                    for (int j = 0; j < CH; j += 2) {
                        acc0 += (((c[j].w * x + c[j].z) * x + c[j].y) * x + c[j].x) * v[j];
                        acc1 += (((c[j + 1].w * x + c[j + 1].z) * x + c[j + 1].y) * x + c[j + 1].x) * v[j + 1];
                    }
                }
                ysc[r * 257 + slot] = acc0 + acc1;
                n += a.Mq; rem += a.Mr;
                const bool adv = rem >= a.Ls;
                if (adv) { rem -= a.Ls; ++n; }
                if constexpr (MQ >= 0) { // the window moves on by MQ or MQ + 1 samples (the span has 4 spare words behind the last window)
                    const Real *wn = xs + (n - nA - (H - 1));
                    const Real e0 = wn[TT - 2], e1 = wn[TT - 1];
#pragma unroll
                    for (int j = 0; j < TT - 2; ++j) u[j] = adv ? u[j + MQ + 1] : u[j + MQ];
                    u[TT - 2] = e0; u[TT - 1] = e1;
                }
            }
        }
        POLY_STAMP(2); // outputs computed
        __syncthreads(); // the tile's outputs are staged (and its source span is free for the next tile)
        POLY_STAMP(3); // barrier
        { // (the channels of a group leave one after the other from the same workgroup: their halves of a line meet in its XCD's L2)
            Real *yo = dst + (kA - a.k_lo) * a.dfs + (int64_t)c * a.dchs;
            const int cnt = (int)(kB - kA + 1);
            const float invR = 1.f / (float)a.R;
            for (int i = tid; i < cnt; i += 256) {
                const int sl = (int)(((float)i + .5f) * invR), r = i - sl * a.R; // output i of the tile: run index r of slot sl
                yo[(int64_t)i * a.dfs] = ys[r * 257 + sl];
            }
        }
        POLY_STAMP(4); // outputs stored
    }
#ifdef POLY_TRACE
    if (a.trace && (tid & 63) == 0) {
        unsigned long long *o = a.trace + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + tid / 64) * 8;
        for (int i = 0; i < 5; ++i) o[i] = tr[i];
        o[5] = tq0; o[6] = tq; o[7] = __builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32); // HW_ID, XCC_ID
    }
#endif
}

// k_poly2: the float32 register-window form of k_poly for a PAIR of interleaved channels (both ends of the stage channel-
// interleaved, e.g. stereo): a thread produces R consecutive FRAMES of the pair.  The two channels of a frame share the
// position, the table row and the cubic's argument, so one set of 16-byte record reads serves both (half the per-tap LDS
// traffic of two k_poly passes), the packed FMAs run over the channel pair — s_d += a_d[j] * (left, right)[j], the
// coefficient broadcast by op_sel — and source frames / output frames move as 8-byte words (whole lines in ONE pass over
// the data instead of half lines in two).  Same arithmetic per channel as k_poly's (sums per cubic coefficient, then
// Horner), in tap order instead of even / odd halves: results agree to rounding.
// IL = false — any OTHER column (mono, planar, odd channel counts): the pair is two SEGMENTS OF ONE COLUMN a whole number of
// phase periods apart.  Output k + h Ls sits exactly h Ms source samples behind output k (positions are k Ms / Ls), so the
// two have the same fraction — the same table row and argument — for every k: member 2 of the pair is the column from
// output h Ls on (PolyArgs::m2_*; launch_poly picks h = half the job's periods), read and written as 4-byte words.
template <int TT, int MQ, bool IL>
#ifndef POLY2_OCC
#define POLY2_OCC 3
#endif
__global__ void __launch_bounds__(256, TT >= 32 ? 2 : POLY2_OCC) k_poly2(PolyArgs a) // (two windows of 32+ frames: 2 workgroups per CU, launch_poly sizes the tile for that)
{
    typedef float v2f __attribute__((ext_vector_type(2)));
    typedef float v4f __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    v4f *tab = reinterpret_cast<v4f *>(smem);
    v2f *xs = reinterpret_cast<v2f *>(smem + (size_t)a.P * a.row * sizeof(v4f));
    v2f *ys = xs + a.span_max; // the tile's output frames, staged [R][257]
    const int cg = 1 << a.lg_cg; // channel PAIRS (IL) / channels a workgroup takes one after the other
    const uint32_t col = (blockIdx.y << a.lg_cg) * (IL ? 2 : 1), ch = col % a.n_channels, clip = col / a.n_channels; // first channel of the group
    const int64_t c_src = IL ? 2 : a.schs, c_dst = IL ? 2 : a.dchs; // from one item of the group to the next
    const float *src0 = (const float *)a.src + (int64_t)clip * a.scs + (int64_t)ch * (IL ? 1 : a.schs);
    float *dst0 = (float *)a.dst + (int64_t)clip * a.dcs + (int64_t)ch * (IL ? 1 : a.dchs);
    const int tid = (int)threadIdx.x;
    const int64_t per_tile = 256LL * a.R, n_tiles = (a.n_out + per_tile - 1) / per_tile;
    constexpr int H = TT / 2;
#ifdef POLY2_CH
    constexpr int CH = TT % POLY2_CH == 0 ? POLY2_CH : 4;
#else
    constexpr int CH = TT % 16 == 0 ? 16 : TT % 12 == 0 ? 12 : TT % 8 == 0 ? 8 : 4; // taps whose LDS reads are in flight together
#endif
    const int64_t nK0 = floor_div(a.k_lo * a.Ms, a.Ls);
    const uint64_t phK0 = (uint64_t)((double)(a.k_lo * a.Ms - nK0 * a.Ls) * a.fx_per_rem);
    auto position = [&](int64_t k, uint64_t &ph) -> int64_t { // (as in k_poly)
        const uint64_t dk = (uint64_t)(k - a.k_lo), lo = dk * a.step_fx;
        ph = phK0 + lo;
        return nK0 + (int64_t)dk * a.Mq + (int64_t)__umul64hi(dk, a.step_fx) + (ph < lo ? 1 : 0);
    };
    const int slot = (tid * a.lane_mul) & 255;
#ifdef POLY_TRACE
    unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tq = __builtin_amdgcn_s_memtime();
    const unsigned long long tq0 = tq;
#endif
    constexpr int NPF = 12; // frames per thread of the NEXT tile's span held in registers (launch_poly keeps the span within 12 x 256)
    v2f pf[NPF];
    const v2f zero = {0.f, 0.f};
    // one frame of the pair at a source position: 8 bytes (IL) or a sample of each member, each with its own range
    auto load2 = [&](const float *q, bool in1, bool in2) -> v2f {
        if constexpr (IL) return in1 ? *reinterpret_cast<const v2f *>(q) : zero;
        else { v2f r; r.x = in1 ? q[0] : 0.f; r.y = in2 ? q[a.m2_src] : 0.f; return r; }
    };
    auto fetch = [&](int64_t tile, int c) {
        const float *src = src0 + c * c_src;
        const int64_t kA = a.k_lo + tile * per_tile, kEnd = a.k_lo + a.n_out, kB = (kA + per_tile < kEnd ? kA + per_tile : kEnd) - 1;
        uint64_t ph;
        const int64_t nA = position(kA, ph) - (H - 1), nB = position(kB, ph) + H;
        // frames [lo, hi) of the span exist (uniform, 32-bit from here on: one unsigned compare and one pointer step per load)
        const int64_t lo64 = a.n_lo - nA, hi64 = (nB + 1 < a.n_src ? nB + 1 : a.n_src) - nA;
        const int lo = (int)(lo64 < 0 ? 0 : lo64 > NPF * 256 ? NPF * 256 : lo64), hi = (int)(hi64 < lo ? lo : hi64 > NPF * 256 ? NPF * 256 : hi64);
        // (member 2 of a split column: its sample n is the column's n + m2_shift)
        const int64_t lo64b = lo64 - a.m2_shift, hi64b = (nB + 1 < a.n_src - a.m2_shift ? nB + 1 : a.n_src - a.m2_shift) - nA;
        const int lo2 = IL ? 0 : (int)(lo64b < 0 ? 0 : lo64b > NPF * 256 ? NPF * 256 : lo64b), hi2 = IL ? 0 : (int)(hi64b < lo2 ? lo2 : hi64b > NPF * 256 ? NPF * 256 : hi64b);
        const float *p = src + (nA + tid) * a.sfs;
        const int64_t step = 256 * a.sfs;
#pragma unroll
        for (int q = 0; q < NPF; ++q, p += step)
            pf[q] = load2(p, (unsigned)(q * 256 + tid - lo) < (unsigned)(hi - lo), (unsigned)(q * 256 + tid - lo2) < (unsigned)(hi2 - lo2));
    };
    if ((int64_t)blockIdx.x < n_tiles) fetch(blockIdx.x, 0);
    { // (the table behind the first span's loads: one round trip for both)
        const v4f *g = (const v4f *)a.tab;
        for (int i = tid; i < a.P * a.row; i += 256) tab[i] = g[i];
    }
    // (staging the next span BEFORE this tile's stores — gfx9 retires loads and stores through one in-order counter — was
    //  measured: no gain at 16 taps, 50 -> 55 us at 40; profiles/r05_ab_experiments.txt §7)
    for (int64_t item = 0;; ++item) { // work items: (tile, channel pair of the group), pairs innermost
        const int64_t tile = blockIdx.x + (item >> a.lg_cg) * gridDim.x;
        if (tile >= n_tiles) break;
        const int c = (int)(item & (cg - 1));
        const float *src = src0 + c * c_src;
        const int64_t kA = a.k_lo + tile * per_tile;
        const int64_t kEnd = a.k_lo + a.n_out, kB = (kA + per_tile < kEnd ? kA + per_tile : kEnd) - 1;
        uint64_t ph_;
        const int64_t nA = position(kA, ph_) - (H - 1), nB = position(kB, ph_) + H;
        const int span = (int)(nB - nA + 1);
#pragma unroll
        for (int q = 0; q < NPF; ++q)
            if (q * 256 + tid < span) xs[q * 256 + tid] = pf[q];
        for (int i = NPF * 256 + tid; i < span; i += 256) { // (not reached with launch_poly's run lengths)
            const int64_t n = nA + i;
            xs[i] = load2(src + n * a.sfs, n >= a.n_lo && n < a.n_src, n >= a.n_lo - a.m2_shift && n < a.n_src - a.m2_shift);
        }
        POLY_STAMP(0);
        __syncthreads();
        POLY_STAMP(1);
        {
            const int64_t tn = blockIdx.x + ((item + 1) >> a.lg_cg) * gridDim.x;
            if (tn < n_tiles) fetch(tn, (int)((item + 1) & (cg - 1)));
        }
        const int64_t k1 = kA + (int64_t)slot * a.R;
        if (k1 <= kB) {
            uint64_t phase;
            const int64_t n0 = position(k1, phase);
            const v2f *w = xs + (n0 - nA - (H - 1));
            v2f u[TT];
#pragma unroll
            for (int j = 0; j < TT; ++j) u[j] = w[j];
            const v2f *wtop = w + (TT - 2);
            v2f *yo = ys + slot;
            const int lg = a.lgP;
            int left = (int)(kB - k1 + 1 < (int64_t)a.R ? kB - k1 + 1 : (int64_t)a.R);
            for (; left > 0; --left) {
                const uint32_t hi = (uint32_t)(phase >> 32);
                const uint32_t i = hi >> (32 - lg);
                const float x = (float)(uint32_t)(hi << lg) * 0x1p-32f;
                const v4f *row = tab + i * (uint32_t)a.row;
                // the cubic per tap ONCE for both channels (three scalar FMAs), then one packed FMA over the channel pair
                // with the coefficient broadcast: 3 + 1 instructions per tap against k_poly's 2 x 2 for two channels
                v2f y0 = zero, y1 = zero;
#pragma unroll
                for (int j0 = 0; j0 < TT; j0 += CH) {
                    v4f cf[CH];
#pragma unroll
                    for (int j = 0; j < CH; ++j) cf[j] = row[j0 + j];
#pragma unroll
                    for (int j = 0; j < CH; j += 2) { // (the value lands in the record's first register: its pair is the operand, low half broadcast)
                        cf[j].x = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(cf[j].w, x, cf[j].z), x, cf[j].y), x, cf[j].x);
                        cf[j + 1].x = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(cf[j + 1].w, x, cf[j + 1].z), x, cf[j + 1].y), x, cf[j + 1].x);
                        const v2f c0 = __builtin_shufflevector(cf[j], cf[j], 0, 1), c1 = __builtin_shufflevector(cf[j + 1], cf[j + 1], 0, 1);
                        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(y0) : "v"(c0), "v"(u[j0 + j]), "v"(y0));
                        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(y1) : "v"(c1), "v"(u[j0 + j + 1]), "v"(y1));
                    }
                }
                *yo = y0 + y1;
                yo += 257;
                const uint64_t next = phase + a.step_fx;
                const bool adv = next < phase; // the fraction wrapped: one more frame
                phase = next;
                wtop += MQ + (adv ? 1 : 0);
                const v2f e0 = wtop[0], e1 = wtop[1]; // (the span has 4 spare frames behind the last window)
                // (explicit selects: left to the compiler, a select between two elements of the window becomes a dynamically
                //  indexed extract — a compare and a select per ELEMENT of the array.  v_cndmask with its mask in a scalar pair
                //  issues at half rate on gfx950, and so does v_bfi_b32 on a mask in a vector register — 4.4 and 4.7 cycles
                //  against 2.7 for v_fma_f32, tools/ubench/valu_ops.hip: the shift is a quarter of the loop's issue time)
                const uint64_t advm = __builtin_amdgcn_ballot_w64(adv);
#pragma unroll
                for (int j = 0; j < TT - 2; ++j) {
                    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(u[j].x) : "v"(u[j + MQ].x), "v"(u[j + MQ + 1].x), "s"(advm));
                    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(u[j].y) : "v"(u[j + MQ].y), "v"(u[j + MQ + 1].y), "s"(advm));
                }
                u[TT - 2] = e0; u[TT - 1] = e1;
            }
        }
        POLY_STAMP(2);
        __syncthreads(); // the tile's outputs are staged and its source span is free
        POLY_STAMP(3);
        {
            float *yo = dst0 + c * c_dst + (kA - a.k_lo + tid) * a.dfs;
            const int64_t step = 256 * a.dfs;
            const int cnt = (int)(kB - kA + 1);
            const int64_t left2 = a.m2_n_out - (kA - a.k_lo); // (member 2 of a split column may end inside the tile)
            const int cnt2 = (int)(left2 < 0 ? 0 : left2 > cnt ? cnt : left2);
            const float invR = 1.f / (float)a.R;
            for (int i = tid; i < cnt; i += 256, yo += step) {
                const int sl = (int)(((float)i + .5f) * invR), r = i - sl * a.R; // frame i of the tile: run index r of slot sl
                const v2f y = ys[r * 257 + sl];
                if constexpr (IL) *reinterpret_cast<v2f *>(yo) = y;
                else {
                    yo[0] = y.x;
                    if (i < cnt2) yo[a.m2_dst] = y.y;
                }
            }
        }
        POLY_STAMP(4);
    }
#ifdef POLY_TRACE
    if (a.trace && (tid & 63) == 0) {
        unsigned long long *o = a.trace + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + tid / 64) * 8;
        for (int i = 0; i < 5; ++i) o[i] = tr[i];
        o[5] = tq0; o[6] = tq; o[7] = 0;
    }
#endif
}

template <typename Real>
static const char *launch_poly(const TwoStage &ts, const void *src, void *dst, int64_t n_lo, int64_t n_src, int64_t k_lo, int64_t n_out, uint32_t n_clips, uint32_t n_channels,
                               const int64_t sstr[3], const int64_t dstr[3], hipStream_t st)
{
    if (n_out <= 0) return nullptr;
    PolyArgs a;
    a.src = src; a.dst = dst; a.tab = sizeof(Real) == 4 ? ts.tab_f : ts.tab_d;
    const int P = sizeof(Real) == 4 ? ts.P2f : ts.P2;
    a.T = ts.T2; a.P = P; a.row = ts.row;
    a.Ls = ts.Ls; a.Ms = ts.Ms; a.Mq = ts.Ms / ts.Ls; a.Mr = ts.Ms % ts.Ls;
    a.n_lo = n_lo; a.n_src = n_src; a.k_lo = k_lo; a.n_out = n_out;
    a.lgP = 0;
    while ((1 << a.lgP) < P) ++a.lgP;
    a.step_fx = (uint64_t)((((unsigned __int128)(uint64_t)a.Mr) << 64) / (unsigned __int128)(uint64_t)a.Ls);
    a.fx_per_rem = 18446744073709551616. / (double)a.Ls;
    a.scs = sstr[0]; a.sfs = sstr[1]; a.schs = sstr[2]; a.dcs = dstr[0]; a.dfs = dstr[1]; a.dchs = dstr[2];
    a.n_channels = n_channels;
    // channels per workgroup: neighbouring channels of interleaved data (source or destination) share their lines
    a.lg_cg = (sstr[2] == 1 || dstr[2] == 1) ? (n_channels % 4 == 0 ? 2 : n_channels % 2 == 0 ? 1 : 0) : 0;
    // float32, both ends channel-interleaved with an even channel count and 8-byte aligned frames, windows that move by at
    // most two frames: channel PAIRS on k_poly2 (one pass over the data, table reads shared by the two channels)
    const bool pair = sizeof(Real) == 4 && (a.Mq == 0 || a.Mq == 1) && n_channels % 2 == 0 && sstr[2] == 1 && dstr[2] == 1 &&
                      ((sstr[0] | sstr[1] | dstr[0] | dstr[1]) & 1) == 0 && (((uintptr_t)src | (uintptr_t)dst) & 7) == 0 && ts.T2 <= 40 && !switches().poly_no_pair;
    if (pair) a.lg_cg = n_channels % 8 == 0 ? 2 : n_channels % 4 == 0 ? 1 : 0;
    // Any other float32 column of more than ~1.7 phase periods (Ls outputs: integer rate pairs have at most 2 f_out of
    // them per period): split h periods in — output k + h Ls has output k's fraction — and the two segments run as the pair
    a.m2_src = a.m2_dst = a.m2_shift = 0; a.m2_n_out = n_out;
    bool split = false;
    if (!pair && sizeof(Real) == 4 && (a.Mq == 0 || a.Mq == 1) && ts.T2 <= 40 && !switches().poly_no_pair) {
        const int64_t h = (n_out + 2 * ts.Ls - 1) / (2 * ts.Ls), n1 = h * ts.Ls, n2 = n_out - n1; // (member 1 is the longer one)
        if (h >= 1 && 10 * n2 >= 7 * n1 && h * ts.Ms < (1LL << 40)) {
            split = true;
            a.n_out = n1; a.m2_n_out = n2; a.m2_shift = h * ts.Ms; a.m2_src = a.m2_shift * sstr[1]; a.m2_dst = n1 * dstr[1];
        }
    }
    const bool two = pair || split; // k_poly2
    n_out = a.n_out;                // (tiles are counted over member 1 of a split column)
    const int cg = 1 << a.lg_cg;
    const size_t unit = two ? 2 * sizeof(Real) : sizeof(Real); // bytes per staged source / output element
    // outputs per thread: as many as keep the tile's source span within the LDS left beside the table (<= 8)
    const size_t tab_bytes = (size_t)P * ts.row * 4 * sizeof(Real);
    const size_t lds_cap = (sizeof(Real) == 4 ? (tab_bytes > 40 * 1024 || (two && ts.T2 >= 32) ? 78 : 52) : 96) * 1024; // three or two (float) / one (double) workgroups per CU
    const double ratio = (double)ts.Ms / (double)ts.Ls;
    int Rmax = 12;
    while (Rmax > 1 && (tab_bytes + (size_t)(256. * Rmax * ratio + 257. * Rmax + ts.T2 + 4) * unit > lds_cap || // (source span + staged outputs)
                        (two && 256. * Rmax * ratio + ts.T2 + 4 > 12. * 256.)))                                      // (k_poly2 holds a whole span in registers)
        --Rmax;
    // Thread t owns R consecutive outputs starting ((t * lane_mul) mod 256) * R into the tile (lane_mul odd: a bijection).
    // Lanes l, l + 1 of a wave are then lane_mul * R outputs apart and their table rows form the arithmetic progression
    // floor(c + l s), s = frac(lane_mul R Ms / Ls) P.  A 16-byte LDS read serves a lane group in one cycle when its 16 lanes
    // fall on 16 different bank quads — (row + tap) mod 16 with the table's odd row stride — and takes one more cycle per
    // extra distinct record on a quad.  Random rows cost 2.5-3 cycles; s within ~0.02 of an odd integer costs 1.
    // (R, lane_mul) is picked by simulating the four lane groups of the four waves over 16 starting phases.
    // R: as long as the tiles fill the workgroups the chip holds, the longest run that fits (a tile's fixed cost is about two
    // outputs' time: job time = a + b / R with b / a = 2.06 — and a partly filled last round costs its share, not a round:
    // 60 / 90 / 120 s take 56 / 72 / 88 us); below that, runs short enough to give every resident workgroup a tile
    // (5 / 10 s stereo 24.3 / 25.0 -> 20.0 / 22.2 us, 10 s mono 26.0 -> 20.2; profiles/r05_ab_experiments.txt §7).
    if (switches().dbg_poly_r > 0) Rmax = std::min(Rmax, switches().dbg_poly_r);
    const uint64_t cols = (uint64_t)n_clips * n_channels / (uint64_t)(pair ? 2 * cg : cg); // channel groups
    if (cols > 65535) return "two-stage: too many columns";
    int dev = 0, n_cu = 256;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    const int occ_limit = two ? (ts.T2 >= 32 ? 2 : 3) : sizeof(Real) == 4 ? 4 : 2; // (registers: k_poly2's launch bounds, k_poly's ~108 / float64's ~200)
    auto lds_of = [&](int r) { return tab_bytes + ((size_t)(256. * r * ratio + ts.T2 + 4) + 257u * (size_t)r) * unit; };
    auto slots_of = [&](int r) { // workgroups per column the chip holds at once (launch geometry below)
        const int per_cu = std::max(1, std::min(occ_limit, (int)((160u * 1024u) / lds_of(r))));
        return std::max<int64_t>(1, (int64_t)per_cu * n_cu / (int64_t)cols);
    };
    int R = Rmax, lane_mul = 1;
    {
        static std::mutex mu;
        std::lock_guard<std::mutex> lk(mu);
        const int which = two ? 2 : sizeof(Real) == 8;
        auto lanes_for = [&](int r) -> double { // best lane multiplier for runs of r (cached per plan): its conflict cost, 1 = none
            if (!ts.lane_for[which][r]) {
                static const int group[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27}, {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                                 {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59}, {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
                const int halves = sizeof(Real) == 8 ? 2 : 1; // a double4 record is two 16-byte reads
                double best = 1e30;
                int best_am = 1;
                for (int am = 1; am < 256 && best > 16. * 16. * halves; am += 2) { // (every group in one cycle: nothing better to look for)
                    double cost = 0.;
                    for (int ph = 0; ph < 16 && cost < best; ++ph)
                        for (int wave = 0; wave < 4; ++wave)
                            for (int g = 0; g < 4; ++g) {
                                int rows[16][16], cnt[16] = {0}, mx = 1; // distinct rows seen per quad
                                for (int q = 0; q < 16; ++q) {
                                    const int tid = 64 * wave + group[g][q];
                                    const double f = ph / 16. + .37 + (double)((tid * am) & 255) * r * ratio;
                                    const int i = (int)((f - std::floor(f)) * P) % P;
                                    const int quad = (halves * ts.row * i) & 15;
                                    bool seen = false;
                                    for (int z = 0; z < cnt[quad]; ++z) seen |= rows[quad][z] == i;
                                    if (!seen) { rows[quad][cnt[quad]++] = i; mx = std::max(mx, cnt[quad]); }
                                }
                                cost += mx;
                            }
                    if (cost < best) { best = cost; best_am = am; }
                }
                ts.lane_for[which][r] = best_am; ts.conf_for[which][r] = (float)(best / (16. * 16. * halves));
            }
            return ts.conf_for[which][r];
        };
        const int64_t slots = slots_of(Rmax);
        if ((n_out + 256LL * Rmax - 1) / (256LL * Rmax) >= slots) { // a round or more: long runs, among them the one whose table reads conflict least
            double best = 1e30;
            for (int r = Rmax; r >= std::min(Rmax, std::max(2, Rmax / 3)) && best > 1.; --r) {
                const double cost = lanes_for(r) * (1. + .02 * (Rmax - r)); // (a shorter run per thread: more tiles per output)
                if (cost < best) { best = cost; R = r; }
            }
        } else // less than one round: shorter runs spread the job over the workgroups the chip holds
            R = (int)std::max<int64_t>(std::min(Rmax, 2), std::min<int64_t>(Rmax, (n_out + 256 * slots - 1) / (256 * slots)));
        lanes_for(R);
        lane_mul = ts.lane_for[which][R];
    }
    a.lane_mul = lane_mul;
    a.R = R;
    a.span_max = (int)(256. * R * ratio + ts.T2 + 4);
    const size_t lds = tab_bytes + ((size_t)a.span_max + 257u * (size_t)R) * unit;
    if (lds > 160 * 1024) return "two-stage: polyphase tile does not fit LDS";
    const int64_t n_tiles = (n_out + 256LL * R - 1) / (256LL * R);
    void (*kern)(PolyArgs) = nullptr;
    switch (ts.T2) {
#define HIPSOXR_POLY_T(t) case t: kern = a.Mq == 0 ? k_poly<Real, t, 0> : a.Mq == 1 ? k_poly<Real, t, 1> : k_poly<Real, t, -1>; break;
        HIPSOXR_POLY_TAPS(HIPSOXR_POLY_T)
#undef HIPSOXR_POLY_T
    }
    if constexpr (sizeof(Real) == 4)
        if (two) switch (ts.T2) {
#define HIPSOXR_POLY_T(t) case t: kern = pair ? (a.Mq == 0 ? k_poly2<t, 0, true> : k_poly2<t, 1, true>) : (a.Mq == 0 ? k_poly2<t, 0, false> : k_poly2<t, 1, false>); break;
            HIPSOXR_POLY2_TAPS(HIPSOXR_POLY_T)
#undef HIPSOXR_POLY_T
        }
    if (!kern) return "two-stage: no polyphase instance for this tap count";
    if (const char *e = ensure_dyn_lds((const void *)kern, lds)) return e;
    // workgroups walk tiles (the table is loaded once per workgroup): exactly as many as the chip holds at once — a
    // partly filled second round of workgroups would double the launch
    int per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)kern, 256, lds));
    const int64_t want = std::max<int64_t>(1, (int64_t)std::max(per_cu, 1) * n_cu / (int64_t)cols);
    unsigned gx = (unsigned)std::min<int64_t>(n_tiles, want);
    if (gx > 8) gx &= ~7u; // columns' workgroups of one tile index on ONE XCD (workgroup b -> XCD b mod 8): interleaved channels share their lines in its L2
    a.trace = nullptr;
#ifdef POLY_TRACE
    const size_t trace_n = (size_t)gx * cols * 4 * 8;
    if (switches().dbg_trace) {
        HIP_TRY(hipMalloc((void **)&a.trace, trace_n * 8));
        HIP_TRY(hipMemset(a.trace, 0, trace_n * 8));
    }
#endif
    hipLaunchKernelGGL(kern, dim3(gx, (unsigned)cols, 1), dim3(256), lds, st, a);
    HIP_TRY(hipGetLastError());
#ifdef POLY_TRACE
    if (a.trace) { // debugging aid only: synchronous dump of the per-wave cycle sums
        std::vector<unsigned long long> h(trace_n);
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipMemcpy(h.data(), a.trace, trace_n * 8, hipMemcpyDeviceToHost));
        if (FILE *f = fopen(switches().dbg_trace, "wb")) { fwrite(h.data(), 8, trace_n, f); fclose(f); }
        (void)hipFree(a.trace);
    }
#endif
    return nullptr;
}

// Whole-signal float job on an interpolated-phase plan: FFT stage + polyphase stage.  *handled = false
// (nothing launched) when the plan or the job is not one the two-stage form serves: the caller's ordinary path takes it.
const char *launch_two_stage(Plan *p, const hipsoxr_job_t &j, void *stream, bool *handled)
{
    *handled = false;
    if (!p->phases || (j.elem != HIPSOXR_F32 && j.elem != HIPSOXR_F64) || j.in_abs0 != 0 || j.out_k0 != 0 || j.clip_table) return nullptr;
    if ((uint64_t)j.out_frames > plan_out_len(*p, (uint64_t)j.in_frames)) return nullptr;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        if (!p->two)
            if (const char *e = twostage_build(p)) return e;
    }
    const TwoStage &ts = *p->two;
    if (!ts.ok) return nullptr;
    const int64_t n = j.in_frames, n_out = j.out_frames;
    if (n_out < 8192 || n < 8192 || n >= (1LL << 30) || n_out >= (1LL << 30)) return nullptr;
    const uint64_t cols = (uint64_t)j.n_clips * j.n_channels;
    if (cols > 65535) return nullptr;
    const size_t es = j.elem == HIPSOXR_F32 ? 4 : 8;
    // the polyphase table and one tile's source span must fit LDS in the job's precision (long stages in float64 do not:
    // the exact engine keeps those)
    if ((size_t)(es == 4 ? ts.P2f : ts.P2) * ts.row * 4 * es + (size_t)(512. * ((double)ts.Ms / (double)ts.Ls + 1.01) + ts.T2 + 4) * es > 150u * 1024u) return nullptr;
    hipStream_t st = (hipStream_t)stream;
    // The intermediate signal runs PAST both ends of the job, as far as the second stage reads it: `pad` samples of it
    // before sample 0 and after the last one (a multiple of 8: 16-byte phases of the columns are kept).  The stage that
    // writes it produces those samples like any others (its own input zero-extended), the stage that reads it sees zeros
    // beyond them — so the composite is the prototype's response to the zero-extended signal at EVERY output, ends
    // included, and no third engine patches the ends.
    //   down: v[m], m in [-pad, 2 n_out + pad): pad >= polyphase half-width in v samples
    //   up:   u[m], m in [-pad, 2 n + pad):     pad >= polyphase half-width (T2 / 2 u samples)
    const double half_mid = ts.up ? .5 * ts.T2 : .5 * ts.T2 * (double)ts.Ls / (double)ts.Ms;
    const int64_t pad = ((int64_t)std::ceil(half_mid) + 4 + 7) / 8 * 8;
    const int64_t n_core = ts.up ? 2 * n : 2 * n_out, n_mid = n_core + 2 * pad;
    void *mid = nullptr;
    HIP_TRY(hipMallocAsync(&mid, (size_t)cols * (size_t)n_mid * es, st));
    // intermediate layout [clip][channel][frames] — or [clip][frames][channel] when the job's own data is interleaved with an
    // even channel count: the FFT stage then takes its channel-pair form (one complex word per frame and pair, contiguous
    // for stereo) instead of pairing blocks over strided columns
    const bool inter = j.n_channels % 2 == 0 && j.in_chan_stride == 1 && j.out_chan_stride == 1;
    const int64_t mstr[3] = {n_mid * (int64_t)j.n_channels, inter ? (int64_t)j.n_channels : 1, inter ? 1 : n_mid};
    const int64_t istr[3] = {j.in_clip_stride, j.in_frame_stride, j.in_chan_stride};
    const int64_t ostr[3] = {j.out_clip_stride, j.out_frame_stride, j.out_chan_stride};
    auto fail = [&](const char *e) { (void)hipFreeAsync(mid, st); return e; };
    hipsoxr_job_t fj = j;
    fj.kernel = j.kernel == HIPSOXR_KERNEL_FFT_F64 ? HIPSOXR_KERNEL_FFT_F64 : HIPSOXR_KERNEL_FFT;
    fj.clip_counter = nullptr; fj.dither = 0;
    bool fft_done = false;
    const char *err = nullptr;
    void *mid0 = (char *)mid + (size_t)(pad * mstr[1]) * es; // sample 0 of the first column
    if ((err = device_bank_ensure(&p->two->fft, engine_prec(j.elem)))) return fail(err);
    if (ts.up) {
        // 1:2 over the input delayed by pad / 2 samples: its output m' is u[m' - pad]
        fj.in_abs0 = pad / 2;
        fj.out = mid; fj.out_clip_stride = mstr[0]; fj.out_frame_stride = mstr[1]; fj.out_chan_stride = mstr[2];
        fj.in_frames = n; fj.out_frames = n_mid;
        if ((err = launch_fft(&p->two->fft, fj, stream, &fft_done))) return fail(err);
        if (!fft_done) { (void)hipFreeAsync(mid, st); return nullptr; }
        err = j.elem == HIPSOXR_F32 ? launch_poly<float>(ts, mid0, j.out, -pad, n_core + pad, 0, n_out, j.n_clips, j.n_channels, mstr, ostr, st)
                                    : launch_poly<double>(ts, mid0, j.out, -pad, n_core + pad, 0, n_out, j.n_clips, j.n_channels, mstr, ostr, st);
        if (err) return fail(err);
    } else {
        err = j.elem == HIPSOXR_F32 ? launch_poly<float>(ts, j.in, mid, 0, n, -pad, n_mid, j.n_clips, j.n_channels, istr, mstr, st)
                                    : launch_poly<double>(ts, j.in, mid, 0, n, -pad, n_mid, j.n_clips, j.n_channels, istr, mstr, st);
        if (err) return fail(err);
        fj.in = mid; fj.in_abs0 = -pad; fj.in_clip_stride = mstr[0]; fj.in_frame_stride = mstr[1]; fj.in_chan_stride = mstr[2];
        fj.in_frames = n_mid; fj.out_frames = n_out;
        if ((err = launch_fft(&p->two->fft, fj, stream, &fft_done))) return fail(err);
        if (!fft_done) { (void)hipFreeAsync(mid, st); return nullptr; } // (declined: the queued first stage wrote the intermediate only; the ordinary path computes the job)
    }
    (void)hipFreeAsync(mid, st);
    (void)mid0;
    *handled = true;
    return nullptr;
}

} // namespace hipsoxr
