// kernels_interp.h — exact engine, interpolated-phase plans and the quad (wave) forms: k_interp, k_gather_wave, k_interp_tile, k_interp_wave.
// Part of the ONE translation unit kernels.hip (included there, inside namespace hipsoxr, behind the conversions and the
// output helpers): a cut by kernel family, not a separate compilation.

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

