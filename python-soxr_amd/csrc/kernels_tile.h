// kernels_tile.h — exact engine, period tiles: k_tile, k_tile_mfma, k_tile_mfma_p, k_tile_mfma64_p.
// Part of the ONE translation unit kernels.hip (included there, inside namespace hipsoxr, behind the conversions and the
// output helpers): a cut by kernel family, not a separate compilation.

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

