// kernels_chain.h — exact engine, small launches and streams: k_chain, k_chain_multi, k_chain_resident.
// Part of the ONE translation unit kernels.hip (included there, inside namespace hipsoxr, behind the conversions and the
// output helpers): a cut by kernel family, not a separate compilation.

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

