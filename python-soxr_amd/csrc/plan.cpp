// plan.cpp — host-side filter design for the MI355X resampler (float64 throughout).
//
// Written from the specification, not from libsoxr's sources (absent from the reference
// checkout): recipe -> (bits, pass-band end, stop-band begin) as libsoxr documents them
// (SURVEY.md §A.1/§A.2), a Kaiser-windowed-sinc prototype by the textbook design rule, and the
// zero-latency polyphase decomposition the kernels consume.  oracle/soxr_oracle.c restates the
// same specification independently; tests require the two banks to agree.
//
// Compile with -ffp-contract=off so that host arithmetic is plain IEEE (no implicit fusing).
#include "plan.h"

#include <cmath>
#include <cstdlib>

namespace hipsoxr {

// Kaiser's empirical estimate undershoots the requested attenuation by 1-3 dB at the stop-band edge;
// design for a little more than the recipe asks (verified in tests/test_plan.py).
static const double kAttMarginDb = 2.0;

const char *quality_spec(unsigned long recipe, QualitySpec *q)
{
    unsigned long r = recipe & 0xf;
    if (r > 7) return "invalid quality recipe";
    q->bits = r == 0 ? 0. : r < 4 ? 16. : 4. + 4. * (double)r;
    q->stopband_begin = 1.;
    double rej = q->bits * 20. * std::log10(2.);
    if (r == 0)
        q->passband_end = 0.;
    else if (r == 1)
        q->passband_end = 1385. / 2048.;
    else
        q->passband_end = 1. - .05 / ((1.6e-6 * rej - 7.5e-4) * rej + .646);
    return nullptr;
}

static int64_t gcd64(int64_t a, int64_t b)
{
    while (b) { int64_t t = a % b; a = b; b = t; }
    return a;
}

const char *reduce_ratio(double in_rate, double out_rate, int64_t *L, int64_t *M)
{
    if (!(in_rate > 0) || !(out_rate > 0)) return "sample rate must be > 0";
    if (in_rate == std::floor(in_rate) && out_rate == std::floor(out_rate) && in_rate < 9e15 &&
        out_rate < 9e15) {
        int64_t a = (int64_t)out_rate, b = (int64_t)in_rate, g = gcd64(a, b);
        *L = a / g; *M = b / g;
        return nullptr;
    }
    // Continued fraction of the double quotient r = out/in, expanded EXACTLY: r = mant * 2^ex is a
    // rational with a power-of-two denominator, so Euclid's algorithm on (numerator, denominator) in
    // 128-bit integers yields its true partial quotients (iterating x -> 1/(x - a) in doubles drifts
    // off them after ~10 levels).  A convergent that reproduces r to 1e-15 (tested in doubles, the
    // way the result will be used) is taken at once; when the next convergent would leave the 31-bit
    // range first, the best semiconvergent inside the range stands in (error < 1/(k k_prev): a drift
    // below one sample in 2^31).
    const int64_t LIM = 2147483647LL;
    const double r = out_rate / in_rate;
    if (!(r > 0) || !std::isfinite(r)) return "rate ratio is out of range";
    int ex = 0;
    const double fr = std::frexp(r, &ex); // r = fr * 2^ex, fr in [0.5, 1)
    unsigned __int128 num = (unsigned __int128)(uint64_t)std::ldexp(fr, 53), den = 1;
    ex -= 53;                              // r = num * 2^ex
    if (ex > 40 || ex < -110) return "rate ratio is out of range";
    if (ex >= 0) num <<= ex; else den <<= -ex;
    int64_t h0 = 0, h1 = 1, k0 = 1, k1 = 0;
    for (int it = 0; it < 128 && den != 0; ++it) {
        const unsigned __int128 a128 = num / den, rem = num % den;
        bool over = a128 > (unsigned __int128)LIM;
        const int64_t ai = over ? LIM : (int64_t)a128;
        if (!over) over = (h1 && ai > (LIM - h0) / h1) || (k1 && ai > (LIM - k0) / k1);
        if (over) {
            if (k1 == 0) return "rate ratio is out of range";
            int64_t amax = LIM;
            if (h1 && (LIM - h0) / h1 < amax) amax = (LIM - h0) / h1;
            if ((LIM - k0) / k1 < amax) amax = (LIM - k0) / k1;
            if (amax >= 1) {
                const int64_t hs = amax * h1 + h0, ks = amax * k1 + k0;
                if (std::fabs((double)hs / (double)ks - r) < std::fabs((double)h1 / (double)k1 - r)) { h1 = hs; k1 = ks; }
            }
            break;
        }
        const int64_t h2 = ai * h1 + h0, k2 = ai * k1 + k0;
        h0 = h1; h1 = h2; k0 = k1; k1 = k2;
        if (std::fabs((double)h1 / (double)k1 - r) <= 1e-15 * r) break;
        num = den; den = rem;
    }
    if (k1 <= 0 || h1 <= 0) return "rate ratio is out of range";
    *L = h1; *M = k1;
    return nullptr;
}

double bessel_i0(double x)
{
    double sum = 1., term = 1., q = x * x * .25;
    for (int k = 1; k < 500; ++k) {
        term *= q / ((double)k * (double)k);
        sum += term;
        if (term < sum * 1e-17) break;
    }
    return sum;
}

// Largest exact-rational bank (entries).  Ratios that would need more phases x taps than this —
// random integer or float rates, reference tests/test_random.py:21-26 — get an interpolated-phase
// plan instead.
static const int64_t kMaxBankElems = (int64_t)1 << 22;

// ---- interpolated-phase plans -----------------------------------------------------------------
// The same prototype as a function of continuous time tau (in input samples),
//     h(tau) = 2 fc sinc(2 fc tau) . I0(beta sqrt(1 - (tau/W)^2)) / I0(beta),   W = T/2,
// normalised to unit DC gain on a 64x over-sampled grid.  An output at fractional input position
// f = ((k M) mod L) / L weights tap j with h(f + T/2-1-j).  The f axis is divided into P intervals;
// on each, every tap is the cubic through h at the interval's four Chebyshev nodes, kept in
// monomial form [P][T][4] so that a kernel evaluates it with three FMAs (Horner) per tap.
namespace {
struct Proto {
    bool cubic = false; // QQ: Lagrange 4-point kernel instead of the windowed sinc
    double fc = 0, W = 0, beta = 0, inv_i0 = 0, stretch = 1;
    double operator()(double tau) const
    {
        if (cubic) {
            double t = std::fabs(tau) / stretch;
            if (t < 1.) return (1. - t * t) * (2. - t) * .5;
            if (t < 2.) return (1. - t) * (2. - t) * (3. - t) / 6.;
            return 0.;
        }
        double u = tau / W, w = 1. - u * u;
        if (w < 0.) w = 0.;
        double s = tau == 0. ? 2. * fc : std::sin(2. * M_PI * fc * tau) / (M_PI * tau);
        return s * bessel_i0(beta * std::sqrt(w)) * inv_i0;
    }
};
const int kGrid = 64;
} // namespace

static void design_interp(Plan *p)
{
    const int32_t T = p->T, P = p->phases;
    Proto h;
    h.cubic = p->q.bits == 0.;
    h.W = .5 * (double)T;
    double scale = 1.;
    if (h.cubic) {
        h.stretch = p->M > p->L ? (double)p->M / (double)p->L : 1.;
    } else {
        const double fn = .5 * (p->in_rate < p->out_rate ? p->in_rate : p->out_rate);
        h.fc = .5 * (p->q.passband_end + p->q.stopband_begin) * fn / p->in_rate;
        h.beta = p->beta;
        h.inv_i0 = 1. / bessel_i0(p->beta);
        const int64_t half = (int64_t)kGrid * T / 2;
        double sum = 0.;
        for (int64_t m = -half; m < half; ++m) sum += h((double)m / (double)kGrid);
        scale = (double)kGrid / sum;
    }
    double node[4];
    for (int c = 0; c < 4; ++c) node[c] = .5 - .5 * std::cos((double)(2 * c + 1) * M_PI / 8.);
    const double s01 = node[0] + node[1], s012 = node[0] + node[1] + node[2], p01 = node[0] * node[1],
                 e2 = node[0] * node[1] + node[0] * node[2] + node[1] * node[2],
                 p012 = node[0] * node[1] * node[2];
    p->proto_scale = scale;
    p->bank.assign((size_t)P * T * 4, 0.);
    std::vector<double> v((size_t)4 * T);
    for (int32_t i = 0; i < P; ++i) {
        for (int c = 0; c < 4; ++c) {
            const double f = ((double)i + node[c]) / (double)P;
            double sum = 0.;
            for (int32_t j = 0; j < T; ++j) {
                v[(size_t)c * T + j] = h(f + (double)(T / 2 - 1 - j)) * scale;
                sum += v[(size_t)c * T + j];
            }
            if (h.cubic)
                for (int32_t j = 0; j < T; ++j) v[(size_t)c * T + j] /= sum;
        }
        for (int32_t j = 0; j < T; ++j) {
            const double v0 = v[j], v1 = v[(size_t)T + j], v2 = v[(size_t)2 * T + j], v3 = v[(size_t)3 * T + j];
            const double d01 = (v1 - v0) / (node[1] - node[0]), d12 = (v2 - v1) / (node[2] - node[1]),
                         d23 = (v3 - v2) / (node[3] - node[2]);
            const double d012 = (d12 - d01) / (node[2] - node[0]), d123 = (d23 - d12) / (node[3] - node[1]);
            const double d3 = (d123 - d012) / (node[3] - node[0]);
            double *a = &p->bank[((size_t)i * T + j) * 4];
            a[3] = d3;
            a[2] = d012 - d3 * s012;
            a[1] = d01 - d012 * s01 + d3 * e2;
            a[0] = v0 - d01 * node[0] + d012 * p01 - d3 * p012;
        }
    }
}

double plan_proto(const Plan &p, double tau)
{
    Proto h;
    h.W = .5 * (double)p.T;
    const double fn = .5 * (p.in_rate < p.out_rate ? p.in_rate : p.out_rate);
    h.fc = .5 * (p.q.passband_end + p.q.stopband_begin) * fn / p.in_rate;
    h.beta = p.beta;
    h.inv_i0 = 1. / bessel_i0(p.beta);
    return h(tau) * p.proto_scale;
}

const char *plan_design(double in_rate, double out_rate, unsigned long recipe, Plan *p, bool force_interp)
{
    if (const char *e = quality_spec(recipe, &p->q)) return e;
    if (const char *e = reduce_ratio(in_rate, out_rate, &p->L, &p->M)) return e;
    p->in_rate = in_rate; p->out_rate = out_rate; p->recipe = recipe;
    const int64_t L = p->L, M = p->M;
    const double fn = .5 * (in_rate < out_rate ? in_rate : out_rate);
    const double fs_hi = (double)L * in_rate;

    if (p->q.bits == 0.) { // QQ: 4-point cubic Lagrange kernel, stretched when down-sampling
        double s = M > L ? (double)M / (double)L : 1.;
        int t = (int)std::ceil(4. * s);
        p->T = (t + 7) / 8 * 8;
        p->att_db = 0.; p->beta = 0.;
        if (force_interp || L * (int64_t)p->T > kMaxBankElems) { p->phases = 256; design_interp(p); return nullptr; }
        p->bank.assign((size_t)(L * p->T), 0.);
        const int32_t T = p->T;
        for (int64_t ph = 0; ph < L; ++ph) {
            double sum = 0.;
            for (int j = 0; j < T; ++j) {
                double t2 = std::fabs((double)(L * ((int64_t)T / 2 - 1 - j) + ph) / (double)L) / s, v;
                if (t2 < 1.) v = (1. - t2 * t2) * (2. - t2) * .5;
                else if (t2 < 2.) v = (1. - t2) * (2. - t2) * (3. - t2) / 6.;
                else v = 0.;
                p->bank[(size_t)(ph * T + j)] = v;
                sum += v;
            }
            for (int j = 0; j < T; ++j) p->bank[(size_t)(ph * T + j)] /= sum;
        }
        return nullptr;
    }

    const double dw = 2. * M_PI * (p->q.stopband_begin - p->q.passband_end) * fn / fs_hi;
    const double A = (p->q.bits + 1.) * 20. * std::log10(2.) + kAttMarginDb;
    const double n_hi = (A - 7.95) / (2.285 * dw) + 1.;
    int64_t t = (int64_t)std::ceil(n_hi / (double)L);
    if (t < 8) t = 8;
    if (t > (1 << 24)) return "rate ratio needs too many taps";
    p->T = (int32_t)((t + 7) / 8 * 8);
    p->att_db = A;
    p->beta = .1102 * (A - 8.7);
    if (force_interp || L * (int64_t)p->T > kMaxBankElems) {
        p->phases = p->q.bits <= 16. ? 16 : p->q.bits <= 20. ? 32 : 128;
        design_interp(p);
        return nullptr;
    }

    const int32_t T = p->T;
    const int64_t half = L * (int64_t)T / 2;
    const double fc = .5 * (p->q.passband_end + p->q.stopband_begin) * fn / fs_hi;
    const double inv_i0 = 1. / bessel_i0(p->beta), inv_half = 1. / (double)half;
    p->bank.assign((size_t)(L * T), 0.);
    double sum = 0.;
    for (int64_t m = -half; m < half; ++m) {
        double u = (double)m * inv_half, w = 1. - u * u, a = 2. * M_PI * fc * (double)m;
        if (w < 0.) w = 0.;
        double s = m == 0 ? 2. * fc : std::sin(a) / (M_PI * (double)m);
        double v = s * bessel_i0(p->beta * std::sqrt(w)) * inv_i0;
        int64_t qq = m + half; // = L*(T-1-j) + phase
        p->bank[(size_t)((qq % L) * T + (T - 1 - qq / L))] = v;
        sum += v;
    }
    const double scale = (double)L / sum;
    for (size_t i = 0; i < p->bank.size(); ++i) p->bank[i] *= scale;
    return nullptr;
}

uint64_t plan_out_len(const Plan &p, uint64_t n_in)
{
    unsigned __int128 num = (unsigned __int128)n_in * (unsigned __int128)p.L * 2u + (unsigned __int128)p.M;
    return (uint64_t)(num / ((unsigned __int128)p.M * 2u));
}

} // namespace hipsoxr
