/*
 * soxr_oracle.c — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load this library,
 * and only as the checker / the timed CPU baseline — never as part of the shipped path
 * (python-soxr_amd/ never imports, links or executes anything under oracle/).
 *
 * WHAT IT RESTATES.  The arithmetic behind python-soxr's hot path lives in a third-party
 * dependency that is ABSENT from the reference checkout: libsoxr ("dofuuz/soxr", a patched fork
 * of libsoxr 0.1.3), an un-vendored git submodule with no recoverable pin
 * (/root/reference/.gitmodules:1-3, /root/reference/libsoxr is empty).  It can be neither
 * compiled nor imported in this image, so this file restates libsoxr's PUBLISHED behaviour, i.e.
 * what its public header and documentation promise and what the reference's own call sites and
 * tests pin:
 *   - quality recipe -> (precision bits, pass-band end, stop-band begin):      oracle_quality()
 *     (recipe constants exported at /root/reference/src/soxr_ext.cpp:447-451; numbers in
 *      SURVEY.md §A.1/§A.2)
 *   - band-limited interpolation by a Kaiser-windowed-sinc polyphase FIR bank: oracle_design_bank()
 *     (the "polyphase FIR filter bank" BASELINE.json:north_star names; textbook Kaiser design,
 *      Kaiser 1974 / Oppenheim & Schafer §7.5 — NOT libsoxr's private coefficient fits)
 *   - zero-latency alignment and the output-length rule floor(n*out/in + 1/2): oracle_out_len()
 *     (lengths pinned by /root/reference/tests/test_resample.py:142-156)
 *   - the soxr_process inner product  y[k] = sum_j c[phase_k][j] * x[n_k + j]: oracle_resample_*()
 *     (call sites /root/reference/src/soxr_ext.cpp:163-166, :245-248, :328-331)
 *   - integer output: round-half-even, saturate, count clips, TPDF dither on int16:
 *     oracle_quantize_*()  (behaviour pinned by /root/reference/tests/test_resample.py:119-130,
 *     :159-176, :194-211 to +-2 LSB)
 *
 * PARITY STATUS: **parity unpinned** against libsoxr itself (no libsoxr binary, source or golden
 * vector exists in this image or in the reference).  The oracle IS pinned against every
 * known-answer test the reference holds for this path — the analytic tone tests
 * (test_quality_sine, atol 1e-4, all five recipes, exact lengths; test_int_sine, +-2 LSB) — by
 * tests/test_oracle_pinning.py, and against an independent implementation
 * (scipy.signal.upfirdn with the same bank).
 *
 * Two arithmetic modes:
 *   *_ref    float64 accumulation, ascending taps (what a float64 CPU engine would do);
 *   *_port   the CANONICAL ORDER the HIP kernels implement, in the engine precision (f32 or f64):
 *                accL = 0; for j = 0 .. T/2-1      : accL = fma(c[j], x[j], accL)
 *                accR = 0; for j = T-1 .. T/2 (desc): accR = fma(c[j], x[j], accR)
 *                y = accL + accR
 *            (both half-chains run from the small outer taps toward the large centre taps; in
 *            f32 this keeps the error at ~4.6e-8 relative RMS vs ~2.9e-7 for one ascending chain).
 *            GPU results must equal *_port BIT FOR BIT; they must equal *_ref within 1e-6 RMS.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------ */
/* recipe -> spec                                                                              */
/* ------------------------------------------------------------------------------------------ */
/* precision bits: 0 for QQ (cubic), 16 for LQ/MQ, 4+4q for HQ(4)/VHQ(6);
 * pass-band end (fraction of the lower rate's Nyquist): LQ 1385/2048, else 1 - 0.05/TO_3dB(rej),
 * rej = bits*20log10(2), TO_3dB(a) = (1.6e-6 a - 7.5e-4) a + 0.646; stop-band begins at 1.0. */
API int oracle_quality(unsigned long recipe, double *bits, double *passband_end,
                       double *stopband_begin)
{
    unsigned long q = recipe & 0xf;
    double b, rej;
    if (q > 7) return -1;
    b = q == 0 ? 0. : q < 4 ? 16. : 4. + 4. * (double)q;
    rej = b * 20. * log10(2.);
    *bits = b;
    *stopband_begin = 1.;
    if (q == 0)
        *passband_end = 0.;
    else if (q == 1)
        *passband_end = 1385. / 2048.;
    else
        *passband_end = 1. - .05 / ((1.6e-6 * rej - 7.5e-4) * rej + .646);
    return 0;
}

static int64_t gcd64(int64_t a, int64_t b)
{
    while (b) { int64_t t = a % b; a = b; b = t; }
    return a;
}

/* out/in as a reduced fraction L/M.  Integer-valued rates: exact gcd.  Otherwise: continued
 * fraction of out/in, accepted when it reproduces the double ratio to 1e-15 relative.
 * Returns 0 on success, -1 when no such fraction with L, M <= 2^31 exists. */
API int oracle_ratio(double in_rate, double out_rate, int64_t *L, int64_t *M)
{
    if (!(in_rate > 0) || !(out_rate > 0)) return -1;
    if (in_rate == floor(in_rate) && out_rate == floor(out_rate) && in_rate < 9e15 &&
        out_rate < 9e15) {
        int64_t a = (int64_t)out_rate, b = (int64_t)in_rate, g = gcd64(a, b);
        *L = a / g; *M = b / g;
        return 0;
    }
    {
        /* Continued fraction of out/in.  A convergent that reproduces the double ratio to 1e-15 is
         * accepted at once; if the next convergent would leave the 31-bit range first, the best
         * semiconvergent inside the range is taken instead (error below 1/(k*k_prev), i.e. still far
         * below anything audible or measurable: a drift of less than one sample in 2^31). */
        const int64_t LIM = 2147483647LL;
        double r = out_rate / in_rate, x = r;
        int64_t h0 = 0, h1 = 1, k0 = 1, k1 = 0;
        int it;
        for (it = 0; it < 64; ++it) {
            double a = floor(x);
            int64_t ai, h2, k2;
            int over = a > (double)LIM;
            ai = over ? LIM : (int64_t)a;
            if (!over) {
                over = (h1 && ai > (LIM - h0) / h1) || (k1 && ai > (LIM - k0) / k1);
            }
            if (over) {
                int64_t amax = LIM;
                if (k1 == 0) return -1; /* ratio itself beyond 2^31 */
                if (h1 && (LIM - h0) / h1 < amax) amax = (LIM - h0) / h1;
                if ((LIM - k0) / k1 < amax) amax = (LIM - k0) / k1;
                if (amax >= 1) {
                    int64_t hs = amax * h1 + h0, ks = amax * k1 + k0;
                    if (fabs((double)hs / (double)ks - r) < fabs((double)h1 / (double)k1 - r)) { h1 = hs; k1 = ks; }
                }
                break;
            }
            h2 = ai * h1 + h0; k2 = ai * k1 + k0;
            h0 = h1; h1 = h2; k0 = k1; k1 = k2;
            if (fabs((double)h1 / (double)k1 - r) <= 1e-15 * r) break;
            if (x - a < 1e-300) break;
            x = 1. / (x - a);
        }
        if (k1 <= 0 || h1 <= 0) return -1;
        *L = h1; *M = k1;
        return 0;
    }
}

/* Modified Bessel function of the first kind, order 0 (power series, float64). */
static double bessel_i0(double x)
{
    double sum = 1., term = 1., q = x * x * .25;
    int k;
    for (k = 1; k < 500; ++k) {
        term *= q / ((double)k * (double)k);
        sum += term;
        if (term < sum * 1e-17) break;
    }
    return sum;
}

#define DESIGN_ATT_MARGIN_DB 2.0 /* Kaiser's estimate falls 1-3 dB short at the stop-band edge */

/* Geometry + Kaiser parameters.  taps per phase T is even and a multiple of 8. */
API int oracle_plan(double in_rate, double out_rate, unsigned long recipe, int64_t *L, int64_t *M,
                    int32_t *T, double *att_db, double *beta)
{
    double bits, pb, sb;
    if (oracle_quality(recipe, &bits, &pb, &sb)) return -1;
    if (oracle_ratio(in_rate, out_rate, L, M)) return -2;
    if (bits == 0.) { /* QQ: 4-point cubic (Lagrange) kernel, stretched when down-sampling */
        double s = *M > *L ? (double)*M / (double)*L : 1.;
        int t = (int)ceil(4. * s);
        *T = (t + 7) / 8 * 8;
        *att_db = 0.; *beta = 0.;
        return 0;
    }
    {
        double fn = .5 * (in_rate < out_rate ? in_rate : out_rate);
        double fs_hi = (double)*L * in_rate;
        double dw = 2. * M_PI * (sb - pb) * fn / fs_hi;
        double A = (bits + 1.) * 20. * log10(2.) + DESIGN_ATT_MARGIN_DB;
        double n_hi = (A - 7.95) / (2.285 * dw) + 1.;
        int64_t t = (int64_t)ceil(n_hi / (double)*L);
        if (t < 8) t = 8;
        if (t > (1 << 24)) return -3;
        *T = (int32_t)((t + 7) / 8 * 8);
        *att_db = A;
        *beta = .1102 * (A - 8.7);
    }
    return 0;
}

/* Bank, phase-major [L][T]:  bank[p][j] = g[L*(T/2-1-j) + p],  g = prototype at rate L*in_rate,
 * support m in [-L*T/2, L*T/2).  Output k sits at input time k*M/L exactly (zero latency):
 *   y[k] = sum_j bank[(k*M) mod L][j] * x[floor(k*M/L) - (T/2-1) + j]. */
API int oracle_design_bank(double in_rate, double out_rate, unsigned long recipe, double *bank)
{
    int64_t L, M, half, m, p;
    int32_t T;
    double att, beta, bits, pb, sb;
    if (oracle_plan(in_rate, out_rate, recipe, &L, &M, &T, &att, &beta)) return -1;
    oracle_quality(recipe, &bits, &pb, &sb);
    if (L * (int64_t)T > ((int64_t)1 << 22)) return -2; /* interpolated-phase plan: oracle_design_interp */
    half = L * (int64_t)T / 2;
    if (bits == 0.) {
        /* 4-point Lagrange cubic kernel k(t), |t| < 2 input samples, stretched by s >= 1;
         * each phase normalised to unit DC gain. */
        double s = M > L ? (double)M / (double)L : 1.;
        for (p = 0; p < L; ++p) {
            double sum = 0.;
            int j;
            for (j = 0; j < T; ++j) {
                double t = fabs((double)(L * ((int64_t)T / 2 - 1 - j) + p) / (double)L) / s, v;
                if (t < 1.) v = (1. - t * t) * (2. - t) * .5;
                else if (t < 2.) v = (1. - t) * (2. - t) * (3. - t) / 6.;
                else v = 0.;
                bank[p * T + j] = v;
                sum += v;
            }
            for (j = 0; j < T; ++j) bank[p * T + j] /= sum;
        }
        return 0;
    }
    {
        double fn = .5 * (in_rate < out_rate ? in_rate : out_rate);
        double fs_hi = (double)L * in_rate;
        double fc = .5 * (pb + sb) * fn / fs_hi; /* -6 dB point, cycles/sample at the high rate */
        double inv_i0 = 1. / bessel_i0(beta), inv_half = 1. / (double)half, sum = 0., scale;
        for (m = -half; m < half; ++m) {
            double u = (double)m * inv_half, w = 1. - u * u, a = 2. * M_PI * fc * (double)m, s, v;
            int64_t q = m + half, jj, pp;
            if (w < 0.) w = 0.;
            s = m == 0 ? 2. * fc : sin(a) / (M_PI * (double)m);
            v = s * bessel_i0(beta * sqrt(w)) * inv_i0;
            /* q = L*(T-1-j) + p */
            jj = (int64_t)T - 1 - q / L; pp = q % L;
            bank[pp * T + jj] = v;
            sum += v;
        }
        scale = (double)L / sum; /* mean DC gain over phases == 1 */
        for (m = 0; m < L * (int64_t)T; ++m) bank[m] *= scale;
    }
    return 0;
}

/* floor(n*L/M + 1/2) in exact integer arithmetic. */
API uint64_t oracle_out_len(uint64_t n_in, int64_t L, int64_t M)
{
    unsigned __int128 num = (unsigned __int128)n_in * (unsigned __int128)L * 2u + (unsigned __int128)M;
    return (uint64_t)(num / ((unsigned __int128)M * 2u));
}

/* ------------------------------------------------------------------------------------------ */
/* the inner product (one channel, planar).  x[i] is the sample with absolute index in_abs0+i;  */
/* the signal is zero outside [in_abs0, in_abs0+n_in).  Outputs k0 .. k0+n_out-1.               */
/* ------------------------------------------------------------------------------------------ */
static inline void locate(int64_t k, int64_t L, int64_t M, int32_t T, int64_t *n0, int64_t *p)
{
    __int128 kM = (__int128)k * M;
    *n0 = (int64_t)(kM / L) - (T / 2 - 1);
    *p = (int64_t)(kM % L);
}

API void oracle_resample_ref(const double *bank, int64_t L, int64_t M, int32_t T, const double *x,
                             int64_t in_abs0, int64_t n_in, double *y, int64_t k0, int64_t n_out)
{
    int64_t i;
    for (i = 0; i < n_out; ++i) {
        int64_t n0, p, j;
        const double *c;
        double acc = 0.;
        locate(k0 + i, L, M, T, &n0, &p);
        c = bank + p * T;
        for (j = 0; j < T; ++j) {
            int64_t a = n0 + j - in_abs0;
            if (a >= 0 && a < n_in) acc += c[j] * x[a];
        }
        y[i] = acc;
    }
}

API void oracle_resample_port_f64(const double *bank, int64_t L, int64_t M, int32_t T,
                                  const double *x, int64_t in_abs0, int64_t n_in, double *y,
                                  int64_t k0, int64_t n_out)
{
    int64_t i;
    for (i = 0; i < n_out; ++i) {
        int64_t n0, p, j;
        const double *c;
        double accL = 0., accR = 0.;
        locate(k0 + i, L, M, T, &n0, &p);
        c = bank + p * T;
        for (j = 0; j < T / 2; ++j) {
            int64_t a = n0 + j - in_abs0;
            double xv = (a >= 0 && a < n_in) ? x[a] : 0.;
            accL = fma(c[j], xv, accL);
        }
        for (j = T - 1; j >= T / 2; --j) {
            int64_t a = n0 + j - in_abs0;
            double xv = (a >= 0 && a < n_in) ? x[a] : 0.;
            accR = fma(c[j], xv, accR);
        }
        y[i] = accL + accR;
    }
}

/* f32 engine: coefficients are the float64 bank rounded to nearest float32. */
API void oracle_resample_port_f32(const double *bank, int64_t L, int64_t M, int32_t T,
                                  const float *x, int64_t in_abs0, int64_t n_in, float *y,
                                  int64_t k0, int64_t n_out)
{
    int64_t i, n = L * (int64_t)T;
    float *cf = (float *)malloc((size_t)n * sizeof(float));
    for (i = 0; i < n; ++i) cf[i] = (float)bank[i];
    for (i = 0; i < n_out; ++i) {
        int64_t n0, p, j;
        const float *c;
        float accL = 0.f, accR = 0.f;
        locate(k0 + i, L, M, T, &n0, &p);
        c = cf + p * T;
        for (j = 0; j < T / 2; ++j) {
            int64_t a = n0 + j - in_abs0;
            float xv = (a >= 0 && a < n_in) ? x[a] : 0.f;
            accL = fmaf(c[j], xv, accL);
        }
        for (j = T - 1; j >= T / 2; --j) {
            int64_t a = n0 + j - in_abs0;
            float xv = (a >= 0 && a < n_in) ? x[a] : 0.f;
            accR = fmaf(c[j], xv, accR);
        }
        y[i] = accL + accR;
    }
    free(cf);
}

/* ------------------------------------------------------------------------------------------ */
/* interpolated-phase plans: ratios whose exact bank would be too large (L*T > 2^22 entries),    */
/* e.g. random integer or float rates (/root/reference/tests/test_random.py:21-26).               */
/* ------------------------------------------------------------------------------------------ */
/* Same prototype, now as a function of continuous time tau (input samples):
 *     h(tau) = 2fc sinc(2 fc tau) * I0(beta sqrt(1-(tau/W)^2)) / I0(beta),  |tau| < W = T/2,
 *     fc = L * (fc of the exact design) cycles per input sample,
 * normalised to unit DC gain (sum over a 64x over-sampled grid).  Tap j of an output at fractional
 * position f = ((k*M) mod L)/L uses c_j(f) = h(f + T/2-1-j).  The fraction axis is cut into P
 * intervals (16 / 32 / 128 for <=16 / 20 / 28 bits, 256 for QQ); on interval i each tap is the cubic
 * through h at the 4 Chebyshev nodes of the interval, stored as monomial coefficients
 *     coef[i][j][0..3]:   c_j = a0 + x (a1 + x (a2 + x a3)),   x = f*P - i  in [0,1).
 * Interpolation error (measured, tests/test_oracle_pinning.py) is >30x below 2^-(bits+1).
 * x is quantised to 24 (f32 engine) or 32 (f64 engine) bits with integer arithmetic, so that oracle
 * and GPU derive bit-identical coefficients:
 *     r = (k*M) mod L;  i = floor(r*P/L);  x = floor(((r*P) mod L) * 2^SH / L) * 2^-SH.        */
#define EXACT_BANK_MAX_ELEMS ((int64_t)1 << 22)
#define INTERP_GRID 64

/* 0 for an exact-bank plan, else the number of phase intervals P. */
API int32_t oracle_plan_phases(double in_rate, double out_rate, unsigned long recipe)
{
    int64_t L, M;
    int32_t T;
    double att, beta, bits, pb, sb;
    if (oracle_plan(in_rate, out_rate, recipe, &L, &M, &T, &att, &beta)) return -1;
    oracle_quality(recipe, &bits, &pb, &sb);
    if (L * (int64_t)T <= EXACT_BANK_MAX_ELEMS) return 0;
    return bits == 0. ? 256 : bits <= 16. ? 16 : bits <= 20. ? 32 : 128;
}

typedef struct {
    double fc, W, beta, inv_i0, s; /* s: QQ stretch */
    int qq;
} proto_t;

static double proto_eval(const proto_t *pr, double tau)
{
    if (pr->qq) {
        double t = fabs(tau) / pr->s;
        if (t < 1.) return (1. - t * t) * (2. - t) * .5;
        if (t < 2.) return (1. - t) * (2. - t) * (3. - t) / 6.;
        return 0.;
    } else {
        double u = tau / pr->W, w = 1. - u * u, sv;
        if (w < 0.) w = 0.;
        sv = tau == 0. ? 2. * pr->fc : sin(2. * M_PI * pr->fc * tau) / (M_PI * tau);
        return sv * bessel_i0(pr->beta * sqrt(w)) * pr->inv_i0;
    }
}

static int proto_setup(double in_rate, double out_rate, unsigned long recipe, proto_t *pr,
                       int64_t *L, int64_t *M, int32_t *T, double *scale)
{
    double att, beta, bits, pb, sb;
    if (oracle_plan(in_rate, out_rate, recipe, L, M, T, &att, &beta)) return -1;
    oracle_quality(recipe, &bits, &pb, &sb);
    memset(pr, 0, sizeof *pr);
    pr->qq = bits == 0.;
    pr->W = .5 * (double)*T;
    if (pr->qq) {
        pr->s = *M > *L ? (double)*M / (double)*L : 1.;
        *scale = 1.;
    } else {
        double fn = .5 * (in_rate < out_rate ? in_rate : out_rate), sum = 0.;
        int64_t half = (int64_t)INTERP_GRID * *T / 2, m;
        pr->fc = .5 * (pb + sb) * fn / in_rate;
        pr->beta = beta;
        pr->inv_i0 = 1. / bessel_i0(beta);
        for (m = -half; m < half; ++m) sum += proto_eval(pr, (double)m / (double)INTERP_GRID);
        *scale = (double)INTERP_GRID / sum;
    }
    return 0;
}

/* Exact (un-interpolated) coefficients of fraction f in [0,1): c[j] = h(f + T/2-1-j), j < T. */
API int oracle_interp_exact_coefs(double in_rate, double out_rate, unsigned long recipe, double f,
                                  double *c)
{
    proto_t pr;
    int64_t L, M;
    int32_t T, j;
    double scale, sum = 0.;
    if (proto_setup(in_rate, out_rate, recipe, &pr, &L, &M, &T, &scale)) return -1;
    for (j = 0; j < T; ++j) {
        c[j] = proto_eval(&pr, f + (double)(T / 2 - 1 - j)) * scale;
        sum += c[j];
    }
    if (pr.qq) for (j = 0; j < T; ++j) c[j] /= sum; /* QQ: every fraction has unit DC gain */
    return 0;
}

/* coef[P][T][4] (float64). */
static int design_interp_table(double in_rate, double out_rate, unsigned long recipe, int32_t P,
                               double *coef)
{
    proto_t pr;
    int64_t L, M;
    int32_t T, i, j, c;
    double scale, xn[4], *v;
    if (proto_setup(in_rate, out_rate, recipe, &pr, &L, &M, &T, &scale)) return -1;
    for (c = 0; c < 4; ++c) xn[c] = .5 - .5 * cos((double)(2 * c + 1) * M_PI / 8.);
    v = (double *)malloc((size_t)T * 4 * sizeof(double));
    for (i = 0; i < P; ++i) {
        for (c = 0; c < 4; ++c) {
            double f = ((double)i + xn[c]) / (double)P, sum = 0.;
            for (j = 0; j < T; ++j) {
                v[c * T + j] = proto_eval(&pr, f + (double)(T / 2 - 1 - j)) * scale;
                sum += v[c * T + j];
            }
            if (pr.qq) for (j = 0; j < T; ++j) v[c * T + j] /= sum;
        }
        for (j = 0; j < T; ++j) {
            /* Newton divided differences on the nodes, then expansion to monomials */
            double v0 = v[j], v1 = v[T + j], v2 = v[2 * T + j], v3 = v[3 * T + j];
            double d01 = (v1 - v0) / (xn[1] - xn[0]), d12 = (v2 - v1) / (xn[2] - xn[1]),
                   d23 = (v3 - v2) / (xn[3] - xn[2]);
            double d012 = (d12 - d01) / (xn[2] - xn[0]), d123 = (d23 - d12) / (xn[3] - xn[1]);
            double d3 = (d123 - d012) / (xn[3] - xn[0]);
            double *a = coef + ((size_t)i * T + j) * 4;
            a[3] = d3;
            a[2] = d012 - d3 * (xn[0] + xn[1] + xn[2]);
            a[1] = d01 - d012 * (xn[0] + xn[1]) + d3 * (xn[0] * xn[1] + xn[0] * xn[2] + xn[1] * xn[2]);
            a[0] = v0 - d01 * xn[0] + d012 * (xn[0] * xn[1]) - d3 * (xn[0] * xn[1] * xn[2]);
        }
    }
    free(v);
    return 0;
}

API int oracle_design_interp(double in_rate, double out_rate, unsigned long recipe, double *coef)
{
    int32_t P = oracle_plan_phases(in_rate, out_rate, recipe);
    if (P <= 0) return -1;
    return design_interp_table(in_rate, out_rate, recipe, P, coef);
}

/* position of output k: first tap's absolute input index, interval, quantised residual */
static inline void locate_interp(int64_t k, int64_t L, int64_t M, int32_t T, int32_t P, int sh,
                                 int64_t *n0, int32_t *iv, uint64_t *xq)
{
    __int128 kM = (__int128)k * M;
    uint64_t r = (uint64_t)(kM % L), t = r * (uint64_t)P, rem;
    *n0 = (int64_t)(kM / L) - (T / 2 - 1);
    *iv = (int32_t)(t / (uint64_t)L);
    rem = t % (uint64_t)L;
    *xq = (rem << sh) / (uint64_t)L;
}

API void oracle_interp_ref(const double *coef, int32_t P, int64_t L, int64_t M, int32_t T,
                           const double *x, int64_t in_abs0, int64_t n_in, double *y, int64_t k0,
                           int64_t n_out)
{
    int64_t i;
    for (i = 0; i < n_out; ++i) {
        int64_t n0, j;
        int32_t iv;
        uint64_t xq;
        double acc = 0., xx;
        const double *a;
        locate_interp(k0 + i, L, M, T, P, 32, &n0, &iv, &xq);
        xx = (double)xq * (1. / 4294967296.);
        a = coef + (size_t)iv * T * 4;
        for (j = 0; j < T; ++j) {
            int64_t q = n0 + j - in_abs0;
            if (q >= 0 && q < n_in)
                acc += (a[4 * j] + xx * (a[4 * j + 1] + xx * (a[4 * j + 2] + xx * a[4 * j + 3]))) * x[q];
        }
        y[i] = acc;
    }
}

/* canonical order, f64 engine: coefficient by fma-Horner, then the two half-chains */
API void oracle_interp_port_f64(const double *coef, int32_t P, int64_t L, int64_t M, int32_t T,
                                const double *x, int64_t in_abs0, int64_t n_in, double *y, int64_t k0,
                                int64_t n_out)
{
    int64_t i;
    for (i = 0; i < n_out; ++i) {
        int64_t n0, j;
        int32_t iv;
        uint64_t xq;
        double accL = 0., accR = 0., xx;
        const double *a;
        locate_interp(k0 + i, L, M, T, P, 32, &n0, &iv, &xq);
        xx = (double)xq * (1. / 4294967296.);
        a = coef + (size_t)iv * T * 4;
        for (j = 0; j < T / 2; ++j) {
            int64_t q = n0 + j - in_abs0;
            double xv = (q >= 0 && q < n_in) ? x[q] : 0.;
            double c = fma(fma(fma(a[4 * j + 3], xx, a[4 * j + 2]), xx, a[4 * j + 1]), xx, a[4 * j]);
            accL = fma(c, xv, accL);
        }
        for (j = T - 1; j >= T / 2; --j) {
            int64_t q = n0 + j - in_abs0;
            double xv = (q >= 0 && q < n_in) ? x[q] : 0.;
            double c = fma(fma(fma(a[4 * j + 3], xx, a[4 * j + 2]), xx, a[4 * j + 1]), xx, a[4 * j]);
            accR = fma(c, xv, accR);
        }
        y[i] = accL + accR;
    }
}

/* f32 engine: polynomial coefficients rounded to float32, Horner and chains in float32 */
API void oracle_interp_port_f32(const double *coef, int32_t P, int64_t L, int64_t M, int32_t T,
                                const float *x, int64_t in_abs0, int64_t n_in, float *y, int64_t k0,
                                int64_t n_out)
{
    int64_t i, n = (int64_t)P * T * 4;
    float *cf = (float *)malloc((size_t)n * sizeof(float));
    for (i = 0; i < n; ++i) cf[i] = (float)coef[i];
    for (i = 0; i < n_out; ++i) {
        int64_t n0, j;
        int32_t iv;
        uint64_t xq;
        float accL = 0.f, accR = 0.f, xx;
        const float *a;
        locate_interp(k0 + i, L, M, T, P, 24, &n0, &iv, &xq);
        xx = (float)xq * (1.f / 16777216.f);
        a = cf + (size_t)iv * T * 4;
        for (j = 0; j < T / 2; ++j) {
            int64_t q = n0 + j - in_abs0;
            float xv = (q >= 0 && q < n_in) ? x[q] : 0.f;
            float c = fmaf(fmaf(fmaf(a[4 * j + 3], xx, a[4 * j + 2]), xx, a[4 * j + 1]), xx, a[4 * j]);
            accL = fmaf(c, xv, accL);
        }
        for (j = T - 1; j >= T / 2; --j) {
            int64_t q = n0 + j - in_abs0;
            float xv = (q >= 0 && q < n_in) ? x[q] : 0.f;
            float c = fmaf(fmaf(fmaf(a[4 * j + 3], xx, a[4 * j + 2]), xx, a[4 * j + 1]), xx, a[4 * j]);
            accR = fmaf(c, xv, accR);
        }
        y[i] = accL + accR;
    }
    free(cf);
}

/* ------------------------------------------------------------------------------------------ */
/* variable rate (SOXR_VR streams, /root/reference/src/soxr_ext.cpp:74, :200-204; usage pattern    */
/* /root/reference/tests/vr.py:60-114).  Untested in the reference: only the API shape is pinned. */
/* ------------------------------------------------------------------------------------------ */
/* Same interpolated-phase table, designed for the largest io ratio.  Time is Q64.64 fixed point;
 * one call covers outputs i = 0..n_out-1 whose input positions are the quadratic
 *     t(i) = T0 + i*S0 + D*i(i-1)/2          (128-bit two's-complement integers, hi:lo words)
 * (D = 0: constant ratio; D != 0: the step slews linearly).  P is a power of two: interval =
 * top log2(P) bits of the fraction, residual = the next 24 (f32) / 32 (f64) bits.  The schedule
 * of (T0, S0, D) across ratio changes is restated by the tests in Python integers. */
typedef unsigned __int128 u128;

static inline void locate_vr(int64_t i, uint64_t t_hi, uint64_t t_lo, uint64_t s_hi, uint64_t s_lo,
                             uint64_t d_hi, uint64_t d_lo, int32_t T, int lgP, int sh, int64_t *n0,
                             int32_t *iv, uint64_t *xq)
{
    u128 T0 = ((u128)t_hi << 64) | t_lo, S0 = ((u128)s_hi << 64) | s_lo, D = ((u128)d_hi << 64) | d_lo;
    u128 n = (u128)(uint64_t)i, m = n * (n - 1) / 2; /* n = 0 -> 0 */
    u128 t = T0 + n * S0 + D * m;                    /* modular: two's-complement D */
    uint64_t frac = (uint64_t)t;
    *n0 = (int64_t)(uint64_t)(t >> 64) - (T / 2 - 1);
    *iv = lgP ? (int32_t)(frac >> (64 - lgP)) : 0;
    *xq = (frac << lgP) >> (64 - sh);
}

#define VR_ARGS uint64_t t_hi, uint64_t t_lo, uint64_t s_hi, uint64_t s_lo, uint64_t d_hi, uint64_t d_lo
#define VR_PASS t_hi, t_lo, s_hi, s_lo, d_hi, d_lo

static int ilog2(int32_t P) { int l = 0; while ((1 << l) < P) ++l; return l; }

API void oracle_vr_ref(const double *coef, int32_t P, int32_t T, const double *x, int64_t in_abs0,
                       int64_t n_in, double *y, int64_t n_out, VR_ARGS)
{
    int64_t i;
    int lgP = ilog2(P);
    for (i = 0; i < n_out; ++i) {
        int64_t n0, j;
        int32_t iv;
        uint64_t xq;
        double acc = 0., xx;
        const double *a;
        locate_vr(i, VR_PASS, T, lgP, 32, &n0, &iv, &xq);
        xx = (double)xq * (1. / 4294967296.);
        a = coef + (size_t)iv * T * 4;
        for (j = 0; j < T; ++j) {
            int64_t q = n0 + j - in_abs0;
            if (q >= 0 && q < n_in)
                acc += (a[4 * j] + xx * (a[4 * j + 1] + xx * (a[4 * j + 2] + xx * a[4 * j + 3]))) * x[q];
        }
        y[i] = acc;
    }
}

API void oracle_vr_port_f64(const double *coef, int32_t P, int32_t T, const double *x, int64_t in_abs0,
                            int64_t n_in, double *y, int64_t n_out, VR_ARGS)
{
    int64_t i;
    int lgP = ilog2(P);
    for (i = 0; i < n_out; ++i) {
        int64_t n0, j;
        int32_t iv;
        uint64_t xq;
        double accL = 0., accR = 0., xx;
        const double *a;
        locate_vr(i, VR_PASS, T, lgP, 32, &n0, &iv, &xq);
        xx = (double)xq * (1. / 4294967296.);
        a = coef + (size_t)iv * T * 4;
        for (j = 0; j < T / 2; ++j) {
            int64_t q = n0 + j - in_abs0;
            double xv = (q >= 0 && q < n_in) ? x[q] : 0.;
            accL = fma(fma(fma(fma(a[4 * j + 3], xx, a[4 * j + 2]), xx, a[4 * j + 1]), xx, a[4 * j]), xv, accL);
        }
        for (j = T - 1; j >= T / 2; --j) {
            int64_t q = n0 + j - in_abs0;
            double xv = (q >= 0 && q < n_in) ? x[q] : 0.;
            accR = fma(fma(fma(fma(a[4 * j + 3], xx, a[4 * j + 2]), xx, a[4 * j + 1]), xx, a[4 * j]), xv, accR);
        }
        y[i] = accL + accR;
    }
}

API void oracle_vr_port_f32(const double *coef, int32_t P, int32_t T, const float *x, int64_t in_abs0,
                            int64_t n_in, float *y, int64_t n_out, VR_ARGS)
{
    int64_t i, n = (int64_t)P * T * 4;
    int lgP = ilog2(P);
    float *cf = (float *)malloc((size_t)n * sizeof(float));
    for (i = 0; i < n; ++i) cf[i] = (float)coef[i];
    for (i = 0; i < n_out; ++i) {
        int64_t n0, j;
        int32_t iv;
        uint64_t xq;
        float accL = 0.f, accR = 0.f, xx;
        const float *a;
        locate_vr(i, VR_PASS, T, lgP, 24, &n0, &iv, &xq);
        xx = (float)xq * (1.f / 16777216.f);
        a = cf + (size_t)iv * T * 4;
        for (j = 0; j < T / 2; ++j) {
            int64_t q = n0 + j - in_abs0;
            float xv = (q >= 0 && q < n_in) ? x[q] : 0.f;
            accL = fmaf(fmaf(fmaf(fmaf(a[4 * j + 3], xx, a[4 * j + 2]), xx, a[4 * j + 1]), xx, a[4 * j]), xv, accL);
        }
        for (j = T - 1; j >= T / 2; --j) {
            int64_t q = n0 + j - in_abs0;
            float xv = (q >= 0 && q < n_in) ? x[q] : 0.f;
            accR = fmaf(fmaf(fmaf(fmaf(a[4 * j + 3], xx, a[4 * j + 2]), xx, a[4 * j + 1]), xx, a[4 * j]), xv, accR);
        }
        y[i] = accL + accR;
    }
    free(cf);
}

/* The table a VR stream uses: always interpolated-phase, whatever the ratio.  coef == NULL: only
 * report T and P. */
API int oracle_design_vr(double in_rate, double out_rate, unsigned long recipe, int32_t *T_out,
                         int32_t *P_out, double *coef)
{
    int64_t L, M;
    int32_t T;
    double att, beta, bits, pb, sb;
    if (oracle_plan(in_rate, out_rate, recipe, &L, &M, &T, &att, &beta)) return -1;
    oracle_quality(recipe, &bits, &pb, &sb);
    *T_out = T;
    *P_out = bits == 0. ? 256 : bits <= 16. ? 16 : bits <= 20. ? 32 : 128;
    return coef ? design_interp_table(in_rate, out_rate, recipe, *P_out, coef) : 0;
}

/* ------------------------------------------------------------------------------------------ */
/* integer output                                                                              */
/* ------------------------------------------------------------------------------------------ */
/* Counter-based TPDF dither in (-1, 1) LSB: a pure function of (seed, channel, absolute output
 * index), so results do not depend on chunking, threads or launch geometry. */
static inline uint64_t mix64(uint64_t z)
{
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27; z *= 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return z;
}
API double oracle_dither(uint32_t seed, uint32_t channel, int64_t k)
{
    uint64_t z = mix64((uint64_t)k * 0x9E3779B97F4A7C15ULL + (((uint64_t)channel << 32) | seed));
    int32_t u1 = (int32_t)(z & 0xFFFFFF), u2 = (int32_t)((z >> 24) & 0xFFFFFF);
    return (double)(u1 - u2) * (1. / 16777216.);
}

/* f32 engine -> int16: v (+ dither, added in f32), rintf (half-even), saturate, count clips. */
API uint64_t oracle_quantize_i16(const float *v, int64_t n, int dither, uint32_t seed,
                                 uint32_t channel, int64_t k0, int16_t *out)
{
    uint64_t clips = 0;
    int64_t i;
    for (i = 0; i < n; ++i) {
        float a = v[i], r;
        if (dither) a = a + (float)oracle_dither(seed, channel, k0 + i);
        r = rintf(a);
        if (r > 32767.f) { r = 32767.f; ++clips; }
        else if (r < -32768.f) { r = -32768.f; ++clips; }
        out[i] = (int16_t)r;
    }
    return clips;
}

/* f64 engine -> int32: rint (half-even), saturate, count clips (no dither). */
API uint64_t oracle_quantize_i32(const double *v, int64_t n, int32_t *out)
{
    uint64_t clips = 0;
    int64_t i;
    for (i = 0; i < n; ++i) {
        double r = rint(v[i]);
        if (r > 2147483647.) { r = 2147483647.; ++clips; }
        else if (r < -2147483648.) { r = -2147483648.; ++clips; }
        out[i] = (int32_t)r;
    }
    return clips;
}
