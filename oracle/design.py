"""Filter design of the CPU oracle — numpy/scipy formulation.  TEST INFRASTRUCTURE ONLY.

Everything the product designs on the host in C++ (python-soxr_amd/csrc/plan.cpp: recipe -> spec,
rate pair -> L/M, Kaiser-windowed-sinc prototype -> polyphase bank, the cubic phase-interpolation
tables) is restated here FROM THE SPECIFICATION in a different
formulation and with different building blocks, so that a comparison of the two is a comparison of
two implementations and not of one piece of code with itself:

    product (plan.cpp)                          here
    ----------------------------------------    -------------------------------------------------
    gcd loop / double-precision continued       fractions.Fraction + exact integer continued
    fraction with 1/(x - a) iteration           fraction (no floating-point state)
    power-series I0 with a term-ratio stop      scipy.special.i0e (Cephes Chebyshev expansions)
    sin(a)/(pi m), special case at m == 0       numpy.sinc
    scalar loop scattering g[m] into            one vectorised prototype, reshaped [T][L] and
    bank[(m+half) % L][T-1-(m+half)/L]          flipped/transposed into [L][T]
    running sum, scale = L / sum                mean over phases of the per-phase DC gains
    Newton divided differences -> monomials     4 x 4 Vandermonde solve at the Chebyshev nodes

The reference itself holds no filter code (libsoxr is an empty submodule: /root/reference/.gitmodules:1-3),
so what is restated is libsoxr's documented recipe numbers (precision bits, pass-band end,
stop-band begin; python-soxr exports the recipe ids at /root/reference/src/soxr_ext.cpp:447-451)
and the textbook Kaiser design rule.  Agreement with the product is to ~1e-13 of the coefficient
scale, not bit for bit (tests/test_design_independent.py); bit-exact comparisons of the ARITHMETIC
take the product's own float64 bank as an input (oracle.bank_provider).
"""
from fractions import Fraction
import math

import numpy as np
from scipy.special import i0e

ATT_MARGIN_DB = 2.0          # Kaiser's length estimate falls 1-3 dB short at the stop-band edge
EXACT_BANK_MAX = 1 << 22     # larger L*T: interpolated-phase plan
LIM31 = 2147483647
GRID = 64                    # over-sampling of the DC-gain normalisation of h(tau)


def quality(recipe):
    """recipe id -> (precision bits, pass-band end, stop-band begin), the band edges as fractions of
    the lower rate's Nyquist frequency."""
    q = int(recipe) & 0xF
    if q > 7:
        raise ValueError("invalid quality recipe")
    bits = 0.0 if q == 0 else 16.0 if q < 4 else 4.0 + 4.0 * q
    if q == 0:
        pb = 0.0
    elif q == 1:
        pb = 1385.0 / 2048.0
    else:
        rej = bits * 20.0 * math.log10(2.0)
        pb = 1.0 - 0.05 / ((1.6e-6 * rej - 7.5e-4) * rej + 0.646)
    return bits, pb, 1.0


def ratio(in_rate, out_rate):
    """out/in = L/M in lowest terms.  Integral rates: exact.  Otherwise the first continued-fraction
    convergent of the double quotient out_rate/in_rate that reproduces it to 1e-15 relative, or —
    when the next convergent would leave the 31-bit range first — the best semiconvergent inside
    the range.  The expansion runs in exact rational arithmetic (the product iterates x -> 1/(x - a)
    in doubles); the acceptance test is the specified double-precision one, |fl(h/k) - r| <= 1e-15 r,
    because at that tolerance (4.5 ulp) an exact test and a rounded one can disagree by one
    convergent, and the plan geometry must be the same on both sides."""
    if not (in_rate > 0 and out_rate > 0):
        raise ValueError("sample rate must be > 0")
    if float(in_rate).is_integer() and float(out_rate).is_integer() and in_rate < 9e15 and out_rate < 9e15:
        f = Fraction(int(out_rate), int(in_rate))
        return f.numerator, f.denominator
    r = float(out_rate) / float(in_rate)       # the double the product starts from
    target = Fraction(r)
    x = target
    h0, h1, k0, k1 = 0, 1, 1, 0
    for _ in range(64):
        a = x.numerator // x.denominator
        over = a > LIM31 or (h1 and a > (LIM31 - h0) // h1) or (k1 and a > (LIM31 - k0) // k1)
        if over:
            if k1 == 0:
                raise ValueError("rate ratio is out of range")
            amax = LIM31
            if h1:
                amax = min(amax, (LIM31 - h0) // h1)
            amax = min(amax, (LIM31 - k0) // k1)
            if amax >= 1:
                hs, ks = amax * h1 + h0, amax * k1 + k0
                if abs(float(hs) / float(ks) - r) < abs(float(h1) / float(k1) - r):
                    h1, k1 = hs, ks
            break
        h0, h1, k0, k1 = h1, a * h1 + h0, k1, a * k1 + k0
        if abs(float(h1) / float(k1) - r) <= 1e-15 * r:
            break
        frac = x - a
        if frac == 0:
            break
        x = 1 / frac
    if h1 <= 0 or k1 <= 0:
        raise ValueError("rate ratio is out of range")
    return h1, k1


def geometry(in_rate, out_rate, recipe):
    """-> dict(L, M, T, att_db, beta, phases, bits, pb, sb).  T: taps per phase, a multiple of 8."""
    bits, pb, sb = quality(recipe)
    L, M = ratio(in_rate, out_rate)
    if bits == 0.0:
        s = M / L if M > L else 1.0
        T = (int(math.ceil(4.0 * s)) + 7) // 8 * 8
        att = beta = 0.0
    else:
        fn = 0.5 * min(in_rate, out_rate)
        dw = 2.0 * math.pi * (sb - pb) * fn / (L * in_rate)       # transition width, rad/sample at L*in_rate
        att = (bits + 1.0) * 20.0 * math.log10(2.0) + ATT_MARGIN_DB
        n_hi = (att - 7.95) / (2.285 * dw) + 1.0                  # Kaiser's length estimate
        t = max(8, int(math.ceil(n_hi / L)))
        if t > (1 << 24):
            raise ValueError("rate ratio needs too many taps")
        T = (t + 7) // 8 * 8
        beta = 0.1102 * (att - 8.7)
    phases = 0
    if L * T > EXACT_BANK_MAX:
        phases = 256 if bits == 0.0 else 16 if bits <= 16 else 32 if bits <= 20 else 128
    return dict(L=L, M=M, T=T, att_db=att, beta=beta, phases=phases, bits=bits, pb=pb, sb=sb)


def _kaiser_ratio(beta, w):
    """I0(beta*sqrt(w)) / I0(beta) for w in [0, 1], via the exponentially scaled I0."""
    a = beta * np.sqrt(np.clip(w, 0.0, None))
    return i0e(a) / i0e(beta) * np.exp(a - beta)


def _lagrange(t):
    """4-point Lagrange (cubic) interpolation kernel, |t| in input samples."""
    t = np.abs(t)
    return np.where(t < 1.0, (1.0 - t * t) * (2.0 - t) * 0.5,
                    np.where(t < 2.0, (1.0 - t) * (2.0 - t) * (3.0 - t) / 6.0, 0.0))


def bank(in_rate, out_rate, recipe):
    """Exact rational plan: float64 bank [L][T], bank[p][j] = g[L*(T/2-1-j) + p] with g the prototype
    at rate L*in_rate on the support m in [-L*T/2, L*T/2); mean DC gain over the phases == 1."""
    g = geometry(in_rate, out_rate, recipe)
    L, M, T = g["L"], g["M"], g["T"]
    if g["phases"]:
        raise ValueError("interpolated-phase plan: use interp_table()")
    half = L * T // 2
    m = np.arange(-half, half, dtype=np.float64)
    if g["bits"] == 0.0:
        s = M / L if M > L else 1.0
        proto = _lagrange(m / L / s)
        b = proto.reshape(T, L)[::-1].T.copy()                    # [p][j] = proto[(T-1-j)*L + p]
        return b / b.sum(axis=1, keepdims=True)                   # every phase: unit DC gain
    fn = 0.5 * min(in_rate, out_rate)
    fc = 0.5 * (g["pb"] + g["sb"]) * fn / (L * in_rate)           # -6 dB point, cycles/sample at L*in_rate
    proto = 2.0 * fc * np.sinc(2.0 * fc * m) * _kaiser_ratio(g["beta"], 1.0 - (m / half) ** 2)
    b = proto.reshape(T, L)[::-1].T.copy()
    return b / b.sum(axis=1).mean()


class _Proto:
    """The prototype as a function of continuous time tau (input samples), unit DC gain."""

    def __init__(self, in_rate, out_rate, recipe, g):
        self.T, self.cubic = g["T"], g["bits"] == 0.0
        if self.cubic:
            self.stretch = g["M"] / g["L"] if g["M"] > g["L"] else 1.0
            self.scale = 1.0
        else:
            fn = 0.5 * min(in_rate, out_rate)
            self.fc = 0.5 * (g["pb"] + g["sb"]) * fn / in_rate
            self.beta, self.W = g["beta"], 0.5 * g["T"]
            half = GRID * g["T"] // 2
            self.scale = 1.0
            self.scale = GRID / self(np.arange(-half, half) / GRID).sum()

    def __call__(self, tau):
        tau = np.asarray(tau, np.float64)
        if self.cubic:
            return _lagrange(tau / self.stretch)
        return self.scale * 2.0 * self.fc * np.sinc(2.0 * self.fc * tau) * \
            _kaiser_ratio(self.beta, 1.0 - (tau / self.W) ** 2)

    def taps(self, f):
        """c[..., j] = h(f + T/2-1-j); QQ: every fraction normalised to unit DC gain."""
        f = np.asarray(f, np.float64)
        c = self(f[..., None] + (self.T // 2 - 1 - np.arange(self.T)))
        return c / c.sum(axis=-1, keepdims=True) if self.cubic else c


def exact_coefs(in_rate, out_rate, recipe, f):
    g = geometry(in_rate, out_rate, recipe)
    return _Proto(in_rate, out_rate, recipe, g).taps(float(f))


def interp_table(in_rate, out_rate, recipe, phases=None):
    """Interpolated-phase plan: [P][T][4] monomial coefficients (a0..a3) of the cubic through h at the
    four Chebyshev nodes of each of the P intervals of the fraction axis."""
    g = geometry(in_rate, out_rate, recipe)
    P = phases or g["phases"]
    if not P:
        raise ValueError("exact plan: use bank()")
    pr = _Proto(in_rate, out_rate, recipe, g)
    nodes = 0.5 - 0.5 * np.cos((2 * np.arange(4) + 1) * np.pi / 8.0)
    f = (np.arange(P)[:, None] + nodes[None, :]) / P              # [P][4]
    v = pr.taps(f)                                                # [P][4][T]
    V = np.vander(nodes, 4, increasing=True)                      # V @ a = v
    a = np.linalg.solve(V, v.transpose(1, 0, 2).reshape(4, -1))   # [4][P*T]
    return np.ascontiguousarray(a.reshape(4, P, g["T"]).transpose(1, 2, 0))


def vr_phases(recipe):
    bits = quality(recipe)[0]
    return 256 if bits == 0.0 else 16 if bits <= 16 else 32 if bits <= 20 else 128
