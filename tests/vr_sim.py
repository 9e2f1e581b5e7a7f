"""Test-side restatement of the variable-rate schedule (engine.cpp, VrState) in Python integers.

The oracle (oracle/soxr_oracle.c, oracle_vr_*) evaluates one quadratic position law per call; which
(T0, S0, D) applies to which outputs — i.e. what set_io_ratio does to the output clock, how many
outputs a given amount of input allows, where the stream ends — is restated here independently of
the C++ engine, with exact integers:

    t(k_s + n) = t_s + n*s0 + delta*n(n-1)/2                 n <= n_slew
               = t(k_s + n_slew) + (n - n_slew)*s1           n >  n_slew        (Q64.64)
"""
from fractions import Fraction

import numpy as np

ONE = 1 << 64


def q64(x):
    return int(Fraction(float(x)) * ONE)        # truncation of an exactly scaled double


def tdiv(a, b):
    """C-style division (truncation toward zero), b > 0."""
    return a // b if a >= 0 else -((-a) // b)


class VrSim:
    def __init__(self, oracle, in_rate, out_rate, quality, dtype):
        self.o = oracle
        self.vp = oracle.VrPlan(in_rate, out_rate, quality)
        self.H = self.vp.T // 2
        self.dtype = np.dtype(dtype)
        self.eng = oracle.engine_of(dtype)
        self.k_s = self.n_slew = 0
        self.t_s = self.delta = 0
        self.s0 = self.s1 = q64(float(in_rate) / float(out_rate))
        self.x = np.zeros(0, np.float64)
        self.k_done = 0

    def pos(self, k):
        n = k - self.k_s
        if n <= self.n_slew:
            return self.t_s + n * self.s0 + self.delta * (n * (n - 1) // 2)
        N = self.n_slew
        return self.t_s + N * self.s0 + self.delta * (N * (N - 1) // 2) + (n - N) * self.s1

    def step(self, k):
        n = k - self.k_s
        return self.s0 + n * self.delta if n < self.n_slew else self.s1

    def set_io_ratio(self, io_ratio, slew_len=0):
        t_now, s_now, s_new = self.pos(self.k_done), self.step(self.k_done), q64(io_ratio)
        self.k_s, self.t_s, self.s1 = self.k_done, t_now, s_new
        if slew_len > 0:
            self.s0, self.n_slew, self.delta = s_now, slew_len, tdiv(s_new - s_now, slew_len)
        else:
            self.s0, self.n_slew, self.delta = s_new, 0, 0

    def _limit(self, ended):
        n_in = len(self.x)
        k = self.k_done
        if ended:
            ok = lambda k: self.pos(k) + tdiv(self.step(k), 2) <= n_in * ONE
        else:
            ok = lambda k: (self.pos(k) >> 64) + self.H <= n_in - 1
        if not ok(k):
            return k
        lo, span = k, 1
        while ok(lo + span):
            lo, span = lo + span, span * 2
        hi = lo + span
        while hi - lo > 1:
            mid = (lo + hi) // 2
            lo, hi = (mid, hi) if ok(mid) else (lo, mid)
        return hi

    def feed(self, chunk, last=False, channel=0, mode="port"):
        """One mono chunk in; the outputs a stream would return for it (all of them: no output cap)."""
        self.x = np.concatenate([self.x, np.asarray(chunk, np.float64)])
        k_end = self._limit(last)
        outs = []
        while self.k_done < k_end:
            slew_end = self.k_s + self.n_slew
            stop = min(k_end, slew_end) if (self.n_slew and self.k_done < slew_end) else k_end
            n = stop - self.k_done
            D = self.delta if self.k_done < slew_end else 0
            m = "ref" if mode == "ref" else "port_" + self.eng
            v = self.o.vr_run(self.vp, self.x, m, n, self.pos(self.k_done), self.step(self.k_done), D)
            if mode == "port":
                v, _ = self.o.quantize(v, self.dtype, channel=channel, k0=self.k_done)
            outs.append(v)
            self.k_done = stop
        if not outs:
            return np.zeros(0, np.float64 if mode == "ref" else self.dtype)
        return np.concatenate(outs)

    def positions(self, k0, k1):
        """Input positions (float64 samples) of outputs k0..k1-1, for analytic checks."""
        return np.array([self.pos(k) / ONE for k in range(k0, k1)])
