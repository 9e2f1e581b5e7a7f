"""CPU baseline of libsoxr's ALGORITHM CLASS: FFT overlap-save rate conversion (numpy / scipy.fft).  TEST INFRASTRUCTURE
ONLY — imported by bench.py's cpu_baseline leg and tests/, never by the product.

libsoxr does not evaluate its VHQ filter as a 296-tap direct form (what oracle/soxr_oracle.c restates and what the
"port" baseline times): its sharp stages are FFT overlap-save (UPSTREAM-UNVERIFIED — there is no
libsoxr source in the image).  This module applies the oracle's OWN prototype (oracle/design.py's bank) that way, with
scipy's pocketfft, so that the CPU column of bench.py also holds a number of that cost class:

    y[k] = sum_n x[n] g(k M - n L)      g = the prototype at rate L f_in, centred on 0 (bank[p][j] = g[L (T/2-1-j) + p])
    block of N_in = M k' inputs --rfft--> X;  Y[q] = X[q] G[q] / M  (q <= N_out / 2;  G = DFT of g on the L N_in grid);
    y = irfft(Y, N_out = L k');  blocks start on period boundaries and overlap by more than the filter; only outputs whose
    filter support lies inside the block are kept.  What is neglected is the aliasing of g's stop band (VHQ: -175 dB).

`resample(pl, x)` agrees with the oracle's float64 direct form to ~1e-10 relative RMS (tests/test_oracle_pinning.py)."""
import numpy as np
import scipy.fft as sfft


class Plan:
    """Spectrum of the prototype for blocks of `periods` input periods (N_in = M periods, N_out = L periods)."""

    def __init__(self, pl, periods=256):
        L, M, T = int(pl.L), int(pl.M), int(pl.T)
        assert pl.phases == 0, "exact-ratio plans only"
        self.L, self.M, self.T, self.k = L, M, T, int(periods)
        self.N_in, self.N_out = M * self.k, L * self.k
        # g on the circular grid of L * N_in points: index m = L (T/2-1-j) + p, negative m wraps
        g = np.zeros(L * self.N_in)
        j = np.arange(T)
        for p in range(L):
            g[(L * (T // 2 - 1 - j) + p) % (L * self.N_in)] = pl.bank[p]
        G = sfft.rfft(g)
        nb = min(self.N_in, self.N_out) // 2 + 1
        self.H = np.zeros(self.N_out // 2 + 1, np.complex128)
        self.H[:nb] = G[:nb] / M
        if self.N_out <= self.N_in:
            self.H[-1] = self.H[-1].real            # the output grid's Nyquist bin is real
        disc = -(-((T // 2 + 2) * L) // M)          # outputs without full support at either end of a block
        self.lead = -(-disc // L)                   # ... in whole periods
        self.hop = self.k - 2 * self.lead
        assert self.hop >= 1, "block too short for this filter"

    def out_len(self, n):
        return (n * self.L + self.M // 2) // self.M if self.M % 2 == 0 else (2 * n * self.L + self.M) // (2 * self.M)


def resample(fp, x, workers=1, out_len=None):
    """x: 1-D float array -> float64 result of `out_len` (default: floor(n L / M + 1/2)) frames."""
    x = np.asarray(x, np.float64)
    n = len(x)
    L, M = fp.L, fp.M
    if out_len is None:
        out_len = (2 * n * L + M) // (2 * M)
    hop_in, hop_out = fp.hop * M, fp.hop * L
    n_blocks = max(1, -(-out_len // hop_out))
    lead_in = fp.lead * M
    total = lead_in + (n_blocks - 1) * hop_in + fp.N_in
    xp = np.zeros(max(total, lead_in + n))
    xp[lead_in:lead_in + n] = x
    blocks = np.lib.stride_tricks.as_strided(xp, shape=(n_blocks, fp.N_in), strides=(hop_in * 8, 8), writeable=False)
    X = sfft.rfft(blocks, axis=1, workers=workers)
    nb = fp.N_out // 2 + 1
    if nb <= X.shape[1]:
        Y = X[:, :nb] * fp.H
    else:                                          # up-sampling: zero-extend the spectrum
        Y = np.zeros((n_blocks, nb), np.complex128)
        Y[:, :X.shape[1]] = X * fp.H[:X.shape[1]]
    # scale: rfft is unnormalised (a sum over N_in samples), Y = X G / M, irfft divides by N_out: a constant input c gives
    # X[0] = c N_in, G[0] = L, Y[0] = c N_in L / M = c N_out, y = c
    y = sfft.irfft(Y, n=fp.N_out, axis=1, workers=workers)
    v0 = fp.lead * L
    return y[:, v0:v0 + hop_out].reshape(-1)[:out_len]
