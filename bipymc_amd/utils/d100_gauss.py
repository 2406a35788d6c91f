"""Equicorrelated d-dimensional Gaussian (reference: bipymc/utils/d100_gauss.py:10-39).

cov_ii = sigma_i^2, cov_ij = rho sigma_i sigma_j with sigma_i = sqrt(i+1).  The
log-density has the O(d) closed form used on the device:
    z = y / sigma, S1 = sum z, S2 = sum z^2,
    ln p = c0 - 0.5 (a S2 - b S1^2),  a = 1/(1-rho),  b = rho / ((1+(d-1) rho)(1-rho)),
    c0 = -0.5 (d ln 2pi + 2 sum ln sigma + (d-1) ln(1-rho) + ln(1+(d-1) rho)).
Unlike the reference's log(pdf) it does not underflow to -inf far from the mode.
"""
import math

import numpy as np

from ._target import LN_2PI, TARGET_GAUSS_EQUICORR


def equicorr_block(rho, sigma):
    """Device parameter block [rho, c0, a, b, 1/sigma_0 .. 1/sigma_{d-1}] of a zero-mean Gaussian with
    cov_ii = sigma_i^2, cov_ij = rho sigma_i sigma_j (BPM_TARGET_GAUSS_EQUICORR, include/bipymc_hip.h)."""
    sg = np.asarray(sigma, dtype=np.float64)
    dim = sg.size
    logdet = 2.0 * np.sum(np.log(sg)) + (dim - 1) * math.log(1.0 - rho) + math.log(1.0 + (dim - 1) * rho)
    c0 = -0.5 * (dim * LN_2PI + logdet)
    a = 1.0 / (1.0 - rho)
    b = rho / ((1.0 + (dim - 1) * rho) * (1.0 - rho))
    return np.concatenate([[rho, c0, a, b], 1.0 / sg])


def equicorr_ln_like(block, y):
    """the closed form the device evaluates, on the host (rows of y)"""
    y = np.asarray(y, dtype=np.float64)
    z = y * block[4:]
    s1 = np.sum(z, axis=-1)
    s2 = np.sum(z * z, axis=-1)
    return block[1] - 0.5 * (block[2] * s2 - block[3] * s1 * s1)


class Gauss_100D(object):
    def __init__(self, rho=0.5, dim=100):
        self.mu = np.zeros(dim)
        self.var = np.sqrt(np.arange(dim) + 1.0)     # (sic) standard deviations, as in d100_gauss.py:17
        self.dim = dim
        self.rho = rho
        sg = self.var
        self.cov = rho * np.outer(sg, sg)
        self.cov[np.diag_indices(dim)] = sg ** 2.0
        blk = equicorr_block(rho, sg)
        self._c0, self._a, self._b = float(blk[1]), float(blk[2]), float(blk[3])
        self._inv_sigma = 1.0 / sg

    def _bpm_target_spec(self):
        return (TARGET_GAUSS_EQUICORR,
                np.concatenate([[self.rho, self._c0, self._a, self._b], self._inv_sigma]), self.dim)

    def ln_like(self, y):
        y = np.asarray(y, dtype=np.float64)
        assert y.shape[-1] == self.dim
        z = y * self._inv_sigma
        s1 = np.sum(z, axis=-1)
        s2 = np.sum(z * z, axis=-1)
        return self._c0 - 0.5 * (self._a * s2 - self._b * s1 * s1)

    def pdf(self, y):
        return np.exp(self.ln_like(y))

    def rvs(self, n_samples):
        # x_i = sigma_i (sqrt(rho) g + sqrt(1-rho) e_i): exact equicorrelated draw
        g = np.random.standard_normal((n_samples, 1))
        e = np.random.standard_normal((n_samples, self.dim))
        return self.var * (math.sqrt(self.rho) * g + math.sqrt(1.0 - self.rho) * e)
