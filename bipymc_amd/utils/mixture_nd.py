"""Two-component Gaussian mixture in even dimension d whose components are block-diagonal
in coordinate pairs (2k, 2k+1); every pair carries the 2x2 blocks of the reference's
BimodeGauss_2D (bipymc/utils/dblgauss_rv.py:10-41).  d = 2 IS that target; d = 8 is the
"bimodal 8-D mixture" of BASELINE config 5 (absent from the reference)."""
import math

import numpy as np

from ._target import TARGET_MIXTURE_PAIRS, pair_block


class BimodeGauss_ND(object):
    def __init__(self, dim=8, mu_g1=(0, 0), mu_g2=(2, 2), sigma_g1=(0.25, 0.25), sigma_g2=(0.25, 0.25),
                 rho_g1=0.8, rho_g2=-0.8, w_g1=0.25, w_g2=0.75):
        assert dim % 2 == 0 and dim >= 2
        self.dim = dim
        self.mu_g1, self.mu_g2 = list(mu_g1), list(mu_g2)
        self.sigma_g1, self.sigma_g2 = list(sigma_g1), list(sigma_g2)
        self.rho_g1, self.rho_g2 = rho_g1, rho_g2
        self.w_g1 = w_g1 / (w_g1 + w_g2)
        self.w_g2 = w_g2 / (w_g1 + w_g2)
        self._params = np.array([math.log(self.w_g1), math.log(self.w_g2)]
                                + pair_block(self.mu_g1, self.sigma_g1, rho_g1)
                                + pair_block(self.mu_g2, self.sigma_g2, rho_g2), dtype=np.float64)

    def _bpm_target_spec(self):
        return TARGET_MIXTURE_PAIRS, self._params, self.dim

    def ln_like(self, y):
        y = np.asarray(y, dtype=np.float64)
        assert y.shape[-1] == self.dim
        p = self._params
        xe, xo = y[..., 0::2], y[..., 1::2]
        comp = []
        for c in range(2):
            mx, my, isx, isy, rho, h, ln_norm = p[2 + 7 * c: 9 + 7 * c]
            u = (xe - mx) * isx
            v = (xo - my) * isy
            q = np.sum((u * u - 2.0 * rho * u * v + v * v) * h, axis=-1)
            comp.append(p[c] + (self.dim // 2) * ln_norm - 0.5 * q)
        m = np.maximum(comp[0], comp[1])
        return m + np.log(np.exp(comp[0] - m) + np.exp(comp[1] - m))

    def rvs(self, n_samples, stratified=False):
        """exact draws; stratified=True: exactly round(w_g1 n) of them from the first component (the mode occupancy of a finite population then
        equals the weights instead of fluctuating by sqrt(w (1 - w) / n): what a 1 % gate on the OVERALL variance 0.0625 + 4 w (1 - w) needs)"""
        if stratified:
            pick1 = np.zeros(n_samples, dtype=bool)
            pick1[:int(round(self.w_g1 * n_samples))] = True
            np.random.shuffle(pick1)
        else:
            pick1 = np.random.uniform(size=n_samples) < self.w_g1
        out = np.empty((n_samples, self.dim))
        for k in range(self.dim // 2):
            for sel, mu, sg, rho in ((pick1, self.mu_g1, self.sigma_g1, self.rho_g1),
                                     (~pick1, self.mu_g2, self.sigma_g2, self.rho_g2)):
                n = int(np.count_nonzero(sel))
                g1 = np.random.standard_normal(n)
                g2 = np.random.standard_normal(n)
                out[sel, 2 * k] = mu[0] + sg[0] * g1
                out[sel, 2 * k + 1] = mu[1] + sg[1] * (rho * g1 + math.sqrt(1 - rho * rho) * g2)
        return out
