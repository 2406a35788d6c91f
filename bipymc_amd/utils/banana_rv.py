"""2-D twisted-Gaussian "banana" (reference: bipymc/utils/banana_rv.py:10-69)."""
import math

import numpy as np

from ._target import TARGET_BANANA_2D, pair_block


class Banana_2D(object):
    def __init__(self, mu1=0, mu2=0, sigma1=1, sigma2=1, rho=0.9, a=1.15, b=0.5):
        self.mu1, self.mu2 = mu1, mu2
        self.sigma1, self.sigma2 = sigma1, sigma2
        self.rho = rho
        self.a, self.b = a, b
        self._params = np.array(pair_block((mu1, mu2), (float(sigma1), float(sigma2)), rho) + [a, b], dtype=np.float64)

    def _bpm_target_spec(self):
        return TARGET_BANANA_2D, self._params, 2

    def inv_transform(self, y1, y2):
        x1_inv = y1 / self.a
        x2_inv = (y2 - self.b * (x1_inv ** 2.0 + self.a ** 2.0)) * self.a
        return x1_inv, x2_inv

    def transform(self, x1, x2):
        return self.a * x1, x2 / self.a + self.b * (x1 ** 2.0 + self.a ** 2.0)

    def _ln_pdf(self, y1, y2):
        mu1, mu2, is1, is2, rho, h, ln_norm, a, b = self._params
        x1 = y1 / a
        x2 = (y2 - b * (x1 * x1 + a * a)) * a
        u = (x1 - mu1) * is1
        v = (x2 - mu2) * is2
        return ln_norm - 0.5 * (u * u - 2.0 * rho * u * v + v * v) * h

    def pdf(self, y1, y2):
        return np.exp(self._ln_pdf(np.asarray(y1, dtype=np.float64), np.asarray(y2, dtype=np.float64)))

    def ln_like(self, y):
        y = np.asarray(y, dtype=np.float64)
        assert y.shape[-1] == 2
        return self._ln_pdf(y[..., 0], y[..., 1])

    def check_prob_lvl(self, y1, y2, pdf_lvl):
        return pdf_lvl < self.pdf(y1, y2)

    def rvs(self, n_samples):
        g1 = np.random.standard_normal(n_samples)
        g2 = np.random.standard_normal(n_samples)
        x1 = self.mu1 + self.sigma1 * g1
        x2 = self.mu2 + self.sigma2 * (self.rho * g1 + math.sqrt(1 - self.rho ** 2) * g2)
        return self.transform(x1, x2)
