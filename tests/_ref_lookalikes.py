"""Stand-ins for the REFERENCE's target objects (bipymc/utils/d100_gauss.py:10-35, dblgauss_rv.py:10-32,
banana_rv.py:10-40) for machines where /root/reference does not exist (the GPU box): same class names and the same
attributes the reference's constructors set, log-densities through scipy.stats like the reference (np.log of a pdf).
Written for the tests; they publish NO `_bpm_target_spec`, so only the look-alike rule of bipymc_amd/utils/_target.py
can put them on the device."""
import numpy as np
from scipy.stats import multivariate_normal as mvn


class Gauss_100D(object):
    __module__ = "bipymc.utils.d100_gauss"        # (the registry checks where the class claims to live: bipymc_amd/utils/_target.py)

    def __init__(self, rho=0.5, dim=100):
        self.dim, self.rho = dim, rho
        self.mu = np.zeros(dim)
        self.var = np.sqrt(1.0 + np.arange(dim))
        self.cov = rho * np.outer(self.var, self.var) + (1.0 - rho) * np.diag(self.var ** 2)
        self.rv_100d = mvn(self.mu, self.cov)

    def ln_like(self, y):
        assert len(y) == self.dim
        return np.log(self.rv_100d.pdf(y))


class BimodeGauss_2D(object):
    __module__ = "bipymc.utils.dblgauss_rv"

    def __init__(self, mu_g1=(0, 0), mu_g2=(2, 2), sigma_g1=(0.25, 0.25), sigma_g2=(0.25, 0.25), rho_g1=0.8, rho_g2=-0.8,
                 w_g1=0.25, w_g2=0.75):
        def cov(s, r):
            return np.array([[s[0] * s[0], r * s[0] * s[1]], [r * s[0] * s[1], s[1] * s[1]]])
        self.mu_g1, self.mu_g2 = list(mu_g1), list(mu_g2)
        self.cov_g1, self.cov_g2 = cov(sigma_g1, rho_g1), cov(sigma_g2, rho_g2)
        self.rv_2d_g1, self.rv_2d_g2 = mvn(self.mu_g1, self.cov_g1), mvn(self.mu_g2, self.cov_g2)
        self.w_g1, self.w_g2 = w_g1 / (w_g1 + w_g2), w_g2 / (w_g1 + w_g2)

    def ln_like(self, y):
        assert len(y) == 2
        return np.log(self.w_g1 * self.rv_2d_g1.pdf(y) + self.w_g2 * self.rv_2d_g2.pdf(y))


class Banana_2D(object):
    __module__ = "bipymc.utils.banana_rv"

    def __init__(self, mu1=0, mu2=0, sigma1=1, sigma2=1, rho=0.9, a=1.15, b=0.5):
        self.mu1, self.mu2, self.sigma1, self.sigma2, self.rho, self.a, self.b = mu1, mu2, sigma1, sigma2, rho, a, b
        c = rho * sigma1 * sigma2
        self.rv_2d_normal = mvn([mu1, mu2], [[sigma1 ** 2, c], [c, sigma2 ** 2]])

    def ln_like(self, y):
        assert len(y) == 2
        x1 = y[0] / self.a
        return np.log(self.rv_2d_normal.pdf([x1, (y[1] - self.b * (x1 ** 2 + self.a ** 2)) * self.a]))


class Gauss_100D_scaled(Gauss_100D):
    """same attributes, different density (an override the verification must catch)"""
    def ln_like(self, y):
        return 0.5 * Gauss_100D.ln_like(self, y)


class Gauss_100D_boxed(Gauss_100D):
    """keeps every attribute and the density near the mode, adds a prior box -- -inf beyond 5 sigma in any coordinate"""
    def ln_like(self, y):
        return -np.inf if np.any(np.abs(np.asarray(y)) > 5.0 * self.var) else Gauss_100D.ln_like(self, y)
