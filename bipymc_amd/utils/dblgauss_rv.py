"""2-D bimodal Gaussian mixture (reference: bipymc/utils/dblgauss_rv.py:10-41)."""
import numpy as np

from .mixture_nd import BimodeGauss_ND


class BimodeGauss_2D(BimodeGauss_ND):
    def __init__(self, mu_g1=[0, 0], mu_g2=[2, 2], sigma_g1=[0.25, 0.25], sigma_g2=[0.25, 0.25],
                 rho_g1=0.8, rho_g2=-0.8, w_g1=0.25, w_g2=0.75):
        super(BimodeGauss_2D, self).__init__(2, mu_g1, mu_g2, sigma_g1, sigma_g2, rho_g1, rho_g2, w_g1, w_g2)

    def pdf(self, y1, y2):
        return np.exp(super(BimodeGauss_2D, self).ln_like(np.stack(np.broadcast_arrays(y1, y2), axis=-1)))

    def rvs(self, n_samples):
        s = super(BimodeGauss_2D, self).rvs(n_samples)
        return (s[:, 0], s[:, 1])
