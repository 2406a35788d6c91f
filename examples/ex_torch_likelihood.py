#!/usr/bin/env python3
"""A user-written likelihood evaluated ON THE GPU by PyTorch (`vectorized="device"`, round 5) -- the model of the reference's
examples/ex_para_fit.py:20-110 (a straight line with an underestimated-error term, emcee's line-fit tutorial) written as a torch function of a whole
block of parameter vectors.  The sampler hands the block over where it lies in device memory (`__cuda_array_interface__`: torch wraps it without a
copy) and takes the log-likelihoods back as a device tensor: nothing crosses PCIe.  Beside it the same model as a vectorised NumPy host callback and
row by row like the reference calls it (samplers.py:36-43).

`import torch` comes BEFORE the sampler is created: the torch wheel carries a HIP runtime of its own and a process initialises only one.
"""
from __future__ import division, print_function

import os
import sys
import time

import numpy as np
import torch                                            # noqa: F401  (first, see above)

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))   # run from a checkout

from bipymc_amd.dream import DreamMpi                   # noqa: E402

# ---- synthetic data of the tutorial: y = m x + b with under-estimated errors (f = fractional extra variance)
M_TRUE, B_TRUE, F_TRUE = -0.9594, 4.294, 0.534
rs = np.random.RandomState(123)
x = np.sort(10 * rs.rand(50))
yerr = 0.1 + 0.5 * rs.rand(50)
y = M_TRUE * x + B_TRUE
y += np.abs(F_TRUE * y) * rs.randn(50)
y += yerr * rs.randn(50)


def lnprob_row(theta):
    """one parameter vector (the reference's contract): flat prior box, Gaussian likelihood with variance yerr^2 + (model f)^2"""
    m, b, lnf = theta
    if not (-5.0 < m < 0.5 and 0.0 < b < 10.0 and -10.0 < lnf < 1.0):
        return -np.inf
    model = m * x + b
    inv_sigma2 = 1.0 / (yerr ** 2 + model ** 2 * np.exp(2 * lnf))
    return -0.5 * np.sum((y - model) ** 2 * inv_sigma2 - np.log(inv_sigma2))


def lnprob_block(thetas):
    """(n, 3) block on the host"""
    m, b, lnf = thetas[:, 0:1], thetas[:, 1:2], thetas[:, 2:3]
    model = m * x[None, :] + b
    inv_sigma2 = 1.0 / (yerr[None, :] ** 2 + model ** 2 * np.exp(2 * lnf))
    ll = -0.5 * np.sum((y[None, :] - model) ** 2 * inv_sigma2 - np.log(inv_sigma2), axis=1)
    ok = (-5.0 < m[:, 0]) & (m[:, 0] < 0.5) & (0.0 < b[:, 0]) & (b[:, 0] < 10.0) & (-10.0 < lnf[:, 0]) & (lnf[:, 0] < 1.0)
    return np.where(ok, ll, -np.inf)


xt = torch.tensor(x, dtype=torch.float64, device="cuda")
yt = torch.tensor(y, dtype=torch.float64, device="cuda")
et = torch.tensor(yerr, dtype=torch.float64, device="cuda")


def lnprob_device(rows):
    """the same on the GPU: `rows` exposes __cuda_array_interface__ (n proposals x 3 coordinates, where they lie); returns n float64 on the device"""
    th = torch.as_tensor(rows, device="cuda")
    m, b, lnf = th[:, 0:1], th[:, 1:2], th[:, 2:3]
    model = m * xt[None, :] + b
    inv_sigma2 = 1.0 / (et[None, :] ** 2 + model ** 2 * torch.exp(2 * lnf))
    ll = -0.5 * torch.sum((yt[None, :] - model) ** 2 * inv_sigma2 - torch.log(inv_sigma2), dim=1)
    ok = (-5.0 < m[:, 0]) & (m[:, 0] < 0.5) & (0.0 < b[:, 0]) & (b[:, 0] < 10.0) & (-10.0 < lnf[:, 0]) & (lnf[:, 0] < 1.0)
    return torch.where(ok, ll, torch.full_like(ll, float("-inf")))


if __name__ == "__main__":
    theta_0 = np.array([-1.0, 4.5, -0.7])
    n_chains, gens = 4096, 600
    for name, fn, kw in (("row by row (the reference's contract)", lnprob_row, {}),
                         ("vectorised NumPy on the host", lnprob_block, dict(vectorized=True)),
                         ("torch on the device (vectorized=\"device\")", lnprob_device, dict(vectorized="device"))):
        s = DreamMpi(fn, theta_0, varepsilon=1e-4, n_chains=n_chains, n_cr_gen=50, burnin_gen=200, seed=7, **kw)
        t0 = time.time()
        s.run_mcmc(n_chains * (gens + 1))
        el = time.time() - t0
        mean, std = s.param_est_moments(n_chains * 300)
        print("%-44s %6.2f s  %9.3g chain-updates/s   m = %.3f +- %.3f  b = %.3f +- %.3f  ln f = %.3f +- %.3f  acceptance %.3f"
              % (name, el, n_chains * gens / el, mean[0], std[0], mean[1], std[1], mean[2], std[2], s.acceptance_fraction))
