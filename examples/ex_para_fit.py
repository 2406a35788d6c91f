#!/usr/bin/env python3
"""Straight-line fit with an arbitrary Python ln_like_fn -- the scenario of the reference's
examples/ex_para_fit.py:20-110 (emcee's line-fit model), run through the MI355X drop-in classes.

The only change against the reference script is the import.  `lnprob` is an ordinary Python callable with
`ln_kwargs` and a prior that returns -inf outside its support, so the sampler takes the host-callback path
(proposals come from the GPU, ln_like is evaluated here, the Metropolis step runs on the GPU).
"""
from __future__ import division, print_function

import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))   # run from a checkout

from bipymc_amd.demc import DeMcMpi      # reference: from bipymc.demc import DeMcMpi
from bipymc_amd.dream import DreamMpi    # reference: from bipymc.dream import DreamMpi


def make_data(seed=42):
    rs = np.random.RandomState(seed)
    m_true, b_true, f_true = -0.9594, 4.294, 0.534
    N = 50
    x = np.sort(10 * rs.rand(N))
    yerr = 0.1 + 0.5 * rs.rand(N)
    y = m_true * x + b_true
    y += np.abs(f_true * y) * rs.randn(N)
    y += yerr * rs.randn(N)
    return x, y, yerr, (m_true, b_true, np.log(f_true))


def lnlike(theta, x, y, yerr):
    m, b, lnf = theta
    model = m * x + b
    inv_sigma2 = 1.0 / (yerr ** 2 + model ** 2 * np.exp(2 * lnf))
    return -0.5 * (np.sum((y - model) ** 2 * inv_sigma2 - np.log(inv_sigma2)))


def lnprior(theta):
    m, b, lnf = theta
    if -5.0 < m < 0.5 and 0.0 < b < 10.0 and -10.0 < lnf < 1.0:
        return 0.0
    return -np.inf


def lnprob(theta, x, y, yerr):
    lp = lnprior(theta)
    if not np.isfinite(lp):
        return -np.inf
    return lp + lnlike(theta, x, y, yerr)


def run(sampler_cls=DreamMpi, n_chains=12, n=500 * 100, comm=None, seed=7):
    x, y, yerr, truth = make_data()
    theta_0 = np.array([-0.8, 4.5, 0.2])
    my_mcmc = sampler_cls(lnprob, theta_0, n_chains=n_chains, mpi_comm=comm,
                          ln_kwargs={'x': x, 'y': y, 'yerr': yerr}, inflate=1e1, seed=seed)
    my_mcmc.run_mcmc(n)
    theta_est, sig_est, chain = my_mcmc.param_est(n_burn=10000)
    return my_mcmc, theta_est, sig_est, truth


if __name__ == "__main__":
    for cls in (DreamMpi, DeMcMpi):
        s, est, sig, truth = run(cls)
        print("=== %s ===" % cls.__name__)
        print("Esimated params: %s" % str(est))
        print("Estimated params sigma: %s " % str(sig))
        print("Truth: %s" % str(truth))
        print("Acceptance fraction: %f" % s.acceptance_fraction)
