#!/usr/bin/env python3
"""The reference's heavy test (tests/test_100dgauss.py: DREAM, 100 chains, 500000 samples) with the
target evaluated on the GPU, and the BASELINE configuration (8192 chains) next to it."""
from __future__ import division, print_function

import time

import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))   # run from a checkout

from bipymc_amd.dream import DreamMpi
from bipymc_amd.utils import d100_gauss

if __name__ == "__main__":
    gauss = d100_gauss.Gauss_100D()
    for n_chains, n_samples, n_burn in ((100, 500000, 200000), (8192, 8192 * 2001, 8192 * 1000)):
        np.random.seed(42)
        my_mcmc = DreamMpi(gauss.ln_like, np.zeros(100), n_chains=n_chains, n_cr_gen=50, burnin_gen=int(n_burn / n_chains))
        t0 = time.time()
        my_mcmc.run_mcmc(n_samples)
        t1 = time.time()
        cnt, s1, s2, sh = my_mcmc._engine.reduce_moments(n_burn)
        mean = sh + s1 / cnt
        var = s2 / cnt - (s1 / cnt) ** 2
        print("n_chains=%d: %.2f s, %.3g chain-updates/s" % (n_chains, t1 - t0, (n_samples - n_chains) / (t1 - t0)))
        print("  max |mean| = %.3f, var/true var in [%.3f, %.3f], acceptance %.3f, p_cr %s"
              % (np.abs(mean).max(), (var / (np.arange(100) + 1.0)).min(), (var / (np.arange(100) + 1.0)).max(),
                 my_mcmc.acceptance_fraction, np.round(my_mcmc.p_cr, 3)))
