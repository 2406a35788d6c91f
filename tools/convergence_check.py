#!/usr/bin/env python3
"""Invariance / convergence of DREAM on the 100-D Gaussian at N=8192 (BASELINE config 2):
(1) start from exact draws of the target: the moments must stay (stationarity);
(2) over-dispersed independent start: how long the population needs to find the correlated scale."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bipymc_amd import _lib as L                      # noqa: E402
from bipymc_amd.engine import HipEngine               # noqa: E402
from bipymc_amd.utils import d100_gauss               # noqa: E402

g = d100_gauss.Gauss_100D()
tid, tp, d = g._bpm_target_spec()
N = 8192
sig2 = np.arange(d) + 1.0


def report(tag, e, n_burn_gens):
    cnt, s1, s2, sh = e.reduce_moments((1 + n_burn_gens) * N)
    mean = sh + s1 / cnt
    var = s2 / cnt - (s1 / cnt) ** 2
    st = e.stats()
    print("%-40s rows=%d  max|mean|/sigma=%.4f  var ratio mean=%.4f min=%.4f max=%.4f  acc=%.3f p_cr=%s"
          % (tag, cnt, np.max(np.abs(mean) / np.sqrt(sig2)), np.mean(var / sig2), np.min(var / sig2), np.max(var / sig2),
             st["local_n_accepted"] / (st["local_n_accepted"] + st["local_n_rejected"]), np.round(st["p_cr"], 3)))


np.random.seed(1)
e = HipEngine(algo=L.ALGO_DREAM, n_chains=N, dim=d, target_id=tid, target_params=tp, seed=42, burnin_gen=200, n_cr_gen=50)
e.set_state(g.rvs(N))
e.begin_run(); e.step(1200)
report("exact start, gens 201..1200", e, 200)
report("exact start, gens 701..1200", e, 700)
e.close()

e = HipEngine(algo=L.ALGO_DREAM, n_chains=N, dim=d, target_id=tid, target_params=tp, seed=42, burnin_gen=200, n_cr_gen=50)
e.set_state(np.random.RandomState(1234).normal(size=(N, d)) * np.sqrt(sig2))
e.begin_run(); e.step(6000)
for b in (250, 1000, 2000, 4000, 5000):
    report("independent start, gens %d..6000" % (b + 1), e, b)
e.close()
