#!/usr/bin/env python3
"""Diagnostic: time per generation DURING DREAM's CR adaptation (burn-in) -- update kernels with Welford moments and
CR statistics plus the per-generation CR reduction kernels -- beside the steady-state figure of tools/bench_configs.py."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bipymc_amd import _lib as L          # noqa: E402
from bipymc_amd.engine import HipEngine   # noqa: E402
from bipymc_amd.utils import d100_gauss, mixture_nd   # noqa: E402

for name, spec, N, G, outl in (("cfg2 gauss100", d100_gauss.Gauss_100D()._bpm_target_spec(), 8192, 300, 0),
                               ("cfg5/8 mixture8", mixture_nd.BimodeGauss_ND(8)._bpm_target_spec(), 32768, 300, 0),
                               ("cfg5 mixture8", mixture_nd.BimodeGauss_ND(8)._bpm_target_spec(), 262144, 100, 0),
                               ("cfg5 mixture8 + outlier check every 50", mixture_nd.BimodeGauss_ND(8)._bpm_target_spec(), 262144, 100, 50)):
    tid, tp, d = spec
    e = HipEngine(algo=L.ALGO_DREAM, n_chains=N, dim=d, target_id=tid, target_params=tp, seed=1, burnin_gen=10 ** 6, n_cr_gen=5,
                  outlier_every=outl)
    e.set_state(np.random.RandomState(0).normal(size=(N, d)) + 1.0)
    e.reserve_history(4 * G + 60)
    e.begin_run()
    e.step(20)
    e.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        e.step(G)
        e.synchronize()
        best = min(best, (time.perf_counter() - t0) / G)
    print("%-40s N=%-7d burn-in generation %.2f us  (%.3e updates/s)" % (name, N, best * 1e6, N / best))
    e.close()
