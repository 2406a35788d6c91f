#!/usr/bin/env python3
"""param_est on the device (bpm_reduce_moments, demc.py:235-248 without moving the history): time and bandwidth of the
reduction over a cfg2-sized history (8192 chains x 100 dims x G generations)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bipymc_amd import _lib as L                      # noqa: E402
from bipymc_amd.engine import HipEngine               # noqa: E402
from bipymc_amd.utils import d100_gauss               # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 1031
t = d100_gauss.Gauss_100D()
tid, tp, d = t._bpm_target_spec()
N = 8192
np.random.seed(0)
e = HipEngine(algo=L.ALGO_DREAM, n_chains=N, dim=d, target_id=tid, target_params=tp, seed=1, burnin_gen=0)
e.set_state(t.rvs(N))
e.reserve_history(G + 2)
e.begin_run()
e.step(G)
e.synchronize()
best = 1e9
for rep in range(5):
    t0 = time.perf_counter()
    cnt, s1, s2, sh = e.reduce_moments(0)
    best = min(best, time.perf_counter() - t0)
nbytes = cnt * d * 8
print("reduce_moments over %d rows x %d dims (%.2f GB): %.1f us host-to-host, %.2f TB/s (%.3f of 8 TB/s)"
      % (cnt, d, nbytes / 1e9, best * 1e6, nbytes / best / 1e12, nbytes / best / 8e12))
H = e.get_history(0, 40).reshape(-1, d)
c2, a1, a2, sh2 = None, None, None, None
mean = sh + s1 / cnt
print("mean[:3] =", mean[:3], " var[:3] =", (s2 / cnt - (s1 / cnt) ** 2)[:3])
