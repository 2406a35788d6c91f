#!/usr/bin/env python3
"""PCIe-inclusive rate of BASELINE config 2 through the host-buffer boundary.

`bench.py`'s value starts with the state resident in HBM and excludes the final D2H of the history
(SURVEY section 8d).  A bipymc user calls `run_mcmc(n)` then `param_est(n_burn)`: the initial state comes from
a host buffer and the result is either (i) the full chain history copied back (`bpm_get_history`, what the
reference's `param_est` returns as `chain_slice`), or (ii) only mean/std reduced on the device
(`bpm_reduce_moments`).  This times both, whole calls, host clock."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bipymc_amd import _lib as L          # noqa: E402
from bipymc_amd.engine import HipEngine   # noqa: E402
from bipymc_amd.utils import d100_gauss   # noqa: E402

N, G = 8192, int(os.environ.get("GENS", "500"))
tid, tp, d = d100_gauss.Gauss_100D()._bpm_target_spec()
x0 = np.random.RandomState(0).normal(size=(N, d)) * np.sqrt(np.arange(d) + 1.0)
e = HipEngine(algo=L.ALGO_DREAM, n_chains=N, dim=d, target_id=tid, target_params=tp, seed=42, burnin_gen=0)
e.reserve_history(2 * G + 2)
e.set_state(x0)
e.begin_run()
e.step(20)
e.synchronize()                                           # warm

for what in ("history", "moments"):
    t0 = time.perf_counter()
    e.set_state(x0)                                       # H2D of the (N, d) state; restarts the history
    e.begin_run()
    e.step(G)
    e.synchronize()
    t1 = time.perf_counter()
    if what == "history":
        H = e.get_history()                               # (G+1, N, d) f64 to pageable host memory
        nbytes = H.nbytes
    else:
        cnt, s1, s2, sh = e.reduce_moments(0)
        nbytes = 3 * d * 8
    t2 = time.perf_counter()
    H = None                                              # (unmapping 3.3 GB is not part of the rate)
    print("%-8s: step %.4f s (%.3e updates/s resident), result D2H %.4f s (%.2f GB, %.1f GB/s) -> %.3e updates/s PCIe-inclusive"
          % (what, t1 - t0, N * G / (t1 - t0), t2 - t1, nbytes / 1e9, nbytes / 1e9 / max(t2 - t1, 1e-9), N * G / (t2 - t0)))
e.close()
