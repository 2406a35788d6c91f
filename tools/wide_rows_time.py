#!/usr/bin/env python3
"""DREAM on the equicorrelated Gaussian at growing row width: time per generation and fraction of the HBM roof on algorithmic bytes
(8 d (2 P + 3) + 16 per chain-update).  d <= 1024: register-resident kernels (2 ... 16 coordinates per lane); beyond:
the looped wide-row kernel (kernels_wide.h).  usage: wide_rows_time.py [d,d,...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bipymc_amd import _lib as L                      # noqa: E402
from bipymc_amd.engine import HipEngine               # noqa: E402
from bipymc_amd.utils import d100_gauss               # noqa: E402

CASES = {100: (8192, 500), 256: (8192, 300), 512: (8192, 200), 640: (8192, 200), 1000: (8192, 100), 1024: (8192, 100), 1500: (4096, 100),
         1800: (8192, 60), 2048: (4096, 100), 4096: (4096, 40), 10000: (4096, 20)}
want = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else sorted(CASES)
for d in want:
    N, G = CASES.get(d, (4096, 40))
    tid, tp, dd = d100_gauss.Gauss_100D(dim=d)._bpm_target_spec()
    e = HipEngine(algo=L.ALGO_DREAM, n_chains=N, dim=d, target_id=tid, target_params=tp, seed=1, burnin_gen=0, keep_history=True)
    rs = np.random.RandomState(0)
    X0 = np.sqrt(np.arange(d) + 1.0) * (np.sqrt(0.5) * rs.standard_normal((N, 1)) + np.sqrt(0.5) * rs.standard_normal((N, d)))
    e.set_state(X0)
    e.reserve_history(3 * G + 4)
    e.begin_run()
    e.step(G); e.synchronize()
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter(); e.step(G); e.synchronize(); best = min(best, (time.perf_counter() - t0) / G)
    B = 8 * d * 9 + 16
    st = e.stats()
    ls = e.launch_stats()
    print("d=%5d N=%5d: %8.1f us per generation, %.2e chain-updates/s, algorithmic %.2f TB/s (%.2f of 8), acc %.3f, %s" % (
        d, N, best * 1e6, N / best, N * B / best / 1e12, N * B / best / 8e12, st["local_n_accepted"] / (st["local_n_accepted"] + st["local_n_rejected"]),
        "own queue" if ls["direct"] > 0 and ls["stream"] == 0 else "HIP stream"), flush=True)
    e.close()
