#!/usr/bin/env python3
"""Throughput of the other BASELINE configurations on one GPU (not the headline bench):
cfg3 DE-MC banana N=65536 snooker 0.1; cfg5 (one GPU's share) DREAM 8-D mixture N=32768; cfg2 for reference.
Prints chain-updates/s and the fraction of the HBM roofline with SURVEY 8(d)'s bytes per update."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bipymc_amd import _lib as L                      # noqa: E402
from bipymc_amd.engine import HipEngine               # noqa: E402
from bipymc_amd.utils import banana_rv, d100_gauss, mixture_nd   # noqa: E402


def run(name, algo, N, spec, x0, bytes_per_update, gens=500, **kw):
    tid, tp, d = spec
    e = HipEngine(algo=algo, n_chains=N, dim=d, target_id=tid, target_params=tp, seed=42, **kw)
    e.set_state(x0)
    e.reserve_history(gens * 3 + 80)
    e.begin_run()
    e.step(60)
    e.synchronize()
    el = 1e9
    for _ in range(3):        # best of three windows (the first one of a process finds the GPU at idle clocks)
        t0 = time.perf_counter()
        e.step(gens)
        e.synchronize()
        el = min(el, time.perf_counter() - t0)
    st = e.stats()
    ups = N * gens / el
    print("%-28s N=%-7d d=%-4d %.3e updates/s  %.2f us/gen  %.1f GB/s algorithmic (%.3f of 8 TB/s)  acc=%.3f"
          % (name, N, d, ups, el / gens * 1e6, ups * bytes_per_update / 1e9, ups * bytes_per_update / 8e12,
             st["local_n_accepted"] / (st["local_n_accepted"] + st["local_n_rejected"])))
    e.close()


rs = np.random.RandomState(0)
run("cfg2 DREAM gauss100", L.ALGO_DREAM, 8192, d100_gauss.Gauss_100D()._bpm_target_spec(),
    rs.normal(size=(8192, 100)) * np.sqrt(np.arange(100) + 1.0), 7216, burnin_gen=0)
run("cfg3 DE-MC banana snooker", L.ALGO_DEMC, 65536, banana_rv.Banana_2D()._bpm_target_spec(),
    rs.normal(size=(65536, 2)) + np.array([0, 1.0]), 97.6, p_snooker=0.1)
run("cfg3 DE-MC banana", L.ALGO_DEMC, 65536, banana_rv.Banana_2D()._bpm_target_spec(),
    rs.normal(size=(65536, 2)) + np.array([0, 1.0]), 96.0)
run("cfg5/8 DREAM mixture8", L.ALGO_DREAM, 32768, mixture_nd.BimodeGauss_ND(8)._bpm_target_spec(),
    np.where(rs.uniform(size=(32768, 1)) < 0.25, 0.0, 2.0) + 0.25 * rs.normal(size=(32768, 8)), 592, burnin_gen=0)
run("cfg5 full DREAM mixture8", L.ALGO_DREAM, 262144, mixture_nd.BimodeGauss_ND(8)._bpm_target_spec(),
    np.where(rs.uniform(size=(262144, 1)) < 0.25, 0.0, 2.0) + 0.25 * rs.normal(size=(262144, 8)), 592, burnin_gen=0, gens=200)
run("DREAM gauss100 N=65536", L.ALGO_DREAM, 65536, d100_gauss.Gauss_100D()._bpm_target_spec(),
    rs.normal(size=(65536, 100)) * np.sqrt(np.arange(100) + 1.0), 7216, burnin_gen=0, gens=200)
