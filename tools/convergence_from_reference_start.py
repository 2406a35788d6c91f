#!/usr/bin/env python3
"""BASELINE config 2 from the REFERENCE'S OWN START (theta_0 = 0, varepsilon = 1e-6 -- every chain within 1e-3 of the origin,
chain.py:25-27): after how many generations does the posterior-moment gate hold (pooled variance ratio within 1 %, every |mean| < 0.05 sigma,
over a trailing window of 1000 generations)?  Population sums per generation (running_moments), no resident history.
usage: convergence_from_reference_start.py [max_gens] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bipymc_amd import _lib as L                      # noqa: E402
from bipymc_amd.engine import HipEngine               # noqa: E402
from bipymc_amd.utils import d100_gauss               # noqa: E402


def run(max_gens=12000, seed=42, window=1000, step=250, N=8192, d=100, verbose=True):
    """-> (first generation T at which rows (T - window, T] pass the gate, list of (T, var_ratio, max|mean|/sigma, acc))"""
    g = d100_gauss.Gauss_100D(rho=0.5, dim=d)
    tid, tp, _ = g._bpm_target_spec()
    sig2 = np.arange(d) + 1.0
    e = HipEngine(algo=L.ALGO_DREAM, n_chains=N, dim=d, target_id=tid, target_params=tp, seed=seed, burnin_gen=200, n_cr_gen=50,
                  keep_history=False, running_moments=True)
    e.init_chains(np.zeros(d), 1e-6)
    e.begin_run()
    log, first = [], None
    T = 0
    while T < max_gens:
        e.step(step)
        T += step
        if T < window:
            continue
        cnt, s1, s2, sh = e.reduce_moments((1 + T - window) * N)
        mean = sh + s1 / cnt
        var = s2 / cnt - (s1 / cnt) ** 2
        st = e.stats()
        vr, mm = float(np.mean(var / sig2)), float(np.max(np.abs(mean) / np.sqrt(sig2)))
        acc = st["local_n_accepted"] / float(st["local_n_accepted"] + st["local_n_rejected"])
        log.append((T, vr, mm, acc))
        ok = abs(vr - 1.0) < 0.01 and mm < 0.05
        if verbose:
            print("generations %6d..%6d: var ratio %.4f  max|mean|/sigma %.4f  acc %.3f  %s" % (T - window + 1, T, vr, mm, acc, "GATE" if ok else ""), flush=True)
        if ok and first is None:
            first = T
        if first is not None and T >= first + 2 * window:
            break
    e.close()
    return first, log


if __name__ == "__main__":
    mg = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
    sd = int(sys.argv[2]) if len(sys.argv) > 2 else 42
    first, _ = run(mg, sd)
    print("first window that passes the gate ends at generation", first)
