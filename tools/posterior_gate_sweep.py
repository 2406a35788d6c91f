#!/usr/bin/env python3
"""How many post-burn-in generations does the per-coordinate posterior gate need?  

cfg2 (DREAM, 100-D Gaussian, N = 8192), cfg3 (DE-MC banana N = 65536, snooker 0.1) and cfg5's share / cfg5 (DREAM mixture d = 8) with
running_moments and no history: the gate of bench.py (batch_moment_gate: every coordinate's variance within 1 %, every mean within 0.01 sigma,
batch-means standard errors) over a growing number of generations, from exact draws of the target and, for cfg2, from the reference's start
(theta_0 = 0, varepsilon = 1e-6; chain.py:25-27) after a transient.  Prints one JSON object per line.

    python tools/posterior_gate_sweep.py [cfg2|cfg2ref|cfg3|cfg5s|cfg5]...
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                            # noqa: E402
from bipymc_amd import _lib as L                        # noqa: E402
from bipymc_amd.engine import HipEngine                 # noqa: E402
from bipymc_amd.utils import banana_rv, d100_gauss, mixture_nd   # noqa: E402


def run(tag, algo, tgt, N, x0, burn, transient, totals, true_mean, true_var, **kw):
    tid, tp, d = tgt._bpm_target_spec()
    e = HipEngine(algo=algo, n_chains=N, dim=d, target_id=tid, target_params=tp, seed=42, keep_history=False, running_moments=True, **kw)
    if x0 is None:
        e.init_chains(np.zeros(d), 1e-6)
    else:
        e.set_state(x0)
    e.begin_run()
    done = 0
    for T in totals:
        t0 = time.perf_counter()
        e.step(burn + transient + T - done)
        e.synchronize()
        el = time.perf_counter() - t0
        done = burn + transient + T
        t1 = time.perf_counter()
        g = bench.batch_moment_gate(e, N, 1 + burn + transient, 1 + done, true_mean, true_var)
        g.update(config=tag, seconds_stepping=round(el, 3), seconds_gate=round(time.perf_counter() - t1, 3))
        st = e.stats()
        g["acceptance_fraction"] = st["local_n_accepted"] / float(st["local_n_accepted"] + st["local_n_rejected"])
        g.pop("gate", None)
        print(json.dumps(g), flush=True)
    e.close()


def main():
    which = sys.argv[1:] or ["cfg2", "cfg2ref", "cfg3", "cfg5s"]
    np.random.seed(20261005)
    totals = [2000, 5000, 10000, 20000, 40000, 80000]
    if "cfg2" in which or "cfg2ref" in which:
        g = d100_gauss.Gauss_100D(rho=0.5, dim=100)
        var = np.arange(100) + 1.0
        if "cfg2" in which:
            run("cfg2 exact start", L.ALGO_DREAM, g, 8192, g.rvs(8192), 200, 0, totals, np.zeros(100), var, burnin_gen=200, n_cr_gen=50)
        if "cfg2ref" in which:
            run("cfg2 reference start, 3000-generation transient", L.ALGO_DREAM, g, 8192, None, 200, 2800, totals, np.zeros(100), var, burnin_gen=200, n_cr_gen=50)
    if "cfg3" in which:
        b = banana_rv.Banana_2D()
        y1, y2 = b.rvs(65536)
        a_, b_ = 1.15, 0.5
        run("cfg3 exact start", L.ALGO_DEMC, b, 65536, np.stack([y1, y2], axis=1), 0, 0, totals, [0.0, b_ * (1 + a_ * a_)],
            [a_ * a_, 1.0 / (a_ * a_) + 2 * b_ * b_], p_snooker=0.1)
    for tag, N in (("cfg5s", 32768), ("cfg5", 262144)):
        if tag in which:
            m = mixture_nd.BimodeGauss_ND(8)
            run("%s exact start (overall moments: mean 1.5, var 0.8125 per axis)" % tag, L.ALGO_DREAM, m, N, m.rvs(N), 300, 0, totals[:5], np.full(8, 1.5), np.full(8, 0.8125),
                burnin_gen=300, n_cr_gen=50)


if __name__ == "__main__":
    main()
