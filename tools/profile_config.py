#!/usr/bin/env python3
"""One BASELINE configuration for a fixed number of generations -- the program to put behind `rocprofv3 ... --`.

    python tools/profile_config.py cfg3|cfg5|cfg5_burnin|cfg5_local|cfg2|cfg2_burnin [gens]

cfg3         DE-MC, banana 2-D, N = 65536, snooker 0.1, steady state          (lanes per chain 1)
cfg5         DREAM, 8-D mixture, N = 262144, steady state (burn-in 0)         (lanes per chain 4)
cfg5_burnin  the same with CR adaptation on every generation and the outlier check every 50
cfg5_local   one GPU's share of cfg5: N = 32768, steady state
cfg2         DREAM, 100-D Gaussian, N = 8192, steady state                    (one wavefront per chain)
cfg2_burnin  the same during CR adaptation
Prints generations/s measured by the host clock (no profiler: the number of tools/bench_configs.py)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bipymc_amd import _lib as L                      # noqa: E402
from bipymc_amd.engine import HipEngine               # noqa: E402
from bipymc_amd.utils import banana_rv, d100_gauss, mixture_nd   # noqa: E402


def main():
    cfg = sys.argv[1]
    gens = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    rs = np.random.RandomState(0)
    np.random.seed(0)
    kw = {}
    if cfg == "cfg3":
        t = banana_rv.Banana_2D()
        algo, N, bpu = L.ALGO_DEMC, 65536, 97.6
        y1, y2 = t.rvs(N)
        x0 = np.stack([y1, y2], axis=1)
        kw = dict(p_snooker=0.1)
    elif cfg in ("cfg5", "cfg5_burnin", "cfg5_local"):
        t = mixture_nd.BimodeGauss_ND(8)
        algo, N, bpu = L.ALGO_DREAM, (32768 if cfg == "cfg5_local" else 262144), 592
        x0 = t.rvs(N)
        kw = dict(burnin_gen=10 ** 6, n_cr_gen=5, outlier_every=50) if cfg == "cfg5_burnin" else dict(burnin_gen=0)
        if cfg == "cfg5_burnin":
            bpu = 592 + 4 * 8 * 8
    else:
        t = d100_gauss.Gauss_100D()
        algo, N, bpu = L.ALGO_DREAM, 8192, 7216
        x0 = t.rvs(N)
        kw = dict(burnin_gen=10 ** 6, n_cr_gen=5) if cfg == "cfg2_burnin" else dict(burnin_gen=0)
        if cfg == "cfg2_burnin":
            bpu = 7216 + 32 * 100
    tid, tp, d = t._bpm_target_spec()
    e = HipEngine(algo=algo, n_chains=N, dim=d, target_id=tid, target_params=tp, seed=42, **kw)
    e.set_state(x0)
    e.reserve_history(gens + 80)
    e.begin_run()
    e.step(60)
    e.synchronize()
    t0 = time.perf_counter()
    e.step(gens)
    e.synchronize()
    el = time.perf_counter() - t0
    ups = N * gens / el
    print("%s N=%d d=%d: %.2f us/generation, %.3e chain-updates/s, %.0f GB/s algorithmic at %g B/update (%.3f of 8 TB/s)"
          % (cfg, N, d, el / gens * 1e6, ups, ups * bpu / 1e9, bpu, ups * bpu / 8e12))
    e.close()


if __name__ == "__main__":
    main()
