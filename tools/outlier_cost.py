"""(BPM_LIB_PATH selects another build of the library.)  Where cfg5's burn-in with the outlier check spends its time: generations with the check never due, due every 50, every 10; one GPU."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bipymc_amd import _lib as L
from bipymc_amd.engine import HipEngine
from bipymc_amd.utils import mixture_nd


def main():
    m = mixture_nd.BimodeGauss_ND(8)
    tid, tp, d = m._bpm_target_spec()
    N = int(os.environ.get("N", "262144"))
    np.random.seed(5)
    x0 = m.rvs(N)
    only = os.environ.get("ONLY")
    for every in ((int(only),) if only else (0, 10 ** 5, 50, 10)):
        e = HipEngine(algo=L.ALGO_DREAM, n_chains=N, dim=d, target_id=tid, target_params=tp, seed=42, burnin_gen=10 ** 6, n_cr_gen=5, outlier_every=every)
        e.set_state(x0)
        e.reserve_history(400)
        e.begin_run()
        e.step(30)
        e.synchronize()
        out = []
        for rep in range(3):
            t0 = time.perf_counter()
            e.step(100)
            e.synchronize()
            out.append((time.perf_counter() - t0) / 100 * 1e6)
        print("outlier_every=%-7d us per generation: %s   resets %d" % (every, " ".join("%.1f" % v for v in out), e.stats()["n_outlier_resets"]), flush=True)
        e.close()


if __name__ == "__main__":
    main()
