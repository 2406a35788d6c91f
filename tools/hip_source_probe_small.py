#!/usr/bin/env python3
"""The HIP-source likelihood path at small dimension against the shipped target of the same shape: DE-MC, banana (d = 2), 65536 chains, snooker 0.1 (cfg3)
and DREAM, 8-D pairwise mixture, 32768 chains (cfg5's share) -- chain-updates/s, steady state."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bipymc_amd import _lib as L
from bipymc_amd.engine import HipEngine
from bipymc_amd.utils import banana_rv

BANANA = """
__device__ double ln_like(const double* v, int d, const double* k) {      // utils/banana_rv.py through the library's parameter block
    const double a = k[7], b = k[8];
    const double x1 = v[0] / a;
    const double x2 = (v[1] - b * (x1 * x1 + a * a)) * a;
    const double u = (x1 - k[0]) * k[2];
    const double w = (x2 - k[1]) * k[3];
    return k[6] - 0.5 * (u * u - 2.0 * k[4] * u * w + w * w) * k[5];
}"""


def run(name, make, gens=400):
    e = make()
    e.reserve_history(3 * gens + 100)
    e.begin_run()
    e.step(50); e.synchronize()
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter(); e.step(gens); e.synchronize(); best = min(best, (time.perf_counter() - t0) / gens)
    N = e.n_chains
    print("%-44s %7.2f us per generation  %.3g chain-updates/s" % (name, best * 1e6, N / best), flush=True)
    e.close()


if __name__ == "__main__":
    t = banana_rv.Banana_2D()
    tid, tp, d = t._bpm_target_spec()
    N = 65536
    np.random.seed(1)
    x0 = np.stack(t.rvs(N), axis=1) if isinstance(t.rvs(4), tuple) else t.rvs(N)

    def shipped():
        e = HipEngine(algo=L.ALGO_DEMC, n_chains=N, dim=2, target_id=tid, target_params=tp, seed=3, p_snooker=0.1)
        e.set_state(x0)
        return e

    def user():
        e = HipEngine(algo=L.ALGO_DEMC, n_chains=N, dim=2, target_id=L.TARGET_HOST_CALLBACK, target_params=None, seed=3, p_snooker=0.1)
        e.set_state(x0)
        e.set_device_likelihood(BANANA, tp)
        print("   ", e.device_likelihood_info())
        return e
    run("cfg3 shipped banana", shipped)
    run("cfg3 banana as HIP source", user)
