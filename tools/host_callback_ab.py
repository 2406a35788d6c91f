#!/usr/bin/env python3
"""A/B of the host-callback path at cfg2's shape (DREAM, 100-D Gaussian, N = 8192, vectorised NumPy ln_like): bpm_propose / bpm_commit with the
read-back in 1 / 2 / 4 / 8 pieces under the library's compaction copy (BPM_PROPOSE_PIECES), each in a child process, same box, 3 s each.
Prints chain-updates/s and the callback's share of the wall clock."""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    from bipymc_amd import _lib as L
    from bipymc_amd.engine import HipEngine
    N, d = 8192, 100
    sig = np.sqrt(np.arange(d) + 1.0)
    rho = 0.5
    c0 = -0.5 * (d * np.log(2 * np.pi) + 2 * np.sum(np.log(sig)) + (d - 1) * np.log(1 - rho) + np.log(1 + (d - 1) * rho))
    a, b = 1.0 / (1 - rho), rho / ((1 - rho) * (1 + (d - 1) * rho))
    isig = 1.0 / sig

    def ln_like(X):
        z = X * isig
        s1 = z.sum(axis=1)
        return c0 - 0.5 * (a * np.einsum("ij,ij->i", z, z) - b * s1 * s1)
    rs = np.random.RandomState(1234)
    X0 = sig * (np.sqrt(0.5) * rs.standard_normal((N, 1)) + np.sqrt(0.5) * rs.standard_normal((N, d)))
    e = HipEngine(algo=L.ALGO_DREAM, n_chains=N, dim=d, target_id=L.TARGET_HOST_CALLBACK, target_params=None, seed=42, burnin_gen=0)
    e.set_state(X0)
    e.set_loglike(ln_like(X0))
    e.reserve_history(8000)
    e.begin_run()
    for _ in range(20):
        for _h in range(2):
            p, _i = e.propose()
            e.commit(ln_like(p))
    gens, t_py, t_prop, t_com, t0 = 0, 0.0, 0.0, 0.0, time.perf_counter()
    while time.perf_counter() - t0 < 3.0:
        for _h in range(2):
            ta = time.perf_counter()
            p, _i = e.propose()
            tb = time.perf_counter()
            ll = ln_like(p)
            tc = time.perf_counter()
            e.commit(ll)
            td = time.perf_counter()
            t_prop += tb - ta; t_py += tc - tb; t_com += td - tc
        gens += 1
    el = time.perf_counter() - t0
    print(json.dumps(dict(pieces=os.environ.get("BPM_PROPOSE_PIECES"), updates_per_s=N * gens / el, us_per_half_generation=el / gens / 2 * 1e6,
                          propose_us=t_prop / gens / 2 * 1e6, callback_us=t_py / gens / 2 * 1e6, commit_us=t_com / gens / 2 * 1e6)))
    e.close()


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child()
    else:
        for rep in range(2):
            for pieces in ("1", "2", "4", "8"):
                env = dict(os.environ, BPM_PROPOSE_PIECES=pieces)
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
                print(r.stdout.decode().strip() or r.stderr.decode()[-400:], flush=True)
