#!/usr/bin/env python3
"""Where does a SHORT timed window go?  (the driver times 20 generations = 0.25 ms.)

After burn-in, time windows of K steady-state generations of BASELINE config 2, for several K, both ways bench.py
does: host clock around step + fence, and the HIP event pair on the sampler's stream.  A fit wall(K) = a + b K
separates the per-window overhead a from the per-generation cost b.

    python tools/window_anatomy.py [reps]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402  (HIP runtime shared with the library)

from bipymc_amd import _lib as L  # noqa: E402
from bipymc_amd.engine import HipEngine  # noqa: E402
from bipymc_amd.utils.d100_gauss import Gauss_100D  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    N, d = 8192, 100
    tid, tp, _ = Gauss_100D(rho=0.5, dim=d)._bpm_target_spec()
    eng = HipEngine(algo=L.ALGO_DREAM, n_chains=N, dim=d, target_id=tid, target_params=tp, seed=42,
                    del_pairs=3, burnin_gen=200, n_cr_gen=50, n_cr=3)
    rs = np.random.RandomState(1234)
    X0 = np.sqrt(np.arange(d) + 1.0) * (np.sqrt(0.5) * rs.standard_normal((N, 1)) + np.sqrt(0.5) * rs.standard_normal((N, d)))
    eng.set_state(X0)
    Ks = [1, 5, 20, 40, 80, 160, 320, 1000]
    eng.reserve_history(1 + 200 + 64 + 100 + 1200 + 100 + 480 + reps * sum(Ks) + 5 * reps * len(Ks) + 64)

    def fence():
        eng.synchronize()
        torch.cuda.synchronize()

    eng.begin_run()
    eng.step(200)
    fence()
    # what the bracket itself costs on an idle GPU
    for name, fn in (("eng.synchronize()", eng.synchronize), ("torch.cuda.synchronize()", torch.cuda.synchronize),
                     ("eng.step_timed(0)", lambda: eng.step_timed(0))):
        ts = []
        for _ in range(20):
            t0 = time.perf_counter()
            fn()
            ts.append((time.perf_counter() - t0) * 1e6)
        print("idle %-26s %.1f us (min %.1f)" % (name, np.median(ts), min(ts)))
    # host side alone: how long does the ENQUEUE of 400 generations take (bpm_step returns without waiting)?
    for rep in range(3):
        fence()
        t0 = time.perf_counter()
        eng.step(400)
        t1 = time.perf_counter()
        fence()
        t2 = time.perf_counter()
        print("enqueue of 800 launches: host %.2f us per launch; until drained %.2f us per launch" % ((t1 - t0) * 1e6 / 800, (t2 - t0) * 1e6 / 800))
    # the driver's exact sequence: burn-in, fence, 5 warm-up generations, fence, 20 timed -- the FIRST steady-state window
    for rep in range(4):
        eng.step(5)
        fence()
        t0 = time.perf_counter()
        ms, nl = eng.step_timed(20)
        fence()
        w = (time.perf_counter() - t0) * 1e6
        print("window %d after burn-in: wall %.1f us, event-timed launch period %.3f us (%d launches)" % (rep, w, ms * 1e3 / max(nl, 1), nl))
    # is a short window paced by the host or by the GPU?  enqueue time of 40 launches vs time until they are done
    for rep in range(4):
        eng.step(5)
        fence()
        t0 = time.perf_counter()
        eng.step(20)
        t1 = time.perf_counter()
        fence()
        t2 = time.perf_counter()
        print("window of 20 (plain bpm_step): enqueue returned after %.1f us, drained after %.1f us" % ((t1 - t0) * 1e6, (t2 - t0) * 1e6))
    # the same 20 generations enqueued BEHIND 100 others (queue never empty): launch period of the last 39 launches
    for rep in range(4):
        fence()
        eng.step(100)
        ms, nl = eng.step_timed(20)
        print("20 generations behind 100 (no idle gap): event-timed launch period %.3f us" % (ms * 1e3 / max(nl, 1)))
    print("K  wall_us(min/med)  event_us(min/med)  per-gen wall/event (med)")
    rows = []
    for K in Ks:
        walls, evs = [], []
        for _ in range(reps):
            eng.step(5)          # the driver's warm-up shape: a short call right before the timed one
            fence()
            t0 = time.perf_counter()
            ms, nl = eng.step_timed(K)
            fence()
            walls.append((time.perf_counter() - t0) * 1e6)
            evs.append(ms * 1e3 * (2.0 * K) / max(nl, 1))     # scaled from the 2K - 1 launch periods the events span to 2K
        w, e = np.array(walls), np.array(evs)
        rows.append((K, np.median(w), np.median(e)))
        print("%5d  %9.1f %9.1f   %9.1f %9.1f   %7.2f %7.2f" % (K, w.min(), np.median(w), e.min(), np.median(e),
                                                              np.median(w) / K, np.median(e) / K), flush=True)
    K = np.array([r[0] for r in rows], dtype=float)
    for name, col in (("wall", 1), ("event", 2)):
        y = np.array([r[col] for r in rows])
        A = np.vstack([np.ones_like(K), K]).T
        a, b = np.linalg.lstsq(A, y, rcond=None)[0]
        print("%s(K) ~ %.1f us + %.3f us * K" % (name, a, b))


if __name__ == "__main__":
    main()
