#!/usr/bin/env python3
"""Stress of what a process that creates and destroys many samplers does to their memory (diagnostic + tests/test_gpu_api.py).

Warm-up: samplers of four shapes, each run through the library's own queue and through HIP-stream launches, histories read back,
destroyed.  Then four repetitions (own queue, HIP stream, own queue, HIP stream) of a short run whose history grows through several
buffers and whose row 1 is written twice (bpm_set_state in the middle of the run).  The four histories must be equal.

With the state in the GPU's hardware-coherent memory type (round 2's experiment; since round 3 only in the experiment build
`make -C bipymc_amd/csrc variant NAME=coherent DEFS=-DBPM_EXPERIMENT_COHERENT`, run with BPM_LIB_PATH=build_variants/libbipymc_coherent.so
BPM_COHERENT_STATE=1) they are not: 4-5 runs of 6 show the first repetition with ~20 % of row 1 still holding its FIRST version, now and then
a repetition with every row different -- with acquire-only and with acquire + release packets alike.  Exit status 1 then."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bipymc_amd import _lib as L                                  # noqa: E402
from bipymc_amd.engine import HipEngine                           # noqa: E402
from bipymc_amd.utils import banana_rv, d100_gauss, mixture_nd    # noqa: E402

tid, tp, d = d100_gauss.Gauss_100D()._bpm_target_spec()
for spec, algo, N, kw, G in ((d100_gauss.Gauss_100D()._bpm_target_spec(), L.ALGO_DREAM, 512, dict(burnin_gen=20, n_cr_gen=4), 150),
                             (mixture_nd.BimodeGauss_ND(8)._bpm_target_spec(), L.ALGO_DREAM, 20000, dict(burnin_gen=60, n_cr_gen=4, del_pairs=2, outlier_every=20), 140),
                             (banana_rv.Banana_2D()._bpm_target_spec(), L.ALGO_DEMC, 4099, dict(p_snooker=0.2), 200),
                             (d100_gauss.Gauss_100D()._bpm_target_spec(), L.ALGO_DREAM, 8192, dict(burnin_gen=30, n_cr_gen=4), 100)):
    t2, p2, d2 = spec
    X = np.random.RandomState(3).normal(size=(N, d2)) + 1.0
    for direct in (True, False, True, True):
        e = HipEngine(algo=algo, n_chains=N, dim=d2, target_id=t2, target_params=p2, seed=11, **kw)
        e.set_launch_path(direct)
        e.set_state(X); e.begin_run(); e.step(G // 2); e.step_timed(G - G // 2 - 3); e.step(3)
        e.stats(); e.get_state(); e.get_history(0, G + 1); e.close()

N = 1024
X0 = np.random.RandomState(9).normal(size=(N, d)) + 0.5


def run(direct):
    e = HipEngine(algo=L.ALGO_DREAM, n_chains=N, dim=d, target_id=tid, target_params=tp, seed=21, burnin_gen=7, n_cr_gen=3)
    e.set_launch_path(direct)
    e.set_state(X0)
    e.begin_run()
    v1 = None
    for n in (0, 1, 2, 5, 1, 70):
        e.step(n)
        e.get_state()
        if n == 1 and v1 is None:
            v1 = e.get_history(1, 2)[0].copy()
        if n == 5:
            e.set_state(e.get_state() * 1.0)
    H = e.get_history(0, e.history_rows())
    mode = e.launch_stats()
    e.close()
    return H, v1, mode


res = [run(True), run(False), run(True), run(False)]
print("state in hardware-coherent memory: %s, packet fences: %s" % (res[0][2]["coherent_state"], res[0][2]["fence"]))
ref = res[1][0]
bad = False
for i, (H, v1, _) in enumerate(res):
    rows = np.nonzero(np.any(H != ref, axis=(1, 2)))[0]
    if rows.size:
        bad = True
        wrong = H[rows[0]] != ref[rows[0]]
        print("repetition %d (%s): %d history rows differ from repetition 1, first %s; in row %d %d elements are wrong, %d of them hold the row's first version"
              % (i, "own queue" if i % 2 == 0 else "HIP stream", rows.size, rows[:6], rows[0], int(wrong.sum()),
                 int((H[rows[0]][wrong] == v1[wrong]).sum()) if rows[0] == 1 else -1))
print("all four histories equal" if not bad else "HISTORIES DIFFER")
sys.exit(1 if bad else 0)
