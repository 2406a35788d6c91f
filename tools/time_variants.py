#!/usr/bin/env python3
"""Diagnostic: generation time of BASELINE config 2 for experiment builds of the library
(`make -C bipymc_amd/csrc variant NAME=x DEFS=...` -> build_variants/libbipymc_x.so).
usage: time_variants.py [name ...]   ('main' = the shipped library); one subprocess per library."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
import numpy as np
sys.path.insert(0, %r)
from bipymc_amd import _lib as L
if sys.argv[1] != "main":
    L.LIB_PATH = %r + "/build_variants/libbipymc_" + sys.argv[1] + ".so"
from bipymc_amd.engine import HipEngine
from bipymc_amd.utils import d100_gauss
tid, tp, d = d100_gauss.Gauss_100D()._bpm_target_spec()
out = []
for N, G in [tuple(int(v) for v in s.split(':')) for s in os.environ.get('SIZES', '8192:1000,65536:200').split(',')]:
    e = HipEngine(algo=L.ALGO_DREAM, n_chains=N, dim=d, target_id=tid, target_params=tp, seed=1, burnin_gen=0, keep_history=not os.environ.get('NO_HISTORY'))
    e.set_state(np.random.RandomState(0).normal(size=(N, d)) * np.sqrt(np.arange(d) + 1.0))
    e.reserve_history(3 * G + 60)
    e.begin_run()
    e.step(50); e.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); e.step(G); e.synchronize(); best = min(best, (time.perf_counter() - t0) / G)
    out.append("N=%%d %%.2f us/gen" %% (N, best * 1e6))
    e.close()
print("%%-16s %%s" %% (sys.argv[1], "   ".join(out)))
''' % (ROOT, ROOT)
for name in (sys.argv[1:] or ["main"]):
    subprocess.run([sys.executable, "-c", CHILD, name], check=False)
