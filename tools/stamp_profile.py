#!/usr/bin/env python3
"""Diagnostic (never part of the product): per-wavefront s_memtime timeline of the update kernel, from a
library built with -DBPM_STAMPS (build_variants/libbipymc_stamps.so).  Shares, not absolute times."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bipymc_amd import _lib as L          # noqa: E402
L.LIB_PATH = os.path.join(ROOT, "build_variants", "libbipymc_stamps.so")
from bipymc_amd.engine import HipEngine   # noqa: E402
from bipymc_amd.utils import d100_gauss   # noqa: E402

WHICH = os.environ.get("STAMP_TARGET", "gauss100")      # gauss100 (DREAM, one wavefront per chain) | banana (DE-MC, one lane per chain)
if WHICH == "banana":
    from bipymc_amd.utils import banana_rv
    tid, tp, d = banana_rv.Banana_2D()._bpm_target_spec()
    ALGO, SIZES, CPW = L.ALGO_DEMC, (8192, 65536, 262144), 64
else:
    tid, tp, d = d100_gauss.Gauss_100D()._bpm_target_spec()
    ALGO, SIZES, CPW = L.ALGO_DREAM, (1024, 8192, 65536), 1
for N in SIZES:
    e = HipEngine(algo=ALGO, n_chains=N, dim=d, target_id=tid, target_params=tp, seed=1, burnin_gen=0)
    lib = e.lib
    lib.bpm_debug_stamps.restype = C.c_int
    lib.bpm_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    e.set_state(np.random.RandomState(0).normal(size=(N, d)) * np.sqrt(np.arange(d) + 1.0))
    n_items = N // 2
    L.check(lib.bpm_debug_stamps(e._h, None, 0))        # allocate
    e.begin_run()
    e.step(30)
    n = N // 2
    out = np.zeros((n, 8), dtype=np.uint64)
    L.check(lib.bpm_debug_stamps(e._h, out.ctypes.data_as(C.c_void_p), n))
    xcc = (out[:, 7] >> np.uint64(60)).astype(np.int64)
    out[:, 7] &= np.uint64(0x0FFFFFFFFFFFFFFF)
    t = out.astype(np.float64)
    # dispatch ramp from the device-wide 100 MHz counter (10 ns ticks): when do wavefronts start / end after the first start
    rt0, rt1 = out[:, 2].astype(np.int64), out[:, 7].astype(np.int64)
    planned = rt0 > 0
    if planned.sum() > 8:
        last = planned & (rt0 >= rt0[planned].max() - 3000)             # only the last launch (the buffer is reused)
        o0 = (rt0[last] - rt0[last].min()) * 10.0
        o1 = (rt1[last] - rt0[last].min()) * 10.0
        print("N=%d last launch, %d wavefronts: start after first start p10 %.0f p50 %.0f p90 %.0f max %.0f ns; end p10 %.0f p50 %.0f p90 %.0f max %.0f ns"
              % (N, last.sum(), np.percentile(o0, 10), np.percentile(o0, 50), np.percentile(o0, 90), o0.max(),
                 np.percentile(o1, 10), np.percentile(o1, 50), np.percentile(o1, 90), o1.max()))
        for x in range(8):
            sx = last & (xcc == x)
            if sx.sum():
                print("      xcd %d: %5d wavefronts, starts %5.0f..%5.0f ns, ends %5.0f..%5.0f ns" % (x, sx.sum(), (rt0[sx].min() - rt0[last].min()) * 10.0,
                      (rt0[sx].max() - rt0[last].min()) * 10.0, (rt1[sx].min() - rt0[last].min()) * 10.0, (rt1[sx].max() - rt0[last].min()) * 10.0))
    t[:, 2] = 0
    ok = t[:, 6] > 0
    t = t[ok]
    t0 = t[:, 0].min()
    names = ["entry", "c known", "philox+header", "partner ids", "proposal (rows in)", "ll reduced", "end"]
    print("N=%d  waves=%d  (s_memtime ticks = shader cycles; 100 MHz realtime not used)" % (N, t.shape[0]))
    prev = t[:, 0]
    for i in range(1, 7):
        if np.median(t[:, i]) == 0:          # stage not on this kernel's path (e.g. in-kernel header draw when records are precomputed)
            continue
        dt = t[:, i] - prev
        print("   %-20s median %7.0f  p90 %7.0f cycles" % (names[i], np.median(dt), np.percentile(dt, 90)))
        prev = t[:, i]
    life = t[:, 6] - t[:, 0]
    print("   wave lifetime        median %7.0f  p90 %7.0f cycles" % (np.median(life), np.percentile(life, 90)))
    e.close()
