#!/usr/bin/env python3
"""Diagnostic: R ranks of a world emulated on ONE GPU (bpm_local_group_step: the real kernels, layouts and host logic of a
multi-GPU run, the all-gathers done by device copies) at BASELINE config 4's per-rank shape (8192 chains of the 100-D
Gaussian per rank).  Run under `rocprofv3 --kernel-trace --stats` to read the per-rank kernel times of an R-GPU run --
update kernel in the sharded mode, replay / scatter kernels -- which is everything but the RCCL transfer.
usage: emulate_ranks.py R [push|replay|rows|dense] [generations] [cfg4|cfg5]
push: owners store accepted rows into the other ranks' replicas from the update kernel (here: seven more buffers of the SAME GPU's HBM
instead of peer memory over xGMI), a one-wavefront kernel orders the ranks; no replay kernel.  BPM_TEST_PATHS=serial makes the emulated
ranks take turns, so a kernel trace shows each rank's kernels alone.
cfg5: BASELINE config 5's per-rank shape instead (32768 chains of the 8-D mixture per rank, 4 lanes per chain, steady state)."""
import ctypes as C
import os
# (this tool uses test hooks / BPM_TEST_PATHS: it runs on the test variant of the library, include/bipymc_hip_test.h)
os.environ.setdefault("BPM_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build_variants", "libbipymc_test.so"))
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bipymc_amd import _lib as L          # noqa: E402
from bipymc_amd.engine import HipEngine   # noqa: E402
from bipymc_amd.utils import d100_gauss, mixture_nd   # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
mode = sys.argv[2] if len(sys.argv) > 2 else "push"
G = int(sys.argv[3]) if len(sys.argv) > 3 else 40
cfg = sys.argv[4] if len(sys.argv) > 4 else "cfg4"
g = d100_gauss.Gauss_100D() if cfg == "cfg4" else mixture_nd.BimodeGauss_ND(8)
tid, tp, d = g._bpm_target_spec()
N = (8192 if cfg == "cfg4" else 32768) * R
np.random.seed(3)
x0 = g.rvs(N)
uid = b"BPMLOCAL" + bytes(120)
ranks = [HipEngine(algo=L.ALGO_DREAM, n_chains=N, dim=d, target_id=tid, target_params=tp, seed=11, rank=r, world_size=R,
                   nccl_uid=uid, lib=L.load_test(), burnin_gen=0) for r in range(R)]
if mode == "push":
    blobs = [e.push_export() for e in ranks]
    for e in ranks:
        e.push_connect(blobs)
for e in ranks:
    e.set_state(x0)
    e.begin_run()
    e.set_exchange(mode=mode)
arr = (C.c_void_p * R)(*[e._h for e in ranks])
L.check(ranks[0].lib.bpm_local_group_step(arr, R, 5), ranks[0].lib)
t0 = time.perf_counter()
L.check(ranks[0].lib.bpm_local_group_step(arr, R, G), ranks[0].lib)
dt = time.perf_counter() - t0
acc = sum(e.stats()["local_n_accepted"] for e in ranks) / float(N * (G + 5))
print("R=%d mode=%s N=%d: %d generations, %.1f us per generation for ALL ranks serialised on one GPU (acceptance %.3f); %s"
      % (R, mode, N, G, dt / G * 1e6, acc, ranks[0].exchange_stats()))

# one rank's kernels alone, its own data warm in the Infinity Cache (the lock-step run above interleaves R replicas on ONE GPU):
# the update kernel and the replay kernel of rank 0's last half generation, re-launched back to back (destructive)
upd, rep = C.c_float(0.0), C.c_float(0.0)
L.check(ranks[0].lib.bpm_debug_time_kernels(ranks[0]._h, 200, C.byref(upd), C.byref(rep)))
print("rank 0 alone, 200 launches each: update kernel %.2f us%s, replay kernel %.2f us per half generation"
      % (upd.value, " (pushes to %d peers included)" % (R - 1) if mode == "push" else "", rep.value))
