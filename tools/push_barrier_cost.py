#!/usr/bin/env python3
"""Diagnostic: what the cross-rank hand-over of the push exchange costs per half generation, measured on ONE GPU.
R ranks as handles of one process, every rank on an AQL queue of its own (BPM_TEST_PATHS=groupqueues): the update kernels push
accepted rows into the other replicas, the one-wavefront push_sync_kernel of every rank announces and waits for the others ACROSS the
queues -- the very packets of a multi-GPU run, with the GPU's memory in place of xGMI.  With small populations per rank the ranks'
update kernels do not compete for CUs, so (generation time of the world) - (generation time of ONE rank alone, same chains) is two
hand-overs: the extra dependent dispatch + announce -> poll latency.  What it cannot show: xGMI store latency (add ~1-2 us per hand-over).
usage: BPM_TEST_PATHS=groupqueues python tools/push_barrier_cost.py [R] [chains_per_rank] [generations]"""
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
from bipymc_amd.utils import d100_gauss   # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n_per = int(sys.argv[2]) if len(sys.argv) > 2 else 512
G = int(sys.argv[3]) if len(sys.argv) > 3 else 400
g = d100_gauss.Gauss_100D()
tid, tp, d = g._bpm_target_spec()
np.random.seed(3)

# one rank alone with the same number of chains (no exchange at all)
one = HipEngine(algo=L.ALGO_DREAM, n_chains=n_per, dim=d, target_id=tid, target_params=tp, seed=11, burnin_gen=0)
one.set_state(g.rvs(n_per))
one.reserve_history(4 * G + 100)
one.begin_run()
one.step(60)
one.synchronize()
t1 = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    one.step(G)
    one.synchronize()
    t1 = min(t1, (time.perf_counter() - t0) / G)
one.close()

N = n_per * R
x0 = g.rvs(N)
uid = b"BPMLOCAL" + bytes(120)
ranks = [HipEngine(algo=L.ALGO_DREAM, n_chains=N, dim=d, target_id=tid, target_params=tp, seed=11, rank=r, world_size=R, nccl_uid=uid, lib=L.load_test(), burnin_gen=0)
         for r in range(R)]
blobs = [e.push_export() for e in ranks]
scope = os.environ.get("PUSH_SCOPE", "system")
for e in ranks:
    e.push_connect(blobs)
    e.set_exchange("push-agent" if scope == "agent" else "push")
    e.set_state(x0)
    e.reserve_history(4 * G + 100)
    e.begin_run()
arr = (C.c_void_p * R)(*[e._h for e in ranks])
L.check(ranks[0].lib.bpm_local_group_step(arr, R, 60), ranks[0].lib)
tw = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    L.check(ranks[0].lib.bpm_local_group_step(arr, R, G), ranks[0].lib)
    tw = min(tw, (time.perf_counter() - t0) / G)
ls = ranks[0].launch_stats()
# the same world with ONE HOST THREAD PER RANK (bpm_step per rank; ctypes releases the GIL): the ranks' hosts enqueue side by side like
# the processes of a multi-GPU run, instead of one thread feeding R queues in turn
import threading
tt = 1e9
bar = threading.Barrier(R + 1)


def drive(e):
    for _ in range(4):
        bar.wait()
        e.step(G)
        e.synchronize()
        bar.wait()


th = [threading.Thread(target=drive, args=(e,)) for e in ranks]
for t in th:
    t.start()
for i in range(4):
    bar.wait()
    t0 = time.perf_counter()
    bar.wait()
    if i > 0:
        tt = min(tt, (time.perf_counter() - t0) / G)
for t in th:
    t.join()
print("[%s-scope fences on the update packets] R=%d ranks x %d chains (d=100), a queue per rank: %.2f us per generation with ONE host thread feeding all queues, %.2f us with a host "
      "thread per rank; one rank alone with %d chains: %.2f us per generation -> %.2f us per hand-over (two per generation); update launches "
      "of rank 0: %d on its own queue, %d on the HIP stream"
      % (scope, R, n_per, tw * 1e6, tt * 1e6, n_per, t1 * 1e6, (tt - t1) * 1e6 / 2.0, ls["direct"], ls["stream"]))
for e in ranks:
    e.close()
