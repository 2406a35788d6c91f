#!/usr/bin/env python3
"""Diagnostic: a world of R rank PROCESSES SHARING ONE GPU over the push exchange (IPC-mapped buffers, no RCCL): time per generation
as every rank sees it.  With tiny per-rank populations the update kernels are short and what remains is the cross-rank hand-over
(push_sync_kernel: announce + wait, one extra dependent dispatch per half generation) -- the part of an N-GPU run that a one-GPU box
can measure; the xGMI transfer of the rows cannot be.
usage: push_world.py R [chains_per_rank] [generations] [dim]        (parent)"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def worker(d_, rank, R, n_per, G, dim):
    from _file_comm import FileComm
    from bipymc_amd import _lib as L
    from bipymc_amd.engine import HipEngine
    from bipymc_amd.utils import d100_gauss
    comm = FileComm(d_, rank, R)
    g = d100_gauss.Gauss_100D(dim=dim)
    tid, tp, d = g._bpm_target_spec()
    N = n_per * R
    np.random.seed(3)
    x0 = g.rvs(N)
    e = HipEngine(algo=L.ALGO_DREAM, n_chains=N, dim=d, target_id=tid, target_params=tp, seed=11, rank=rank, world_size=R,
                  nccl_uid=HipEngine.push_uid(), burnin_gen=0)
    e.push_connect(comm.allgather(e.push_export()))
    comm.Barrier()
    assert e.push_selftest()
    e.set_state(x0)
    e.reserve_history(3 * G + 80)
    e.begin_run()
    comm.Barrier()
    e.step(50)
    e.synchronize()
    best = 1e9
    for _ in range(3):
        comm.Barrier()
        t0 = time.perf_counter()
        e.step(G)
        e.synchronize()
        best = min(best, (time.perf_counter() - t0) / G)
    ms, nl = e.step_timed(G)
    print("rank %d of %d: %d chains per rank, d = %d: %.2f us per generation by the host clock, %.2f us per update-kernel launch period "
          "by dispatch time stamps; %s; %s" % (rank, R, n_per, dim, best * 1e6, ms * 1e3 / max(nl, 1), e.exchange_stats(), e.launch_stats()), flush=True)
    comm.Barrier()
    e.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7]))
        sys.exit(0)
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    n_per = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    G = int(sys.argv[3]) if len(sys.argv) > 3 else 300
    dim = int(sys.argv[4]) if len(sys.argv) > 4 else 100
    assert R <= 5, "at most 6 GPU processes on a box"
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    with tempfile.TemporaryDirectory() as td:
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", td, str(r), str(R), str(n_per), str(G), str(dim)], env=env)
                 for r in range(R)]
        rc = 0
        for p in procs:
            rc = rc or p.wait(timeout=600)
    sys.exit(rc)
