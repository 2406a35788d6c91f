#!/usr/bin/env python3
"""EXPERIMENT (round 5): a generation loop resident on ONE XCD for the small-state configurations.

cfg3 (DE-MC, banana, N = 65536, snooker 0.1: 1 MB of state) and cfg5's per-GPU share (DREAM, 8-D mixture, N = 32768: 2 MB) fit one XCD's 4 MB L2.
The shipped path pays two whole-GPU dependent dispatches per generation; the experiment build (make -C bipymc_amd/csrc variant NAME=xcd
DEFS=-DBPM_EXPERIMENT_XCD, kernels.h: xcd_resident_kernel) runs up to 64 generations as ONE launch whose 32 worker workgroups all sit on one XCD and meet
at an L2-level barrier between half generations.  This script times both on the same start and compares the final states bit for bit:

    python tools/xcd_resident.py            # parent: runs every (config, mode) in a child process, prints one JSON line each
modes: shipped (product library, own AQL queue) | stream (product library, HIP stream launches) | xcd (experiment: the resident loop) |
       barrier (experiment: the same launch shape with NO work: what the barriers alone cost)
"""
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
XCD_LIB = os.path.join(ROOT, "build_variants", "libbipymc_xcd.so")
G_WARM, G_TIMED = 256, 4096


def child(cfg, mode):
    from bipymc_amd import _lib as L
    from bipymc_amd.engine import HipEngine
    from bipymc_amd.utils import banana_rv, mixture_nd
    np.random.seed(20261005)
    if cfg == "cfg3":
        tgt, N, algo, kw = banana_rv.Banana_2D(), 65536, L.ALGO_DEMC, dict(p_snooker=0.1)
        y1, y2 = tgt.rvs(N)
        x0 = np.stack([y1, y2], axis=1)
    else:
        tgt, N, algo, kw = mixture_nd.BimodeGauss_ND(8), 32768, L.ALGO_DREAM, dict(burnin_gen=0)
        x0 = tgt.rvs(N)
    tid, tp, d = tgt._bpm_target_spec()
    e = HipEngine(algo=algo, n_chains=N, dim=d, target_id=tid, target_params=tp, seed=42, **kw)
    if mode == "stream":
        e.set_launch_path(direct=False)
    e.set_state(x0)
    e.reserve_history(G_WARM + G_TIMED + 2)
    e.begin_run()
    e.step(G_WARM)
    e.synchronize()
    t0 = time.perf_counter()
    e.step(G_TIMED)
    e.synchronize()
    el = time.perf_counter() - t0
    X = e.get_state()
    H = e.get_history(G_WARM + G_TIMED - 1, G_WARM + G_TIMED + 1)
    st = e.stats()
    print(json.dumps(dict(config=cfg, mode=mode, n_chains=N, dim=d, generations=G_TIMED, us_per_generation=el / G_TIMED * 1e6,
                          chain_updates_per_s=N * G_TIMED / el, accepted=int(st["local_n_accepted"]),
                          state_sha=hashlib.sha256(X.tobytes()).hexdigest()[:16], history_sha=hashlib.sha256(H.tobytes()).hexdigest()[:16],
                          launch=e.launch_stats())), flush=True)
    e.close()


def main():
    if len(sys.argv) == 3:
        return child(sys.argv[1], sys.argv[2])
    for cfg in ("cfg3", "cfg5s"):
        for mode in ("shipped", "stream", "xcd", "xcd-inv-sc1", "barrier"):
            env = dict(os.environ)
            if mode in ("xcd", "barrier", "xcd-inv-sc1"):
                # (xcd-inv-sc1: the same loop with `buffer_inv sc1` -- the agent-scope invalidate -- behind every barrier instead of `buffer_inv sc0`)
                env.update(BPM_LIB_PATH=XCD_LIB.replace("_xcd.so", "_xcd1.so") if mode == "xcd-inv-sc1" else XCD_LIB, BPM_XCD="2" if mode == "barrier" else "1")
            r = subprocess.run([sys.executable, os.path.abspath(__file__), cfg, mode], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
            out = r.stdout.decode().strip().splitlines()
            if r.returncode != 0 or not out:
                print(json.dumps(dict(config=cfg, mode=mode, error=r.stderr.decode()[-600:])), flush=True)
            else:
                print(out[-1], flush=True)


if __name__ == "__main__":
    main()
