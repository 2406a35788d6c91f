#!/usr/bin/env python3
"""Soak of the push exchange (diagnostic): worlds of R rank PROCESSES SHARING ONE GPU (IPC-mapped buffers, every rank on its own AQL queue) run LONG
sequences -- thousands of generations, CR adaptation, table windows rebuilt inside the loop, several bpm_step calls -- under both fence scopes, and
every rank's final replica, ln-likes, adapted p_cr and a checksum over its whole history must equal the single-rank run's, bit for bit.  A race in
the hand-over (a flag seen before the rows it announces, a window built from another rank's records) shows up as a difference here.
usage: push_soak.py [R] [repeats]   -> one line per (case, scope, seed), exit code 1 on any difference"""
import hashlib
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# name: (target, algo, chains per rank, kwargs, generations, step calls)
CASES = {
    "dream_gauss100_cfg4_rank_shape": ("gauss100", "dream", 8192, dict(burnin_gen=300, n_cr_gen=20), 1500, 3),
    "dream_gauss100_small": ("gauss100", "dream", 96, dict(burnin_gen=2000, n_cr_gen=50), 12000, 7),
    "dream_mix8_outlier": ("mix8", "dream", 2048, dict(burnin_gen=600, n_cr_gen=10, del_pairs=2, outlier_every=50), 3000, 4),
    "demc_banana_snooker": ("banana", "demc", 4096, dict(p_snooker=0.1), 6000, 5),
}


def spec_of(case):
    from bipymc_amd import _lib as L
    from bipymc_amd.utils import banana_rv, d100_gauss, mixture_nd
    tgt, algo, n_per, kw, G, calls = CASES[case]
    t = dict(gauss100=lambda: d100_gauss.Gauss_100D(), mix8=lambda: mixture_nd.BimodeGauss_ND(8), banana=lambda: banana_rv.Banana_2D())[tgt]()
    return t._bpm_target_spec(), (L.ALGO_DREAM if algo == "dream" else L.ALGO_DEMC), n_per, kw, G, calls


def run_calls(e, G, calls):
    per = G // calls
    done = 0
    for c in range(calls):
        n = per if c < calls - 1 else G - done
        e.step(n)
        done += n
    e.synchronize()


def hist_sha(e, G, col_lo, col_hi, piece=256):
    h = hashlib.sha256()
    for g0 in range(0, G + 1, piece):
        H = e.get_history(g0, min(G + 1, g0 + piece))
        h.update(np.ascontiguousarray(H[:, col_lo:col_hi]).tobytes())
    return h.hexdigest()[:16]


def worker(d_, rank, R, case, seed, scope):
    from _file_comm import FileComm
    from bipymc_amd.engine import HipEngine
    comm = FileComm(d_, rank, R)
    (tid, tp, d), algo, n_per, kw, G, calls = spec_of(case)
    N = n_per * R
    x0 = np.random.RandomState(seed).normal(size=(N, d)) + 0.5
    e = HipEngine(algo=algo, n_chains=N, dim=d, target_id=tid, target_params=tp, seed=seed, rank=rank, world_size=R, nccl_uid=HipEngine.push_uid(), **kw)
    e.push_connect(comm.allgather(e.push_export()))
    comm.Barrier()
    assert e.push_selftest()
    e.set_exchange(mode="push-agent" if scope == "agent" else "push")
    e.set_state(x0)
    e.reserve_history(G + 2)
    e.begin_run(flip=0.4)
    comm.Barrier()
    t0 = time.perf_counter()
    run_calls(e, G, calls)
    dt = time.perf_counter() - t0
    st = e.stats()
    out = dict(state=hashlib.sha256(np.ascontiguousarray(e.get_state()).tobytes()).hexdigest()[:16],
               ll=hashlib.sha256(np.ascontiguousarray(e.get_loglike()).tobytes()).hexdigest()[:16],
               p_cr=[float(v) for v in st["p_cr"]], acc=int(st["local_n_accepted"]), resets=int(st["n_outlier_resets"]),
               hist=hist_sha(e, G, 0, n_per), us_per_gen=dt / G * 1e6, push_gens=e.exchange_stats()["push_gens"])
    import json
    json.dump(out, open(os.path.join(d_, "soak_rank%d.json" % rank), "w"))
    comm.Barrier()
    e.close()


def single(case, seed, R):
    from bipymc_amd.engine import HipEngine
    (tid, tp, d), algo, n_per, kw, G, calls = spec_of(case)
    N = n_per * R
    x0 = np.random.RandomState(seed).normal(size=(N, d)) + 0.5
    e = HipEngine(algo=algo, n_chains=N, dim=d, target_id=tid, target_params=tp, seed=seed, **kw)
    e.set_state(x0)
    e.reserve_history(G + 2)
    e.begin_run(flip=0.4)
    run_calls(e, G, calls)
    st = e.stats()
    ll = np.ascontiguousarray(e.get_loglike())
    out = dict(state=hashlib.sha256(np.ascontiguousarray(e.get_state()).tobytes()).hexdigest()[:16],
               ll=[hashlib.sha256(ll[r * n_per:(r + 1) * n_per].tobytes()).hexdigest()[:16] for r in range(R)],
               p_cr=[float(v) for v in st["p_cr"]], acc=int(st["local_n_accepted"]), resets=int(st["n_outlier_resets"]),
               hist=[hist_sha(e, G, r * n_per, (r + 1) * n_per) for r in range(R)])
    e.close()
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], int(sys.argv[6]), sys.argv[7])
        sys.exit(0)
    import json
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    bad = 0
    for case in CASES:
        for seed in range(11, 11 + reps):
            ref = single(case, seed, R)
            for scope in ("agent", "system"):
                with tempfile.TemporaryDirectory() as d_:
                    env = dict(os.environ, BPM_PUSH_TIMEOUT_S="60")
                    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", d_, str(r), str(R), case, str(seed), scope], env=env)
                          for r in range(R)]
                    rcs = [p.wait(timeout=900) for p in ps]
                    if any(rcs):
                        print("%s seed %d %s: worker exit codes %s" % (case, seed, scope, rcs), flush=True)
                        bad += 1
                        continue
                    outs = [json.load(open(os.path.join(d_, "soak_rank%d.json" % r))) for r in range(R)]
                same = all(o["state"] == ref["state"] and o["ll"] == ref["ll"][r] and o["p_cr"] == ref["p_cr"] and o["hist"] == ref["hist"][r] and
                           o["resets"] == ref["resets"] for r, o in enumerate(outs)) and sum(o["acc"] for o in outs) == ref["acc"]
                bad += 0 if same else 1
                print("%-32s seed %d  %-6s fences  R=%d x %d chains, %5d generations in %d calls: %s  (%.0f us per generation, %d accepted, %d outlier resets)"
                      % (case, seed, scope, R, CASES[case][2], CASES[case][4], CASES[case][5], "IDENTICAL to the single-rank run" if same else "DIFFERENT",
                         outs[0]["us_per_gen"], ref["acc"], ref["resets"]), flush=True)
    print("push soak: %d difference(s)" % bad)
    sys.exit(1 if bad else 0)
