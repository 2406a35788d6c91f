#!/usr/bin/env python3
"""A/B of experiment switches (BPM_TEST_PATHS values, read once per process -> one child process per value): time per generation and
a hash of the final state / p_cr for a set of workloads; every variant must leave the same bits.
usage: ab_paths.py "<paths A>" "<paths B>" ...      ("" = the default)      workloads: WORKLOADS below or AB_WORKLOADS=cfg2,cfg2burn,..."""
import hashlib
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def workloads():
    from bipymc_amd import _lib as L
    from bipymc_amd.utils import banana_rv, d100_gauss, mixture_nd
    g, b, m = d100_gauss.Gauss_100D(), banana_rv.Banana_2D(), mixture_nd.BimodeGauss_ND(8)
    return {
        "cfg2": (L.ALGO_DREAM, g, 8192, 500, dict(burnin_gen=0)),
        "cfg2burn": (L.ALGO_DREAM, g, 8192, 300, dict(burnin_gen=10 ** 6, n_cr_gen=5)),
        "cfg3": (L.ALGO_DEMC, b, 65536, 500, dict(p_snooker=0.1)),
        "cfg5share": (L.ALGO_DREAM, m, 32768, 500, dict(burnin_gen=0)),
        "cfg5": (L.ALGO_DREAM, m, 262144, 200, dict(burnin_gen=0)),
        "cfg5burn": (L.ALGO_DREAM, m, 262144, 100, dict(burnin_gen=10 ** 6, n_cr_gen=5)),
        "cfg2small": (L.ALGO_DREAM, g, 2048, 500, dict(burnin_gen=0)),
        "gauss65536": (L.ALGO_DREAM, g, 65536, 100, dict(burnin_gen=0)),
        "gauss16384": (L.ALGO_DREAM, g, 16384, 300, dict(burnin_gen=0)),
        "gauss32768": (L.ALGO_DREAM, g, 32768, 200, dict(burnin_gen=0)),
        "cfg5shareburn": (L.ALGO_DREAM, m, 32768, 300, dict(burnin_gen=10 ** 6, n_cr_gen=5)),
    }


def child(names):
    from bipymc_amd.engine import HipEngine
    W = workloads()
    for name in names:
        algo, tgt, N, G, kw = W[name]
        tid, tp, d = tgt._bpm_target_spec()
        np.random.seed(5)
        x0 = tgt.rvs(N)
        if isinstance(x0, tuple):
            x0 = np.stack(x0, axis=1)
        e = HipEngine(algo=algo, n_chains=N, dim=d, target_id=tid, target_params=tp, seed=42, **kw)
        e.set_state(x0)
        e.reserve_history(4 * G + 100)
        e.begin_run()
        e.step(60)
        e.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            e.step(G)
            e.synchronize()
            best = min(best, (time.perf_counter() - t0) / G)
        st = e.stats()
        h = hashlib.sha256(np.ascontiguousarray(e.get_state()).tobytes() + np.ascontiguousarray(st["p_cr"]).tobytes()
                           + np.ascontiguousarray(e.get_history(3 * G + 60, 3 * G + 61)).tobytes()).hexdigest()[:12]
        print("RESULT %s %.3f %s %s" % (name, best * 1e6, h, e.launch_stats()["fence"]), flush=True)
        e.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(sys.argv[2].split(","))
        sys.exit(0)
    variants = sys.argv[1:] or [""]
    names = os.environ.get("AB_WORKLOADS", "cfg2,cfg2burn,cfg3,cfg5share").split(",")
    table = {}
    for rep in range(int(os.environ.get("AB_REPS", "2"))):
        for v in variants:
            env = dict(os.environ)
            env.pop("BPM_TEST_PATHS", None)
            if v.startswith("lib="):           # an experiment build of the library instead of a test path
                env["BPM_LIB_PATH"] = os.path.join(ROOT, v[4:])
            elif v:                            # (a test path: read by the test variant of the library only)
                env["BPM_TEST_PATHS"] = v
                env["BPM_LIB_PATH"] = os.path.join(ROOT, "build_variants", "libbipymc_test.so")
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", ",".join(names)], env=env, capture_output=True, text=True, timeout=900)
            if out.returncode != 0:
                print("variant %r failed:\n%s" % (v, out.stderr[-1500:]))
                continue
            for ln in out.stdout.splitlines():
                if ln.startswith("RESULT"):
                    _, name, us, h, fence = ln.split()
                    table.setdefault(name, {}).setdefault(v, []).append((float(us), h))
    print("%-12s" % "workload" + "".join("%-34s" % ("[" + (v or "default") + "]") for v in variants))
    for name in names:
        row, hashes = "%-12s" % name, set()
        for v in variants:
            r = table.get(name, {}).get(v, [])
            hashes.update(h for _, h in r)
            row += "%-34s" % (" / ".join("%.2f" % u for u, _ in r) + " us")
        print(row + ("same bits" if len(hashes) == 1 else "BITS DIFFER: %s" % sorted(hashes)))
