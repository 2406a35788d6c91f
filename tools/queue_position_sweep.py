#!/usr/bin/env python3
"""Diagnostic: every kind of call the generation loop makes on the library's own AQL queue, started at every position 236 ... 254 of an epoch of 256
packets (bpm_debug_queue_pad) -- in the ring's first lap (a fresh process per position) and in later laps (one process, all positions) -- against
the same calls launched on the HIP stream: states, histories, p_cr bit-identical, no wait running into its limit.  The epoch marker, the drain's fence
kernel and barrier packet, the timing signals and the table builds all meet the epoch boundary somewhere in this sweep.
usage: queue_position_sweep.py            (parent)   -> exit code 1 on any difference or failure"""
import ctypes as C
import hashlib
import os
# (this tool uses test hooks / BPM_TEST_PATHS: it runs on the test variant of the library, include/bipymc_hip_test.h)
os.environ.setdefault("BPM_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build_variants", "libbipymc_test.so"))
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make(direct):
    from bipymc_amd import _lib as L
    from bipymc_amd.engine import HipEngine
    from bipymc_amd.utils import d100_gauss
    tid, tp, d = d100_gauss.Gauss_100D(dim=8)._bpm_target_spec()
    e = HipEngine(algo=L.ALGO_DREAM, n_chains=40, dim=d, target_id=tid, target_params=tp, seed=5, burnin_gen=30, n_cr_gen=3)
    if not direct:
        e.set_launch_path(direct=0)
    e.set_state(np.random.RandomState(2).normal(size=(40, d)))
    e.begin_run()
    return e


def ops(e, pad=None):
    """the call mix; pad(k) is called in front of each call with the call's index"""
    h = hashlib.sha256()
    calls = [lambda: e.step(3), lambda: e.step_timed(2), lambda: h.update(e.get_history(0, 4).tobytes()), lambda: e.step(1), lambda: e.step(70),
             lambda: h.update(e.get_loglike().tobytes()), lambda: e.step(2), lambda: e.synchronize(), lambda: e.step(5), lambda: h.update(e.get_history().tobytes())]
    for k, c in enumerate(calls):
        if pad:
            pad(k)
        c()
    e.synchronize()
    h.update(e.get_state().tobytes())
    h.update(np.ascontiguousarray(e.stats()["p_cr"]).tobytes())
    return h.hexdigest()[:16]


def child(first_pos, laps):
    from bipymc_amd import _lib as L
    ref_e = make(False)
    ref = ops(ref_e)
    ref_e.close()
    w = C.c_int64(0)
    bad = 0
    for lap in range(laps):
        for pos in ([first_pos] if laps == 1 else range(236, 255)):
            e = make(True)
            # every call of the mix starts `k` packets further from the marker: pos, pos - 1, ... (clamped)
            got = ops(e, pad=lambda k: L.check(e.lib.bpm_debug_queue_pad(e._h, max(0, min(254, pos - (k % 4))), C.byref(w))))
            ls = e.launch_stats()
            e.close()
            ok = got == ref and ls["direct"] > 0
            bad += 0 if ok else 1
            print("lap %d position %d: %s (write index %d)" % (lap, pos, "identical to the stream launches" if ok else "DIFFERENT", w.value), flush=True)
    return bad


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        sys.exit(1 if child(int(sys.argv[2]), int(sys.argv[3])) else 0)
    env = dict(os.environ, BPM_QUEUE_TIMEOUT_S="20")
    bad = 0
    for pos in range(236, 255):                       # first lap: a fresh process each
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(pos), "1"], env=env, capture_output=True, text=True, timeout=300)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("lap")]
        print("fresh process, " + (line[0] if line else "NO RESULT: " + r.stderr[-300:]), flush=True)
        bad += 1 if r.returncode else 0
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "0", "3"], env=env, capture_output=True, text=True, timeout=900)
    print("\n".join(ln for ln in r.stdout.splitlines() if ln.startswith("lap")), flush=True)
    bad += 1 if r.returncode else 0
    print("queue position sweep: %d failure(s)" % bad)
    sys.exit(1 if bad else 0)
