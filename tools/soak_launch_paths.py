#!/usr/bin/env python3
"""Soak test of the launch paths (not part of the suite: minutes): the seeded random call sequences of
tests/test_gpu_api.py::test_random_call_sequences_equal_on_every_launch_path for many more seeds, then tools/coherent_memory_hazard.py
again and again.  Exit status 1 at the first difference.

    python tools/soak_launch_paths.py [n_seeds] [n_hazard_runs]
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n_hazard = int(sys.argv[2]) if len(sys.argv) > 2 else 10

import test_gpu_api as T  # noqa: E402

bad = 0
for seed in range(1000, 1000 + n_seeds):
    for case in ("dream_gauss100", "demc_banana_snooker", "dream_mix8"):
        try:
            T.test_random_call_sequences_equal_on_every_launch_path.__wrapped__(case, seed) if hasattr(
                T.test_random_call_sequences_equal_on_every_launch_path, "__wrapped__") else T.test_random_call_sequences_equal_on_every_launch_path(case, seed)
        except AssertionError as e:
            bad += 1
            print("DIFFERENT: case %s seed %d: %s" % (case, seed, str(e)[:200]), flush=True)
    if (seed - 999) % 10 == 0:
        print("random sequences: %d seeds done, %d differences" % (seed - 999, bad), flush=True)
for i in range(n_hazard):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "coherent_memory_hazard.py")], cwd=ROOT, capture_output=True, text=True, timeout=600)
    ok = out.returncode == 0 and "all four histories equal" in out.stdout
    bad += 0 if ok else 1
    print("hazard tool run %d: %s" % (i + 1, "equal" if ok else "DIFFERENT"), flush=True)
print("soak: %d differences" % bad)
sys.exit(1 if bad else 0)
