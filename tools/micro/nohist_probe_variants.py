"""What in bench.py's process makes rocprofv3 show a second mode of slow launches?  Variants of tools/micro/nohist_probe.py (env VARIANT):
timed = bpm_step_timed instead of bpm_step; burnin = 200 burn-in generations first (bench.py's sampler); thread = a watchdog-like Python thread;
all = everything."""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from bipymc_amd import _lib as L
from bipymc_amd.engine import HipEngine
from bipymc_amd.utils import d100_gauss
v = os.environ.get("VARIANT", "")
t = d100_gauss.Gauss_100D(rho=0.5)
tid, tp, d = t._bpm_target_spec()
N = 8192
np.random.seed(1)
burn = 200 if ("burnin" in v or v == "all") else 0
e = HipEngine(algo=L.ALGO_DREAM, n_chains=N, dim=d, target_id=tid, target_params=tp, seed=42, burnin_gen=burn, n_cr_gen=50, n_cr=3)
e.set_state(t.rvs(N))
e.reserve_history(1400)
if "thread" in v or v == "all":
    stop = threading.Event()
    threading.Thread(target=lambda: [time.sleep(0.05) for _ in iter(lambda: stop.is_set(), True)], daemon=True).start()
e.begin_run()
if burn:
    e.step(burn)
e.step(50); e.synchronize()
t0 = time.perf_counter()
if "timed" in v or v == "all":
    e.step_timed(1000, read=False)
else:
    e.step(1000)
e.synchronize()
print("variant %-8s us per generation %.2f" % (v or "-", (time.perf_counter() - t0) / 1000 * 1e6))
e.close()
