import os, sys, time, numpy as np
if os.environ.get("WITH_TORCH"):
    import torch      # (first: the process then runs on the HIP runtime the torch wheel carries)
sys.path.insert(0, "/root/repo")
from bipymc_amd import _lib as L
from bipymc_amd.engine import HipEngine
from bipymc_amd.utils import d100_gauss
import os
keep = int(os.environ.get("KEEP", "0"))
t = d100_gauss.Gauss_100D()
tid, tp, d = t._bpm_target_spec()
N = 8192
np.random.seed(1)
e = HipEngine(algo=L.ALGO_DREAM, n_chains=N, dim=d, target_id=tid, target_params=tp, seed=42, burnin_gen=0, keep_history=bool(keep), running_moments=not keep)
e.set_state(t.rvs(N))
if keep: e.reserve_history(1300)
e.begin_run(); e.step(100); e.synchronize()
t0 = time.perf_counter(); e.step(1000); e.synchronize(); el = time.perf_counter() - t0
print("keep_history", keep, "us per generation %.2f" % (el / 1000 * 1e6))
e.close()
