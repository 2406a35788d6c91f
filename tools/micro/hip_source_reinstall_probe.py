import sys, os, numpy as np
sys.path.insert(0, "/root/repo")
from bipymc_amd import _lib as L
from bipymc_amd.engine import HipEngine
import resource
SRC = "__device__ double ln_like(const double* x, int d, const double* p) { double s = 0; for (int j = 0; j < d; ++j) s += x[j] * x[j] * p[0]; return -0.5 * s; }"
e = HipEngine(algo=L.ALGO_DREAM, n_chains=256, dim=20, target_id=L.TARGET_HOST_CALLBACK, target_params=None, seed=1, burnin_gen=3, n_cr_gen=1)
e.set_state(np.random.RandomState(0).normal(size=(256, 20)))
for i in range(12):
    e.set_device_likelihood(SRC.replace("p[0]", "(p[0] + %d.0)" % i), [1.0])
    e.begin_run(); e.step(6)
    if i in (0, 5, 11):
        print(i, "maxrss MB", resource.getrusage(resource.RUSAGE_SELF).ru_maxrss // 1024, e.device_likelihood_info()[0], e.launch_stats()["direct"])
e.close()
# many samplers one after the other
for i in range(6):
    e = HipEngine(algo=L.ALGO_DEMC, n_chains=64, dim=2, target_id=L.TARGET_HOST_CALLBACK, target_params=None, seed=i)
    e.set_state(np.zeros((64, 2))); e.set_device_likelihood(SRC, [1.0]); e.begin_run(); e.step(4); e.close()
print("ok, maxrss MB", resource.getrusage(resource.RUSAGE_SELF).ru_maxrss // 1024)
