import os, sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from bipymc_amd import _lib as L
from bipymc_amd.engine import HipEngine
N, d = 8192, 100
sig = np.sqrt(np.arange(d) + 1.0); rho = 0.5
c0 = -0.5 * (d * np.log(2 * np.pi) + 2 * np.sum(np.log(sig)) + (d - 1) * np.log(1 - rho) + np.log(1 + (d - 1) * rho))
a, b = 1.0 / (1 - rho), rho / ((1 - rho) * (1 + (d - 1) * rho))
src = """__device__ double ln_like(const double* x, int d, const double* p) {
    double s1 = 0.0, s2 = 0.0;
    for (int j = 0; j < d; ++j) { const double z = x[j] * p[3 + j]; s1 += z; s2 += z * z; }
    return p[0] - 0.5 * (p[1] * s2 - p[2] * s1 * s1);
}"""
if os.environ.get("HIP_SOURCE_TERMS"):      # the per-coordinate form of the same likelihood
    src = """#define BPM_LN_LIKE_TERMS 2
__device__ void ln_like_terms(double xj, int j, int d, const double* p, double* acc) { const double z = xj * p[3 + j]; acc[0] += z; acc[1] += z * z; }
__device__ double ln_like_finish(const double* acc, int d, const double* p) { return p[0] - 0.5 * (p[1] * acc[1] - p[2] * acc[0] * acc[0]); }"""
rs = np.random.RandomState(1)
X0 = sig * (np.sqrt(0.5) * rs.standard_normal((N, 1)) + np.sqrt(0.5) * rs.standard_normal((N, d)))
e = HipEngine(algo=L.ALGO_DREAM, n_chains=N, dim=d, target_id=L.TARGET_HOST_CALLBACK, target_params=None, seed=42, burnin_gen=int(os.environ.get("HIP_SOURCE_BURNIN", "0")), n_cr_gen=5, n_cr=3)
e.set_state(X0); e.set_device_likelihood(src, np.concatenate([[c0, a, b], 1 / sig])); e.reserve_history(1000); e.begin_run()
e.step(50); e.synchronize()
t0 = time.perf_counter(); e.step(300); e.synchronize(); el = time.perf_counter() - t0
print("us per generation %.2f  updates/s %.3g" % (el / 300 * 1e6, N * 300 / el), e.device_likelihood_info(), e.launch_stats())
e.close()
