#!/usr/bin/env python3
"""A user-written likelihood at device speed: ln_like_fn as a few lines of HIP source (bipymc_amd.HipLikelihood).

The reference's examples pass a Python function (examples/ex_para_fit.py:39-72); with bipymc_amd such a function runs as a host callback.  Written as
HIP C it is compiled at construction (hiprtc, half a second) into a kernel of the generation loop: a straight-line fit y = m x + c with unknown noise,
200 data points, 4096 chains."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))   # run from a checkout

from bipymc_amd import DreamMpi, HipLikelihood          # noqa: E402

rs = np.random.RandomState(0)
xs = np.linspace(0.0, 10.0, 200)
ys = 1.7 * xs - 0.4 + 0.8 * rs.standard_normal(xs.size)

SRC = """
__device__ double ln_like(const double* th, int d, const double* p) {      // th = (m, c, log sigma); p = [n, x_0 .. x_{n-1}, y_0 .. y_{n-1}]
    const int n = (int)p[0];
    const double m = th[0], c = th[1], ls = th[2];
    if (ls < -5.0 || ls > 5.0) return -INFINITY;                           // a flat prior on log sigma
    const double is2 = exp(-2.0 * ls);
    double s = 0.0;
    for (int i = 0; i < n; ++i) { const double r = p[1 + n + i] - (m * p[1 + i] + c); s += r * r; }
    return -0.5 * s * is2 - n * ls;
}
"""


def main():
    ll = HipLikelihood(SRC, params=np.concatenate([[xs.size], xs, ys]))
    sampler = DreamMpi(ll, theta_0=np.array([1.0, 0.0, 0.0]), varepsilon=1e-2, n_chains=4096, n_cr_gen=50, burnin_gen=300, seed=1)
    sampler.run_mcmc(4096 * 1500)
    mean, std, chain = sampler.param_est(n_burn=4096 * 700)
    print("m = %.3f +- %.3f   c = %.3f +- %.3f   sigma = %.3f" % (mean[0], std[0], mean[1], std[1], np.exp(mean[2])))
    return mean, std


if __name__ == "__main__":
    main()
