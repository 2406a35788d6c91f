"""A user-written likelihood at device speed: `ln_like_fn` given as HIP source.

The reference takes any Python callable as ln_like_fn (bipymc/samplers.py:36-43) and evaluates it row by row on the host; here that path is the
host callback (7-9e6 chain-updates/s at 8192 chains x 100 dimensions) or a device-resident framework callback (vectorized="device", 3e7).  A
likelihood written as a few lines of HIP C runs INSIDE the update kernel instead -- the library compiles its own update kernel around the function at
construction (hiprtc, about two seconds; include/bipymc_hip.h: bpm_set_device_likelihood), no host code per generation, 4e8 at that shape:

    ll = HipLikelihood('''
        __device__ double ln_like(const double* x, int d, const double* p) {      // p: the parameter block below (what ln_kwargs is to a callable)
            double s = 0.0;
            for (int j = 0; j < d; ++j) { const double z = (x[j] - p[j]) / p[d + j]; s += z * z; }
            return -0.5 * s;
        }''', params=np.concatenate([mu, sigma]))
    sampler = DreamMpi(ll, theta_0, n_chains=8192, ...)        # run_mcmc / param_est as ever

`python_fn` (optional): the same function for the host -- `ll(theta)` then works like any ln_like_fn (and the CPU test engine uses it)."""
import ctypes as C

import numpy as np


class HipLikelihood(object):
    """terms=K (optional, 1 ... 8): the PER-COORDINATE form for a likelihood that is a function of K sums over the coordinates.  `source` then defines

        __device__ void ln_like_terms(double xj, int j, int d, const double* p, double* acc)     // adds coordinate j's contribution into acc[0 .. K)
        __device__ double ln_like_finish(const double* acc, int d, const double* p)              // the value from the K sums

    instead of ln_like: inside the update kernel every lane of a chain adds the terms of its own coordinates and the sums meet in the kernel's reduction
    tree -- the shape of the shipped targets, and their speed (the plain form runs on one lane per chain)."""

    def __init__(self, source, params=(), python_fn=None, terms=0):
        self.terms = int(terms)
        if not 0 <= self.terms <= 8:
            raise ValueError("terms must be 0 (plain ln_like) or 1 ... 8")
        self.source = ("#define BPM_LN_LIKE_TERMS %d\n" % self.terms if self.terms else "") + str(source)
        self.params = np.ascontiguousarray(np.asarray(params, dtype=np.float64).reshape(-1))
        self.python_fn = python_fn

    def __call__(self, theta, **kwargs):
        if self.python_fn is None:
            raise TypeError("this HipLikelihood has no python_fn: it can only be evaluated on the device")
        return self.python_fn(theta, **kwargs)

    def check(self, arch=None):
        """Compile only (no GPU needed): raises ValueError with the compiler's log when the source does not build for `arch` (default gfx950)."""
        from . import _lib as L
        lib = L.load()
        log = C.create_string_buffer(1 << 16)
        if lib.bpm_check_device_likelihood(self.source.encode(), arch.encode() if arch else None, log, len(log)) != 0:
            raise ValueError(log.value.decode(errors="replace"))
        return True
