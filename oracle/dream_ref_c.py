"""ORACLE (test infrastructure): ctypes loader of oracle/_build/libdream_ref.so (oracle/csrc/dream_ref.c),
the plain-C + OpenMP restatement of the DREAM generation used as bench.py's CPU baseline and as an
independent cross-check of oracle/sampler_ref.py."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libdream_ref.so")
_lib = None


def load(build=True):
    global _lib
    if _lib is None:
        if not os.path.exists(_SO) and build:
            subprocess.check_call(["make", "-C", _HERE])
        lib = C.CDLL(_SO)
        dp = C.POINTER(C.c_double)
        lib.dream_run.restype = C.c_long
        lib.dream_run.argtypes = [dp, dp, C.c_int, C.c_int, dp, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, C.c_int, C.c_int,
                                  dp, C.c_double, C.c_double, C.c_double, C.c_double, dp, C.c_int]
        lib.dream_ref_max_threads.restype = C.c_int
        _lib = lib
    return _lib


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def dream_run(X, ll, params, seed, t0, k0, n_gens, del_pairs=3, p_cr=(1 / 3., 1 / 3., 1 / 3.), gamma_scale=1.0,
              flip=0.5, epsilon=1e-12, u_epsilon=1e-2, keep_history=False, n_threads=0):
    """Advances X (N, d) and ll (N) in place by n_gens generations; returns (accepted, history or None)."""
    lib = load()
    assert X.dtype == np.float64 and X.flags.c_contiguous and ll.dtype == np.float64
    N, d = X.shape
    params = np.ascontiguousarray(params, dtype=np.float64)
    p_cr = np.ascontiguousarray(p_cr, dtype=np.float64)
    hist = np.empty((n_gens, N, d)) if keep_history else None
    acc = lib.dream_run(_p(X), _p(ll), N, d, _p(params), int(seed), int(t0), int(k0), int(n_gens), int(del_pairs), p_cr.size,
                        _p(p_cr), float(gamma_scale), float(flip), float(epsilon), float(u_epsilon),
                        _p(hist) if keep_history else None, int(n_threads))
    if acc < 0:
        raise ValueError("dream_ref.c: dim <= 512 and del_pairs <= 10 required")
    return int(acc), hist


def max_threads():
    return int(load().dream_ref_max_threads())
