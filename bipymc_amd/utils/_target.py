"""Device-target protocol: a target object whose bound `ln_like` method is handed to
DeMcMpi/DreamMpi is recognised through `__self__._bpm_target_spec()` and evaluated
on the GPU; any other callable takes the host-callback path (samplers.py:36-43)."""
import math

import numpy as np

LN_2PI = math.log(2.0 * math.pi)
TARGET_HOST_CALLBACK, TARGET_GAUSS_EQUICORR, TARGET_MIXTURE_PAIRS, TARGET_BANANA_2D = 0, 1, 2, 3


def pair_block(mu, sg, rho):
    """[mx, my, 1/sx, 1/sy, rho, 1/(1-rho^2), ln_norm] of one bivariate normal block."""
    h = 1.0 / (1.0 - rho * rho)
    ln_norm = -(LN_2PI + math.log(sg[0]) + math.log(sg[1]) + 0.5 * math.log(1.0 - rho * rho))
    return [float(mu[0]), float(mu[1]), 1.0 / sg[0], 1.0 / sg[1], float(rho), h, ln_norm]


def log_binormal(u, v, rho, h, ln_norm):
    return ln_norm - 0.5 * (u * u - 2.0 * rho * u * v + v * v) * h


def resolve(ln_like_fn, ln_kwargs, dim):
    """-> (target_id, params or None).  Device target only for a bound method of an object that
    publishes a spec of matching dimension and when no extra kwargs are frozen in."""
    owner = getattr(ln_like_fn, "__self__", None)
    spec = getattr(owner, "_bpm_target_spec", None)
    if spec is not None and not ln_kwargs and getattr(ln_like_fn, "__name__", "") == "ln_like":
        tid, params, tdim = spec()
        if tdim == dim:
            return tid, np.ascontiguousarray(params, dtype=np.float64)
    return TARGET_HOST_CALLBACK, None
