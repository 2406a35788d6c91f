"""bipymc_amd: MI355X-native DE-MC / DREAM population sampler.

Drop-in for the parallel samplers of wgurecky/bipymc (`bipymc.demc.DeMcMpi`,
`bipymc.dream.DreamMpi`): same constructor, `run_mcmc`, `param_est` and
`ln_like_fn` callback surface; the per-generation hot path runs as hand-written
HIP kernels behind the C ABI of include/bipymc_hip.h (libbipymc_hip.so).
"""
__version__ = "0.1.0"

from .demc import DeMcMpi  # noqa: F401
from .dream import DreamMpi  # noqa: F401
from .device_likelihood import HipLikelihood  # noqa: F401
