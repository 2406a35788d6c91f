"""Serial-API DE-MC sampler on MI355X -- drop-in for `bipymc.samplers.DeMc`
(reference: bipymc/samplers.py:237-336, base class :10-84) in its default `delayed_accept=True` form.

In that form every chain of a generation proposes from the population as it stood at the start of the
generation and the updates are "banked" (samplers.py:300-308): the generation is one embarrassingly
parallel launch (C ABI algo `BPM_ALGO_DEMC_SYNC`).  `delayed_accept=False` updates chains one after the
other inside a generation, which is sequential by construction, and is not offered.
"""
from __future__ import division, print_function

import numpy as np

from . import _lib as L
from .chain import DetachedChain
from .utils import _target


class DeMc(object):
    def __init__(self, log_like_fn, n_chains=8, ln_kwargs={}, **proposal_kwargs):
        assert n_chains >= 4                                         # samplers.py:249
        self.n_chains = int(n_chains)
        self.log_like_fn = log_like_fn
        self._ln_kwargs = dict(ln_kwargs)
        self._frozen_ln_like_fn = lambda theta: self.log_like_fn(theta, **self._ln_kwargs)    # samplers.py:36-43
        self.am_chains = []
        self.n_accepted = 1                                          # samplers.py:30-31 (never reset)
        self.n_rejected = 0
        self._kw = dict(proposal_kwargs)
        self._engine = None

    @property
    def frozen_ln_like_fn(self):
        return self._frozen_ln_like_fn

    def run_mcmc(self, n, theta_0, **kwargs):
        self._mcmc_run(n, theta_0, **kwargs)

    def _mcmc_run(self, n, theta_0, varepsilon=1e-6, **kwargs):
        theta_0 = np.asarray(theta_0, dtype=np.float64).reshape(-1)
        dim = len(theta_0)
        if not kwargs.get("delayed_accept", True):
            raise NotImplementedError("delayed_accept=False is sequential inside a generation; use DeMcMpi instead")
        gamma = kwargs.get("gamma", None)                            # samplers.py:264
        seed = kwargs.get("seed", self._kw.get("seed", None))
        if seed is None:
            seed = int(np.random.randint(0, 2 ** 62))
        tid, tparams = _target.resolve(self.log_like_fn, self._ln_kwargs, dim)
        from . import demc as _demc
        factory = _demc._engine_factory                                  # (the HIP engine; CPU tests patch the module attribute)
        if self._engine is not None:
            self._engine.close()
        eng = factory(algo=L.ALGO_DEMC_SYNC, n_chains=self.n_chains, dim=dim, target_id=tid, target_params=tparams,
                      seed=seed, device=self._kw.get("device", 0))
        self._engine = eng
        self._device_target = tid != L.TARGET_HOST_CALLBACK
        # _init_chains (samplers.py:255-259): chain.py:27 jitter with variance varepsilon * inflate
        eng.init_chains(theta_0, np.asarray(varepsilon, dtype=np.float64) * kwargs.get("inflate", 1e1))
        # ln_like_fn given as HIP source (device_likelihood.py): compiled into the generation loop, the sampler is stepped like one with a shipped target
        from .device_likelihood import HipLikelihood
        hip = isinstance(self.log_like_fn, HipLikelihood) and not self._device_target and hasattr(eng, "set_device_likelihood")
        if hip:
            eng.set_device_likelihood(self.log_like_fn.source, self.log_like_fn.params)      # (evaluates the jittered start states too)
        elif not self._device_target:
            X = eng.get_state()
            eng.set_loglike(np.array([self._call(x) for x in X], dtype=np.float64))
        # var_ball(varepsilon * 1e-3, dim) (samplers.py:283): VARIANCE varepsilon*1e-3 -> std for the device jitter
        eps_std = float(np.sqrt(np.max(np.asarray(varepsilon, dtype=np.float64)) * 1e-3))
        n_gens = max(0, -(-(int(n) - self.n_chains) // self.n_chains))          # while j < n - n_chains: j += n_chains
        eng.begin_run(epsilon=eps_std, gamma=gamma, shuffle=False, flip=0.0)
        eng.reserve_history(1 + n_gens)
        if self._device_target or hip:
            eng.step(n_gens)
        else:
            for _ in range(n_gens):
                props, _ids = eng.propose()
                eng.commit(np.array([self._call(p) for p in props], dtype=np.float64))
        eng.synchronize()
        st = eng.stats()
        if st["n_nan_alpha"] > 0:
            raise ValueError("probabilities contain NaN")             # samplers.py:336
        self.n_accepted += int(st["local_n_accepted"])
        self.n_rejected += int(st["local_n_rejected"]) - 1
        hist = eng.get_history()
        self.am_chains = [DetachedChain(i, hist[:, i, :]) for i in range(self.n_chains)]

    def _call(self, theta):
        v = self._frozen_ln_like_fn(np.array(theta, dtype=np.float64))
        return float(np.asarray(v).reshape(-1)[0]) if np.ndim(v) else float(v)

    @property
    def acceptance_fraction(self):
        return self.n_accepted / (self.n_accepted + self.n_rejected)   # samplers.py:75-80

    @property
    def chain(self):
        return self.am_chains[0]

    @property
    def current_pos(self):
        return self.chain.current_pos

    @property
    def super_chain(self):
        return self._super_chain()

    def _super_chain(self):
        """samplers.py:320-326: row g*n_chains + i = chain i at generation g."""
        return self._engine.get_history().reshape(-1, self.am_chains[0].dim).copy()

    def param_est(self, n_burn):
        chain_slice = self.super_chain[n_burn:, :]                      # samplers.py:311-315
        return np.mean(chain_slice, axis=0), np.std(chain_slice, axis=0), chain_slice
