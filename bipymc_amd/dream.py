"""Parallel DREAM sampler on MI355X -- drop-in for `bipymc.dream.DreamMpi`
(reference: bipymc/dream.py:12-144).

Adds to DeMcMpi the DREAM proposal (crossover subspace mask, `del_pairs` distinct pairs,
uniform + normal jitter, gamma = 1 jumps every 5th generation) and the adaptation of the
crossover probabilities `p_cr` during burn-in; all of it runs in the HIP update kernel
(bipymc_amd/csrc/kernels.h) and the once-per-generation CR reduction (level 1 inside the burn-in flavours of the update kernel or
`cr_level1_kernel`, `cr_mid_kernel` beyond 512 partial sums, `cr_final_kernel`).
"""
from __future__ import division, print_function

import numpy as np

from . import _lib as L
from .demc import DeMcMpi


class DreamMpi(DeMcMpi):
    _ALGO = L.ALGO_DREAM

    def __init__(self, ln_like_fn, theta_0=None, varepsilon=1e-6, n_chains=8,
                 mpi_comm=None, ln_kwargs={}, **kwargs):
        self.gamma_scale = kwargs.get("gamma_scale", 1.0)        # dream.py:20
        self.del_pairs = kwargs.get("del_pairs", 3)              # dream.py:22
        self.burnin_gen = kwargs.get("burnin_gen", 300)          # dream.py:24
        self.p_cr_update_gen = kwargs.get("n_cr_gen", 50)        # dream.py:26
        self.n_cr = kwargs.get("n_cr", 3)                        # dream.py:27
        self.CR = (np.array(range(self.n_cr)) + 1) / self.n_cr   # dream.py:113
        super(DreamMpi, self).__init__(ln_like_fn, theta_0=theta_0, varepsilon=varepsilon, n_chains=n_chains,
                                       mpi_comm=mpi_comm, ln_kwargs=ln_kwargs, **kwargs)

    def _engine_kwargs(self, kwargs):
        return dict(gamma_scale=self.gamma_scale, del_pairs=self.del_pairs, burnin_gen=self.burnin_gen,
                    n_cr_gen=self.p_cr_update_gen, n_cr=self.n_cr)

    # ---- crossover statistics live on the device (dream.py:109-140) -------------
    @property
    def p_cr(self):
        return self._engine.stats()["p_cr"]

    @property
    def delta_m(self):
        return self._engine.stats()["delta_m"]

    @property
    def n_cr_updates(self):
        return self._engine.stats()["n_cr_updates"]

    @property
    def p_cr_update(self):
        return self.p_cr                                         # dream.py:137 aliases the two

    @property
    def in_burnin(self):
        return True                                              # dream.py:142-144

    def _adapt_state(self):
        st = self._engine.stats()
        return dict(t_abs=int(st["t_abs"]), seed=self.seed, p_cr=st["p_cr"], delta_m=st["delta_m"],
                    n_cr_updates=st["n_cr_updates"])

    def _restore_adapt_state(self, adapt):
        if adapt and "p_cr" in adapt and len(adapt["p_cr"]) == self.n_cr:
            self._engine.set_adapt_state(adapt["p_cr"], adapt["delta_m"], adapt["n_cr_updates"],
                                         int(adapt.get("t_abs", -1)))
        elif adapt and "t_abs" in adapt:
            self._engine.set_adapt_state(t_abs=int(adapt["t_abs"]))
