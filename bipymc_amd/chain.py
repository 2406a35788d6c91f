"""Per-chain view of the device-resident history (reference: bipymc/chain.py:9-134).

The reference keeps one growing (T, dim) ndarray per chain and re-allocates it on every
append (chain.py:51-54).  Here the history lives on the GPU as one (T, n_local, dim) block,
appended by the update kernel; `McmcChain` is a lazy view of column `global_id`.
"""
import numpy as np


class McmcChain(object):
    def __init__(self, sampler, global_id, local_index):
        assert isinstance(global_id, int) and global_id >= 0     # chain.py:21-23
        self._sampler = sampler
        self.global_id = global_id
        self._li = local_index
        self._dim = sampler.dim
        self._override = None

    @property
    def chain(self):
        """(T, dim) history of this chain, row 0 = initial state (chain.py:113-115)."""
        if self._override is not None:
            return self._override
        return self._sampler._local_history()[:, self._li, :]

    @chain.setter
    def chain(self, input_chain):
        input_chain = np.asarray(input_chain, dtype=np.float64)
        assert input_chain.shape[1] == self._dim                 # chain.py:119
        self._override = input_chain

    def load_chain_state(self, chain_state):
        self.chain = chain_state                                 # chain.py:95-96

    def append_sample(self, theta_new):
        """chain.py:51-54.  Samples are appended by the update kernel on the device; a host-side append would
        fork this chain's history from the sampler's, so it is refused rather than silently diverging."""
        raise NotImplementedError("bipymc_amd: chain histories are appended on the GPU by run_mcmc(); "
                                  "use load_chain_state() to replace a chain's host-side view")

    def auto_corr(self, lag):
        pass                                                     # chain.py:98-102: a stub in the reference as well

    def __getitem__(self, get_index):
        if isinstance(get_index, slice):
            return self.chain[get_index]
        return self.chain[get_index, :]

    @property
    def current_pos(self):
        return self.chain[-1, :]                                 # chain.py:122-124

    @property
    def chain_len(self):
        return self.chain.shape[0]

    @property
    def dim(self):
        return self._dim


class DetachedChain(object):
    """A chain copied out of its sampler (what the reference pickles to the collection rank,
    demc.py:296-325)."""

    def __init__(self, global_id, chain):
        self.global_id = int(global_id)
        self.chain = np.asarray(chain, dtype=np.float64)

    @property
    def current_pos(self):
        return self.chain[-1, :]

    @property
    def chain_len(self):
        return self.chain.shape[0]

    @property
    def dim(self):
        return self.chain.shape[1]
