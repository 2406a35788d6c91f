"""Parallel DE-MC sampler on MI355X -- drop-in for `bipymc.demc.DeMcMpi`
(reference: bipymc/demc.py:10-338; base classes bipymc/samplers.py:10-84, 237-336).

Same constructor, `run_mcmc(n, **kwargs)`, `param_est(n_burn)`, `acceptance_fraction`,
`am_chains`, `save_state` / `load_state`, `super_chain_mpi`, `gather_all_chains`,
`iter_local_chains`, `iter_all_chains`, `get_chain`, `get_chain_rank`.  The generation loop
(demc.py:79-140) runs inside libbipymc_hip.so (include/bipymc_hip.h): this class only
translates arguments, drives `bpm_step`, and assembles results.

Differences a caller can see (see DESIGN.md "Deviations"):
  * randomness comes from counter-based Philox streams keyed by `seed=` (default: one
    draw from `np.random`, so `np.random.seed(42)` before construction still pins a run);
  * `mpi_comm` may be None (single GPU), "torch" / a torch.distributed group, or an
    mpi4py-style communicator; one process drives one GPU;
  * a `ln_like_fn` that is the bound `ln_like` of a shipped target object
    (bipymc_amd.utils.*) is evaluated on the GPU; any other callable is called on the host
    once per chain update with a 1-D float64 array, exactly like the reference.
"""
from __future__ import division, print_function

import numpy as np

from . import _lib as L
from . import comm as _comm
from .chain import DetachedChain, McmcChain
from .utils import _target


def _visible_device_count():
    """hipGetDeviceCount without creating a context (bpm_device_count)."""
    try:
        return int(L.device_count())
    except Exception:
        return 0


def _hostname():
    import socket
    return socket.gethostname()


def _default_engine_factory(**kw):
    from .engine import HipEngine
    return HipEngine(**kw)


# The engine behind the sampler classes.  Not a constructor argument: the product has one engine (the HIP library);
# the CPU tests replace this module attribute with the oracle engine (tests/_oracle_engine.py) to exercise the host logic.
_engine_factory = _default_engine_factory


class DeMcMpi(object):
    """!
    @brief DE-MC population sampler, one process per MI355X.
    """
    _ALGO = L.ALGO_DEMC

    def __init__(self, ln_like_fn, theta_0=None, varepsilon=1e-6, n_chains=8,
                 mpi_comm=None, ln_kwargs={}, **kwargs):
        assert n_chains >= 4                                   # samplers.py:249
        varepsilon = np.asarray(varepsilon, dtype=np.float64)
        self.n_chains = int(n_chains)
        self.comm = _comm.wrap(mpi_comm)
        self.local_n_accepted = 0
        self.local_n_rejected = 1                              # demc.py:18
        self.n_accepted = 1                                    # samplers.py:30-31
        self.n_rejected = 0
        if theta_0 is not None:
            self.dim = len(np.asarray(theta_0).reshape(-1))
        else:
            self.dim = int(kwargs.get("dim", 1))
        self.h5_file = kwargs.get("h5_file", "sampler_checkpoint.h5")
        self.warm_start = kwargs.get("warm_start", False)
        self.checkpoint = kwargs.get("checkpoint", 0)
        self.log_like_fn = ln_like_fn
        # how ln_like_fn is called when it is NOT one of the shipped device targets (samplers.py:36-43 calls it row by row):
        #   vectorized=False   one Python call per proposal row, like the reference
        #   vectorized=True    one call per block of rows: ln_like_fn((n, dim) float64 ndarray, **ln_kwargs) -> n values
        #   vectorized="device"  the block stays on the GPU: ln_like_fn is handed an object with `__cuda_array_interface__` (engine.DeviceRows:
        #                      `torch.as_tensor(rows, device="cuda")` wraps it without a copy) and returns n float64 values in device memory
        #                      (a torch / cupy array) -- nothing crosses PCIe (include/bipymc_hip.h: bpm_propose_device / bpm_commit_device)
        # callback_chunks > 1: the half generation's proposals are handed to ln_like_fn in that many pieces, the DMA of piece k + 1 under the evaluation
        # of piece k (vectorized=True: one call per piece); default: one call per half generation
        #   ln_like_fn = HipLikelihood(source, params)  (device_likelihood.py): compiled into a kernel of the generation loop -- no callback at all
        vec = kwargs.get("vectorized", False)
        self._device_callback = isinstance(vec, str) and vec == "device"
        if isinstance(vec, str) and not self._device_callback:
            raise ValueError("vectorized must be False, True or \"device\"")
        self._vectorized = bool(vec)
        self._callback_chunks = kwargs.get("callback_chunks", None)
        # callback_threads > 1 (vectorized=True only; ln_like_fn must then be thread-safe): the pieces of a half generation are evaluated by a pool of
        # host threads -- NumPy releases the interpreter lock inside its loops, and ONE core reading 3 MB of freshly DMA'd proposals is what bounds
        # the host path at cfg2's shape (bench.py: host_callback_config)
        self._callback_threads = int(kwargs.get("callback_threads", 1))
        self._pool = None
        self._ln_kwargs = dict(ln_kwargs)
        self._freeze_ln_like_fn(**self._ln_kwargs)
        if self.n_chains % self.comm.size != 0:
            raise ValueError("n_chains must be a multiple of the communicator size "
                             "(unequal blocks break the reference's Allgather too, demc.py:39,93)")
        # ---- device engine --------------------------------------------
        seed = kwargs.get("seed", None)
        if seed is None:
            seed = int(np.random.randint(0, 2 ** 62)) if self.comm.rank == 0 else None
        self.seed = int(self.comm.bcast(seed, root=0))
        # which ln_like_fn runs on the device: the shipped targets, the reference's own target objects (recognised and
        # verified, utils/_target.py), else the host callback of samplers.py:36-43
        self._target_id, self._target_params, self.target_rule = _target.resolve_info(ln_like_fn, self._ln_kwargs, self.dim)
        if kwargs.get("force_host_callback", False):
            self._target_id, self._target_params, self.target_rule = L.TARGET_HOST_CALLBACK, None, "host-callback"
        factory = _engine_factory
        uid = None
        # How the ranks exchange state where the reference calls comm.Allgather twice per generation (demc.py:93-94,116-117):
        # exchange="auto" (default): an RCCL communicator is created AND the ranks' buffers are mapped into each other; when the
        # mapping works on every rank, owners push accepted rows straight into the peers' replicas ("push"), else accept bytes
        # travel through RCCL and are replayed ("replay").  "push": no RCCL communicator at all (ranks sharing one GPU, nodes
        # without RCCL).  "replay" / "rows" / "dense": the RCCL exchanges of include/bipymc_hip.h.
        self.exchange = kwargs.get("exchange", "auto")
        if self.exchange not in ("auto", "push", "replay", "rows", "dense"):
            raise ValueError("exchange must be one of auto, push, replay, rows, dense")
        if self.comm.size > 1:
            L.want_dmabuf_ipc()            # (multi-rank only; a no-op when the launcher exported the variable, as bench.py's does)
            if self.comm.rank == 0:
                from .engine import HipEngine
                if factory is not _default_engine_factory:
                    uid = b"\0" * L.UID_BYTES
                else:
                    uid = HipEngine.push_uid() if self.exchange == "push" else HipEngine.unique_id()
            uid = self.comm.bcast(uid, root=0)
        self._engine = factory(
            algo=self._ALGO, n_chains=self.n_chains, dim=self.dim, target_id=self._target_id,
            target_params=self._target_params, seed=self.seed,
            device=kwargs.get("device", self._default_device()), rank=self.comm.rank, world_size=self.comm.size,
            nccl_uid=uid, p_snooker=kwargs.get("p_snooker", 0.0), outlier_every=kwargs.get("outlier_every", 0),
            keep_history=kwargs.get("keep_history", True),
            # without a history param_est_moments answers from per-generation population sums (whole-generation burn-ins)
            running_moments=kwargs.get("running_moments", not kwargs.get("keep_history", True)), **self._engine_kwargs(kwargs))
        self.n_local = self.n_chains // self.comm.size
        # ln_like_fn given as HIP source: a host-callback target whose likelihood is a kernel between the proposal and the commit kernel
        # (include/bipymc_hip.h: bpm_set_device_likelihood); an engine without the entry point (the CPU test engine) calls its python_fn
        from .device_likelihood import HipLikelihood
        self._hip_likelihood = None
        if isinstance(ln_like_fn, HipLikelihood) and not self.uses_device_target:
            if hasattr(self._engine, "set_device_likelihood"):
                self._engine.set_device_likelihood(ln_like_fn.source, ln_like_fn.params)
                self._hip_likelihood = ln_like_fn
            elif ln_like_fn.python_fn is None:
                raise TypeError("HipLikelihood without python_fn on an engine that cannot compile it")
        self._connect_exchange()
        self._hist_cache = None
        self._hist_cache_rows = -1
        self.am_chains = []
        if not self.warm_start:
            self.init_chains(theta_0, varepsilon, **kwargs)
        else:
            self.init_warmstart_chain(self.h5_file)

    # ---- hooks for DreamMpi --------------------------------------------
    def _engine_kwargs(self, kwargs):
        return {}

    def _connect_exchange(self):
        """world > 1: every rank publishes what the others need to map its exchange buffer (bpm_push_export), the communicator
        moves the blobs (the one collective the reference's constructor has no counterpart for), every rank maps its peers
        (bpm_push_connect) and the connection is tested (bpm_push_selftest).  The decision is collective: push only when it
        works on EVERY rank; an engine without the entry points (host-callback target, the CPU test engine) skips it."""
        eng = self._engine
        self.exchange_used = None
        if self.comm.size == 1 or not hasattr(eng, "push_export"):
            return
        if self.exchange in ("auto", "push") and self.uses_device_target:
            err = None
            try:
                blob = eng.push_export()
            except Exception as e:                                    # noqa: BLE001 -- reported through the collective below
                blob, err = None, str(e)
            blobs = self.comm.allgather((blob, err))
            ok = all(b[0] is not None for b in blobs)
            if ok:
                try:
                    eng.push_connect([b[0] for b in blobs])
                except Exception as e:                                # noqa: BLE001
                    ok, err = False, str(e)
            oks = self.comm.allgather((ok, err))
            ok = all(o[0] for o in oks)
            if ok:
                self.comm.Barrier()
                try:                                                  # (a rank that raises here must still enter the collective below)
                    mine = bool(eng.push_selftest())
                except Exception as e:                                # noqa: BLE001
                    mine, err = False, "self-test: %s" % e
                oks = self.comm.allgather((mine, err))
                ok = all(o[0] for o in oks)
            if ok:
                eng.set_exchange("push")
                self.exchange_used = "push"
                return
            why = "; ".join("rank %d: %s" % (i, o[1]) for i, o in enumerate(oks) if o[1]) or "the connection self-test failed"
            if self.exchange == "push":
                raise RuntimeError("the push exchange could not be connected: " + why)
            import warnings
            # rows wider than 512 coordinates run on the looped kernel, which has no replay / packed-rows form (bpm_set_exchange refuses both):
            # they degrade to the dense all-gather -- the reference's own MPI_Allgather (demc.py:93-94,116-117).  The decision depends on dim
            # alone: the same on every rank.
            fallback = "dense" if self.dim > 512 else "replay"
            warnings.warn("bipymc_amd: push exchange not available (%s); %s instead" % (
                why, "whole blocks travel through the RCCL all-gather" if fallback == "dense" else "accept bytes travel through RCCL"))
            eng.set_exchange(fallback)
            self.exchange_used = fallback
            return
        if self.uses_device_target:
            eng.set_exchange(self.exchange)
            self.exchange_used = self.exchange

    def _default_device(self):
        """One process per GPU: the node-local rank as the launcher exports it (torchrun: LOCAL_RANK; Open MPI, MVAPICH,
        Slurm), else rank modulo the number of visible devices.  A launcher that already narrowed the visible devices to
        one per process (ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES) gets device 0.  EVERY rank enters the one collective
        of this function, whatever it found locally, and all ranks raise together afterwards: a mis-launch is an error
        message on every rank, not a hang of the well-configured ones in an allgather (the reference's ranks share the
        host's cores and pick nothing, demc.py:15)."""
        import os
        if self.comm.size == 1:
            return 0
        local = None
        for var in ("LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK", "MV2_COMM_WORLD_LOCAL_RANK", "MPI_LOCALRANKID", "SLURM_LOCALID"):
            if os.environ.get(var, "") != "":
                local = int(os.environ[var])
                break
        ndev = _visible_device_count()
        if local is None:
            local = self.comm.rank % max(ndev, 1)
        error = None
        if ndev <= 1:          # one device exposed per process (or none at all: bpm_create will say so)
            device = 0
        elif local >= ndev:
            device = -1
            error = "rank %d: local rank %d but only %d visible GPU(s): launch one process per GPU" % (self.comm.rank, local, ndev)
        else:
            device = local
        host = _hostname()
        seen = self.comm.allgather((host, device, ndev, error))
        errors = [e[3] for e in seen if e[3]]
        # two ranks of one node on the same device of a multi-device view would deadlock in ncclCommInitRank: say so instead
        multi = [(e[0], e[1]) for e in seen if e[2] > 1 and e[1] >= 0]
        dup = sorted(set(k for k in multi if multi.count(k) > 1))
        if dup:
            errors.append("several ranks resolved to the same GPU: %s; set LOCAL_RANK (or pass device=)"
                          % ", ".join("host %s GPU %d" % k for k in dup))
        if errors:
            raise RuntimeError("; ".join(errors))
        return device

    # ---- samplers.py:36-47 -----------------------------------------------
    def _freeze_ln_like_fn(self, **kwargs):
        self._frozen_ln_like_fn = lambda theta: self.log_like_fn(theta, **kwargs)

    @property
    def frozen_ln_like_fn(self):
        return self._frozen_ln_like_fn

    @property
    def uses_device_target(self):
        return self._target_id != L.TARGET_HOST_CALLBACK

    # ---- chains ------------------------------------------------------------
    def init_chains(self, theta_0, varepsilon=1e-6, **kwargs):
        """demc.py:34-44: contiguous id blocks per rank; chain.py:25-27 jitter on the device."""
        rank_chain_ids = np.array_split(np.array(range(self.n_chains)), self.comm.size)[self.comm.rank]
        self.rank_chain_ids = rank_chain_ids
        if theta_0 is None:
            theta_0 = np.zeros(self.dim)
        self._engine.init_chains(theta_0, np.asarray(varepsilon, dtype=np.float64))
        self._after_state_reset()

    def init_warmstart_chain(self, h5_file):
        """demc.py:46-51."""
        self.init_chains(np.zeros(self.dim), 0.0)
        self.load_state(h5_file)

    def _after_state_reset(self):
        self.am_chains = [McmcChain(self, int(c), i) for i, c in enumerate(self.rank_chain_ids)]
        self._hist_cache = None
        self._hist_cache_rows = -1
        if not self.uses_device_target:
            if self._hip_likelihood is not None:
                self._engine.refresh_device_loglike()
                return
            if self._device_callback:
                self._engine.set_loglike_device(self.log_like_fn(self._engine.state_device(), **self._ln_kwargs))
                return
            X = self._engine.get_state()
            lo = self.comm.rank * self.n_local
            self._engine.set_loglike(self._eval_ln_like(X[lo:lo + self.n_local]))

    def _call_ln_like(self, theta):
        v = self._frozen_ln_like_fn(np.array(theta, dtype=np.float64))
        return float(np.asarray(v).reshape(-1)[0]) if np.ndim(v) else float(v)

    def _get_local_chain_state(self):
        """demc.py:53-57."""
        lo = self.comm.rank * self.n_local
        return self._engine.get_state()[lo:lo + self.n_local]

    def _local_history(self):
        rows = self._engine.history_rows()
        if self._hist_cache is None or rows != self._hist_cache_rows:
            self._hist_cache = self._engine.get_history(0, rows)
            self._hist_cache_rows = rows
        return self._hist_cache

    # ---- the run ---------------------------------------------------------------
    def run_mcmc(self, n, **kwargs):
        self._mcmc_run(n, **kwargs)

    def _n_generations(self, n):
        """Iterations of `while j < int((n - n_chains) / size)` (demc.py:79): j grows by the
        number of local chains per generation."""
        target = int((n - self.n_chains) / self.comm.size)
        return max(0, -(-target // self.n_local))

    def _run_opts(self, kwargs):
        return dict(flip=kwargs.get("flip", 0.5), shuffle=kwargs.get("shuffle", True),
                    epsilon=kwargs.get("epsilon", None), u_epsilon=kwargs.get("u_epsilon", None),
                    gamma=kwargs.get("gamma", None))

    def _mcmc_run(self, n, **kwargs):
        if not self.am_chains:
            raise RuntimeError("ERROR: chains not initilized")       # demc.py:65-66
        n_gens = self._n_generations(n)
        eng = self._engine
        self.comm.Barrier()            # ranks enter the run together (the push exchange bounds how long a rank waits for its peers)
        eng.begin_run(**self._run_opts(kwargs))
        eng.reserve_history(eng.history_rows() + n_gens)
        done = 0
        chunk = int(self.checkpoint) if self.checkpoint and self.checkpoint > 0 else n_gens
        while done < n_gens:
            todo = min(chunk, n_gens - done)
            if self.uses_device_target or self._hip_likelihood is not None:
                eng.step(todo)
            else:
                for _ in range(todo):
                    self._host_generation()
            done += todo
            if self.checkpoint and self.checkpoint > 0 and done % self.checkpoint == 0:
                self.save_state(self.h5_file)                        # demc.py:138-140
        eng.synchronize()
        st = eng.stats()
        self._last_stats = st
        if st["n_nan_alpha"] > 0:
            # np.random.choice(p=[nan, nan]) in the reference (samplers.py:336)
            raise ValueError("probabilities contain NaN")
        self.local_n_accepted = int(st["local_n_accepted"])
        self.local_n_rejected = int(st["local_n_rejected"])
        counts = self.comm.allgather((self.local_n_accepted, self.local_n_rejected))   # demc.py:143-150
        self.n_accepted = int(sum(c[0] for c in counts))
        self.n_rejected = int(sum(c[1] for c in counts))
        self._hist_cache = None
        self.comm.Barrier()

    def _host_generation(self):
        """One generation with a Python ln_like_fn: two propose/commit half generations (demc.py:103-109,126-132 call ln_like through
        _mut_prop_ratio, samplers.py:328-332).  The proposals come back in pieces, the DMA of the next piece under the evaluation of this one;
        with vectorized="device" they never leave the GPU."""
        eng = self._engine
        if self._device_callback:
            for _ in range(2):
                eng.commit_device(self.log_like_fn(eng.propose_device(), **self._ln_kwargs))
            return
        # Default: the one-call form (bpm_propose / bpm_commit; the library overlaps the read-back with its own compaction copy).  The piecewise form
        # pays off when the callback is expensive per row; with a cheap vectorised NumPy likelihood it measured no gain at cfg2's shape (the evaluation
        # then reads freshly DMA'd memory itself, and every piece costs a few Python calls: profiles/r05_host_callback.txt)
        chunks = self._callback_chunks
        if (chunks is None or chunks <= 1) and self._callback_threads <= 1 and hasattr(eng, "propose"):
            for _ in range(2):
                props, _ids = eng.propose()
                eng.commit(self._eval_ln_like(props))
            return
        chunks = 1 if chunks is None else chunks
        def piece(rows, ids):
            if len(ids) and ids.min() < 0:                           # idle work items of a rank of a world: no call, any value
                act = ids >= 0
                ll = np.zeros(len(ids))
                ll[act] = self._eval_ln_like(rows[act])
                return ll
            return self._eval_ln_like(rows)
        if self._vectorized and self._callback_threads > 1 and hasattr(eng, "propose_begin"):
            if self._pool is None:
                from concurrent.futures import ThreadPoolExecutor
                self._pool = ThreadPoolExecutor(max_workers=self._callback_threads)
            chunks = max(chunks, self._callback_threads)
            for _ in range(2):
                eng.propose_begin(chunks)
                futs = [self._pool.submit(lambda k=k: piece(*eng.propose_chunk(k))) for k in range(chunks)]
                for k, f in enumerate(futs):
                    eng.commit_chunk(k, f.result())                  # (the library is called from this thread only once the values exist)
                eng.commit_end()
            return
        for _ in range(2):
            for k, rows, ids in eng.propose_chunks(chunks):
                eng.commit_chunk(k, piece(rows, ids))
            eng.commit_end()

    def _eval_ln_like(self, thetas):
        """ln_like_fn on a block of rows: one Python call per row like the reference (samplers.py:36-43), or
        -- opt-in `vectorized=True` -- one call on the whole (n, dim) block returning n values."""
        if self._vectorized and len(thetas):
            # (the block is handed over as it lies in the library's pinned staging when it is contiguous -- even dim --: a view, valid during the call)
            block = thetas if isinstance(thetas, np.ndarray) and thetas.dtype == np.float64 and thetas.flags["C_CONTIGUOUS"] else np.array(thetas, dtype=np.float64)
            v = np.asarray(self.log_like_fn(block, **self._ln_kwargs), dtype=np.float64)
            return v.reshape(len(thetas))
        return np.array([self._call_ln_like(p) for p in thetas], dtype=np.float64)

    @property
    def acceptance_fraction(self):
        """samplers.py:75-80."""
        return self.n_accepted / (self.n_accepted + self.n_rejected)

    # ---- results -----------------------------------------------------------------
    def param_est(self, n_burn, collection_rank=0):
        """demc.py:235-248."""
        self.comm.Barrier()
        chain_slice = self.super_chain_mpi(collection_rank)
        if self.comm.rank == collection_rank:
            chain_slice = chain_slice[n_burn:, :]
            mean_theta = np.mean(chain_slice, axis=0)
            std_theta = np.std(chain_slice, axis=0)
            return mean_theta, std_theta, chain_slice
        return None, None, None

    def param_est_moments(self, n_burn):
        """mean and std (ddof=0) of the super-chain rows >= n_burn, as `param_est` computes them (demc.py:242-246),
        reduced on the GPU(s) without moving the history to the host; identical on every rank."""
        cnt, s1, s2, sh = self._engine.reduce_moments(int(n_burn))
        parts = self.comm.allgather((cnt, s1, s2))
        n = float(sum(p[0] for p in parts))
        S1 = np.sum([p[1] for p in parts], axis=0)
        S2 = np.sum([p[2] for p in parts], axis=0)
        return sh + S1 / n, np.sqrt(np.maximum(S2 / n - (S1 / n) ** 2, 0.0))

    def super_chain_mpi(self, collection_rank=0):
        return self._super_chain(collection_rank)

    def _super_chain(self, collection_rank=0):
        """demc.py:260-270: row g*n_chains + i = chain i at generation g.  The device history is
        already generation-major, so this is a concatenation along the chain axis."""
        local = self._local_history()                               # (T, n_local, d)
        if self.comm.size == 1:
            return local.reshape(-1, self.dim).copy()
        parts = self.comm.allgather(local)
        if self.comm.rank != collection_rank:
            return None
        return np.concatenate(parts, axis=1).reshape(-1, self.dim)

    def gather_all_chains(self, collection_rank=0):
        return list(self.iter_all_chains(collection_rank))

    def iter_local_chains(self):
        for chain in self.am_chains:
            yield chain

    def iter_all_chains(self, collection_rank=0, verbose=0):
        if self.comm.size == 1:
            for chain in self.am_chains:
                yield chain
            return
        parts = self.comm.allgather(self._local_history())
        if self.comm.rank == collection_rank:
            full = np.concatenate(parts, axis=1)
            for c_id in range(self.n_chains):
                yield DetachedChain(c_id, full[:, c_id, :])
        else:
            for c_id in range(self.n_chains):
                yield None

    def get_chain(self, c_id, collection_rank=0, verbose=0):
        """demc.py:296-325.  Which branch is taken depends only on (c_id, collection_rank), never on the calling rank: either
        nobody communicates (the owner IS the collection rank: it returns its own chain, everyone else None) or every rank
        enters the same collective."""
        assert 0 <= c_id < self.n_chains
        r = self.get_chain_rank(c_id)
        if self.comm.size == 1:
            return self.am_chains[c_id]
        if r == collection_rank:                                   # demc.py:301-304: no communication at all
            return self.am_chains[c_id - r * self.n_local] if self.comm.rank == r else None
        parts = self.comm.allgather(self._local_history()[:, c_id - r * self.n_local, :]
                                    if self.comm.rank == r else None)
        if self.comm.rank == collection_rank:
            return DetachedChain(c_id, parts[r])
        return None

    def get_chain_rank(self, c_id):
        """demc.py:327-338."""
        assert 0 <= c_id < self.n_chains
        return int(c_id // self.n_local)

    # ---- checkpoint (demc.py:198-233; chain.py:59-93) --------------------------------
    def save_state(self, h5_file=""):
        from . import checkpoint
        if not h5_file:
            h5_file = self.h5_file
        full = self._super_chain(0)
        if self.comm.rank == 0:
            T = full.shape[0] // self.n_chains
            checkpoint.write(h5_file, full.reshape(T, self.n_chains, self.dim), self._adapt_state())
        self.comm.Barrier()

    def load_state(self, h5_file=""):
        from . import checkpoint
        if not h5_file:
            h5_file = self.h5_file
        hist, adapt = checkpoint.read(h5_file, self.n_chains, self.dim)   # (T, N, d)
        lo = self.comm.rank * self.n_local
        self._engine.set_history(hist[:, lo:lo + self.n_local, :], hist[-1])
        self._after_state_reset()
        self._restore_adapt_state(adapt)
        self.comm.Barrier()

    def _adapt_state(self):
        st = self._engine.stats()
        return dict(t_abs=int(st["t_abs"]), seed=self.seed)

    def _restore_adapt_state(self, adapt):
        if adapt and "t_abs" in adapt:
            self._engine.set_adapt_state(t_abs=int(adapt["t_abs"]))
