"""Thin object wrapper over the C ABI: one `HipEngine` = one `bpm_handle_t`.

Mirrors, call for call, what DeMcMpi/DreamMpi need from the device: construction
(demc.py:14-32, dream.py:17-30), chain initialisation (chain.py:25-27), the
generation loop (demc.py:63-151), history (chain.py:51-54) and statistics.
"""
import ctypes as C

import numpy as np

from . import _lib as L


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _iptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


_dp_type = C.POINTER(C.c_double)
_ip_type = C.POINTER(C.c_int32)


class DeviceRows(object):
    """n rows of dim float64 coordinates in DEVICE memory (row stride ld doubles), read-only: what a device-resident likelihood is handed.  Exposes
    the CUDA array interface, which PyTorch-ROCm and CuPy-ROCm consume without a copy: `torch.as_tensor(rows, device="cuda")`."""

    def __init__(self, ptr, n, dim, ld, ids_ptr):
        self.ptr, self.n, self.dim, self.ld, self._ids_ptr = int(ptr), int(n), int(dim), int(ld), int(ids_ptr)
        self.shape = (self.n, self.dim)

    @property
    def __cuda_array_interface__(self):
        # (read-only by contract; PyTorch refuses the interface's read-only flag, so it is not set)
        return dict(shape=(self.n, self.dim), typestr="<f8", data=(self.ptr, False), strides=(self.ld * 8, 8), version=3)

    @property
    def ids(self):
        return _DeviceVector(self._ids_ptr, self.n, "<i4")

    def __len__(self):
        return self.n


class _DeviceVector(object):
    def __init__(self, ptr, n, typestr):
        self.ptr, self.n, self.typestr = int(ptr), int(n), typestr

    @property
    def __cuda_array_interface__(self):
        return dict(shape=(self.n,), typestr=self.typestr, data=(self.ptr, False), strides=None, version=3)


def _device_pointer(obj, n, what):
    """device address of a contiguous float64 vector of n values: anything with `__cuda_array_interface__` (torch / cupy arrays) or data_ptr()"""
    cai = getattr(obj, "__cuda_array_interface__", None)
    if cai is not None:
        shape = tuple(cai["shape"])
        size = int(np.prod(shape)) if shape else 1
        if cai["typestr"] not in ("<f8", "=f8", "|f8"):
            raise TypeError("%s must be float64 on the device (got %s)" % (what, cai["typestr"]))
        if cai.get("strides") is not None and len(shape) == 1 and tuple(cai["strides"]) != (8,):
            raise ValueError("%s must be contiguous" % what)
        if len(shape) > 1 and size != max(shape):
            raise ValueError("%s must be one value per row (got shape %s)" % (what, shape))
        if n is not None and size != n:
            raise ValueError("%s: %d values for %d rows" % (what, size, n))
        return int(cai["data"][0])
    if hasattr(obj, "data_ptr"):
        return int(obj.data_ptr())
    raise TypeError("%s must live in device memory: an object with __cuda_array_interface__ (torch.Tensor on the GPU, cupy.ndarray)" % what)


class HipEngine(object):
    def __init__(self, algo, n_chains, dim, target_id, target_params, seed, device=0, rank=0, world_size=1,
                 nccl_uid=None, gamma_scale=1.0, del_pairs=3, burnin_gen=300, n_cr_gen=50, n_cr=3,
                 p_snooker=0.0, outlier_every=0, keep_history=True, running_moments=False, lib=None):
        """lib: the ctypes library to run on -- default the product library; tests that need the hooks of include/bipymc_hip_test.h pass
        `_lib.load_test()` (the same sources built with -DBPM_TEST_HOOKS)."""
        self._h = C.c_void_p()
        self.lib = lib if lib is not None else L.load()
        self.n_chains, self.dim = int(n_chains), int(dim)
        self.rank, self.world_size = int(rank), int(world_size)
        if self.n_chains % self.world_size != 0:
            raise ValueError("n_chains must be divisible by the communicator size")
        self.n_local = self.n_chains // self.world_size
        self.lo = self.rank * self.n_local
        tp = np.ascontiguousarray(target_params if target_params is not None else [], dtype=np.float64)
        cfg = L.BpmConfig()
        cfg.abi_version = L.ABI_VERSION
        cfg.algo = int(algo)
        cfg.n_chains = self.n_chains
        cfg.dim = self.dim
        cfg.target_id = int(target_id)
        cfg.n_target_params = int(tp.size)
        cfg.target_params = _dptr(tp) if tp.size else None
        cfg.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        cfg.device = int(device)
        cfg.rank = self.rank
        cfg.world_size = self.world_size
        cfg.nccl_uid = bytes(nccl_uid) if nccl_uid is not None else None
        cfg.gamma_scale = float(gamma_scale)
        cfg.del_pairs = int(del_pairs)
        cfg.burnin_gen = int(burnin_gen)
        cfg.n_cr_gen = int(n_cr_gen)
        cfg.n_cr = int(n_cr)
        cfg.p_snooker = float(p_snooker)
        cfg.outlier_every = int(outlier_every)
        cfg.keep_history = 1 if keep_history else 0
        cfg.running_moments = 1 if running_moments else 0
        self._ck(self.lib.bpm_create(C.byref(cfg), C.byref(self._h)))
        self.algo, self.target_id = int(algo), int(target_id)
        self.n_cr = int(n_cr) if algo == L.ALGO_DREAM else 1

    def _ck(self, rc):
        L.check(rc, self.lib)

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(L.UID_BYTES)
        L.check(L.load().bpm_get_unique_id(buf))
        return buf.raw

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            h, self._h = self._h, C.c_void_p()
            self._ck(self.lib.bpm_destroy(h))         # non-zero: the queue failed and buffers were leaked (include/bipymc_hip.h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- state ------------------------------------------------------
    def init_chains(self, theta_0, varepsilon):
        t0 = np.ascontiguousarray(np.asarray(theta_0, dtype=np.float64).reshape(-1))
        if t0.size != self.dim:
            raise ValueError("theta_0 has %d entries, dim is %d" % (t0.size, self.dim))
        var = np.ascontiguousarray(np.broadcast_to(np.asarray(varepsilon, dtype=np.float64), (self.dim,)))
        self._ck(self.lib.bpm_init_chains(self._h, _dptr(t0), _dptr(var)))

    def set_state(self, X):
        X = np.ascontiguousarray(X, dtype=np.float64)
        if X.shape != (self.n_chains, self.dim):
            raise ValueError("state must have shape (n_chains, dim)")
        self._ck(self.lib.bpm_set_state(self._h, _dptr(X)))

    def get_state(self):
        X = np.empty((self.n_chains, self.dim), dtype=np.float64)
        self._ck(self.lib.bpm_get_state(self._h, _dptr(X)))
        return X

    def set_history(self, hist_local, X):
        hist_local = np.ascontiguousarray(hist_local, dtype=np.float64)
        X = np.ascontiguousarray(X, dtype=np.float64)
        assert hist_local.ndim == 3 and hist_local.shape[1:] == (self.n_local, self.dim)
        assert X.shape == (self.n_chains, self.dim)
        self._ck(self.lib.bpm_set_history(self._h, hist_local.shape[0], _dptr(hist_local), _dptr(X)))

    def set_loglike(self, ll_local):
        ll = np.ascontiguousarray(ll_local, dtype=np.float64)
        assert ll.shape == (self.n_local,)
        self._ck(self.lib.bpm_set_loglike(self._h, _dptr(ll)))

    def get_loglike(self):
        ll = np.empty(self.n_local, dtype=np.float64)
        self._ck(self.lib.bpm_get_loglike(self._h, _dptr(ll)))
        return ll

    # ---- running ----------------------------------------------------
    def begin_run(self, flip=0.5, shuffle=True, epsilon=None, u_epsilon=None, gamma=None):
        o = L.BpmRunOpts()
        o.flip = float(flip)
        o.shuffle = 1 if shuffle else 0
        o.epsilon = -1.0 if epsilon is None else float(epsilon)
        o.u_epsilon = -1.0 if u_epsilon is None else float(u_epsilon)
        o.gamma = -1.0 if gamma is None else float(gamma)
        self._ck(self.lib.bpm_begin_run(self._h, C.byref(o)))

    def step(self, n_gens):
        self._ck(self.lib.bpm_step(self._h, int(n_gens)))

    def step_timed(self, n_gens, read=True):
        """n_gens generations, synchronous, timed on the device -> (ms from the end of the first update launch to the end of the
        last, launches in that interval).  read=False returns nothing: fetch the figures with last_step_time() afterwards (reading
        the events costs tens of microseconds of host time)."""
        if not read:
            self._ck(self.lib.bpm_step_timed(self._h, int(n_gens), None, None))
            return None
        ms = C.c_float(0.0)
        n = C.c_int64(0)
        self._ck(self.lib.bpm_step_timed(self._h, int(n_gens), C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)

    def last_step_time(self):
        ms = C.c_float(0.0)
        n = C.c_int64(0)
        self._ck(self.lib.bpm_get_step_time(self._h, C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)

    def _need_hooks(self, name):
        if not hasattr(self.lib, name):
            raise L.BpmError("%s is part of the test surface (include/bipymc_hip_test.h): create the engine with lib=_lib.load_test()" % name)

    def step_profiled(self, n_gens):
        """-> (summed update-kernel time in ms, number of launches).  Test variant only."""
        self._need_hooks("bpm_step_profiled")
        ms = C.c_double(0.0)
        n = C.c_int64(0)
        self._ck(self.lib.bpm_step_profiled(self._h, int(n_gens), C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)

    def synchronize(self):
        self._ck(self.lib.bpm_synchronize(self._h))

    def propose(self, fresh=False):
        """-> (proposals (n, dim), global chain ids (n,)) of this rank's chains of the open half generation.  The arrays are views of two buffers the
        engine keeps (a fresh 3 MB NumPy array per half generation is 800 page faults under the library's copy: a fifth of the call at cfg2's shape):
        valid until the next propose(); fresh=True returns arrays of the caller's own."""
        if fresh or getattr(self, "_prop_host", None) is None:
            prop = np.empty((self.n_local, self.dim), dtype=np.float64)
            ids = np.empty(self.n_local, dtype=np.int32)
            if not fresh:
                self._prop_host, self._ids_host = prop, ids
        else:
            prop, ids = self._prop_host, self._ids_host
        n = C.c_int32(0)
        self._ck(self.lib.bpm_propose(self._h, _dptr(prop), _iptr(ids), C.byref(n)))
        return prop[:n.value], ids[:n.value]

    def commit(self, ll_prop):
        ll = np.ascontiguousarray(ll_prop, dtype=np.float64)
        self._ck(self.lib.bpm_commit(self._h, _dptr(ll) if ll.size else None))

    # ---- the same half generation, read-back overlapped with the caller's evaluation (bpm_propose_begin / _chunk, bpm_commit_chunk / _end) --------
    def propose_chunks(self, n_chunks):
        """Generator over the half generation's proposals in n_chunks pieces: yields (k, rows, ids) with rows an (n, dim) float64 VIEW into the
        library's pinned staging (valid until commit_end) and ids the global chain ids (-1 = idle work item); while the caller evaluates piece k
        the DMA of the following pieces proceeds.  Hand the values in with commit_chunk(k, ll) and finish with commit_end()."""
        self._ck(self.lib.bpm_propose_begin(self._h, int(n_chunks)))
        for k in range(int(n_chunks)):
            rows, ids = _dp_type(), _ip_type()
            n, ld = C.c_int32(0), C.c_int32(0)
            self._ck(self.lib.bpm_propose_chunk(self._h, k, C.byref(rows), C.byref(ids), C.byref(n), C.byref(ld)))
            if n.value == 0:
                yield k, np.empty((0, self.dim)), np.empty(0, dtype=np.int32)
                continue
            a = np.ctypeslib.as_array(rows, shape=(n.value, ld.value))[:, :self.dim]
            yield k, a, np.ctypeslib.as_array(ids, shape=(n.value,))

    def propose_begin(self, n_chunks):
        self._ck(self.lib.bpm_propose_begin(self._h, int(n_chunks)))

    def propose_chunk(self, k):
        """piece k of the open half generation (waits for its DMA only; callable from a worker thread) -> (rows view, ids view)"""
        rows, ids = _dp_type(), _ip_type()
        n, ld = C.c_int32(0), C.c_int32(0)
        self._ck(self.lib.bpm_propose_chunk(self._h, int(k), C.byref(rows), C.byref(ids), C.byref(n), C.byref(ld)))
        if n.value == 0:
            return np.empty((0, self.dim)), np.empty(0, dtype=np.int32)
        return np.ctypeslib.as_array(rows, shape=(n.value, ld.value))[:, :self.dim], np.ctypeslib.as_array(ids, shape=(n.value,))

    def commit_chunk(self, k, ll):
        ll = np.ascontiguousarray(ll, dtype=np.float64)
        self._ck(self.lib.bpm_commit_chunk(self._h, int(k), _dptr(ll) if ll.size else None))

    def commit_end(self):
        self._ck(self.lib.bpm_commit_end(self._h))

    # ---- ... and with the likelihood evaluated on the device by the caller's framework (bpm_propose_device / bpm_commit_device) --------------------
    def propose_device(self):
        """-> DeviceRows: the half generation's proposals where they lie in device memory (`__cuda_array_interface__`: torch.as_tensor / cupy.asarray
        take it without a copy), work-item order; .ids is the same for the global chain ids (-1 = idle work item)"""
        rows, ids = C.c_void_p(), C.c_void_p()
        n, ld = C.c_int32(0), C.c_int32(0)
        self._ck(self.lib.bpm_propose_device(self._h, C.byref(rows), C.byref(ids), C.byref(n), C.byref(ld)))
        return DeviceRows(rows.value or 0, n.value, self.dim, ld.value, ids.value or 0)

    def commit_device(self, ll):
        """ll: n float64 values in device memory, in the order of the rows (anything with `__cuda_array_interface__` or data_ptr())"""
        self._ck(self.lib.bpm_commit_device(self._h, C.c_void_p(_device_pointer(ll, self._pending_rows(), "ln_like values"))))

    def _pending_rows(self):
        return None

    def state_device(self):
        rows = C.c_void_p()
        n, ld = C.c_int32(0), C.c_int32(0)
        self._ck(self.lib.bpm_state_device(self._h, C.byref(rows), C.byref(n), C.byref(ld)))
        return DeviceRows(rows.value or 0, n.value, self.dim, ld.value, 0)

    def set_loglike_device(self, ll):
        self._ck(self.lib.bpm_set_loglike_device(self._h, C.c_void_p(_device_pointer(ll, self.n_local, "ln_like values"))))

    def set_device_likelihood(self, source, params=()):
        """ln_like_fn as HIP source (include/bipymc_hip.h: bpm_set_device_likelihood): compiled with hiprtc into a kernel that runs between the proposal
        and the commit kernel; `step` then drives this host-callback sampler.  The current states are evaluated at once."""
        p = np.ascontiguousarray(np.asarray(params, dtype=np.float64).reshape(-1))
        src = source.encode() if isinstance(source, str) else bytes(source)
        self._ck(self.lib.bpm_set_device_likelihood(self._h, src, p.ctypes.data_as(C.POINTER(C.c_double)) if p.size else None, int(p.size)))
        self.has_device_likelihood = True

    def device_likelihood_info(self):
        """-> (fused, why): fused True -- the update kernel itself was compiled around the likelihood (one launch per half generation); False -- the
        proposal / likelihood / commit kernels run, `why` says why the fused form is not in use"""
        fused = C.c_int32(0)
        why = C.create_string_buffer(1 << 14)
        self._ck(self.lib.bpm_get_device_likelihood_info(self._h, C.byref(fused), why, len(why)))
        return bool(fused.value), why.value.decode(errors="replace")

    def refresh_device_loglike(self):
        self._ck(self.lib.bpm_refresh_device_loglike(self._h))

    def reserve_history(self, rows):
        self._ck(self.lib.bpm_reserve_history(self._h, int(rows)))

    # ---- results ----------------------------------------------------
    def stats(self):
        st = L.BpmStats()
        self._ck(self.lib.bpm_get_stats(self._h, C.byref(st)))
        n = st.n_cr
        return dict(local_n_accepted=st.local_n_accepted, local_n_rejected=st.local_n_rejected,
                    n_nan_alpha=st.n_nan_alpha, k_gen=st.k_gen, t_abs=st.t_abs, history_rows=st.history_rows,
                    n_outlier_resets=st.n_outlier_resets,
                    p_cr=np.array(st.p_cr[:n]), delta_m=np.array(st.delta_m[:n]),
                    n_cr_updates=np.array(st.n_cr_updates[:n]))

    EXCHANGE_MODES = {"dense": 0, "rows": 1, "replay": 2, "push": 3}

    def set_exchange(self, mode="replay", cap=0):
        """world_size > 1: "push" (owners store accepted rows straight into the peers' replicas; the default once connected),
        "replay" (accept bytes through RCCL + recomputation), "rows" (accepted rows in packed blocks; cap = rows per sub-block
        per half generation) or "dense" (all-gather of whole blocks)"""
        if mode == "push-agent":          # the push exchange with agent-scope packet fences (include/bipymc_hip.h: bpm_set_exchange)
            mode, cap = "push", 1
        self._ck(self.lib.bpm_set_exchange(self._h, self.EXCHANGE_MODES[mode], int(cap)))

    def exchange_stats(self):
        out = (C.c_int64 * 8)()
        self._ck(self.lib.bpm_get_exchange_stats(self._h, out))
        return dict(mode=["dense", "rows", "replay", "push"][out[0]], cap=int(out[1]), chunks=int(out[2]), replays=int(out[3]),
                    replay_gens=int(out[4]), push_gens=int(out[5]), push_connected=bool(int(out[6]) & 3), push_flags_fine_grained=(int(out[6]) & 3) == 2,
                    # bpm_push_selftest's arena probe: stores into the peers' arenas arrived under system- / agent-scope packet fences; run on the own queue
                    arena_probe_system=bool(int(out[6]) & 4), arena_probe_agent=bool(int(out[6]) & 8), arena_probe_on_own_queue=bool(int(out[6]) & 16),
                    barriers=int(out[7]) & ((1 << 62) - 1), push_fence_scope="agent" if int(out[7]) >> 62 & 1 else "system")

    # ---- push exchange (world_size > 1): map the ranks' buffers into each other ------------------------------------------
    def push_export(self):
        """-> bytes: what the other ranks need to map this rank's exchange buffer (bpm_push_export)"""
        buf = C.create_string_buffer(L.PUSH_BLOB_BYTES)
        self._ck(self.lib.bpm_push_export(self._h, buf))
        return buf.raw

    def push_connect(self, blobs):
        """blobs: the exports of ALL ranks in rank order (bpm_push_connect)"""
        blobs = [bytes(b) for b in blobs]
        if len(blobs) != self.world_size or any(len(b) != L.PUSH_BLOB_BYTES for b in blobs):
            raise ValueError("push_connect needs one export per rank")
        buf = C.create_string_buffer(b"".join(blobs), L.PUSH_BLOB_BYTES * self.world_size)
        self._ck(self.lib.bpm_push_connect(self._h, buf))

    def push_selftest(self):
        """collective over the ranks (one engine per process): -> True when this rank received every peer's pattern"""
        arr = (C.c_void_p * 1)(self._h)
        ok = C.c_int32(0)
        self._ck(self.lib.bpm_push_selftest(arr, 1, C.byref(ok)))
        return bool(ok.value)

    @staticmethod
    def push_uid():
        """the nccl_uid of a world WITHOUT an RCCL communicator: the push exchange is its only one"""
        return b"BPMPUSH" + bytes(L.UID_BYTES - 7)

    def launch_stats(self):
        """How the update kernels were dispatched: the library's own AQL queue or the HIP stream (bpm_get_launch_stats)."""
        out = (C.c_int64 * 6)()
        self._ck(self.lib.bpm_get_launch_stats(self._h, out))
        return dict(has_queue=bool(out[0]), direct=int(out[1]), stream=int(out[2]), queue_active=bool(out[3]),
                    coherent_state=bool(out[4]), fence={3: "acquire+release", 1: "acquire", 0: "none"}.get(int(out[5]), int(out[5])))

    def set_launch_path(self, direct=True, fence=-1):
        """direct=False: HIP stream launches only; fence: 3 acquire + release, 1 acquire only, 0 none, -1 keep (bpm_set_launch_path)."""
        self._ck(self.lib.bpm_set_launch_path(self._h, 1 if direct else 0, int(fence)))

    def history_rows(self):
        return int(self.stats()["history_rows"])

    def get_history(self, g_lo=0, g_hi=None):
        """(g_hi - g_lo, n_local, dim): row g, local chain i."""
        if g_hi is None:
            g_hi = self.history_rows()
        out = np.empty((max(0, g_hi - g_lo), self.n_local, self.dim), dtype=np.float64)
        self._ck(self.lib.bpm_get_history(self._h, int(g_lo), int(g_hi), _dptr(out)))
        return out

    def get_loglike_history(self, g_lo=0, g_hi=None):
        if g_hi is None:
            g_hi = self.history_rows()
        out = np.empty((max(0, g_hi - g_lo), self.n_local), dtype=np.float64)
        self._ck(self.lib.bpm_get_loglike_history(self._h, int(g_lo), int(g_hi), _dptr(out)))
        return out

    def reduce_moments(self, n_burn=0):
        """-> (count, S1, S2, shift): raw moments of the local super-chain rows >= n_burn."""
        s1 = np.empty(self.dim); s2 = np.empty(self.dim); sh = np.empty(self.dim)
        n = C.c_int64(0)
        self._ck(self.lib.bpm_reduce_moments(self._h, int(n_burn), _dptr(s1), _dptr(s2), _dptr(sh), C.byref(n)))
        return int(n.value), s1, s2, sh

    def set_adapt_state(self, p_cr=None, delta_m=None, n_cr_updates=None, t_abs=-1):
        keep = [np.ascontiguousarray(a, dtype=np.float64) if a is not None else None
                for a in (p_cr, delta_m, n_cr_updates)]
        self._ck(self.lib.bpm_set_adapt_state(self._h, *[None if a is None else _dptr(a) for a in keep], int(t_abs)))

    def eval_loglike(self, X):
        X = np.ascontiguousarray(np.atleast_2d(X), dtype=np.float64)
        assert X.shape[1] == self.dim
        out = np.empty(X.shape[0], dtype=np.float64)
        self._ck(self.lib.bpm_eval_loglike(self._h, _dptr(X), X.shape[0], _dptr(out)))
        return out

    # ---- parity hooks (test variant of the library only: HipEngine(lib=_lib.load_test())) ----------------
    def set_trace(self, on=True):
        self._need_hooks("bpm_set_trace")
        self._ck(self.lib.bpm_set_trace(self._h, 1 if on else 0))

    def get_trace(self):
        ti = np.empty((self.n_local, L.TRACE_I32), dtype=np.int32)
        tf = np.empty((self.n_local, L.TRACE_F64), dtype=np.float64)
        tm = np.empty((self.n_local, self.dim), dtype=np.uint8)
        self._ck(self.lib.bpm_get_trace(self._h, _iptr(ti), _dptr(tf), tm.ctypes.data_as(C.POINTER(C.c_uint8))))
        return dict(cr_idx=ti[:, 0], d_prime=ti[:, 1], jump=ti[:, 2], accepted=ti[:, 3], snooker=ti[:, 4],
                    partners=ti[:, 5:5 + L.MAX_PARTNERS], alpha=tf[:, 0], ll_prop=tf[:, 1], delta=tf[:, 2],
                    gamma=tf[:, 3], mask=tm.astype(bool))

    def debug_perm(self, t, shuffle=True, flip_prob=0.5):
        order = np.empty(self.n_chains, dtype=np.int32)
        inv = np.empty(self.n_chains, dtype=np.int32)
        flip = C.c_int32(0)
        self._ck(self.lib.bpm_debug_perm(self._h, int(t), 1 if shuffle else 0, float(flip_prob), _iptr(order),
                                        _iptr(inv), C.byref(flip)))
        return order, inv, bool(flip.value)


def selftest_philox(n=4096, seed=42, device=0):
    """the inline Philox against rocRAND's device engine (a test hook: runs on the test variant of the library)"""
    lib = L.load_test()
    mine = np.empty((n, 4), dtype=np.uint32)
    ref = np.empty((n, 4), dtype=np.uint32)
    L.check(lib.bpm_selftest_philox(int(device), int(n), int(seed), mine.ctypes.data_as(C.POINTER(C.c_uint32)),
                                    ref.ctypes.data_as(C.POINTER(C.c_uint32))), lib)
    return mine, ref
