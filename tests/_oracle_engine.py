"""Test-only engine: the CPU oracle behind the `HipEngine` interface, so the host-side classes
(DeMcMpi/DreamMpi: argument translation, generation count, sharding, result assembly) can be
exercised without a GPU, single-rank and under torch.distributed/gloo.  Never used by the product."""
import numpy as np

from oracle import sampler_ref as R


class OracleEngine(object):
    def __init__(self, algo, n_chains, dim, target_id, target_params, seed, device=0, rank=0, world_size=1,
                 nccl_uid=None, gamma_scale=1.0, del_pairs=3, burnin_gen=300, n_cr_gen=50, n_cr=3,
                 p_snooker=0.0, outlier_every=0, keep_history=True, running_moments=False, ll_fn=None):
        self.n_chains, self.dim, self.rank, self.world_size = n_chains, dim, rank, world_size
        self.n_local = n_chains // world_size
        self.lo = rank * self.n_local
        self.algo, self.target_id = algo, target_id
        allgather = None
        if world_size > 1:
            import torch
            import torch.distributed as dist

            def allgather(local):
                t = torch.from_numpy(np.ascontiguousarray(local))
                out = [torch.empty_like(t) for _ in range(world_size)]
                dist.all_gather(out, t)
                return torch.cat(out, dim=0).numpy()
        kw = dict(gamma_scale=gamma_scale, burnin_gen=burnin_gen, n_cr_gen=n_cr_gen, n_cr=n_cr if algo == R.ALGO_DREAM else 1,
                  del_pairs=del_pairs if algo == R.ALGO_DREAM else 1, p_snooker=p_snooker,
                  outlier_every=outlier_every)
        self.s = R.OracleSampler(algo, n_chains, dim, target_id, target_params, seed, rank=rank, world=world_size,
                                 allgather=allgather, ll_fn=ll_fn, **kw)
        self._k = 0
        self._opts = None
        self._pending = None

    def close(self):
        pass

    def init_chains(self, theta_0, varepsilon):
        self.s.init_jitter(theta_0, varepsilon)

    def set_state(self, X):
        self.s.set_state(X)

    def set_history(self, hist_local, X):
        self.s.set_state(X)
        self.s.history = [h.copy() for h in hist_local]
        self.s.ll_history = [self.s._ll(h) for h in hist_local]
        self.s.w_count = 0
        self.s.w_mean[:] = 0
        self.s.w_m2[:] = 0

    def get_state(self):
        return self.s.X.copy()

    def set_loglike(self, ll_local):
        self.s.ll[self.lo:self.lo + self.n_local] = ll_local

    def get_loglike(self):
        return self.s.ll[self.lo:self.lo + self.n_local].copy()

    def begin_run(self, flip=0.5, shuffle=True, epsilon=None, u_epsilon=None, gamma=None):
        if epsilon is None:
            epsilon = 1e-12 if self.algo == R.ALGO_DREAM else 1e-15
        self._opts = (float(np.clip(flip, 0, 1)), bool(shuffle), float(epsilon),
                      1e-2 if u_epsilon is None else float(u_epsilon), gamma)
        self._k = 0
        self.s.local_n_accepted, self.s.local_n_rejected = 0, 1

    def step(self, n_gens):
        for _ in range(int(n_gens)):
            self.s._generation(self._k, *self._opts)
            self._k += 1

    def synchronize(self):
        pass

    def reserve_history(self, rows):
        pass

    def stats(self):
        return dict(local_n_accepted=self.s.local_n_accepted, local_n_rejected=self.s.local_n_rejected,
                    n_nan_alpha=self.s.n_nan, k_gen=self._k, t_abs=self.s.t, history_rows=len(self.s.history),
                    n_outlier_resets=self.s.n_outlier_resets, p_cr=self.s.cr.p_cr.copy(), delta_m=self.s.cr.delta_m.copy(),
                    n_cr_updates=self.s.cr.n_cr_updates.copy())

    def history_rows(self):
        return len(self.s.history)

    def get_history(self, g_lo=0, g_hi=None):
        return np.stack(self.s.history[g_lo:g_hi], axis=0)

    def reduce_moments(self, n_burn=0):
        """raw moments of this rank's super-chain rows >= n_burn (row g*N + i), about chain 0's current state"""
        H = np.stack(self.s.history, axis=0)                      # (T, n_local, d)
        T = H.shape[0]
        g = np.arange(T)[:, None] * self.n_chains + (self.lo + np.arange(self.n_local))[None, :]
        sel = g >= n_burn
        sh = self.s.X[0].copy()
        rows = H[sel] - sh
        return int(sel.sum()), rows.sum(axis=0), (rows ** 2).sum(axis=0), sh

    def set_adapt_state(self, p_cr=None, delta_m=None, n_cr_updates=None, t_abs=-1):
        if p_cr is not None:
            self.s.cr.p_cr = np.array(p_cr, dtype=float)
            self.s.cr.delta_m = np.array(delta_m, dtype=float)
            self.s.cr.n_cr_updates = np.array(n_cr_updates, dtype=float)
        if t_abs >= 0:
            self.s.t = int(t_abs)


def factory(**kw):
    return OracleEngine(**kw)
