"""
ORACLE (test infrastructure, not product code).

CPU (NumPy) restatement of the per-generation hot path of wgurecky/bipymc's
parallel DE-MC / DREAM samplers.  Every function cites the reference lines it
follows (paths relative to /root/reference).  Two layers:

  1. "core": the deterministic arithmetic of one chain update with EVERY random
     draw an explicit input.  Pinned bit-for-bit against step vectors recorded
     from the genuine reference (tests/golden/steps_*.npz, produced by
     oracle/gen_golden.py) and against its ln_like known answers
     (tests/golden/targets_known_answers.json).
  2. "engine": `OracleSampler`, the generation driver of demc.py:63-151 restated
     in the batched, half-generation-synchronous form the MI355X kernels use,
     drawing its randomness from the counter-based layout of
     oracle/philox_ref.py.  The HIP path is compared against this on identical
     seeds: bit-exact on every integer quantity (shuffle order, pool split,
     CR index, mask, pair ids, gamma pick), <= 1e-12 relative on float state
     after one generation.

Documented deviations of layer 2 from the reference (DESIGN.md "Deviations"):
  * p_cr is re-estimated once per generation from all chains' statistics, not
    after every single chain update (dream.py:92-93 is sequential);
  * the per-chain history std of dream.py:128 comes from running Welford
    moments instead of an O(T) rescan;
  * ln_like of the current state is cached instead of re-evaluated
    (samplers.py:330 calls it twice per update);
  * targets return true log-densities instead of log(pdf) (no -inf underflow);
  * the permutation of demc.py:84-86 is a keyed bijection, not a Fisher-Yates
    shuffle of the NumPy global stream.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module; the product path never does.
"""
import math

import numpy as np

from . import philox_ref as P

LN_2PI = math.log(2.0 * math.pi)

ALGO_DEMC = 0
ALGO_DREAM = 1
ALGO_DEMC_SYNC = 2      # serial DeMc of samplers.py:237-308 with delayed_accept=True (banked, synchronous updates)

TARGET_HOST = 0
TARGET_GAUSS_EQUICORR = 1
TARGET_MIXTURE_PAIRS = 2
TARGET_BANANA_2D = 3


# =====================================================================
# Layer 0: targets (bipymc/utils/*.py), as closed-form log-densities
# =====================================================================
def gauss_equicorr_params(rho, sigma):
    """Parameter block of the equicorrelated Gaussian (utils/d100_gauss.py:14-27:
    cov_ii = sigma_i^2, cov_ij = rho sigma_i sigma_j).  Layout
    [rho, c0, a, b, 1/sigma_0 .. 1/sigma_{d-1}] with
    ll(y) = c0 - 0.5 (a S2 - b S1^2), S1 = sum z, S2 = sum z^2, z = y / sigma."""
    sigma = np.asarray(sigma, dtype=np.float64)
    d = sigma.size
    rho = float(rho)
    logdet = 2.0 * np.sum(np.log(sigma)) + (d - 1) * math.log(1.0 - rho) + math.log(1.0 + (d - 1) * rho)
    c0 = -0.5 * (d * LN_2PI + logdet)
    a = 1.0 / (1.0 - rho)
    b = rho / ((1.0 + (d - 1) * rho) * (1.0 - rho))
    return np.concatenate([[rho, c0, a, b], 1.0 / sigma])


def ll_gauss_equicorr(x, params):
    """log N(x; 0, Sigma) for utils/d100_gauss.py:33-35 (true log-density)."""
    x = np.asarray(x, dtype=np.float64)
    c0, a, b = params[1], params[2], params[3]
    z = x * params[4:]
    s1 = np.sum(z, axis=-1)
    s2 = np.sum(z * z, axis=-1)
    return c0 - 0.5 * (a * s2 - b * s1 * s1)


def mixture_pairs_params(w1, w2, mu1, mu2, sig1, sig2, rho1, rho2):
    """Two-component Gaussian mixture whose components are block-diagonal in
    coordinate pairs (2k, 2k+1); every pair has the 2x2 blocks of
    utils/dblgauss_rv.py:11-24.  d = 2 is the reference's BimodeGauss_2D.
    Layout [lw1, lw2, then per component: mx, my, 1/sx, 1/sy, rho, 1/(1-rho^2), ln_norm]."""
    wsum = float(w1) + float(w2)
    out = [math.log(w1 / wsum), math.log(w2 / wsum)]
    for mu, sg, rho in ((mu1, sig1, rho1), (mu2, sig2, rho2)):
        h = 1.0 / (1.0 - rho * rho)
        ln_norm = -(LN_2PI + math.log(sg[0]) + math.log(sg[1]) + 0.5 * math.log(1.0 - rho * rho))
        out += [float(mu[0]), float(mu[1]), 1.0 / sg[0], 1.0 / sg[1], float(rho), h, ln_norm]
    return np.array(out, dtype=np.float64)


def ll_mixture_pairs(x, params):
    """log(w1 N1(x) + w2 N2(x)) (utils/dblgauss_rv.py:26-32) by log-sum-exp."""
    x = np.asarray(x, dtype=np.float64)
    xe, xo = x[..., 0::2], x[..., 1::2]
    npairs = xe.shape[-1]
    comp = []
    for c in range(2):
        mx, my, isx, isy, rho, h, ln_norm = params[2 + 7 * c: 9 + 7 * c]
        u = (xe - mx) * isx
        v = (xo - my) * isy
        q = np.sum((u * u - 2.0 * rho * u * v + v * v) * h, axis=-1)
        comp.append(params[c] + npairs * ln_norm - 0.5 * q)
    m = np.maximum(comp[0], comp[1])
    return m + np.log(np.exp(comp[0] - m) + np.exp(comp[1] - m))


def banana_params(mu1=0.0, mu2=0.0, sigma1=1.0, sigma2=1.0, rho=0.9, a=1.15, b=0.5):
    """utils/banana_rv.py:11-24.  Layout [mu1, mu2, 1/s1, 1/s2, rho, 1/(1-rho^2), ln_norm, a, b]."""
    h = 1.0 / (1.0 - rho * rho)
    ln_norm = -(LN_2PI + math.log(sigma1) + math.log(sigma2) + 0.5 * math.log(1.0 - rho * rho))
    return np.array([mu1, mu2, 1.0 / sigma1, 1.0 / sigma2, rho, h, ln_norm, a, b], dtype=np.float64)


def ll_banana(x, params):
    """utils/banana_rv.py:26-37: Gaussian in the un-twisted coordinates (unit Jacobian)."""
    x = np.asarray(x, dtype=np.float64)
    mu1, mu2, is1, is2, rho, h, ln_norm, a, b = params
    x1 = x[..., 0] / a
    x2 = (x[..., 1] - b * (x1 * x1 + a * a)) * a
    u = (x1 - mu1) * is1
    v = (x2 - mu2) * is2
    return ln_norm - 0.5 * (u * u - 2.0 * rho * u * v + v * v) * h


def eval_target(target_id, params, x):
    if target_id == TARGET_GAUSS_EQUICORR:
        return ll_gauss_equicorr(x, params)
    if target_id == TARGET_MIXTURE_PAIRS:
        return ll_mixture_pairs(x, params)
    if target_id == TARGET_BANANA_2D:
        return ll_banana(x, params)
    raise ValueError("unknown target id %r" % (target_id,))


# =====================================================================
# Layer 1: core arithmetic of one update, every draw an explicit input
# =====================================================================
def split_pools(shuffle_idx, flip):
    """demc.py:95-100: halves of the shuffled id order (first gets ceil(N/2)), swapped on flip.
    Returns (a_ids, b_ids): a is updated first against pool b."""
    a_ids, b_ids = np.array_split(np.asarray(shuffle_idx), 2)
    if flip:
        a_ids, b_ids = b_ids, a_ids
    return a_ids, b_ids


def cr_values(n_cr):
    """dream.py:113."""
    return (np.array(range(n_cr)) + 1) / n_cr


def cr_mask(z, cr, forced_dim):
    """dream.py:52-58: dims with z <= cr; if none, the one forced dim."""
    m = (z <= cr)
    if np.count_nonzero(m) == 0:
        m = m.copy()
        m[forced_dim] = True
    return m


def dream_gamma_base(gamma_scale, del_pairs, d_prime):
    """dream.py:61."""
    return gamma_scale * 2.38 / np.sqrt(2. * del_pairs * d_prime)


def dream_proposal(cur, A, B, gamma, eps_u, eps_n, mask):
    """dream.py:85-89.  A, B: (..., P, d) partner rows; others (..., d) / (...,)."""
    cur = np.asarray(cur)
    update_dims = np.zeros(cur.shape)
    update_dims[mask] = 1.0
    g = np.asarray(gamma)[..., None] if np.ndim(gamma) else gamma
    prop = ((np.ones(cur.shape[-1]) + eps_u) * g * np.sum(A - B, axis=-2) + eps_n) * update_dims
    prop = prop + cur
    return prop


def demc_gamma_base(dim, gamma=None):
    """demc.py:162."""
    return gamma if gamma is not None else 2.38 / np.sqrt(2. * dim)


def demc_proposal(cur, a, b, gamma, eps_n):
    """demc.py:180-182 (same association order: gamma*(a-b), += cur, += eps)."""
    g = np.asarray(gamma)[..., None] if np.ndim(gamma) else gamma
    prop = g * (a - b)
    prop = prop + cur
    prop = prop + eps_n
    return prop


def mut_prop_ratio(ll_cur, ll_prop):
    """samplers.py:328-332."""
    with np.errstate(over="ignore", invalid="ignore"):
        alpha = np.minimum(1.0, np.exp(ll_prop - ll_cur))
    return np.clip(alpha, 0.0, 1.0)


def metropolis_accept(alpha, u):
    """samplers.py:334-336: choice([True, False], p=[alpha, 1-alpha]) == (u < alpha)."""
    return u < alpha


def cr_delta(cur, prop, std):
    """dream.py:128-130: squared proposed jump normalised by the chain's own history std."""
    std = np.array(std, dtype=np.float64, copy=True)
    std[std == 0] = 1e-12
    return np.sum(((cur - prop) ** 2.0 / std ** 2.0), axis=-1)


class CrState(object):
    """dream.py:109-117."""

    def __init__(self, n_cr):
        self.n_cr = n_cr
        self.CR = cr_values(n_cr)
        self.p_cr = np.ones(n_cr) / n_cr
        self.n_cr_updates = np.zeros(n_cr)
        self.delta_m = np.zeros(n_cr)

    def copy(self):
        c = CrState(self.n_cr)
        c.p_cr = self.p_cr.copy()
        c.n_cr_updates = self.n_cr_updates.copy()
        c.delta_m = self.delta_m.copy()
        return c

    def update_sequential(self, cr_idx, delta):
        """dream.py:125-140 for ONE chain update (the reference's order)."""
        self.n_cr_updates[cr_idx] += 1.0
        self.delta_m[cr_idx] += delta
        self._refresh()

    def update_batched(self, cr_idx, delta):
        """Same accumulators, all updates of one generation at once (cr_idx < 0 = gated off)."""
        cr_idx = np.asarray(cr_idx)
        delta = np.asarray(delta, dtype=np.float64)
        any_upd = False
        for m in range(self.n_cr):
            sel = cr_idx == m
            if np.any(sel):
                any_upd = True
                self.n_cr_updates[m] += float(np.count_nonzero(sel))
                self.delta_m[m] += np.sum(delta[sel])
        if any_upd:
            self._refresh()

    def _refresh(self):
        if np.count_nonzero(self.n_cr_updates) == self.n_cr:
            self.p_cr = self.delta_m / self.n_cr_updates
        self.p_cr = self.p_cr / np.sum(self.p_cr)


def welford_push(count, mean, m2, row):
    """Running per-chain moments over history rows (replaces np.std(chain.chain), dream.py:128)."""
    count = count + 1
    d = row - mean
    mean = mean + d / count
    m2 = m2 + d * (row - mean)
    return count, mean, m2


def welford_std(count, m2):
    return np.sqrt(m2 / count)


def others_position_to_id(i, pos):
    """samplers.py:274: np.delete(np.array(range(N)), i)[pos] -- the pool of chain i is every OTHER chain, in id order."""
    pos = np.asarray(pos)
    return pos + (pos >= np.asarray(i))


def demc_sync_generation(X, ll, ids, pair_a, pair_b, gamma, eps_n, accept, ll_fn):
    """One generation of the serial DeMc with delayed_accept=True (samplers.py:268-308), every draw an explicit input.

    X (N, d): states at the START of the generation; ids: the chains updated here (all N in the reference; a rank's
    block in a sharded run); pair_a / pair_b: CHAIN IDS of the mutation pair of each chain (what
    `np.random.choice(valid_pool_ids, replace=False, size=2)` returns, :275); gamma: scalar, no gamma = 1 jump in this
    sampler (:263,281); eps_n (n, d): the var_ball draw (:283); accept: callable alpha -> bool array (the
    metropolis_accept decisions, :289).  Every proposal is made from X -- also for partners that already accepted in
    this sweep -- because updates are banked until every chain has proposed (:296-308).
    Returns (new rows of `ids`, new ll of `ids`, prop, ll_prop, alpha, accepted)."""
    cur = X[ids]
    prop = demc_proposal(cur, X[pair_a], X[pair_b], np.full(len(ids), gamma), eps_n)      # samplers.py:281-283
    ll_prop = ll_fn(prop)
    alpha = mut_prop_ratio(ll[ids], ll_prop)                                              # samplers.py:287-289
    accepted = accept(alpha)
    new_rows = np.where(accepted[:, None], prop, cur)                                     # banked_prop_array, :290-303
    new_ll = np.where(accepted, ll_prop, ll[ids])
    return new_rows, new_ll, prop, ll_prop, alpha, accepted


def snooker_third(wz, w1, w2, m):
    """Three distinct pool positions (z, z1, z2) from three words."""
    iz = P.mulhi(wz, m)
    i1 = P.mulhi(w1, np.asarray(m) - 1)
    i1 = i1 + (i1 >= iz)
    lo = np.minimum(iz, i1)
    hi = np.maximum(iz, i1)
    i2 = P.mulhi(w2, np.asarray(m) - 2)
    i2 = i2 + (i2 >= lo)
    i2 = i2 + (i2 >= hi)
    return iz, i1, i2


def snooker_proposal(cur, z, z1, z2, gamma_s, eps_n):
    """ter Braak & Vrugt (2008) snooker update (NOT in the reference; parity unpinned):
    x' = x + gamma_s * <z1 - z2, x - z> / |x - z|^2 * (x - z) (+ eps_n)."""
    diff = cur - z
    n2 = np.sum(diff * diff, axis=-1)
    proj = np.sum((z1 - z2) * diff, axis=-1) / n2
    prop = cur + (np.asarray(gamma_s) * proj)[..., None] * diff + eps_n
    return prop, n2


# =====================================================================
# Layer 2: the generation driver with the counter-based draw layout
# =====================================================================
class OracleSampler(object):
    """demc.py:63-151 + dream.py:32-140, batched per half generation.

    State is kept for ALL N chains (the replicated state matrix); `rank`/`world`
    select which contiguous block (demc.py:39) this instance updates, and
    `allgather(local_block) -> (N, d)` supplies the exchange of demc.py:93-94 /
    116-117 for world > 1.
    """

    def __init__(self, algo, n_chains, dim, target_id, target_params, seed,
                 rank=0, world=1, allgather=None, ll_fn=None,
                 gamma_scale=1.0, del_pairs=3, burnin_gen=300, n_cr_gen=50, n_cr=3,
                 p_snooker=0.0, outlier_every=0):
        assert n_chains >= 4                       # samplers.py:249
        assert n_chains % world == 0
        assert 1 <= del_pairs <= P.MAX_PAIRS
        self.algo, self.N, self.d = algo, int(n_chains), int(dim)
        self.target_id = target_id
        self.params = None if target_params is None else np.asarray(target_params, dtype=np.float64)
        self.ll_fn = ll_fn
        self.seed = int(seed)
        self.rank, self.world, self.allgather = rank, world, allgather
        self.n_local = self.N // world
        self.lo = rank * self.n_local
        self.hi = self.lo + self.n_local
        self.gamma_scale, self.P = float(gamma_scale), int(del_pairs)
        self.burnin_gen, self.n_cr_gen, self.n_cr = int(burnin_gen), int(n_cr_gen), int(n_cr)
        self.p_snooker = float(p_snooker)
        self.outlier_every = int(outlier_every)
        self.n_outlier_resets = 0
        self.cr = CrState(self.n_cr)
        self.X = np.zeros((self.N, self.d))
        self.ll = np.zeros(self.N)
        self.history = []            # list of (n_local, d) rows blocks, one per history row
        self.ll_history = []
        self.w_count = 0
        self.w_mean = np.zeros((self.n_local, self.d))
        self.w_m2 = np.zeros((self.n_local, self.d))
        self.t = 0                   # absolute generation counter (never reset)
        self.local_n_accepted = 0
        self.local_n_rejected = 1    # demc.py:67-68
        self.n_nan = 0
        self.trace = None            # optional per-generation integer trace for parity tests

    # ---- targets ---------------------------------------------------
    def _ll(self, x):
        if self.ll_fn is not None:
            return np.array([self.ll_fn(r) for r in np.atleast_2d(x)], dtype=np.float64)
        return eval_target(self.target_id, self.params, x)

    # ---- state -----------------------------------------------------
    def init_jitter(self, theta_0, varepsilon):
        """chain.py:25-27: state0 = theta_0 + N(0, diag(varepsilon)); varepsilon is a VARIANCE.
        Draws: chain subsequence, t = 2^47-1 (never reached by generations), dim blocks words 2,3."""
        theta_0 = np.asarray(theta_0, dtype=np.float64).reshape(-1)
        var = np.broadcast_to(np.asarray(varepsilon, dtype=np.float64), (self.d,))
        ids = np.arange(self.N)
        X = np.tile(theta_0, (self.N, 1))
        if np.all(var > 0):                         # util.py:12
            w = P.chain_block(self.seed, ids[:, None], P.T_INIT, P.SLOT_DIM0 + np.arange(self.d)[None, :])
            X = X + np.sqrt(var)[None, :] * P.box_muller(w[..., 2], w[..., 3])
        self.set_state(X)

    def set_state(self, X):
        self.X = np.array(X, dtype=np.float64).reshape(self.N, self.d)
        self.ll = self._ll(self.X)
        self.history = [self.X[self.lo:self.hi].copy()]
        self.ll_history = [self.ll[self.lo:self.hi].copy()]
        self.w_count = 0
        self.w_mean[:] = 0
        self.w_m2[:] = 0
        self._welford_sync()

    def _welford_sync(self):
        while self.w_count < len(self.history):
            self.w_count, self.w_mean, self.w_m2 = welford_push(
                self.w_count, self.w_mean, self.w_m2, self.history[self.w_count])

    # ---- one generation ----------------------------------------------
    def run(self, n_gens, flip=0.5, shuffle=True, epsilon=None, u_epsilon=1e-2, gamma=None):
        """One `run_mcmc` call: k restarts at 0, counters reset (demc.py:67-68,78)."""
        self.local_n_accepted = 0
        self.local_n_rejected = 1
        flip = float(np.clip(flip, 0.0, 1.0))
        if epsilon is None:
            epsilon = 1e-12 if self.algo == ALGO_DREAM else 1e-15   # dream.py:40, demc.py:161
        for k in range(n_gens):
            self._generation(k, flip, shuffle, float(epsilon), float(u_epsilon), gamma)

    def _generation(self, k, flip_prob, shuffle, epsilon, u_epsilon, gamma_kw):
        if self.algo == ALGO_DEMC_SYNC:
            return self._generation_sync(k, epsilon, gamma_kw)
        t, N = self.t, self.N
        flip = P.flip_draw(self.seed, t, flip_prob)
        order = P.shuffle_idx(self.seed, t, N, shuffle)
        a_ids, b_ids = split_pools(order, flip)
        adapt_on = (self.algo == ALGO_DREAM) and (self.burnin_gen > k)     # dream.py:92
        if adapt_on:
            self._welford_sync()
        hist_len = len(self.history)
        cr_idx_all = np.full(N, -1, dtype=np.int64)
        delta_all = np.zeros(N)
        tr = dict(flip=flip, order=order, a_ids=a_ids, b_ids=b_ids) if self.trace is not None else None
        for phase, (upd_ids, pool_ids) in enumerate(((a_ids, b_ids), (b_ids, a_ids))):
            mine = upd_ids[(upd_ids >= self.lo) & (upd_ids < self.hi)]
            if mine.size:
                res = self._update(k, t, mine, pool_ids, epsilon, u_epsilon, gamma_kw, adapt_on, hist_len)
                if tr is not None:
                    tr["phase%d" % phase] = res
                self.X[mine] = res["new_state"]
                self.ll[mine] = res["new_ll"]
                cr_idx_all[mine] = res["cr_idx_eff"]
                delta_all[mine] = res["delta"]
                na = int(np.count_nonzero(res["accepted"]))
                self.local_n_accepted += na
                self.local_n_rejected += mine.size - na
                self.n_nan += int(np.count_nonzero(np.isnan(res["alpha"])))
            if self.world > 1:
                self.X = self.allgather(self.X[self.lo:self.hi].copy())
        if self.world > 1 and self.algo == ALGO_DREAM:
            cr_idx_all = self.allgather(cr_idx_all[self.lo:self.hi].astype(np.float64).reshape(-1, 1)).reshape(-1).astype(np.int64)
            delta_all = self.allgather(delta_all[self.lo:self.hi].reshape(-1, 1)).reshape(-1)
        self.history.append(self.X[self.lo:self.hi].copy())
        self.ll_history.append(self.ll[self.lo:self.hi].copy())
        if adapt_on:
            self._welford_sync()
            self.cr.update_batched(cr_idx_all, delta_all)
        if tr is not None:
            tr["p_cr"] = self.cr.p_cr.copy()
            self.trace.append(tr)
        self.t += 1
        if (self.algo == ALGO_DREAM and self.outlier_every > 0 and (k + 1) < self.burnin_gen
                and (k + 1) % self.outlier_every == 0):
            self._outlier_check()

    def _outlier_check(self):
        """DREAM outlier-chain reset (Vrugt et al. 2009; NOT in the reference, parity unpinned): chains whose
        mean ln_like over the last half of their history is below Q1 - 2 IQR restart from the best chain."""
        rows = len(self.ll_history)
        omega = np.zeros(self.n_local)
        cnt = np.zeros(self.n_local)
        for g in range(rows // 2, rows):            # sequential accumulation, as the kernel does; rows of unknown ln_like (NaN) do not count
            v = self.ll_history[g]
            ok = ~np.isnan(v)
            omega = np.where(ok, omega + np.where(ok, v, 0.0), omega)
            cnt = cnt + ok
        with np.errstate(divide="ignore", invalid="ignore"):
            omega = omega / cnt
        ll_all = self.ll[self.lo:self.hi].copy()
        if self.world > 1:
            omega = self.allgather(omega.reshape(-1, 1)).reshape(-1)
            ll_all = self.allgather(ll_all.reshape(-1, 1)).reshape(-1)
        else:
            ll_all = self.ll.copy()
        q1, q3 = np.percentile(omega, [25.0, 75.0])
        cut = q1 - 2.0 * (q3 - q1)
        best = int(np.argmax(omega))
        out = np.nonzero(omega < cut)[0]
        if out.size == 0:
            return
        self.X[out] = self.X[best]
        for c in out:
            if self.lo <= c < self.hi:
                self.ll[c] = ll_all[best]
                self.history[-1][c - self.lo] = self.X[best]
                self.ll_history[-1][c - self.lo] = ll_all[best]
        self.n_outlier_resets += int(out.size)
        self.w_count = 0
        self.w_mean[:] = 0
        self.w_m2[:] = 0

    def _generation_sync(self, k, epsilon, gamma_kw):
        """samplers.py:268-308 with delayed_accept=True: every chain proposes from the states at the start
        of the generation, pair drawn from all OTHER chains (np.delete, :274-275), no gamma jumps, updates banked."""
        t, N, d, seed = self.t, self.N, self.d, self.seed
        ids = np.arange(self.lo, self.hi)
        n = ids.size
        h0 = P.chain_block(seed, ids, t, P.SLOT_HDR0)
        npairs = (d + 1) // 2
        wd = P.chain_block(seed, ids[:, None], t, P.SLOT_DIM0 + np.arange(npairs)[None, :])
        if epsilon > 0:
            n0, n1 = P.box_muller_pair_f32(wd[..., 2], wd[..., 3])
            eps_n = epsilon * np.stack([n0, n1], axis=-1).reshape(n, 2 * npairs)[:, :d]
        else:
            eps_n = np.zeros((n, d))
        wp = P.chain_block(seed, ids, t, P.SLOT_PAIR0)
        ia, ib = P.distinct_pair(wp[:, 0], wp[:, 1], N - 1)
        ia = others_position_to_id(ids, ia)          # positions in np.delete(range(N), i)
        ib = others_position_to_id(ids, ib)
        gamma = demc_gamma_base(d, gamma_kw)
        ua = P.u01_53(h0[:, 2], h0[:, 3])
        new_local, new_ll, prop, ll_prop, alpha, accepted = demc_sync_generation(
            self.X, self.ll, ids, ia, ib, gamma, eps_n, lambda al: metropolis_accept(al, ua), self._ll)
        self.ll[ids] = new_ll
        na = int(np.count_nonzero(accepted))
        self.local_n_accepted += na
        self.local_n_rejected += n - na
        self.n_nan += int(np.count_nonzero(np.isnan(alpha)))
        if self.world > 1:
            self.X = self.allgather(new_local)
        else:
            self.X = new_local.copy()
        if self.trace is not None:
            self.trace.append(dict(pa=ia, pb=ib, accepted=accepted, alpha=alpha, ll_prop=ll_prop))
        self.history.append(self.X[self.lo:self.hi].copy())
        self.ll_history.append(self.ll[self.lo:self.hi].copy())
        self.t += 1

    def _update(self, k, t, ids, pool_ids, epsilon, u_epsilon, gamma_kw, adapt_on, hist_len):
        seed, d = self.seed, self.d
        n = ids.size
        M = pool_ids.size
        cur = self.X[ids]
        ll_cur = self.ll[ids]
        # one header block per chain update: (select16|gamma16, forced dim / snooker gamma, accept hi, accept lo)
        h0 = P.chain_block(seed, ids, t, P.SLOT_HDR0)
        sel_hi, sel_lo = P.split16(h0[:, 0])
        u_sel = sel_hi * 2.0 ** -16          # DREAM: CR select; DE-MC: snooker select
        u_gam = sel_lo * 2.0 ** -16          # gamma = 1 jump select
        npairs = (d + 1) // 2
        wd = P.chain_block(seed, ids[:, None], t, P.SLOT_DIM0 + np.arange(npairs)[None, :])   # (n, npairs, 4)

        def interleave(first, second):
            """(n, npairs) x 2 -> (n, d): coordinates 2pi, 2pi+1"""
            return np.stack([first, second], axis=-1).reshape(n, 2 * npairs)[:, :d]

        if epsilon > 0:
            n0, n1 = P.box_muller_pair_f32(wd[..., 2], wd[..., 3])
            eps_n = epsilon * interleave(n0, n1)
        else:
            eps_n = np.zeros((n, d))
        out = dict(ids=ids)
        log_corr = np.zeros(n)
        if self.algo == ALGO_DREAM:
            # CR index ~ Categorical(p_cr)   dream.py:51
            uc = u_sel
            cum = np.cumsum(self.cr.p_cr)     # sequential running sum, as the kernel does
            cr_idx = np.minimum((uc[:, None] >= cum[None, :]).sum(axis=1), self.n_cr - 1)
            cr = self.cr.CR[cr_idx]
            zk = interleave(*P.split16(wd[..., 0]))                      # 16-bit uniforms k * 2^-16
            forced = P.mulhi(h0[:, 1], d)
            mask = zk <= P.mask_threshold(cr)[:, None]                   # == (k * 2^-16 <= cr), dream.py:53
            none = ~mask.any(axis=1)
            mask[none, forced[none]] = True
            d_prime = mask.sum(axis=1)
            gamma_base = dream_gamma_base(self.gamma_scale, self.P, d_prime)
            pa = np.empty((n, self.P), dtype=np.int64)
            pb = np.empty((n, self.P), dtype=np.int64)
            for p in range(self.P):
                wp = P.chain_block(seed, ids, t, P.SLOT_PAIR0 + p // 2)
                pa[:, p], pb[:, p] = P.distinct_pair(wp[:, 2 * (p % 2)], wp[:, 2 * (p % 2) + 1], M)
            if k % 5 == 0:                       # dream.py:77-80
                jump = ~(u_gam < 0.2)
                gamma = np.where(jump, 1.0, gamma_base)
            else:
                jump = np.zeros(n, dtype=bool)
                gamma = gamma_base
            if u_epsilon > 0:                                            # util.py:18-28
                eps_u = -u_epsilon + (2.0 * u_epsilon) * P.u_sym16(interleave(*P.split16(wd[..., 1])))
            else:
                eps_u = np.zeros((n, d))
            A = self.X[pool_ids[pa]]             # (n, P, d)
            B = self.X[pool_ids[pb]]
            prop = dream_proposal(cur, A, B, gamma, eps_u, eps_n, mask)
            out.update(cr_idx=cr_idx, mask=mask, d_prime=d_prime, pa=pool_ids[pa], pb=pool_ids[pb], jump=jump)
            if adapt_on and hist_len > self.n_cr_gen:         # dream.py:92, 123-124
                li = ids - self.lo
                std = welford_std(self.w_count, self.w_m2[li])
                delta = cr_delta(cur, prop, std)
                cr_idx_eff = cr_idx
            else:
                delta = np.zeros(n)
                cr_idx_eff = np.full(n, -1, dtype=np.int64)
        else:
            gamma_base = demc_gamma_base(d, gamma_kw)
            wp = P.chain_block(seed, ids, t, P.SLOT_PAIR0)
            ia, ib = P.distinct_pair(wp[:, 0], wp[:, 1], M)
            if k % 10 == 0:                      # demc.py:174-177
                jump = ~(u_gam < 0.1)
                gamma = np.where(jump, 1.0, gamma_base)
            else:
                jump = np.zeros(n, dtype=bool)
                gamma = np.full(n, gamma_base)
            prop = demc_proposal(cur, self.X[pool_ids[ia]], self.X[pool_ids[ib]], gamma, eps_n)
            snk = np.zeros(n, dtype=bool)
            if self.p_snooker > 0 and M >= 3:
                snk = u_sel < self.p_snooker
                ws = P.chain_block(seed, ids, t, P.SLOT_SNK)
                iz, i1, i2 = snooker_third(ws[:, 0], ws[:, 1], ws[:, 2], M)
                gamma_s = 1.2 + P.u01_32(h0[:, 1])
                zz = self.X[pool_ids[iz]]
                with np.errstate(divide="ignore", invalid="ignore"):
                    sprop, n2 = snooker_proposal(cur, zz, self.X[pool_ids[i1]], self.X[pool_ids[i2]], gamma_s, eps_n)
                    snk = snk & (n2 > 0)
                    dn = sprop - zz
                    corr = 0.5 * (d - 1) * (np.log(np.sum(dn * dn, axis=-1)) - np.log(n2))
                prop = np.where(snk[:, None], sprop, prop)
                log_corr = np.where(snk, corr, 0.0)
                out.update(snooker=snk, iz=pool_ids[iz], i1=pool_ids[i1], i2=pool_ids[i2])
            out.update(pa=pool_ids[ia], pb=pool_ids[ib], jump=jump)
            delta = np.zeros(n)
            cr_idx_eff = np.full(n, -1, dtype=np.int64)
        ll_prop = self._ll(prop)
        alpha = mut_prop_ratio(ll_cur, ll_prop + log_corr)
        ua = P.u01_53(h0[:, 2], h0[:, 3])
        accepted = metropolis_accept(alpha, ua)
        new_state = np.where(accepted[:, None], prop, cur)
        new_ll = np.where(accepted, ll_prop, ll_cur)
        out.update(prop=prop, ll_prop=ll_prop, alpha=alpha, accepted=accepted, new_state=new_state,
                   new_ll=new_ll, delta=delta, cr_idx_eff=cr_idx_eff)
        return out

    # ---- results -----------------------------------------------------
    def history_array(self):
        """(T, n_local, d): history[g][i] = chain lo+i at generation g (row 0 = initial)."""
        return np.stack(self.history, axis=0)

    def super_chain(self):
        """demc.py:260-270 for world == 1: row g*N + i = chain i at generation g."""
        assert self.world == 1
        return self.history_array().reshape(-1, self.d)

    def param_est(self, n_burn):
        """demc.py:235-248."""
        s = self.super_chain()[n_burn:, :]
        return np.mean(s, axis=0), np.std(s, axis=0), s
