#!/usr/bin/env python3
"""
ORACLE tooling (test infrastructure, not product code).

Generates the golden fixtures under tests/golden/ by importing the genuine
reference (wgurecky/bipymc at /root/reference) in THIS container and recording
what it computes.  The reference never travels to the GPU box: only the numbers
this script writes (small .npz/.json files) are committed.

The reference needs `h5py` and `mpi4py`, which are not installed here; as
described in SURVEY.md section 8(c) two in-memory stand-in modules are registered
in `sys.modules` before the import: an empty `h5py` (only `h5py.File` is named,
never called on the hot path) and a single-rank `mpi4py.MPI.COMM_WORLD`
(`Allgather` copies send -> recv).  Nothing of the reference is modified.

Fixtures written:
  G1  targets_known_answers.json   ln_like of the three shipped targets at fixed points
  G2  steps_dream_bimodal.npz      every random draw + every intermediate of each chain
      steps_dream_gauss16.npz      update of short reference runs (DREAM d=2 N=10 with CR
      steps_demc_banana.npz        adaptation active; DREAM d=16 equicorrelated Gaussian
                                   N=8; DE-MC banana N=8)
      steps_demc_serial_banana.npz the SERIAL sampler bipymc.samplers.DeMc (samplers.py:261-308,
                                   delayed_accept=True): pool = np.delete(range(N), i), no gamma
                                   jump, updates banked until every chain has proposed
  G3  e2e_anchor_cfg1.json         end-to-end moments of a seed-42 cfg1-like run
      e2e_anchor_cfg1_seeds.json   p_cr / acceptance of the same run under six seeds (their spread)

Usage:  python oracle/gen_golden.py [--out tests/golden]
"""
import argparse
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"


def _install_standins():
    h5py = types.ModuleType("h5py")

    class File(object):  # named by chain.py isinstance checks only
        def __init__(self, *a, **k):
            raise RuntimeError("h5py stand-in: no file I/O in the golden generator")

    h5py.File = File
    sys.modules["h5py"] = h5py

    mpi4py = types.ModuleType("mpi4py")
    MPI = types.ModuleType("mpi4py.MPI")

    class _Comm(object):
        size = 1
        rank = 0

        def Get_size(self):
            return 1

        def Get_rank(self):
            return 0

        def Barrier(self):
            pass

        def Allgather(self, send, recv):
            s = np.asarray(send[0])
            r = recv[0]
            r[...] = s.reshape(r.shape)

    MPI.COMM_WORLD = _Comm()
    MPI.DOUBLE = "DOUBLE"
    MPI.INT = "INT"
    MPI.ANY_SOURCE = -1
    MPI.ANY_TAG = -1
    MPI.Status = type("Status", (), {})
    mpi4py.MPI = MPI
    sys.modules["mpi4py"] = mpi4py
    sys.modules["mpi4py.MPI"] = MPI


def _import_reference():
    os.environ.setdefault("MPLBACKEND", "agg")
    sys.dont_write_bytecode = True
    _install_standins()
    sys.path.insert(0, REF)
    from bipymc.dream import DreamMpi
    from bipymc.demc import DeMcMpi
    from bipymc.utils import d100_gauss, dblgauss_rv, banana_rv
    return DreamMpi, DeMcMpi, d100_gauss, dblgauss_rv, banana_rv


def record_serial_demc(sampler, n, theta_0, **run_kwargs):
    """Run the serial `bipymc.samplers.DeMc.run_mcmc(n, theta_0)` (samplers.py:261-308) recording, per chain
    update, the draws it made (pair = CHAIN IDS chosen from np.delete(range(N), i); normal jitter; accept
    decision) and the proposal / ratio it computed, and per generation the chain states before and after."""
    cls = type(sampler)
    rec = Recorder()
    updates = []
    orig_ratio = cls._mut_prop_ratio
    orig_init = cls._init_chains
    frozen = sampler._frozen_ln_like_fn
    mark = {"i0": 0}

    def init(self, theta_0, varepsilon=1e-6, **kw):
        orig_init(self, theta_0, varepsilon, **kw)
        mark["i0"] = len(rec.calls)                       # draws of chain.py:27 end here

    def ratio(self, fn, current_theta, mut_theta):
        alpha = orig_ratio(self, fn, current_theta, mut_theta)
        updates.append(dict(current=np.array(current_theta).copy(), prop=np.array(mut_theta).copy(), alpha=float(alpha),
                            ll_prop=float(frozen(mut_theta)), ll_cur=float(frozen(current_theta)), n_calls=len(rec.calls)))
        return alpha

    cls._mut_prop_ratio = ratio
    cls._init_chains = init
    try:
        with rec:
            sampler.run_mcmc(n, theta_0, **run_kwargs)
    finally:
        cls._mut_prop_ratio = orig_ratio
        cls._init_chains = orig_init
    # per update the reference draws: choice (pair ids), multivariate_normal (jitter), then -- after the ratio --
    # choice (accept)
    calls = rec.calls
    pos = mark["i0"]
    for u in updates:
        seg = calls[pos:u["n_calls"]]
        assert [c[0] for c in seg] == ["choice", "multivariate_normal"], [c[0] for c in seg]
        u["pair"] = seg[0][1].astype(np.int64)
        u["eps_n"] = seg[1][1].reshape(-1)
        name, acc = calls[u["n_calls"]]
        assert name == "choice"
        u["accept_draw"] = bool(acc)
        pos = u["n_calls"] + 1
    assert pos == len(calls)
    return updates


class Recorder(object):
    """Logs every np.random call the reference makes and the sampler internals
    around each `_update_chain_pool` call."""

    NAMES = ("choice", "uniform", "shuffle", "multivariate_normal")

    def __init__(self):
        self.calls = []
        self._orig = {}

    def __enter__(self):
        for name in self.NAMES:
            orig = getattr(np.random, name)
            self._orig[name] = orig
            setattr(np.random, name, self._wrap(name, orig))
        return self

    def __exit__(self, *exc):
        for name, orig in self._orig.items():
            setattr(np.random, name, orig)

    def _wrap(self, name, orig):
        def f(*a, **k):
            if name == "shuffle":
                orig(*a, **k)
                self.calls.append((name, np.array(a[0]).copy()))
                return None
            out = orig(*a, **k)
            self.calls.append((name, np.array(out).copy()))
            return out
        return f


def record_run(sampler, n, is_dream, **run_kwargs):
    """Run `sampler.run_mcmc(n)` recording per-update and per-generation data."""
    cls = type(sampler)
    updates = []
    gens = []
    rec = Recorder()
    orig_update = cls._update_chain_pool
    orig_ratio = cls._mut_prop_ratio
    frozen = sampler._frozen_ln_like_fn
    cur = {}

    def upd(self, k, c_id, current_chain, pool, pool_ids, **kw):
        i0 = len(rec.calls)
        cur.clear()
        cur.update(k=int(k), c_id=int(c_id), current=current_chain.current_pos.copy(),
                   pool=np.array(pool).copy(), pool_ids=np.array(pool_ids).copy(),
                   hist_len=int(current_chain.chain_len))
        if is_dream:
            cur.update(p_cr_before=np.array(self.p_cr, dtype=float).copy(),
                       delta_m_before=self.delta_m.copy(),
                       n_cr_updates_before=self.n_cr_updates.copy(),
                       hist_std=np.std(current_chain.chain, axis=0))
        acc0 = self.local_n_accepted
        orig_update(self, k, c_id, current_chain, pool, pool_ids, **kw)
        cur["accepted"] = bool(self.local_n_accepted > acc0)
        cur["new_state"] = current_chain.current_pos.copy()
        if is_dream:
            cur.update(p_cr_after=np.array(self.p_cr, dtype=float).copy(),
                       delta_m_after=self.delta_m.copy(),
                       n_cr_updates_after=self.n_cr_updates.copy())
        cur["draws"] = rec.calls[i0:]
        updates.append(dict(cur))

    def ratio(self, fn, current_theta, mut_theta):
        alpha = orig_ratio(self, fn, current_theta, mut_theta)
        cur.update(prop=np.array(mut_theta).copy(), alpha=float(alpha),
                   ll_prop=float(frozen(mut_theta)), ll_cur=float(frozen(current_theta)))
        return alpha

    cls._update_chain_pool = upd
    cls._mut_prop_ratio = ratio
    try:
        with rec:
            sampler.run_mcmc(n, **run_kwargs)
    finally:
        cls._update_chain_pool = orig_update
        cls._mut_prop_ratio = orig_ratio

    # generation-level draws: a `choice` of a bool followed by a `shuffle`
    calls = rec.calls
    for i, (name, val) in enumerate(calls):
        if name == "shuffle":
            assert calls[i - 1][0] == "choice"
            gens.append(dict(flip=bool(calls[i - 1][1]), shuffle_idx=val.astype(np.int64)))
    return updates, gens


def parse_dream_draws(u, dim, del_pairs):
    """Split the np.random call log of one DREAM update (dream.py:51-99) into named draws."""
    d = list(u["draws"])
    it = iter(d)
    name, cr = next(it); assert name == "choice"
    name, z = next(it); assert name == "uniform" and z.shape == (dim,)
    out = dict(cr=float(np.asarray(cr).reshape(-1)[0]), z=z)
    nxt = next(it)
    forced = -1
    if np.count_nonzero(z <= out["cr"]) == 0:
        assert nxt[0] == "choice"
        forced = int(nxt[1])
        nxt = next(it)
    out["forced_dim"] = forced
    assert nxt[0] == "choice" and nxt[1].shape == (2, del_pairs)  # dead draw, dream.py:62
    pairs = []
    for _ in range(del_pairs):
        name, pr = next(it); assert name == "choice" and pr.shape == (2,)
        pairs.append(pr.astype(np.int64))
    out["pairs"] = np.array(pairs)
    nxt = next(it)
    gamma_pick = np.nan
    if u["k"] % 5 == 0:
        assert nxt[0] == "choice"
        gamma_pick = float(nxt[1])
        nxt = next(it)
    out["gamma_pick"] = gamma_pick
    assert nxt[0] == "uniform" and nxt[1].shape == (dim,)
    out["eps_u"] = nxt[1]
    name, en = next(it); assert name == "multivariate_normal"
    out["eps_n"] = en.reshape(-1)
    name, acc = next(it); assert name == "choice"
    out["accept_draw"] = bool(acc)
    rest = list(it)
    assert not rest, rest
    return out


def parse_demc_draws(u, dim):
    """Split the call log of one DE-MC update (demc.py:169-188)."""
    it = iter(u["draws"])
    name, pr = next(it); assert name == "choice" and pr.shape == (2,)
    out = dict(pair=pr.astype(np.int64))
    nxt = next(it)
    gamma_pick = np.nan
    if u["k"] % 10 == 0:
        assert nxt[0] == "choice"
        gamma_pick = float(nxt[1])
        nxt = next(it)
    out["gamma_pick"] = gamma_pick
    assert nxt[0] == "multivariate_normal"
    out["eps_n"] = nxt[1].reshape(-1)
    name, acc = next(it); assert name == "choice"
    out["accept_draw"] = bool(acc)
    assert not list(it)
    return out


def pack_dream(updates, gens, dim, del_pairs, meta):
    n = len(updates)
    P = [parse_dream_draws(u, dim, del_pairs) for u in updates]
    pool_n = max(len(u["pool"]) for u in updates)

    def padpool(u):
        a = np.full((pool_n, dim), np.nan)
        a[:len(u["pool"])] = u["pool"]
        return a

    def padids(u):
        a = np.full((pool_n,), -1, dtype=np.int64)
        a[:len(u["pool_ids"])] = u["pool_ids"]
        return a

    out = dict(
        k=np.array([u["k"] for u in updates]), c_id=np.array([u["c_id"] for u in updates]),
        hist_len=np.array([u["hist_len"] for u in updates]),
        current=np.array([u["current"] for u in updates]),
        pool=np.array([padpool(u) for u in updates]), pool_ids=np.array([padids(u) for u in updates]),
        pool_len=np.array([len(u["pool"]) for u in updates]),
        cr=np.array([p["cr"] for p in P]), z=np.array([p["z"] for p in P]),
        forced_dim=np.array([p["forced_dim"] for p in P]),
        pairs=np.array([p["pairs"] for p in P]),
        gamma_pick=np.array([p["gamma_pick"] for p in P]),
        eps_u=np.array([p["eps_u"] for p in P]), eps_n=np.array([p["eps_n"] for p in P]),
        accept_draw=np.array([p["accept_draw"] for p in P]),
        prop=np.array([u["prop"] for u in updates]), alpha=np.array([u["alpha"] for u in updates]),
        ll_prop=np.array([u["ll_prop"] for u in updates]), ll_cur=np.array([u["ll_cur"] for u in updates]),
        accepted=np.array([u["accepted"] for u in updates]),
        new_state=np.array([u["new_state"] for u in updates]),
        hist_std=np.array([u["hist_std"] for u in updates]),
        p_cr_before=np.array([u["p_cr_before"] for u in updates]),
        p_cr_after=np.array([u["p_cr_after"] for u in updates]),
        delta_m_before=np.array([u["delta_m_before"] for u in updates]),
        delta_m_after=np.array([u["delta_m_after"] for u in updates]),
        n_cr_updates_before=np.array([u["n_cr_updates_before"] for u in updates]),
        n_cr_updates_after=np.array([u["n_cr_updates_after"] for u in updates]),
        gen_flip=np.array([g["flip"] for g in gens]),
        gen_shuffle_idx=np.array([g["shuffle_idx"] for g in gens]),
        meta=np.array(json.dumps(meta)),
    )
    assert len(out["k"]) == n
    return out


def pack_demc(updates, gens, dim, meta):
    P = [parse_demc_draws(u, dim) for u in updates]
    return dict(
        k=np.array([u["k"] for u in updates]), c_id=np.array([u["c_id"] for u in updates]),
        current=np.array([u["current"] for u in updates]),
        pool=np.array([u["pool"] for u in updates]), pool_ids=np.array([u["pool_ids"] for u in updates]),
        pair=np.array([p["pair"] for p in P]), gamma_pick=np.array([p["gamma_pick"] for p in P]),
        eps_n=np.array([p["eps_n"] for p in P]), accept_draw=np.array([p["accept_draw"] for p in P]),
        prop=np.array([u["prop"] for u in updates]), alpha=np.array([u["alpha"] for u in updates]),
        ll_prop=np.array([u["ll_prop"] for u in updates]), ll_cur=np.array([u["ll_cur"] for u in updates]),
        accepted=np.array([u["accepted"] for u in updates]),
        new_state=np.array([u["new_state"] for u in updates]),
        gen_flip=np.array([g["flip"] for g in gens]),
        gen_shuffle_idx=np.array([g["shuffle_idx"] for g in gens]),
        meta=np.array(json.dumps(meta)),
    )


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    args = ap.parse_args()
    out = os.path.abspath(args.out)
    os.makedirs(out, exist_ok=True)
    DreamMpi, DeMcMpi, d100_gauss, dblgauss_rv, banana_rv = _import_reference()
    from mpi4py import MPI

    # ---- G1: target known answers -------------------------------------
    g100 = d100_gauss.Gauss_100D()
    bim = dblgauss_rv.BimodeGauss_2D()
    ban = banana_rv.Banana_2D(sigma1=1, sigma2=1)
    i = np.arange(100)
    pts100 = {
        "zeros": np.zeros(100), "ones": np.ones(100), "linspace": np.linspace(-1, 1, 100),
        "alt_half_sigma": ((-1.0) ** i) * 0.5 * np.sqrt(i + 1.0), "tens": 10.0 * np.ones(100),
        "forties": 40.0 * np.ones(100),
    }
    rs = np.random.RandomState(7)
    for j in range(4):
        pts100["rand%d" % j] = rs.normal(size=100) * np.sqrt(i + 1.0) * (0.5 + j)
    pts2 = [(0, 0), (2, 2), (1, 1), (1.5, 1.5), (0.3, -0.2), (5, 5), (-3, 4), (2.1, 1.7), (-0.2, 0.25)]
    ptsb = [(0, 0), (0, 1.16125), (1, 1), (-2, 3), (3, 8), (0.5, -1), (-1.3, 2.2), (2.5, 4.0)]
    with np.errstate(divide="ignore"):
        g1 = {
            "Gauss_100D": [{"name": k, "x": v.tolist(), "ln_like": float(g100.ln_like(v)),
                            "logpdf_true": float(g100.rv_100d.logpdf(v))} for k, v in pts100.items()],
            "BimodeGauss_2D": [{"x": list(map(float, p)), "ln_like": float(bim.ln_like(np.array(p, dtype=float)))}
                               for p in pts2],
            "Banana_2D": [{"x": list(map(float, p)), "ln_like": float(ban.ln_like(np.array(p, dtype=float)))}
                          for p in ptsb],
        }
    # a 16-D equicorrelated Gaussian built by the reference class (dim kwarg, d100_gauss.py:14)
    g16 = d100_gauss.Gauss_100D(rho=0.5, dim=16)
    rs = np.random.RandomState(8)
    g1["Gauss_16D"] = []
    for j in range(4):
        v = rs.normal(size=16) * np.sqrt(np.arange(16) + 1.0)
        g1["Gauss_16D"].append({"x": v.tolist(), "ln_like": float(g16.ln_like(v))})
    with open(os.path.join(out, "targets_known_answers.json"), "w") as f:
        json.dump(g1, f, indent=1)

    # ---- G2: step-level vectors ---------------------------------------
    # (i) cfg1-like DREAM on the 2-D bimodal, CR adaptation switching on after n_cr_gen
    np.random.seed(42)
    meta = dict(target="BimodeGauss_2D", n_chains=10, dim=2, n=10 * 31, n_cr_gen=12, burnin_gen=25,
                del_pairs=3, n_cr=3, gamma_scale=1.0, varepsilon=1e-6, seed=42)
    s = DreamMpi(bim.ln_like, np.zeros(2), n_chains=10, mpi_comm=MPI.COMM_WORLD,
                 n_cr_gen=meta["n_cr_gen"], burnin_gen=meta["burnin_gen"])
    init = np.array([c.current_pos for c in s.am_chains])
    ups, gens = record_run(s, meta["n"], True)
    pk = pack_dream(ups, gens, 2, 3, meta)
    pk["init_state"] = init
    mean, std, chain = s.param_est(0)
    pk["final_super_chain_tail"] = chain[-20:]
    pk["acceptance_fraction"] = np.array(s.acceptance_fraction)
    np.savez_compressed(os.path.join(out, "steps_dream_bimodal.npz"), **pk)

    # (ii) DREAM on a 16-D equicorrelated Gaussian
    np.random.seed(43)
    meta = dict(target="Gauss_16D", n_chains=8, dim=16, n=8 * 13, n_cr_gen=5, burnin_gen=9,
                del_pairs=3, n_cr=3, gamma_scale=1.0, varepsilon=1e-6, seed=43, rho=0.5)
    s = DreamMpi(g16.ln_like, np.zeros(16), n_chains=8, mpi_comm=MPI.COMM_WORLD,
                 n_cr_gen=meta["n_cr_gen"], burnin_gen=meta["burnin_gen"])
    init = np.array([c.current_pos for c in s.am_chains])
    ups, gens = record_run(s, meta["n"], True)
    pk = pack_dream(ups, gens, 16, 3, meta)
    pk["init_state"] = init
    np.savez_compressed(os.path.join(out, "steps_dream_gauss16.npz"), **pk)

    # (iii) DE-MC on the banana
    np.random.seed(44)
    meta = dict(target="Banana_2D", n_chains=8, dim=2, n=8 * 25, varepsilon=1e-6, seed=44)
    s = DeMcMpi(ban.ln_like, np.zeros(2), n_chains=8, mpi_comm=MPI.COMM_WORLD)
    init = np.array([c.current_pos for c in s.am_chains])
    ups, gens = record_run(s, meta["n"], False)
    pk = pack_demc(ups, gens, 2, meta)
    pk["init_state"] = init
    np.savez_compressed(os.path.join(out, "steps_demc_banana.npz"), **pk)

    # (iv) the SERIAL DeMc of samplers.py:237-308 (delayed_accept=True, the default) on the banana
    from bipymc.samplers import DeMc
    np.random.seed(45)
    meta = dict(target="Banana_2D", n_chains=8, dim=2, n=8 * 21, varepsilon=1e-6, inflate=1e1, seed=45, delayed_accept=True)
    s = DeMc(ban.ln_like, n_chains=8)
    ups = record_serial_demc(s, meta["n"], np.zeros(2))
    N = meta["n_chains"]
    n_gens = len(ups) // N
    assert len(ups) == n_gens * N and n_gens == 20
    hist = np.array([c.chain for c in s.am_chains])            # (N, T, d): chain.py:51-54, one row per generation
    assert hist.shape == (N, n_gens + 1, 2)
    pk = dict(
        current=np.array([u["current"] for u in ups]), pair=np.array([u["pair"] for u in ups]),
        eps_n=np.array([u["eps_n"] for u in ups]), accept_draw=np.array([u["accept_draw"] for u in ups]),
        prop=np.array([u["prop"] for u in ups]), alpha=np.array([u["alpha"] for u in ups]),
        ll_prop=np.array([u["ll_prop"] for u in ups]), ll_cur=np.array([u["ll_cur"] for u in ups]),
        history=np.transpose(hist, (1, 0, 2)).copy(),           # (T, N, d): row g = every chain after generation g
        super_chain=s.super_chain, n_accepted=np.array(s.n_accepted), n_rejected=np.array(s.n_rejected),
        meta=np.array(json.dumps(meta)))
    np.savez_compressed(os.path.join(out, "steps_demc_serial_banana.npz"), **pk)

    # ---- G3: end-to-end anchor (cfg1 shape, shortened) ------------------
    np.random.seed(42)
    s = DreamMpi(bim.ln_like, np.zeros(2), n_chains=10, mpi_comm=MPI.COMM_WORLD,
                 n_cr_gen=50, burnin_gen=2000)
    s.run_mcmc(20000)
    mean, std, chain = s.param_est(n_burn=8000)
    g3 = dict(config="DREAM BimodeGauss_2D n_chains=10 n=20000 n_cr_gen=50 burnin_gen=2000 n_burn=8000 seed=42",
              mean=mean.tolist(), std=std.tolist(), acceptance_fraction=float(s.acceptance_fraction),
              p_cr=np.asarray(s.p_cr).tolist(), rows=int(chain.shape[0]),
              super_chain_shape=list(s.param_est(0)[2].shape))
    with open(os.path.join(out, "e2e_anchor_cfg1.json"), "w") as f:
        json.dump(g3, f, indent=1)

    # ---- G3b: the same run under further seeds.  p_cr of ONE reference run is a noisy estimate (sequential updates of few chains:
    # seed 42 ends at p_cr[0] = 0.18 with n_cr_updates = (2027, 8039, 9424), the other seeds at 0.30-0.34): a statistical anchor for
    # p_cr and the acceptance fraction has to be the family, not one member
    runs = []
    for seed in (42, 1, 2, 3, 4, 5):
        np.random.seed(seed)
        s = DreamMpi(bim.ln_like, np.zeros(2), n_chains=10, mpi_comm=MPI.COMM_WORLD, n_cr_gen=50, burnin_gen=2000)
        s.run_mcmc(20000)
        runs.append(dict(seed=seed, p_cr=np.asarray(s.p_cr).tolist(), acceptance_fraction=float(s.acceptance_fraction),
                         n_cr_updates=np.asarray(s.n_cr_updates).tolist()))
    pcr = np.array([r["p_cr"] for r in runs])
    acc = np.array([r["acceptance_fraction"] for r in runs])
    g3b = dict(config="DREAM BimodeGauss_2D n_chains=10 n=20000 n_cr_gen=50 burnin_gen=2000, np.random.seed(s) for s in runs",
               runs=runs, p_cr_median=np.median(pcr, axis=0).tolist(), p_cr_std=pcr.std(axis=0).tolist(),
               acceptance_median=float(np.median(acc)), acceptance_std=float(acc.std()))
    with open(os.path.join(out, "e2e_anchor_cfg1_seeds.json"), "w") as f:
        json.dump(g3b, f, indent=1)
    print("golden fixtures written to", out)


if __name__ == "__main__":
    main()
