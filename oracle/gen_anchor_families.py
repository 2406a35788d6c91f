#!/usr/bin/env python3
"""
ORACLE tooling (test infrastructure, not product code).

End-to-end anchor FAMILIES recorded from the genuine reference (wgurecky/bipymc at /root/reference), imported in
THIS container exactly as oracle/gen_golden.py does (in-memory stand-ins for the absent h5py / mpi4py packages,
SURVEY.md section 8c).  A single reference run is one draw of a noisy process, so every scenario is run under
several `np.random.seed`s and the family (per-run numbers + median / min / max / std) is what the tests hold
the oracle engine and the device against.  Only numbers are written; the reference never travels to the GPU box.

Scenarios (the reference's own test files):
  gauss100_dream   tests/test_100dgauss.py:105-110   DreamMpi(Gauss_100D, zeros(100), n_chains=100, n_cr_gen=50,
                                                     burnin_gen=2000), run_mcmc(500000), n_burn=200000
  gauss100_demc    tests/test_100dgauss.py:100-103   DeMcMpi(Gauss_100D, zeros(100), n_chains=200), same n / n_burn
  banana_dream     tests/test_banana.py:123-127      DreamMpi(Banana_2D(1,1), [0,0], n_chains=10, n_cr_gen=50,
                                                     burnin_gen=2000), run_mcmc(100000), n_burn=20000
  banana_demc      tests/test_banana.py:118-121      DeMcMpi(Banana_2D(1,1), [0,0], n_chains=20), same n / n_burn
  bimodal_demc     tests/test_dblgauss.py:130-133    DeMcMpi(BimodeGauss_2D(), [0,0], n_chains=20), n=100000, n_burn=40000

Per run: acceptance fraction, p_cr / n_cr_updates / delta_m (DREAM), param_est(n_burn) mean and std per coordinate,
variance ratios against the analytic variances, the banana's level fractions (test_banana.py:66-72), and a
TRAJECTORY every `stride` generations read off the chains' histories: population variance ratio (mean over
coordinates of var_i(x_ij)/sigma_j^2), the fraction of chains that moved in the window, and p_cr at that point.

Usage:  python oracle/gen_anchor_families.py [--out tests/golden] [--workers 7] [--only gauss100_dream,...]
The d = 100 runs take 12-16 minutes each (the reference makes ~650 chain updates per second per core).
"""
import argparse
import json
import os
import sys
import time

# one BLAS thread per worker process (the d = 100 runs spend their time in numpy.linalg.svd inside multivariate_normal,
# util.py:13): must be in the environment BEFORE numpy is imported, in the spawned workers too
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ[_v] = "1"

import numpy as np  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

SCENARIOS = {
    # name: (algo, target, n_chains, n, n_burn, kwargs, seeds, stride)
    "gauss100_dream": ("dream", "gauss100", 100, 500000, 200000, dict(n_cr_gen=50, burnin_gen=2000), (42, 1, 2), 100),
    "gauss100_demc": ("demc", "gauss100", 200, 500000, 200000, {}, (42, 1, 2), 50),
    "banana_dream": ("dream", "banana", 10, 100000, 20000, dict(n_cr_gen=50, burnin_gen=2000), (42, 1, 2, 3, 4, 5), 500),
    "banana_demc": ("demc", "banana", 20, 100000, 20000, {}, (42, 1, 2, 3, 4, 5), 250),
    "bimodal_demc": ("demc", "bimodal", 20, 100000, 40000, {}, (42, 1, 2, 3, 4, 5), 250),
}


def _run(task):
    name, seed = task
    import gen_golden as G
    DreamMpi, DeMcMpi, d100_gauss, dblgauss_rv, banana_rv = G._import_reference()
    from mpi4py import MPI
    algo, tname, n_chains, n, n_burn, kw, _, stride = SCENARIOS[name]
    if tname == "gauss100":
        target = d100_gauss.Gauss_100D()
        theta_0 = np.zeros(100)
        true_var = np.arange(100) + 1.0                       # d100_gauss.py:17 (`var` holds sigma = sqrt(i+1))
        true_mean = np.zeros(100)
    elif tname == "banana":
        target = banana_rv.Banana_2D(sigma1=1.0, sigma2=1.0)
        theta_0 = [0.0, 0.0]
        a, b = 1.15, 0.5
        true_var = np.array([a * a, 1.0 / (a * a) + 2.0 * b * b])
        true_mean = np.array([0.0, b * (1.0 + a * a)])
    else:
        target = dblgauss_rv.BimodeGauss_2D()
        theta_0 = [0.0, 0.0]
        true_var = np.array([0.8125, 0.8125])
        true_mean = np.array([1.5, 1.5])
    np.random.seed(seed)
    cls = DreamMpi if algo == "dream" else DeMcMpi
    s = cls(target.ln_like, theta_0, n_chains=n_chains, mpi_comm=MPI.COMM_WORLD, **kw)

    pcr_traj = []
    if algo == "dream":
        orig = cls._update_chain_pool
        count = [0]

        def upd(self, k, c_id, current_chain, pool, pool_ids, **kwargs):
            orig(self, k, c_id, current_chain, pool, pool_ids, **kwargs)
            count[0] += 1
            if count[0] % (stride * n_chains) == 0:
                pcr_traj.append(np.array(self.p_cr, dtype=float).copy())

        cls._update_chain_pool = upd
    t0 = time.time()
    try:
        with np.errstate(divide="ignore", invalid="ignore"):
            s.run_mcmc(n)
    finally:
        if algo == "dream":
            cls._update_chain_pool = orig
    wall = time.time() - t0
    mean, std, chain = s.param_est(n_burn=n_burn)
    H = np.array([c.chain for c in s.am_chains])              # (N, T, d): chain.py:51-54, one row per generation
    N, T, d = H.shape
    gens = list(range(stride, T, stride))
    pop_var = [float(np.mean(H[:, g, :].var(axis=0) / true_var)) for g in gens]
    pop_mean_dev = [float(np.max(np.abs(H[:, g, :].mean(axis=0) - true_mean) / np.sqrt(true_var))) for g in gens]
    moved = np.any(H[:, 1:, :] != H[:, :-1, :], axis=2)       # (N, T-1): an accepted update changes the row
    win_acc = [float(moved[:, g - stride:g].mean()) for g in gens]
    vr = std ** 2 / true_var
    out = dict(
        scenario=name, seed=int(seed), algo=algo, target=tname, n_chains=n_chains, n=n, n_burn=n_burn, kwargs=kw,
        wall_s=round(wall, 1), updates_per_s=round((n - n_chains) / wall, 1), history_rows_per_chain=int(T),
        acceptance_fraction=float(s.acceptance_fraction), n_accepted=int(s.n_accepted), n_rejected=int(s.n_rejected),
        mean=mean.tolist(), std=std.tolist(),
        var_ratio_pooled=float(vr.mean()), var_ratio_min=float(vr.min()), var_ratio_max=float(vr.max()),
        mean_over_sigma_maxabs=float(np.max(np.abs(mean - true_mean) / np.sqrt(true_var))),
        traj=dict(stride=stride, gens=gens, pop_var_ratio=pop_var, pop_mean_dev_max=pop_mean_dev, window_acceptance=win_acc),
    )
    if algo == "dream":
        out.update(p_cr=np.asarray(s.p_cr, dtype=float).tolist(), n_cr_updates=np.asarray(s.n_cr_updates).tolist(),
                   delta_m=np.asarray(s.delta_m).tolist())
        out["traj"]["p_cr"] = [p.tolist() for p in pcr_traj[:len(gens)]]
    if tname == "banana":
        y1, y2 = chain[:, 0], chain[:, 1]
        out["frac_q50"] = float(np.count_nonzero(target.check_prob_lvl(y1, y2, 0.18)) / y1.size)   # test_banana.py:66-72
        out["frac_q95"] = float(np.count_nonzero(target.check_prob_lvl(y1, y2, 0.018)) / y1.size)
    return out


def _family(runs):
    fam = {}
    for key in ("acceptance_fraction", "var_ratio_pooled", "var_ratio_min", "var_ratio_max", "mean_over_sigma_maxabs",
                "frac_q50", "frac_q95", "p_cr", "n_cr_updates"):
        if key in runs[0]:
            a = np.array([r[key] for r in runs], dtype=float)
            fam[key] = dict(median=np.median(a, axis=0).tolist(), min=a.min(axis=0).tolist(), max=a.max(axis=0).tolist(),
                            std=a.std(axis=0).tolist())
    for key in ("pop_var_ratio", "window_acceptance", "p_cr"):
        if key in runs[0]["traj"]:
            a = np.array([r["traj"][key] for r in runs], dtype=float)
            fam["traj_" + key] = dict(median=np.median(a, axis=0).tolist(), min=a.min(axis=0).tolist(), max=a.max(axis=0).tolist())
    fam["traj_gens"] = runs[0]["traj"]["gens"]
    return fam


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(HERE, "..", "tests", "golden"))
    ap.add_argument("--workers", type=int, default=7)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    out = os.path.abspath(args.out)
    names = [x for x in args.only.split(",") if x] or list(SCENARIOS)
    tasks = [(nm, sd) for nm in names for sd in SCENARIOS[nm][6]]
    tasks.sort(key=lambda t: -SCENARIOS[t[0]][3] * (100 if "gauss100" in t[0] else 1))   # long runs first
    import multiprocessing as mp
    with mp.get_context("spawn").Pool(args.workers) as pool:
        results = []
        for r in pool.imap_unordered(_run, tasks):
            print("done %-16s seed %-3d %7.1f s  acc %.4f  var_ratio %.4f" % (
                r["scenario"], r["seed"], r["wall_s"], r["acceptance_fraction"], r["var_ratio_pooled"]), flush=True)
            results.append(r)
    for nm in names:
        runs = sorted([r for r in results if r["scenario"] == nm], key=lambda r: SCENARIOS[nm][6].index(r["seed"]))
        algo, tname, n_chains, n, n_burn, kw, seeds, stride = SCENARIOS[nm]
        doc = dict(
            config="%s %s n_chains=%d n=%d n_burn=%d %s; np.random.seed(s) for s in %s; genuine reference via oracle/gen_anchor_families.py"
                   % (algo, tname, n_chains, n, n_burn, json.dumps(kw), list(seeds)),
            scenario=nm, algo=algo, target=tname, n_chains=n_chains, n=n, n_burn=n_burn, kwargs=kw,
            family=_family(runs), runs=runs)
        with open(os.path.join(out, "e2e_anchor_%s.json" % nm), "w") as f:
            json.dump(doc, f, indent=1)
        print("wrote", nm, flush=True)


if __name__ == "__main__":
    main()
