"""Shared by the CPU and the GPU tests: the reference's OWN scenarios held against the families the genuine reference produced
(tests/golden/e2e_anchor_<scenario>.json, recorded by oracle/gen_anchor_families.py from /root/reference under several np.random seeds).

A family is (median, min, max, std) over the reference's seeds of: acceptance fraction, p_cr, the banana's level fractions, pooled post-burn-in
variance ratio, and a trajectory every `stride` generations (population variance ratio, window acceptance).  What is held against it: a run of THIS
build's sampler -- the CPU oracle engine (tests/test_oracle_golden.py) or the device through the drop-in classes (tests/test_gpu_api.py) -- in the
same configuration.  Tolerances: written here, each with its reason; they cover the family's own spread (3 or 6 reference seeds) and this
sampler's seed-to-seed spread (measured with 4 seeds of the oracle engine, round 5), never a systematic shift of more than a few standard errors.

Reference scenarios (file:line in /root/reference):
  gauss100_dream  tests/test_100dgauss.py:105-110   gauss100_demc  tests/test_100dgauss.py:100-103
  banana_dream    tests/test_banana.py:123-127      banana_demc    tests/test_banana.py:118-121      bimodal_demc  tests/test_dblgauss.py:130-133
"""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

SCENARIOS = ("gauss100_dream", "gauss100_demc", "banana_dream", "banana_demc", "bimodal_demc")

# |x - family median| <= tol.  d = 100 (100 / 200 chains, 5000 / 2500 generations): the reference's runs agree among themselves to 4e-4 in the acceptance
# fraction and 5e-3 in p_cr, this sampler's seeds to 8e-4 / 4e-3 -- 0.004 and 0.02 are ~4.5 combined standard deviations.  d = 2 (10 / 20 chains): the
# families themselves spread by 0.003-0.026 (acceptance) and 0.015 (p_cr).
TOL = {
    # var_ratio_pooled: at rho = 0.5 half of every coordinate's variance is ONE mode shared by all coordinates, and 100 chains estimate it to +-14 % per
    # snapshot with a long autocorrelation: the pooled post-burn-in variance of a run scatters by 0.03-0.05 (reference seeds 0.935 ... 0.983, this
    # sampler's 0.976 ... 1.057) -- 0.15 is ~3 standard deviations of the difference of two such runs.  The DE-MC run has not converged at n_burn
    # (0.40 of the target variance, still growing): its family is tight (0.402 ... 0.415), this sampler's seeds 0.407 ... 0.426.
    "gauss100_dream": dict(acceptance_fraction=0.004, p_cr=0.02, var_ratio_pooled=0.15, traj_until=1300),
    "gauss100_demc": dict(acceptance_fraction=0.004, var_ratio_pooled=0.04, traj_until=2450),
    "banana_dream": dict(acceptance_fraction=0.02, p_cr=0.06, frac=0.05),          # frac: the reference's own tolerance (test_banana.py:71-72)
    "banana_demc": dict(acceptance_fraction=0.03, frac=0.05),
    "bimodal_demc": dict(acceptance_fraction=0.08, mean_abs=0.1),                  # mean: the reference's own assertion (test_dblgauss.py:67-69)
}
BANANA_FRAC_ANALYTIC = (0.5070, 0.9507)      # fraction of the banana's mass above the pdf levels 0.18 / 0.018


def load(scenario):
    return json.load(open(os.path.join(GOLDEN, "e2e_anchor_%s.json" % scenario)))


def true_moments(target):
    if target == "gauss100":
        return np.zeros(100), np.arange(100) + 1.0
    if target == "banana":
        a, b = 1.15, 0.5
        return np.array([0.0, b * (1.0 + a * a)]), np.array([a * a, 1.0 / (a * a) + 2.0 * b * b])
    return np.full(2, 1.5), np.full(2, 0.8125)


def banana_level_fractions(chain):
    """check_prob_lvl of the reference (banana_rv.py:39-40) at the levels of tests/test_banana.py:66-67, restated: pdf of the twisted Gaussian"""
    a, b, rho = 1.15, 0.5, 0.9
    x1 = chain[:, 0] / a
    x2 = (chain[:, 1] - b * (x1 ** 2 + a * a)) * a
    q = (x1 * x1 - 2.0 * rho * x1 * x2 + x2 * x2) / (1.0 - rho * rho)
    pdf = np.exp(-0.5 * q) / (2.0 * np.pi * np.sqrt(1.0 - rho * rho))
    return float(np.mean(pdf > 0.18)), float(np.mean(pdf > 0.018))


def summarize(doc, history, acceptance_fraction, p_cr=None):
    """history: (T, N, d), row g = every chain after generation g (row 0 = the start).  -> the statistics a family holds"""
    tm, tv = true_moments(doc["target"])
    T, N, d = history.shape
    rows = history.reshape(T * N, d)[doc["n_burn"]:]                  # the interleaved super chain behind n_burn rows (demc.py:235-270)
    vr = rows.var(axis=0) / tv
    gens = doc["family"]["traj_gens"]
    stride = gens[0]
    moved = np.any(history[1:] != history[:-1], axis=2)              # (T - 1, N): an accepted update changes the row
    out = dict(acceptance_fraction=float(acceptance_fraction), var_ratio_pooled=float(vr.mean()),
               mean=rows.mean(axis=0),
               traj_pop_var_ratio=[float(np.mean(history[g].var(axis=0) / tv)) for g in gens if g < T],
               traj_window_acceptance=[float(moved[g - stride:g].mean()) for g in gens if g < T])
    if p_cr is not None:
        out["p_cr"] = [float(v) for v in p_cr]
    if doc["target"] == "banana":
        out["frac_q50"], out["frac_q95"] = banana_level_fractions(rows)
    return out


def check(scenario, got, who):
    """assert that `got` (summarize) lies inside the reference's family for this scenario"""
    doc = load(scenario)
    fam, tol = doc["family"], TOL[scenario]
    msg = "%s, %s" % (scenario, who)
    assert abs(got["acceptance_fraction"] - fam["acceptance_fraction"]["median"]) <= tol["acceptance_fraction"], (msg, got["acceptance_fraction"], fam["acceptance_fraction"])
    if "p_cr" in tol:
        dev = np.abs(np.array(got["p_cr"]) - np.array(fam["p_cr"]["median"]))
        assert np.all(dev <= tol["p_cr"]), (msg, got["p_cr"], fam["p_cr"]["median"])
        assert abs(sum(got["p_cr"]) - 1.0) < 1e-9
    if "var_ratio_pooled" in tol:
        assert abs(got["var_ratio_pooled"] - fam["var_ratio_pooled"]["median"]) <= tol["var_ratio_pooled"], (msg, got["var_ratio_pooled"], fam["var_ratio_pooled"])
    if "frac" in tol:
        for key, exact in zip(("frac_q50", "frac_q95"), BANANA_FRAC_ANALYTIC):
            assert abs(got[key] - fam[key]["median"]) <= tol["frac"] and abs(got[key] - exact) <= tol["frac"], (msg, key, got[key], fam[key])
    if "mean_abs" in tol:
        tm, _ = true_moments(doc["target"])
        assert np.all(np.abs(got["mean"] - tm) <= tol["mean_abs"]), (msg, got["mean"])
    if "traj_until" in tol:
        # the TRANSIENT from the reference's start (every chain within 1e-3 of the origin): the population's variance grows along the same curve,
        # the acceptance fraction falls along the same curve -- the reference's seeds agree to 1-3 % there, this sampler's to 5-10 %
        # ... POINT BY POINT within a factor of 1.35 (population variance; 1.8 in the exponential growth phase, where a run-to-run factor is a shift in time of
        # a few generations) / 1.15 (window acceptance), and ON AVERAGE over the transient within 12 % / 6 %: with 100-200 chains a single snapshot of
        # the population variance carries ~7 % of noise on either side (half of every coordinate's variance is one mode shared by all coordinates)
        gens = fam["traj_gens"]
        lr_pv, lr_wa = [], []
        for i, g in enumerate(gens):
            if g > tol["traj_until"] or i >= len(got["traj_pop_var_ratio"]):
                break
            pv, ref_pv = got["traj_pop_var_ratio"][i], fam["traj_pop_var_ratio"]["median"][i]
            if ref_pv < 0.05:
                assert abs(np.log(pv / ref_pv)) <= np.log(1.8), (msg, "population variance ratio at generation %d (growth phase)" % g, pv, ref_pv)
            else:
                assert abs(np.log(pv / ref_pv)) <= np.log(1.35), (msg, "population variance ratio at generation %d" % g, pv, ref_pv)
                lr_pv.append(np.log(pv / ref_pv))
            wa, ref_wa = got["traj_window_acceptance"][i], fam["traj_window_acceptance"]["median"][i]
            assert abs(np.log(wa / ref_wa)) <= np.log(1.15), (msg, "window acceptance at generation %d" % g, wa, ref_wa)
            lr_wa.append(np.log(wa / ref_wa))
        assert len(lr_pv) >= 5 and abs(np.mean(lr_pv)) <= 0.12, (msg, "population variance along the transient, mean log ratio", float(np.mean(lr_pv)))
        assert abs(np.mean(lr_wa)) <= 0.06, (msg, "window acceptance along the transient, mean log ratio", float(np.mean(lr_wa)))
