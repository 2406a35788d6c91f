"""Distributional checks of the on-device random decisions (through the trace hook of the C ABI): the
frequencies the reference's NumPy draws would have (dream.py:51-80, demc.py:169-177, samplers.py:334-336)."""
import numpy as np
import pytest

from oracle import philox_ref as P
from oracle import sampler_ref as R

pytestmark = pytest.mark.gpu


def _engine(**kw):
    from bipymc_amd.engine import HipEngine
    return HipEngine(**kw)


def _hooks_engine(**kw):
    """on the test variant of the library: the per-chain decision trace is part of the test surface (include/bipymc_hip_test.h)"""
    from bipymc_amd import _lib as L
    from bipymc_amd.engine import HipEngine
    return HipEngine(lib=L.load_test(), **kw)


def test_dream_decision_frequencies():
    N, d, G = 2048, 20, 40
    params = R.gauss_equicorr_params(0.5, np.sqrt(np.arange(d) + 1.0))
    e = _hooks_engine(algo=R.ALGO_DREAM, n_chains=N, dim=d, target_id=R.TARGET_GAUSS_EQUICORR, target_params=params, seed=123,
                burnin_gen=0, n_cr=3, del_pairs=3)
    X = np.random.RandomState(0).normal(size=(N, d)) * np.sqrt(np.arange(d) + 1.0)
    e.set_state(X)
    e.set_trace(True)
    e.begin_run()
    cr_counts = np.zeros(3)
    mask_rate = {0: [], 1: [], 2: []}
    jump_on, jump_n = 0, 0
    acc, alpha_sum, n_upd = 0, 0.0, 0
    partner_hist = np.zeros(N)
    same_pool_violations = 0
    for g in range(G):
        order, flip = P.shuffle_idx(123, g, N), P.flip_draw(123, g, 0.5)     # (== the device's: test_gpu_parity.py::test_shuffle_and_flip_bit_exact)
        e.step(1)
        tr = e.get_trace()
        cr_counts += np.bincount(tr["cr_idx"], minlength=3)
        for m in range(3):
            sel = tr["cr_idx"] == m
            if sel.any():
                mask_rate[m].append(tr["mask"][sel].mean())
        if g % 5 == 0:
            jump_on += tr["jump"].sum(); jump_n += N
        else:
            assert not tr["jump"].any()
        acc += tr["accepted"].sum(); alpha_sum += tr["alpha"].sum(); n_upd += N
        P_ = tr["partners"][:, :6]
        np.add.at(partner_hist, P_.reshape(-1), 1)
        assert np.all(P_[:, 0::2] != P_[:, 1::2])                         # the two members of a pair differ (dream.py:66)
        # partners come from the OTHER pool (demc.py:103-109,126-132)
        first = np.zeros(N, dtype=bool); first[order[:N // 2]] = True
        same_pool_violations += int(np.sum(first[P_] == first[:, None]))
    assert same_pool_violations == 0
    # CR ~ Categorical(p_cr = 1/3 each) (dream.py:51)
    assert np.all(np.abs(cr_counts / cr_counts.sum() - 1 / 3) < 0.01)
    # P(z <= CR_m) = CR_m; with CR = 1/3 a chain whose mask came out empty gets one forced dimension (dream.py:55-57)
    p13 = 1 / 3 + (2 / 3) ** d / d
    assert abs(np.mean(mask_rate[0]) - p13) < 0.01 and abs(np.mean(mask_rate[1]) - 2 / 3) < 0.01 and np.mean(mask_rate[2]) == 1.0
    # gamma = 1 with probability 0.8 on every 5th generation (dream.py:77-80)
    assert abs(jump_on / jump_n - 0.8) < 0.015
    # Bernoulli(alpha) acceptance (samplers.py:334-336): accepted fraction == mean alpha
    assert abs(acc / n_upd - alpha_sum / n_upd) < 0.01
    # partners uniform over chains: chi-square-like spread
    exp = partner_hist.mean()
    assert abs(partner_hist.std() / np.sqrt(exp) - 1.0) < 0.1


def test_demc_decision_frequencies():
    N, G = 4096, 40
    e = _hooks_engine(algo=R.ALGO_DEMC, n_chains=N, dim=2, target_id=R.TARGET_BANANA_2D, target_params=R.banana_params(), seed=321,
                p_snooker=0.1)
    e.set_state(np.random.RandomState(1).normal(size=(N, 2)) + np.array([0, 1.0]))
    e.set_trace(True)
    e.begin_run()
    snk, jump_on, jump_n, n_upd = 0, 0, 0, 0
    for g in range(G):
        e.step(1)
        tr = e.get_trace()
        snk += tr["snooker"].sum(); n_upd += N
        if g % 10 == 0:
            jump_on += tr["jump"].sum(); jump_n += N
        else:
            assert not tr["jump"].any()
        P_ = tr["partners"][:, :5]
        assert np.all(P_[:, 0] != P_[:, 1])
        assert np.all((P_[:, 2] != P_[:, 3]) & (P_[:, 2] != P_[:, 4]) & (P_[:, 3] != P_[:, 4]))
    assert abs(snk / n_upd - 0.1) < 0.006                                   # snooker with probability p_snooker
    assert abs(jump_on / jump_n - 0.9) < 0.01                               # demc.py:174-177


def test_jitter_moments_on_device():
    """e_u ~ U(-u_eps, u_eps) multiplies, e_n ~ N(0, eps^2) adds (util.py:5-28): recover both from one update of a
    population whose pairs are (0, 1)-valued so that the pair sum is exactly known."""
    N, d = 4096, 16
    params = R.gauss_equicorr_params(0.0, np.full(d, 1e6))                  # flat target: everything is accepted
    e = _engine(algo=R.ALGO_DEMC, n_chains=N, dim=d, target_id=R.TARGET_GAUSS_EQUICORR, target_params=params, seed=9)
    X = np.zeros((N, d))
    e.set_state(X)
    e.begin_run(epsilon=0.5)
    order, flip = P.shuffle_idx(9, 0, N), P.flip_draw(9, 0, 0.5)
    e.step(1)
    first_group = order[N // 2:] if flip else order[:N // 2]                # updated against a pool that is still all zeros
    X1 = e.get_state()[first_group]                                         # x' = 0 + gamma*(0 - 0) + e_n
    assert abs(X1.mean()) < 0.006 and abs(X1.std() - 0.5) < 0.006
    from scipy import stats
    assert stats.kstest((X1 / 0.5).reshape(-1)[:20000], "norm").pvalue > 1e-3
    # uniform multiplicative jitter of DREAM: pairs differ by exactly 1 in every coordinate
    e = _hooks_engine(algo=R.ALGO_DREAM, n_chains=N, dim=d, target_id=R.TARGET_GAUSS_EQUICORR, target_params=params, seed=10,
                burnin_gen=0, n_cr=1, del_pairs=1)
    X = np.zeros((N, d)); X[::2] = 1.0                                      # a - b in {-1, 0, 1}
    e.set_state(X)
    e.set_trace(True)
    e.begin_run(epsilon=0.0, u_epsilon=0.05)
    order, flip = P.shuffle_idx(10, 0, N), P.flip_draw(10, 0, 0.5)
    e.step(1)
    first_group = np.zeros(N, dtype=bool)
    first_group[order[N // 2:] if flip else order[:N // 2]] = True          # their pool still held the initial states
    tr = e.get_trace()
    X1 = e.get_state()
    ab = X[tr["partners"][:, 0]] - X[tr["partners"][:, 1]]
    sel = first_group & (np.abs(ab[:, 0]) == 1) & (tr["accepted"] == 1)
    assert sel.sum() > 500
    ratio = ((X1 - X)[sel] / (tr["gamma"][sel, None] * ab[sel]))            # = 1 + e_u   (n_cr = 1: every coordinate moves)
    assert abs(ratio.mean() - 1.0) < 0.002 and ratio.min() >= 0.95 and ratio.max() <= 1.05
    assert abs(ratio.std() - 0.05 / np.sqrt(3)) < 0.001


def _equicorr_draws(N, d, rho, seed):
    rs = np.random.RandomState(seed)
    return np.sqrt(np.arange(d) + 1.0) * (np.sqrt(rho) * rs.standard_normal((N, 1)) + np.sqrt(1.0 - rho) * rs.standard_normal((N, d)))


@pytest.mark.parametrize("d", [3, 8, 100])
def test_snooker_only_sampler_leaves_the_gaussian_invariant(d):
    """VERDICT r03 next 5(a).  The snooker update (ter Braak & Vrugt 2008) is an extension the reference cannot pin; at p_snooker = 0.1 a wrong Jacobian
    term 0.5 (d - 1) (ln|x' - z|^2 - ln|x - z|^2) is diluted ten-fold, and at d = 2 the exponent is 1/2 -- its dependence on d was never tested.  Here
    EVERY update is a snooker update (p_snooker = 1.0), the population starts as exact draws of the equicorrelated Gaussian (rho = 0.5, sigma_i^2 = i + 1)
    and must stay there: over 500 generations of 8192 chains the pooled variance within 1 %, every mean within 0.05 sigma, at d = 3, 8 and 100 (one lane,
    four lanes, one wavefront per chain).  An exponent of d instead of d - 1 shifts the stationary radial law by a factor |x - z| -- at d = 3 the variance by
    tens of per cent.  (d = 100: 32768 chains -- with rho = 0.5 the pooled variance of a SAMPLE of 8192 exact draws already scatters by 0.8 %, the common
    factor carries a quarter of every coordinate's variance and a line move per update mixes 100 dimensions slowly: 8192 chains gave 0.9899, the start's own
    sampling error (r0 below: the start's pooled variance ratio, reported with a failure).)"""
    N, G, rho = (32768 if d == 100 else 8192), 500, 0.5
    params = R.gauss_equicorr_params(rho, np.sqrt(np.arange(d) + 1.0))
    e = _engine(algo=R.ALGO_DEMC, n_chains=N, dim=d, target_id=R.TARGET_GAUSS_EQUICORR, target_params=params, seed=2026 + d, p_snooker=1.0)
    X0 = _equicorr_draws(N, d, rho, 77 + d)
    r0 = float(np.mean(X0.var(axis=0) / (np.arange(d) + 1.0)))     # the start's own pooled variance ratio (sampling error of N exact draws)
    e.set_state(X0)
    e.reserve_history(G + 2)
    e.begin_run()
    e.step(G)
    e.synchronize()
    cnt, s1, s2, sh = e.reduce_moments(N)                         # every row behind the start
    st = e.stats()
    assert cnt == G * N
    mean, var = sh + s1 / cnt, s2 / cnt - (s1 / cnt) ** 2
    sig2 = np.arange(d) + 1.0
    acc = st["local_n_accepted"] / float(st["local_n_accepted"] + st["local_n_rejected"])
    assert 0.02 < acc < 0.9, acc                                   # the chains do move
    assert abs(np.mean(var / sig2) - 1.0) < 0.01, (d, np.mean(var / sig2), r0, acc)
    assert np.max(np.abs(mean) / np.sqrt(sig2)) < 0.05, (d, np.max(np.abs(mean) / np.sqrt(sig2)))
    # ... and the correlation structure: the variance of the standardised coordinates' sum is d (1 + (d - 1) rho)
    X = e.get_state() / np.sqrt(sig2)
    assert abs(np.var(X.sum(axis=1)) / (d * (1 + (d - 1) * rho)) - 1.0) < 0.08
    e.close()


def test_cfg2_from_the_references_own_start_reaches_the_posterior_gate():
    """the moment gate of bench.py tests STATIONARITY (start = exact draws).  Here BASELINE config 2 starts where the reference
    starts it -- theta_0 = 0, varepsilon = 1e-6: 8192 chains within 1e-3 of the origin (bipymc/chain.py:25-27, tests/test_100dgauss.py:67-69)
    -- and must GET there: the trailing 1000 generations pass the gate (pooled variance ratio within 1 %, every |mean| < 0.05 sigma) by generation 3000
    (measured: 2250 for seeds 42 and 7, profiles/r04_convergence_from_reference_start.txt -- the figure DESIGN.md section 7 quotes) and the window ending at
    generation 4000 passes too.  Population sums per generation (running_moments), no history."""
    from bipymc_amd.utils import d100_gauss
    N, d, window, step = 8192, 100, 1000, 250
    tid, tp, _ = d100_gauss.Gauss_100D(rho=0.5, dim=d)._bpm_target_spec()
    e = _engine(algo=R.ALGO_DREAM, n_chains=N, dim=d, target_id=tid, target_params=tp, seed=42, burnin_gen=200, n_cr_gen=50, keep_history=False,
                running_moments=True)
    e.init_chains(np.zeros(d), 1e-6)
    x0 = e.get_state()
    assert np.abs(x0).max() < 1e-2 and 5e-4 < x0.std() < 2e-3          # the reference's start: N(0, 1e-6 I) around theta_0 = 0
    e.begin_run()
    sig2 = np.arange(d) + 1.0
    passed, T = {}, 0
    while T < 4000:
        e.step(step)
        T += step
        if T < window:
            continue
        cnt, s1, s2, sh = e.reduce_moments((1 + T - window) * N)
        assert cnt == window * N
        mean, var = sh + s1 / cnt, s2 / cnt - (s1 / cnt) ** 2
        vr, mm = float(np.mean(var / sig2)), float(np.max(np.abs(mean) / np.sqrt(sig2)))
        passed[T] = (abs(vr - 1.0) < 0.01 and mm < 0.05, vr, mm)
    e.close()
    first = min([t for t, p in passed.items() if p[0]] or [10 ** 9])
    assert first <= 3000, passed
    assert passed[4000][0], passed[4000]
    assert passed[1000][1] < 0.9, passed[1000]                          # (and it did start far away: the first 1000 generations are NOT the posterior)
