"""The launch path that SHIPS -- no trace, the library's own AQL queue, acquire-only packets with agent-scope stores of what the
next kernel reads, the specialised (`HOT`) kernel instantiations -- compared DIRECTLY with the CPU oracle at the sizes BASELINE.json
names (VERDICT r02 "what's weak" item 1: until now the oracle met the traced, general kernel at N <= 64 and the shipped path met
the stream path: a self-comparison).

Reference arithmetic restated by the oracle: bipymc/dream.py:32-140 (DREAM update + CR adaptation), bipymc/demc.py:63-151,153-196
(generation driver, DE-MC update), bipymc/samplers.py:328-336 (Metropolis).  Integer outcomes (accept counts) must be equal; floats
after G generations: rtol 1e-10 / atol 1e-12 (libm exp/log, reduction order and the float32 Box-Muller of the 1e-12-scale jitter;
one generation agrees to 1e-12, tests/test_gpu_parity.py).
"""
import numpy as np
import pytest

from oracle import sampler_ref as R

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-10, 1e-12


def _run_both(algo, N, d, target_id, params, seed, X0, gens, okw, hist_rows):
    from bipymc_amd.engine import HipEngine
    eng = HipEngine(algo=algo, n_chains=N, dim=d, target_id=target_id, target_params=params, seed=seed, **okw)
    ora = R.OracleSampler(algo, N, d, target_id, params, seed, **okw)
    ls0 = eng.launch_stats()
    assert ls0["has_queue"], "no direct AQL queue on this box: " + str(ls0)
    assert ls0["fence"] == "acquire" and not ls0["coherent_state"], ls0
    eng.set_state(X0)
    ora.set_state(X0)
    eng.begin_run()
    eng.step(gens)                         # NO trace: phase_args_hot() holds, the HOT instantiations run
    eng.synchronize()
    ls = eng.launch_stats()
    assert ls["direct"] - ls0["direct"] == 2 * gens and ls["stream"] == ls0["stream"], (ls0, ls)   # every update kernel through the own queue
    ora.run(gens)
    st = eng.stats()
    # ---- integers: exact
    assert st["local_n_accepted"] == ora.local_n_accepted, (st["local_n_accepted"], ora.local_n_accepted)
    assert st["local_n_rejected"] == ora.local_n_rejected
    assert st["n_nan_alpha"] == ora.n_nan == 0
    # ---- floats
    X = eng.get_state()
    np.testing.assert_allclose(X, ora.X, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(eng.get_loglike(), ora.ll, rtol=RTOL, atol=1e-9)
    for g in hist_rows:                    # two history rows, straight out of the device history
        np.testing.assert_allclose(eng.get_history(g, g + 1)[0], ora.history[g], rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(eng.get_loglike_history(g, g + 1)[0], ora.ll_history[g], rtol=RTOL, atol=1e-9)
    if algo == R.ALGO_DREAM:
        np.testing.assert_allclose(st["p_cr"], ora.cr.p_cr, rtol=1e-9)
        np.testing.assert_allclose(st["delta_m"], ora.cr.delta_m, rtol=1e-6)        # divides by per-chain history variances: see test_gpu_parity
        assert np.array_equal(st["n_cr_updates"], ora.cr.n_cr_updates)
    frac_bit_equal = float(np.mean(X == ora.X))
    eng.close()
    return frac_bit_equal


def test_cfg2_shape_on_the_shipped_path_equals_the_oracle():
    """BASELINE configs[1]: DREAM, 100-D Gaussian, N = 8192, del_pairs 3, n_cr 3 -- 4 burn-in generations (CR adaptation: HOT = 3,
    gate open from generation 3) + 8 steady-state ones (HOT = 1: plan records, write-through stores, acquire-only packets)."""
    N, d = 8192, 100
    params = R.gauss_equicorr_params(0.5, np.sqrt(np.arange(d) + 1.0))
    rs = np.random.RandomState(11)
    X0 = np.sqrt(np.arange(d) + 1.0) * (np.sqrt(0.5) * rs.standard_normal((N, 1)) + np.sqrt(0.5) * rs.standard_normal((N, d)))
    f = _run_both(R.ALGO_DREAM, N, d, R.TARGET_GAUSS_EQUICORR, params, 42, X0, 12,
                  dict(del_pairs=3, n_cr=3, burnin_gen=4, n_cr_gen=2), hist_rows=(3, 12))
    assert f > 0.5          # (most entries are bit-identical; the rest differ in the last ulps)


def test_cfg3_shape_on_the_shipped_path_equals_the_oracle():
    """BASELINE configs[2]: DE-MC, banana, N = 65536, snooker probability 0.1 (extension: oracle parity only) -- 11 generations, so
    that k = 0 and k = 10 (gamma = 1 jumps w.p. 0.9, demc.py:174-177) are both inside."""
    N = 65536
    bp = R.banana_params()
    rs = np.random.RandomState(5)
    X0 = rs.normal(size=(N, 2)) + np.array([0.0, 1.0])
    _run_both(R.ALGO_DEMC, N, 2, R.TARGET_BANANA_2D, bp, 7, X0, 11, dict(p_snooker=0.1), hist_rows=(1, 11))


def test_cfg5_share_on_the_shipped_path_equals_the_oracle():
    """BASELINE configs[4], one GPU's share: DREAM, 8-D pairwise mixture (extension of dblgauss_rv.py:11-32), N = 32768 -- 3 burn-in
    generations with CR adaptation + 7 steady-state ones (4 lanes per chain, HOT = 4 / 2)."""
    N, d = 32768, 8
    mp = R.mixture_pairs_params(0.25, 0.75, [0, 0], [2, 2], [0.25, 0.25], [0.25, 0.25], 0.8, -0.8)
    rs = np.random.RandomState(9)
    X0 = np.where(rs.uniform(size=(N, 1)) < 0.25, 0.0, 2.0) + 0.25 * rs.normal(size=(N, d))
    _run_both(R.ALGO_DREAM, N, d, R.TARGET_MIXTURE_PAIRS, mp, 3, X0, 10,
              dict(del_pairs=3, n_cr=3, burnin_gen=3, n_cr_gen=1), hist_rows=(2, 10))


def test_cfg4_and_cfg5_total_populations_on_one_gpu_equal_the_oracle():
    """BASELINE configs[3] and configs[4] name 65536 chains at d = 100 and 262144 chains at d = 8 (sharded over 8 GPUs there; the sharded runs are pinned
    to the one-rank run bit for bit in tests/test_gpu_api.py and tests/test_gpu_push.py).  Here the ONE-rank run of those total populations meets the oracle
    directly: the in-kernel draw path of one wavefront per chain (> 16384 chains: no plan records), the lean form of four lanes per chain
    (>= 49152 chains), level 2 of the CR reduction (cr_mid_kernel) and the outlier check at full size."""
    N, d = 65536, 100
    params = R.gauss_equicorr_params(0.5, np.sqrt(np.arange(d) + 1.0))
    rs = np.random.RandomState(12)
    X0 = np.sqrt(np.arange(d) + 1.0) * (np.sqrt(0.5) * rs.standard_normal((N, 1)) + np.sqrt(0.5) * rs.standard_normal((N, d)))
    _run_both(R.ALGO_DREAM, N, d, R.TARGET_GAUSS_EQUICORR, params, 42, X0, 6, dict(del_pairs=3, n_cr=3, burnin_gen=3, n_cr_gen=1), hist_rows=(2, 6))
    N, d = 262144, 8
    mp = R.mixture_pairs_params(0.25, 0.75, [0, 0], [2, 2], [0.25, 0.25], [0.25, 0.25], 0.8, -0.8)
    rs = np.random.RandomState(13)
    X0 = np.where(rs.uniform(size=(N, 1)) < 0.25, 0.0, 2.0) + 0.25 * rs.normal(size=(N, d))
    X0[:5] = 30.0                                                         # a few chains in the far tail for the outlier check
    _run_both(R.ALGO_DREAM, N, d, R.TARGET_MIXTURE_PAIRS, mp, 3, X0, 8, dict(del_pairs=3, n_cr=3, burnin_gen=6, n_cr_gen=1, outlier_every=4), hist_rows=(3, 8))


@pytest.mark.parametrize("N,d,tgt", [(40000, 8, "mix"), (70001, 2, "gauss"), (8195, 100, "gauss"), (5000, 20, "gauss"), (4099, 300, "gauss")])
def test_cr_statistics_summed_inside_the_update_kernels_equal_the_oracle(N, d, tgt):
    """Round 4: level 1 of the CR reduction (dream.py:119-140) is written by the burn-in flavours of the update kernel themselves -- a wavefront's 16 / 64
    chains (d = 8 / d = 2) or a workgroup of 16 wavefronts (d = 100) per chunk of positions -- cr_mid_kernel passes fold more than 1024 chunks
    (N = 40000 at d = 8: 2500; N = 70001 at d = 2: 1095), cr_final_kernel finishes; odd populations (a ragged last chunk in both halves); d = 20
    (4 chains per wavefront) and d = 300 (one wavefront per chain with 8 coordinates per lane: level 1 from the slots).  Against the oracle over 7
    burn-in generations, on the shipped path (own queue, no trace)."""
    if tgt == "mix":
        params = R.mixture_pairs_params(0.25, 0.75, [0, 0], [2, 2], [0.25, 0.25], [0.25, 0.25], 0.8, -0.8)
        tid = R.TARGET_MIXTURE_PAIRS
        rs = np.random.RandomState(9)
        X0 = np.where(rs.uniform(size=(N, 1)) < 0.25, 0.0, 2.0) + 0.25 * rs.normal(size=(N, d))
    else:
        params = R.gauss_equicorr_params(0.5, np.sqrt(np.arange(d) + 1.0))
        tid = R.TARGET_GAUSS_EQUICORR
        X0 = np.random.RandomState(4).normal(size=(N, d)) * np.sqrt(np.arange(d) + 1.0)
    _run_both(R.ALGO_DREAM, N, d, tid, params, 31, X0, 7, dict(del_pairs=3, n_cr=3, burnin_gen=100, n_cr_gen=1), hist_rows=(2, 7))


@pytest.mark.parametrize("N,d", [(300, 20), (500, 8)])
def test_general_kernel_on_the_own_queue_equals_the_oracle(N, d):
    """the GENERAL instantiation of the update kernel (four CR values: no specialised one) of the shapes with 16 / 4 lanes per chain declares a
    private segment of 36 bytes without executing a scratch instruction (tests/test_abi.py reads the disassembly); the library's own queue dispatches it with
    that size in the packet.  It has to run there -- own queue, no trace -- and equal the oracle like every other kernel."""
    params = R.gauss_equicorr_params(0.5, np.sqrt(np.arange(d) + 1.0))
    X0 = np.random.RandomState(8).normal(size=(N, d)) * np.sqrt(np.arange(d) + 1.0)
    _run_both(R.ALGO_DREAM, N, d, R.TARGET_GAUSS_EQUICORR, params, 17, X0, 9, dict(del_pairs=2, n_cr=4, burnin_gen=5, n_cr_gen=2), hist_rows=(3, 9))


@pytest.mark.parametrize("algo,N,d,shuffle,flip", [("demc", 77, 2, True, 0.5), ("demc", 10, 1, True, 0.3), ("demc", 129, 3, False, 1.0),
                                                    ("dream", 101, 8, True, 0.5), ("dream", 64, 5, False, 0.0), ("dream", 33, 17, True, 0.7),
                                                    ("dream", 12, 32, True, 0.5)])
def test_histories_appended_in_shuffle_order_come_back_in_chain_order(algo, N, d, shuffle, flip):
    """Fewer than 64 lanes per chain: the update kernels append a generation's history row (and its ln-likes) in that generation's shuffle
    order and bpm_get_history de-permutes it (sampler.hip: normalize_history).  Whole histories against the oracle's (chain.py:51-54:
    one row per generation, chain i at column block i) for odd and even populations, padded rows (odd d), shuffle off, every flip -- read
    in two pieces and out of order, then the run goes on and the rest is read."""
    from bipymc_amd.engine import HipEngine
    A = R.ALGO_DEMC if algo == "demc" else R.ALGO_DREAM
    params = R.gauss_equicorr_params(0.5, np.sqrt(np.arange(d) + 1.0)) if d > 1 else R.gauss_equicorr_params(0.0, np.ones(1))
    kw = dict(p_snooker=0.2) if algo == "demc" else dict(del_pairs=2, n_cr=3, burnin_gen=5, n_cr_gen=2)
    eng = HipEngine(algo=A, n_chains=N, dim=d, target_id=R.TARGET_GAUSS_EQUICORR, target_params=params, seed=9, **kw)
    ora = R.OracleSampler(A, N, d, R.TARGET_GAUSS_EQUICORR, params, 9, **kw)
    X0 = np.random.RandomState(2).normal(size=(N, d))
    eng.set_state(X0)
    ora.set_state(X0)
    eng.begin_run(flip=flip, shuffle=shuffle)
    eng.step(9)
    tail = eng.get_history(6, 10)                         # rows 6..9 first ...
    head = eng.get_history(0, 6)                          # ... then the older ones
    eng.step(4)
    rest = eng.get_history(10, 14)
    llh = eng.get_loglike_history(0, 14)
    ora.run(13, flip=flip, shuffle=shuffle)
    H = np.concatenate([head, tail, rest], axis=0)
    np.testing.assert_allclose(H, np.stack(ora.history, axis=0), rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(llh, np.stack(ora.ll_history, axis=0), rtol=RTOL, atol=1e-9)
    assert eng.stats()["local_n_accepted"] == ora.local_n_accepted
    eng.close()


@pytest.mark.parametrize("N,d,pairs", [(96, 130, 3), (40, 300, 2), (64, 64, 3), (50, 33, 1), (48, 640, 3), (40, 1024, 2), (44, 512, 3), (36, 514, 3)])
def test_wide_rows_on_the_shipped_path_equal_the_oracle(N, d, pairs):
    """One wavefront per chain with 2, 4 or 8 coordinates per lane (d = 33 ... 512) and the looped kernel beyond, on the shipped path -- own queue,
    acquire-only packets, 16-byte write-through stores, plan records -- against the oracle (dream.py:32-140), burn-in and steady state."""
    params = R.gauss_equicorr_params(0.3, np.sqrt(np.arange(d) + 1.0))
    X0 = np.random.RandomState(21).normal(size=(N, d)) * np.sqrt(np.arange(d) + 1.0)
    _run_both(R.ALGO_DREAM, N, d, R.TARGET_GAUSS_EQUICORR, params, 5, X0, 9, dict(del_pairs=pairs, n_cr=3, burnin_gen=4, n_cr_gen=2), hist_rows=(4, 9))


@pytest.mark.parametrize("algo,N,d,kw", [
    (R.ALGO_DREAM, 24, 1800, dict(del_pairs=3, n_cr=3, burnin_gen=4, n_cr_gen=2)),
    (R.ALGO_DREAM, 16, 4096, dict(del_pairs=3, n_cr=3, burnin_gen=4, n_cr_gen=2)),
    (R.ALGO_DREAM, 12, 10000, dict(del_pairs=3, n_cr=3, burnin_gen=3, n_cr_gen=1)),
    (R.ALGO_DREAM, 21, 1025, dict(del_pairs=2, n_cr=4, burnin_gen=4, n_cr_gen=2)),          # odd d (a padded row), pairs at run time, four CR values
    (R.ALGO_DREAM, 33, 513, dict(del_pairs=3, n_cr=3, burnin_gen=4, n_cr_gen=2)),            # the first width on the looped kernel
    (R.ALGO_DREAM, 10, 2047, dict(del_pairs=7, n_cr=1, burnin_gen=0)),                       # CR = 1 only: no counting pass
    (R.ALGO_DEMC, 30, 1500, dict(p_snooker=0.3)),
    (R.ALGO_DEMC, 9, 3001, dict(p_snooker=0.0)),
])
def test_looped_wide_row_kernel_on_the_shipped_path_equals_the_oracle(algo, N, d, kw):
    """d > 512: one wavefront per chain LOOPING over its row in chunks of 256 coordinates (kernels_wide.h) -- no dimension
    limit, no scratch memory, so it runs on the library's own queue like every other shape (round 3: d <= 2048 only, the widest shape spilled and ran on
    the HIP stream).  The reference has no limit (dream.py:52-58,61,85-89 work on self.dim).  Against the oracle as everywhere: accept counts equal,
    state / whole history / p_cr to 1e-10, burn-in with CR adaptation and steady state; d = 4096 and 10 000 included."""
    from bipymc_amd.engine import HipEngine
    params = R.gauss_equicorr_params(0.3, np.sqrt(np.arange(d) % 50 + 1.0))
    eng = HipEngine(algo=algo, n_chains=N, dim=d, target_id=R.TARGET_GAUSS_EQUICORR, target_params=params, seed=5, **kw)
    ora = R.OracleSampler(algo, N, d, R.TARGET_GAUSS_EQUICORR, params, 5, **kw)
    X0 = np.random.RandomState(21).normal(size=(N, d)) * np.sqrt(np.arange(d) % 50 + 1.0)
    eng.set_state(X0)
    ora.set_state(X0)
    ls0 = eng.launch_stats()
    eng.begin_run()
    G = 11 if algo == R.ALGO_DEMC else 9                  # (DE-MC: k = 0 and k = 10, the gamma = 1 generations of demc.py:174-177)
    eng.step(G)
    eng.synchronize()
    ora.run(G)
    ls = eng.launch_stats()
    assert ls["direct"] - ls0["direct"] == 2 * G and ls["stream"] == ls0["stream"], (ls0, ls)      # the library's own queue
    st = eng.stats()
    assert st["local_n_accepted"] == ora.local_n_accepted and st["local_n_rejected"] == ora.local_n_rejected
    np.testing.assert_allclose(eng.get_state(), ora.X, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(eng.get_history(), np.stack(ora.history, axis=0), rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(eng.get_loglike(), ora.ll, rtol=1e-9, atol=1e-7)
    if algo == R.ALGO_DREAM:
        # (p_cr: ratios of sums over >= 500 coordinates of (jump / history std)^2 after a handful of history rows -- the running Welford moments
        # and NumPy's two-pass std agree to ~1e-9 there)
        np.testing.assert_allclose(st["p_cr"], ora.cr.p_cr, rtol=1e-7)
        assert np.array_equal(st["n_cr_updates"], ora.cr.n_cr_updates)
    eng.close()


def test_the_one_dimension_limit_is_the_philox_slot():
    """A coordinate pair's draws are addressed by a 16-bit slot (philox.h: SLOT_BITS): dim < 131056; no other limit (memory aside)."""
    from bipymc_amd import _lib as L
    from bipymc_amd.engine import HipEngine
    with pytest.raises(L.BpmError, match="below 131056"):
        HipEngine(algo=R.ALGO_DEMC, n_chains=4, dim=131056, target_id=R.TARGET_GAUSS_EQUICORR,
                  target_params=R.gauss_equicorr_params(0.3, np.ones(131056)), seed=5)
    e = HipEngine(algo=R.ALGO_DEMC, n_chains=4, dim=131055, target_id=R.TARGET_GAUSS_EQUICORR,
                  target_params=R.gauss_equicorr_params(0.0, np.ones(131055)), seed=5)
    e.set_state(np.zeros((4, 131055)))
    e.begin_run()
    e.step(2)
    assert np.all(np.isfinite(e.get_state()))
    e.close()


def _target_case(kind, N, d, rs):
    """(target id, parameter block, start state) of one of the three device targets"""
    if kind == "gauss":
        return (R.TARGET_GAUSS_EQUICORR, R.gauss_equicorr_params(float(rs.choice([0.0, 0.5, 0.9])), np.sqrt(np.arange(d) % 50 + 1.0)),
                rs.normal(size=(N, d)) * np.sqrt(np.arange(d) % 50 + 1.0))
    if kind == "mix":
        return (R.TARGET_MIXTURE_PAIRS, R.mixture_pairs_params(0.25, 0.75, [0, 0], [2, 2], [0.25, 0.25], [0.25, 0.25], 0.8, -0.8),
                np.where(rs.uniform(size=(N, 1)) < 0.25, 0.0, 2.0) + 0.25 * rs.normal(size=(N, d)))
    return R.TARGET_BANANA_2D, R.banana_params(), rs.normal(size=(N, 2)) * 1.1 + np.array([0.0, 1.1])


# ---- every entry of the dispatch tables against the oracle -----------------------------------------------------------------------------
# bipymc_amd/csrc/sampler.hip: g_fused_gauss[3][7], g_fused_mixture[3][7], g_fused_banana[3] -- rows = [DE-MC (one pair) | DREAM del_pairs = 3
# (compile-time) | DREAM any del_pairs], columns = kernel shape by row width (pick_shape): 0: d <= 2 one lane per chain, 1: d <= 8 four lanes,
# 2: d <= 32 sixteen lanes, 3: d <= 128 one wavefront, 4: d <= 256 two pairs per lane, 5: d <= 512 four pairs per lane, 6: the looped wide-row
# kernel.  VERDICT r04 weak 1: DREAM x banana had no GPU test at all and DE-MC x mixture none against the oracle.
#
#   instantiation                                   -> test
#   g_fused_gauss[v][shape]   v = 0, 1, 2; 0..6     -> test_every_dispatch_table_entry_equals_the_oracle[gauss-*]   (+ cfg2 / the wide-row / random tests above)
#   g_fused_mixture[v][shape] v = 0, 1, 2; 0..6     -> test_every_dispatch_table_entry_equals_the_oracle[mix-*]     (+ cfg5's share; DE-MC: v = 0 was statistical only)
#   g_fused_banana[v]         v = 0, 1, 2           -> test_every_dispatch_table_entry_equals_the_oracle[banana-*]  (+ cfg3; DREAM: v = 1, 2 were untested)
# per-generation trace parity of the two combinations that had none: tests/test_gpu_parity.py::test_dream_banana_generation_parity,
# ::test_demc_mixture_generation_parity; the reference's own scenarios (tests/test_banana.py:123-127, tests/test_dblgauss.py:130-133):
# tests/test_gpu_api.py::test_dream_banana_reference_scenario, ::test_demc_bimodal_and_banana_reference_scenarios.
_SHAPE_DIMS = {0: 2, 1: 8, 2: 20, 3: 100, 4: 200, 5: 400, 6: 600}
_TABLE = [(kind, v, sh) for kind in ("gauss", "mix") for v in (0, 1, 2) for sh in range(7)] + [("banana", v, 0) for v in (0, 1, 2)]


@pytest.mark.parametrize("kind,v,shape", _TABLE, ids=["%s-v%d-shape%d" % t for t in _TABLE])
def test_every_dispatch_table_entry_equals_the_oracle(kind, v, shape):
    """One case per entry of sampler.hip's g_fused_* tables (see the table above): DE-MC with snooker 0.2 over k = 0 ... 10 (both gamma = 1
    generations, demc.py:174-177), DREAM with del_pairs 3 (compile-time pair count) and 2 (run-time), 3 burn-in generations with CR adaptation
    (dream.py:92,119-140: the burn-in flavours) + 6 steady ones (the HOT flavours) -- on the shipped path, against the oracle."""
    d = 2 if kind == "banana" else _SHAPE_DIMS[shape]
    rs = np.random.RandomState(100 * v + shape + (0 if kind == "gauss" else 50 if kind == "mix" else 90))
    if v == 0:
        algo, N, gens, kw = R.ALGO_DEMC, (48 if d > 128 else 130), 11, dict(p_snooker=0.2)
    else:
        pairs = 3 if v == 1 else 2
        algo, N, gens, kw = R.ALGO_DREAM, (40 if d > 128 else 96), 9, dict(del_pairs=pairs, n_cr=3, burnin_gen=3, n_cr_gen=1)
    tid, params, X0 = _target_case(kind, N, d, rs)
    _run_both(algo, N, d, tid, params, 300 + 10 * v + shape, X0, gens, kw, hist_rows=(2, gens))


def _random_case(rs):
    """a small random configuration of the path: algorithm, target (every algorithm x target combination), population, dimension, pair count,
    CR values, burn-in, snooker, outlier check"""
    dream = rs.rand() < 0.6
    if dream:
        kind = rs.choice(["gauss", "mix", "banana"], p=[0.45, 0.3, 0.25])
        d = (int(rs.choice([1, 2, 3, 5, 8, 9, 16, 31, 32, 33, 64, 100, 129, 200, 513])) if kind == "gauss" else
             int(rs.choice([2, 4, 6, 8, 12, 30])) if kind == "mix" else 2)
        n_cr = int(rs.choice([1, 2, 3, 3, 3, 4, 8]))
        pairs = int(rs.choice([1, 2, 3, 3, 3, 4, 7]))
        N = int(rs.choice([4 * pairs + 4, 24, 50, 97, 256, 1000]))
        N = max(N, 2 * (2 * pairs + 1) + 2)
        kw = dict(del_pairs=pairs, n_cr=n_cr, burnin_gen=int(rs.choice([0, 3, 100])), n_cr_gen=int(rs.choice([1, 2])))
        if kind != "gauss" and rs.rand() < 0.4 and kw["burnin_gen"] > 0:
            kw["outlier_every"] = 3
        algo = R.ALGO_DREAM
    else:
        kind = rs.choice(["banana", "gauss", "mix"], p=[0.3, 0.35, 0.35])
        d = 2 if kind == "banana" else (int(rs.choice([1, 2, 3, 8, 17, 100, 300, 600])) if kind == "gauss" else int(rs.choice([2, 4, 8, 30, 130, 514])))
        N = int(rs.choice([8, 13, 64, 257, 1000, 4096]))
        if d > 128:
            N = min(N, 257)
        kw = dict(p_snooker=float(rs.choice([0.0, 0.1, 0.5, 1.0])))
        algo = R.ALGO_DEMC
    tid, params, X0 = _target_case(kind, N, d, rs)
    return algo, N, d, tid, params, X0, kw


@pytest.mark.parametrize("seed", range(48))
def test_random_configurations_on_the_shipped_path_equal_the_oracle(seed):
    """Differential test over random small configurations (algorithm x target: all six combinations; 1 ... 600 dimensions, 1 ... 7 pairs, 1 ... 8 CR
    values, burn-in on / off / ending inside the run, snooker probabilities 0 ... 1, the outlier check every 3 generations): whatever kernel shape and
    flavour the library picks -- specialised or general instantiation, one lane / 4 / 16 lanes / one wavefront per chain, the looped wide-row kernel -- on
    its own queue, without a trace, against the oracle over 8 generations.  Integers exact, floats as in the rest of this file."""
    rs = np.random.RandomState(1000 + seed)
    algo, N, d, tid, params, X0, kw = _random_case(rs)
    _run_both(algo, N, d, tid, params, 77 + seed, X0, 8, kw, hist_rows=(1, 8))
