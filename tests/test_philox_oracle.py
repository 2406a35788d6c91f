"""RNG layer of the oracle: Philox4x32-10 known answers, draw conversions, keyed bijection."""
import json
import os

import numpy as np

from oracle import philox_ref as P


def test_random123_known_answers():
    # Random123 kat_vectors (philox4x32 10 rounds): counter, key -> output
    kats = [
        ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
        ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
        ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
         [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
    ]
    for c, k, o in kats:
        out = P.philox4x32_10(np.array(c, dtype=np.uint32), np.array(k, dtype=np.uint32))
        assert out.tolist() == o


def test_rocrand_host_engine_vectors(golden_dir):
    k = json.load(open(os.path.join(golden_dir, "philox_kat_rocrand.json")))
    assert len(k["rocrand"]) >= 8
    for q in k["rocrand"]:
        off = int(q["offset"])
        out = P.block(int(q["seed"]), int(q["subseq"]), off // 4)
        assert out.tolist() == q["out"]


def test_vectorised_equals_scalar():
    ids = np.arange(50)
    blk = P.chain_block(42, ids[:, None], 7, P.SLOT_DIM0 + np.arange(5)[None, :])
    for i in (0, 13, 49):
        for j in range(5):
            assert blk[i, j].tolist() == P.chain_block(42, i, 7, P.SLOT_DIM0 + j).tolist()


def test_conversions():
    assert P.u01_32(0) == 0.0 and P.u01_32(0xFFFFFFFF) < 1.0
    assert P.u01_53(0xFFFFFFFF, 0xFFFFFFFF) == 1.0 - 2.0 ** -53
    assert P.mulhi(0xFFFFFFFF, 10) == 9 and P.mulhi(0, 10) == 0
    w = np.random.RandomState(0).randint(0, 2 ** 32, size=(2, 200000), dtype=np.uint64).astype(np.uint32)
    ia, ib = P.distinct_pair(w[0], w[1], 5)
    assert np.all(ia != ib) and ia.min() == 0 and ia.max() == 4 and ib.min() == 0 and ib.max() == 4
    # every ordered pair equally likely
    cnt = np.zeros((5, 5))
    np.add.at(cnt, (ia, ib), 1)
    off = cnt[~np.eye(5, dtype=bool)]
    assert np.all(np.abs(off / off.mean() - 1) < 0.05)
    n = P.box_muller(w[0], w[1])
    assert abs(n.mean()) < 0.01 and abs(n.std() - 1) < 0.01 and np.all(np.isfinite(n))
    assert np.isfinite(P.box_muller(np.uint32(0xFFFFFFFF), np.uint32(0)))
    assert np.isfinite(P.box_muller(np.uint32(0), np.uint32(0)))


def test_bijection_and_inverse():
    for n in (1, 2, 3, 4, 5, 7, 10, 64, 100, 1000, 8192, 65536, 100003):
        for t in (0, 1, 99):
            keys = P.shuffle_keys(42, t)
            p = P.perm(np.arange(n), n, keys)
            assert np.array_equal(np.sort(p), np.arange(n))
            assert np.array_equal(P.perm_inv(p, n, keys), np.arange(n))


def test_partition_is_well_mixed():
    """Co-membership of two chains in the same half, over generations, is ~ (h-1)/(N-1)."""
    N, T = 16, 4000
    same = np.zeros((N, N))
    first = np.zeros(N)
    for t in range(T):
        order = P.shuffle_idx(7, t, N)
        half = np.zeros(N, dtype=bool)
        half[order[:N // 2]] = True
        same += (half[:, None] == half[None, :])
        first += half
    same /= T
    expect = (N // 2 - 1) / (N - 1)
    off = same[~np.eye(N, dtype=bool)]
    assert abs(off.mean() - expect) < 0.01
    assert off.min() > expect - 0.06 and off.max() < expect + 0.06
    assert np.all(np.abs(first / T - 0.5) < 0.04)
    flips = np.mean([P.flip_draw(7, t, 0.5) for t in range(T)])
    assert abs(flips - 0.5) < 0.03
    assert not P.flip_draw(7, 0, 0.0) and P.flip_draw(7, 0, 1.0)


def test_engine_draw_layout_is_pinned(golden_dir):
    """Regression pin of the engine-level oracle (draw layout v3: slots, 16-bit fields, float32 Box-Muller,
    Feistel shuffle): tests/golden/engine_layout_v3.npz was written by this very code; any change of the
    layout must be deliberate (regenerate the fixture and bump the layout version in DESIGN.md section 4)."""
    import os
    from oracle import sampler_ref as R
    g = np.load(os.path.join(golden_dir, "engine_layout_v3.npz"))
    p = R.gauss_equicorr_params(0.5, np.sqrt(np.arange(6) + 1.0))
    s = R.OracleSampler(R.ALGO_DREAM, 12, 6, R.TARGET_GAUSS_EQUICORR, p, seed=2024, burnin_gen=6, n_cr_gen=2)
    s.init_jitter(np.linspace(-1, 1, 6), 1e-2)
    s.run(10)
    np.testing.assert_allclose(s.X, g["dream_state"], rtol=1e-13)
    np.testing.assert_allclose(s.cr.p_cr, g["dream_p_cr"], rtol=1e-12)
    assert s.local_n_accepted == int(g["dream_acc"][0])
    s = R.OracleSampler(R.ALGO_DEMC, 9, 2, R.TARGET_BANANA_2D, R.banana_params(), seed=2025, p_snooker=0.3)
    s.init_jitter(np.zeros(2), 1e-1)
    s.run(12)
    np.testing.assert_allclose(s.X, g["demc_state"], rtol=1e-13)
    assert s.local_n_accepted == int(g["demc_acc"][0])
    s = R.OracleSampler(R.ALGO_DEMC_SYNC, 8, 3, R.TARGET_GAUSS_EQUICORR, R.gauss_equicorr_params(0.5, np.sqrt(np.arange(3) + 1.0)), seed=2026)
    s.init_jitter(np.zeros(3), 1e-1)
    s.run(12, epsilon=1e-3)
    np.testing.assert_allclose(s.X, g["sync_state"], rtol=1e-13)
    assert s.local_n_accepted == int(g["sync_acc"][0])
