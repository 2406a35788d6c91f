"""The plain-C restatement (oracle/csrc/dream_ref.c: bench.py's CPU baseline) against the NumPy oracle:
same seeds, same draw layout -> same accept decisions, state equal to 1e-12."""
import numpy as np
import pytest

from oracle import dream_ref_c as CR
from oracle import sampler_ref as R


@pytest.mark.parametrize("N,d,P", [(16, 100, 3), (9, 7, 2), (64, 2, 3), (10, 33, 1)])
def test_c_restatement_equals_numpy_oracle(N, d, P):
    params = R.gauss_equicorr_params(0.5, np.sqrt(np.arange(d) + 1.0))
    X0 = np.random.RandomState(N + d).normal(size=(N, d)) * np.sqrt(np.arange(d) + 1.0)
    ora = R.OracleSampler(R.ALGO_DREAM, N, d, R.TARGET_GAUSS_EQUICORR, params, 77, del_pairs=P, burnin_gen=0)
    ora.set_state(X0)
    ora.run(12, flip=0.3)
    X = X0.copy()
    ll = R.ll_gauss_equicorr(X, params)
    acc, hist = CR.dream_run(X, ll, params, 77, 0, 0, 12, del_pairs=P, flip=0.3, keep_history=True, n_threads=2)
    assert acc == ora.local_n_accepted
    np.testing.assert_allclose(X, ora.X, rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(ll, ora.ll, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(hist, ora.history_array()[1:], rtol=1e-12, atol=1e-15)


def test_thread_count_does_not_change_results():
    d, N = 20, 64
    params = R.gauss_equicorr_params(0.5, np.sqrt(np.arange(d) + 1.0))
    X0 = np.random.RandomState(1).normal(size=(N, d))
    out = []
    for nt in (1, 4):
        X = X0.copy()
        ll = R.ll_gauss_equicorr(X, params)
        CR.dream_run(X, ll, params, 5, 0, 0, 8, n_threads=nt)
        out.append(X)
    assert np.array_equal(out[0], out[1])
