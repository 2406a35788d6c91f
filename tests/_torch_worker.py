"""Worker for the GPU tests that need PyTorch on the device beside the library (tests/test_gpu_parity.py, tests/test_gpu_api.py): run as a child process
that imports torch FIRST.  The torch wheel carries a HIP runtime of its own; a process can initialise only one, and the library binds to whichever is
loaded already -- the order a user's script has (`import torch` at the top, the sampler created later)."""
import json
import os
import sys

import torch                                            # noqa: F401  (first: see above)

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
import numpy as np                                      # noqa: E402
import pytest                                           # noqa: E402

from oracle import sampler_ref as R                     # noqa: E402


def device_parity(algo, d, N, kw):
    from bipymc_amd.engine import HipEngine
    params = R.gauss_equicorr_params(0.4, np.sqrt(np.arange(d) + 1.0))
    tp = torch.tensor(params, dtype=torch.float64, device="cuda")

    def py_ll(theta):
        return float(R.ll_gauss_equicorr(theta, params))

    def torch_ll(rows):
        X = torch.as_tensor(rows, device="cuda")                         # zero-copy view of the library's buffer, strided (n, d)
        assert X.shape == (len(rows), d) and X.dtype == torch.float64 and X.is_cuda
        z = X * tp[4:4 + d]
        s1, s2 = z.sum(dim=1), (z * z).sum(dim=1)
        return tp[1] - 0.5 * (tp[2] * s2 - tp[3] * s1 * s1)

    eng = HipEngine(algo=algo, n_chains=N, dim=d, target_id=R.TARGET_HOST, target_params=None, seed=79, **kw)
    okw = {k: v for k, v in kw.items() if k in ("del_pairs", "burnin_gen", "n_cr_gen", "p_snooker")}
    ora = R.OracleSampler(algo, N, d, R.TARGET_HOST, None, 79, ll_fn=py_ll, **okw)
    X0 = np.random.RandomState(5).normal(size=(N, d)) * np.sqrt(np.arange(d) + 1.0)
    eng.set_state(X0)
    st_rows = eng.state_device()
    np.testing.assert_array_equal(torch.as_tensor(st_rows, device="cuda").cpu().numpy(), X0)      # the local chains where they lie
    eng.set_loglike_device(torch_ll(st_rows))
    np.testing.assert_allclose(eng.get_loglike(), [py_ll(x) for x in X0], rtol=1e-12)
    ora.set_state(X0)
    eng.begin_run()
    for g in range(9):
        for ph in range(2):
            rows = eng.propose_device()
            ids = torch.as_tensor(rows.ids, device="cuda").cpu().numpy()
            assert len(rows) == len(ids) and np.all(ids >= 0) and len(set(ids.tolist())) == len(ids)
            eng.commit_device(torch_ll(rows))
        ora._generation(g, 0.5, True, 1e-12 if algo == R.ALGO_DREAM else 1e-15, 1e-2, None)
    st = eng.stats()
    assert st["local_n_accepted"] == ora.local_n_accepted and st["local_n_rejected"] == ora.local_n_rejected
    np.testing.assert_allclose(eng.get_state(), ora.X, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(eng.get_history(), ora.history_array(), rtol=1e-10, atol=1e-12)
    if algo == R.ALGO_DREAM:
        np.testing.assert_allclose(st["p_cr"], ora.cr.p_cr, rtol=1e-8)
    # host memory, a wrong dtype or a wrong length are refused
    from bipymc_amd._lib import BpmError
    rows = eng.propose_device()
    with pytest.raises(TypeError):
        eng.commit_device(np.zeros(len(rows)))
    with pytest.raises(TypeError):
        eng.commit_device(torch.zeros(len(rows), dtype=torch.float32, device="cuda"))
    with pytest.raises((BpmError, ValueError)):
        eng.commit_device(torch.zeros(len(rows), dtype=torch.float64))               # a CPU tensor: host memory
    eng.commit_device(torch_ll(rows))
    eng.close()
    print("device_parity ok")


def transports():
    """One likelihood, five ways to call it (samplers.py:36-43 calls ln_like_fn row by row): per row, vectorised over a half generation in one piece,
    vectorised with the read-back in five overlapped pieces, the pieces evaluated by a pool of three host threads, and as a torch function on the
    device (vectorized="device": no PCIe).  The host forms give bit-identical chains; the device form agrees to rounding with the same accept count."""
    from bipymc_amd import DreamMpi
    d, N = 12, 4096
    isig = 1.0 / np.sqrt(np.arange(d) + 1.0)
    isig_t = torch.tensor(isig, dtype=torch.float64, device="cuda")
    calls = dict(row=0, block=[], dev=[])

    def ll_row(theta):
        calls["row"] += 1
        return float(-0.5 * np.sum((theta * isig) ** 2))

    def ll_block(thetas):
        calls["block"].append(len(thetas))
        return -0.5 * np.sum((thetas * isig) ** 2, axis=1)

    def ll_dev(rows):
        X = torch.as_tensor(rows, device="cuda")
        calls["dev"].append(tuple(X.shape))
        return -0.5 * ((X * isig_t) ** 2).sum(dim=1)

    kw = dict(n_chains=N, seed=8, burnin_gen=6, n_cr_gen=2)
    G = 12
    a = DreamMpi(ll_row, np.zeros(d), **kw)
    b = DreamMpi(ll_block, np.zeros(d), vectorized=True, **kw)
    c = DreamMpi(ll_block, np.zeros(d), vectorized=True, callback_chunks=5, **kw)
    e = DreamMpi(ll_dev, np.zeros(d), vectorized="device", **kw)
    f = DreamMpi(lambda th: -0.5 * np.sum((th * isig) ** 2, axis=1), np.zeros(d), vectorized=True, callback_chunks=6, callback_threads=3, **kw)
    f.run_mcmc(N * (G + 1))
    for s_ in (a, b, c, e):
        assert not s_.uses_device_target
        s_.run_mcmc(N * (G + 1))
    assert calls["row"] == N + N * G and calls["dev"][0] == (N, d) and len(calls["dev"]) == 1 + 2 * G
    assert calls["block"].count(N) == 2 and max(c_ for c_ in calls["block"] if c_ != N) == N // 2      # one piece ... or five of ~N / 10 rows
    Ha, Hb, Hc, He = (s_.param_est(0)[2] for s_ in (a, b, c, e))
    assert np.array_equal(Ha, Hb) and np.array_equal(Ha, Hc)
    assert np.array_equal(Ha, f.param_est(0)[2])                     # ... and the pieces evaluated by a pool of three host threads
    np.testing.assert_allclose(He, Ha, rtol=1e-9, atol=1e-12)
    assert a.n_accepted == b.n_accepted == c.n_accepted == e.n_accepted
    with pytest.raises(ValueError):
        DreamMpi(ll_block, np.zeros(d), vectorized="gpu", **kw)
    print("transports ok")


if __name__ == "__main__":
    if sys.argv[1] == "device_parity":
        device_parity(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), json.loads(sys.argv[5]))
    elif sys.argv[1] == "transports":
        transports()
