"""Host-side classes (bipymc_amd.demc.DeMcMpi / dream.DreamMpi) on CPU, with the oracle standing in
for the device engine (tests only): constructor/kwargs surface, generation count, result
assembly and row order, counters, checkpoint round trip -- mirroring how the reference's own
tests drive the samplers (tests/test_dblgauss.py:43-62,130-140)."""
import os

import numpy as np
import pytest

from bipymc_amd.demc import DeMcMpi
from bipymc_amd.dream import DreamMpi
from bipymc_amd.utils import banana_rv, d100_gauss, dblgauss_rv, mixture_nd
from oracle import sampler_ref as R

pytestmark = pytest.mark.usefixtures("oracle_engine")      # the oracle engine behind DeMcMpi / DreamMpi (conftest.py)


def test_generation_count_matches_reference_loop():
    """while j < int((n - n_chains) / size): j += local updates   (demc.py:79,107)"""
    t = dblgauss_rv.BimodeGauss_2D()
    s = DreamMpi(t.ln_like, np.zeros(2), n_chains=10, seed=1)
    for n, gens in ((10, 0), (11, 1), (20, 1), (21, 2), (100000, 9999), (5, 0)):
        assert s._n_generations(n) == gens


def test_dream_public_surface_and_row_order():
    np.random.seed(42)
    t = dblgauss_rv.BimodeGauss_2D()
    s = DreamMpi(t.ln_like, np.zeros(2), n_chains=10, mpi_comm=None, n_cr_gen=5, burnin_gen=30)
    assert s.uses_device_target and s.dim == 2 and s.n_chains == 10
    assert s.del_pairs == 3 and s.n_cr == 3 and s.burnin_gen == 30 and s.p_cr_update_gen == 5
    np.testing.assert_array_equal(s.CR, [1 / 3, 2 / 3, 1.0])
    s.run_mcmc(10 * 51)                                   # 50 generations
    mean, std, chain = s.param_est(n_burn=100)
    _, _, full = s.param_est(n_burn=0)
    assert full.shape == (510, 2) and chain.shape == (410, 2)
    assert np.array_equal(chain, full[100:])
    # interleaved super chain: row g*N + i = chain i at generation g (demc.py:260-270)
    for i in (0, 3, 9):
        c = s.am_chains[i]
        assert c.global_id == i and c.chain.shape == (51, 2) and c.chain_len == 51 and c.dim == 2
        assert np.array_equal(full[i::10], c.chain)
        assert np.array_equal(c.current_pos, c.chain[-1])
    np.testing.assert_allclose(mean, full[100:].mean(0))
    np.testing.assert_allclose(std, full[100:].std(0))
    # counters: 50 generations x 10 chains, rejected starts at 1 (demc.py:67-68)
    assert s.n_accepted + s.n_rejected == 501
    assert 0 < s.acceptance_fraction < 1
    assert s.p_cr.shape == (3,) and abs(s.p_cr.sum() - 1) < 1e-12
    assert s.n_cr_updates.sum() > 0
    # a second run continues the history and resets the counters (demc.py:67-68,78)
    s.run_mcmc(10 * 3)
    assert s.param_est(0)[2].shape == (530, 2)
    assert s.n_accepted + s.n_rejected == 21
    assert len(list(s.iter_local_chains())) == 10 and len(s.gather_all_chains()) == 10
    assert s.get_chain(4).global_id == 4 and s.get_chain_rank(7) == 0


def test_seed_from_numpy_global_state_is_reproducible():
    t = banana_rv.Banana_2D()
    out = []
    for _ in range(2):
        np.random.seed(7)
        s = DeMcMpi(t.ln_like, np.zeros(2), n_chains=8)
        s.run_mcmc(8 * 20)
        out.append(s.param_est(0)[2])
    assert np.array_equal(out[0], out[1])
    np.random.seed(8)
    s = DeMcMpi(t.ln_like, np.zeros(2), n_chains=8)
    s.run_mcmc(8 * 20)
    assert not np.array_equal(out[0], s.param_est(0)[2])


def test_constructor_contract():
    t = banana_rv.Banana_2D()
    with pytest.raises(AssertionError):
        DeMcMpi(t.ln_like, np.zeros(2), n_chains=3)          # samplers.py:249
    s = DeMcMpi(t.ln_like, None, n_chains=8, dim=2, seed=3, inflate=10.0)  # unknown kwargs ignored
    assert s.dim == 2 and np.allclose(s._get_local_chain_state(), 0, atol=1e-2)
    assert s.h5_file == "sampler_checkpoint.h5" and s.checkpoint == 0 and s.warm_start is False
    # varepsilon per dimension (examples/ex_exp_fit.py:135-141)
    s = DeMcMpi(t.ln_like, np.array([1.0, 2.0]), varepsilon=np.array([1e-2, 1e-8]), n_chains=64, seed=3)
    X = s._get_local_chain_state()
    assert 0.05 < X[:, 0].std() < 0.2 and X[:, 1].std() < 1e-3
    assert abs(X[:, 0].mean() - 1.0) < 0.05
    # frozen ln_like with kwargs (samplers.py:36-43) -> not a device target
    f = lambda theta, scale=1.0: -0.5 * np.sum(theta ** 2) / scale
    s2 = DeMcMpi.__new__(DeMcMpi)
    from bipymc_amd.utils import _target
    assert _target.resolve(f, {"scale": 2.0}, 2)[0] == 0
    assert _target.resolve(t.ln_like, {}, 2)[0] == 3 and _target.resolve(t.ln_like, {}, 3)[0] == 0
    assert _target.resolve(d100_gauss.Gauss_100D().ln_like, {}, 100)[0] == 1
    assert _target.resolve(mixture_nd.BimodeGauss_ND(8).ln_like, {}, 8)[0] == 2


def test_targets_match_oracle_param_blocks_and_values():
    g = d100_gauss.Gauss_100D()
    assert np.array_equal(g._bpm_target_spec()[1], R.gauss_equicorr_params(0.5, np.sqrt(np.arange(100) + 1.0)))
    x = np.random.RandomState(0).normal(size=(5, 100))
    np.testing.assert_allclose(g.ln_like(x), R.ll_gauss_equicorr(x, g._bpm_target_spec()[1]), rtol=1e-14)
    np.testing.assert_allclose(np.cov(g.rvs(200000).T)[:3, :3], g.cov[:3, :3], rtol=0.05)
    b = dblgauss_rv.BimodeGauss_2D()
    assert np.array_equal(b._bpm_target_spec()[1], R.mixture_pairs_params(0.25, 0.75, [0, 0], [2, 2], [.25, .25], [.25, .25], 0.8, -0.8))
    y1, y2 = b.rvs(100000)
    assert abs(y1.mean() - 1.5) < 0.02 and abs(y1.var() - 0.8125) < 0.02
    assert b.pdf(0.0, 0.0) == pytest.approx(np.exp(0.05924291847653619), rel=1e-12)
    n = banana_rv.Banana_2D()
    assert np.array_equal(n._bpm_target_spec()[1], R.banana_params())
    y1, y2 = n.rvs(200000)
    assert abs(y2.mean() - 1.16125) < 0.02 and abs(y1.var() - 1.3225) < 0.03
    assert n.check_prob_lvl(0.0, 1.16125, 0.1)
    m = mixture_nd.BimodeGauss_ND(8)
    xs = m.rvs(50000)
    assert xs.shape == (50000, 8) and abs(xs.mean() - 1.5) < 0.02
    # d = 2 marginal of the 8-D mixture is the reference's 2-D target
    np.testing.assert_allclose(mixture_nd.BimodeGauss_ND(2).ln_like(np.array([0.3, -0.2])), b.ln_like(np.array([0.3, -0.2])))


def test_checkpoint_round_trip(tmp_path):
    t = dblgauss_rv.BimodeGauss_2D()
    f = str(tmp_path / "ck.npz")
    s = DreamMpi(t.ln_like, np.zeros(2), n_chains=8, n_cr_gen=3, burnin_gen=10, seed=5,
                 h5_file=f, checkpoint=4)
    s.run_mcmc(8 * 13)                                  # 12 generations, checkpoints at 4, 8, 12
    full = s.param_est(0)[2]
    s2 = DreamMpi(t.ln_like, None, n_chains=8, dim=2, n_cr_gen=3, burnin_gen=10, seed=5,
                  h5_file=f, warm_start=True)
    assert np.array_equal(s2.param_est(0)[2], full)      # chain histories restored (demc.py:217-233)
    np.testing.assert_allclose(s2.p_cr, s.p_cr)           # and what the reference forgets
    s2.run_mcmc(8 * 3)
    assert s2.param_est(0)[2].shape == (8 * 15, 2)
    assert np.array_equal(s2.param_est(0)[2][:8 * 13], full)


def test_checkpoint_format_is_decided_by_the_file(tmp_path):
    """`read()` must pick the format from what is on disk.  An HDF5 file (the reference's write_chain_h5 layout,
    chain.py:59-70) with neither h5py nor libhdf5 available is an explicit error, not a silent look for the .npz twin; a missing
    file names both candidates; a path ending in .npz is always the NumPy twin."""
    from bipymc_amd import checkpoint
    hist = np.random.RandomState(0).normal(size=(5, 4, 3))
    f = str(tmp_path / "state.npz")
    assert checkpoint.write(f, hist, {"t_abs": 4}) == f
    back, adapt = checkpoint.read(f, 4, 3)
    assert np.array_equal(back, hist) and int(adapt["t_abs"]) == 4
    with np.load(f) as z:
        assert sorted(k for k in z.files if k.startswith("chains/")) == ["chains/chain_id_%d" % i for i in range(4)]
        assert z["chains/chain_id_2"].shape == (5, 3)              # (T, dim) per chain, as chain.py:64-66 writes
    with pytest.raises(IOError, match="neither"):
        checkpoint.read(str(tmp_path / "absent.h5"), 4, 3)
    if checkpoint.hdf5_backend() is None:
        assert checkpoint.write(str(tmp_path / "s.h5"), hist) == str(tmp_path / "s.h5") + ".npz"
        with open(str(tmp_path / "real.h5"), "wb") as fh:          # something that IS an HDF5 file by its signature
            fh.write(b"\x89HDF\r\n\x1a\n" + b"\0" * 64)
        with pytest.raises(IOError, match="h5py"):
            checkpoint.read(str(tmp_path / "real.h5"), 4, 3)


def _h5dump():
    import shutil
    for c in ("h5dump", "/opt/conda/bin/h5dump"):
        p = shutil.which(c) or (c if os.path.exists(c) else None)
        if p:
            return p
    return None


def test_checkpoint_hdf5_layout_of_the_reference(tmp_path):
    """The reference's on-disk format itself: HDF5, one gzip dataset /chains/chain_id_<i> of shape (T, dim) float64 per chain
    (chain.py:59-93; reader mc_plot/vis_mcmc_chains.py:16-41).  Written and read with h5py when it is importable, else with the
    HDF5 C library h5py itself wraps (bipymc_amd/_hdf5.py, ctypes); the file is inspected with the HDF5 project's own `h5dump`
    when that tool is present: an independent reader sees the reference's layout."""
    import subprocess
    from bipymc_amd import checkpoint
    backend = checkpoint.hdf5_backend()
    if backend is None:
        pytest.skip("neither h5py nor libhdf5 can be loaded on this machine: the HDF5 form of the checkpoint layout cannot run here")
    hist = np.random.RandomState(1).normal(size=(6, 5, 2))
    f = str(tmp_path / "state.h5")
    assert checkpoint.write(f, hist, {"t_abs": 5, "seed": 2 ** 61 + 7, "p_cr": np.array([0.2, 0.3, 0.5]),
                                      "delta_m": np.array([1.0, 2.0, 3.0]), "n_cr_updates": np.array([4.0, 5.0, 6.0])}) == f
    assert checkpoint._is_hdf5(f)
    back, adapt = checkpoint.read(f, 5, 2)
    assert np.array_equal(back, hist)
    assert int(adapt["t_abs"]) == 5 and int(adapt["seed"]) == 2 ** 61 + 7
    assert np.array_equal(adapt["p_cr"], [0.2, 0.3, 0.5]) and np.array_equal(adapt["n_cr_updates"], [4.0, 5.0, 6.0])
    tool = _h5dump()
    if tool:
        hdr = subprocess.check_output([tool, "-H", "-p", f]).decode()
        for i in range(5):
            assert 'DATASET "chain_id_%d"' % i in hdr
        assert 'GROUP "chains"' in hdr
        assert "H5T_IEEE_F64LE" in hdr and "( 6, 2 ) / ( 6, 2 )" in hdr
        assert "COMPRESSION DEFLATE { LEVEL 4 }" in hdr and "CHUNKED" in hdr      # create_dataset(..., compression="gzip")
        data = subprocess.check_output([tool, "-d", "/chains/chain_id_3", "-y", "-w", "0", "-m", "%.17g", f]).decode()
        vals = [float(t) for t in data.split("DATA {")[1].split("}")[0].replace(",", " ").split()]
        assert np.array_equal(np.array(vals).reshape(6, 2), hist[:, 3, :])          # an independent reader, bit for bit
    # a file laid out the way the reference's own writer does (chain.py:64-66: create_dataset per chain, no side group)
    g = str(tmp_path / "ref_style.h5")
    if backend == "h5py":
        import h5py
        with h5py.File(g, "w") as h:
            for i in range(5):
                h.create_dataset("/chains/chain_id_" + str(i), data=hist[:, i, :], compression="gzip")
    else:
        from bipymc_amd import _hdf5
        with _hdf5.File(g, "w") as h:
            h.create_group("/chains")
            for i in range(5):
                h.write("/chains/chain_id_" + str(i), hist[:, i, :], gzip=True)
    back2, adapt2 = checkpoint.read(g, 5, 2)
    assert np.array_equal(back2, hist) and adapt2 == {}


def test_checkpoint_round_trip_hdf5_through_the_sampler(tmp_path):
    """save_state / warm start through an .h5 file (the reference's default h5_file name ends in .h5, demc.py:24)."""
    from bipymc_amd import checkpoint
    if checkpoint.hdf5_backend() is None:
        pytest.skip("neither h5py nor libhdf5 can be loaded on this machine")
    t = dblgauss_rv.BimodeGauss_2D()
    f = str(tmp_path / "sampler_checkpoint.h5")
    s = DreamMpi(t.ln_like, np.zeros(2), n_chains=8, n_cr_gen=3, burnin_gen=10, seed=5,
                 h5_file=f, checkpoint=4)
    s.run_mcmc(8 * 13)
    assert checkpoint._is_hdf5(f) and not os.path.exists(f + ".npz")
    full = s.param_est(0)[2]
    s2 = DreamMpi(t.ln_like, None, n_chains=8, dim=2, n_cr_gen=3, burnin_gen=10, seed=5,
                  h5_file=f, warm_start=True)
    assert np.array_equal(s2.param_est(0)[2], full)
    np.testing.assert_allclose(s2.p_cr, s.p_cr)
    s2.run_mcmc(8 * 3)
    assert np.array_equal(s2.param_est(0)[2][:8 * 13], full)


def test_statistical_cfg1_shape_on_oracle():
    """BASELINE config 1 plumbing (shortened): DREAM N=10 on the bimodal target through the public API."""
    t = dblgauss_rv.BimodeGauss_2D()
    s = DreamMpi(t.ln_like, np.zeros(2), n_chains=10, n_cr_gen=50, burnin_gen=500, seed=11)
    s.run_mcmc(30000)
    mean, std, _ = s.param_est(n_burn=10000)
    assert abs(mean[0] - 1.5) < 0.25 and abs(mean[1] - 1.5) < 0.25      # both modes visited
    assert 0.6 < std[0] < 1.2


def test_attribute_and_method_surface_of_the_reference_classes():
    """Every attribute and method SURVEY section 8(b) lists as readable on the reference's DreamMpi / DeMcMpi
    (demc.py:14-338, dream.py:17-140, samplers.py:30-80, chain.py:13-124) exists here, with the reference's meaning."""
    t = dblgauss_rv.BimodeGauss_2D()
    common = ["am_chains", "n_chains", "dim", "comm", "rank_chain_ids", "n_accepted", "n_rejected", "local_n_accepted",
              "local_n_rejected", "acceptance_fraction", "h5_file", "checkpoint", "warm_start", "frozen_ln_like_fn",
              "run_mcmc", "param_est", "super_chain_mpi", "gather_all_chains", "iter_local_chains", "iter_all_chains",
              "get_chain", "get_chain_rank", "save_state", "load_state", "init_chains", "init_warmstart_chain"]
    dream_only = ["CR", "p_cr", "n_cr_updates", "delta_m", "gamma_scale", "del_pairs", "burnin_gen", "p_cr_update_gen", "n_cr"]
    d = DreamMpi(t.ln_like, np.zeros(2), n_chains=8, mpi_comm=None, seed=2,
                 ln_kwargs={}, inflate=3.0)                       # unknown kwargs are accepted silently (samplers.py:36-43)
    m = DeMcMpi(t.ln_like, np.zeros(2), n_chains=8, mpi_comm=None, seed=2)
    for name in common + dream_only:
        assert hasattr(d, name), name
    for name in common:
        assert hasattr(m, name), name
    assert d.gamma_scale == 1.0 and d.burnin_gen == 300 and d.p_cr_update_gen == 50          # dream.py:20-27 defaults
    assert list(d.rank_chain_ids) == list(range(8)) and d.comm.size == 1 and d.comm.rank == 0
    assert d.local_n_accepted == 0 and d.local_n_rejected == 1                               # demc.py:67-68 before any update
    assert abs(d.frozen_ln_like_fn(np.array([0.0, 0.0])) - t.ln_like(np.array([0.0, 0.0]))) < 1e-14
    c = d.am_chains[3]
    for name in ("chain", "current_pos", "global_id", "chain_len", "dim", "append_sample"):   # chain.py:13-124
        assert hasattr(c, name), name
    with pytest.raises(AssertionError):
        DreamMpi(t.ln_like, np.zeros(2), n_chains=3, mpi_comm=None)   # samplers.py:249


def test_default_device_enters_its_collective_before_it_raises(monkeypatch):
    """A mis-launched rank must not leave the others waiting in an allgather: every rank gathers, then all
    raise the same error.  (The reference's ranks pick no device, demc.py:15.)"""
    import bipymc_amd.demc as D

    class Comm(object):
        size = 2

        def __init__(self, rank, others):
            self.rank, self.others, self.calls = rank, others, 0

        def allgather(self, obj):
            self.calls += 1
            out = list(self.others)
            out.insert(self.rank, obj)
            return out

    class Stub(object):
        pass
    monkeypatch.setattr(D, "_visible_device_count", lambda: 2)
    monkeypatch.setattr(D, "_hostname", lambda: "box")
    for var in ("OMPI_COMM_WORLD_LOCAL_RANK", "MV2_COMM_WORLD_LOCAL_RANK", "MPI_LOCALRANKID", "SLURM_LOCALID"):
        monkeypatch.delenv(var, raising=False)
    # this rank is fine, the other one reports an error: both raise
    monkeypatch.setenv("LOCAL_RANK", "0")
    s = Stub()
    s.comm = Comm(0, [("box", -1, 2, "rank 1: local rank 5 but only 2 visible GPU(s): launch one process per GPU")])
    with pytest.raises(RuntimeError, match="local rank 5"):
        D.DeMcMpi._default_device(s)
    assert s.comm.calls == 1
    # this rank is the broken one: it still enters the collective first
    monkeypatch.setenv("LOCAL_RANK", "5")
    s.comm = Comm(1, [("box", 0, 2, None)])
    with pytest.raises(RuntimeError, match="local rank 5"):
        D.DeMcMpi._default_device(s)
    assert s.comm.calls == 1
    # a rank that sees one device only (launcher narrowed the view) takes part as well, and duplicates are named
    monkeypatch.setattr(D, "_visible_device_count", lambda: 1)
    monkeypatch.setenv("LOCAL_RANK", "1")
    s.comm = Comm(1, [("box", 0, 1, None)])
    assert D.DeMcMpi._default_device(s) == 0 and s.comm.calls == 1
    monkeypatch.setattr(D, "_visible_device_count", lambda: 2)
    monkeypatch.setenv("LOCAL_RANK", "0")
    s.comm = Comm(1, [("box", 0, 2, None)])
    with pytest.raises(RuntimeError, match="same GPU"):
        D.DeMcMpi._default_device(s)


def test_exchange_is_chosen_collectively(monkeypatch):
    """DeMcMpi._connect_exchange (world > 1): the push exchange is used only when EVERY rank exported, mapped its peers and passed the
    self-test; one failing rank sends all ranks to the RCCL replay exchange (exchange="auto") or raises on all of them (exchange="push").
    The reference has no counterpart: its ranks meet in comm.Allgather (demc.py:93-94,116-117)."""
    import warnings
    import bipymc_amd.demc as D

    class Eng(object):
        def __init__(self, fail_at=None):
            self.fail_at, self.calls, self.mode = fail_at, [], None

        def push_export(self):
            self.calls.append("export")
            if self.fail_at == "export":
                raise RuntimeError("no arena")
            return b"x" * 256

        def push_connect(self, blobs):
            self.calls.append("connect")
            assert len(blobs) == 2
            if self.fail_at == "connect":
                raise RuntimeError("hipIpcOpenMemHandle failed")

        def push_selftest(self):
            self.calls.append("selftest")
            if self.fail_at == "selftest-raises":
                raise RuntimeError("hipErrorLaunchFailure")
            return self.fail_at != "selftest"

        def set_exchange(self, mode, cap=0):
            self.mode = mode

    class Comm(object):
        """this process is rank 0 of 2; `other` scripts what rank 1 contributes to each collective, in order"""
        rank, size = 0, 2

        def __init__(self, other):
            self.other = list(other)

        def allgather(self, obj):
            return [obj, self.other.pop(0)]

        def Barrier(self):
            pass

    def run(eng, other, exchange="auto", dim=100):
        s = object.__new__(D.DeMcMpi)
        s._engine, s.comm, s.exchange, s._target_id, s.dim = eng, Comm(other), exchange, 1, dim      # (a device target)
        s._connect_exchange()
        return s

    good = [(b"y" * 256, None), (True, None), (True, None)]
    s = run(Eng(), good)
    assert s.exchange_used == "push" and s._engine.mode == "push" and s._engine.calls == ["export", "connect", "selftest"]
    # the OTHER rank could not map this one: nobody runs the self-test, everybody takes the fallback
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        s = run(Eng(), [(b"y" * 256, None), (False, "connect: hipIpcOpenMemHandle failed")])
    assert s.exchange_used == "replay" and s._engine.mode == "replay" and "selftest" not in s._engine.calls
    assert any("rank 1: connect" in str(x.message) for x in w)
    # this rank's self-test fails, the other's passes: fallback too
    with warnings.catch_warnings(record=True):
        warnings.simplefilter("always")
        s = run(Eng(fail_at="selftest"), good)
    assert s.exchange_used == "replay"
    # this rank's self-test RAISES: it still enters the collective (the other rank is not left waiting) and everybody falls back
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        s = run(Eng(fail_at="selftest-raises"), good)
    assert s.exchange_used == "replay" and s.comm.other == [] and any("self-test: hipErrorLaunchFailure" in str(x.message) for x in w)
    # rows wider than 512 coordinates (the looped kernel has no replay form: bpm_set_exchange refuses it, sampler.hip) fall back to the DENSE
    # all-gather instead of raising in the constructor; at 512 the replay exchange still serves
    for dim, want in ((640, "dense"), (513, "dense"), (512, "replay")):
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            s = run(Eng(fail_at="selftest"), good, dim=dim)
        assert s.exchange_used == want and s._engine.mode == want, (dim, s.exchange_used)
        assert any(("whole blocks" if want == "dense" else "accept bytes") in str(x.message) for x in w)
    # exchange="push" (no RCCL to fall back to): an error on every rank
    with pytest.raises(RuntimeError, match="push exchange could not be connected"):
        run(Eng(fail_at="export"), [(None, "export: no arena"), (False, None)], exchange="push")
    # an RCCL exchange asked for by name: no mapping at all
    s = run(Eng(), [], exchange="dense")
    assert s.exchange_used == "dense" and s._engine.calls == []


def test_hip_likelihood_needs_an_engine_that_compiles_it_or_a_python_fn(oracle_engine):
    """ln_like_fn = HipLikelihood(source, params): the HIP engine compiles the source into the generation loop (tests/test_gpu_api.py); an engine without
    bpm_set_device_likelihood (the CPU test engine) can only go through python_fn -- without one the constructor says so before any chain exists."""
    from bipymc_amd import DreamMpi, HipLikelihood
    src = "__device__ double ln_like(const double* x, int d, const double* p) { double s = 0; for (int j = 0; j < d; ++j) s += x[j] * x[j]; return -0.5 * s; }"
    with pytest.raises(TypeError, match="python_fn"):
        DreamMpi(HipLikelihood(src), np.zeros(3), n_chains=8, seed=2)
    ll = HipLikelihood(src, params=[1.0, 2.0], python_fn=lambda th: -0.5 * float(np.sum(np.asarray(th) ** 2)))
    assert ll(np.ones(3)) == -1.5 and ll.params.dtype == np.float64 and ll.params.shape == (2,)
