"""The push exchange (world_size > 1): owners store accepted rows straight into the other ranks' replicas through mapped buffers,
a one-wavefront kernel per half generation orders the ranks -- where the reference's ranks meet in comm.Allgather twice per
generation (bipymc/demc.py:93-94,116-117).  Checked the way the reference's own property holds: a world of R ranks reproduces the
single-rank run BIT FOR BIT (histories, p_cr, counters).  Ranks as (a) R handles of this process (local group, up to R = 8) and
(b) R PROCESSES SHARING THIS ONE GPU, connected through hipIpc handles (R = 2, 4: the box allows 6 GPU processes) -- RCCL is not
involved anywhere."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)


def _single(case, world=1):
    from _push_worker import case_spec, start_state
    from bipymc_amd.engine import HipEngine
    spec, algo, N, kw, G = case_spec(case)
    if N is None:
        N = 8192 * world
    tid, tp, d = spec
    one = HipEngine(algo=algo, n_chains=N, dim=d, target_id=tid, target_params=tp, seed=11, **kw)
    one.set_state(start_state(case, N, d))
    one.begin_run(flip=0.4)
    one.step(G // 2)
    one.step(G - G // 2)
    st = one.stats()
    res = dict(state=one.get_state(), ll=one.get_loglike(), hist_a=one.get_history(1, 2)[0], hist_b=one.get_history(G, G + 1)[0],
               hist=one.get_history() if N <= 1024 else None, p_cr=st["p_cr"], n_cr_updates=st["n_cr_updates"],
               acc=np.array([st["local_n_accepted"], st["local_n_rejected"], st["n_outlier_resets"]]), N=N, d=d, G=G)
    one.close()
    return res


@pytest.mark.parametrize("case", ["dream_gauss100", "dream_mix8_outlier", "demc_banana_snooker"])
@pytest.mark.parametrize("R", [2, 4, 8])
def test_push_exchange_local_group_equals_single_rank(case, R):
    from _push_worker import case_spec, start_state
    from bipymc_amd import _lib as L
    from bipymc_amd.engine import HipEngine
    spec, algo, N, kw, G = case_spec(case)
    tid, tp, d = spec
    ref = _single(case)
    uid = b"BPMLOCAL" + bytes(120)
    ranks = [HipEngine(algo=algo, n_chains=N, dim=d, target_id=tid, target_params=tp, seed=11, rank=r, world_size=R, nccl_uid=uid, **kw)
             for r in range(R)]
    blobs = [e.push_export() for e in ranks]
    for e in ranks:
        e.push_connect(blobs)
    arr = (C.c_void_p * R)(*[e._h for e in ranks])
    ok = C.c_int32(0)
    L.check(ranks[0].lib.bpm_push_selftest(arr, R, C.byref(ok)))
    assert ok.value == 1
    x0 = start_state(case, N, d)
    for e in ranks:
        assert e.exchange_stats()["mode"] == "push"
        e.set_state(x0)
        e.begin_run(flip=0.4)
    L.check(ranks[0].lib.bpm_local_group_step(arr, R, G // 2))
    L.check(ranks[0].lib.bpm_local_group_step(arr, R, G - G // 2))
    n_local = N // R
    H = np.concatenate([e.get_history() for e in ranks], axis=1)
    assert np.array_equal(H, ref["hist"])                                  # every chain's whole history, bit for bit
    for r, e in enumerate(ranks):
        st = e.stats()
        assert np.array_equal(e.get_state(), ref["state"])                 # every replica
        assert np.array_equal(e.get_loglike(), ref["ll"][r * n_local:(r + 1) * n_local])
        assert np.array_equal(st["p_cr"], ref["p_cr"]) and np.array_equal(st["n_cr_updates"], ref["n_cr_updates"])
        assert st["n_outlier_resets"] == ref["acc"][2]
        assert e.exchange_stats()["push_gens"] == G
    assert sum(e.stats()["local_n_accepted"] for e in ranks) == ref["acc"][0]
    for e in ranks:
        e.close()


def _run_processes(case, R, tmp_path):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["BPM_PUSH_TIMEOUT_S"] = "60"
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_push_worker.py"), str(tmp_path), str(r), str(R), case], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(R)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o.decode("utf-8", "replace"))
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d:\n%s" % (r, outs[r][-3000:])
    return [np.load(os.path.join(str(tmp_path), "out_rank%d.npz" % r)) for r in range(R)]


@pytest.mark.parametrize("case,R", [("dream_gauss100", 2), ("dream_gauss100", 4), ("dream_mix8_outlier", 4), ("demc_banana_snooker", 2),
                                    ("cfg4_shape", 4)])
def test_push_exchange_between_processes_sharing_the_gpu(case, R, tmp_path):
    """R processes, one GPU, buffers mapped with hipIpcOpenMemHandle, every rank's kernels on its own AQL queue."""
    ref = _single(case, world=R)
    res = _run_processes(case, R, tmp_path)
    n_local = ref["N"] // R
    for r, o in enumerate(res):
        assert o["xmode"][0] == 1 and o["xmode"][1] == ref["G"]               # push exchange, every generation
        assert o["xmode"][2] >= 2 * ref["G"] and o["xmode"][3] == 0           # update kernels through the rank's own queue
        assert np.array_equal(o["state"], ref["state"])                        # every replica equals the single-rank state
        assert np.array_equal(o["ll"], ref["ll"][r * n_local:(r + 1) * n_local])
        assert np.array_equal(o["hist_a"], ref["hist_a"][r * n_local:(r + 1) * n_local])
        assert np.array_equal(o["hist_b"], ref["hist_b"][r * n_local:(r + 1) * n_local])
        if ref["hist"] is not None:
            assert np.array_equal(o["hist"], ref["hist"][:, r * n_local:(r + 1) * n_local])
        assert np.array_equal(o["p_cr"], ref["p_cr"]) and np.array_equal(o["n_cr_updates"], ref["n_cr_updates"])
        assert o["acc"][2] == ref["acc"][2]
    assert sum(int(o["acc"][0]) for o in res) == ref["acc"][0]


def test_sampler_classes_over_the_push_exchange(tmp_path):
    """DreamMpi with a communicator of 2 processes on this GPU, exchange="push": the class connects the ranks by itself
    (bipymc_amd/demc.py:_connect_exchange) and param_est equals the single-process sampler's."""
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from _file_comm import FileComm
from bipymc_amd import DreamMpi
from bipymc_amd.utils import d100_gauss
d_, rank, world = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
comm = FileComm(d_, rank, world) if world > 1 else None
t = d100_gauss.Gauss_100D(rho=0.5, dim=10)
s = DreamMpi(t.ln_like, np.zeros(10), n_chains=64, mpi_comm=comm, n_cr_gen=3, burnin_gen=10, seed=77, exchange="push" if world > 1 else "auto")
s.run_mcmc(64 * 30)
mean, std, chain = s.param_est(64 * 5)
mm, ss = s.param_est_moments(64 * 5)
if rank == 0:
    np.savez(os.path.join(d_, "cls_w%d.npz" % world), mean=mean, std=std, chain=chain, mm=mm, ss=ss, p_cr=s.p_cr, acc=s.n_accepted,
             used=str(getattr(s, "exchange_used", None)))
'''
    root = os.path.join(HERE, "..")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    d1 = tmp_path / "w1"
    d1.mkdir()
    subprocess.check_call([sys.executable, "-c", code, str(d1), "0", "1"], cwd=root, env=env, timeout=300)
    d2 = tmp_path / "w2"
    d2.mkdir()
    procs = [subprocess.Popen([sys.executable, "-c", code, str(d2), str(r), "2"], cwd=root, env=env) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    a, b = np.load(str(d1 / "cls_w1.npz")), np.load(str(d2 / "cls_w2.npz"))
    assert str(b["used"]) == "push"
    assert np.array_equal(a["chain"], b["chain"]) and np.array_equal(a["p_cr"], b["p_cr"]) and int(a["acc"]) == int(b["acc"])
    np.testing.assert_allclose(a["mm"], b["mm"], rtol=1e-12, atol=1e-14)
