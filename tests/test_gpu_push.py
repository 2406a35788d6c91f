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


def _test_lib_path():
    from bipymc_amd import _lib as L
    assert os.path.exists(L.TEST_LIB_PATH), "build_variants/libbipymc_test.so missing: make -C bipymc_amd/csrc"
    return L.TEST_LIB_PATH


def _single(case, world=1):
    from _push_worker import single_rank_reference
    return single_rank_reference(case, world)


@pytest.mark.parametrize("case", ["dream_gauss100", "dream_mix8_outlier", "demc_banana_snooker", "dream_gauss100_long"])
@pytest.mark.parametrize("R", [2, 4, 8])
def test_push_exchange_local_group_equals_single_rank(case, R):
    """R handles of this process on HIP streams; the cross-rank barrier as announce-all / join / wait-all."""
    from _push_worker import local_group_check
    nd, ns = local_group_check(case, R)
    assert nd == 0 and ns > 0


@pytest.mark.parametrize("case,R", [("dream_gauss700", 2), ("dream_gauss700", 4), ("dream_gauss1300", 2)])
def test_push_exchange_with_wide_rows(case, R):
    """d = 700 and d = 1300 (the looped wide-row kernel, one and several passes over a chunk of 256 coordinates) over the push exchange."""
    from _push_worker import local_group_check
    nd, ns = local_group_check(case, R)
    assert nd == 0 and ns > 0


@pytest.mark.parametrize("case,R", [("dream_gauss100", 2), ("dream_gauss100_long", 4), ("dream_mix8_outlier", 4), ("dream_gauss100", 8),
                                    ("demc_banana_snooker", 8), ("dream_gauss700", 2)])
def test_push_exchange_local_group_with_a_queue_per_rank(case, R):
    """The same with every rank on an AQL queue of its own (BPM_TEST_PATHS=groupqueues, read at library load: a child process): the
    ranks' one-wavefront barrier kernels announce and WAIT FOR EACH OTHER across queues inside one process -- what the ranks of a
    multi-GPU world do across GPUs."""
    env = dict(os.environ)
    env["BPM_TEST_PATHS"] = "groupqueues"
    env["BPM_LIB_PATH"] = _test_lib_path()            # (BPM_TEST_PATHS is read by the test variant of the library only)
    env["BPM_PUSH_TIMEOUT_S"] = "20"
    env["BPM_QUEUE_TIMEOUT_S"] = "60"
    out = subprocess.run([sys.executable, os.path.join(HERE, "_push_worker.py"), "--group", case, str(R)], env=env, capture_output=True,
                         text=True, timeout=400)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("GROUP ok")][0]
    nd, ns = (int(t.split("=")[1]) for t in line.split()[2:4])
    assert nd > 0 and ns == 0, line                                          # every update kernel through the rank's own queue


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


@pytest.mark.parametrize("case,R", [("dream_gauss100", 2), ("dream_gauss100", 4), ("dream_gauss100_long", 2), ("dream_mix8_outlier", 4), ("demc_banana_snooker", 2),
                                    ("cfg4_shape", 4), ("dream_gauss700", 2)])
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


def test_a_rank_that_waits_for_a_missing_peer_gives_up_and_says_so():
    """Every cross-rank wait inside push_sync_kernel is bounded (BPM_PUSH_TIMEOUT_S): a rank whose peer never arrives must come back with an
    error at the next bpm_synchronize, not hang (the reference's ranks WOULD hang in comm.Allgather, demc.py:93-94).  Two ranks of a local
    group with queues of their own, only rank 0 is ever stepped; bound 2 s.  In a child process (BPM_TEST_PATHS is read at library load)."""
    code = r'''
import os, sys, time, numpy as np
sys.path.insert(0, os.getcwd())
from bipymc_amd import _lib as L
from bipymc_amd.engine import HipEngine
from bipymc_amd.utils import d100_gauss
tid, tp, d = d100_gauss.Gauss_100D(dim=6)._bpm_target_spec()
uid = b"BPMLOCAL" + bytes(120)
ranks = [HipEngine(algo=L.ALGO_DREAM, n_chains=64, dim=d, target_id=tid, target_params=tp, seed=1, rank=r, world_size=2, nccl_uid=uid, lib=L.load_test(), burnin_gen=0)
         for r in range(2)]
blobs = [e.push_export() for e in ranks]
for e in ranks:
    e.push_connect(blobs)
    e.set_state(np.random.RandomState(0).normal(size=(64, d)))
    e.begin_run()
t0 = time.time()
ranks[0].step(3)                       # rank 1 never steps: rank 0's entry barrier waits for it
try:
    ranks[0].synchronize()
    print("NOERROR")
except L.BpmError as err:
    print("ERROR after %.1f s: %s" % (time.time() - t0, err))
# a sampler whose push exchange timed out can leave it for another exchange (bench.py: validate_exchange), never re-enter it
try:
    ranks[0].set_exchange(mode="push")
    print("PUSH-AGAIN accepted")
except L.BpmError as err:
    print("PUSH-AGAIN refused: %s" % err)
ranks[0].set_exchange(mode="dense")
ranks[0].synchronize()
print("LEFT-PUSH synchronize ok")
'''
    env = dict(os.environ)
    env["BPM_TEST_PATHS"] = "groupqueues"
    env["BPM_LIB_PATH"] = _test_lib_path()            # (BPM_TEST_PATHS is read by the test variant of the library only)
    env["BPM_PUSH_TIMEOUT_S"] = "2"
    env["BPM_QUEUE_TIMEOUT_S"] = "60"
    out = subprocess.run([sys.executable, "-c", code], cwd=os.path.join(HERE, ".."), env=env, capture_output=True, text=True, timeout=200)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-2500:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith(("ERROR", "NOERROR"))][0]
    assert line.startswith("ERROR after") and "waited longer than the limit for rank 1" in line, line
    assert float(line.split()[2]) < 60.0
    assert "PUSH-AGAIN refused" in out.stdout and "ran into its limit earlier" in out.stdout and "LEFT-PUSH synchronize ok" in out.stdout, out.stdout[-1500:]


def test_connect_refuses_wrong_blobs_and_can_be_repeated():
    """bpm_push_connect checks what it is given (rank order, world, sampler shape) and leaves nothing mapped when it fails: the caller's
    communicator delivered the blobs (the reference's counterpart is mpi_comm, demc.py:15), so a mix-up must be an error, not a wrong run."""
    import ctypes as C
    from bipymc_amd import _lib as L
    from bipymc_amd.engine import HipEngine
    from bipymc_amd.utils import d100_gauss
    tid, tp, d = d100_gauss.Gauss_100D(dim=6)._bpm_target_spec()
    uid = b"BPMLOCAL" + bytes(120)
    ranks = [HipEngine(algo=L.ALGO_DREAM, n_chains=32, dim=d, target_id=tid, target_params=tp, seed=1, rank=r, world_size=2, nccl_uid=uid, lib=L.load_test()) for r in range(2)]
    other = HipEngine(algo=L.ALGO_DREAM, n_chains=64, dim=d, target_id=tid, target_params=tp, seed=1, rank=1, world_size=2, nccl_uid=uid, lib=L.load_test())
    blobs = [e.push_export() for e in ranks]
    with pytest.raises(L.BpmError, match="not the export of rank"):
        ranks[0].push_connect(blobs[::-1])                               # rank order mixed up
    with pytest.raises(L.BpmError, match="another sampler shape"):
        ranks[0].push_connect([blobs[0], other.push_export()])           # a rank of another world
    with pytest.raises(L.BpmError, match="needs bpm_push_connect first"):
        ranks[0].set_exchange("push")
    single = HipEngine(algo=L.ALGO_DREAM, n_chains=32, dim=d, target_id=tid, target_params=tp, seed=1)
    with pytest.raises(L.BpmError, match="no push exchange"):
        single.push_export()
    for e in ranks:
        e.push_connect(blobs)                                            # and now for real
    with pytest.raises(L.BpmError, match="already connected"):
        ranks[0].push_connect(blobs)
    arr = (C.c_void_p * 2)(*[e._h for e in ranks])
    ok = C.c_int32(0)
    L.check(ranks[0].lib.bpm_push_selftest(arr, 2, C.byref(ok)))
    assert ok.value == 1 and ranks[0].exchange_stats()["push_flags_fine_grained"]
    for e in ranks + [other, single]:
        e.close()


def _run_mode(mode, R, tmp_path, *extra, env_extra=None, timeout=300):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["BPM_PUSH_TIMEOUT_S"] = "20"
    env.update(env_extra or {})
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_push_worker.py"), mode, str(tmp_path), str(r), str(R)] + [str(x) for x in extra],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(R)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o.decode("utf-8", "replace"))
    return [p.returncode for p in procs], outs


def _fields(line):
    return dict(kv.split("=") for kv in line.split()[1:] if "=" in kv)


@pytest.mark.parametrize("wrong", [0, 1])
def test_selftest_probes_every_peers_arena_from_the_librarys_own_queue(wrong, tmp_path):
    """bpm_push_selftest stores probe rows into every peer's ARENA (first / last row of its block, the block's last slots, its
    om block) from the library's own queue, under system- and agent-scope packet fences, and every rank verifies what arrived.  wrong = 0: ok on
    both ranks, the state that was in place is untouched, the world then runs.  wrong = 1: rank 1 hands out the arena handle of ANOTHER buffer --
    rank 0's stores land elsewhere, rank 1 answers ok = 0 (-> the callers fall back to an RCCL exchange), nothing faults, nothing hangs."""
    rcs, outs = _run_mode("--selftest", 2, tmp_path, wrong)
    assert rcs == [0, 0], "\n".join(o[-2500:] for o in outs)
    st = [_fields([ln for ln in o.splitlines() if ln.startswith("SELFTEST")][0]) for o in outs]
    assert all(f["own_queue"] == "1" and f["state_intact"] == "1" for f in st), st
    if not wrong:
        assert all(f["ok"] == "1" and f["arena_system"] == "1" and f["arena_agent"] == "1" for f in st), st
        sums = [[ln for ln in o.splitlines() if ln.startswith("RAN")][0].split("sum=")[1] for o in outs]
        assert sums[0] == sums[1]                                   # the replicas agree after a run behind the self-test
    else:
        assert st[1]["ok"] == "0" and st[1]["arena_system"] == "0", st      # rank 1 never received rank 0's probe rows
        assert st[0]["ok"] == "1", st                                       # (rank 1 mapped rank 0 correctly: the decision is collective, demc.py)
        assert not any("RAN" in o for o in outs)


def test_destroy_orders_the_teardown_across_ranks(tmp_path):
    """the library itself orders the teardown.  Rank 0 closes right behind its last step, with no barrier of the caller's,
    while rank 1 is busy for another second: bpm_destroy announces "closing" and waits (bounded) for rank 1's announcement before it unmaps --
    both ranks end cleanly and rank 1's results are the single-rank run's."""
    ref = _single("dream_gauss100", world=2)
    rcs, outs = _run_mode("--teardown", 2, tmp_path, "late", env_extra={"BPM_PUSH_CLOSE_TIMEOUT_S": "20"})
    assert rcs == [0, 0], "\n".join(o[-2500:] for o in outs)
    waited = float([ln for ln in outs[0].splitlines() if ln.startswith("CLOSED")][0].split()[3])
    assert 0.3 < waited < 15.0, outs[0][-500:]                              # it waited for rank 1 (about a second), not for the bound
    o = np.load(os.path.join(str(tmp_path), "late_rank1.npz"))
    assert np.array_equal(o["state"], ref["state"]) and np.array_equal(o["p_cr"], ref["p_cr"])


def test_a_rank_that_goes_on_after_a_peer_closed_is_told_so_at_once(tmp_path):
    """... and a rank that steps again after its peer has closed (a mis-sequenced caller; the reference's ranks would hang in comm.Allgather,
    demc.py:93-94) gets an error that names the closed rank within seconds -- not after BPM_PUSH_TIMEOUT_S, and never a store into freed memory."""
    rcs, outs = _run_mode("--teardown", 2, tmp_path, "more", env_extra={"BPM_PUSH_CLOSE_TIMEOUT_S": "4", "BPM_PUSH_TIMEOUT_S": "60"})
    assert rcs == [0, 0], "\n".join(o[-2500:] for o in outs)
    line = [ln for ln in outs[1].splitlines() if ln.startswith("MORE")][0]
    assert line.startswith("MORE error after") and "rank 0 closed its sampler" in line, line
    assert float(line.split()[3]) < 30.0, line
