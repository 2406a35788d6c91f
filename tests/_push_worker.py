"""One rank of a world whose ranks are PROCESSES SHARING ONE GPU (tests/test_gpu_push.py): the push exchange through IPC-mapped
buffers, no RCCL.  usage: _push_worker.py <dir> <rank> <world> <case>"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def case_spec(case):
    from bipymc_amd import _lib as L
    from bipymc_amd.utils import banana_rv, d100_gauss, mixture_nd
    if case == "dream_gauss100":
        return d100_gauss.Gauss_100D()._bpm_target_spec(), L.ALGO_DREAM, 64, dict(burnin_gen=8, n_cr_gen=3), 20
    if case == "dream_gauss100_long":      # crosses two 64-generation table windows: the windows built INSIDE the generation loop
        return d100_gauss.Gauss_100D()._bpm_target_spec(), L.ALGO_DREAM, 64, dict(burnin_gen=8, n_cr_gen=3), 150
    if case == "dream_mix8_outlier":
        return (mixture_nd.BimodeGauss_ND(8)._bpm_target_spec(), L.ALGO_DREAM, 480,
                dict(burnin_gen=30, n_cr_gen=3, del_pairs=2, outlier_every=10), 45)
    if case == "demc_banana_snooker":
        return banana_rv.Banana_2D()._bpm_target_spec(), L.ALGO_DEMC, 40, dict(p_snooker=0.3), 25
    if case == "dream_gauss700":           # the looped wide-row kernel (d > 512) with pushes
        return d100_gauss.Gauss_100D(dim=700)._bpm_target_spec(), L.ALGO_DREAM, 48, dict(burnin_gen=6, n_cr_gen=2), 14
    if case == "dream_gauss1300":          # the looped wide-row kernel (kernels_wide.h) pushing its accepted rows chunk by chunk
        return d100_gauss.Gauss_100D(dim=1300)._bpm_target_spec(), L.ALGO_DREAM, 32, dict(burnin_gen=6, n_cr_gen=2), 12
    if case == "cfg4_shape":          # BASELINE configs[3] per-rank shape: 8192 chains of the 100-D Gaussian per rank
        return d100_gauss.Gauss_100D()._bpm_target_spec(), L.ALGO_DREAM, None, dict(burnin_gen=4, n_cr_gen=2), 12
    raise ValueError(case)


def start_state(case, N, d):
    return np.random.RandomState(3).normal(size=(N, d)) + 0.5


def main():
    d_, rank, world, case = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    from _file_comm import FileComm
    from bipymc_amd.engine import HipEngine
    comm = FileComm(d_, rank, world)
    spec, algo, N, kw, G = case_spec(case)
    if N is None:
        N = 8192 * world
    tid, tp, d = spec
    if case.startswith("class_"):
        raise ValueError
    e = HipEngine(algo=algo, n_chains=N, dim=d, target_id=tid, target_params=tp, seed=11, rank=rank, world_size=world,
                  nccl_uid=HipEngine.push_uid(), **kw)
    e.push_connect(comm.allgather(e.push_export()))
    comm.Barrier()
    assert e.push_selftest(), "push self-test failed on rank %d" % rank
    e.set_state(start_state(case, N, d))
    e.begin_run(flip=0.4)
    comm.Barrier()
    e.step(G // 2)
    e.step(G - G // 2)                  # (a second call: another entry barrier)
    e.synchronize()
    st = e.stats()
    xs = e.exchange_stats()
    ls = e.launch_stats()
    rows = (1, G)
    np.savez(os.path.join(d_, "out_rank%d.npz" % rank), state=e.get_state(), ll=e.get_loglike(),
             hist_a=e.get_history(rows[0], rows[0] + 1)[0], hist_b=e.get_history(rows[1], rows[1] + 1)[0],
             hist=e.get_history() if N <= 1024 else np.zeros(0), p_cr=st["p_cr"], n_cr_updates=st["n_cr_updates"],
             acc=np.array([st["local_n_accepted"], st["local_n_rejected"], st["n_outlier_resets"]]),
             xmode=np.array([xs["mode"] == "push", xs["push_gens"], ls["direct"], ls["stream"]], dtype=np.int64))
    comm.Barrier()                      # nobody unmaps while a peer may still be reading its results
    e.close()


def single_rank_reference(case, world=1):
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


def local_group_check(case, R):
    """R ranks as handles of THIS process (bpm_local_group_step) over the push exchange == the single-rank run, bit for bit.
    -> (update launches through the ranks' own queues, through HIP streams)"""
    import ctypes as C
    from bipymc_amd import _lib as L
    from bipymc_amd.engine import HipEngine
    spec, algo, N, kw, G = case_spec(case)
    tid, tp, d = spec
    ref = single_rank_reference(case)
    uid = b"BPMLOCAL" + bytes(120)
    ranks = [HipEngine(algo=algo, n_chains=N, dim=d, target_id=tid, target_params=tp, seed=11, rank=r, world_size=R, nccl_uid=uid, lib=L.load_test(), **kw)
             for r in range(R)]
    ls0 = ranks[0].launch_stats()
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
    L.check(ranks[0].lib.bpm_local_group_step(arr, R, G // 2), ranks[0].lib)
    L.check(ranks[0].lib.bpm_local_group_step(arr, R, G - G // 2), ranks[0].lib)
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
    ls = ranks[0].launch_stats()
    for e in ranks:
        e.close()
    return ls["direct"] - ls0["direct"], ls["stream"] - ls0["stream"]


def selftest_main():
    """_push_worker.py --selftest <dir> <rank> <world> <wrong>: connect, run bpm_push_selftest, print what it said.  wrong = 1: rank 1
    publishes the export of its sampler with the ARENA HANDLE OF ANOTHER sampler of the same shape (a decoy that stays alive): rank 0 then
    maps -- and probes -- the wrong buffer, and rank 1 must see that nothing arrived in its arena.  Blob layout: sampler.hip, struct PushBlob
    (the arena's hipIpcMemHandle_t at bytes 80..144)."""
    d_, rank, world, wrong = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    from _file_comm import FileComm
    from bipymc_amd import _lib as L
    from bipymc_amd.engine import HipEngine
    from bipymc_amd.utils import mixture_nd
    comm = FileComm(d_, rank, world)
    tid, tp, d = mixture_nd.BimodeGauss_ND(8)._bpm_target_spec()
    kw = dict(algo=L.ALGO_DREAM, n_chains=96 * world, dim=d, target_id=tid, target_params=tp, seed=11, rank=rank, world_size=world,
              nccl_uid=HipEngine.push_uid(), burnin_gen=20, n_cr_gen=3, outlier_every=10)
    e = HipEngine(**kw)
    blob = e.push_export()
    decoy = None
    if wrong and rank == 1:
        decoy = HipEngine(**kw)
        other = decoy.push_export()
        blob = blob[:80] + other[80:144] + blob[144:]
    e.push_connect(comm.allgather(blob))
    x0 = start_state("selftest", 96 * world, d)
    e.set_state(x0)                                     # (a state is in place: the self-test must leave it as it found it)
    comm.Barrier()
    ok = e.push_selftest()
    xs = e.exchange_stats()
    same = bool(np.array_equal(e.get_state(), x0))
    print("SELFTEST rank=%d ok=%d arena_system=%d arena_agent=%d own_queue=%d state_intact=%d" %
          (rank, ok, xs["arena_probe_system"], xs["arena_probe_agent"], xs["arena_probe_on_own_queue"], same), flush=True)
    oks = comm.allgather(bool(ok))
    if all(oks):                                        # a good connection goes on to run: the probe left nothing behind
        e.begin_run()
        comm.Barrier()
        e.step(12)
        e.synchronize()
        print("RAN rank=%d sum=%r" % (rank, float(e.get_state().sum())), flush=True)
    comm.Barrier()
    e.close()
    if decoy is not None:
        decoy.close()


def teardown_main():
    """_push_worker.py --teardown <dir> <rank> <world> <variant>.  variant "late": rank 0 closes right behind its last step while rank 1
    is still busy with its own results for a while -- bpm_destroy's hand-over makes rank 0 wait for it (bounded); nobody faults, results are
    the single-rank run's.  variant "more": rank 1 steps AGAIN after rank 0 has closed -- an error naming the closed rank, at once."""
    import time
    d_, rank, world, variant = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    from _file_comm import FileComm
    from bipymc_amd import _lib as L
    from bipymc_amd.engine import HipEngine
    comm = FileComm(d_, rank, world)
    spec, algo, N, kw, G = case_spec("dream_gauss100")
    tid, tp, d = spec
    e = HipEngine(algo=algo, n_chains=N, dim=d, target_id=tid, target_params=tp, seed=11, rank=rank, world_size=world,
                  nccl_uid=HipEngine.push_uid(), **kw)
    e.push_connect(comm.allgather(e.push_export()))
    comm.Barrier()
    assert e.push_selftest()
    e.set_state(start_state("dream_gauss100", N, d))
    e.begin_run(flip=0.4)
    comm.Barrier()
    e.step(G // 2)
    e.step(G - G // 2)
    if rank == 0:
        e.synchronize()
        t0 = time.time()
        e.close()                                       # NO barrier of the caller's in front of it
        print("CLOSED rank=0 after %.2f s" % (time.time() - t0), flush=True)
        return
    if variant == "late":
        time.sleep(1.0)                                 # (rank 0 is inside bpm_destroy by now, waiting for this rank's announcement)
        e.synchronize()
        st = e.stats()
        np.savez(os.path.join(d_, "late_rank%d.npz" % rank), state=e.get_state(), p_cr=st["p_cr"], acc=np.array([st["local_n_accepted"]]))
        t0 = time.time()
        e.close()
        print("CLOSED rank=%d after %.2f s" % (rank, time.time() - t0), flush=True)
    else:
        e.synchronize()
        time.sleep(1.5)                                 # rank 0 has announced "closing" (and is waiting, or gone)
        t0 = time.time()
        try:
            e.step(2)
            e.synchronize()
            print("MORE no error", flush=True)
        except L.BpmError as err:
            print("MORE error after %.2f s: %s" % (time.time() - t0, err), flush=True)
        e.close()


if __name__ == "__main__":
    if sys.argv[1] == "--selftest":
        selftest_main()
    elif sys.argv[1] == "--teardown":
        teardown_main()
    elif sys.argv[1] == "--group":
        nd, ns = local_group_check(sys.argv[2], int(sys.argv[3]))
        print("GROUP ok direct=%d stream=%d" % (nd, ns))
    else:
        main()
