"""world_size-2 gloo worker for tests/test_dist_gloo.py (CPU; the oracle stands in for the device engine)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def run(rank, world, port, out_dir, case):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        from _oracle_engine import factory
        import bipymc_amd.demc as _demc
        _demc._engine_factory = factory            # (this worker process only: the oracle engine stands in for the HIP engine)
        from bipymc_amd.demc import DeMcMpi
        from bipymc_amd.dream import DreamMpi
        from bipymc_amd.utils import banana_rv, d100_gauss
        if case == "dream":
            t = d100_gauss.Gauss_100D(rho=0.5, dim=6)
            s = DreamMpi(t.ln_like, np.zeros(6), n_chains=12, mpi_comm="torch", n_cr_gen=3, burnin_gen=8, seed=1234)
            s.run_mcmc(12 * 16)
        else:
            t = banana_rv.Banana_2D()
            s = DeMcMpi(t.ln_like, np.zeros(2), n_chains=8, mpi_comm="torch", seed=99, p_snooker=0.2)
            s.run_mcmc(8 * 21, flip=0.3)
        assert s.comm.size == world and s.comm.rank == rank
        assert list(s.rank_chain_ids) == list(range(rank * s.n_local, (rank + 1) * s.n_local))     # demc.py:39
        mean, std, chain = s.param_est(n_burn=24)
        res = dict(rank=rank, n_accepted=s.n_accepted, n_rejected=s.n_rejected, local_acc=s.local_n_accepted)
        if rank == 0:
            res.update(mean=mean, std=std, chain=chain, full=s.param_est(0)[2])
        else:
            assert mean is None and std is None and chain is None                                  # demc.py:247-248
            s.param_est(0)
        if case == "dream":
            res["p_cr"] = s.p_cr
        mm, ss = s.param_est_moments(24)          # device-style reduction, combined across ranks: same on every rank
        res["mom_mean"], res["mom_std"] = mm, ss
        c = s.get_chain(s.n_chains - 1, 0)          # a chain owned by the last rank, fetched to rank 0
        if rank == 0:
            res["last_chain"] = c.chain
        else:
            assert c is None
        # owner == collection rank: nobody communicates (demc.py:301-304); the other ranks get None and the next
        # collective still pairs up on every rank
        c0 = s.get_chain(0, 0)
        assert (c0 is not None) == (rank == 0)
        if rank == 0:
            assert c0.global_id == 0
        own_last = s.get_chain(s.n_chains - 1, world - 1)
        assert (own_last is not None) == (rank == world - 1)
        # every chain to every collection rank, in the same order on all ranks
        for coll in range(world):
            for c_id in range(s.n_chains):
                ch = s.get_chain(c_id, coll)
                assert (ch is not None) == (rank == coll), (c_id, coll, rank)
                if ch is not None:
                    assert ch.chain.shape == (s.am_chains[0].chain.shape[0], s.dim)
        got = [ch for ch in s.iter_all_chains(0)]
        assert len(got) == s.n_chains and all((g is not None) == (rank == 0) for g in got)
        np.savez(os.path.join(out_dir, "%s_rank%d.npz" % (case, rank)), **res)
    finally:
        dist.barrier()
        dist.destroy_process_group()
