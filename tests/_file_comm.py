"""A communicator over a shared directory (tests only): rank/size, bcast, allgather, Barrier -- the mpi4py-style surface
bipymc_amd.comm.wrap() accepts where the reference takes MPI.COMM_WORLD (demc.py:15).  Lets N processes that share ONE GPU
run the world_size > 1 path of the sampler classes without MPI, torch.distributed or RCCL (which refuses two ranks on one device)."""
import os
import pickle
import time


class FileComm(object):
    def __init__(self, directory, rank, size, timeout=120.0):
        self.dir, self.rank, self.size, self.timeout = directory, int(rank), int(size), float(timeout)
        self._n = 0

    def Get_rank(self):
        return self.rank

    def Get_size(self):
        return self.size

    def _path(self, n, r):
        return os.path.join(self.dir, "c%06d_r%d.pkl" % (n, r))

    def allgather(self, obj):
        n = self._n
        self._n += 1
        tmp = self._path(n, self.rank) + ".tmp"
        with open(tmp, "wb") as f:
            pickle.dump(obj, f)
        os.rename(tmp, self._path(n, self.rank))
        out = []
        t0 = time.time()
        for r in range(self.size):
            p = self._path(n, r)
            while not os.path.exists(p):
                if time.time() - t0 > self.timeout:
                    raise RuntimeError("FileComm: rank %d waited %.0f s for rank %d (collective %d)" % (self.rank, self.timeout, r, n))
                time.sleep(0.002)
            with open(p, "rb") as f:
                out.append(pickle.load(f))
        return out

    def bcast(self, obj, root=0):
        return self.allgather(obj if self.rank == root else None)[root]

    def Barrier(self):
        self.allgather(None)
