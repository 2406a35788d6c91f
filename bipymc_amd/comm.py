"""Communicator adapters.  The reference takes an mpi4py communicator (`mpi_comm=MPI.COMM_WORLD`,
demc.py:15) and uses it for the state Allgather, barriers and the chain gather to root.  On
MI355X the state exchange is an RCCL all-gather issued inside libbipymc_hip.so; the host
communicator is only needed for control-plane objects (the RCCL unique id, accept counters,
histories for param_est).  Accepted `mpi_comm` values:

  None                      single process
  "torch" / a ProcessGroup  torch.distributed (one process per GPU, launched by torchrun)
  an mpi4py-style object    anything with .rank/.size (or Get_rank/Get_size), .bcast, .allgather
"""


class SingleComm(object):
    rank = 0
    size = 1

    def Get_rank(self):
        return 0

    def Get_size(self):
        return 1

    def bcast(self, obj, root=0):
        return obj

    def allgather(self, obj):
        return [obj]

    def Barrier(self):
        pass


class TorchDistComm(object):
    """torch.distributed as the control-plane communicator (gloo or nccl(=RCCL) backend)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised (launch with torch.distributed.run)")
        self._dist = dist
        self._group = group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)

    def Get_rank(self):
        return self.rank

    def Get_size(self):
        return self.size

    def bcast(self, obj, root=0):
        box = [obj]
        self._dist.broadcast_object_list(box, src=root, group=self._group)
        return box[0]

    def allgather(self, obj):
        out = [None] * self.size
        self._dist.all_gather_object(out, obj, group=self._group)
        return out

    def Barrier(self):
        self._dist.barrier(group=self._group)


class _MpiLike(object):
    def __init__(self, comm):
        self._c = comm
        self.rank = comm.rank if hasattr(comm, "rank") else comm.Get_rank()
        self.size = comm.size if hasattr(comm, "size") else comm.Get_size()

    def Get_rank(self):
        return self.rank

    def Get_size(self):
        return self.size

    def bcast(self, obj, root=0):
        return self._c.bcast(obj, root=root) if self.size > 1 else obj

    def allgather(self, obj):
        return self._c.allgather(obj) if self.size > 1 else [obj]

    def Barrier(self):
        if self.size > 1:
            self._c.Barrier()


def wrap(mpi_comm):
    if mpi_comm is None:
        return SingleComm()
    if isinstance(mpi_comm, (SingleComm, TorchDistComm, _MpiLike)):
        return mpi_comm
    if isinstance(mpi_comm, str):
        if mpi_comm == "torch":
            return TorchDistComm()
        raise ValueError("unknown communicator %r" % (mpi_comm,))
    if type(mpi_comm).__module__.startswith("torch"):
        return TorchDistComm(mpi_comm)
    return _MpiLike(mpi_comm)
