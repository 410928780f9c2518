"""The CONTROL plane of a multi-process job, behind one small interface -- rendezvous of the device communicator's id,
barriers, max-over-ranks timing, and the few host-array collectives the layout mirror offers (`gather_local_array`,
`fill_jtj` on host arrays).  Device data never passes through it: row blocks and normal-equation sums travel between
device buffers through `_lib.Comm` (gst_comm_*: RCCL over xGMI, or HIP IPC).

What the reference's callers have is an mpi4py communicator inside a `ResourceAllocation`
(pygsti/baseobjs/resourceallocation.py:43-120; Gatherv / Allgatherv :316-348, allreduce_sum :441-508): `MpiControl` wraps
one.  Tests and bench.py run where mpi4py does not exist and use `control_gloo.GlooControl` (a torch.distributed gloo group --
the only module of the package that imports torch).  A control plane is three primitives --

    bcast_bytes(payload, src)   barrier()   max_float(x)

-- plus host-array collectives every implementation derives from two more, `allgather_array` and `allreduce_sum`."""
import numpy as np


class ControlPlane:
    rank, size = 0, 1

    # -- the three primitives ------------------------------------------------------------------------------------------
    def bcast_bytes(self, payload, src=0):
        raise NotImplementedError

    def barrier(self):
        raise NotImplementedError

    def max_float(self, x):
        raise NotImplementedError

    # -- host arrays ---------------------------------------------------------------------------------------------------
    def allgather_array(self, send):
        """[rank 0's array, rank 1's, ...] on every rank; `send` has the same shape and dtype everywhere."""
        raise NotImplementedError

    def gather_array(self, send, dst):
        """The same list on rank `dst` only (None elsewhere).  Default: an all-gather."""
        got = self.allgather_array(send)
        return got if self.rank == dst else None

    def allreduce_sum(self, arr):
        """In-place sum over the ranks (float64 numpy array)."""
        raise NotImplementedError

    def all_floats(self, x):
        return [float(v[0]) for v in self.allgather_array(np.array([float(x)]))]


class SingleControl(ControlPlane):
    """One process: every collective is the identity."""

    def __init__(self, rank=0, size=1):
        self.rank, self.size = 0, 1

    def bcast_bytes(self, payload, src=0):
        return payload

    def barrier(self):
        pass

    def max_float(self, x):
        return float(x)

    def allgather_array(self, send):
        return [np.array(send)]

    def allreduce_sum(self, arr):
        return arr


class MpiControl(ControlPlane):
    """Over an mpi4py communicator -- `ResourceAllocation.comm` of the reference's callers (resourceallocation.py:43-120)."""

    def __init__(self, comm):
        self.comm = comm
        self.rank, self.size = int(comm.Get_rank()), int(comm.Get_size())

    def bcast_bytes(self, payload, src=0):
        return self.comm.bcast(payload, root=src)

    def barrier(self):
        self.comm.Barrier()

    def max_float(self, x):
        out = np.empty(1)
        self.comm.Allreduce(np.array([float(x)]), out, op=_mpi_op(self.comm, "MAX"))
        return float(out[0])

    def allgather_array(self, send):
        send = np.ascontiguousarray(send)
        recv = np.empty((self.size,) + send.shape, send.dtype)
        self.comm.Allgather(send, recv)
        return [recv[r] for r in range(self.size)]

    def gather_array(self, send, dst):
        send = np.ascontiguousarray(send)
        recv = np.empty((self.size,) + send.shape, send.dtype) if self.rank == dst else None
        self.comm.Gather(send, recv, root=dst)
        return None if recv is None else [recv[r] for r in range(self.size)]

    def allreduce_sum(self, arr):
        buf = np.ascontiguousarray(arr)
        out = np.empty_like(buf)
        self.comm.Allreduce(buf, out)          # (op defaults to SUM)
        arr[...] = out
        return arr


def _mpi_op(comm, name):
    """mpi4py's MPI.MAX without importing mpi4py at module level (a duck-typed communicator in tests has none)."""
    try:
        from mpi4py import MPI
        return getattr(MPI, name)
    except ImportError:
        return name


_CURRENT = None


def current():
    """The control plane this process joined (pygsti_amd.dist.init), or None."""
    return _CURRENT


def set_current(ctl):
    global _CURRENT
    _CURRENT = ctl
