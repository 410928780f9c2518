"""The control plane over a torch.distributed *gloo* group -- what tests and bench.py use (mpi4py does not exist on the
boxes this build runs on).  The ONLY module of the package that imports torch; PyTorch is plumbing here and never sees
device data.  See pygsti_amd/control.py for the interface."""
import os

import numpy as np

from .control import ControlPlane


class GlooControl(ControlPlane):
    def __init__(self, group=None):
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo")
        self.group = group
        self.rank, self.size = dist.get_rank(group), dist.get_world_size(group)

    def bcast_bytes(self, payload, src=0):
        import torch.distributed as dist
        box = [payload if self.rank == src else None]
        dist.broadcast_object_list(box, src=src, group=self.group)
        return box[0]

    def barrier(self):
        import torch.distributed as dist
        dist.barrier(group=self.group)

    def max_float(self, x):
        import torch
        import torch.distributed as dist
        t = torch.tensor([float(x)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def allgather_array(self, send):
        import torch
        import torch.distributed as dist
        s = torch.from_numpy(np.ascontiguousarray(send))
        recv = [torch.empty_like(s) for _ in range(self.size)]
        dist.all_gather(recv, s, group=self.group)
        return [r.numpy() for r in recv]

    def gather_array(self, send, dst):
        import torch
        import torch.distributed as dist
        s = torch.from_numpy(np.ascontiguousarray(send))
        recv = [torch.empty_like(s) for _ in range(self.size)] if self.rank == dst else None
        dist.gather(s, recv, dst=dst, group=self.group)
        return None if recv is None else [r.numpy() for r in recv]

    def allreduce_sum(self, arr):
        import torch
        import torch.distributed as dist
        buf = arr if arr.flags.c_contiguous else np.ascontiguousarray(arr)
        dist.all_reduce(torch.from_numpy(buf), group=self.group)
        if buf is not arr:
            arr[...] = buf
        return arr

    def shutdown(self):
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()
