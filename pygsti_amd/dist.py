"""One-process-per-GPU sharding helpers (torch.distributed: backend "nccl" == RCCL over xGMI on ROCm,
"gloo" for the CPU tests).  PyTorch is plumbing here -- process groups and the gather collective; the
compute never goes through it.

How the path shards: layout atoms are disjoint circuit groups with contiguous element
slices and no data dependence between them (distlayout.py:326-332, 404-415), so rank r of N owns the
atoms r, r+N, ...; every rank fills its own rows of the 'e' / 'ep' array and NO collective is needed
inside a fill -- exactly as the reference's bulk_fill_* leaves rows distributed.  Only when the caller
asks for the assembled array (the reference's `layout.gather_local_array`, distlayout.py:1143-1147,
built on Gatherv / Allgatherv, resourceallocation.py:316-348) do row blocks travel: `gather_elements`
below is that collective -- an all-gather of max-padded row blocks (every rank gets the result, as
`allgather_local_array` does) or a gather to rank 0.
"""
import os

import numpy as np


def env_rank_size():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


class RankAlloc:
    """What create_layout needs to know about the process grid (stand-in for ResourceAllocation)."""

    def __init__(self, rank, size):
        self.comm_rank, self.comm_size = rank, size
        self.comm = None
        self.is_host_leader = True


def owned_row_blocks(layout, rank, size):
    """[(start, stop)] element ranges owned by `rank` (atoms rank, rank+size, ...)."""
    return [(at.element_slice.start, at.element_slice.stop) for a, at in enumerate(layout.all_atoms) if a % size == rank]


def gather_elements(local, layout, group=None, dst=None):
    """Assemble a full element-dimension array from per-rank row blocks.

    local : torch tensor of shape (num_elements, ...) in which THIS rank's rows are filled (device
            tensor under nccl, CPU tensor under gloo).  Rows owned by other ranks are ignored.
    dst   : None -> all ranks receive the assembled tensor (all_gather of padded blocks);
            int  -> only that rank does (gather).
    """
    import torch
    import torch.distributed as dist
    rank, size = dist.get_rank(group), dist.get_world_size(group)
    blocks = [owned_row_blocks(layout, r, size) for r in range(size)]
    rows = [sum(b - a for a, b in bl) for bl in blocks]
    pad = max(rows)
    tail = tuple(local.shape[1:])
    send = torch.zeros((pad,) + tail, dtype=local.dtype, device=local.device)
    off = 0
    for a, b in blocks[rank]:
        send[off:off + (b - a)] = local[a:b]
        off += b - a
    if dst is None:
        recv = [torch.empty_like(send) for _ in range(size)]
        dist.all_gather(recv, send, group=group)
    else:
        recv = [torch.empty_like(send) for _ in range(size)] if rank == dst else None
        dist.gather(send, recv, dst=dst, group=group)
        if rank != dst:
            return None
    out = torch.empty_like(local)
    for r in range(size):
        off = 0
        for a, b in blocks[r]:
            out[a:b] = recv[r][off:off + (b - a)]
            off += b - a
    return out
