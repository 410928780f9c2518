"""One process per GPU: process-group plumbing and the row-block exchange.

How the path shards: layout atoms are disjoint circuit groups with contiguous element slices and no
data dependence between them (distlayout.py:326-332, 404-415), so rank r of N owns the atoms r, r+N, ...; every rank
fills its own rows of the 'e' / 'ep' array and NO collective is needed inside a fill -- exactly as the reference's
bulk_fill_* leaves rows distributed.  Data moves only when the caller asks for the assembled array (the reference's
`layout.gather_local_array` / `allgather_local_array`, distlayout.py:1010-1156 and copalayout.py:479-518, built on
Gatherv / Allgatherv, resourceallocation.py:316-348) or for summed normal equations (`fill_jtj` / `fill_jtf`,
distlayout.py:1220-1359, on `allreduce_sum`, resourceallocation.py:441-508).

Two planes:
  * control (rendezvous, barriers, max-over-ranks timing, small host arrays): a `control.ControlPlane` -- an mpi4py
    communicator where the reference's callers have one (`control.MpiControl`, resourceallocation.py:43-120), a gloo group
    for tests and bench.py (`control_gloo.GlooControl`, the one module that imports torch).  This module imports neither;
  * data (row blocks of probabilities / Jacobians, J^T J sums): `_lib.Comm`, the C ABI's gst_comm_* -- RCCL over xGMI
    between device pointers, or the intra-node IPC transport (ranks sharing a GPU; fallback when RCCL cannot start).
"""
import os

import numpy as np

from . import _lib, control as _control


def env_rank_size():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


class RankAlloc:
    """What create_layout needs to know about the process grid (stand-in for ResourceAllocation)."""

    def __init__(self, rank, size):
        self.comm_rank, self.comm_size = rank, size
        self.comm = None
        self.is_host_leader = True


def owned_row_blocks(layout, rank, size):
    """[(start, stop)] element ranges `rank` contributes to element-dimension gathers: the atoms of its atom-processor
    when it is that processor's first rank (atoms k, k+na, ...; every rank is one when atoms are the only axis)."""
    return [(r0, r1) for r0, r1, _, _ in layout.owned_blocks("e", rank)]


def row_blocks(layout, size):
    """The block list of the C ABI's row exchanges: (owner rank, first row, rows) per atom, in atom order."""
    return [(layout.atom_owner_rank(a), at.element_slice.start, at.element_slice.stop - at.element_slice.start)
            for a, at in enumerate(layout.all_atoms)]


class _StdoutToStderr:
    """While active, file descriptor 1 points at stderr, and C stdio is flushed before it is restored: RCCL and gloo
    print banners to the process's stdout (fully buffered when it is a pipe, i.e. they would surface at exit), which
    must not end up in front of a caller's own output -- bench.py prints exactly one JSON line there."""

    def __enter__(self):
        import sys
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        import ctypes
        import sys
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


class DistContext:
    """Control plane + data communicator (gst_comm) of this process.  `comm` is None when size == 1 or when no device
    transport could be created (`comm_error` says why): callers then stay on the host paths."""

    def __init__(self, rank, size, local_rank, group, comm, comm_error=None, control=None):
        self.rank, self.size, self.local_rank = rank, size, local_rank
        self.group, self.comm, self.comm_error = group, comm, comm_error
        self.control = control if control is not None else _control.SingleControl()

    @property
    def transport(self):
        return None if self.comm is None else self.comm.info()["transport"]

    def barrier(self):
        if self.size > 1:
            self.control.barrier()

    def max_over_ranks(self, x):
        return float(x) if self.size == 1 else self.control.max_float(x)

    def all_floats(self, x):
        """[x of rank 0, x of rank 1, ...] on every rank."""
        return [float(x)] if self.size == 1 else self.control.all_floats(x)

    def all_ok(self, ok):
        """True iff `ok` on every rank."""
        return self.max_over_ranks(0.0 if ok else 1.0) == 0.0

    def broadcast_bytes(self, payload, src=0):
        return payload if self.size == 1 else self.control.bcast_bytes(payload, src)

    def close(self):
        if self.comm is not None:
            self.comm.close()
            self.comm = None

    def shutdown(self):
        """close() + leave the control group (gloo: destroy the process group)."""
        self.close()
        if hasattr(self.control, "shutdown"):
            self.control.shutdown()
        _control.set_current(None)


def init(device=None, transport="auto", want_comm=True, control=None, mpi_comm=None):
    """Join the job: the control plane first, then the device communicator.
    control  : a `control.ControlPlane`; or
    mpi_comm : an mpi4py communicator (the reference's `ResourceAllocation.comm`) -> `control.MpiControl`; or neither: the
               job `torch.distributed.run` (or any launcher exporting RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT) started,
               joined as a gloo group (`control_gloo.GlooControl`).
    transport: 'rccl', 'ipc' or 'auto' (RCCL, and if any rank fails to create it -- e.g. ranks sharing a GPU -- all ranks fall
               back to IPC together).  GST_TRANSPORT in the environment overrides it."""
    rank, size, local_rank = env_rank_size()
    if control is None and mpi_comm is not None:
        control = _control.MpiControl(mpi_comm)
    if control is not None:
        rank, size = control.rank, control.size
    if size == 1:
        return DistContext(0, 1, local_rank, None, None)
    _lib.lib()          # libgstfwd (and with it ROCm's HIP runtime) is loaded BEFORE torch brings its bundled copies
    if control is None:
        from .control_gloo import GlooControl
        with _StdoutToStderr():
            control = GlooControl()
    _control.set_current(control)
    ctx = DistContext(rank, size, local_rank, getattr(control, "group", None), None, control=control)
    if not want_comm:
        return ctx
    with _StdoutToStderr():
        _create_comm(ctx, rank, size, local_rank, device, transport)
    return ctx


def _comm_with_deadline(rank, size, uid, device, code, name, err):
    """Create the communicator under a deadline (GST_COMM_INIT_TIMEOUT seconds, default 90): ncclCommInitRank is a
    collective that blocks for as long as a peer is missing, so a rank that died, or a fabric that never comes up, must
    not hang the job.  The call runs on a helper thread (ctypes releases the GIL); on expiry the rank reports failure --
    loudly, on stderr -- and `_create_comm` moves every rank to the next transport together.  (The helper thread cannot
    be cancelled; it is a daemon and dies with the process.)"""
    import sys
    import threading
    timeout = float(os.environ.get("GST_COMM_INIT_TIMEOUT", "90"))
    box = {}

    def work():
        try:
            box["comm"] = _lib.Comm(rank, size, uid, device, code)
        except Exception as e:
            box["err"] = "%s: %s" % (type(e).__name__, e)

    th = threading.Thread(target=work, name="gst-comm-init", daemon=True)
    th.start()
    th.join(timeout)
    if th.is_alive():
        msg = "%s communicator not up after %.0f s on rank %d (GST_COMM_INIT_TIMEOUT)" % (name, timeout, rank)
        print("[pygsti_amd.dist] " + msg + " -- falling back", file=sys.stderr, flush=True)
        return None, msg
    return box.get("comm"), box.get("err", err)


def _create_comm(ctx, rank, size, local_rank, device, transport):
    transport = os.environ.get("GST_TRANSPORT", transport)
    if device is None:
        device = local_rank % max(_lib.device_count(), 1)
    errors = []
    for tr in (["rccl", "ipc"] if transport == "auto" else [transport]):
        code = {"rccl": _lib.TRANSPORT_RCCL, "ipc": _lib.TRANSPORT_IPC}[tr]
        uid, err = None, None
        if rank == 0:
            try:
                uid = _lib.Comm.unique_id(code)
            except Exception as e:          # e.g. no librccl on this box
                err = "%s: %s" % (type(e).__name__, e)
        uid = ctx.broadcast_bytes(uid)
        comm = None
        if uid is not None:
            comm, err = _comm_with_deadline(rank, size, uid, device, code, tr, err)
        if ctx.all_ok(comm is not None):
            ctx.comm = comm
            break
        if comm is not None:
            comm.close()
        errors.append("%s transport: %s" % (tr, err or "failed on another rank"))
    ctx.comm_error = "; ".join(errors) if errors else None
    if errors and rank == 0:
        import sys
        print("[pygsti_amd.dist] device transport fallback: %s -> %s" % (
            ctx.comm_error, "NONE (host-staged exchange)" if ctx.comm is None else
            {_lib.TRANSPORT_RCCL: "rccl", _lib.TRANSPORT_IPC: "ipc"}.get(ctx.comm.transport, "?")), file=sys.stderr, flush=True)


# ---- device arrays -------------------------------------------------------------------------------------------------
def allgather_elements_dev(ctx, layout, d_full, row_doubles=1, plan=None):
    """`allgather_local_array` on a device array [num_elements][row_doubles] in which this rank's atoms are filled:
    afterwards (stream order under RCCL) every rank holds every row."""
    if ctx.size > 1:
        ctx.comm.allgather_rows(d_full, row_doubles, row_blocks(layout, ctx.size), plan)


def gather_elements_dev(ctx, layout, d_local, d_full, row_doubles=1, root=0, plan=None):
    """`gather_local_array` (Gatherv to `root`) of device row blocks: non-root ranks pass their own atoms' rows packed
    in atom order (d_local), the root its full-size array with its own rows in place."""
    if ctx.size > 1:
        ctx.comm.gather_rows(d_local, d_full, row_doubles, row_blocks(layout, ctx.size), root, plan)


# ---- host arrays (the reference's semantics on numpy arrays; small ones: probabilities, objective terms) --------------
def _ctl(control=None):
    ctl = control if control is not None else _control.current()
    if ctl is None:
        raise RuntimeError("no control plane: call pygsti_amd.dist.init() (gloo), or pass an mpi4py communicator / a ControlPlane")
    return ctl


def _exchange_padded(ctl, send, dst):
    return ctl.allgather_array(send) if dst is None else ctl.gather_array(send, dst)


def gather_elements(local, layout, group=None, dst=None, control=None):
    """Assemble a full element-dimension HOST array from per-rank row blocks through the control plane.

    local : numpy array of shape (num_elements, ...) in which THIS rank's rows are filled.
    dst   : None -> all ranks receive the assembled array (all-gather of padded blocks); int -> only that rank does."""
    ctl = _ctl(control)
    loc = np.ascontiguousarray(local)
    rank, size = ctl.rank, ctl.size
    blocks = [owned_row_blocks(layout, r, size) for r in range(size)]
    rows = [sum(b - a for a, b in bl) for bl in blocks]
    pad = max(rows)
    send = np.zeros((pad,) + tuple(loc.shape[1:]), loc.dtype)
    off = 0
    for a, b in blocks[rank]:
        send[off:off + (b - a)] = loc[a:b]
        off += b - a
    recv = _exchange_padded(ctl, send, dst)
    if recv is None:
        return None
    out = np.empty_like(loc)
    for r in range(size):
        off = 0
        for a, b in blocks[r]:
            out[a:b] = recv[r][off:off + (b - a)]
            off += b - a
    return out


def gather_blocks(local, layout, array_type, group=None, dst=None, within_atom_proc=False, control=None):
    """Assemble a full HOST array of `array_type` ('e', 'ep', 'ep2', 'epp') from the blocks the ranks hold under the
    layout's processor grid -- rows by atom-processor, columns by parameter-processor (`layout.owned_blocks`) -- through
    the control plane (gather_local_array, distlayout.py:1010-1156).

    dst : None -> every rank receives the assembled array; int -> only that rank does (the others get None).
    within_atom_proc : only the blocks of this rank's own atom-processor are placed (what fill_jtj needs: whole rows of
        the rank's own atoms); rows of other atom-processors keep the local array's content."""
    ctl = _ctl(control)
    loc = np.ascontiguousarray(local)
    rank, size = ctl.rank, ctl.size
    if array_type in ("e",) or layout.processor_grid[1] * layout.processor_grid[2] == 1:
        if not within_atom_proc:
            return gather_elements(loc, layout, None, dst, ctl)

    def cut(arr, blk):
        r0, r1, c1, c2 = blk
        v = arr[r0:r1]
        if c1 is not None: v = v[:, c1]
        if c2 is not None: v = v[:, :, c2]
        return v
    blocks = [layout.owned_blocks(array_type, r) for r in range(size)]
    counts = [sum(cut(loc, b).size for b in bl) for bl in blocks]          # (shapes only: views)
    pad = max(max(counts), 1)
    send = np.zeros(pad, loc.dtype)
    off = 0
    for b in blocks[rank]:
        v = np.ascontiguousarray(cut(loc, b)).reshape(-1)
        send[off:off + v.size] = v
        off += v.size
    recv = _exchange_padded(ctl, send, dst)
    if recv is None:
        return None
    out = loc.copy() if within_atom_proc else np.empty_like(loc)
    na, np1, np2 = layout.processor_grid
    for r in range(size):
        if within_atom_proc and r // (np1 * np2) != layout.atom_proc_index:
            continue
        buf = recv[r]; off = 0
        for b in blocks[r]:
            v = cut(out, b)
            v[...] = buf[off:off + v.size].reshape(v.shape)
            off += v.size
    return out


def allreduce_sum_host(arr, group=None, expect_size=1, comm=None, control=None):
    """In-place sum of a host numpy array over the ranks (`allreduce_sum`, resourceallocation.py:441-508).

    expect_size : the number of ranks the CALLER's layout was built for.  When it is > 1 the sum must really happen:
        through the control plane this process joined (pygsti_amd.dist.init), else through `comm` -- an mpi4py-style
        communicator (the reference's `ResourceAllocation.comm`) -- else a RuntimeError: handing back one rank's
        partial J^T J as if it were the sum would be silent corruption."""
    ctl = control if control is not None else _control.current()
    if ctl is None:
        if expect_size > 1:
            if comm is not None and hasattr(comm, "Allreduce"):
                buf = np.ascontiguousarray(arr)
                out = np.empty_like(buf)
                comm.Allreduce(buf, out)          # mpi4py: op defaults to SUM
                arr[...] = out
                return arr
            raise RuntimeError("layout spans %d ranks but this process has neither a control plane "
                               "(pygsti_amd.dist.init) nor an MPI communicator: cannot sum over ranks" % expect_size)
        return arr
    if ctl.size == 1:
        if expect_size > 1:
            raise RuntimeError("layout spans %d ranks but the process group has one" % expect_size)
        return arr
    return ctl.allreduce_sum(arr)
