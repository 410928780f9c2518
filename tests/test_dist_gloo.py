"""N>1 path on CPU: two processes (gloo, 127.0.0.1) shard a layout into atoms, each fills ONLY its own
rows (here through the numpy program interpreter standing in for the device) and the row blocks are
assembled with pygsti_amd.dist.gather_elements; the result must equal the single-process array."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, size, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(size))
    import torch
    import torch.distributed as dist
    from pygsti_amd import modelpacks as MP, dist as gdist
    from pygsti_amd.layout import HipCOPALayout
    from _interp import run_programs
    ctx = gdist.init(want_comm=False)          # the control plane (gloo); no device communicator on a CPU box
    assert (ctx.rank, ctx.size) == (rank, size) and ctx.comm is None
    pack = MP.smq1Q_XYI
    model = pack.target_model().depolarize(0.01, 0.01)
    circuits = pack.create_gst_circuits(8)
    lay = HipCOPALayout(circuits, model, num_atoms=4, rank=rank, size=size)
    G, R, E = lay.model_arrays(model)
    local = torch.full((lay.num_elements, 3), float("nan"), dtype=torch.float64)
    for atom in lay.atoms:                          # only this rank's atoms
        w, off = atom.plan().program()
        n = len(atom.circuit_indices); nO = lay.num_outcomes
        o, written, _ = run_programs(w, off, G, R, E, np.arange(n + 1) * nO, np.tile(np.arange(nO), n), np.arange(n * nO), n * nO)
        local[atom.element_slice, 0] = torch.from_numpy(o)
        local[atom.element_slice, 1] = rank
        local[atom.element_slice, 2] = torch.arange(atom.element_slice.start, atom.element_slice.stop, dtype=torch.float64)
    full = gdist.gather_elements(local, lay)                  # all ranks
    root = gdist.gather_elements(local, lay, dst=0)           # rank 0 only
    ok_root = (root is None) if rank != 0 else bool(torch.equal(root, full))
    # the layout's own gather (distlayout.py:1010-1156 semantics) on a host numpy array, + block bookkeeping
    arr = local.numpy().copy()
    g_all = lay.allgather_local_array("ep", arr)
    g_root = lay.gather_local_array("ep", arr)
    ok_root = ok_root and np.array_equal(g_all, full.numpy()) and ((g_root is None) if rank != 0 else np.array_equal(g_root, g_all))
    blocks = gdist.row_blocks(lay, size)
    ok_root = ok_root and [b[0] for b in blocks] == [a % size for a in range(len(lay.all_atoms))] \
        and sum(b[2] for b in blocks) == lay.global_num_elements
    part = np.full(3, float(rank + 1)); gdist.allreduce_sum_host(part)
    ok_root = ok_root and np.array_equal(part, np.full(3, 3.0)) and ctx.max_over_ranks(rank) == 1.0
    q.put((rank, full.numpy(), ok_root, [(a.element_slice.start, a.element_slice.stop) for a in lay.atoms]))
    dist.destroy_process_group()


def test_two_rank_sharding_and_gather():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = sorted([q.get(timeout=180) for _ in range(2)], key=lambda t: t[0])
    for p in procs: p.join(timeout=60)
    (r0, full0, ok0, sl0), (r1, full1, ok1, sl1) = res
    assert ok0 and ok1
    assert np.array_equal(full0, full1) and not np.isnan(full0).any()
    assert not set(sl0) & set(sl1) and len(sl0) == len(sl1) == 2
    # assembled probabilities equal the single-process ones, element for element
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from pygsti_amd import modelpacks as MP
    from pygsti_amd.layout import HipCOPALayout
    from _interp import run_programs
    pack = MP.smq1Q_XYI
    model = pack.target_model().depolarize(0.01, 0.01)
    circuits = pack.create_gst_circuits(8)
    lay = HipCOPALayout(circuits, model, num_atoms=4)
    G, R, E = lay.model_arrays(model)
    ref = np.empty(lay.num_elements)
    for atom in lay.all_atoms:
        w, off = atom.plan().program()
        n = len(atom.circuit_indices); nO = lay.num_outcomes
        o, _, _ = run_programs(w, off, G, R, E, np.arange(n + 1) * nO, np.tile(np.arange(nO), n), np.arange(n * nO), n * nO)
        ref[atom.element_slice] = o
    assert np.array_equal(full0[:, 0].view(np.uint64), ref.view(np.uint64))
    assert np.array_equal(full0[:, 2], np.arange(lay.num_elements))
    owners = full0[:, 1]
    for a, at in enumerate(lay.all_atoms):
        assert (owners[at.element_slice] == a % 2).all()
