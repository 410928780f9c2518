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
    from pygsti_amd import modelpacks as MP, dist as gdist
    from pygsti_amd.layout import HipCOPALayout
    from _interp import run_programs
    ctx = gdist.init(want_comm=False)          # the control plane (gloo); no device communicator on a CPU box
    assert (ctx.rank, ctx.size) == (rank, size) and ctx.comm is None
    from pygsti_amd import control as CTL
    assert isinstance(ctx.control, CTL.ControlPlane) and CTL.current() is ctx.control and "torch" not in gdist.__dict__
    pack = MP.smq1Q_XYI
    model = pack.target_model().depolarize(0.01, 0.01)
    circuits = pack.create_gst_circuits(8)
    lay = HipCOPALayout(circuits, model, num_atoms=4, rank=rank, size=size)
    G, R, E = lay.model_arrays(model)
    local = np.full((lay.num_elements, 3), np.nan)
    for atom in lay.atoms:                          # only this rank's atoms
        w, off = atom.plan().program()
        n = len(atom.circuit_indices); nO = lay.num_outcomes
        o, written, _ = run_programs(w, off, G, R, E, np.arange(n + 1) * nO, np.tile(np.arange(nO), n), np.arange(n * nO), n * nO)
        local[atom.element_slice, 0] = o
        local[atom.element_slice, 1] = rank
        local[atom.element_slice, 2] = np.arange(atom.element_slice.start, atom.element_slice.stop, dtype=np.float64)
    full = gdist.gather_elements(local, lay)                  # all ranks
    root = gdist.gather_elements(local, lay, dst=0)           # rank 0 only
    ok_root = (root is None) if rank != 0 else bool(np.array_equal(root, full))
    # the layout's own gather (distlayout.py:1010-1156 semantics) on a host numpy array, + block bookkeeping
    arr = local.copy()
    g_all = lay.allgather_local_array("ep", arr)
    g_root = lay.gather_local_array("ep", arr)
    ok_root = ok_root and np.array_equal(g_all, full) and ((g_root is None) if rank != 0 else np.array_equal(g_root, g_all))
    blocks = gdist.row_blocks(lay, size)
    # sequential dealing (distlayout.py:327-329): 4 atoms on 2 ranks -> owners 0 0 1 1, a rank's rows one contiguous range
    ok_root = ok_root and [b[0] for b in blocks] == [a * size // len(lay.all_atoms) for a in range(len(lay.all_atoms))] \
        and sum(b[2] for b in blocks) == lay.global_num_elements \
        and lay.local_element_slice == slice(lay.atoms[0].element_slice.start, lay.atoms[-1].element_slice.stop) \
        and sum(a.num_elements for a in lay.atoms) == lay.local_element_slice.stop - lay.local_element_slice.start
    part = np.full(3, float(rank + 1)); gdist.allreduce_sum_host(part)
    ok_root = ok_root and np.array_equal(part, np.full(3, 3.0)) and ctx.max_over_ranks(rank) == 1.0
    q.put((rank, full, ok_root, [(a.element_slice.start, a.element_slice.stop) for a in lay.atoms]))
    ctx.shutdown()


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
        assert (owners[at.element_slice] == a // 2).all()          # sequential blocks: atoms 0,1 -> rank 0; 2,3 -> rank 1


def _grid_worker(rank, size, port, q, grid, n_atoms):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(size))
    from pygsti_amd import modelpacks as MP, dist as gdist
    from pygsti_amd.layout import HipCOPALayout
    ctx = gdist.init(want_comm=False)
    pack = MP.smq1Q_XYI
    model = pack.target_model().depolarize(0.01, 0.01)
    lay = HipCOPALayout(pack.create_gst_circuits(2), model, num_atoms=n_atoms, rank=rank, size=size, processor_grid=grid)
    nE, nP = lay.global_num_elements, model.num_params
    want = {"e": np.arange(nE) * 1.0,
            "ep": np.arange(nE)[:, None] * 1000.0 + np.arange(nP)[None, :],
            "epp": np.arange(nE)[:, None, None] * 1e6 + np.arange(nP)[None, :, None] * 1e3 + np.arange(nP)[None, None, :]}
    ok = True
    for t, full in want.items():
        # what THIS rank's fills produce: its atoms' rows, its parameter slice(s) -- everything else NaN
        loc = np.full(full.shape, np.nan)
        for at in lay.atoms:
            es = at.element_slice
            if t == "e": loc[es] = full[es]
            elif t == "ep": loc[es, lay.host_param_slice] = full[es, lay.host_param_slice]
            else: loc[es, lay.host_param_slice, lay.host_param2_slice] = full[es, lay.host_param_slice, lay.host_param2_slice]
        g_all = lay.allgather_local_array(t, loc)
        g_root = lay.gather_local_array(t, loc)
        ok = ok and np.array_equal(g_all, full) and ((g_root is None) if rank != 0 else np.array_equal(g_root, full))
        if t == "ep":       # whole rows of the rank's own atoms (what fill_jtj contracts), other atom-processors' rows untouched
            comp = lay._complete_columns(loc)
            for a, at in enumerate(lay.all_atoms):
                mine = lay._atom_proc[a] == lay.atom_proc_index
                ok = ok and (np.array_equal(comp[at.element_slice], full[at.element_slice]) if mine else np.isnan(comp[at.element_slice]).all())
    shares = [(lay._row_share(at).start, lay._row_share(at).stop) for at in lay.atoms]
    q.put((rank, ok, shares, [(at.element_slice.start, at.element_slice.stop) for at in lay.atoms]))
    ctx.shutdown()


@pytest.mark.parametrize("grid,n_atoms", [((1, 2), 2), ((2, 2), 4), ((1, 2, 2), 1)])
def test_processor_grid_gathers_on_cpu(grid, n_atoms):
    """Parameter-processors (distlayout.py:424-660) over gloo: every rank holds its atoms' rows x its parameter slice(s) of
    full-size host arrays; gather / all-gather of 'e', 'ep', 'epp' arrays assemble the global array, `_complete_columns`
    gives an atom-processor's ranks whole rows, and their row shares partition each atom for the JtJ products."""
    import torch.multiprocessing as mp
    size = int(np.prod(grid))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() * 3 + size * 31 + n_atoms) % 2000
    procs = [ctx.Process(target=_grid_worker, args=(r, size, port, q, grid, n_atoms)) for r in range(size)]
    for p in procs: p.start()
    res = sorted([q.get(timeout=240) for _ in range(size)], key=lambda t: t[0])
    for p in procs: p.join(timeout=60)
    assert all(r[1] for r in res)
    per = size // grid[0]
    for ap in range(grid[0]):
        group = res[ap * per:(ap + 1) * per]
        atoms = group[0][3]
        assert all(g[3] == atoms for g in group)
        for k, (a0, a1) in enumerate(atoms):      # the group's row shares tile each of its atoms
            cuts = sorted(g[2][k] for g in group)
            assert cuts[0][0] == a0 and cuts[-1][1] == a1 and all(cuts[i][1] == cuts[i + 1][0] for i in range(len(cuts) - 1))
