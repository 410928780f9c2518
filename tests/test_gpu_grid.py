"""The processor grid of the reference's distributed layouts (distforwardsim.py:445-485 `_compute_processor_distribution`,
distlayout.py:424-660): atoms over `na` atom-processors AND parameter columns over the np1 (x np2) parameter-processors
of each, on hardware -- N processes sharing GPU 0, every rank filling only its own (rows, columns) block, the layout's
gathers and `fill_jtj` / `fill_jtf` assembling what a single process computes."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, load_fixture, assert_bitwise
from pygsti_amd.layout import _slice_up_range

pytestmark = pytest.mark.gpu


def _run(tmp_path, n_atoms, grid):
    n_ranks = int(np.prod(grid))
    port = 29500 + (os.getpid() * 11 + n_ranks * 17 + n_atoms * 3 + len(grid)) % 2000
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE=str(n_ranks),
                   LOCAL_RANK=str(r), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_grid_worker.py"), str(tmp_path), str(n_atoms),
                                       "x".join(str(g) for g in grid)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o.decode(errors="replace"))
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, outs[r][-3000:])
    return [dict(np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))) for r in range(n_ranks)]


@pytest.fixture(scope="module")
def single():
    """What one process computes for the same layout (4 atoms): the reference values of every comparison below."""
    from test_host_mirror import _model_from_fixture
    from pygsti_amd import modelpacks as MP
    from pygsti_amd.forwardsim import HipMapForwardSimulator
    fx = load_fixture("smq1Q_XYI_L4_depol")
    pack = MP.smq1Q_XYI
    model = _model_from_fixture(fx, pack)
    out = {}
    for n_atoms in (1, 2, 4):
        sim = HipMapForwardSimulator(model, num_atoms=n_atoms, devices=[0])
        lay = sim.create_layout(pack.create_gst_circuits(4), array_types=("e", "ep", "epp"))
        nE, nP = lay.global_num_elements, model.num_params
        J = np.empty((nE, nP)); P = np.empty(nE); H = np.empty((nE, nP, nP))
        sim.bulk_fill_dprobs(J, lay, P); sim.bulk_fill_hprobs(H, lay)
        counts = np.round(1000.0 * (0.5 + 0.4 * np.sin(np.arange(nE) * 0.77)))
        jtj_l = np.empty((nP, nP)); jtf_l = np.empty(nP); ls = np.empty(nE); hess = np.empty((nP, nP))
        obj = sim.bulk_fill_lsq_step(jtj_l, jtf_l, lay, counts, np.full(nE, 1000.0), objective="chi2", lsvec_to_fill=ls)
        sim.bulk_fill_objective_hessian(hess, lay, counts, np.full(nE, 1000.0), objective="chi2", row_block=30)
        out[n_atoms] = (J, P, H, (jtj_l, jtf_l, obj, ls, hess))
    # the element order depends on the atoms; the values, circuit by circuit, are the fixture's
    idx = np.concatenate([np.arange(*lay.indices_for_index(i).indices(nE)) for i in range(lay.num_circuits)])
    assert_bitwise(out[4][0][idx], fx["dprobs_map"], "single-process Jacobian vs the reference")
    return out


@pytest.mark.parametrize("n_atoms,grid", [(1, (1, 2)), (2, (2, 2)), (4, (2, 1, 2)), (1, (1, 2, 2)), (2, (1, 3))])
def test_processor_grid_assembles_the_single_process_results(tmp_path, single, n_atoms, grid):
    res = _run(tmp_path, n_atoms, grid)
    J, P, H, (jtj_l, jtf_l, obj_l, ls_l, hess_l) = single[n_atoms]
    nE, nP = J.shape
    na, np1, np2 = (tuple(grid) + (1, 1))[:3]
    f = np.sin(np.arange(nE) * 0.37) + 0.1
    jtj, jtf = J.T @ J, J.T @ f
    w = 1.0 + 0.5 * np.cos(np.arange(nE) * 0.11)
    Jw = J * w[:, None]
    jtj_w, jtf_w = Jw.T @ Jw, Jw.T @ f
    for r, d in enumerate(res):
        assert_bitwise(d["J_all"], J, "all-gathered Jacobian on rank %d" % r)
        assert_bitwise(d["P_all"], P, "all-gathered probabilities on rank %d" % r)
        assert_bitwise(d["H_all"], H, "all-gathered Hessian on rank %d" % r)
        assert bool(d["root_none"]) == (r != 0)
        assert np.abs(d["jtj"] - jtj).max() <= 1e-12 * np.abs(jtj).max()
        assert np.abs(d["jtf"] - jtf).max() <= 1e-12 * np.abs(jtf).max()
        # device-resident route (bulk_fill_jtj_jtf: column blocks exchanged between device buffers, gst_comm_exchange_blocks)
        assert np.abs(d["jtj_d"] - jtj_w).max() <= 1e-12 * np.abs(jtj_w).max()
        assert np.abs(d["jtf_d"] - jtf_w).max() <= 1e-12 * np.abs(jtf_w).max()
        assert_bitwise(d["P_d_all"], P, "probabilities of the device-resident route on rank %d" % r)
        # no device communicator attached: parameter-processor (0, 0) alone contributes; and the fused LM step /
        # objective Hessian count every atom once whatever the grid (they over-counted by np1*np2 before round 4)
        assert np.abs(d["jtj_n"] - jtj_w).max() <= 1e-12 * np.abs(jtj_w).max()
        assert np.abs(d["jtf_n"] - jtf_w).max() <= 1e-12 * np.abs(jtf_w).max()
        assert np.abs(d["jtj_l"] - jtj_l).max() <= 1e-12 * np.abs(jtj_l).max()
        assert np.abs(d["jtf_l"] - jtf_l).max() <= 1e-12 * np.abs(jtf_l).max()
        assert abs(float(d["obj"][0]) - obj_l) <= 1e-12 * abs(obj_l)
        assert_bitwise(d["ls_all"], ls_l, "lsvec rows under the grid on rank %d" % r)
        assert np.abs(d["hess"] - hess_l).max() <= 1e-11 * np.abs(hess_l).max()
        # the rank really computed only its block: rows of its atom-processor x its column slice(s)
        g = d["gps"]
        q = r % (np1 * np2)
        s1, s2 = _slice_up_range(nP, np1)[q // np2], _slice_up_range(nP, np2)[q % np2]
        assert (g[0], g[1], g[2], g[3]) == (s1.start, s1.stop, s2.start, s2.stop)
        rows = int(d["n_filled"]) // (g[1] - g[0])
        assert 0 < rows <= nE and rows * (g[1] - g[0]) == int(d["n_filled"]) and int(d["h_filled"]) == rows * (g[1] - g[0]) * (g[3] - g[2])
        assert (rows < nE) == (na > 1)
    assert_bitwise(res[0]["J_root"], J, "Jacobian gathered to rank 0")
    # the atom-processors partition the atoms, their parameter-processors the columns
    by_ap = {}
    for r, d in enumerate(res):
        by_ap.setdefault(r // (np1 * np2), []).append(d)
    assert len(by_ap) == na
    starts = [tuple(v[0]["owned"].tolist()) for v in by_ap.values()]
    assert sum(len(s) for s in starts) == n_atoms and len(set(sum((list(s) for s in starts), []))) == n_atoms
    for v in by_ap.values():
        assert all(tuple(d["owned"].tolist()) == tuple(v[0]["owned"].tolist()) for d in v)
