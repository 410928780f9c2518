"""Worker of tests/test_gpu_grid.py: one rank of an N-process job on ONE box (the ranks share GPU 0).

Every rank builds the layout with the same processor grid (na, np1[, np2]) -- atoms over the atom-processors, parameter
columns over the parameter-processors of each (distforwardsim.py:445-485, distlayout.py:424-660) -- fills ONLY its own
block of the probabilities, the FD Jacobian and an FD-of-FD Hessian through the host-mirror simulator, and the layout's
gathers / normal-equation products assemble the global results.  They go to <out>/rank<r>.npz for the parent test."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    out_dir, n_atoms = sys.argv[1], int(sys.argv[2])
    grid = tuple(int(x) for x in sys.argv[3].split("x"))
    from conftest import load_fixture
    from test_host_mirror import _model_from_fixture
    from pygsti_amd import modelpacks as MP, dist as gdist
    from pygsti_amd.forwardsim import HipMapForwardSimulator
    ctx = gdist.init(device=0, transport="ipc")          # host arrays travel through the control group, device blocks over IPC
    assert ctx.comm is not None, ctx.comm_error
    rank, size = ctx.rank, ctx.size
    fx = load_fixture("smq1Q_XYI_L4_depol")
    pack = MP.smq1Q_XYI
    model = _model_from_fixture(fx, pack)
    circuits = pack.create_gst_circuits(4)
    sim = HipMapForwardSimulator(model, num_atoms=n_atoms, processor_grid=grid, devices=[0])
    lay = sim.create_layout(circuits, resource_alloc=gdist.RankAlloc(rank, size), array_types=("e", "ep", "epp"))
    nE, nP = lay.global_num_elements, model.num_params
    J = lay.allocate_local_array("ep"); J[...] = np.nan
    P = lay.allocate_local_array("e"); P[...] = np.nan
    sim.bulk_fill_dprobs(J, lay, P)
    filled = ~np.isnan(J)
    J_all = lay.allgather_local_array("ep", J)
    J_root = lay.gather_local_array("ep", J)
    P_all = lay.allgather_local_array("e", P)
    f = np.sin(np.arange(nE) * 0.37) + 0.1
    jtj = np.empty((nP, nP)); jtf = np.empty(nP)
    lay.fill_jtj(J, jtj); lay.fill_jtf(J, f, jtf)
    H = lay.allocate_local_array("epp"); H[...] = np.nan
    sim.bulk_fill_hprobs(H, lay)
    H_all = lay.allgather_local_array("epp", H)
    # the same products without the Jacobian leaving the GPUs: column blocks exchanged between device buffers
    lay.device_comm = ctx.comm
    w = 1.0 + 0.5 * np.cos(np.arange(nE) * 0.11)
    jtj_d = np.empty((nP, nP)); jtf_d = np.empty(nP); P_d = np.full(nE, np.nan)
    sim.bulk_fill_jtj_jtf(jtj_d, jtf_d, lay, row_scale=w, f=f, pr_array_to_fill=P_d)
    gdist.allreduce_sum_host(jtj_d, expect_size=size); gdist.allreduce_sum_host(jtf_d, expect_size=size)
    P_d_all = lay.allgather_local_array("e", P_d)
    # ... and without a device communicator attached: one rank per atom-processor contributes, the sums stay right
    lay.device_comm = None
    jtj_n = np.empty((nP, nP)); jtf_n = np.empty(nP)
    sim.bulk_fill_jtj_jtf(jtj_n, jtf_n, lay, row_scale=w, f=f)
    gdist.allreduce_sum_host(jtj_n, expect_size=size); gdist.allreduce_sum_host(jtf_n, expect_size=size)
    # one Levenberg-Marquardt step's data reduction and the objective Hessian under the grid (every atom counted once)
    counts = np.round(1000.0 * (0.5 + 0.4 * np.sin(np.arange(nE) * 0.77)))
    Ntot = np.full(nE, 1000.0)
    jtj_l = np.empty((nP, nP)); jtf_l = np.empty(nP); ls = np.full(nE, np.nan)
    obj = np.array([sim.bulk_fill_lsq_step(jtj_l, jtf_l, lay, counts, Ntot, objective="chi2", lsvec_to_fill=ls)])
    gdist.allreduce_sum_host(jtj_l, expect_size=size); gdist.allreduce_sum_host(jtf_l, expect_size=size)
    gdist.allreduce_sum_host(obj, expect_size=size)
    ls_all = lay.allgather_local_array("e", ls)
    hess = np.empty((nP, nP))
    sim.bulk_fill_objective_hessian(hess, lay, counts, Ntot, objective="chi2", row_block=30)
    gdist.allreduce_sum_host(hess, expect_size=size)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), jtj_n=jtj_n, jtf_n=jtf_n, jtj_l=jtj_l, jtf_l=jtf_l, obj=obj, ls_all=ls_all, hess=hess, jtj_d=jtj_d, jtf_d=jtf_d, P_d_all=P_d_all, J_all=J_all, J_root=J_root if rank == 0 else np.zeros(0), P_all=P_all,
             jtj=jtj, jtf=jtf, H_all=H_all, n_filled=int(filled.sum()), h_filled=int((~np.isnan(H)).sum()),
             gps=np.array([lay.global_param_slice.start, lay.global_param_slice.stop, lay.global_param2_slice.start, lay.global_param2_slice.stop]),
             owned=np.array([a.element_slice.start for a in lay.atoms]), root_none=(J_root is None))
    ctx.shutdown()


if __name__ == "__main__":
    main()
