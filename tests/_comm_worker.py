"""Worker of tests/test_gpu_comm.py: one rank of an N-process job on ONE box (ranks may share GPU 0).

Every rank builds the same multi-atom layout of smq2Q_XYICNOT L<=2, fills ONLY its own atoms' rows of the FD Jacobian
(the golden fixture's 96 columns) and of the probabilities straight into full-size DEVICE arrays through the C ABI, then
the row blocks travel between device buffers (gst_comm_*): all-gather, gather to rank 0, all-reduce.  Results go to
<out>/rank<r>.npz for the parent test to compare bit for bit with the reference's single-process vectors."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    out_dir, transport, n_atoms = sys.argv[1], sys.argv[2], int(sys.argv[3])
    distinct = len(sys.argv) > 4 and sys.argv[4] == "distinct"      # one GPU per rank (a multi-GPU node): device = LOCAL_RANK
    dev = int(os.environ.get("LOCAL_RANK", "0")) if distinct else 0
    from conftest import load_fixture
    from test_host_mirror import _model_from_fixture
    from pygsti_amd import modelpacks as MP, dist as gdist, _lib
    from pygsti_amd.layout import HipCOPALayout
    ctx = gdist.init(device=dev, transport=transport)
    assert ctx.comm is not None, ctx.comm_error
    rank, size = ctx.rank, ctx.size
    fx = load_fixture("smq2Q_XYICNOT_L2_depol")
    pack = MP.smq2Q_XYICNOT
    model = _model_from_fixture(fx, pack)
    circuits = pack.create_gst_circuits(2)
    lay = HipCOPALayout(circuits, model, num_atoms=n_atoms, devices=[dev], rank=rank, size=size)
    cols = np.asarray(fx["dprobs_cols"], np.int64)
    nE, nC = lay.global_num_elements, len(cols)
    plans = [at.plan() for at in lay.atoms]
    p0 = plans[0]
    d_J = p0.device_malloc(nE * nC * 8); d_P = p0.device_malloc(nE * 8)
    p0.memcpy_h2d(d_J, np.full(nE * nC, np.nan)); p0.memcpy_h2d(d_P, np.full(nE, np.nan))
    G, R, E = lay.model_arrays(model)
    for at, pl in zip(lay.atoms, plans):
        pl.set_model(G, R, E); pl.set_param_map(*lay.param_map(model))
        r0 = at.element_slice.start
        pl.fill_dprobs_dev(d_J + r0 * nC * 8, nC, cols, None, 1e-7, d_P + r0 * 8)
        pl.sync()
    # (1) Gatherv to rank 0 from packed local blocks (what non-root ranks of a real job hold)
    n_loc = sum(at.num_elements for at in lay.atoms)
    d_loc = p0.device_malloc(max(n_loc, 1) * nC * 8)
    off = 0
    for at in lay.atoms:
        tmp = np.empty((at.num_elements, nC)); p0.memcpy_d2h(tmp, d_J + at.element_slice.start * nC * 8)
        p0.memcpy_h2d(d_loc + off * nC * 8, tmp); off += at.num_elements
    d_root = p0.device_malloc(nE * nC * 8)
    if rank == 0:
        p0.memcpy_h2d(d_root, np.full(nE * nC, np.nan))
        for at in lay.atoms:       # the root's own blocks are in place
            tmp = np.empty((at.num_elements, nC)); p0.memcpy_d2h(tmp, d_J + at.element_slice.start * nC * 8)
            p0.memcpy_h2d(d_root + at.element_slice.start * nC * 8, tmp)
    gdist.gather_elements_dev(ctx, lay, d_loc, d_root if rank == 0 else None, nC, 0, p0)
    p0.sync(); ctx.barrier()
    J_root = np.empty((nE, nC)); p0.memcpy_d2h(J_root, d_root)
    # (2) all-gather in place (Jacobian rows and probabilities)
    gdist.allgather_elements_dev(ctx, lay, d_J, nC, p0)
    gdist.allgather_elements_dev(ctx, lay, d_P, 1, p0)
    p0.sync(); ctx.barrier()
    J = np.empty((nE, nC)); p0.memcpy_d2h(J, d_J)
    P = np.empty(nE); p0.memcpy_d2h(P, d_P)
    # (2b) alternating destinations A / B / A / B: a re-published allocation keeps its id, so no peer re-opens it
    opens0 = ctx.comm.info()["ipc_opens"]
    for _ in range(3):
        gdist.allgather_elements_dev(ctx, lay, d_J, nC, p0)
        gdist.allgather_elements_dev(ctx, lay, d_P, 1, p0)
    p0.sync(); ctx.barrier()
    reopens = ctx.comm.info()["ipc_opens"] - opens0
    # (3) all-reduce of a rank-dependent vector, twice with different lengths (staging growth)
    sums = []
    for n in (1000, 5000):
        v = np.sin(np.arange(n) * (rank + 1.0)) * 10.0 ** (rank - 3)
        d_v = p0.device_malloc(n * 8); p0.memcpy_h2d(d_v, v)
        ctx.comm.allreduce_sum(d_v, n, p0); p0.sync()
        got = np.empty(n); p0.memcpy_d2h(got, d_v); sums.append(got)
        p0.device_free(d_v)
    # (4) the fan-in without a copy: rank 0's assembled array mapped into every rank; the fills write their rows there directly
    d_direct = p0.device_malloc(nE * nC * 8) if rank == 0 else 0
    if rank == 0:
        p0.memcpy_h2d(d_direct, np.full(nE * nC, np.nan))
    mapped = ctx.comm.map_root_buffer(d_direct, 0)
    assert mapped != 0 and (rank != 0 or mapped == d_direct)
    for at, pl in zip(lay.atoms, plans):
        pl.fill_dprobs_dev(mapped + at.element_slice.start * nC * 8, nC, cols, None, 1e-7, None)
        pl.sync()
    ctx.barrier()
    J_direct = np.zeros((0, nC))
    if rank == 0:
        J_direct = np.empty((nE, nC)); p0.memcpy_d2h(J_direct, d_direct)
    ctx.barrier()
    idx = np.concatenate([np.arange(*lay.indices_for_index(i).indices(nE)) for i in range(len(circuits))])
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), J=J[idx], P=P[idx], J_root=J_root[idx] if rank == 0 else np.zeros(0), J_direct=J_direct[idx] if rank == 0 else np.zeros(0),
             s0=sums[0], s1=sums[1], transport=ctx.transport, reopens=reopens, owned=np.array([a.element_slice.start for a in lay.atoms]))
    for d in (d_J, d_P, d_loc, d_root) + ((d_direct,) if rank == 0 else ()):
        p0.device_free(d)
    ctx.shutdown()


if __name__ == "__main__":
    main()
