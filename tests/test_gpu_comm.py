"""The multi-GPU exchange through the C ABI (gst_comm_*), on hardware.

A 1-GPU box cannot host two RCCL ranks (RCCL refuses ranks that share a device), so:
  * the RCCL binding (dlopen, communicator, grouped send/recv, all-reduce, barrier) is exercised with a one-rank
    communicator on the real device;
  * the N-rank data path -- per-rank fills into device arrays, row blocks between DEVICE buffers, no host staging --
    runs as two / three processes sharing GPU 0 over the IPC transport, and the assembled Jacobian must equal the
    reference's single-process golden vector BIT FOR BIT (north star: atoms over ranks, gather of Jacobian blocks)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, load_fixture, assert_bitwise

pytestmark = pytest.mark.gpu


def test_rccl_one_rank_communicator_on_device():
    from pygsti_amd import _lib
    from conftest import plan_from_fixture
    uid = _lib.Comm.unique_id(_lib.TRANSPORT_RCCL)
    comm = _lib.Comm(0, 1, uid, 0, _lib.TRANSPORT_RCCL)
    info = comm.info()
    assert info["transport"] == "rccl" and info["size"] == 1 and info["rccl_version"] >= 21800, info
    fx = load_fixture("smq1Q_XYI_L4_depol")
    pl = plan_from_fixture(fx, device=0)
    nE = int(fx["nE"]); nP = 60
    d_J = pl.device_malloc(nE * nP * 8); d_p = pl.device_malloc(nE * 8)
    pl.fill_dprobs_dev(d_J, nP, np.arange(nP), None, 1e-7, d_p)
    # stream-ordered behind the fill, on the plan's stream
    comm.allgather_rows(d_J, nP, [(0, 0, nE)], pl)
    comm.gather_rows(None, d_J, nP, [(0, 0, nE)], 0, pl)
    comm.allreduce_sum(d_p, nE, pl)
    pl.sync()
    comm.barrier()
    J = np.empty((nE, nP)); pl.memcpy_d2h(J, d_J)
    p = np.empty(nE); pl.memcpy_d2h(p, d_p)
    assert_bitwise(J, fx["dprobs_map"], "Jacobian after the one-rank exchange")
    assert_bitwise(p, fx["probs"], "probabilities after the one-rank all-reduce")
    pl.device_free(d_J); pl.device_free(d_p)
    comm.close()


def test_rccl_send_recv_group_executes_with_one_rank(monkeypatch):
    """The branch a one-rank job never enters: ncclGroupStart ... ncclSend / ncclRecv ... ncclGroupEnd with a PEER.  RCCL allows
    a rank to send to itself inside a group, so the test hook `comm_self=1` routes the blocks that stay on their rank through
    exactly that code (gst_comm_exchange_blocks) instead of a local copy: the first multi-GPU run is then not the first
    execution of it.  Same block list as the column exchange of the processor grid: several blocks, ragged sizes, a gap."""
    from pygsti_amd import _lib
    from conftest import plan_from_fixture, force
    force(monkeypatch, comm_self=1)
    uid = _lib.Comm.unique_id(_lib.TRANSPORT_RCCL)
    comm = _lib.Comm(0, 1, uid, 0, _lib.TRANSPORT_RCCL)
    fx = load_fixture("smq1Q_XYI_L4_depol")
    pl = plan_from_fixture(fx, device=0)
    n = 100_000
    src = np.random.default_rng(5).standard_normal(n)
    d_src = pl.device_malloc(n * 8); d_dst = pl.device_malloc(n * 8)
    pl.memcpy_h2d(d_src, src); pl.memcpy_h2d(d_dst, np.full(n, np.nan))
    # (src rank, dst rank, src offset, dst offset, count) in doubles
    blocks = [(0, 0, 0, 50_000, 1_000), (0, 0, 1_000, 0, 7), (0, 0, 20_000, 60_000, 33_333), (0, 0, 99_999, 99_999, 1)]
    comm.exchange_blocks(d_src, d_dst, blocks, pl)
    pl.sync()
    comm.barrier()
    got = np.empty(n); pl.memcpy_d2h(got, d_dst)
    want = np.full(n, np.nan)
    for _, _, so, do, cnt in blocks:
        want[do:do + cnt] = src[so:so + cnt]
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(got[~np.isnan(want)], want[~np.isnan(want)])
    pl.device_free(d_src); pl.device_free(d_dst)
    comm.close()


def _run_job(tmp_path, n_ranks, transport, n_atoms, distinct=False):
    port = 29500 + (os.getpid() * 7 + n_ranks * 13 + n_atoms) % 2000
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE=str(n_ranks),
                   LOCAL_RANK=str(r), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_comm_worker.py"), str(tmp_path),
                                       transport, str(n_atoms)] + (["distinct"] if distinct else []), env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
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


@pytest.mark.parametrize("n_ranks,n_atoms,transport", [(2, 2, "ipc"), (2, 4, "ipc"), (3, 5, "ipc"), (8, 8, "ipc"), (2, 2, "auto")])
def test_ranks_sharing_one_gpu_assemble_the_reference_jacobian_bitwise(tmp_path, n_ranks, n_atoms, transport):
    fx = load_fixture("smq2Q_XYICNOT_L2_depol")
    res = _run_job(tmp_path, n_ranks, transport, n_atoms)
    v = [np.sin(np.arange(5000) * (r + 1.0)) * 10.0 ** (r - 3) for r in range(n_ranks)]
    for r, d in enumerate(res):
        assert str(d["transport"]) in ("ipc", "rccl")
        assert int(d["reopens"]) == 0, "alternating destinations re-opened %d IPC mappings on rank %d" % (int(d["reopens"]), r)
        assert_bitwise(d["J"], fx["dprobs_map"], "all-gathered Jacobian on rank %d" % r)
        assert_bitwise(d["P"], fx["probs"], "all-gathered probabilities on rank %d" % r)
        # rank-order sums: bit-reproducible under the IPC transport
        for key, n in (("s0", 1000), ("s1", 5000)):
            want = v[0][:n].copy()
            for q in range(1, n_ranks):
                want = want + v[q][:n]
            if str(d["transport"]) == "ipc":
                assert_bitwise(d[key], want, "all-reduce %s on rank %d" % (key, r))
            else:
                np.testing.assert_allclose(d[key], want, rtol=1e-14, atol=1e-300)
    assert_bitwise(res[0]["J_root"], fx["dprobs_map"], "Jacobian gathered to rank 0 (Gatherv)")
    # the fan-in without a copy: every rank's fill wrote its rows straight into rank 0's array (gst_comm_map_root_buffer)
    assert_bitwise(res[0]["J_direct"], fx["dprobs_map"], "Jacobian filled directly into rank 0's assembled array")
    owned = [set(d["owned"].tolist()) for d in res]
    assert not set.intersection(*owned) and sum(len(o) for o in owned) == n_atoms


def _n_devices():
    from pygsti_amd import _lib
    try:
        return _lib.device_count()
    except Exception:
        return 0


@pytest.mark.skipif(_n_devices() < 2, reason="needs a multi-GPU node: one MI355X per rank (RCCL refuses ranks that share a device)")
@pytest.mark.parametrize("n_ranks", [2, 4, 8])
def test_ranks_on_distinct_gpus_over_rccl_assemble_the_reference_jacobian_bitwise(tmp_path, n_ranks):
    """BASELINE configs[3] as far as a test can carry it: N ranks on N DISTINCT devices, the data plane forced to
    TRANSPORT_RCCL (grouped ncclSend/ncclRecv over xGMI for the row blocks, ncclAllReduce for the sums) -- no IPC
    fall-back is accepted.  The all-gathered and the gathered-to-rank-0 Jacobians must be the reference's single-process
    vectors bit for bit (moving rows changes no bit), the all-reduced sums agree with the rank-order sum to 1e-14
    relative (RCCL's reduction order is its own).  Skipped on the 1-GPU boxes of this build; the round-end driver's
    8-GPU node runs it."""
    if _n_devices() < n_ranks:
        pytest.skip("%d devices on this node" % _n_devices())
    fx = load_fixture("smq2Q_XYICNOT_L2_depol")
    res = _run_job(tmp_path, n_ranks, "rccl", n_ranks, distinct=True)
    v = [np.sin(np.arange(5000) * (r + 1.0)) * 10.0 ** (r - 3) for r in range(n_ranks)]
    for r, d in enumerate(res):
        assert str(d["transport"]) == "rccl", "rank %d ended on the %s transport" % (r, d["transport"])
        assert_bitwise(d["J"], fx["dprobs_map"], "all-gathered Jacobian on rank %d" % r)
        assert_bitwise(d["P"], fx["probs"], "all-gathered probabilities on rank %d" % r)
        for key, n in (("s0", 1000), ("s1", 5000)):
            want = v[0][:n].copy()
            for q in range(1, n_ranks):
                want = want + v[q][:n]
            np.testing.assert_allclose(d[key], want, rtol=1e-14, atol=1e-300)
    assert_bitwise(res[0]["J_root"], fx["dprobs_map"], "Jacobian gathered to rank 0 (Gatherv) over RCCL")
    assert_bitwise(res[0]["J_direct"], fx["dprobs_map"], "Jacobian filled directly into rank 0's array (IPC handle sent through RCCL)")
    owned = [set(d["owned"].tolist()) for d in res]
    assert sum(len(o) for o in owned) == n_ranks and len(set.union(*owned)) == n_ranks
