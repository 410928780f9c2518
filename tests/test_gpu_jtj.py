"""Normal equations on the device (kernel K10: layout.fill_jtj / fill_jtf,
pygsti/layouts/distlayout.py:1220-1359): J^T J and J^T f of a device-resident, row-scaled Jacobian against numpy
on the reference's golden Jacobian.  A plain fp64 GEMM: tolerance relative 1e-12 (summation order differs from BLAS)."""
import numpy as np
import pytest

from conftest import load_fixture, plan_from_fixture
from pygsti_amd import _lib, modelpacks as MP
from pygsti_amd.forwardsim import HipMapForwardSimulator
from test_host_mirror import _model_from_fixture

pytestmark = pytest.mark.gpu


def _close(a, b, rtol=1e-12):
    return np.abs(a - b).max() <= rtol * max(1.0, np.abs(b).max())


@pytest.mark.parametrize("name", ["smq1Q_XYI_L128_depol", "smq2Q_XYICNOT_L2_depol"])
def test_jtj_jtf_dev_vs_numpy(name):
    fx = load_fixture(name)
    pl = plan_from_fixture(fx)
    cols = fx["dprobs_cols"]; nE, n = int(fx["nE"]), len(cols)
    ld = n + 5
    d_J = pl.device_malloc(nE * ld * 8); d_jtj = pl.device_malloc(n * n * 8); d_jtf = pl.device_malloc(n * 8)
    d_w = pl.device_malloc(nE * 8); d_f = pl.device_malloc(nE * 8)
    rng = np.random.default_rng(0)
    w = rng.random(nE) + 0.5; f = rng.standard_normal(nE)
    pl.memcpy_h2d(d_w, w); pl.memcpy_h2d(d_f, f)
    J = fx["dprobs_map"]
    # unscaled
    pl.fill_dprobs_dev(d_J, ld, cols, None, 1e-7); pl.fill_jtj_dev(d_J, nE, n, ld, d_jtj); pl.fill_jtf_dev(d_J, nE, n, ld, d_f, d_jtf)
    jtj = pl.memcpy_d2h(np.empty((n, n)), d_jtj); jtf = pl.memcpy_d2h(np.empty(n), d_jtf)
    assert _close(jtj, J.T @ J) and _close(jtf, J.T @ f)
    assert np.array_equal(jtj, jtj.T)
    # row-scaled in place (J_s = diag(w) J), then both contractions on J_s
    pl.fill_dprobs_dev(d_J, ld, cols, None, 1e-7); pl.fill_jtj_dev(d_J, nE, n, ld, d_jtj, d_w); pl.fill_jtf_dev(d_J, nE, n, ld, d_f, d_jtf)
    Js = J * w[:, None]
    assert _close(pl.memcpy_d2h(np.empty((n, n)), d_jtj), Js.T @ Js)
    assert _close(pl.memcpy_d2h(np.empty(n), d_jtf), Js.T @ f)
    for d in (d_J, d_jtj, d_jtf, d_w, d_f): pl.device_free(d)


def test_simulator_level_normal_equations_multi_atom():
    fx = load_fixture("smq1Q_XYI_L128_depol")
    pack = MP.smq1Q_XYI
    model = _model_from_fixture(fx, pack)
    circuits = pack.create_gst_circuits(128)
    rng = np.random.default_rng(1)
    for natoms, mode in ((1, "fd"), (3, "fd"), (2, "analytic")):
        sim = HipMapForwardSimulator(num_atoms=natoms, derivative_mode=mode); model.sim = sim
        lay = sim.create_layout(circuits)
        w = rng.random(lay.num_elements) + 0.5; f = rng.standard_normal(lay.num_elements)
        J = np.empty((lay.num_elements, 60)); sim.bulk_fill_dprobs(J, lay)
        jtj = np.empty((60, 60)); jtf = np.empty(60); pr = np.empty(lay.num_elements)
        sim.bulk_fill_jtj_jtf(jtj, jtf, lay, row_scale=w, f=f, pr_array_to_fill=pr)
        Js = J * w[:, None]
        assert _close(jtj, Js.T @ Js, 1e-11) and _close(jtf, Js.T @ f, 1e-11)
        p2 = np.empty(lay.num_elements); sim.bulk_fill_probs(p2, lay)
        assert np.array_equal(pr, p2)


@pytest.mark.parametrize("n_rows,n_cols,pad", [(5003, 1616, 3), (3077, 1616, 0), (70001, 200, 1), (4099, 129, 0)])
def test_jtj_benchmark_shape_synthetic(n_rows, n_cols, pad):
    """The shape bench.py --jtj runs (1,616 columns: 13 x 13 column tiles incl. the partial 80-column edge tile and every
    off-diagonal pair, 48 row slabs with a ragged last panel) on a synthetic Jacobian whose row count is not a multiple
    of the 16-row panel; plus a tall narrow case that crosses the 64-slab cap."""
    fx = load_fixture("smq1Q_XYI_L4_depol")
    pl = plan_from_fixture(fx)
    rng = np.random.default_rng(n_rows + n_cols)
    ld = n_cols + pad
    Jp = rng.standard_normal((n_rows, ld)) * np.exp(rng.uniform(-3, 3, size=(1, ld)))
    J = Jp[:, :n_cols]
    w = rng.random(n_rows) + 0.5; f = rng.standard_normal(n_rows)
    d_J = pl.device_malloc(Jp.nbytes); d_jtj = pl.device_malloc(n_cols * n_cols * 8); d_jtf = pl.device_malloc(n_cols * 8)
    d_w = pl.device_malloc(n_rows * 8); d_f = pl.device_malloc(n_rows * 8)
    pl.memcpy_h2d(d_J, Jp); pl.memcpy_h2d(d_w, w); pl.memcpy_h2d(d_f, f)
    pl.fill_jtj_dev(d_J, n_rows, n_cols, ld, d_jtj); pl.fill_jtf_dev(d_J, n_rows, n_cols, ld, d_f, d_jtf)
    jtj = pl.memcpy_d2h(np.empty((n_cols, n_cols)), d_jtj); jtf = pl.memcpy_d2h(np.empty(n_cols), d_jtf)
    want = J.T @ J
    # entry-wise: relative to the norms of the two columns (Cauchy-Schwarz scale of each entry)
    scale = np.sqrt(np.outer(np.diag(want), np.diag(want)))
    assert (np.abs(jtj - want) <= 1e-12 * scale).all(), np.abs((jtj - want) / scale).max()
    assert np.array_equal(jtj, jtj.T)
    assert np.abs(jtf - J.T @ f).max() <= 1e-12 * np.abs(J.T @ f).max()
    # the padding columns of the row-major array must not leak in, and the scaled call must scale rows exactly once
    pl.fill_jtj_dev(d_J, n_rows, n_cols, ld, d_jtj, d_w)
    Js = J * w[:, None]
    want = Js.T @ Js
    scale = np.sqrt(np.outer(np.diag(want), np.diag(want)))
    assert (np.abs(pl.memcpy_d2h(np.empty((n_cols, n_cols)), d_jtj) - want) <= 1e-12 * scale).all()
    for d in (d_J, d_jtj, d_jtf, d_w, d_f): pl.device_free(d)


def test_jtj_on_the_2q_design_jacobian():
    """J^T J / J^T f of the REAL 2Q Jacobian (smq2Q_XYICNOT lite germs L<=64: 55,832 x 1,616, FD, left resident) against
    numpy on the same Jacobian copied to the host -- the bench workload's column count and value distribution."""
    pack = MP.smq2Q_XYICNOT
    model = pack.target_model().depolarize(op_noise=0.01, spam_noise=0.01)
    from pygsti_amd.layout import HipCOPALayout
    lay = HipCOPALayout(pack.create_gst_circuits(64, lite=True), model)
    pl = lay.atoms[0].plan()
    pl.set_model(*lay.model_arrays(model)); pl.set_param_map(*lay.param_map(model))
    nE, nP = lay.num_elements, model.num_params
    d_J = pl.device_malloc(nE * nP * 8); d_jtj = pl.device_malloc(nP * nP * 8); d_jtf = pl.device_malloc(nP * 8); d_f = pl.device_malloc(nE * 8)
    f = np.random.default_rng(5).standard_normal(nE); pl.memcpy_h2d(d_f, f)
    pl.fill_dprobs_dev(d_J, nP, np.arange(nP), None, 1e-7)
    pl.fill_jtj_dev(d_J, nE, nP, nP, d_jtj); pl.fill_jtf_dev(d_J, nE, nP, nP, d_f, d_jtf)
    J = pl.memcpy_d2h(np.empty((nE, nP)), d_J)
    jtj = pl.memcpy_d2h(np.empty((nP, nP)), d_jtj); jtf = pl.memcpy_d2h(np.empty(nP), d_jtf)
    want = J.T @ J
    scale = np.sqrt(np.outer(np.diag(want), np.diag(want))) + 1e-300
    assert (np.abs(jtj - want) <= 1e-12 * scale).all(), np.abs((jtj - want) / scale).max()
    assert np.abs(jtf - J.T @ f).max() <= 1e-12 * np.abs(J.T @ f).max()
    for d in (d_J, d_jtj, d_jtf, d_f): pl.device_free(d)


def test_jtj_block_sparse_jacobian():
    """Round 3: panels (16 rows) whose 128-column tiles hold only zeros are skipped -- exact,
    what is skipped is a product with zeros.  A synthetic Jacobian with the structure of a GST one: groups of rows that are
    zero in whole column blocks (block edges NOT aligned with the 128-column tiles or the 16-row panels), single non-zeros
    in otherwise empty tiles, all-zero rows, zero row weights; with and without the row scale; numpy to 1e-12 (the dense
    form of the kernel, bit-identical on this input in round 3, has been removed)."""
    fx = load_fixture("smq1Q_XYI_L4_depol")
    pl = plan_from_fixture(fx)
    rng = np.random.default_rng(42)
    n_rows, n_cols, ld = 20011, 730, 733
    Jp = rng.standard_normal((n_rows, ld))
    edges = [0, 80, 336, 592, 730]                                   # "SPAM" then three "gates"
    r = 0
    while r < n_rows:
        n = int(rng.integers(3, 90))
        for b in range(1, 4):
            if rng.random() < 0.45:
                Jp[r:r + n, edges[b]:edges[b + 1]] = 0.0
        if rng.random() < 0.05:
            Jp[r:r + n, :] = 0.0
        r += n
    Jp[977, 400] = 3.0; Jp[12001, 729] = -2.0                         # lone non-zeros
    J = Jp[:, :n_cols]
    w = rng.random(n_rows) + 0.5
    w[rng.random(n_rows) < 0.1] = 0.0
    d_J = pl.device_malloc(Jp.nbytes); d_jtj = pl.device_malloc(n_cols * n_cols * 8); d_w = pl.device_malloc(n_rows * 8)
    pl.memcpy_h2d(d_J, Jp); pl.memcpy_h2d(d_w, w)
    pl.fill_jtj_dev(d_J, n_rows, n_cols, ld, d_jtj)
    got = pl.memcpy_d2h(np.empty((n_cols, n_cols)), d_jtj)
    want = J.T @ J
    scale = np.sqrt(np.outer(np.diag(want), np.diag(want))) + 1e-300
    assert (np.abs(got - want) <= 1e-12 * scale).all(), np.abs((got - want) / scale).max()
    assert np.array_equal(got, got.T)
    pl.fill_jtj_dev(d_J, n_rows, n_cols, ld, d_jtj, d_w)             # scales J in place, once
    got_s = pl.memcpy_d2h(np.empty((n_cols, n_cols)), d_jtj)
    Js = J * w[:, None]
    want = Js.T @ Js
    scale = np.sqrt(np.outer(np.diag(want), np.diag(want))) + 1e-300
    assert (np.abs(got_s - want) <= 1e-12 * scale).all()
    back = pl.memcpy_d2h(np.empty((n_rows, ld)), d_J)
    assert np.array_equal(back[:, :n_cols], Js) and np.array_equal(back[:, n_cols:], Jp[:, n_cols:])    # padding untouched
    for d in (d_J, d_jtj, d_w): pl.device_free(d)


@pytest.mark.parametrize("n_rows,n_cols,pad,sparse", [(20011, 730, 3, True), (5003, 1616, 0, False), (70001, 200, 1, False),
                                                      (33, 7, 2, False), (16400, 513, 0, True),
                                                      # n_cols % 8 == 0, even ld: the branch-free two-panels-ahead form with
                                                      # its live-panel bitmaps (windows of 64 panels, some wholly dead)
                                                      (20011, 736, 2, True), (70000, 264, 0, True), (17, 8, 0, False),
                                                      (40003, 1616, 8, True)])
def test_normal_eqs_without_touching_J(n_rows, n_cols, pad, sparse):
    """gst_fill_normal_eqs_dev (round 5): the row weights are applied while the rows are staged, so d_J is only read --
    and since every weighted element is rounded exactly as the in-place scaling stores it, J_s^T J_s and J_s^T f come out
    with the SAME BITS as gst_fill_jtj_dev(..., d_row_scale) + gst_fill_jtf_dev on a copy.  Block-sparse input (zero
    blocks, zero weights, a lone non-zero whose weight is zero: the masks are those of diag(w) J), ragged shapes, padding
    columns, either output alone, no weights at all."""
    fx = load_fixture("smq1Q_XYI_L4_depol")
    pl = plan_from_fixture(fx)
    rng = np.random.default_rng(n_rows * 7 + n_cols)
    ld = n_cols + pad
    Jp = rng.standard_normal((n_rows, ld)) * np.exp(rng.uniform(-3, 3, size=(1, ld)))
    if sparse:
        r = 0
        while r < n_rows:
            n = int(rng.integers(3, 90))
            c0 = int(rng.integers(0, n_cols)); c1 = int(rng.integers(c0, n_cols + 1))
            if rng.random() < 0.6: Jp[r:r + n, c0:c1] = 0.0
            r += n
        if n_rows > 12000:
            Jp[3000:5500, :] = 0.0                                     # 156 dead panels in a row (whole bitmap windows)
            Jp[9000:11000, 128:] = 0.0                                 # ... and a stretch where only the first tile lives
        Jp[960:976, 128:256] = 0.0; Jp[966, 200] = 5.0                # a lone non-zero in its (panel, tile) ...
    w = rng.random(n_rows) + 0.5
    w[rng.random(n_rows) < 0.1] = 0.0
    if sparse: w[966] = 0.0                                            # ... annihilated by its weight
    f = rng.standard_normal(n_rows)
    d_J = pl.device_malloc(Jp.nbytes); d_K = pl.device_malloc(Jp.nbytes)
    d_a = pl.device_malloc(n_cols * n_cols * 8); d_b = pl.device_malloc(n_cols * n_cols * 8)
    d_ya = pl.device_malloc(n_cols * 8); d_yb = pl.device_malloc(n_cols * 8)
    d_w = pl.device_malloc(n_rows * 8); d_f = pl.device_malloc(n_rows * 8)
    pl.memcpy_h2d(d_J, Jp); pl.memcpy_h2d(d_K, Jp); pl.memcpy_h2d(d_w, w); pl.memcpy_h2d(d_f, f)
    # the in-place route on the copy
    pl.fill_jtj_dev(d_K, n_rows, n_cols, ld, d_a, d_w); pl.fill_jtf_dev(d_K, n_rows, n_cols, ld, d_f, d_ya)
    a = pl.memcpy_d2h(np.empty((n_cols, n_cols)), d_a); ya = pl.memcpy_d2h(np.empty(n_cols), d_ya)
    # the new route
    pl.fill_normal_eqs_dev(d_J, n_rows, n_cols, ld, d_w, d_f, d_b, d_yb)
    b = pl.memcpy_d2h(np.empty((n_cols, n_cols)), d_b); yb = pl.memcpy_d2h(np.empty(n_cols), d_yb)
    # J_s^T J_s: the same bits.  J_s^T f: the same bits on the small shapes; on the block-sparse path (>= 16,384 rows, >= 4
    # column tiles) it is summed by the pass that marks the live panels -- one read of J for both, another (fixed) order
    fused = n_rows >= 16384 and n_cols > 384
    assert np.array_equal(a, b)
    assert np.abs(ya - yb).max() <= 1e-13 * np.abs(ya).max() and (fused or np.array_equal(ya, yb))
    assert np.array_equal(pl.memcpy_d2h(np.empty((n_rows, ld)), d_J), Jp)             # J as the caller left it
    Js = Jp[:, :n_cols] * w[:, None]
    want = Js.T @ Js
    scale = np.sqrt(np.outer(np.diag(want), np.diag(want))) + 1e-300
    assert (np.abs(b - want) <= 1e-12 * scale).all() and np.abs(yb - Js.T @ f).max() <= 1e-12 * np.abs(Js.T @ f).max()
    # either output alone; no weights = the plain products
    pl.memcpy_h2d(d_b, np.zeros((n_cols, n_cols))); pl.memcpy_h2d(d_yb, np.zeros(n_cols))
    pl.fill_normal_eqs_dev(d_J, n_rows, n_cols, ld, d_w, d_jtj=d_b)
    assert np.array_equal(pl.memcpy_d2h(np.empty((n_cols, n_cols)), d_b), a) and not pl.memcpy_d2h(np.empty(n_cols), d_yb).any()
    pl.fill_normal_eqs_dev(d_J, n_rows, n_cols, ld, d_w, d_f=d_f, d_jtf=d_yb)
    assert np.array_equal(pl.memcpy_d2h(np.empty(n_cols), d_yb), ya)                  # (J^T f alone: the streaming kernel of gst_fill_jtf_dev)
    pl.fill_normal_eqs_dev(d_J, n_rows, n_cols, ld, d_w, d_f, d_b, d_yb)
    y2 = pl.memcpy_d2h(np.empty(n_cols), d_yb)
    pl.fill_normal_eqs_dev(d_J, n_rows, n_cols, ld, d_w, d_f, d_b, d_yb)
    assert np.array_equal(pl.memcpy_d2h(np.empty(n_cols), d_yb), y2) and np.array_equal(y2, yb)     # deterministic
    pl.fill_jtj_dev(d_J, n_rows, n_cols, ld, d_a); pl.fill_jtf_dev(d_J, n_rows, n_cols, ld, d_f, d_ya)
    pl.fill_normal_eqs_dev(d_J, n_rows, n_cols, ld, None, d_f, d_b, d_yb)
    assert np.array_equal(pl.memcpy_d2h(np.empty((n_cols, n_cols)), d_a), pl.memcpy_d2h(np.empty((n_cols, n_cols)), d_b))
    ya0, yb0 = pl.memcpy_d2h(np.empty(n_cols), d_ya), pl.memcpy_d2h(np.empty(n_cols), d_yb)
    assert np.abs(ya0 - yb0).max() <= 1e-13 * np.abs(ya0).max() and (fused or np.array_equal(ya0, yb0))
    with pytest.raises(ValueError):
        pl.fill_normal_eqs_dev(d_J, n_rows, n_cols, ld, d_w)                           # nothing to compute
    with pytest.raises(ValueError):
        pl.fill_normal_eqs_dev(d_J, n_rows, n_cols, ld, d_w, d_jtf=d_yb)               # J^T f without f
    for d in (d_J, d_K, d_a, d_b, d_ya, d_yb, d_w, d_f): pl.device_free(d)


def test_normal_eqs_keep_resident_zeros():
    """An exact Jacobian in tracked memory keeps its structural zeros resident through the weighted products of
    gst_fill_normal_eqs_dev -- which never write to it, whatever the weights are (an infinite weight ends the claim of the
    in-place route; here there is nothing to end) -- so the next exact fill into the same block is still bit-identical
    to a fill into fresh memory."""
    fx = load_fixture("smq2Q_XYICNOT_L2_depol")
    pl = plan_from_fixture(fx)
    nE, nP = int(fx["nE"]), int(fx["nP"])
    idx = np.arange(nP)
    d = pl.device_malloc(nE * nP * 8, tracked=True); d2 = pl.device_malloc(nE * nP * 8)
    d_w = pl.device_malloc(nE * 8); d_jtj = pl.device_malloc(nP * nP * 8)
    w = np.random.default_rng(3).random(nE) + 0.5; w[5] = np.inf
    pl.memcpy_h2d(d_w, w)

    def fill(dst):
        pl.fill_dprobs_dev(dst, nP, idx, None, 1e-7, None, _lib.DERIV_ANALYTIC); pl.sync()
        return bool(pl.stats()["last_zeros_resident"])
    assert not fill(d)
    pl.fill_normal_eqs_dev(d, nE, nP, nP, d_w, d_jtj=d_jtj)
    assert fill(d)                                                     # repeated fill: the zeros are not stored again
    pl.memcpy_h2d(d2, np.full((nE, nP), np.nan))
    assert not fill(d2)
    a = pl.memcpy_d2h(np.empty((nE, nP)), d); b = pl.memcpy_d2h(np.empty((nE, nP)), d2)
    assert np.array_equal(a, b) and (a == 0).mean() > 0.1
    for x in (d, d2, d_w, d_jtj): pl.device_free(x)
