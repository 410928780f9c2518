"""GPU tests of the reference-shaped simulator surface (HipMapForwardSimulator + HipCOPALayout):
bulk_fill_probs / bulk_fill_dprobs / bulk_fill_hprobs, parameter blocking, multi-atom layouts,
rectangle iterator, device-resident fills -- all checked against the reference's golden vectors or
the CPU oracle, bit for bit."""
import numpy as np
import pytest

from conftest import load_fixture, assert_bitwise
from pygsti_amd import modelpacks as MP
from pygsti_amd.forwardsim import HipMapForwardSimulator
from test_host_mirror import _model_from_fixture

pytestmark = pytest.mark.gpu


def _setup(name, pack, L, **simkw):
    fx = load_fixture(name)
    model = _model_from_fixture(fx, pack)
    sim = HipMapForwardSimulator(**simkw)
    model.sim = sim
    circuits = pack.create_gst_circuits(L)
    layout = sim.create_layout(circuits, array_types=("e", "ep", "epp"))
    return fx, model, sim, circuits, layout


def _by_circuit(arr, layout, circuits):
    """Re-order an element array of `layout` into the reference's (1-atom) element order."""
    return np.concatenate([arr[layout.indices_for_index(i)] for i in range(len(circuits))])


@pytest.mark.parametrize("natoms", [1, 3])
def test_bulk_fill_probs_and_dprobs_1q(natoms):
    fx, model, sim, circuits, layout = _setup("smq1Q_XYI_L4_depol", MP.smq1Q_XYI, 4, num_atoms=natoms)
    probs = layout.allocate_local_array("e", "d")
    sim.bulk_fill_probs(probs, layout)
    assert_bitwise(_by_circuit(probs, layout, circuits), fx["probs"], "bulk_fill_probs")
    J = layout.allocate_local_array("ep", "d")
    pr = layout.allocate_local_array("e", "d")
    sim.bulk_fill_dprobs(J, layout, pr_array_to_fill=pr)
    assert_bitwise(_by_circuit(pr, layout, circuits), fx["probs"], "pr_array_to_fill")
    assert_bitwise(_by_circuit(J, layout, circuits), fx["dprobs_map"], "bulk_fill_dprobs")


def test_parameter_blocking_gives_identical_jacobian():
    fx, model, sim, circuits, layout = _setup("smq1Q_XYI_L128_depol", MP.smq1Q_XYI, 128, param_blk_sizes=(17,))
    assert layout.param_dimension_blk_sizes[0] == 17
    J = layout.allocate_local_array("ep", "d")
    sim.bulk_fill_dprobs(J, layout)
    assert_bitwise(J, fx["dprobs_map"], "blocked dprobs")


def test_bulk_fill_hprobs_and_rectangles():
    fx, model, sim, circuits, layout = _setup("smq1Q_XYI_L4_depol", MP.smq1Q_XYI, 4)
    rows, cols = fx["hprobs_rows"], fx["hprobs_cols"]
    got = {}
    for s1, s2, h in sim.iter_hprobs_by_rectangle(layout, [(rows, cols)]):
        got["h"] = h
    assert_bitwise(got["h"], fx["hprobs_map"], "rectangle")
    # dprobs12 by-product: outer product of the two FD Jacobian blocks (distforwardsim.py:332-334)
    for s1, s2, h, d12 in sim.iter_hprobs_by_rectangle(layout, [(slice(0, 4), slice(4, 9))], True):
        assert_bitwise(d12, fx["dprobs_map"][:, 0:4, None] * fx["dprobs_map"][:, None, 4:9], "dprobs12")
    # small full bulk_fill_hprobs through blocks == one-shot
    sub = circuits[:12]
    lay = sim.create_layout(sub)
    H1 = np.empty((lay.num_elements, 60, 60)); sim.bulk_fill_hprobs(H1, lay)
    sim2 = HipMapForwardSimulator(param_blk_sizes=(25, 16)); model.sim = sim2
    lay2 = sim2.create_layout(sub)
    H2 = np.empty((lay2.num_elements, 60, 60)); sim2.bulk_fill_hprobs(H2, lay2)
    assert_bitwise(H1, H2, "blocked hessian")


def test_2q_multi_atom_column_subset(oracle_built):
    fx, model, sim, circuits, layout = _setup("smq2Q_XYICNOT_L2_depol", MP.smq2Q_XYICNOT, 2, num_atoms=4)
    cols = fx["dprobs_cols"]
    J = np.empty((layout.num_elements, len(cols)))
    for atom in layout.atoms:
        sim._bulk_fill_dprobs_atom(J[atom.element_slice], None, atom, cols)
    assert_bitwise(_by_circuit(J, layout, circuits), fx["dprobs_map"], "2Q multi-atom dprobs")
    d = sim.bulk_probs(circuits[:5])
    assert set(d[circuits[3]].keys()) == {("00",), ("01",), ("10",), ("11",)}
    assert abs(sum(d[circuits[3]].values()) - 1.0) < 1e-12


def test_device_resident_fill_matches_host_fill():
    fx = load_fixture("smq2Q_XYICNOT_L2_depol")
    from conftest import plan_from_fixture
    pl = plan_from_fixture(fx)
    nE, cols = int(fx["nE"]), fx["dprobs_cols"]
    ld = len(cols) + 3
    d_out = pl.device_malloc(nE * ld * 8)
    d_pr = pl.device_malloc(nE * 8)
    pl.fill_dprobs_dev(d_out, ld, cols, np.arange(len(cols)) + 3, 1e-7, d_pr)
    pl.sync()
    out = np.empty((nE, ld)); pr = np.empty(nE)
    pl.memcpy_d2h(out, d_out); pl.memcpy_d2h(pr, d_pr)
    assert_bitwise(out[:, 3:], fx["dprobs_map"], "device-resident dprobs")
    assert_bitwise(pr, fx["probs"], "device-resident probs")
    pl.fill_probs_dev(d_pr); pl.sync()
    assert_bitwise(pl.memcpy_d2h(np.empty(nE), d_pr), fx["probs"], "fill_probs_dev")
    pl.device_free(d_out); pl.device_free(d_pr)


def test_full_size_properties(oracle_built):
    """BASELINE-size run (2Q L<=1024 lite, 24,394 circuits): EVERY probability and 17 whole finite-difference columns
    (preparation, effects, all six gates) equal the CPU checker -- the reference's own C++ reps walking the
    reference-format prefix table of the same design -- bit for bit; results are independent of the task granularity;
    probabilities of each circuit sum to 1 (trace preservation of the depolarized target)."""
    from conftest import design_checker
    import bench
    pack = MP.smq2Q_XYICNOT
    model = pack.target_model().depolarize(0.01, 0.01)
    circuits = pack.create_gst_circuits(1024, lite=True)
    cols = bench.parity_columns(model.dim, 6, 4, model.num_params)
    sims = [HipMapForwardSimulator(target_tasks=t) for t in (0, 97)]
    outs = []
    for sim in sims:
        model.sim = sim
        lay = sim.create_layout(circuits)
        p = np.empty(lay.num_elements); sim.bulk_fill_probs(p, lay)
        J = np.empty((lay.num_elements, len(cols)))
        sim._bulk_fill_dprobs_atom(J, None, lay.atoms[0], cols)
        outs.append((p, J))
    assert_bitwise(outs[0][0], outs[1][0], "probs vs task granularity")
    assert_bitwise(outs[0][1], outs[1][1], "dprobs vs task granularity")
    orc = design_checker(oracle_built, pack, model, circuits, lay)
    Jo, po = orc.dprobs(cols, eps=1e-7, return_probs=True)
    assert_bitwise(outs[0][0], po, "all %d probabilities of the lite design vs the %s checker" % (len(po), orc.kind))
    assert_bitwise(outs[0][1], Jo, "%d whole FD columns of the lite design vs the %s checker" % (len(cols), orc.kind))
    p = outs[0][0].reshape(-1, 4)
    assert np.abs(p.sum(axis=1) - 1.0).max() < 1e-12 and p.min() > -1e-12


def test_baseline_size_checksums(oracle_built):
    """The bench workload itself (2Q L<=1024 FULL design: 136,275 circuits x 1,616 parameters, a 7 GB Jacobian that
    stays in HBM): ALL 545,100 probabilities and 17 whole columns of the resident Jacobian equal the CPU checker (the
    reference's own C++ reps on the reference-format prefix table of the design) bit for bit -- the comparison bench.py
    repeats in its `parity` object -- and, for the other 1,599 columns, size-independent properties: J^T f -- a checksum of
    every Jacobian element, reduced on the device in a fixed order -- is BITWISE the same for two different task
    decompositions (any prefix-sharing schedule gives the same states); it is linear in f; and the
    analytic Jacobian's checksum agrees with the finite-difference one to the FD truncation error."""
    from pygsti_amd import _lib
    from pygsti_amd.layout import HipCOPALayout
    pack = MP.smq2Q_XYICNOT
    model = pack.target_model().depolarize(0.01, 0.01)
    circuits = pack.create_gst_circuits(1024, lite=False)
    assert len(circuits) == 136275
    nP = model.num_params
    pidx = np.arange(nP, dtype=np.int64)
    rng = np.random.default_rng(42)
    sums = {}
    for tt in (0, 777):
        layout = HipCOPALayout(circuits, model, num_atoms=1, devices=[0], rank=0, size=1, target_tasks=tt)
        plan = layout.atoms[0].plan()
        plan.set_model(*layout.model_arrays(model)); plan.set_param_map(*layout.param_map(model))
        nE = layout.num_elements
        assert nE == 545100
        f1 = rng.standard_normal(nE) if tt == 0 else f1
        f2 = rng.standard_normal(nE) if tt == 0 else f2
        bufs = [plan.device_malloc(n) for n in (nE * nP * 8, nE * 8, nE * 8, nP * 8)]
        d_J, d_p, d_f, d_y = bufs
        try:
            def jtf(f):
                plan.memcpy_h2d(d_f, f)
                plan.fill_jtf_dev(d_J, nE, nP, nP, d_f, d_y)
                return plan.memcpy_d2h(np.empty(nP), d_y)
            plan.fill_dprobs_dev(d_J, nP, pidx, None, 1e-7, d_p, _lib.DERIV_FD)
            probs = plan.memcpy_d2h(np.empty(nE), d_p)
            y1, y2, y12 = jtf(f1), jtf(f2), jtf(f1 + f2)
            sums[tt] = (probs, y1)
            if tt == 0:
                assert np.abs(probs.reshape(-1, 4).sum(axis=1) - 1.0).max() < 1e-12 and probs.min() > -1e-12
                import bench
                from conftest import design_checker
                cols = bench.parity_columns(model.dim, 6, 4, nP)
                orc = design_checker(oracle_built, pack, model, circuits, layout)
                Jo, po = orc.dprobs(cols, eps=1e-7, return_probs=True)
                assert_bitwise(probs, po, "all 545,100 probabilities of the bench workload vs the %s checker" % orc.kind)
                d_col = plan.device_malloc(nE * 8)
                for j, c in enumerate(cols):
                    plan.copy_block_dev(d_col, 1, d_J + int(c) * 8, nP, nE, 1)
                    assert_bitwise(plan.memcpy_d2h(np.empty(nE), d_col), Jo[:, j], "column %d of the resident Jacobian vs the checker" % c)
                plan.device_free(d_col)
                assert np.abs(y12 - (y1 + y2)).max() <= 1e-9 * np.abs(y1).max()          # linearity of the checksum
                plan.fill_dprobs_dev(d_J, nP, pidx, None, 1e-7, d_p, _lib.DERIV_ANALYTIC)
                ya = jtf(f1)
                assert np.abs(ya - y1).max() <= 2e-4 * np.abs(ya).max()                   # FD truncation (eps = 1e-7, depth 1030)
                assert np.abs(ya - y1).max() > 0
        finally:
            for b in bufs:
                plan.device_free(b)
        plan.close()
    assert_bitwise(sums[0][0], sums[777][0], "probabilities vs task decomposition at full size")
    assert_bitwise(sums[0][1], sums[777][1], "J^T f checksum vs task decomposition at full size")


def test_dataset_restricted_layout_on_device():
    """bulk_fill_probs / bulk_fill_dprobs on a layout that holds only the outcomes a data set observed (ragged effect
    CSR on the device): rows equal the corresponding rows of the reference's all-outcome vectors, bit for bit."""
    from test_host_mirror import _sparse_dataset
    fx, model, sim, circuits, full = _setup("smq2Q_XYICNOT_L2_depol", MP.smq2Q_XYICNOT, 2, num_atoms=3)
    ds = _sparse_dataset(circuits)
    lay = sim.create_layout(circuits, dataset=ds)
    assert lay.num_elements < full.num_elements
    p = lay.allocate_local_array("e", "d"); sim.bulk_fill_probs(p, lay)
    cols = fx["dprobs_cols"]
    J = np.empty((lay.num_elements, len(cols)))
    for atom in lay.atoms:
        sim._bulk_fill_dprobs_atom(J[atom.element_slice], None, atom, cols)
    names = ["00", "01", "10", "11"]
    rows = np.concatenate([[4 * i + names.index(o[0]) for o in lay.outcomes_for_index(i)] for i in range(len(circuits))])
    order = np.concatenate([np.arange(*lay.indices_for_index(i).indices(lay.num_elements)) for i in range(len(circuits))])
    assert_bitwise(p[order], fx["probs"][rows], "dataset-restricted probs")
    assert_bitwise(J[order], fx["dprobs_map"][rows], "dataset-restricted dprobs")


def test_two_phase_fills_over_several_atoms_bitwise():
    """The single-process multi-GPU form of bulk_fill_probs / bulk_fill_dprobs (all atoms enqueued on their own streams,
    then collected) -- forced here on one device -- gives the same bits as the atom-by-atom loop."""
    fx, model, sim, circuits, layout = _setup("smq1Q_XYI_L128_depol", MP.smq1Q_XYI, 128, num_atoms=4)
    sim.concurrent_fills = True
    p = layout.allocate_local_array("e", "d"); sim.bulk_fill_probs(p, layout)
    J = layout.allocate_local_array("ep", "d"); pr = layout.allocate_local_array("e", "d")
    sim.bulk_fill_dprobs(J, layout, pr_array_to_fill=pr)
    assert_bitwise(_by_circuit(p, layout, circuits), fx["probs"], "two-phase probs")
    assert_bitwise(_by_circuit(pr, layout, circuits), fx["probs"], "two-phase pr_array_to_fill")
    assert_bitwise(_by_circuit(J, layout, circuits), fx["dprobs_map"], "two-phase dprobs")


def test_several_preparations_and_povms_on_device():
    """Two preparations, two POVMs (2 and 3 effects) through the host mirror's simulator: probabilities, FD Jacobian and an
    FD-of-FD Hessian block bit-identical to the reference's vectors (tests/golden/smq1Q_multispam_L2.npz), whatever the
    number of atoms."""
    from test_host_mirror import _multispam_case
    fx, model, circuits, ref = _multispam_case()
    for natoms in (1, 2):
        sim = HipMapForwardSimulator(model, num_atoms=natoms)
        lay = sim.create_layout(circuits)
        nE, nP = lay.num_elements, model.num_params
        mine = np.empty(nE, np.int64)                      # my element -> the reference's element
        for i in range(len(circuits)):
            sl = lay.indices_for_index(i)
            mine[sl] = [ref[(i, o[0])] for o in lay.outcomes_for_index(i)]
        p = np.empty(nE); sim.bulk_fill_probs(p, lay)
        assert_bitwise(p, fx["probs"][mine], "probs, %d atoms" % natoms)
        J = np.empty((nE, nP)); sim.bulk_fill_dprobs(J, lay)
        assert_bitwise(J[:, fx["dprobs_cols"]], fx["dprobs_map"][mine], "dprobs, %d atoms" % natoms)
        if natoms == 1:
            r, c = fx["hprobs_rows"], fx["hprobs_cols"]
            H = np.empty((nE, len(r), len(c)))
            sim._bulk_fill_hprobs_atom(H, None, None, lay.atoms[0], r, c)
            assert_bitwise(H, fx["hprobs_map"][mine], "hprobs block")
