"""Adapter for the real pyGSTi (runs only where the reference is importable, e.g. the build container with
PYTHONPATH=/tmp/pgref; skipped on the GPU box).  Without a GPU we check everything up to the launch: the
subclass plugs into `model.sim`, pyGSTi's own MapCOPALayout is created, the atom's prefix table becomes a
libgstfwd plan whose programs -- interpreted in numpy -- reproduce pyGSTi's bulk_fill_probs bit for bit,
the parameter map matches pyGSTi's gpindices, and a fill raises GstDeviceError (no silent CPU fallback)."""
import numpy as np
import pytest

pygsti = pytest.importorskip("pygsti")

from _interp import run_programs                      # noqa: E402
from conftest import assert_bitwise                    # noqa: E402
from pygsti_amd import _lib                            # noqa: E402
from pygsti_amd import pygsti_adapter as A             # noqa: E402


def test_adapter_plan_matches_pygsti():
    from pygsti.modelpacks import smq1Q_XYI
    from pygsti.forwardsims import MapForwardSimulator
    model = smq1Q_XYI.target_model().depolarize(op_noise=0.01, spam_noise=0.01)
    circuits = list(smq1Q_XYI.create_gst_experiment_design(8).all_circuits_needing_data)
    ref = model.copy(); ref.sim = MapForwardSimulator(num_atoms=2)
    lay_ref = ref.sim.create_layout(circuits, array_types=("e", "ep"))
    p_ref = np.empty(lay_ref.num_elements); ref.sim.bulk_fill_probs(p_ref, lay_ref)

    model.sim = A.HipMapForwardSimulator(num_atoms=2)
    assert isinstance(model.sim, MapForwardSimulator) and model.sim.model is model
    layout = model.sim.create_layout(circuits, array_types=("e", "ep"))      # pyGSTi's own MapCOPALayout
    assert len(layout.atoms) == 2
    # the adapter skips pyGSTi's prefix-cache pass (the library builds its own trie): trivial table, same results
    assert all(at.cache_size == 0 and all(row[1] is None for row in at.table.contents) for at in layout.atoms)
    assert all(at.cache_size > 0 for at in lay_ref.atoms)
    out = np.empty(layout.num_elements)
    for atom in layout.atoms:
        plan = A.atom_plan(model, atom)
        G, R, E = A.atom_arrays(model, atom)
        kind, obj, elem = A.atom_param_map(model, atom)
        v = model.to_vector()
        for p in range(model.num_params):
            if kind[p] >= 0:
                assert (G, R, E)[kind[p]][obj[p]].ravel()[elem[p]] == v[p]
        w, off = plan.program()
        n = len(atom.elbl_indices_by_expcircuit)
        eff_ptr = np.zeros(n + 1, np.int64); el, ed = [], []
        for i in range(n):
            el.extend(atom.elbl_indices_by_expcircuit[i]); ed.extend(atom.elindices_by_expcircuit[i]); eff_ptr[i + 1] = len(el)
        o, written, _ = run_programs(w, off, G, R, E, eff_ptr, np.array(el), np.array(ed), atom.num_elements)
        assert (written == 1).all()
        out[atom.element_slice] = o
    assert_bitwise(out, p_ref, "adapter plan vs pyGSTi bulk_fill_probs")
    if _lib.device_count() == 0:
        with pytest.raises(_lib.GstDeviceError):
            model.sim.bulk_fill_probs(np.empty(layout.num_elements), layout)


def test_adapter_non_full_parameterisations():
    """No element map for TP / CPTP members (loud), but their deriv_wrt_params feed gst_set_derivs: the arrays the
    adapter extracts are the ones the committed fixture was generated with."""
    from pygsti.modelpacks import smq1Q_XYI
    model = smq1Q_XYI.target_model("CPTPLND")
    model.sim = A.HipMapForwardSimulator(derivative_mode="analytic")
    layout = model.sim.create_layout(list(smq1Q_XYI.create_gst_experiment_design(1).all_circuits_needing_data))
    atom = layout.atoms[0]
    A.atom_plan(model, atom); A.atom_arrays(model, atom)
    with pytest.raises(NotImplementedError):
        A.atom_param_map(model, atom)
    objs = A.atom_derivs(model, atom)
    D = model.dim
    assert len(objs) == len(atom.op_labels) + len(atom.rho_labels) + len(atom._hip_eff_labels)
    for k, oi, idx, dm in objs:
        assert dm.shape == ((D * D if k == 0 else D), len(idx)) and idx.max() < model.num_params
    # same extraction as the fixture generator's (tests/golden/make_golden.py 'tp'), on the TP model of the fixture
    from conftest import load_fixture
    from oracle import oracle as O
    fx = load_fixture("smq1Q_XYI_L4_TP")
    mtp = smq1Q_XYI.target_model("full TP").depolarize(op_noise=0.01, spam_noise=0.01)
    mtp.sim = A.HipMapForwardSimulator(derivative_mode="analytic", num_atoms=1)
    lay = mtp.sim.create_layout(list(smq1Q_XYI.create_gst_experiment_design(4).all_circuits_needing_data))
    A.atom_plan(mtp, lay.atoms[0]); A.atom_arrays(mtp, lay.atoms[0])
    atom_tp = lay.atoms[0]
    mine = A.atom_derivs(mtp, atom_tp)
    ref = O.derivs_from_fixture(fx)
    assert len(mine) == len(ref)
    # (effect indices follow the iteration order of a set, which differs between processes: match by label)
    names_mine = {0: [str(l) for l in atom_tp.op_labels], 1: [str(l) for l in atom_tp.rho_labels], 2: [str(l) for l in atom_tp._hip_eff_labels]}
    names_ref = {0: list(fx["op_labels"]), 1: list(fx["rho_labels"]), 2: list(fx["eff_labels"])}
    by_label = {(k, names_ref[k][o]): (i, d) for k, o, i, d in ref}
    for k, o, i, d in mine:
        i2, d2 = by_label[(k, names_mine[k][o])]
        assert np.array_equal(i, i2) and np.array_equal(d, d2)
    # FD mode over the TP model: element-subset map + the complement effect, as the fixture records them
    k_m, o_m, e_m, comp = A.atom_tp_map(mtp, atom_tp)
    k_r, o_r, e_r = O.tp_param_map(fx)
    assert np.array_equal(k_m, k_r) and np.array_equal(e_m, e_r)
    assert [names_mine[k][o] for k, o in zip(k_m, o_m)] == [names_ref[k][o] for k, o in zip(k_r, o_r)]
    assert names_mine[2][comp[0]] == names_ref[2][int(fx["comp_index"])]
    assert [names_mine[2][o] for o in comp[2]] == [names_ref[2][o] for o in fx["comp_others"]]
    assert np.array_equal(comp[1], fx["comp_identity"])
    with pytest.raises(NotImplementedError):      # CPTPLND members are not element subsets: FD stays refused, loudly
        A.atom_tp_map(model, atom)
    if _lib.device_count() == 0:      # no device here: the TP FD request reaches the library and fails there, loudly
        mtp.sim = A.HipMapForwardSimulator(derivative_mode="fd", num_atoms=1)
        lay = mtp.sim.create_layout(list(smq1Q_XYI.create_gst_experiment_design(1).all_circuits_needing_data), array_types=("ep",))
        with pytest.raises(_lib.GstDeviceError):
            mtp.sim.bulk_fill_dprobs(np.empty((lay.num_elements, mtp.num_params)), lay)
        model.sim = A.HipMapForwardSimulator(derivative_mode="fd", num_atoms=1)
        lay = model.sim.create_layout(list(smq1Q_XYI.create_gst_experiment_design(1).all_circuits_needing_data), array_types=("ep",))
        with pytest.raises(_lib.GstDeviceError):       # CPTPLND FD: the general (model-set) path, refused only for lack of a device
            model.sim.bulk_fill_dprobs(np.empty((lay.num_elements, model.num_params)), lay)


def test_adapter_general_parameterisation_fd_model_sets():
    """Default-constructed simulator on a CPTPLND model (the default protocol's parameterisation): finite differences go
    through the model sets -- the adapter's host stepping reproduces, array for array, the dense models the reference
    wrote into the committed fixture (tests/golden/make_golden.py 'tp'), and the fill reaches the device."""
    from pygsti.modelpacks import smq1Q_XYI
    from conftest import load_fixture
    fx = load_fixture("smq1Q_XYI_L4_CPTPLND")
    m = smq1Q_XYI.target_model("CPTPLND")
    m.from_vector(m.to_vector() + 0.01 * np.random.default_rng(3).standard_normal(m.num_params))
    assert np.array_equal(m.to_vector(), fx["paramvec"])
    m.sim = A.HipMapForwardSimulator()                       # defaults: derivative_mode="auto" (exact derivatives for Lindblad members)
    lay = m.sim.create_layout(list(smq1Q_XYI.create_gst_experiment_design(4).all_circuits_needing_data), array_types=("ep",))
    atom = lay.atoms[0]
    A.atom_plan(m, atom)
    v0 = m.to_vector().copy()
    G, R, E = A.atom_model_sets(m, atom, np.arange(m.num_params), float(fx["derivative_eps"]))
    assert np.array_equal(m.to_vector(), v0)
    eff_mine = [str(l) for l in atom._hip_eff_labels]; eff_ref = list(fx["eff_labels"])
    perm = [eff_mine.index(l) for l in eff_ref]              # (set iteration order differs between processes)
    assert [str(l) for l in atom.op_labels] == list(fx["op_labels"])
    assert np.array_equal(G, fx["mm_gates"]) and np.array_equal(R, fx["mm_rhos"]) and np.array_equal(E[:, perm], fx["mm_effects"])
    # default: the device builds the Lindblad members itself (mode "lindblad": its first step needs the device) ...
    if _lib.device_count() == 0:
        with pytest.raises(_lib.GstDeviceError):
            m.sim._prepare(atom, derivatives=True)
        assert atom._hip_plan._hip_mode == "lindblad"
    # ... and the host-stepped validation path on request (finite differences forced: `auto` would take exact derivatives)
    m.sim.lindblad_on_device = False
    m.sim.derivative_mode = "fd"
    plan = m.sim._prepare(atom, derivatives=True)
    assert plan._hip_mode == "models"
    if _lib.device_count() == 0:
        with pytest.raises(_lib.GstDeviceError):
            m.sim.bulk_fill_dprobs(np.empty((lay.num_elements, m.num_params)), lay)
    # FD-of-FD Hessian blocks of the same model: composed from model-set Jacobians (no NotImplementedError any more);
    # the two-level stepping leaves the model where it was
    if _lib.device_count() == 0:
        H = np.empty((lay.num_elements, 2, 3))
        with pytest.raises(_lib.GstDeviceError):
            m.sim._bulk_fill_hprobs_atom(H, None, None, atom, np.array([0, 2]), np.array([0, 3, 6]), None)
        assert np.array_equal(m.to_vector(), v0)


@pytest.mark.parametrize("param,pack_name", [("CPTPLND", "smq1Q_XYI"), ("GLND", "smq1Q_XYI"), ("H+S", "smq1Q_XYI"), ("CPTPLND", "smq2Q_XYICNOT")])
def test_adapter_describes_lindblad_models_for_the_device(param, pack_name):
    """`atom_lindblad`: the members of a CPTPLND / GLND / H+S model as (static factor, coefficient blocks, the reference's
    own term superoperators) -- what gst_set_lindblad takes.  The host restatement of the device's builder on that
    description reproduces pyGSTi's dense members, and the dense model after every set_parameter_value step
    (`atom_model_sets`, the path this description replaces), to 1e-14; a `full` model is refused (it has its own, exact
    path); the native construction from Pauli matrices gives the same term superoperators."""
    import importlib
    from pygsti_amd import lindblad as LBM
    pack = importlib.import_module("pygsti.modelpacks." + pack_name)
    model = pack.target_model(param)
    rng = np.random.default_rng(2)
    model.from_vector(model.to_vector() + 0.01 * rng.standard_normal(model.num_params))
    model.sim = A.HipMapForwardSimulator()
    circuits = list(pack.create_gst_experiment_design(1, lite=True).all_circuits_needing_data) if pack_name.startswith("smq2Q") \
        else list(pack.create_gst_experiment_design(2).all_circuits_needing_data)
    layout = model.sim.create_layout(circuits, array_types=("e", "ep"))
    atom = layout.atoms[0]
    A.atom_plan(model, atom)
    lm = A.atom_lindblad(model, atom)
    assert lm.num_params == model.num_params and len(lm.members) == len(atom.op_labels) + 2
    th = model.to_vector()
    G, R, E = lm.dense(th)
    G0, R0, E0 = A.atom_arrays(model, atom)
    assert np.abs(G - G0).max() < 1e-14 and np.abs(R - R0).max() < 1e-14 and np.abs(E - E0).max() < 1e-14
    cols = np.sort(rng.choice(model.num_params, 12, replace=False))
    Gs, Rs, Es = lm.model_sets(th, cols, 1e-7)
    Gr, Rr, Er = A.atom_model_sets(model, atom, cols, 1e-7)
    assert np.abs(Gs - Gr).max() < 1e-14 and np.abs(Rs - Rr).max() < 1e-14 and np.abs(Es - Er).max() < 1e-14
    assert np.array_equal(model.to_vector(), th)                      # (the model is left where it was)
    # the same members built from Pauli matrices alone
    nq = 1 if model.dim == 4 else 2
    for m in lm.members:
        native = LBM.LindbladMember(m.kind, m.obj, m.param0, m.blocks, m.static, nq)
        assert np.abs(native.term_re - m.term_re).max() < 1e-14 and np.abs(native.term_im - m.term_im).max() < 1e-14
    # the plan takes the description (host side of gst_set_lindblad: validation, term sharing)
    plan = atom._hip_plan
    plan.set_lindblad(lm)
    if _lib.device_count() == 0:
        with pytest.raises(_lib.GstDeviceError):
            plan.set_lindblad_params(th)
    full = pack.target_model()
    full.sim = A.HipMapForwardSimulator()
    lay2 = full.sim.create_layout(circuits[:5], array_types=("e",))
    A.atom_plan(full, lay2.atoms[0])
    with pytest.raises(NotImplementedError):
        A.atom_lindblad(full, lay2.atoms[0])


def test_atom_plan_from_raw_circuits_is_cheap_and_equivalent():
    """SURVEY 8(f) row f2 in the drop-in: the plan is built from the layout's raw circuits with C-level label conversion
    (gst_plan_create_from_circuits), not by a Python loop over every gate of every row.  Same circuits, same element
    CSR as the table path; a fraction of pyGSTi's own layout-creation time (measured on the 2Q L<=1024 lite design in the
    build container: create_layout 6.3 s, atom_plan 0.97 s including the library's plan compile)."""
    import time
    from pygsti.modelpacks import smq2Q_XYICNOT
    model = smq2Q_XYICNOT.target_model().depolarize(op_noise=0.01, spam_noise=0.01)
    circuits = list(smq2Q_XYICNOT.create_gst_experiment_design(16, lite=True).all_circuits_needing_data)
    model.sim = A.HipMapForwardSimulator()
    t0 = time.perf_counter()
    layout = model.sim.create_layout(circuits, array_types=("e", "ep"))
    t_layout = time.perf_counter() - t0
    atom = layout.atoms[0]
    t0 = time.perf_counter()
    plan = A.atom_plan(model, atom)
    t_plan = time.perf_counter() - t0
    assert t_plan < 0.5 * t_layout + 0.2, (t_plan, t_layout)
    st = plan.stats()
    assert st["n_circuits"] == len(atom.table.contents) and st["n_elements"] == atom.num_elements
    assert st["sum_depth"] == sum(len(row[2]) - 1 for row in atom.table.contents)
    # the same atom through the reference-format table path gives the same plan
    op_lookup = {l: i for i, l in enumerate(atom.op_labels)}; rho_lookup = {l: i for i, l in enumerate(atom.rho_labels)}
    R = len(atom.table.contents)
    t_dest = np.array([r[0] for r in atom.table.contents], np.int32)
    row_ptr = np.zeros(R + 1, np.int64); gidx = []; t_rho = np.empty(R, np.int32)
    for k, row in enumerate(atom.table.contents):
        t_rho[k] = rho_lookup[row[2][0]]; gidx.extend(op_lookup[g] for g in row[2][1:]); row_ptr[k + 1] = len(gidx)
    eff_ptr = np.zeros(R + 1, np.int64); el, ed = [], []
    for i in range(R):
        el.extend(atom.elbl_indices_by_expcircuit[i]); ed.extend(atom.elindices_by_expcircuit[i]); eff_ptr[i + 1] = len(el)
    ref = _lib.Plan.from_table(model.dim, len(atom.op_labels), len(atom.rho_labels), len(atom._hip_eff_labels), atom.num_elements, 0,
                               t_dest, -np.ones(R, np.int32), -np.ones(R, np.int32), t_rho, row_ptr, np.array(gidx, np.int32),
                               eff_ptr, np.array(el, np.int32), np.array(ed, np.int32))
    w1, o1 = plan.program(); w2, o2 = ref.program()
    assert np.array_equal(w1, w2) and np.array_equal(o1, o2)


def test_adapter_takes_exact_derivatives_for_implicit_models():
    """An implicit model of the reference (`create_crosstalk_free_model`: EmbeddedOp / ComposedOp layer operations sharing
    their factors' parameters) under the default simulator: no element map, no TP map, no Lindblad description -- the
    `auto` derivative mode resolves to the chain rule over the layers' deriv_wrt_params ("derivs" mode: no host
    re-densification per column); the matrices handed to gst_set_derivs are the reference's own, and the dense layer
    operations are the ones the committed fixture holds (tests/golden/3q_crosstalk_free.npz)."""
    from pygsti.processors import QubitProcessorSpec
    from pygsti.models import modelconstruction as mc
    from pygsti.circuits import Circuit
    from conftest import load_fixture
    fx = load_fixture("3q_crosstalk_free")
    ps = QubitProcessorSpec(3, ['Gxpi2', 'Gypi2', 'Gcnot'], geometry='line')
    m = mc.create_crosstalk_free_model(ps, ideal_gate_type='full', ideal_spam_type='full')
    m.from_vector(m.to_vector() + 0.02 * np.random.default_rng(77).standard_normal(m.num_params))
    assert np.array_equal(m.to_vector(), fx["paramvec"])
    m.sim = A.HipMapForwardSimulator()
    circs = [Circuit([[('Gxpi2', 0), ('Gcnot', 1, 2)], ('Gypi2', 1), [('Gypi2', 1), ('Gxpi2', 2)]], line_labels=(0, 1, 2)),
             Circuit([('Gcnot', 0, 1), ('Gxpi2', 2)], line_labels=(0, 1, 2))]
    lay = m.sim.create_layout(circs, array_types=("ep",))
    atom = lay.atoms[0]
    A.atom_plan(m, atom)
    assert m.sim._effective_mode(atom) == "analytic"
    G, R, E = A.atom_arrays(m, atom)
    mine = [str(l) for l in atom.op_labels]
    ref = [str(l) for l in fx["op_labels"]]
    for k, l in enumerate(mine):
        assert np.array_equal(G[k], fx["gates"][ref.index(l)]), l
    dv = A.atom_derivs(m, atom)
    assert len(dv) == len(mine) + 1 + len(atom._hip_eff_labels)
    shared = sum(1 for a in range(len(dv)) for b in range(a) if set(dv[a][2].tolist()) & set(dv[b][2].tolist()))
    assert shared >= 1                                    # Gypi2:1 alone and inside the composed layer
    # round 5: the layer structure itself goes to the device (gst_set_composite) -- three shared leaves behind every layer;
    # the host restatement of the device's builders reproduces the model's dense layers and derivative matrices exactly
    cm, spam_map = A.atom_composite(m, atom)
    assert sorted(cm.leaf_dims) == [4, 4, 16] and (spam_map[0] >= 0).sum() == 64 + 8 * 64
    assert np.array_equal(cm.dense_gates(cm.values(m.to_vector())), G)
    for (k, oi, idx, dm), (qs, mine_d) in zip([d for d in dv if d[0] == 0], cm.gate_derivs(cm.values(m.to_vector()))):
        assert np.array_equal(np.sort(idx), qs) and np.array_equal(dm[:, np.argsort(idx)], mine_d)
    if _lib.device_count() == 0:
        with pytest.raises(_lib.GstDeviceError):
            m.sim._prepare(atom, derivatives=True)
        assert atom._hip_plan._hip_mode == "composite"
        m.sim.composite_on_device = False                     # the host route of rounds 3-4 on request
        with pytest.raises(_lib.GstDeviceError):
            m.sim._prepare(atom, derivatives=True)
        assert atom._hip_plan._hip_mode != "composite"


def test_adapter_runs_the_qutrit_model_pack():
    """A state dimension that is none of 4 / 16 / 64: the reference's qutrit pack (D = 9, Gell-Mann basis).  The adapter hands
    9 x 9 gates and element indices i * 9 + j to the C ABI, which pads to 16 internally; the plan's programs interpreted in
    numpy on the un-padded arrays give pyGSTi's probabilities bit for bit, and the element map matches gpindices."""
    from pygsti.circuits import Circuit
    from pygsti.modelpacks.legacy import stdQT_XYIMS as std
    from pygsti.forwardsims import MapForwardSimulator
    m = std.target_model()
    m.set_all_parameterizations("full")
    m = m.depolarize(op_noise=0.01, spam_noise=0.01)
    circs = [Circuit([(l.name, 'T0') for l in (f1 + g * 2 + f2)], line_labels=('T0',))
             for g in std.germs_lite[:4] for f1 in std.prepStrs[:3] for f2 in std.effectStrs[:3]]
    circs = list(dict.fromkeys(circs))
    ref = m.copy(); ref.sim = MapForwardSimulator()
    lay_ref = ref.sim.create_layout(circs, array_types=("e",))
    p_ref = np.empty(lay_ref.num_elements); ref.sim.bulk_fill_probs(p_ref, lay_ref)
    m.sim = A.HipMapForwardSimulator()
    lay = m.sim.create_layout(circs, array_types=("e", "ep"))
    atom = lay.atoms[0]
    plan = A.atom_plan(m, atom)
    assert plan.D == 9
    G, R, E = A.atom_arrays(m, atom)
    plan.set_model(G, R, E)
    kind, obj, elem = A.atom_param_map(m, atom)
    plan.set_param_map(kind, obj, elem)
    assert m.sim._effective_mode(atom) == "fd"             # element-parameterised: the Map simulator's own finite differences
    Gb, Rb, Eb = plan.get_model()
    assert np.array_equal(Gb, G) and np.array_equal(Rb, R) and np.array_equal(Eb, E)
    w, off = plan.program()
    n = len(atom.elbl_indices_by_expcircuit)
    eff_ptr = np.zeros(n + 1, np.int64); el, ed = [], []
    for i in range(n):
        el.extend(atom.elbl_indices_by_expcircuit[i]); ed.extend(atom.elindices_by_expcircuit[i]); eff_ptr[i + 1] = len(el)
    out, written, _ = run_programs(w, off, G, R, E, eff_ptr, np.array(el), np.array(ed), atom.num_elements)
    assert (written == 1).all()
    assert_bitwise(out, p_ref, "qutrit: adapter plan vs pyGSTi bulk_fill_probs")


def test_adapter_general_leaves_for_cptplnd_implicit_models():
    """`create_crosstalk_free_model(ideal_gate_type='CPTPLND')`: one- and two-qubit gates = static target x exp(Lindblad
    generator), embedded and composed into 64 x 64 layers.  The description has GENERAL leaves (the exponentiated generators);
    what the host computes for them per model update -- values, deriv_wrt_params, values after each parameter step -- turns,
    through the restatement of the device's builders, into the reference's dense layers, layer derivatives and stepped dense
    models; the drop-in takes this route (and reaches the device call) for FD and for exact derivatives."""
    from pygsti.processors import QubitProcessorSpec
    from pygsti.models import modelconstruction as mc
    from pygsti.circuits import Circuit
    ps = QubitProcessorSpec(3, ['Gxpi2', 'Gypi2', 'Gcnot'], geometry='line')
    m = mc.create_crosstalk_free_model(ps, ideal_gate_type='CPTPLND', ideal_spam_type='full')
    m.from_vector(m.to_vector() + 0.01 * np.random.default_rng(7).standard_normal(m.num_params))
    m.sim = A.HipMapForwardSimulator()
    circs = [Circuit([[('Gxpi2', 0), ('Gcnot', 1, 2)], ('Gypi2', 1), [('Gypi2', 1), ('Gxpi2', 2)], ('Gcnot', 1, 0)], line_labels=(0, 1, 2))]
    lay = m.sim.create_layout(circs, array_types=("ep",))
    atom = lay.atoms[0]
    A.atom_plan(m, atom)
    cm, spam_map = A.atom_composite(m, atom)
    assert sorted(len(cm.leaf_general[l]) for l in cm.general_leaves) == [12, 12, 240]
    v0 = m.to_vector().copy()
    vals, dvs, fds = A.composite_general_data(m, cm, True, 1e-7)
    assert np.array_equal(m.to_vector(), v0)                         # (the stepping leaves the model where it was)
    G, R, E = A.atom_arrays(m, atom)
    v = cm.values(v0, vals)
    assert np.abs(cm.dense_gates(v) - G).max() < 1e-15
    for (k, oi, idx, dm), (qs, mine) in zip([d for d in A.atom_derivs(m, atom) if d[0] == 0], cm.gate_derivs(v, dvs)):
        assert np.array_equal(np.sort(idx), qs) and np.abs(dm[:, np.argsort(idx)] - mine).max() < 1e-14
    cols = np.array([0, 70, 576, 583, 590, 600, 700, 839])
    Gs, Rs, Es = cm.model_sets(v, R, E, spam_map, cols, 1e-7, fds)
    Gr, Rr, Er = A.atom_model_sets(m, atom, cols, 1e-7)
    assert np.abs(Gs - Gr).max() < 1e-15 and np.array_equal(Rs, Rr) and np.array_equal(Es, Er)
    if _lib.device_count() == 0:
        for mode in ("auto", "fd"):
            m.sim.derivative_mode = mode
            with pytest.raises(_lib.GstDeviceError):
                m.sim._prepare(atom, derivatives=True)
            assert atom._hip_plan._hip_mode == "composite"
        assert np.array_equal(m.to_vector(), v0)
