"""Lindblad-parameterised members without pyGSTi (pygsti_amd/lindblad.py): the host statement of what the device's
model builder computes, pinned to vectors of the reference (tests/golden/lindblad_*.npz, written by
make_golden_lindblad.py from `target_model('CPTPLND')` models; the dense models after every finite-difference step are
the `mm_*` arrays of the big CPTPLND fixtures)."""
import numpy as np
import pytest

from conftest import load_fixture
from pygsti_amd import lindblad as LB

CASES = [("smq1Q_XYI_L4_CPTPLND", 1), ("smq2Q_XYICNOT_L1_CPTPLND", 2)]


@pytest.mark.parametrize("name,nq", CASES)
def test_native_lindblad_members_match_the_reference(name, nq):
    fx, lb = load_fixture(name), load_fixture("lindblad_" + name)
    model = LB.LindbladModel.from_fixture(lb, nq)
    th = lb["paramvec"]
    assert model.num_params == int(fx["nP"])
    for m, mem in enumerate(model.members):
        loc = th[mem.param0:mem.param0 + mem.n_params]
        # error generator (lindbladerrorgen.py:699-703) and its exponential (experrorgenop.py:120)
        assert np.abs(mem.errorgen(loc) - lb["m%d_errgen" % m]).max() < 1e-15
        assert np.abs(mem.exp(loc) - lb["m%d_exp" % m]).max() < 1e-15
    G, R, E = model.dense(th)
    assert np.abs(G - fx["gates"]).max() < 1e-15 and np.abs(R - fx["rhos"]).max() < 1e-15 and np.abs(E - fx["effects"]).max() < 1e-15
    # the dense model after every FD step the reference took (set_parameter_value(i, theta_i + eps))
    Gs, Rs, Es = model.model_sets(th, fx["dprobs_cols"], float(fx["derivative_eps"]))
    assert np.abs(Gs - fx["mm_gates"]).max() < 1e-14 and np.abs(Rs - fx["mm_rhos"]).max() < 1e-14 and np.abs(Es - fx["mm_effects"]).max() < 1e-14


def test_coefficient_blocks():
    """'other' blocks: cholesky gives a positive semidefinite Hermitian matrix, elements a Hermitian one, both with the
    reference's parameter placement (lindbladcoefficients.py:315-470); diagonal cholesky squares."""
    rng = np.random.default_rng(0)
    n = 3
    v = rng.standard_normal(n * n)
    c = LB.block_coefficients(LB.BLOCK_OTHER, LB.MODE_CHOLESKY, n, v).reshape(n, n)
    assert np.allclose(c, c.conj().T) and np.linalg.eigvalsh(c).min() > -1e-12
    p = v.reshape(n, n)
    C = np.array([[p[0, 0], 0, 0], [p[1, 0] + 1j * p[0, 1], p[1, 1], 0], [p[2, 0] + 1j * p[0, 2], p[2, 1] + 1j * p[1, 2], p[2, 2]]])
    assert np.allclose(c, C @ C.conj().T)
    h = LB.block_coefficients(LB.BLOCK_OTHER, LB.MODE_ELEMENTS, n, v).reshape(n, n)
    assert np.allclose(h, h.conj().T) and h[1, 0] == p[1, 0] + 1j * p[0, 1] and h[2, 2] == p[2, 2]
    d = LB.block_coefficients(LB.BLOCK_OTHER_DIAGONAL, LB.MODE_CHOLESKY, n, v[:n])
    assert np.allclose(d, v[:n] ** 2)
    assert LB.block_num_params(LB.BLOCK_OTHER, LB.MODE_CHOLESKY, 15) == 225 and LB.block_num_params(LB.BLOCK_HAM, 0, 15) == 15


def test_term_superoperators_are_trace_preserving_generators():
    """Every Lindblad term generator annihilates the trace: its first row vanishes in the Pauli-product basis (whose
    first element is the normalised identity); Hamiltonian terms are real antisymmetric there."""
    for nq in (1, 2):
        for bt in (LB.BLOCK_HAM, LB.BLOCK_OTHER_DIAGONAL, LB.BLOCK_OTHER):
            S = LB.term_superops(nq, bt)
            assert np.abs(S[:, 0, :]).max() < 1e-12
        H = LB.term_superops(nq, LB.BLOCK_HAM)
        assert np.abs(H.imag).max() < 1e-12 and np.abs(H.real + H.real.transpose(0, 2, 1)).max() < 1e-12
    assert LB.pauli_labels(2)[:4] == ["IX", "IY", "IZ", "XI"]


def test_from_target_reproduces_the_reference_models_static_factors():
    """`LindbladModel.from_target(pack.target_model(), ...)` = the reference's `target_model('CPTPLND')`: same static
    factors (target superoperators, preparation, base effects), zero error generators, the reference's parameter order
    (preparation, POVM, operations)."""
    from pygsti_amd import modelpacks as MP
    for name, pack in (("smq1Q_XYI_L4_CPTPLND", MP.smq1Q_XYI), ("smq2Q_XYICNOT_L1_CPTPLND", MP.smq2Q_XYICNOT)):
        fx, lb = load_fixture(name), load_fixture("lindblad_" + name)
        target = pack.target_model()
        labels = [() if l == "[]" else l for l in fx["op_labels"]]
        keys = list(target.operations.keys())
        order = []
        for l in fx["op_labels"]:
            match = [k for k in keys if (str(k) == str(l)) or (l == "[]" and (k == () or str(k) in ("[]", "()")))]
            assert len(match) == 1, (l, keys)
            order.append(match[0])
        eff = ["Mdefault_" + str(l).split("_", 1)[1] for l in fx["eff_labels"]]
        model = LB.LindbladModel.from_target(target, order, eff, "CPTPLND")
        assert model.num_params == int(fx["nP"])
        ref = {(int(lb["m%d_kind" % m]), int(lb["m%d_obj" % m])): m for m in range(int(lb["n_members"]))}
        for mem in model.members:
            m = ref[(mem.kind, mem.obj)]
            assert np.abs(mem.static - lb["m%d_static" % m]).max() < 1e-15, (name, mem.kind, mem.obj)
            assert mem.param0 == int(lb["m%d_param0" % m])
        G, R, E = model.dense(np.zeros(model.num_params))
        assert np.abs(G - np.array([target.operations[k] for k in order])).max() < 1e-15


def test_lindblad_explicit_model_interface():
    """The host mirror's CPTPLND model object: zero error generators give the target model; `from_vector` moves the dense
    members; the layout and simulator accept it (element index, plan compilation); the description follows the plan's
    object order."""
    from pygsti_amd import modelpacks as MP
    from pygsti_amd.forwardsim import HipMapForwardSimulator
    pack = MP.smq1Q_XYI
    target = pack.target_model()
    m = LB.LindbladExplicitModel(target, "CPTPLND")
    assert m.num_params == 60 and m.dim == 4
    for k in target.operations:
        assert np.abs(m.operations[k] - target.operations[k]).max() < 1e-15
    th = 0.01 * np.random.default_rng(3).standard_normal(60)
    m.from_vector(th)
    assert np.array_equal(m.to_vector(), th)
    lb = load_fixture("lindblad_smq1Q_XYI_L4_CPTPLND"); fx = load_fixture("smq1Q_XYI_L4_CPTPLND")
    if np.allclose(th, lb["paramvec"]):
        assert np.abs(np.array(list(m.operations.values())) - fx["gates"]).max() < 1e-14
    sim = HipMapForwardSimulator(m)
    lay = sim.create_layout(pack.create_gst_circuits(2))
    assert lay.num_elements == 2 * len(pack.create_gst_circuits(2))
    desc = m.lindblad_description(lay.model_gate_labels, lay.effect_labels)
    assert desc is m.lindblad_description(lay.model_gate_labels, lay.effect_labels)
    assert [mm.kind for mm in desc.members] == [LB.KIND_RHO, LB.KIND_POVM] + [LB.KIND_GATE] * 3
    m2 = m.copy()
    assert np.array_equal(m2.to_vector(), th) and m2.sim is not m.sim


def test_member_derivatives_match_the_reference():
    """d(error generator)/d(parameter), d exp(L)/d(parameter) and d(dense member)/d(parameter) on the host -- the statement
    of what lindblad_deriv_kernel computes -- against LindbladErrorgen.deriv_wrt_params, ExpErrorgenOp.deriv_wrt_params
    (fixture `m*_derrgen`, `m*_dexp`) and the members' deriv_wrt_params (`dv_deriv` of the big fixture)."""
    fx, lb = load_fixture("smq1Q_XYI_L4_CPTPLND"), load_fixture("lindblad_smq1Q_XYI_L4_CPTPLND")
    model = LB.LindbladModel.from_fixture(lb, 1)
    th = lb["paramvec"]
    for m, mem in enumerate(model.members):
        loc = th[mem.param0:mem.param0 + mem.n_params]
        assert np.abs(mem.errorgen_deriv(loc).transpose(1, 2, 0).reshape(16, -1) - lb["m%d_derrgen" % m]).max() < 1e-14
        assert np.abs(mem.exp_deriv(loc).transpose(1, 2, 0).reshape(16, -1) - lb["m%d_dexp" % m]).max() < 1e-13
    off = 0
    for k, o, n in zip(fx["dv_kind"], fx["dv_obj"], fx["dv_ncols"]):
        K = 16 if k == 0 else 4
        ref = fx["dv_deriv"][off:off + K * n].reshape(K, n); off += K * n
        mem = [mm for mm in model.members if (mm.kind == k if k != 2 else mm.kind == LB.KIND_POVM) and (k == 2 or mm.obj == o)][0]
        dd = mem.dense_deriv(th[mem.param0:mem.param0 + mem.n_params])
        mine = dd[0] if k != 2 else dd[int(o) - mem.obj]
        assert np.abs(mine - ref).max() < 1e-13
    # 'elements' and diagonal blocks: coefficient Jacobians against finite differences
    rng = np.random.default_rng(1)
    for bt, md, n in ((LB.BLOCK_OTHER, LB.MODE_ELEMENTS, 3), (LB.BLOCK_OTHER, LB.MODE_CHOLESKY, 3), (LB.BLOCK_OTHER_DIAGONAL, LB.MODE_CHOLESKY, 3)):
        v = rng.standard_normal(LB.block_num_params(bt, md, n))
        J = LB.block_coefficient_derivs(bt, md, n, v)
        for q in range(len(v)):
            dv = np.zeros_like(v); dv[q] = 1e-6
            fd = (LB.block_coefficients(bt, md, n, v + dv) - LB.block_coefficients(bt, md, n, v - dv)) / 2e-6
            assert np.abs(J[:, q] - fd).max() < 1e-8


def test_three_qubit_members_match_the_reference():
    """A Lindblad member of FULL dimension 64 (4,032 parameters: 63 Hamiltonian + 63 x 63 Cholesky coefficients over the
    three-qubit Paulis): the native construction -- term superoperators from Pauli matrices, coefficients, generator,
    exponential, composition with the static factor -- against the reference's own members of a three-qubit explicit CPTPLND
    model (tests/golden/make_golden_r6.py: `errorgen.to_dense()`, `ExpErrorgenOp.to_dense()`, the dense members)."""
    fx, lb = load_fixture("3q_explicit_CPTPLND"), load_fixture("lindblad_3q_explicit_CPTPLND")
    model = LB.LindbladModel.from_fixture(lb, 3)
    th = lb["paramvec"]
    assert model.num_params == len(th) == 5 * 4032 and model.D == 64
    assert model.members[0].term_re is model.members[1].term_re          # (one 132 MB table, shared)
    for m, mem in enumerate(model.members):
        loc = th[mem.param0:mem.param0 + mem.n_params]
        L = mem.errorgen(loc)
        assert np.abs(L - lb["m%d_errgen" % m]).max() < 1e-13 * max(1.0, np.abs(L).max()), m
        assert np.abs(mem.exp(loc) - lb["m%d_exp" % m]).max() < 1e-13, m
    G, R, E = model.dense(th)
    assert np.abs(G - fx["gates"]).max() < 1e-13 and np.abs(R - fx["rhos"]).max() < 1e-13 and np.abs(E - fx["effects"]).max() < 1e-13
