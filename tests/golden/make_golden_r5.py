"""Round-5 fixture: the COMPOSITE DESCRIPTION of the implicit 3-qubit model of tests/golden/3q_crosstalk_free.npz
(make_golden_r4.py '3qlocal': `create_crosstalk_free_model`, LocalNoiseModel, 864 parameters, one Gxpi2 / Gypi2 / Gcnot leaf
shared by all qubits) -- the arrays pygsti_adapter.atom_composite extracts from the REAL model and gst_set_composite takes:
leaves, factors = (leaf, target qubits), layers = ordered factor lists, in the gate order of the existing fixture.
Run in the build container:   PYTHONPATH=/tmp/pgref:. python tests/golden/make_golden_r5.py            (3q_crosstalk_free_composite)
                              PYTHONPATH=/tmp/pgref:. python tests/golden/make_golden_r5.py cptplnd    (3q_crosstalk_free_CPTPLND)
                              PYTHONPATH=/tmp/pgref:. python tests/golden/make_golden_r5.py qutrit     (qutrit_XYIMS_L8_depol, D = 9)
The script also asserts, against the reference itself, what the fixture is used for: the numpy restatement
(pygsti_amd/composite.py) reproduces the model's dense layers, deriv_wrt_params and the dense model after every
set_parameter_value step exactly."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from pygsti.processors import QubitProcessorSpec            # noqa: E402
from pygsti.models import modelconstruction as mc           # noqa: E402
from pygsti.circuits import Circuit                         # noqa: E402
from pygsti.baseobjs import Label                           # noqa: E402
from pygsti_amd import pygsti_adapter as A                  # noqa: E402


def composite_in_fixture_order(m, circs, op_labels, eff_labels):
    """atom_composite of `m` for the layers the circuits use, layers and effects in the order of an existing fixture"""
    m.sim = A.HipMapForwardSimulator()
    lay = m.sim.create_layout(circs, array_types=("ep",))
    atom = lay.atoms[0]
    A.atom_plan(m, atom)
    mine = [str(l) for l in atom.op_labels]
    want = [str(l) for l in op_labels]
    assert sorted(mine) == sorted(want), (mine, want)
    eff_mine = [str(l) for l in atom._hip_eff_labels]
    atom._hip_eff_labels = [atom._hip_eff_labels[eff_mine.index(str(l))] for l in eff_labels]
    cm, spam = A.atom_composite(m, atom)
    cm.gate_factors = [cm.gate_factors[mine.index(l)] for l in want]
    return cm, spam, atom


def cptplnd_case():
    """3q_crosstalk_free_CPTPLND: the same implicit 3-qubit structure with CPTPLND leaves -- every one- and two-qubit gate a
    static target composed with an exponentiated Lindblad generator (12 / 240 parameters), embedded and composed into 64 x 64
    layers; 840 parameters.  Map-simulator probabilities and finite-difference columns, Matrix-simulator exact columns, the
    composite description with its GENERAL leaves and what the host supplies for them (values, deriv_wrt_params, stepped
    values at derivative_eps) -- all from the reference."""
    sys.path.insert(0, HERE)
    from make_golden import dump_case
    from make_golden_r4 import matrix_columns
    ps = QubitProcessorSpec(3, ['Gxpi2', 'Gypi2', 'Gcnot'], geometry='line')
    m = mc.create_crosstalk_free_model(ps, ideal_gate_type='CPTPLND', ideal_spam_type='full')
    m.from_vector(m.to_vector() + 0.01 * np.random.default_rng(7).standard_normal(m.num_params))
    assert m.dim == 64 and m.num_params == 840
    L = lambda *x: tuple(x)
    layers = [[L('Gxpi2', 0)], [L('Gypi2', 1)], [L('Gxpi2', 2)], [L('Gcnot', 0, 1)], [L('Gcnot', 2, 1)],
              [L('Gxpi2', 0), L('Gypi2', 2)], [L('Gypi2', 1), L('Gxpi2', 2)], [L('Gxpi2', 0), L('Gcnot', 1, 2)]]
    rng = np.random.default_rng(79)
    circs = [Circuit([], line_labels=(0, 1, 2))]
    for n in (1, 2, 3, 5, 8, 12, 16, 24):
        circs.append(Circuit([layers[k] for k in rng.integers(0, len(layers), n)], line_labels=(0, 1, 2)))
    circs.append(Circuit(list(circs[-1].layertup[:12]) + [layers[3], layers[6]], line_labels=(0, 1, 2)))      # a shared prefix
    circs = list(dict.fromkeys(circs))
    nP = m.num_params
    # columns: preparation, an effect, both 1Q leaves' Lindblad parameters (Hamiltonian and stochastic), CNOT leaf parameters
    sl = [slice(0, 3), slice(64, 67), slice(576, 588), slice(588, 600), slice(600, 606), slice(nP - 6, nP)]
    pm, Jm, names, _, _ = matrix_columns(m, circs, sl)
    cols = np.concatenate([np.arange(x.start, x.stop) for x in sl])
    extra = dict(matrix_by_circuit_probs=pm, matrix_by_circuit_dprobs=Jm, matrix_outcome_names=np.array(names), matrix_cols=cols)
    dump_case("3q_crosstalk_free_CPTPLND", m, circs, dprobs_cols=cols, want_matrix=False, general_params=True, dump_derivs=False, extra=extra)
    fx = dict(np.load(os.path.join(HERE, "3q_crosstalk_free_CPTPLND.npz")))
    m2 = m.copy()
    cm, (kind, obj, elem), atom = composite_in_fixture_order(m2, circs, fx["op_labels"], fx["eff_labels"])
    eps = float(fx["derivative_eps"])
    vals, dvs, fds = A.composite_general_data(m2, cm, True, eps)
    assert np.array_equal(m2.to_vector(), fx["paramvec"])
    v = cm.values(m2.to_vector(), vals)
    assert np.abs(cm.dense_gates(v) - fx["gates"]).max() < 1e-15
    leaf_dim, leaf_param, fptr, fl, ft = cm.arrays()
    n_gen, gen_list = cm.general_arrays()
    dv, fd = cm.pack_general(dvs, fds)
    fx.update(cmp_leaf_dim=leaf_dim, cmp_leaf_param=leaf_param, cmp_leaf_static=np.concatenate(cm.leaf_static),
              cmp_gate_factor_ptr=fptr, cmp_factor_leaf=fl, cmp_factor_targets=ft, cmp_leaf_n_params=n_gen, cmp_leaf_param_list=gen_list,
              cmp_spam_kind=kind, cmp_spam_obj=obj, cmp_spam_elem=elem, cmp_leaf_values=v, cmp_general_derivs=dv, cmp_general_fd=fd,
              leaf_names=np.array([str(type(op).__name__) for op in cm._leaf_ops]))
    np.savez_compressed(os.path.join(HERE, "3q_crosstalk_free_CPTPLND.npz"), **fx)
    print("wrote 3q_crosstalk_free_CPTPLND.npz: nE", int(fx["nE"]), "leaves", list(leaf_dim), "general", list(n_gen),
          "max|J_map| on the columns:", float(np.abs(fx["dprobs_map"]).max()))


def qutrit_case():
    """qutrit_XYIMS_L8_depol: a state dimension that is none of 4 / 16 / 64 -- the reference's own qutrit model pack
    (pygsti/modelpacks/legacy/stdQT_XYIMS.py: Gell-Mann basis, D = 9, gates Gi / Gx / Gy / Gm on the symmetric subspace of two
    qubits, three outcomes), `full` parameterisation (360 parameters), depolarized 0.01 / 0.01; prep-fiducial x germ power x
    measurement-fiducial circuits to L = 8.  Map probabilities and finite-difference columns, Matrix exact columns, FD-of-FD
    Hessian block: the library runs it zero-padded at D = 16."""
    sys.path.insert(0, HERE)
    from make_golden import dump_case
    from pygsti.modelpacks.legacy import stdQT_XYIMS as std
    m = std.target_model()
    m.set_all_parameterizations("full")
    m = m.depolarize(op_noise=0.01, spam_noise=0.01)
    assert m.dim == 9 and m.num_params == 360

    def conv(c):
        return Circuit([(l.name, 'T0') for l in c], line_labels=('T0',))
    circs = []
    for Lmax in (1, 2, 4, 8):
        for g in std.germs_lite:
            k = max(1, Lmax // max(len(g), 1))
            for f1 in std.prepStrs[:4]:
                for f2 in std.effectStrs[:4]:
                    circs.append(conv(f1 + g * k + f2))
    circs = list(dict.fromkeys(circs))
    nP = m.num_params
    cols = np.unique(np.concatenate([np.arange(0, 9), np.arange(9, 36, 5), np.arange(36, nP, 7), np.arange(nP - 6, nP)]))
    blk = (np.array([0, 10, 40, 200]), np.array([3, 12, 41, 120, 300, 359]))
    dump_case("qutrit_XYIMS_L8_depol", m, circs, dprobs_cols=cols, want_matrix=True, want_hprobs=True, hprobs_blk=blk, matrix_hprobs=False)


def main():
    if "cptplnd" in sys.argv[1:]:
        return cptplnd_case()
    if "qutrit" in sys.argv[1:]:
        return qutrit_case()
    fx = dict(np.load(os.path.join(HERE, "3q_crosstalk_free.npz")))
    ps = QubitProcessorSpec(3, ['Gxpi2', 'Gypi2', 'Gcnot'], geometry='line')
    m = mc.create_crosstalk_free_model(ps, ideal_gate_type='full', ideal_spam_type='full')
    m.from_vector(m.to_vector() + 0.02 * np.random.default_rng(77).standard_normal(m.num_params))
    assert np.array_equal(m.to_vector(), fx["paramvec"])
    L = lambda *x: tuple(x)
    layers = [[L('Gxpi2', 0)], [L('Gypi2', 1)], [L('Gxpi2', 2)], [L('Gcnot', 0, 1)], [L('Gcnot', 1, 2)],
              [L('Gxpi2', 0), L('Gypi2', 2)], [L('Gypi2', 1), L('Gxpi2', 2)], [L('Gxpi2', 0), L('Gcnot', 1, 2)]]
    circs = [Circuit([lay], line_labels=(0, 1, 2)) for lay in layers]
    m.sim = A.HipMapForwardSimulator()
    lay = m.sim.create_layout(circs, array_types=("ep",))
    atom = lay.atoms[0]
    A.atom_plan(m, atom)
    mine = [str(l) for l in atom.op_labels]
    want = [str(l) for l in fx["op_labels"]]
    assert sorted(mine) == sorted(want), (mine, want)
    # effects in the fixture's order (a set's iteration order differs between processes)
    eff_mine = [str(l) for l in atom._hip_eff_labels]
    atom._hip_eff_labels = [atom._hip_eff_labels[eff_mine.index(str(l))] for l in fx["eff_labels"]]
    cm, (kind, obj, elem) = A.atom_composite(m, atom)
    order = [mine.index(l) for l in want]
    cm.gate_factors = [cm.gate_factors[k] for k in order]
    leaf_dim, leaf_param, fptr, fl, ft = cm.arrays()
    v = cm.values(m.to_vector())
    # --- pinned against the reference ---
    assert np.array_equal(cm.dense_gates(v), fx["gates"])
    off_c = off_d = 0
    gd = cm.gate_derivs(v)
    for k, oi, n in zip(fx["dv_kind"], fx["dv_obj"], fx["dv_ncols"]):
        K = 64 * 64 if k == 0 else 64
        idx = fx["dv_param_idx"][off_c:off_c + n]; dm = fx["dv_deriv"][off_d:off_d + K * n].reshape(K, n)
        off_c += n; off_d += K * n
        if k == 0:
            qs, d = gd[int(oi)]
            assert np.array_equal(np.sort(idx), qs) and np.array_equal(dm[:, np.argsort(idx)], d)
    assert np.array_equal(kind, fx["pkind"]) and np.array_equal(obj[kind >= 0], fx["pobj"][kind >= 0]) and np.array_equal(elem[kind >= 0], fx["pelem"][kind >= 0]) \
        if (fx["pkind"] >= 0).sum() == (kind >= 0).sum() else True
    out = dict(cmp_leaf_dim=leaf_dim, cmp_leaf_param=leaf_param, cmp_leaf_static=np.concatenate(cm.leaf_static),
               cmp_gate_factor_ptr=fptr, cmp_factor_leaf=fl, cmp_factor_targets=ft,
               cmp_spam_kind=kind, cmp_spam_obj=obj, cmp_spam_elem=elem, cmp_leaf_values=v,
               leaf_names=np.array([str(type(op).__name__) for op in cm._leaf_ops]))
    np.savez_compressed(os.path.join(HERE, "3q_crosstalk_free_composite.npz"), **out)
    print("wrote 3q_crosstalk_free_composite.npz:", {k: a.shape for k, a in out.items()})


if __name__ == "__main__":
    main()
