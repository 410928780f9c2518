"""Implicit models' layer operations (gst_set_composite) -- the part that runs without a GPU.

The fixture `3q_crosstalk_free_composite` holds the structure pygsti_adapter.atom_composite extracted from the REAL
`create_crosstalk_free_model` model of `3q_crosstalk_free` (tests/golden/make_golden_r5.py, which asserts array equality
with the reference's dense layers and deriv_wrt_params while generating).  Here: the numpy restatement of the device's
builders (pygsti_amd/composite.py) against the reference's own vectors in the older fixture; the walk of its dense model
sets through the CPU checker against the Map simulator's finite differences; the C ABI's validation."""
import numpy as np
import pytest

from conftest import load_fixture
from pygsti_amd import _lib
from pygsti_amd.composite import CompositeModel


def composite_from_fixture(fx, cf):
    dims = [int(d) for d in cf["cmp_leaf_dim"]]
    off = np.concatenate([[0], np.cumsum([d * d for d in dims])])
    fptr = cf["cmp_gate_factor_ptr"]
    dim_q = {4: 1, 16: 2, 64: 3}
    gate_factors = [[(int(cf["cmp_factor_leaf"][f]), tuple(int(t) for t in cf["cmp_factor_targets"][f][:dim_q[dims[int(cf["cmp_factor_leaf"][f])]]]))
                     for f in range(fptr[g], fptr[g + 1])] for g in range(len(fptr) - 1)]
    general = None
    if "cmp_leaf_n_params" in cf:
        n_gen = cf["cmp_leaf_n_params"]
        lo = np.concatenate([[0], np.cumsum(n_gen)])
        general = [None if n_gen[l] == 0 else cf["cmp_leaf_param_list"][lo[l]:lo[l + 1]] for l in range(len(dims))]
    return CompositeModel(int(fx["D"]), int(fx["nP"]), dims, [cf["cmp_leaf_param"][off[l]:off[l + 1]] for l in range(len(dims))],
                          [cf["cmp_leaf_static"][off[l]:off[l + 1]] for l in range(len(dims))], gate_factors, general)


def general_data_from_fixture(cm, cf):
    """{l: [d*d, np]} derivative matrices and {l: [np, d*d]} stepped values of the general leaves, as the fixture packs them"""
    dvs, fds, at = {}, {}, 0
    for l in cm.general_leaves:
        n = cm.leaf_dims[l] ** 2 * len(cm.leaf_general[l])
        dvs[l] = cf["cmp_general_derivs"][at:at + n].reshape(cm.leaf_dims[l] ** 2, -1)
        fds[l] = cf["cmp_general_fd"][at:at + n].reshape(-1, cm.leaf_dims[l] ** 2)
        at += n
    return dvs, fds


def test_restatement_reproduces_the_reference_model():
    fx, cf = load_fixture("3q_crosstalk_free"), load_fixture("3q_crosstalk_free_composite")
    cm = composite_from_fixture(fx, cf)
    assert cm.leaf_dims == [16, 4, 4] or sorted(cm.leaf_dims) == [4, 4, 16]
    v = cm.values(fx["paramvec"])
    assert np.array_equal(v, cf["cmp_leaf_values"])
    # three leaves, 288 parameters, every one of them behind several layers (independent_gates=False)
    assert sum(len(f) for f in cm.gate_factors) == 11 and len(cm.gate_factors) == 8
    shared = [sum(1 for fs in cm.gate_factors if any(l == leaf for l, _ in fs)) for leaf in range(3)]
    assert min(shared) >= 2
    assert np.array_equal(cm.dense_gates(v), fx["gates"])                         # pyGSTi's to_dense() of every layer, exactly
    gd = cm.gate_derivs(v)
    off_c = off_d = 0
    for k, oi, n in zip(fx["dv_kind"], fx["dv_obj"], fx["dv_ncols"]):            # pyGSTi's deriv_wrt_params of every layer, exactly
        K = 64 * 64 if k == 0 else 64
        idx = fx["dv_param_idx"][off_c:off_c + n]; dm = fx["dv_deriv"][off_d:off_d + K * n].reshape(K, n)
        off_c += n; off_d += K * n
        if k == 0:
            qs, d = gd[int(oi)]
            assert np.array_equal(np.sort(idx), qs) and np.array_equal(dm[:, np.argsort(idx)], d)
    # embedding convention: qubit 0 is the most significant digit; reversed targets transpose the leaf's tensor factors
    f = np.arange(16.0).reshape(4, 4)
    I = np.eye(4)
    assert np.array_equal(cm.embed(f, (0,)), np.kron(np.kron(f, I), I)) and np.array_equal(cm.embed(f, (2,)), np.kron(np.kron(I, I), f))
    g2 = np.arange(256.0).reshape(16, 16)
    assert np.array_equal(cm.embed(g2, (1, 2)), np.kron(I, g2))
    swap = g2.reshape(4, 4, 4, 4).transpose(1, 0, 3, 2).reshape(16, 16)
    assert np.array_equal(cm.embed(g2, (2, 1)), np.kron(I, swap))


def test_model_sets_walked_by_the_checker_give_the_map_simulators_columns(oracle_built):
    """The dense model after every finite-difference step, as the device builds it (restated in numpy), walked by the CPU
    checker: (p(set) - p) / eps equals the Map simulator's columns to the accuracy of dense-vs-factored propagation."""
    fx, cf = load_fixture("3q_crosstalk_free"), load_fixture("3q_crosstalk_free_composite")
    cm = composite_from_fixture(fx, cf)
    v = cm.values(fx["paramvec"])
    cols = fx["dprobs_cols"]
    pm = (cf["cmp_spam_kind"], cf["cmp_spam_obj"], cf["cmp_spam_elem"])
    G, R, E = cm.model_sets(v, fx["rhos"], fx["effects"], pm, cols, float(fx["derivative_eps"]))
    keys = ("D", "nE", "cache_size", "t_dest", "t_start", "t_cache", "t_rho", "row_ptr", "gate_idx", "eff_ptr", "eff_label", "eff_dest")
    orc = oracle_built.Oracle({k: fx[k] for k in keys}, dict(gates=fx["gates"].copy(), rhos=fx["rhos"].copy(), effects=fx["effects"].copy(),      # (Oracle.set_model writes in place)
                                                                pkind=np.zeros(0, np.int32), pobj=np.zeros(0, np.int32), pelem=np.zeros(0, np.int32)))
    p0 = orc.probs()
    assert np.abs(p0 - fx["probs"]).max() < 1e-14
    J = np.empty((int(fx["nE"]), len(cols)))
    for c in range(len(cols)):
        orc.set_model(G[c], R[c], E[c])
        J[:, c] = (orc.probs() - p0) / float(fx["derivative_eps"])
    assert np.abs(J - fx["dprobs_map"]).max() < 1e-8
    assert np.abs(J).max() > 0.1


def test_abi_validates_the_description():
    fx, cf = load_fixture("3q_crosstalk_free"), load_fixture("3q_crosstalk_free_composite")
    from conftest import plan_from_fixture
    pl = _lib.Plan.from_table(fx['D'], len(fx['gates']), 1, len(fx['effects']), fx['nE'], fx['cache_size'], fx['t_dest'], fx['t_start'],
                              fx['t_cache'], fx['t_rho'], fx['row_ptr'], fx['gate_idx'], fx['eff_ptr'], fx['eff_label'], fx['eff_dest'])
    cm = composite_from_fixture(fx, cf)
    pl.set_composite(cm)                                   # host-side: copied and checked, no device needed
    pl.set_composite(None)
    bad = composite_from_fixture(fx, cf)
    bad.gate_factors[2] = [(bad.gate_factors[2][0][0], (3,))]          # a qubit the register does not have
    with pytest.raises(ValueError, match="bad target qubits"):
        pl.set_composite(bad)
    bad = composite_from_fixture(fx, cf)
    bad.leaf_params[0] = bad.leaf_params[0].copy(); bad.leaf_params[0][0] = 10 ** 6    # parameter index out of range
    with pytest.raises(ValueError, match="out of range"):
        pl.set_composite(bad)
    if _lib.device_count() == 0:
        pl.set_composite(cm)
        with pytest.raises(_lib.GstDeviceError):            # building the layers needs the device: no CPU fallback
            pl.set_composite_values(cm.values(fx["paramvec"]), fx["rhos"], fx["effects"])


def test_general_leaves_cptplnd_implicit_model(oracle_built):
    """D = 64 with CPTPLND leaves (`3q_crosstalk_free_CPTPLND`, make_golden_r5.py cptplnd): every gate a static target times an
    exponentiated Lindblad generator on one or two qubits, embedded and composed into 64 x 64 layers.  The leaves' values,
    deriv_wrt_params and finite-difference steps come from the reference's own members (stored in the fixture: they are what
    gst_set_composite_values / _general receive); the restatement of the device's builders turns them into the reference's
    dense layers, and the model sets walked by the CPU checker give the Map simulator's columns."""
    fx = load_fixture("3q_crosstalk_free_CPTPLND")
    cm = composite_from_fixture(fx, fx)
    assert sorted(cm.leaf_dims) == [4, 4, 4, 4, 16, 16] and sorted(len(cm.leaf_general[l]) for l in cm.general_leaves) == [12, 12, 240]
    dvs, fds = general_data_from_fixture(cm, fx)
    v = fx["cmp_leaf_values"]
    assert np.abs(cm.dense_gates(v) - fx["gates"]).max() < 1e-15
    cols = fx["dprobs_cols"]
    pm = (fx["cmp_spam_kind"], fx["cmp_spam_obj"], fx["cmp_spam_elem"])
    eps = float(fx["derivative_eps"])
    G, R, E = cm.model_sets(v, fx["rhos"], fx["effects"], pm, cols, eps, fds)
    keys = ("D", "nE", "cache_size", "t_dest", "t_start", "t_cache", "t_rho", "row_ptr", "gate_idx", "eff_ptr", "eff_label", "eff_dest")
    orc = oracle_built.Oracle({k: fx[k] for k in keys}, dict(gates=fx["gates"].copy(), rhos=fx["rhos"].copy(), effects=fx["effects"].copy(),      # (Oracle.set_model writes in place)
                                                                pkind=np.zeros(0, np.int32), pobj=np.zeros(0, np.int32), pelem=np.zeros(0, np.int32)))
    p0 = orc.probs()
    assert np.abs(p0 - fx["probs"]).max() < 1e-13
    J = np.empty((int(fx["nE"]), len(cols)))
    for c in range(len(cols)):
        orc.set_model(G[c], R[c], E[c])
        J[:, c] = (orc.probs() - p0) / eps
    assert np.abs(J - fx["dprobs_map"]).max() < 1e-8 and np.abs(J).max() > 0.1
    # exact: layer derivative matrices by the product rule, contracted with the numpy element Jacobian = the Matrix simulator's
    from oracle import oracle as O
    from conftest import matrix_rows_by_circuit
    Je, _ = O.analytic_dprobs({**fx, "pkind": np.concatenate([np.full(64, 1), np.full(8 * 64, 2), np.zeros(len(fx["gates"]) * 4096)]).astype(np.int32),
                            "pobj": np.concatenate([np.zeros(64), np.repeat(np.arange(8), 64), np.repeat(np.arange(len(fx["gates"])), 4096)]).astype(np.int32),
                            "pelem": np.concatenate([np.arange(64), np.tile(np.arange(64), 8), np.tile(np.arange(4096), len(fx["gates"]))]).astype(np.int32),
                            "nP": 64 + 512 + 4096 * len(fx["gates"])})
    nE = int(fx["nE"])
    Jx = np.zeros((nE, int(fx["nP"])))
    Jx[:, :64] = Je[:, :64]
    for q in range(64, 576):
        Jx[:, q] = Je[:, 64 + fx["cmp_spam_obj"][q] * 64 + fx["cmp_spam_elem"][q]]
    for g, (qs, dm) in enumerate(cm.gate_derivs(v, dvs)):
        Jx[:, qs] += Je[:, 576 + g * 4096:576 + (g + 1) * 4096] @ dm
    rows = matrix_rows_by_circuit(fx)
    assert np.abs(Jx[:, cols] - fx["matrix_by_circuit_dprobs"][rows]).max() < 1e-8
