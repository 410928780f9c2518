#!/usr/bin/env python3
"""Generate the Lindblad-member descriptions of the CPTPLND fixtures by IMPORTING the reference.

    PYTHONPATH=/tmp/pgref OMP_NUM_THREADS=1 python3 tests/golden/make_golden_lindblad.py

Build container only (needs the scratch build of the reference); only the `.npz` it writes --
pure data -- is committed.  For the two CPTPLND models of make_golden.py (`smq1Q_XYI_L4_CPTPLND`,
`smq2Q_XYICNOT_L1_CPTPLND`, rebuilt here with the same seeds and checked against the parameter vectors those fixtures
hold) it records, per model member in the fixtures' object order,

  inputs : what a Lindblad-parameterised member consists of in the reference --
           ComposedOp([static target, ExpErrorgenOp(LindbladErrorgen)]) (modelmembers/operations/composedop.py,
           experrorgenop.py:49-213, lindbladerrorgen.py:494-742), ComposedState(static state, error map)
           (modelmembers/states/composedstate.py), ComposedPOVM(error map, computational-basis POVM)
           (modelmembers/povms/composedpovm.py) -- as plain arrays: the static factor (dense superoperator / state vector /
           base effect vectors, Pauli-product basis), the first model parameter, the coefficient blocks' types, parameter
           modes and sizes (lindbladcoefficients.py: 'ham'/'elements', 'other'/'cholesky');
  outputs: the member's dense error generator `errorgen.to_dense()` and its exponential `ExpErrorgenOp.to_dense()`
           (scipy.linalg.expm, experrorgenop.py:120) at the fixture's parameter vector, and for the 1Q model the
           member's `deriv_wrt_params()` of the error generator (lindbladerrorgen.py:1342).

The dense models after every finite-difference step are already in the big fixtures (`mm_*`); tests pin the native
construction (pygsti_amd/lindblad.py, the device's model builder) to both.
"""
import os

import numpy as np

import pygsti  # noqa: F401
from pygsti.modelpacks import smq1Q_XYI, smq2Q_XYICNOT

HERE = os.path.dirname(os.path.abspath(__file__))
BLOCK_TYPES = {"ham": 0, "other_diagonal": 1, "other": 2}
PARAM_MODES = {"elements": 0, "cholesky": 1, "static": 2, "depol": 3, "reldepol": 4}


def describe(model, fixture_name, want_derivs):
    fx = dict(np.load(os.path.join(HERE, fixture_name + ".npz")))
    assert np.array_equal(model.to_vector(), fx["paramvec"]), "model rebuilt with other parameters than the fixture's"
    out = {"paramvec": model.to_vector()}
    members = []
    for kind, labels, typ in ((0, fx["op_labels"], "op"), (1, fx["rho_labels"], "prep")):
        for oi, lbl in enumerate(labels):
            key = pygsti.baseobjs.Label(()) if lbl == "[]" else pygsti.baseobjs.Label(tuple(str(lbl).split(":")[:1]) + tuple(int(q) for q in str(lbl).split(":")[1:]))
            member = (model.operations if kind == 0 else model.preps)[key if kind == 0 else str(lbl)]
            if kind == 0:
                static, expop = member.factorops
                static_dense = static.to_dense("HilbertSchmidt").real
            else:
                static, expop = member.state_vec, member.error_map
                static_dense = static.to_dense("HilbertSchmidt").real
            members.append((kind, oi, 1, member, static_dense, expop))
    # one POVM (the plan's effects are its effects, in the fixture's effect order)
    povm = model.povms["Mdefault"]
    eff_names = [str(l).split("_", 1)[1] for l in fx["eff_labels"]]
    base = np.array([povm.base_povm[n].to_dense("HilbertSchmidt").real for n in eff_names])
    members.append((2, 0, len(eff_names), povm, base, povm.error_map))

    for m, (kind, oi, n_eff, member, static_dense, expop) in enumerate(members):
        L = expop.errorgen
        gp = member.gpindices_as_array()
        assert np.array_equal(gp, np.arange(gp[0], gp[0] + len(gp))), "contiguous parameter slices expected"
        blocks = [(BLOCK_TYPES[b._block_type], PARAM_MODES[b._param_mode], len(b._bel_labels), b.num_params) for b in L.coefficient_blocks]
        assert set(str(L.matrix_basis.name).split("*")) == {"pp"}, L.matrix_basis.name      # ('pp*pp' for two qubits: the same basis)
        for b in L.coefficient_blocks:
            assert set(str(b._basis.name).split("*")) == {"PP"}, b._basis.name
            out["m%d_bel_labels_%d" % (m, len(b._bel_labels))] = np.array([str(x) for x in b._bel_labels])
        pre = "m%d_" % m
        out[pre + "kind"] = np.int32(kind); out[pre + "obj"] = np.int32(oi); out[pre + "n_eff"] = np.int32(n_eff)
        out[pre + "param0"] = np.int64(gp[0]); out[pre + "n_params"] = np.int64(len(gp))
        out[pre + "blocks"] = np.array(blocks, np.int32)            # (block type, param mode, basis size, params)
        out[pre + "static"] = np.ascontiguousarray(static_dense, np.float64)
        out[pre + "errgen"] = np.ascontiguousarray(L.to_dense("HilbertSchmidt"), np.float64)
        out[pre + "exp"] = np.ascontiguousarray(expop.to_dense("HilbertSchmidt"), np.float64)
        if want_derivs:
            out[pre + "derrgen"] = np.ascontiguousarray(L.deriv_wrt_params(), np.float64)       # [D*D][n_params]
            out[pre + "dexp"] = np.ascontiguousarray(expop.deriv_wrt_params(), np.float64)       # [D*D][n_params]
    out["n_members"] = np.int32(len(members))
    path = os.path.join(HERE, "lindblad_" + fixture_name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1e3))


def main():
    m = smq1Q_XYI.target_model("CPTPLND")
    m.from_vector(m.to_vector() + 0.01 * np.random.default_rng(3).standard_normal(m.num_params))
    describe(m, "smq1Q_XYI_L4_CPTPLND", True)
    m = smq2Q_XYICNOT.target_model("CPTPLND")
    m.from_vector(m.to_vector() + 0.003 * np.random.default_rng(9).standard_normal(m.num_params))
    describe(m, "smq2Q_XYICNOT_L1_CPTPLND", False)


if __name__ == "__main__":
    main()
