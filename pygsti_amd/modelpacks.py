"""Model packs: the two stock gate sets the BASELINE configs use, re-stated for the host mirror.

Mirrors the slice of pyGSTi's modelpack interface that feeds the forward simulator
(pygsti/modelpacks/_modelpack.py: `target_model`, `prep_fiducials`, `meas_fiducials`, `germs`,
`create_gst_experiment_design(max_max_length, lite=...)` :331-400) -- enough to build the BASELINE
workloads on a box where pyGSTi itself is absent.  Circuits are plain tuples of gate-label strings
(the reference's `str(Label)`: 'Gxpi2:0', 'Gcnot:0:1', '[]' for the idle layer).

The generated experiment designs are pinned (count, total depth and sha256 of the integerised list,
in the reference's circuit order) by tests/test_modelpacks.py against tests/golden/designs.npz.
"""
import itertools

import numpy as np

from . import _packdata as _pd
from .model import ExplicitDenseModel

_SQ2 = np.sqrt(2.0)
_I2 = np.eye(2, dtype=complex)
_X = np.array([[0, 1], [1, 0]], dtype=complex)
_Y = np.array([[0, -1j], [1j, 0]], dtype=complex)
_Z = np.array([[1, 0], [0, -1]], dtype=complex)


def _pauli_basis(nq):
    """Normalised Pauli-product basis {I,X,Y,Z}^{(x)nq} / sqrt(2)^nq ('pp' in the reference)."""
    mats = []
    for idx in itertools.product(range(4), repeat=nq):
        m = np.array([[1.0 + 0j]])
        for k in idx:
            m = np.kron(m, (_I2, _X, _Y, _Z)[k] / _SQ2)
        mats.append(m)
    return mats


def _superop_pp(U, basis):
    """Superoperator of rho -> U rho U^dag in the Pauli-product basis, rounded to exact -1/0/1
    entries for the Clifford gates used here."""
    D = len(basis)
    S = np.empty((D, D))
    for i, Pi in enumerate(basis):
        for j, Pj in enumerate(basis):
            S[i, j] = np.real(np.trace(Pi.conj().T @ U @ Pj @ U.conj().T))
    R = np.round(S)
    assert np.abs(S - R).max() < 1e-12, "non-Clifford gate: keep unrounded values"
    return R + 0.0   # +0.0 turns -0.0 into 0.0


def _vec_pp(rho, basis):
    return np.array([np.real(np.trace(P.conj().T @ rho)) for P in basis])


def _rot(P, theta):
    return np.cos(theta / 2) * np.eye(P.shape[0]) - 1j * np.sin(theta / 2) * P


def repeat_with_max_length(germ, max_len):
    """germ repeated floor(max_len / len(germ)) times ('whole germ powers' truncation,
    pygsti/circuits/gstcircuits.py:931-941)."""
    if len(germ) == 0:
        return ()
    return tuple(germ) * (max_len // len(germ))


def _gen_max_length(max_max_length):
    L, out = 1, []
    while L <= max_max_length:
        out.append(L)
        L *= 2
    return out


class ModelPack:
    def __init__(self, name, nq, gate_labels, unitaries, prep_fids, meas_fids, germs_lite, germs_full):
        self.name = name
        self.nq = nq
        self.gate_labels = tuple(gate_labels)        # order of target_model().operations in the reference
        self._unitaries = unitaries
        self._prep, self._meas = prep_fids, meas_fids
        self._germs_lite, self._germs_full = germs_lite, germs_full

    def prep_fiducials(self):
        return list(self._prep)

    def meas_fiducials(self):
        return list(self._meas)

    def germs(self, lite=True):
        return list(self._germs_lite if lite else self._germs_full)

    def target_model(self):
        basis = _pauli_basis(self.nq)
        d = 2 ** self.nq
        ops = {lbl: _superop_pp(self._unitaries[lbl], basis) for lbl in self.gate_labels}
        rho = np.zeros((d, d), complex); rho[0, 0] = 1.0
        effects = {}
        for k in range(d):
            E = np.zeros((d, d), complex); E[k, k] = 1.0
            effects[format(k, "0%db" % self.nq)] = _vec_pp(E, basis)
        return ExplicitDenseModel(ops, {"rho0": _vec_pp(rho, basis)}, {"Mdefault": effects})

    def create_gst_circuits(self, max_max_length, lite=True, germs=None):
        """The circuit list of `create_gst_experiment_design(max_max_length, lite=lite)
        .all_circuits_needing_data`, in the reference's order: for each L in 1,2,4,..: the
        fiducial-pair plaquettes (prep_i + germ^p + meas_j, prep index outer, meas index inner) of
        every germ whose power changed, then -- after the first L -- the LGST circuits
        (gstcircuits.py:272-520 with nest=True, include_lgst=True), de-duplicated keeping first
        occurrences."""
        if isinstance(max_max_length, (list, tuple)):
            max_lengths = list(max_max_length)
        else:
            max_lengths = _gen_max_length(max_max_length)
        germs = self.germs(lite) if germs is None else list(germs)
        prep, meas = self._prep, self._meas
        seen = {}

        def add(c):
            if c not in seen:
                seen[c] = None

        def lgst_list():
            # create_lgst_circuits (circuitconstruction.py:503-506): eStr, prepStr, prepStr+eStr,
            # prepStr+g+eStr with loop order g, prepStr, eStr
            for e in meas: yield tuple(e)
            for p in prep: yield tuple(p)
            for p in prep:
                for e in meas: yield tuple(p) + tuple(e)
            for g in self._lgst_op_order():
                for p in prep:
                    for e in meas: yield tuple(p) + (g,) + tuple(e)

        done_bases = set()
        for i, L in enumerate(max_lengths):
            if i == 0:
                for p in prep:                       # the empty-germ plaquette = all fiducial pairs
                    for e in meas: add(tuple(p) + tuple(e))
                done_bases.add(())
            for germ in germs:
                base = repeat_with_max_length(germ, L)
                if len(base) == 0 or base in done_bases:
                    continue
                done_bases.add(base)
                for p in prep:
                    for e in meas: add(tuple(p) + base + tuple(e))
            if i == 0:
                for c in lgst_list(): add(c)
        return list(seen.keys())

    def _lgst_op_order(self):
        # processor_spec().primitive_op_labels order (gate names in pack order, then qubits)
        return self._lgst_ops


def _make_1q():
    U = {"[]": _I2, "Gxpi2:0": _rot(_X, np.pi / 2), "Gypi2:0": _rot(_Y, np.pi / 2)}
    mp = ModelPack("smq1Q_XYI", 1, _pd.SMQ1Q_GATES, U, _pd.SMQ1Q_PREP_FIDUCIALS, _pd.SMQ1Q_MEAS_FIDUCIALS,
                   _pd.SMQ1Q_GERMS_LITE, _pd.SMQ1Q_GERMS_FULL)
    mp._lgst_ops = ("Gxpi2:0", "Gypi2:0")
    return mp


def _make_2q():
    # qubit 0 is the left tensor factor (state |q0 q1>)
    cnot = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 1], [0, 0, 1, 0]], dtype=complex)
    U = {"[]": np.eye(4, dtype=complex),
         "Gxpi2:1": np.kron(_I2, _rot(_X, np.pi / 2)), "Gypi2:1": np.kron(_I2, _rot(_Y, np.pi / 2)),
         "Gxpi2:0": np.kron(_rot(_X, np.pi / 2), _I2), "Gypi2:0": np.kron(_rot(_Y, np.pi / 2), _I2),
         "Gcnot:0:1": cnot}
    mp = ModelPack("smq2Q_XYICNOT", 2, _pd.SMQ2Q_GATES, U, _pd.SMQ2Q_PREP_FIDUCIALS, _pd.SMQ2Q_MEAS_FIDUCIALS,
                   _pd.SMQ2Q_GERMS_LITE, _pd.SMQ2Q_GERMS_FULL)
    mp._lgst_ops = ("Gxpi2:1", "Gxpi2:0", "Gypi2:1", "Gypi2:0", "Gcnot:0:1")
    return mp


smq1Q_XYI = _make_1q()
smq2Q_XYICNOT = _make_2q()
