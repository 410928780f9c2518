"""Lindblad-parameterised model members (CPTPLND, GLND, H+S ...) without pyGSTi: the host side of SURVEY 8(f) row f4.

In the reference a CPTPLND model's members are compositions of a STATIC factor with an exponentiated Lindblad error
generator (`Model` built with `target_model('CPTPLND')`):

    gate   ComposedOp([StaticUnitaryOp U, ExpErrorgenOp(LindbladErrorgen L)])      dense = expm(L) . U
           (modelmembers/operations/composedop.py; experrorgenop.py:49-213, `_update_rep` :114-123 = scipy expm)
    prep   ComposedState(static state rho0, ExpErrorgenOp)                          dense = expm(L) . rho0
           (modelmembers/states/composedstate.py)
    POVM   ComposedPOVM(ExpErrorgenOp, ComputationalBasisPOVM)                      effect_i = expm(L)^T . e_i
           (modelmembers/povms/composedpovm.py, composedeffect.py)

and the error generator is LINEAR in complex coefficients c (lindbladerrorgen.py:658-742),

    L = Re( sum_k c_k S_k ),     c = concatenation of the coefficient blocks' block_data,

with S_k the Lindblad term superoperators of the blocks (lindbladcoefficients.py:664-702, 1073-1198;
tools/lindbladtools.py:489-553), written in the Pauli-product basis:
    'ham'             S = -i [P, .]                                           one per basis element P
    'other_diagonal'  S = P . P^dag - 1/2 {P^dag P, .}
    'other'           S_mn = P_n . P_m^dag - 1/2 {P_m^dag P_n, .}             one per PAIR (m, n), row-major
and c a function of the block's real parameters (lindbladcoefficients.py:164-470):
    'elements'  ham / other_diagonal: c = v;         other: c_ii = v_ii, c_ij = v_ij + i v_ji (i > j), Hermitian
    'cholesky'  other_diagonal: c = v^2;             other: c = C C^dag, C lower triangular, C_ii = v_ii,
                                                     C_ij = v_ij + i v_ji (i > j)   (v read as an n x n matrix, row-major)

This module builds the S_k from Pauli matrices, evaluates c(v), L and the dense members on the host (numpy/scipy) -- the
readable statement of what `gst_set_lindblad` makes the DEVICE do for the base model and for every finite-difference
step -- and packs the description the C ABI takes.  tests/test_lindblad.py pins it to vectors of the reference
(tests/golden/lindblad_*.npz, make_golden_lindblad.py).
"""
import itertools

import numpy as np

BLOCK_HAM, BLOCK_OTHER_DIAGONAL, BLOCK_OTHER = 0, 1, 2
MODE_ELEMENTS, MODE_CHOLESKY = 0, 1
KIND_GATE, KIND_RHO, KIND_POVM = 0, 1, 2

_PAULI = {"I": np.array([[1, 0], [0, 1]], complex), "X": np.array([[0, 1], [1, 0]], complex),
          "Y": np.array([[0, -1j], [1j, 0]], complex), "Z": np.array([[1, 0], [0, -1]], complex)}


def pauli_labels(n_qubits, with_identity=False):
    """'IX', 'IY', ... in the reference's order (itertools.product over 'IXYZ'; the all-identity label first)."""
    labels = ["".join(p) for p in itertools.product("IXYZ", repeat=n_qubits)]
    return labels if with_identity else labels[1:]


def pauli_matrix(label):
    """Un-normalised Pauli product (the reference's 'PP' basis element)."""
    m = np.ones((1, 1), complex)
    for ch in label:
        m = np.kron(m, _PAULI[ch])
    return m


def _pp_transform(n_qubits):
    """Columns = the normalised Pauli products ('pp' basis), vectorised row-major in the matrix-unit basis."""
    d = 2 ** n_qubits
    return np.stack([pauli_matrix(l).reshape(-1) / np.sqrt(d) for l in pauli_labels(n_qubits, True)], axis=1)


def _left_right(a, b):
    """Superoperator of rho -> a rho b on row-major vectorised matrices: kron(a, b^T)."""
    return np.kron(a, b.T)


_TERM_CACHE = {}
_MEMBER_TERMS = {}


def term_superops(n_qubits, block_type):
    """[K][D][D] complex Lindblad term superoperators of one coefficient block over the full Pauli basis, in the 'pp'
    basis (K = 4^n - 1 for 'ham' / 'other_diagonal', (4^n - 1)^2 for 'other')."""
    key = (n_qubits, block_type)
    if key in _TERM_CACHE:
        return _TERM_CACHE[key]
    d = 2 ** n_qubits
    eye = np.eye(d, dtype=complex)
    mats = [pauli_matrix(l) for l in pauli_labels(n_qubits)]
    std = []
    if block_type == BLOCK_HAM:
        for p in mats:                                          # -i (P rho - rho P)
            std.append(-1j * (_left_right(p, eye) - _left_right(eye, p)))
    else:
        pairs = [(m, m) for m in mats] if block_type == BLOCK_OTHER_DIAGONAL else [(m, n) for m in mats for n in mats]
        for lm, ln in pairs:                                    # Ln rho Lm^dag - 1/2 (Lm^dag Ln rho + rho Lm^dag Ln)
            lmd = lm.conj().T
            mn = lmd @ ln
            std.append(_left_right(ln, lmd) - 0.5 * (_left_right(mn, eye) + _left_right(eye, mn)))
    b = _pp_transform(n_qubits)
    out = np.array([b.conj().T @ s @ b for s in std])
    _TERM_CACHE[key] = out
    return out


def block_num_params(block_type, mode, n):
    return n * n if block_type == BLOCK_OTHER else n


def block_coefficients(block_type, mode, n, v):
    """block_data.ravel() (complex) of one block from its real parameters v."""
    v = np.asarray(v, float)
    if block_type != BLOCK_OTHER:
        return (v if mode == MODE_ELEMENTS else v * v).astype(complex)
    p = v.reshape(n, n)
    low = np.tril(p, -1) + 1j * np.triu(p, 1).T + np.diag(np.diag(p))      # C_ij = p_ij + i p_ji (i > j), C_ii = p_ii
    if mode == MODE_CHOLESKY:
        return (low @ low.conj().T).reshape(-1)
    herm = low + np.tril(low, -1).conj().T
    return herm.reshape(-1)


def block_coefficient_derivs(block_type, mode, n, v):
    """d(block_data.ravel()) / d(parameters): complex [n_coeffs][n_params] (lindbladcoefficients.py `block_data_jacobian`,
    :178, :235, :348-380, :453-470)."""
    v = np.asarray(v, float)
    if block_type != BLOCK_OTHER:
        return (np.eye(n) if mode == MODE_ELEMENTS else 2.0 * np.diag(v)).astype(complex)
    p = v.reshape(n, n)
    low = np.tril(p, -1) + 1j * np.triu(p, 1).T + np.diag(np.diag(p))
    out = np.zeros((n * n, n * n), complex)
    for a in range(n):
        for b in range(n):
            d = np.zeros((n, n), complex)              # dC / dp_ab: one entry
            if a == b: d[a, a] = 1.0
            elif a > b: d[a, b] = 1.0
            else: d[b, a] = 1.0j
            if mode == MODE_CHOLESKY:
                dc = d @ low.conj().T + low @ d.conj().T
            else:
                dc = d + np.tril(d, -1).conj().T
            out[:, a * n + b] = dc.reshape(-1)
    return out


class LindbladMember:
    """One model member = (static factor) composed with exp(error generator).  `blocks`: [(block type, mode, n)]."""

    def __init__(self, kind, obj, param0, blocks, static, n_qubits):
        self.kind, self.obj, self.param0 = int(kind), int(obj), int(param0)
        self.blocks = [(int(t), int(m), int(n)) for t, m, n in blocks]
        self.static = np.ascontiguousarray(static, np.float64)
        self.n_qubits = int(n_qubits)
        self.D = 4 ** self.n_qubits
        self.n_eff = self.static.shape[0] if self.kind == KIND_POVM else 1
        self.n_params = sum(block_num_params(t, m, n) for t, m, n in self.blocks)
        for t, _, n in self.blocks:
            assert n == self.D - 1, "blocks over the full Pauli basis"
        # L = sum_k Re(c_k) term_re[k] + Im(c_k) term_im[k]; members over the same basis with the same block types SHARE the
        # arrays (three qubits: 2 x 132 MB per table)
        self._term_key = ("pauli", self.n_qubits, tuple(t for t, _, _ in self.blocks))
        if self._term_key not in _MEMBER_TERMS:
            terms = np.concatenate([term_superops(self.n_qubits, t) for t, _, _ in self.blocks], axis=0)
            _MEMBER_TERMS[self._term_key] = (np.ascontiguousarray(terms.real), np.ascontiguousarray(-terms.imag))
        self.term_re, self.term_im = _MEMBER_TERMS[self._term_key]
        self.n_coeffs = self.term_re.shape[0]

    @classmethod
    def from_terms(cls, kind, obj, param0, blocks, static, terms):
        """A member whose Lindblad term superoperators are GIVEN ([K][D][D] complex, the reference's
        `LindbladErrorgen.combined_lindblad_term_superops`) instead of built from Pauli matrices: what the pyGSTi
        adapter passes on (any basis, any subset of basis elements per block)."""
        self = cls.__new__(cls)
        self.kind, self.obj, self.param0 = int(kind), int(obj), int(param0)
        self.blocks = [(int(t), int(m), int(n)) for t, m, n in blocks]
        self.static = np.ascontiguousarray(static, np.float64)
        terms = np.asarray(terms)
        self.D = terms.shape[1]
        self.n_qubits = int(round(np.log(self.D) / np.log(4)))
        self.n_eff = self.static.shape[0] if self.kind == KIND_POVM else 1
        self.n_params = sum(block_num_params(t, m, n) for t, m, n in self.blocks)
        assert terms.shape[0] == self.n_params, "one term per coefficient"     # (every supported block has as many coefficients as parameters)
        self.term_re = np.ascontiguousarray(terms.real, np.float64)
        self.term_im = np.ascontiguousarray(-terms.imag, np.float64)
        self.n_coeffs = terms.shape[0]
        self._term_key = None
        return self

    def coefficients(self, theta):
        out, off = [], 0
        for t, m, n in self.blocks:
            k = block_num_params(t, m, n)
            out.append(block_coefficients(t, m, n, theta[off:off + k]))
            off += k
        return np.concatenate(out)

    def errorgen(self, theta):
        """Dense error generator at this member's parameters theta (lindbladerrorgen.py:699-703)."""
        c = self.coefficients(theta)
        return np.einsum("k,kij->ij", c.real, self.term_re) + np.einsum("k,kij->ij", c.imag, self.term_im)

    def coefficient_derivs(self, theta):
        """dc/dtheta, complex [n_coeffs][n_params] (block diagonal over the member's blocks)."""
        out = np.zeros((self.n_coeffs, self.n_params), complex)
        ko = po = 0
        for t, m, n in self.blocks:
            k = block_num_params(t, m, n)
            d = block_coefficient_derivs(t, m, n, theta[po:po + k])
            out[ko:ko + d.shape[0], po:po + k] = d
            ko += d.shape[0]; po += k
        return out

    def errorgen_deriv(self, theta):
        """d(error generator)/d(parameters): [n_params][D][D] (LindbladErrorgen.deriv_wrt_params, lindbladerrorgen.py:1342)."""
        dc = self.coefficient_derivs(theta)
        return np.einsum("kp,kij->pij", dc.real, self.term_re) + np.einsum("kp,kij->pij", dc.imag, self.term_im)

    def exp_deriv(self, theta):
        """d exp(L)/d(parameters): [n_params][D][D] -- the Frechet derivative of the matrix exponential in the direction
        dL/dtheta_p, read off the exponential of the block matrix [[L, dL], [0, L]] (ExpErrorgenOp.deriv_wrt_params,
        experrorgenop.py:213-260, computes the same thing from a series)."""
        import scipy.linalg
        L = self.errorgen(theta)
        dL = self.errorgen_deriv(theta)
        D = self.D
        out = np.empty_like(dL)
        for p in range(dL.shape[0]):
            blk = np.zeros((2 * D, 2 * D))
            blk[:D, :D] = L; blk[D:, D:] = L; blk[:D, D:] = dL[p]
            out[p] = scipy.linalg.expm(blk)[:D, D:]
        return out

    def dense_deriv(self, theta):
        """d(dense member)/d(parameters) as `deriv_wrt_params()` lays it out: [n_elem][n_params] per object -- a list with
        one array for a gate or a preparation, one per effect for a POVM."""
        dE = self.exp_deriv(theta)                                # [p][i][j]
        if self.kind == KIND_GATE:
            return [np.einsum("pil,lj->ijp", dE, self.static).reshape(self.D * self.D, -1)]
        if self.kind == KIND_RHO:
            return [np.einsum("pil,l->ip", dE, self.static.reshape(-1))]
        return [np.einsum("l,plj->jp", self.static[e], dE) for e in range(self.n_eff)]

    def exp(self, theta):
        import scipy.linalg
        return scipy.linalg.expm(self.errorgen(theta))          # what ExpErrorgenOp._update_rep calls (experrorgenop.py:120)

    def dense(self, theta):
        """Dense member: [D][D] gate, [D] state, [n_eff][D] effect vectors."""
        e = self.exp(theta)
        if self.kind == KIND_GATE:
            return e @ self.static
        if self.kind == KIND_RHO:
            return e @ self.static.reshape(-1)
        return self.static @ e                                   # row i = e_i^T E  (= (E^T e_i)^T)


class LindbladModel:
    """A model whose every member is Lindblad-parameterised: `dense(theta)` gives what `Plan.set_model` takes, and
    `model_sets(theta, params, eps)` the dense model after each `set_parameter_value(p, theta_p + eps)` -- the host
    restatement of the device's model builder."""

    def __init__(self, members, n_params, n_gates, n_rhos, n_effects):
        self.members = list(members)
        self.num_params = int(n_params)
        self.n_gates, self.n_rhos, self.n_effects = int(n_gates), int(n_rhos), int(n_effects)
        self.D = self.members[0].D

    @classmethod
    def from_fixture(cls, lb, n_qubits):
        """From a tests/golden/lindblad_*.npz description (make_golden_lindblad.py)."""
        members = []
        for m in range(int(lb["n_members"])):
            pre = "m%d_" % m
            blocks = [(int(t), int(mode), int(n)) for t, mode, n, _ in lb[pre + "blocks"]]
            members.append(LindbladMember(lb[pre + "kind"], lb[pre + "obj"], lb[pre + "param0"], blocks, lb[pre + "static"], n_qubits))
        n_g = 1 + max(mm.obj for mm in members if mm.kind == KIND_GATE)
        n_r = 1 + max(mm.obj for mm in members if mm.kind == KIND_RHO)
        n_e = sum(mm.n_eff for mm in members if mm.kind == KIND_POVM)
        return cls(members, len(lb["paramvec"]), n_g, n_r, n_e)

    @classmethod
    def from_target(cls, target, gate_labels, effect_labels, parameterization="CPTPLND"):
        """What the reference's `target_model(parameterization)` is: every member of the dense `target` model (an
        ExplicitDenseModel: static superoperators / state / effect vectors) composed with an exponentiated error
        generator whose parameters start at zero.  Parameter order as in the reference: the preparation, the POVM, then
        the operations in the model's own order; `gate_labels` is the plan's gate order (object indices).  CPTPLND: 'ham'/'elements' + 'other'/'cholesky';
        GLND: 'other'/'elements'; H+S: 'ham' + 'other_diagonal'/'cholesky'; H+s: 'other_diagonal'/'elements'."""
        D = target.dim
        nq = int(round(np.log(D) / np.log(4)))
        assert 4 ** nq == D
        n = D - 1
        blocks = {"CPTPLND": [(BLOCK_HAM, MODE_ELEMENTS, n), (BLOCK_OTHER, MODE_CHOLESKY, n)],
                  "GLND": [(BLOCK_HAM, MODE_ELEMENTS, n), (BLOCK_OTHER, MODE_ELEMENTS, n)],
                  "H+S": [(BLOCK_HAM, MODE_ELEMENTS, n), (BLOCK_OTHER_DIAGONAL, MODE_CHOLESKY, n)],
                  "H+s": [(BLOCK_HAM, MODE_ELEMENTS, n), (BLOCK_OTHER_DIAGONAL, MODE_ELEMENTS, n)]}[parameterization]
        per = sum(block_num_params(t, m, k) for t, m, k in blocks)
        members, off = [], 0
        rho = next(iter(target.preps.values()))
        members.append(LindbladMember(KIND_RHO, 0, off, blocks, rho, nq)); off += per
        base = np.array([target.effect_vector(l) for l in effect_labels])
        members.append(LindbladMember(KIND_POVM, 0, off, blocks, base, nq)); off += per
        gate_labels = list(gate_labels)
        for l in target.operations:                  # parameters in the MODEL's operation order; `obj` = the plan's index
            if l in gate_labels:
                members.append(LindbladMember(KIND_GATE, gate_labels.index(l), off, blocks, target.operations[l], nq))
            off += per
        return cls(members, off, len(gate_labels), 1, len(effect_labels))

    def dense(self, theta):
        D = self.D
        gates = np.zeros((self.n_gates, D, D)); rhos = np.zeros((self.n_rhos, D)); effects = np.zeros((self.n_effects, D))
        for m in self.members:
            d = m.dense(np.asarray(theta)[m.param0:m.param0 + m.n_params])
            if m.kind == KIND_GATE: gates[m.obj] = d
            elif m.kind == KIND_RHO: rhos[m.obj] = d
            else: effects[m.obj:m.obj + m.n_eff] = d
        return gates, rhos, effects

    def model_sets(self, theta, params, eps):
        """Dense models after stepping each of `params` by eps (one member changes per step; models/model.py:1198-1310)."""
        theta = np.asarray(theta, float)
        g0, r0, e0 = self.dense(theta)
        G = np.repeat(g0[None], len(params), 0); R = np.repeat(r0[None], len(params), 0); E = np.repeat(e0[None], len(params), 0)
        for k, p in enumerate(params):
            for m in self.members:
                if m.param0 <= p < m.param0 + m.n_params:
                    th = theta[m.param0:m.param0 + m.n_params].copy()
                    th[p - m.param0] = th[p - m.param0] + eps
                    d = m.dense(th)
                    if m.kind == KIND_GATE: G[k, m.obj] = d
                    elif m.kind == KIND_RHO: R[k, m.obj] = d
                    else: E[k, m.obj:m.obj + m.n_eff] = d
        return G, R, E


class LindbladExplicitModel:
    """The host mirror's model object for Lindblad parameterisations -- what the reference's
    `pack.target_model('CPTPLND')` returns, reduced to the interface the forward simulator and the layout consume
    (pygsti_amd/model.py: `operations`, `preps`, `povms`, `dim`, `num_params`, `to_vector` / `from_vector`,
    `effect_labels`, `effect_vector`, `sim`).  Dense members are derived from the parameter vector on demand (host, for
    inspection); the simulator hands `lindblad_description()` and the vector to the device, which builds them itself."""

    def __init__(self, target, parameterization="CPTPLND", theta=None):
        self.target = target
        self.parameterization = parameterization
        self.dim = target.dim
        self._gate_labels = list(target.operations.keys())
        self._effect_labels = list(target.effect_labels)
        self._lm = LindbladModel.from_target(target, self._gate_labels, self._effect_labels, parameterization)
        self._theta = np.zeros(self._lm.num_params) if theta is None else np.array(theta, float)
        assert self._theta.size == self._lm.num_params
        self._sim = None
        self._desc = {}

    @property
    def num_params(self):
        return self._lm.num_params

    def to_vector(self):
        return self._theta.copy()

    def from_vector(self, v, close=False):
        v = np.asarray(v, float)
        assert v.size == self.num_params
        self._theta = v.copy()

    @property
    def sim(self):
        if self._sim is None:
            from .forwardsim import HipMapForwardSimulator
            self.sim = HipMapForwardSimulator()
        return self._sim

    @sim.setter
    def sim(self, simulator):
        self._sim = simulator
        if simulator is not None:
            simulator.model = self

    def _dense(self):
        return self._lm.dense(self._theta)

    @property
    def operations(self):
        from collections import OrderedDict
        G = self._dense()[0]
        return OrderedDict((l, G[i]) for i, l in enumerate(self._gate_labels))

    @property
    def preps(self):
        from collections import OrderedDict
        return OrderedDict((k, self._dense()[1][i]) for i, k in enumerate(self.target.preps.keys()))

    @property
    def povms(self):
        from collections import OrderedDict
        E = self._dense()[2]
        out = OrderedDict()
        for i, full in enumerate(self._effect_labels):
            pk, ok = full.split("_", 1)
            out.setdefault(pk, OrderedDict())[ok] = E[i]
        return out

    @property
    def effect_labels(self):
        return list(self._effect_labels)

    def effect_vector(self, full_label):
        return self._dense()[2][self._effect_labels.index(full_label)]

    def lindblad_description(self, gate_labels, effect_labels):
        """The members in a plan's object order (gate_labels / effect_labels as the layout lists them)."""
        key = (tuple(gate_labels), tuple(effect_labels))
        if key not in self._desc:
            self._desc[key] = LindbladModel.from_target(self.target, list(gate_labels), list(effect_labels), self.parameterization)
        return self._desc[key]

    def copy(self):
        m = LindbladExplicitModel(self.target, self.parameterization, self._theta)
        if self._sim is not None:
            m.sim = self._sim.copy()
        return m
