"""Drop-in for the real pyGSTi (when it is importable): a MapForwardSimulator subclass whose three
per-atom seams run on the MI355X through libgstfwd.

    from pygsti_amd.pygsti_adapter import HipMapForwardSimulator
    results = pygsti.protocols.GateSetTomography(...).run(data, simulator=HipMapForwardSimulator())

Only the seams `DistributableForwardSimulator` subclasses are meant to override are replaced
(pygsti/forwardsims/distforwardsim.py:105-108,148-152,236-241; Map implementations at
mapforwardsim.py:372-391); layout creation (`MapCOPALayout`, which reads `sim.calclib`,
layouts/maplayout.py:79,283), parameter blocking, the MPI grid, serialization and everything above
stay pyGSTi's own.  The precedent for this plug-in style is the reference's own test
test/unit/protocols/test_gst.py:224-302.

The per-atom device plan is built once from the atom's prefix table (`atom.table.contents`,
`elbl_indices_by_expcircuit`, `elindices_by_expcircuit`; layouts/maplayout.py:101-134) and cached on
the atom; each call only re-uploads the dense model arrays (the reference re-marshals the whole
layout on every call, mapforwardsim_calc_densitymx.pyx:170-181).

Scope: `densitymx` models whose members can be densified.  Fully parameterised members (one parameter
per dense element: FullArbitraryOp / FullState / FullPOVMEffect) and full-TP models run in both derivative modes and
for Hessians, finite differences bit-identical to the Map simulator.  Any other parameterisation (CPTPLND, composed
members, shared parameters ...): finite-difference Jacobians by stepping the model on the host exactly as the reference
does and evaluating every perturbed dense model on the device (gst_fill_dprobs_models; to rounding, <= 1e-8 against the
Map simulator), and with `derivative_mode="analytic"` exact Jacobians / Hessians through the members'
`deriv_wrt_params()` / `hessian_wrt_params()` (gst_set_derivs); their FD-of-FD Hessian blocks are composed from
model-set Jacobians exactly as `_mapfill_hprobs_atom` composes them.  There is no silent fallback to the CPU path.
"""
import numpy as np

from . import _lib
from . import lmstep as _lm

try:   # pyGSTi is optional: this module is importable (and useless) without it
    from pygsti.forwardsims.mapforwardsim import MapForwardSimulator as _MapForwardSimulator
    from pygsti.layouts.maplayout import MapCOPALayout as _MapCOPALayout
    from pygsti.objectivefns import objectivefns as _objfns
    from pygsti.tools import slicetools as _slct
    HAVE_PYGSTI = True
except Exception:   # pragma: no cover - exercised only where pyGSTi is absent
    _MapForwardSimulator = object
    _MapCOPALayout = object
    _objfns = None
    _slct = None
    HAVE_PYGSTI = False


def _to_array(slc_or_indices):
    """slicetools.to_array: a slice (or index list) as an index array."""
    if isinstance(slc_or_indices, slice):
        assert slc_or_indices.stop is not None, "open-ended parameter slice"
        return np.arange(slc_or_indices.start or 0, slc_or_indices.stop, slc_or_indices.step or 1)
    return np.asarray(slc_or_indices, dtype=np.int64)


def atom_arrays(model, atom):
    """Dense model arrays in the atom's own index order (op_labels sorted, full_effect_labels a set whose
    iteration order defines effect indices -- captured once)."""
    D = model.dim
    op_labels, rho_labels = list(atom.op_labels), list(atom.rho_labels)
    eff_labels = getattr(atom, "_hip_eff_labels", None)
    if eff_labels is None:
        eff_labels = atom._hip_eff_labels = list(atom.full_effect_labels)

    def dense(lbl, typ, shape):
        return np.ascontiguousarray(np.real(model._circuit_layer_operator(lbl, typ).to_dense("minimal")),
                                    dtype=np.float64).reshape(shape)
    gates = np.array([dense(l, "op", (D, D)) for l in op_labels]).reshape(len(op_labels), D, D)
    rhos = np.array([dense(l, "prep", (D,)) for l in rho_labels]).reshape(len(rho_labels), D)
    effects = np.array([dense(l, "povm", (D,)) for l in eff_labels]).reshape(len(eff_labels), D)
    return gates, rhos, effects


def atom_param_map(model, atom):
    """Parameter -> (kind, object, element) for `full` members; loud failure otherwise."""
    D, nP = model.dim, model.num_params
    kind = -np.ones(nP, np.int32); obj = np.zeros(nP, np.int32); elem = np.zeros(nP, np.int32)
    eff_labels = atom._hip_eff_labels
    for k, labels, typ in ((_lib.KIND_GATE, list(atom.op_labels), "op"), (_lib.KIND_RHO, list(atom.rho_labels), "prep"),
                           (_lib.KIND_EFFECT, eff_labels, "povm")):
        n_el = D * D if k == _lib.KIND_GATE else D
        for oi, lbl in enumerate(labels):
            member = model._circuit_layer_operator(lbl, typ)
            idx = member.gpindices_as_array()
            if len(idx) == 0:
                continue        # static member: no parameters
            dw = None
            if len(idx) != n_el or not np.array_equal((dw := np.asarray(member.deriv_wrt_params())).reshape(n_el, -1),
                                                       np.eye(n_el)):
                raise NotImplementedError(
                    "member %s is not fully parameterised (deriv_wrt_params is not the identity); the device path "
                    "supports one-parameter-per-dense-element members only in this round" % str(lbl))
            kind[idx] = k; obj[idx] = oi; elem[idx] = np.arange(n_el)
    return kind, obj, elem


_ELEMENT_SUBSET_MEMBERS = ("FullArbitraryOp", "FullTPOp", "FullState", "TPState", "FullPOVMEffect")


def atom_tp_map(model, atom):
    """Parameter map of a "full TP"-style model for the FD mode: TPState / FullTPOp parameters are plain dense elements
    (from_vector writes them into the dense array, modelmembers/states/tpstate.py, operations/fulltpop.py), and a
    TPPOVM's last effect is the complement identity - sum(others) (povms/complementeffect.py:72-78).
    Returns (kind, obj, elem, complement) with complement = None or (index, identity[D], others); loud failure for
    any other kind of member."""
    D, nP = model.dim, model.num_params
    kind = -np.ones(nP, np.int32); obj = np.zeros(nP, np.int32); elem = np.zeros(nP, np.int32)
    eff_labels = atom._hip_eff_labels
    eff_members = [model._circuit_layer_operator(l, "povm") for l in eff_labels]
    complement = None
    for k, labels, typ in ((_lib.KIND_GATE, list(atom.op_labels), "op"), (_lib.KIND_RHO, list(atom.rho_labels), "prep"),
                           (_lib.KIND_EFFECT, eff_labels, "povm")):
        n_el = D * D if k == _lib.KIND_GATE else D
        for oi, lbl in enumerate(labels):
            member = model._circuit_layer_operator(lbl, typ)
            idx = member.gpindices_as_array()
            if len(idx) == 0:
                continue
            cls = type(member).__name__
            if cls == "ComplementPOVMEffect":
                if complement is not None:
                    raise NotImplementedError("more than one complement effect in one atom")
                others = []
                for oe in member.other_effects:
                    hits = [j for j, m in enumerate(eff_members) if m is oe]
                    if len(hits) != 1:
                        raise NotImplementedError("a complement effect's terms are not all in this atom")
                    others.append(hits[0])
                ident = np.ascontiguousarray(np.real(member.identity.to_dense()), dtype=np.float64).ravel()
                complement = (oi, ident, np.array(others, np.int32))
                continue
            dm = np.asarray(member.deriv_wrt_params()).reshape(n_el, -1) if cls in _ELEMENT_SUBSET_MEMBERS else None
            rows = None
            if dm is not None and dm.shape[1] == len(idx):
                nz = [np.nonzero(dm[:, j])[0] for j in range(dm.shape[1])]
                if all(len(r) == 1 and dm[r[0], j] == 1.0 for j, r in enumerate(nz)):
                    rows = np.array([r[0] for r in nz])
            if rows is None or len(set(rows.tolist())) != len(rows):
                raise NotImplementedError(
                    "member %s (%s) is not one-parameter-per-dense-element: finite differences over its parameters "
                    "are not on the device" % (str(lbl), cls))
            if (kind[idx] != -1).any():
                raise NotImplementedError("parameter shared between members")
            kind[idx] = k; obj[idx] = oi; elem[idx] = rows
    return kind, obj, elem, complement


def atom_model_sets(model, atom, param_indices, eps):
    """The dense model after each finite-difference step of mapfill_dprobs_atom (mapforwardsim_calc_densitymx.pyx:362-381:
    undo the previous parameter and step the current one with set_parameter_values, models/model.py:1198-1310) -- the
    input of gst_fill_dprobs_models for members whose parameters are not dense elements (CPTPLND, composed, ...).  Only
    the members a parameter belongs to are re-densified; the model is left at its original parameter vector.
    Returns (gates [n, nG, D, D], rhos [n, nR, D], effects [n, nEl, D])."""
    D = model.dim
    base_g, base_r, base_e = atom_arrays(model, atom)
    n = len(param_indices)
    G = np.repeat(base_g[None], n, axis=0); R = np.repeat(base_r[None], n, axis=0); E = np.repeat(base_e[None], n, axis=0)
    members = []
    for arr, labels, typ, shape in ((G, list(atom.op_labels), "op", (D, D)), (R, list(atom.rho_labels), "prep", (D,)),
                                    (E, atom._hip_eff_labels, "povm", (D,))):
        for oi, lbl in enumerate(labels):
            idx = model._circuit_layer_operator(lbl, typ).gpindices_as_array()
            if len(idx):
                members.append((arr, oi, lbl, typ, shape, set(int(q) for q in idx)))
    orig = model.to_vector().copy()
    prev = None
    for c, i in enumerate(int(q) for q in param_indices):
        if prev is None:
            model.set_parameter_value(i, orig[i] + eps)
        else:
            model.set_parameter_values([prev, i], [orig[prev], orig[i] + eps])
        for arr, oi, lbl, typ, shape, owned in members:
            if i in owned:
                arr[c, oi] = np.real(model._circuit_layer_operator(lbl, typ).to_dense("minimal")).reshape(shape)
        prev = i
    if prev is not None:
        model.set_parameter_value(prev, orig[prev])
    return G, R, E


def atom_lindblad(model, atom):
    """Describe a model whose every member is (static factor) x exp(Lindblad error generator) -- what
    `target_model('CPTPLND')` / 'GLND' / 'H+S' build: ComposedOp([static, ExpErrorgenOp(LindbladErrorgen)]),
    ComposedState(static state, ExpErrorgenOp), ComposedPOVM(ExpErrorgenOp, base POVM) -- for gst_set_lindblad, so that
    the DEVICE builds the dense members for the base model and for every finite-difference step (no to_dense() per
    column on the host).  The term superoperators are the reference's own (`combined_lindblad_term_superops`,
    lindbladerrorgen.py:585-590).  Raises NotImplementedError for anything else (the caller then steps the model on the
    host: atom_model_sets)."""
    from . import lindblad as LBM
    D = model.dim
    if D not in (4, 16):
        raise NotImplementedError("Lindblad members are built on the device for one and two qubits")
    BT = {"ham": LBM.BLOCK_HAM, "other_diagonal": LBM.BLOCK_OTHER_DIAGONAL, "other": LBM.BLOCK_OTHER}
    MD = {"elements": LBM.MODE_ELEMENTS, "cholesky": LBM.MODE_CHOLESKY}

    def member_of(kind, obj, member, static_dense, expop):
        if type(expop).__name__ != "ExpErrorgenOp" or type(expop.errorgen).__name__ != "LindbladErrorgen":
            raise NotImplementedError("not an exponentiated Lindblad error generator")
        L = expop.errorgen
        blocks = []
        for b in L.coefficient_blocks:
            if b._block_type not in BT or b._param_mode not in MD:
                raise NotImplementedError("coefficient block %s/%s" % (b._block_type, b._param_mode))
            blocks.append((BT[b._block_type], MD[b._param_mode], len(b._bel_labels)))
        gp = np.asarray(member.gpindices_as_array())
        if len(gp) == 0 or not np.array_equal(gp, np.arange(gp[0], gp[0] + len(gp))) or len(gp) != L.num_params:
            raise NotImplementedError("a member's parameters are not one contiguous slice of the model's")
        terms = L.combined_lindblad_term_superops
        if not isinstance(terms, np.ndarray) or terms.shape[1:] != (D, D):
            raise NotImplementedError("sparse Lindblad term superoperators")
        return LBM.LindbladMember.from_terms(kind, obj, int(gp[0]), blocks, static_dense, terms)

    def static_dense(obj, shape):
        if obj.num_params != 0:
            raise NotImplementedError("the static factor has parameters")
        return np.ascontiguousarray(np.real(obj.to_dense("HilbertSchmidt")), dtype=np.float64).reshape(shape)

    members = []
    for oi, lbl in enumerate(atom.op_labels):
        m = model._circuit_layer_operator(lbl, "op")
        f = getattr(m, "factorops", None)
        if type(m).__name__ != "ComposedOp" or f is None or len(f) != 2:
            raise NotImplementedError("operation %s is not ComposedOp([static, ExpErrorgenOp])" % str(lbl))
        members.append(member_of(LBM.KIND_GATE, oi, m, static_dense(f[0], (D, D)), f[1]))
    for oi, lbl in enumerate(atom.rho_labels):
        m = model._circuit_layer_operator(lbl, "prep")
        if type(m).__name__ != "ComposedState":
            raise NotImplementedError("preparation %s is not a ComposedState" % str(lbl))
        members.append(member_of(LBM.KIND_RHO, oi, m, static_dense(m.state_vec, (D,)), m.error_map))
    eff_labels = atom._hip_eff_labels
    povm_names = {str(l).split("_", 1)[0] for l in eff_labels}
    if len(povm_names) != 1:
        raise NotImplementedError("effects of several POVMs in one atom")
    povm = model.povms[povm_names.pop()]
    if type(povm).__name__ != "ComposedPOVM":
        raise NotImplementedError("the POVM is not a ComposedPOVM")
    base = np.array([static_dense(povm.base_povm[str(l).split("_", 1)[1]], (D,)) for l in eff_labels])
    members.append(member_of(LBM.KIND_POVM, 0, povm, base, povm.error_map))
    if len(eff_labels) > D:
        raise NotImplementedError("a POVM with more than dim effects")
    # members that share an error generator (same gpindices for two gates) are stepped TOGETHER by the reference's
    # set_parameter_value; the device tables give a parameter to exactly one member
    spans = sorted((m.param0, m.param0 + m.n_params) for m in members if m.n_params > 0)
    if any(b[0] < a[1] for a, b in zip(spans, spans[1:])):
        raise NotImplementedError("two members share parameters")
    return LBM.LindbladModel(members, model.num_params, len(atom.op_labels), len(atom.rho_labels), len(eff_labels))


def atom_derivs(model, atom):
    """[(kind, object, gpindices, deriv_wrt_params [n_elem, n])] of every member the atom uses -- the input of
    gst_set_derivs for parameterisations that are not one-parameter-per-element (TP, CPTP, ...), exactly what
    MatrixForwardSimulator._doperation consumes (matrixforwardsim.py:126-190)."""
    D = model.dim
    out = []
    for k, labels, typ in ((_lib.KIND_GATE, list(atom.op_labels), "op"), (_lib.KIND_RHO, list(atom.rho_labels), "prep"),
                           (_lib.KIND_EFFECT, atom._hip_eff_labels, "povm")):
        n_el = D * D if k == _lib.KIND_GATE else D
        for oi, lbl in enumerate(labels):
            member = model._circuit_layer_operator(lbl, typ)
            idx = member.gpindices_as_array()
            if len(idx) == 0:
                continue
            dm = np.ascontiguousarray(np.real(member.deriv_wrt_params()), dtype=np.float64).reshape(n_el, len(idx))
            out.append((k, oi, np.asarray(idx, np.int64), dm))
    return out


def atom_second_derivs(model, atom):
    """hessian_wrt_params [n_elem, n, n] (or None for members linear in their parameters) of the objects atom_derivs
    lists, in its order -- what MatrixForwardSimulator._hoperation consumes (matrixforwardsim.py:192-224)."""
    D = model.dim
    out = []
    for k, labels, typ in ((_lib.KIND_GATE, list(atom.op_labels), "op"), (_lib.KIND_RHO, list(atom.rho_labels), "prep"),
                           (_lib.KIND_EFFECT, atom._hip_eff_labels, "povm")):
        n_el = D * D if k == _lib.KIND_GATE else D
        for lbl in labels:
            member = model._circuit_layer_operator(lbl, typ)
            n = len(member.gpindices_as_array())
            if n == 0:
                continue
            out.append(np.ascontiguousarray(np.real(member.hessian_wrt_params()), dtype=np.float64).reshape(n_el, n, n)
                       if member.has_nonzero_hessian() else None)
    return out


def _element_selection(member, n_el):
    """int64 [n_el]: the model parameter that IS each dense element of `member` (-1: not a parameter), for members whose
    deriv_wrt_params is a 0/1 selection with at most one parameter per element and one element per parameter (FullArbitraryOp,
    FullTPOp, FullState, TPState, FullPOVMEffect, static members ...); NotImplementedError otherwise."""
    idx = np.asarray(member.gpindices_as_array(), np.int64)
    sel = -np.ones(n_el, np.int64)
    if len(idx) == 0:
        return sel
    dm = np.asarray(member.deriv_wrt_params())
    if np.iscomplexobj(dm):
        dm = np.real(dm)
    dm = dm.reshape(n_el, len(idx))
    rows, cols = np.nonzero(dm)
    if len(rows) != len(idx) or len(set(cols.tolist())) != len(idx) or len(set(rows.tolist())) != len(rows) or not np.all(dm[rows, cols] == 1.0):
        raise NotImplementedError("member %s is not one-parameter-per-dense-element" % type(member).__name__)
    sel[rows] = idx[cols]
    return sel


def atom_composite(model, atom):
    """The layer structure of an implicit model for gst_set_composite: every operation label of the atom resolved into an
    ordered product of EMBEDDED LEAVES -- `ComposedOp.factorops` in order of application, `EmbeddedOp.embedded_op` on
    `target_labels` (modelmembers/operations/composedop.py, embeddedop.py), nested as deep as they come -- with leaves
    identified by object identity (so a leaf that stands behind several layers, `independent_gates=False`, is ONE leaf and
    its parameters are shared, exactly as set_parameter_value moves them, models/model.py:1198-1310).  Leaves must be
    one-parameter-per-element or static (FullArbitraryOp, FullTPOp, StaticArbitraryOp ...).
    Returns (composite.CompositeModel, (kind, obj, elem) parameter map of the SPAM elements); NotImplementedError for
    anything else (Lindblad leaves, leaves on more than three qubits, non-qubit registers ...)."""
    from .composite import CompositeModel
    D, nP = model.dim, model.num_params
    if D not in (4, 16, 64):
        raise NotImplementedError("composite layers are built on the device for one to three qubits")
    nq = {4: 1, 16: 2, 64: 3}[D]
    leaves, leaf_dims, leaf_params, leaf_static, leaf_general = {}, [], [], [], []
    derivs_ok = [True]

    def leaf_of(op):
        key = id(op)
        if key not in leaves:
            d = int(op.dim)
            if d not in (4, 16, 64) or d > D:
                raise NotImplementedError("a leaf operation of dimension %d" % d)
            try:
                leaf_params.append(_element_selection(op, d * d))           # ELEMENT leaf: its elements are the parameters
                leaf_general.append(None)
            except NotImplementedError:
                # GENERAL leaf (an exponentiated Lindblad generator ...): the host keeps densifying / differentiating this
                # small operation with the reference's own code; embedding and products happen on the device
                idx = np.asarray(op.gpindices_as_array(), np.int64)
                if len(idx) == 0 or len(set(idx.tolist())) != len(idx) or not hasattr(op, "deriv_wrt_params"):
                    raise NotImplementedError("leaf %s" % type(op).__name__)
                try:                    # the base class HAS the attribute and raises when called (linearop.py): ask once
                    op.deriv_wrt_params()
                except NotImplementedError:
                    derivs_ok[0] = False    # finite differences over this leaf's host-stepped values still work
                leaf_params.append(-np.ones(d * d, np.int64))
                leaf_general.append(np.sort(idx))
            leaf_static.append(np.ascontiguousarray(np.real(op.to_dense("HilbertSchmidt")), dtype=np.float64).ravel())
            leaf_dims.append(d)
            leaves[key] = (len(leaf_dims) - 1, op)         # (the op is kept alive: ids are only unique among live objects)
        return leaves[key][0]

    def expand(op, qubits):
        """-> [(leaf, target positions)] in order of application; `qubits`: register position of each qubit of op's space"""
        name = type(op).__name__
        if name == "ComposedOp":
            out = []
            for f in op.factorops:
                out.extend(expand(f, qubits))
            return out
        if name == "EmbeddedOp":
            lbls = list(op.state_space.sole_tensor_product_block_labels)
            if len(lbls) != len(qubits):
                raise NotImplementedError("an embedded operation over a non-qubit space")
            inner = tuple(qubits[lbls.index(t)] for t in op.target_labels)
            return expand(op.embedded_op, inner)
        if getattr(op, "factorops", None) is not None or hasattr(op, "embedded_op"):
            raise NotImplementedError("layer member %s" % name)
        if int(op.dim) != 4 ** len(qubits):
            raise NotImplementedError("a leaf that does not fill its target qubits")
        return [(leaf_of(op), tuple(qubits))]

    gate_factors = []
    for lbl in atom.op_labels:
        op = model._circuit_layer_operator(lbl, "op")
        lbls = list(op.state_space.sole_tensor_product_block_labels)
        if len(lbls) != nq:
            raise NotImplementedError("a register that is not %d qubits" % nq)
        gate_factors.append(expand(op, tuple(range(nq))))
    if not any(len(f) > 1 or (len(f) == 1 and leaf_dims[f[0][0]] < D) for f in gate_factors):
        raise NotImplementedError("every layer is one dense operation: not an implicit model")
    kind = -np.ones(nP, np.int32); obj = np.zeros(nP, np.int32); elem = np.zeros(nP, np.int32)
    for k, labels, typ in ((_lib.KIND_RHO, list(atom.rho_labels), "prep"), (_lib.KIND_EFFECT, atom._hip_eff_labels, "povm")):
        for oi, lbl in enumerate(labels):
            sel = _element_selection(model._circuit_layer_operator(lbl, typ), D)
            hit = np.nonzero(sel >= 0)[0]
            if (kind[sel[hit]] != -1).any():
                raise NotImplementedError("a SPAM parameter shared between members")
            kind[sel[hit]] = k; obj[sel[hit]] = oi; elem[sel[hit]] = hit
    leaf_owned = np.concatenate(leaf_params + [g for g in leaf_general if g is not None]) if leaf_params else np.zeros(0, np.int64)
    if (kind[leaf_owned[leaf_owned >= 0]] != -1).any():
        raise NotImplementedError("a parameter shared between a layer operation and a SPAM member")
    cm = CompositeModel(D, nP, leaf_dims, leaf_params, leaf_static, gate_factors, leaf_general)
    cm._leaf_ops = [op for _, op in sorted(leaves.values(), key=lambda t: t[0])]
    cm.derivs_ok = derivs_ok[0]          # every general leaf can give deriv_wrt_params: the exact route is available
    return cm, (kind, obj, elem)


def composite_general_data(model, cm, want_derivs, fd_eps=None):
    """What gst_set_composite_values / gst_set_composite_general need from the GENERAL leaves of `cm` at the model's current
    parameters, computed by the reference's own members on the (small) leaves only: their dense elements; with want_derivs
    their deriv_wrt_params() in ascending parameter order; with fd_eps their elements after each of their parameters'
    finite-difference steps -- the model is stepped exactly as mapfill_dprobs_atom steps it
    (mapforwardsim_calc_densitymx.pyx:362-381) and left at its original vector.  -> (values {l: [d*d]}, derivs {l: [d*d, np]}
    or None, fd {l: [np, d*d]} or None)"""
    vals, derivs, fds = {}, ({} if want_derivs else None), ({} if fd_eps is not None else None)

    def dense(op):
        return np.ascontiguousarray(np.real(op.to_dense("HilbertSchmidt")), dtype=np.float64).ravel()
    for l in cm.general_leaves:
        op, d = cm._leaf_ops[l], cm.leaf_dims[l]
        vals[l] = dense(op)
        if want_derivs:
            idx = np.asarray(op.gpindices_as_array(), np.int64)
            derivs[l] = np.ascontiguousarray(np.real(op.deriv_wrt_params()), dtype=np.float64).reshape(d * d, len(idx))[:, np.argsort(idx)]
    if fd_eps is not None and cm.general_leaves:
        orig = model.to_vector().copy()
        prev = None
        for l in cm.general_leaves:
            fds[l] = np.empty((len(cm.leaf_general[l]), cm.leaf_dims[l] ** 2))
            for c, q in enumerate(int(x) for x in cm.leaf_general[l]):
                if prev is None:
                    model.set_parameter_value(q, orig[q] + fd_eps)
                else:
                    model.set_parameter_values([prev, q], [orig[prev], orig[q] + fd_eps])
                fds[l][c] = dense(cm._leaf_ops[l])
                prev = q
        if prev is not None:
            model.set_parameter_value(prev, orig[prev])
    return vals, derivs, fds


def atom_plan(model, atom, device=-1, target_tasks=0):
    """The libgstfwd plan of a `_MapCOPALayoutAtom`, built once and cached on the atom.

    The adapter's layouts are created with max_cache_size = 0 (see `create_layout`), so every row of the atom's prefix
    table is a whole circuit -- (iDest, None, (rho, gate, gate, ...), None) -- and the plan is built from the RAW CIRCUITS
    (gst_plan_create_from_circuits: the library compiles its own prefix trie anyway).  The labels are turned into integers
    by C-level iteration (`np.fromiter(map(dict.__getitem__, chain.from_iterable(...)))`): the per-label Python loop this
    replaces was 31.9 M list appends for the 2Q L<=1024 design (reference: convert_maplayout,
    mapforwardsim_calc_densitymx.pyx:55-77).  A table that does carry cached prefixes (someone else's layout) still goes
    through gst_plan_create_from_table."""
    import itertools
    plan = getattr(atom, "_hip_plan", None)
    if plan is not None:
        return plan
    op_lookup = {l: i for i, l in enumerate(atom.op_labels)}
    rho_lookup = {l: i for i, l in enumerate(atom.rho_labels)}
    contents = atom.table.contents
    R = len(contents)
    # effect CSR: elbl_indices_by_expcircuit / elindices_by_expcircuit are dicts (or lists) of per-circuit integer lists
    el_src, ed_src = atom.elbl_indices_by_expcircuit, atom.elindices_by_expcircuit
    el_lists = [el_src[i] for i in range(R)]
    ed_lists = [ed_src[i] for i in range(R)]
    eff_ptr = np.zeros(R + 1, np.int64)
    np.cumsum(np.fromiter(map(len, el_lists), np.int64, count=R), out=eff_ptr[1:])
    n_el = int(eff_ptr[-1])
    el = np.fromiter(itertools.chain.from_iterable(el_lists), np.int32, count=n_el)
    ed = np.fromiter(itertools.chain.from_iterable(ed_lists), np.int32, count=n_el)
    atom._hip_eff_labels = list(atom.full_effect_labels)
    nG, nR, nEl = len(atom.op_labels), len(atom.rho_labels), len(atom._hip_eff_labels)
    if all(row[1] is None for row in contents):
        # cache-free table: row k IS circuit iDest_k
        dest = np.fromiter((row[0] for row in contents), np.int64, count=R)
        rems = [row[2] for row in contents]
        lens = np.fromiter(map(len, rems), np.int64, count=R) - 1
        assert (lens >= 0).all(), "every expanded circuit starts with a state preparation"
        order = np.argsort(dest, kind="stable")
        if not np.array_equal(dest[order], np.arange(R)):
            raise ValueError("prefix table rows do not enumerate the expanded circuits")
        circ_rho_rows = np.fromiter((rho_lookup[r[0]] for r in rems), np.int32, count=R)
        gates_rows = np.fromiter(map(op_lookup.__getitem__, itertools.chain.from_iterable(r[1:] for r in rems)),
                                 np.int32, count=int(lens.sum()))
        if np.array_equal(order, np.arange(R)):
            circ_rho, circ_len, circ_gates = circ_rho_rows, lens, gates_rows
        else:                                       # rows in evaluation order, circuits in index order: permute
            ptr_rows = np.zeros(R + 1, np.int64); np.cumsum(lens, out=ptr_rows[1:])
            circ_rho, circ_len = circ_rho_rows[order], lens[order]
            circ_gates = np.concatenate([gates_rows[ptr_rows[k]:ptr_rows[k + 1]] for k in order]) if R else gates_rows
        circ_ptr = np.zeros(R + 1, np.int64); np.cumsum(circ_len, out=circ_ptr[1:])
        plan = _lib.Plan.from_circuits(model.dim, nG, nR, nEl, atom.num_elements, circ_rho, circ_ptr, circ_gates, eff_ptr, el, ed,
                                       device=device, target_tasks=target_tasks)
    else:
        t_dest = np.empty(R, np.int32); t_start = np.empty(R, np.int32); t_cache = np.empty(R, np.int32)
        t_rho = -np.ones(R, np.int32); row_ptr = np.zeros(R + 1, np.int64); gidx = []
        for k, (iDest, iStart, remainder, iCache) in enumerate(contents):   # convert_maplayout, pyx:55-77
            t_dest[k] = iDest
            t_start[k] = -1 if iStart is None else iStart
            t_cache[k] = -1 if iCache is None else iCache
            rem = list(remainder)
            if iStart is None:
                t_rho[k] = rho_lookup[rem[0]]; rem = rem[1:]
            gidx.extend(op_lookup[g] for g in rem)
            row_ptr[k + 1] = len(gidx)
        plan = _lib.Plan.from_table(model.dim, nG, nR, nEl, atom.num_elements, atom.cache_size, t_dest, t_start, t_cache, t_rho,
                                    row_ptr, np.array(gidx, np.int32), eff_ptr, el, ed, device=device, target_tasks=target_tasks)
    atom._hip_plan = plan
    return plan


class AtomFillLogic:
    """Everything the adapter does per layout atom -- choosing how a model's parameters reach the device ("elements",
    "tp-elements", "lindblad", "models", "derivs") and the three `_bulk_fill_*_atom` seams -- written against the DUCK
    TYPES of a pyGSTi model and layout atom only (`model._circuit_layer_operator`, `to_vector`, `set_parameter_value(s)`,
    `from_vector`, `num_params`, `dim`; `atom.op_labels`, `rho_labels`, `full_effect_labels`, `table.contents`,
    `elbl_indices_by_expcircuit`, `elindices_by_expcircuit`, `num_elements`, `cache_size`).  HipMapForwardSimulator mixes
    it into pyGSTi's MapForwardSimulator; tests/test_gpu_adapter_modes.py drives it on the GPU box -- where pyGSTi does
    not exist -- with stand-in models and atoms built from the committed fixtures."""
    # (expects on `self`: model, derivative_eps, hessian_eps, derivative_mode, lindblad_on_device, _hip_device -- no
    #  defaults for those here: class attributes of a mixin would shadow pyGSTi's `model` property)
    composite_on_device = True      # implicit models: dense layers / FD model sets / derivative matrices built on the device

    def _lindblad_description(self, layout_atom):
        """The atom's Lindblad member description (None when the model is not of that family), cached per model."""
        if getattr(layout_atom, "_hip_lb_model", None) is not self.model:
            try:
                layout_atom._hip_lb = atom_lindblad(self.model, layout_atom)
            except NotImplementedError:
                layout_atom._hip_lb = None
            layout_atom._hip_lb_model = self.model
        return layout_atom._hip_lb

    def _effective_mode(self, layout_atom):
        """'fd' or 'analytic' for this atom.  derivative_mode='auto' (the adapter's default): finite differences -- the
        Map simulator's own values, bit for bit -- wherever parameters are dense elements (`full`, full TP) or the model is
        stepped on the host; EXACT derivatives for Lindblad-parameterised models (CPTPLND, GLND, H+S) whose members the
        device builds itself: there the device's exponential (scaled Taylor) and the reference's (scipy's Pade approximant)
        differ in the last bit of the perturbed member, which a finite-difference quotient amplifies by (occurrences of
        the member in the circuit) / eps -- <= 1e-8 against the Map simulator for circuits of depth <= 16 only (7e-8 at depth
        1,030, 1.2e-7 on the 1Q L<=128 design; tests/test_gpu_lindblad.py::test_cptplnd_models_at_depth_error_profile) -- while the exact route agrees
        with the Matrix simulator to <= 1e-8 at every depth (observed 1e-12).  Models of any OTHER parameterisation
        (implicit models: embedded / composed layer operations sharing their factors' parameters, ...) also take the
        exact route: the chain rule over the members' deriv_wrt_params runs on the device, while their finite differences
        would have the host re-densify the model for every column ("models" mode, still there under 'fd').  'fd' and
        'analytic' force one mode."""
        if self.derivative_mode != "auto":
            return self.derivative_mode
        key = getattr(layout_atom, "_hip_auto_key", (None, None))
        if key[0] is not self.model or key[1] != (bool(self.lindblad_on_device), bool(self.composite_on_device)):
            mode = "fd"
            try:
                atom_param_map(self.model, layout_atom)
            except NotImplementedError:
                try:
                    atom_tp_map(self.model, layout_atom)
                except NotImplementedError:
                    # neither element maps nor TP element subsets: Lindblad members (built on the device) and every other
                    # parameterisation whose members give deriv_wrt_params take exact derivatives; a member that cannot
                    # give them (deriv_wrt_params unimplemented) sends the atom back to finite differences over
                    # host-stepped dense model sets -- what 'fd' did for every such model before 'auto' existed
                    mode = "analytic"
                    cmp_desc = self._composite_description(layout_atom) if self.composite_on_device else None
                    if cmp_desc is not None and not getattr(cmp_desc[0], "derivs_ok", True):
                        mode = "fd"         # a general leaf without deriv_wrt_params: FD over its host-stepped values
                    elif not (self.lindblad_on_device and self._lindblad_description(layout_atom) is not None) and cmp_desc is None:
                        try:
                            atom_derivs(self.model, layout_atom)
                        except (NotImplementedError, AttributeError):
                            mode = "fd"
            layout_atom._hip_auto_mode, layout_atom._hip_auto_key = mode, (self.model, (bool(self.lindblad_on_device), bool(self.composite_on_device)))
        return layout_atom._hip_auto_mode

    def _prepare(self, layout_atom, derivatives=False, hessian=False):
        plan = atom_plan(self.model, layout_atom, self._hip_device)
        dmode = self._effective_mode(layout_atom) if derivatives else None
        # Lindblad-parameterised atoms with `lindblad_on_device`: the DEVICE builds the dense members from the parameter
        # vector for EVERY fill of this model -- probabilities included, from the first call on -- so that what
        # bulk_fill_probs returns never depends on whether a derivative was requested before (the device's scaled-Taylor
        # exponential and the host's Pade approximant differ in the last bit, which a line search would see).  Skips the
        # host to_dense() of every member as well.  Only exact HESSIAN requests leave this route (they need the host's
        # deriv / hessian matrices, `derivs` mode below); the next fill re-enters it.
        # (forced derivative_mode="fd" at depth: the library offers FD over device-built members only where it meets the 1e-8
        #  bar -- GST_LINDBLAD_FD_MAX_DEPTH -- so such a request walks host-stepped dense model sets below: the reference's own
        #  perturbed members, as before `lindblad_on_device` existed)
        lb_fd_deep = derivatives and not hessian and dmode != "analytic" and plan.stats()["max_depth"] > _lib.LINDBLAD_FD_MAX_DEPTH
        if self.lindblad_on_device and not hessian and not lb_fd_deep and self._lindblad_description(layout_atom) is not None:
            if getattr(plan, "_hip_mode", None) != "lindblad":
                plan.set_derivs(self.model.num_params, [])
                plan.set_complement_effect(-1)
                plan.set_lindblad(layout_atom._hip_lb)
                plan._hip_mode = "lindblad"
            plan.set_lindblad_params(self.model.to_vector())
            return plan
        # Implicit models (layers = products of embedded leaves, parameters shared between layers) with `composite_on_device`:
        # the device builds the dense layers from the leaves' few elements -- for the base model, for every finite-difference
        # column and for the exact route's derivative matrices -- so nothing of size D x D is densified on the host, per model
        # update or per column.  Exact HESSIAN requests leave this route (a layer is bilinear in two leaves).
        if self.composite_on_device and not hessian and self._composite_description(layout_atom) is not None:
            cm, spam_map = layout_atom._hip_cmp
            if getattr(plan, "_hip_mode", None) != "composite":
                self._leave_lindblad(plan, layout_atom)
                plan.set_derivs(self.model.num_params, [])
                plan.set_complement_effect(-1)
                plan.set_param_map(*spam_map)
                plan.set_composite(cm)
                plan._hip_mode = "composite"
            D = self.model.dim

            def vec(lbl, typ):
                return np.ascontiguousarray(np.real(self.model._circuit_layer_operator(lbl, typ).to_dense("minimal")), dtype=np.float64).reshape(D)
            rhos = np.array([vec(l, "prep") for l in layout_atom.rho_labels]).reshape(len(layout_atom.rho_labels), D)
            effects = np.array([vec(l, "povm") for l in layout_atom._hip_eff_labels]).reshape(len(layout_atom._hip_eff_labels), D)
            if cm.general_leaves:
                # general leaves (exponentiated Lindblad generators ...): the reference's own members densify / differentiate
                # the SMALL leaves; finite differences need every leaf parameter's stepped leaf, exact derivatives the leaves'
                # deriv_wrt_params -- per model update, never per layer and never at the register's dimension
                want_fd = derivatives and dmode != "analytic"
                vals, dvs, fds = composite_general_data(self.model, cm, derivatives and dmode == "analytic",
                                                        self.derivative_eps if want_fd else None)
                plan.set_composite_values(cm.values(self.model.to_vector(), vals), rhos, effects)
                if derivatives:
                    plan.set_composite_general(*cm.pack_general(dvs, fds), fd_eps=self.derivative_eps)
            else:
                plan.set_composite_values(cm.values(self.model.to_vector()), rhos, effects)
            return plan
        self._leave_composite(plan)
        plan.set_model(*atom_arrays(self.model, layout_atom))
        if not derivatives:
            return plan
        self._element_map(layout_atom)
        if layout_atom._hip_pmap is not None:
            if plan.n_params != self.model.num_params or getattr(plan, "_hip_mode", None) != "elements":
                self._leave_lindblad(plan, layout_atom)
                plan.set_derivs(self.model.num_params, [])
                plan.set_complement_effect(-1)
                plan.set_param_map(*layout_atom._hip_pmap)
                plan._hip_mode = "elements"
        elif dmode != "analytic":
            # TP-style models: element-subset members + a complement effect, bit-identical finite differences
            if getattr(layout_atom, "_hip_tpmap_model", None) is not self.model:
                try:
                    layout_atom._hip_tpmap = atom_tp_map(self.model, layout_atom)
                except NotImplementedError:
                    # CPTPLND, composed members, shared parameters ...: the general path -- the model is stepped on the
                    # host as the reference does and the device evaluates every perturbed dense model
                    layout_atom._hip_tpmap = None
                layout_atom._hip_tpmap_model = self.model
            if layout_atom._hip_tpmap is None:
                # Lindblad-parameterised members (CPTPLND, GLND, H+S) under an FD-of-FD Hessian request (everything else
                # returned at the top): the device builds them from the parameter vector, for the base model and every step
                if self._lindblad_description(layout_atom) is not None and self.lindblad_on_device and not lb_fd_deep:
                    if getattr(plan, "_hip_mode", None) != "lindblad":
                        plan.set_derivs(self.model.num_params, [])
                        plan.set_complement_effect(-1)
                        plan.set_lindblad(layout_atom._hip_lb)
                        plan._hip_mode = "lindblad"
                    plan.set_lindblad_params(self.model.to_vector())      # (replaces the host-densified set_model above)
                    return plan
                self._leave_lindblad(plan, layout_atom)
                plan._hip_mode = "models"
                return plan
            k, o, e, comp = layout_atom._hip_tpmap
            if plan.n_params != self.model.num_params or getattr(plan, "_hip_mode", None) != "tp-elements":
                self._leave_lindblad(plan, layout_atom)
                plan.set_derivs(self.model.num_params, [])
                plan.set_param_map(k, o, e)
                if comp is not None:
                    plan.set_complement_effect(comp[0], comp[1], comp[2])
                plan._hip_mode = "tp-elements"
        else:
            # exact derivatives of a general parameterisation (Lindblad-parameterised models whose member derivatives the
            # device computes itself -- Frechet derivative of the exponential -- returned at the top; their exact Hessians
            # come here): the members' deriv_wrt_params() from the host, re-sent every call (they move with the parameters)
            self._leave_lindblad(plan, layout_atom)
            plan.set_derivs(self.model.num_params, atom_derivs(self.model, layout_atom))
            plan._hip_mode = "derivs"
        return plan

    def _element_map(self, layout_atom):
        """atom_param_map of the current model, cached on the atom (None: not one-parameter-per-element)"""
        if getattr(layout_atom, "_hip_pmap_model", None) is not self.model:
            try:
                layout_atom._hip_pmap = atom_param_map(self.model, layout_atom)     # one parameter per dense element
            except NotImplementedError:
                layout_atom._hip_pmap = None                                        # TP, CPTP, ...: chain rule
            layout_atom._hip_pmap_model = self.model
        return layout_atom._hip_pmap

    def _composite_description(self, layout_atom):
        """(CompositeModel, SPAM parameter map) of an implicit model's atom, or None: explicit models whose members are
        dense elements keep the bit-exact element-map route; anything atom_composite cannot resolve keeps the chain-rule /
        host-stepped routes.  Cached per model."""
        if getattr(layout_atom, "_hip_cmp_model", None) is not self.model:
            layout_atom._hip_cmp = None
            if getattr(layout_atom, "_hip_eff_labels", None) is None:
                layout_atom._hip_eff_labels = list(layout_atom.full_effect_labels)
            if self._element_map(layout_atom) is None:
                try:
                    layout_atom._hip_cmp = atom_composite(self.model, layout_atom)
                except (NotImplementedError, AttributeError):
                    layout_atom._hip_cmp = None
            layout_atom._hip_cmp_model = self.model
        return layout_atom._hip_cmp

    @staticmethod
    def _leave_composite(plan):
        if getattr(plan, "_hip_mode", None) == "composite":
            plan.set_composite(None)
            plan._hip_mode = None

    def _leave_lindblad(self, plan, layout_atom):
        """A plan that was in the device-built Lindblad mode and now serves another mode: drop the description (later
        GST_DERIV_FD fills would still take the Lindblad route, with a stale theta) and re-install the host's dense model
        (set_lindblad_params had replaced it)."""
        if getattr(plan, "_hip_mode", None) == "lindblad":
            plan.set_lindblad(None)
            plan.set_model(*atom_arrays(self.model, layout_atom))
            plan._hip_mode = None

    def _bulk_fill_probs_atom(self, array_to_fill, layout_atom, resource_alloc):
        plan = self._prepare(layout_atom)
        if array_to_fill.flags.c_contiguous:
            plan.fill_probs(array_to_fill)
        else:
            array_to_fill[:] = plan.fill_probs()

    def _bulk_fill_dprobs_atom(self, array_to_fill, dest_param_slice, layout_atom, param_slice, resource_alloc):
        plan = self._prepare(layout_atom, derivatives=True)
        _lm.pin_destination(array_to_fill)        # pyGSTi's own 'ep' array: page-locked on its first real fill
        nP = self.model.num_params
        pidx = np.arange(nP) if param_slice is None else _to_array(param_slice)
        didx = None if dest_param_slice is None else _to_array(dest_param_slice)
        if getattr(plan, "_hip_mode", None) == "models":
            G, R, E = atom_model_sets(self.model, layout_atom, pidx, self.derivative_eps)
            plan.set_model(*atom_arrays(self.model, layout_atom))
            plan.fill_dprobs_models(G, R, E, array_to_fill, didx, self.derivative_eps)
            return
        mode = _lib.DERIV_ANALYTIC if self._effective_mode(layout_atom) == "analytic" else _lib.DERIV_FD
        plan.fill_dprobs(array_to_fill, pidx, didx, self.derivative_eps, mode=mode)

    def _bulk_fill_hprobs_atom(self, array_to_fill, dest_param_slice1, dest_param_slice2, layout_atom,
                               param_slice1, param_slice2, resource_alloc):
        plan = self._prepare(layout_atom, derivatives=True, hessian=True)
        hmode = getattr(plan, "_hip_mode", None)
        if hmode == "derivs":
            # exact Hessians of a general parameterisation: members that are not linear in their parameters also send
            # their hessian_wrt_params (the objects and their order are atom_derivs')
            plan.set_second_derivs(atom_second_derivs(self.model, layout_atom))
        elif hmode == "models":
            # FD of FD over a general parameterisation, composed as _mapfill_hprobs_atom does (mapforwardsim.py:420-436):
            # the model is moved to theta + eps e_i on the host, its FD Jacobian over block 2 comes from the device
            # (model sets), and (dprobs2 - dprobs) / eps is the reference's own numpy line
            eps = self.hessian_eps
            nP = self.model.num_params
            i1 = np.arange(nP) if param_slice1 is None else _to_array(param_slice1)
            i2 = np.arange(nP) if param_slice2 is None else _to_array(param_slice2)
            d1 = np.arange(len(i1)) if dest_param_slice1 is None else _to_array(dest_param_slice1)
            d2 = np.arange(len(i2)) if dest_param_slice2 is None else _to_array(dest_param_slice2)
            orig = self.model.to_vector().copy()

            def dprobs_here():
                G, R, E = atom_model_sets(self.model, layout_atom, i2, eps)
                plan.set_model(*atom_arrays(self.model, layout_atom))
                return plan.fill_dprobs_models(G, R, E, eps=eps)
            dprobs = dprobs_here()
            try:
                for i, i_final in sorted(zip((int(q) for q in i1), (int(q) for q in d1))):
                    vec = orig.copy(); vec[i] += eps
                    self.model.from_vector(vec, close=True)
                    array_to_fill[:, i_final, d2] = (dprobs_here() - dprobs) / eps
            finally:
                self.model.from_vector(orig)
            return
        elif hmode == "lindblad":
            # FD of FD composed as _mapfill_hprobs_atom does (mapforwardsim.py:420-436), every Jacobian from the device's
            # own model builder: theta + eps e_i is just another parameter vector
            eps = self.hessian_eps
            nP = self.model.num_params
            i1 = np.arange(nP) if param_slice1 is None else _to_array(param_slice1)
            i2 = np.arange(nP) if param_slice2 is None else _to_array(param_slice2)
            d1 = np.arange(len(i1)) if dest_param_slice1 is None else _to_array(dest_param_slice1)
            d2 = np.arange(len(i2)) if dest_param_slice2 is None else _to_array(dest_param_slice2)
            orig = self.model.to_vector().copy()
            dprobs = plan.fill_dprobs(param_idx=i2, eps=eps)
            try:
                for i, i_final in sorted(zip((int(q) for q in i1), (int(q) for q in d1))):
                    vec = orig.copy(); vec[i] += eps
                    plan.set_lindblad_params(vec)
                    array_to_fill[:, i_final, d2] = (plan.fill_dprobs(param_idx=i2, eps=eps) - dprobs) / eps
            finally:
                plan.set_lindblad_params(orig)
            return
        elif hmode != "elements" and not (hmode == "tp-elements" and self._effective_mode(layout_atom) == "fd"):
            raise NotImplementedError("Hessians on the device: fully parameterised models (FD or exact), full-TP models "
                                      "(FD of FD as the Map simulator computes them, or exact), any parameterisation "
                                      "with derivative_mode='analytic' (exact)")
        nP = self.model.num_params
        i1 = np.arange(nP) if param_slice1 is None else _to_array(param_slice1)
        i2 = np.arange(nP) if param_slice2 is None else _to_array(param_slice2)
        d1 = None if dest_param_slice1 is None else _to_array(dest_param_slice1)
        d2 = None if dest_param_slice2 is None else _to_array(dest_param_slice2)
        # derivative_mode="analytic": exact second derivatives (MatrixForwardSimulator's values) 
        # (D = 4, 16, 64); otherwise the Map simulator's FD-of-FD, bit for bit
        mode = _lib.DERIV_ANALYTIC if (self._effective_mode(layout_atom) == "analytic") else _lib.DERIV_FD
        if array_to_fill.flags.c_contiguous:
            plan.fill_hprobs(array_to_fill, i1, i2, d1, d2, self.hessian_eps, mode)
        else:
            tmp = np.ascontiguousarray(array_to_fill)
            plan.fill_hprobs(tmp, i1, i2, d1, d2, self.hessian_eps, mode)
            array_to_fill[...] = tmp


    def device_scaled_jacobian(self, layout, counts, total_counts, objective="logl", min_prob_clip=1e-4, radius=1e-4,
                               prob_clip_interval=None, lsvec_to_fill=None, pr_array_to_fill=None):
        """The device half of `TimeIndependentMDCObjectiveFunction.dlsvec` (objectivefns.py:4595-4665) for every atom of
        `layout`: probabilities and Jacobian (this simulator's derivative mode for the model's parameterisation), the
        objective's element-wise maps (lsvec; the dlsvec row factor w = (0.5 / lsvec) * dterms), and J_s^T J_s of the scaled
        Jacobian J_s = diag(w) J with the factors applied on the fly (gst_fill_normal_eqs_dev: J_s is never stored, the
        device block stays dprobs) -- nothing of size (nE, nP) leaves HBM.  `counts` / `total_counts`: per-element
        arrays in layout order (the objective's `.counts` / `.total_counts`); objective = 'chi2' | 'logl'.
        Returns (parts, jtj): parts = [(plan, device pointer of the atom's [nE_atom][nP] dprobs block, element slice,
        device pointer of its row factors)] (what lmstep.DeviceJacobian wraps), jtj = the (nP, nP) sum over this process's atoms.  The optional host vectors
        receive lsvec and the (clipped) probabilities, as the reference's dlsvec leaves them in `objective.obj` /
        `objective.probs`; `self.last_objective_sum` holds sum(terms).  Not available for models that must be stepped on
        the host ("models" mode: no device-resident Jacobian)."""
        nP = self.model.num_params
        counts = np.asarray(counts, np.float64); total_counts = np.asarray(total_counts, np.float64)
        jtj = np.zeros((nP, nP)); part = np.empty((nP, nP))
        parts = []
        total = 0.0
        pidx = np.arange(nP, dtype=np.int64)
        for atom in layout.atoms:
            plan = self._prepare(atom, derivatives=True)
            if getattr(plan, "_hip_mode", None) == "models":
                raise NotImplementedError("the fused LM step needs a device-resident Jacobian; this model is stepped on the host")
            es = getattr(atom, "element_slice", slice(0, atom.num_elements))
            nE = atom.num_elements
            mode = _lib.DERIV_ANALYTIC if self._effective_mode(atom) == "analytic" else _lib.DERIV_FD
            sizes = (nE * nP * 8, nE * 8, nE * 8, nE * 8, nE * 8, nE * 8, nP * nP * 8)
            d_J, d_pr, d_c, d_N, d_ls, d_w, d_jtj = [plan.workspace("lsq%d" % k, nb) for k, nb in enumerate(sizes)]
            plan.memcpy_h2d(d_c, np.ascontiguousarray(counts[es])); plan.memcpy_h2d(d_N, np.ascontiguousarray(total_counts[es]))
            plan.fill_dprobs_dev(d_J, nP, pidx, None, self.derivative_eps, d_pr, mode)
            total += plan.objective_rows_dev(objective, d_pr, d_c, d_N, nE, d_ls, d_w, None, min_prob_clip, radius,
                                             prob_clip_interval)
            plan.fill_normal_eqs_dev(d_J, nE, nP, nP, d_w, d_jtj=d_jtj)
            plan.memcpy_d2h(part, d_jtj)
            jtj += part
            if lsvec_to_fill is not None:
                self._d2h_rows(plan, lsvec_to_fill, es, d_ls)
            if pr_array_to_fill is not None:
                self._d2h_rows(plan, pr_array_to_fill, es, d_pr)
            parts.append((plan, d_J, es, d_w))
        self.last_objective_sum = total
        return parts, jtj

    @staticmethod
    def _d2h_rows(plan, host_vec, es, d_ptr):
        view = host_vec[es]
        if isinstance(view, np.ndarray) and view.flags.c_contiguous:
            plan.memcpy_d2h(view, d_ptr)
        else:
            tmp = np.empty(es.stop - es.start); plan.memcpy_d2h(tmp, d_ptr); host_vec[es] = tmp

    def bulk_fill_lsq_step(self, jtj, jtf, layout, counts, total_counts, objective="logl", min_prob_clip=1e-4, radius=1e-4,
                           prob_clip_interval=None, lsvec_to_fill=None, pr_array_to_fill=None):
        """What one Levenberg-Marquardt iteration of a GST fit needs from the data, computed without the Jacobian leaving
        the GPU: `device_scaled_jacobian`, then J_s^T lsvec (optimize/simplerlm.py:677-678).
        jtj (nP, nP) and jtf (nP,) are summed over this process's atoms (ranks all-reduce, as `layout.fill_jtj` does);
        returns sum(terms) of them.  This is the direct call; from the reference's own optimizer the same step is reached
        through `hip_objfn_builders()` (objective.dlsvec -> lmstep.DeviceJacobian -> layout.fill_jtj / fill_jtf)."""
        nP = self.model.num_params
        n_el = sum(a.num_elements for a in layout.atoms)
        ls = np.empty(n_el) if lsvec_to_fill is None else lsvec_to_fill
        parts, part_jtj = self.device_scaled_jacobian(layout, counts, total_counts, objective, min_prob_clip, radius,
                                                      prob_clip_interval, ls, pr_array_to_fill)
        dj = _lm.DeviceJacobian(parts, part_jtj, n_el, nP)
        jtj[...] = dj.jtj()
        jtf[...] = dj.jtf(ls)
        return self.last_objective_sum


class HipMapForwardSimulator(AtomFillLogic, _MapForwardSimulator):
    """MapForwardSimulator whose atom fills run on the GPU.  `derivative_mode="auto"` (default): the Map simulator's
    finite differences, bit for bit, for `full` / full-TP models and exact derivatives for Lindblad-parameterised models
    (see AtomFillLogic._effective_mode); `"fd"`: finite differences for every model; `"analytic"`: exact first
    derivatives (what MatrixForwardSimulator returns, to <= 1e-8), several times faster, and exact Hessian blocks."""

    def __init__(self, model=None, max_cache_size=None, num_atoms=None, processor_grid=None, param_blk_sizes=None,
                 derivative_eps=1e-7, hessian_eps=1e-5, device=-1, derivative_mode="auto", lindblad_on_device=True,
                 composite_on_device=True):
        if not HAVE_PYGSTI:
            raise ImportError("pygsti is not importable; use pygsti_amd.forwardsim.HipMapForwardSimulator instead")
        if derivative_mode not in ("auto", "fd", "analytic"):
            raise ValueError("derivative_mode must be 'auto', 'fd' or 'analytic'")
        super().__init__(model, max_cache_size, num_atoms, processor_grid, param_blk_sizes, derivative_eps, hessian_eps)
        self._hip_device = device
        self.derivative_mode = derivative_mode
        # Lindblad-parameterised models (CPTPLND, GLND, H+S): True = the device builds the dense members from the
        # parameter vector for every finite-difference step (gst_set_lindblad); False = the model is stepped on the host
        # and the device evaluates the dense sets (gst_fill_dprobs_models; the validation path)
        self.lindblad_on_device = bool(lindblad_on_device)
        # Implicit models (embedded / composed layer operations over shared leaves): True = the device builds the dense
        # layers, every finite-difference column's model and the exact route's derivative matrices from the leaves
        # (gst_set_composite); False = the members' to_dense() / deriv_wrt_params() on the host (the validation path)
        self.composite_on_device = bool(composite_on_device)

    def copy(self, keep_model_attached=True):
        out = HipMapForwardSimulator(self.model if keep_model_attached else None, self._max_cache_size, self._num_atoms,
                                     self._processor_grid, self._pblk_sizes, self.derivative_eps, self.hessian_eps,
                                     self._hip_device, self.derivative_mode, self.lindblad_on_device, self.composite_on_device)
        return out

    def _to_nice_serialization(self):
        """MapForwardSimulator's state (max_cache_size and the two step sizes, mapforwardsim.py:174-181) plus this class's
        own options: pyGSTi writes the simulator into every GST checkpoint and results directory
        (protocols/gst.py:1497-1504), and a run resumed from one must not silently change its derivative mode."""
        state = super()._to_nice_serialization()
        state.update({"hip_derivative_mode": self.derivative_mode, "hip_device": int(self._hip_device),
                      "hip_lindblad_on_device": bool(self.lindblad_on_device),
                      "hip_composite_on_device": bool(self.composite_on_device)})
        return state

    @classmethod
    def _from_nice_serialization(cls, state):
        # (like the base class, mapforwardsim.py:183-188: the parent model and the processor distribution are not stored)
        return cls(None, state["max_cache_size"], derivative_eps=state.get("derivative_epsilon", 1e-7),
                   hessian_eps=state.get("hessian_epsilon", 1e-5), device=state.get("hip_device", -1),
                   derivative_mode=state.get("hip_derivative_mode", "auto"),
                   lindblad_on_device=state.get("hip_lindblad_on_device", True),
                   composite_on_device=state.get("hip_composite_on_device", True))

    def create_layout(self, circuits, dataset=None, resource_alloc=None, array_types=('E',), derivative_dimensions=None,
                      verbosity=0, layout_creation_circuit_cache=None, **kwargs):
        """pyGSTi's own MapCOPALayout (element indexing, atoms, parameter blocks, MPI grid -- everything upstream relies
        on), built WITHOUT the prefix-cache assessment pass: the library compiles its own prefix trie from the full gate
        strings (gst_plan_create_from_table re-derives them), so the table's caching choices are never used, and with
        max_cache_size = 0 `PrefixTable.__init__` skips `_cache_hits` (layouts/prefixtable.py:76-84), the O(rows x cache)
        tuple-comparison pass that is half of layout creation (2Q L<=1024 lite: 14.6 -> 7.7 s; what remains is pyGSTi's
        per-circuit completion / POVM separation, models/model.py:1600-1775, which callers amortise with
        `layout_creation_circuit_cache`).  Results are unchanged: every state is still rho followed by the circuit's
        gates applied left to right.

        The layout comes back as `HipMapCOPALayout`: the same object with three methods of the LM path overridden
        (lmstep.LayoutNormalEquations: `fill_jtj` / `fill_jtf` accept a device-resident Jacobian, `allocate_local_array`
        marks pyGSTi's 'ep' arrays for page-locking on their first real fill)."""
        keep = self._max_cache_size
        self._max_cache_size = 0
        try:
            layout = super().create_layout(circuits, dataset, resource_alloc, array_types, derivative_dimensions, verbosity,
                                           layout_creation_circuit_cache, **kwargs)
        finally:
            self._max_cache_size = keep
        if type(layout) is _MapCOPALayout:
            layout.__class__ = HipMapCOPALayout      # (no new state: a method-only subclass)
        return layout


class HipMapCOPALayout(_lm.LayoutNormalEquations, _MapCOPALayout):
    """pyGSTi's MapCOPALayout whose normal-equation products accept the device-resident Jacobian of `dlsvec`
    (layouts/distlayout.py:1220-1359 otherwise) -- see lmstep.LayoutNormalEquations."""


if HAVE_PYGSTI:
    class HipChi2Function(_lm.DeviceLMStepLogic, _objfns.Chi2Function):
        """`Chi2Function` (objectivefns.py:4972-4983) whose `dlsvec` leaves the scaled Jacobian on the device."""

    class HipPoissonPicDeltaLogLFunction(_lm.DeviceLMStepLogic, _objfns.PoissonPicDeltaLogLFunction):
        """`PoissonPicDeltaLogLFunction` (objectivefns.py:5056-5068) whose `dlsvec` leaves the scaled Jacobian on the device."""


def hip_objfn_builders(objective="logl", always_perform_mle=False, only_perform_mle=False):
    """`GSTObjFnBuilders.create_from(objective)` (protocols/gst.py:790-834) with the two objective classes above in place
    of the stock ones -- same names, descriptions, regularisation and penalties, so estimates and reports read the same:

        proto = pygsti.protocols.GateSetTomography(model, objfn_builders=hip_objfn_builders(), ...)
        results = proto.run(data, simulator=HipMapForwardSimulator())

    Every LM iteration of that run then goes objective.dlsvec -> device (fill + maps + J^T J) -> layout.fill_jtj /
    fill_jtf, and no (nE, nP) host array is ever filled.  The builders serialize by class path
    (objectivefns.py:287-307), so checkpoints and results round-trip wherever pygsti_amd is importable."""
    if not HAVE_PYGSTI:
        raise ImportError("pygsti is not importable")
    from pygsti.protocols.gst import GSTObjFnBuilders
    stock = GSTObjFnBuilders.create_from(objective, always_perform_mle=always_perform_mle, only_perform_mle=only_perform_mle)
    swap = {_objfns.Chi2Function: HipChi2Function, _objfns.PoissonPicDeltaLogLFunction: HipPoissonPicDeltaLogLFunction}

    def convert(b):
        cls = swap.get(b.cls_to_build)
        if cls is None:
            return b
        return _objfns.ObjectiveFunctionBuilder(cls, b.name, b.description, b.regularization, b.penalties, **b.additional_args)
    return GSTObjFnBuilders([convert(b) for b in stock.iteration_builders], [convert(b) for b in stock.final_builders])
