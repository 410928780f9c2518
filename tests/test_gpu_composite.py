"""Implicit models' layer operations built ON THE DEVICE (gst_set_composite): the reference's own
implicit 3-qubit model (`create_crosstalk_free_model`: EmbeddedOp / ComposedOp layers over three SHARED leaves, 864
parameters; fixtures `3q_crosstalk_free` + `3q_crosstalk_free_composite`) with zero host densification --

  * the dense layers the device builds are pyGSTi's to_dense() of every layer (<= 1e-15; the host restatement is exact),
    probabilities <= 1e-10 against the Map simulator's;
  * GST_DERIV_FD columns (the device moves the parameter's leaf elements, rebuilds every layer that contains the leaf --
    shared parameters move several layers at once, as set_parameter_value does -- and walks the complete dense model):
    <= 1e-8 against the Map simulator's finite differences;
  * GST_DERIV_ANALYTIC columns (device-built derivative matrices by the product rule + chain rule): <= 1e-8 against the
    Matrix simulator, and equal to the host-supplied deriv_wrt_params route (gst_set_derivs) to rounding;
  * the drop-in's per-atom logic selects this route for such a model (stand-in members shaped like pyGSTi's), never asks a
    layer for to_dense() or deriv_wrt_params(), and leaves it for exact Hessians."""
import numpy as np
import pytest

from conftest import load_fixture, matrix_rows_by_circuit
from test_composite import composite_from_fixture

pytestmark = pytest.mark.gpu


def _plan(fx):
    from pygsti_amd import _lib
    return _lib.Plan.from_table(fx['D'], len(fx['gates']), 1, len(fx['effects']), fx['nE'], fx['cache_size'], fx['t_dest'], fx['t_start'],
                                fx['t_cache'], fx['t_rho'], fx['row_ptr'], fx['gate_idx'], fx['eff_ptr'], fx['eff_label'], fx['eff_dest'])


def test_device_built_layers_fd_and_exact_columns():
    from pygsti_amd import _lib
    from oracle import oracle as O
    fx, cf = load_fixture("3q_crosstalk_free"), load_fixture("3q_crosstalk_free_composite")
    cm = composite_from_fixture(fx, cf)
    pl = _plan(fx)
    pl.set_param_map(cf["cmp_spam_kind"], cf["cmp_spam_obj"], cf["cmp_spam_elem"])
    pl.set_composite(cm)
    pl.set_composite_values(cm.values(fx["paramvec"]), fx["rhos"], fx["effects"])
    G, R, E = pl.get_model()
    assert np.abs(G - fx["gates"]).max() < 1e-15 and np.array_equal(R, fx["rhos"]) and np.array_equal(E, fx["effects"])
    p = pl.fill_probs()
    assert np.abs(p - fx["probs"]).max() < 1e-10
    cols = fx["dprobs_cols"]
    eps = float(fx["derivative_eps"])
    Jf = pl.fill_dprobs(param_idx=cols, eps=eps)
    assert np.abs(Jf - fx["dprobs_map"]).max() < 1e-8, np.abs(Jf - fx["dprobs_map"]).max()
    assert np.abs(Jf).max() > 0.1
    # the same columns through host-built model sets (the numpy restatement of the builder): same walk, same sets to rounding
    pm = (cf["cmp_spam_kind"], cf["cmp_spam_obj"], cf["cmp_spam_elem"])
    Gs, Rs, Es = cm.model_sets(cm.values(fx["paramvec"]), fx["rhos"], fx["effects"], pm, cols, eps)
    Jm = pl.fill_dprobs_models(Gs, Rs, Es, eps=eps)
    assert np.abs(Jf - Jm).max() < 1e-8
    # into a column window of a wider device array, scattered destinations
    nE = int(fx["nE"])
    ld = len(cols) + 5
    d = pl.device_malloc(nE * ld * 8)
    pl.memcpy_h2d(d, np.full(nE * ld, np.nan))
    dest = np.arange(len(cols))[::-1] + 3
    pl.fill_dprobs_dev(d, ld, cols, dest, eps, None, _lib.DERIV_FD); pl.sync()
    W = pl.memcpy_d2h(np.empty((nE, ld)), d)
    assert np.array_equal(W[:, dest], Jf) and np.isnan(W[:, :3]).all() and np.isnan(W[:, -2:]).all()
    pl.device_free(d)
    # exact columns
    Ja = pl.fill_dprobs(param_idx=cols, mode=_lib.DERIV_ANALYTIC)
    rows = matrix_rows_by_circuit(fx)
    assert np.array_equal(fx["matrix_cols"], cols)
    assert np.abs(Ja - fx["matrix_by_circuit_dprobs"][rows]).max() < 1e-8, np.abs(Ja - fx["matrix_by_circuit_dprobs"][rows]).max()
    assert np.abs(Ja - Jf).max() < 1e-4                    # (forward differences: eps times the second derivative)
    # ALL 864 columns: exact vs FD, and exact vs the host-supplied deriv_wrt_params route
    allc = np.arange(int(fx["nP"]))
    Jall = pl.fill_dprobs(param_idx=allc, mode=_lib.DERIV_ANALYTIC)
    Jfd = pl.fill_dprobs(param_idx=allc, eps=eps)
    assert np.abs(Jall - Jfd).max() < 1e-4 and np.array_equal(Jall[:, cols], Ja)
    ref = _plan(fx)
    ref.set_model(fx["gates"], fx["rhos"], fx["effects"])
    ref.set_derivs(int(fx["nP"]), O.derivs_from_fixture(fx))
    Jh = ref.fill_dprobs(param_idx=allc, mode=_lib.DERIV_ANALYTIC)
    assert np.abs(Jall - Jh).max() < 1e-12, np.abs(Jall - Jh).max()
    # a new parameter vector: only the leaves' 288 numbers (and the SPAM vectors) travel
    theta2 = fx["paramvec"] + 0.01 * np.random.default_rng(5).standard_normal(int(fx["nP"]))
    R2 = theta2[:64].reshape(1, 64)
    E2 = fx["effects"].copy()
    for q in range(64, 576):
        E2[cf["cmp_spam_obj"][q], cf["cmp_spam_elem"][q]] = theta2[q]
    pl.set_composite_values(cm.values(theta2), R2, E2)
    G2 = pl.get_model()[0]
    assert np.abs(G2 - cm.dense_gates(cm.values(theta2))).max() < 1e-15
    ref.set_model(cm.dense_gates(cm.values(theta2)), R2, E2)
    assert np.abs(pl.fill_probs() - ref.fill_probs()).max() < 1e-14
    # cleared: the plan is an ordinary dense plan again
    pl.set_composite(None)
    pl.set_model(fx["gates"], fx["rhos"], fx["effects"])
    assert np.abs(pl.fill_probs() - fx["probs"]).max() < 1e-10


# ---- the drop-in's per-atom logic over stand-in members shaped like pyGSTi's ------------------------------------------------
class _Space:
    def __init__(self, labels):
        self.sole_tensor_product_block_labels = tuple(labels)


class FullArbitraryOp:
    def __init__(self, dense, gp):
        self._dense, self._gp = np.asarray(dense, float), np.asarray(gp, np.int64)
        self.dim = self._dense.shape[0]
        self.densified = 0

    def to_dense(self, on_space="minimal"):
        self.densified += 1
        return self._dense

    def gpindices_as_array(self):
        return self._gp

    def deriv_wrt_params(self, wrt_filter=None):
        return np.eye(self._dense.size)


class EmbeddedOp:
    def __init__(self, leaf, targets, nq=3):
        self.embedded_op, self.target_labels, self.state_space = leaf, tuple(targets), _Space(range(nq))

    def gpindices_as_array(self):
        return self.embedded_op.gpindices_as_array()

    def to_dense(self, on_space="minimal"):
        raise AssertionError("a LAYER was densified on the host")

    def deriv_wrt_params(self, wrt_filter=None):
        raise AssertionError("a LAYER's deriv_wrt_params was requested")


class ComposedOp(EmbeddedOp):
    def __init__(self, factors, nq=3):
        self.factorops, self.state_space = list(factors), _Space(range(nq))

    def gpindices_as_array(self):
        return np.unique(np.concatenate([f.gpindices_as_array() for f in self.factorops]))


class _Vec:
    def __init__(self, dense, gp):
        self._dense, self._gp = np.asarray(dense, float), np.asarray(gp, np.int64)

    def to_dense(self, on_space="minimal"):
        return self._dense

    def gpindices_as_array(self):
        return self._gp

    def deriv_wrt_params(self, wrt_filter=None):
        return np.eye(self._dense.size)


class _ImplicitModel:
    def __init__(self, fx, cf, vec):
        self.dim, self.num_params, self._vec = 64, int(fx["nP"]), np.asarray(vec, float)
        cm = composite_from_fixture(fx, cf)
        vals = cm.values(vec)
        leaves = [FullArbitraryOp(vals[cm.leaf_off[l]:cm.leaf_off[l + 1]].reshape(d, d), cm.leaf_params[l]) for l, d in enumerate(cm.leaf_dims)]
        self.leaves = leaves
        self._members = {}
        for lbl, fs in zip(fx["op_labels"], cm.gate_factors):
            embs = [EmbeddedOp(leaves[l], tg) for l, tg in fs]
            self._members[("op", str(lbl))] = embs[0] if len(embs) == 1 else ComposedOp(embs)
        self._members[("prep", str(fx["rho_labels"][0]))] = _Vec(vec[:64], np.arange(64))
        for e, lbl in enumerate(fx["eff_labels"]):
            gp = np.nonzero((cf["cmp_spam_kind"] == 2) & (cf["cmp_spam_obj"] == e))[0]
            gp = gp[np.argsort(cf["cmp_spam_elem"][gp])]
            self._members[("povm", str(lbl))] = _Vec(vec[gp], gp)

    def _circuit_layer_operator(self, lbl, typ):
        return self._members[(typ, str(lbl))]

    def to_vector(self):
        return self._vec.copy()


def test_adapter_logic_takes_the_device_built_route_for_implicit_models():
    import test_gpu_adapter_modes as M
    from pygsti_amd import _lib
    fx, cf = load_fixture("3q_crosstalk_free"), load_fixture("3q_crosstalk_free_composite")
    atom = M._Atom(fx)
    model = _ImplicitModel(fx, cf, fx["paramvec"])
    sim = M._Sim(model, "auto")
    nE, nP = int(fx["nE"]), int(fx["nP"])
    p = np.empty(nE); sim._bulk_fill_probs_atom(p, atom, None)
    assert atom._hip_plan._hip_mode == "composite"
    assert np.abs(p - fx["probs"]).max() < 1e-10
    cols = fx["dprobs_cols"]
    J = np.full((nE, nP), np.nan)
    sim._bulk_fill_dprobs_atom(J, cols, atom, cols, None)              # `auto` -> exact derivatives for implicit models
    assert sim._effective_mode(atom) == "analytic"
    rows = matrix_rows_by_circuit(fx)
    assert np.abs(J[:, cols] - fx["matrix_by_circuit_dprobs"][rows]).max() < 1e-8
    fd = M._Sim(model, "fd")
    atom2 = M._Atom(fx)
    Jf = np.empty((nE, len(cols)))
    fd._bulk_fill_dprobs_atom(Jf, None, atom2, cols, None)             # finite differences: device-built model sets
    assert atom2._hip_plan._hip_mode == "composite"
    assert np.abs(Jf - fx["dprobs_map"]).max() < 1e-8
    # (the stand-in layers raise on to_dense / deriv_wrt_params: nothing of size D x D was densified on the host)
    # the fused LM step runs on this route as well
    jtj = np.empty((nP, nP)); jtf = np.empty(nP)
    rng = np.random.default_rng(3)
    N = np.full(nE, 1000.0); cnt = rng.multinomial(1000, np.full(8, 0.125), size=nE // 8).astype(float).ravel()
    lay = type("L", (), {"atoms": [atom]})()
    total = sim.bulk_fill_lsq_step(jtj, jtf, lay, cnt, N, "chi2", 1e-4, 1e-4, None)
    assert np.isfinite(total) and np.isfinite(jtj).all() and np.abs(jtj - jtj.T).max() <= 1e-9 * np.abs(jtj).max()
    # switched off: the host routes (these stand-ins then fail loudly, which is the point of the switch's default)
    off = M._Sim(model, "auto"); off.composite_on_device = False
    with pytest.raises(AssertionError, match="densified on the host"):
        off._bulk_fill_probs_atom(np.empty(nE), M._Atom(fx), None)


def test_general_leaves_cptplnd_implicit_model_on_the_device():
    """`3q_crosstalk_free_CPTPLND`: D = 64 layers over CPTPLND leaves (static target x exp(Lindblad generator) on one / two
    qubits; 840 parameters, the leaves shared between layers).  The host hands over what the reference's own members give for
    the SMALL leaves (values, deriv_wrt_params, values after each parameter step: stored in the fixture); embedding, layer
    products, the per-column dense models and the layers' derivative matrices are the device's.
      layers <= 1e-15 from to_dense(); probabilities <= 1e-10 (Map); FD columns <= 1e-8 (Map: the perturbed leaves ARE the
      reference's, so no exponential-algorithm mismatch is amplified); exact columns <= 1e-8 (Matrix)."""
    from pygsti_amd import _lib
    from test_composite import general_data_from_fixture
    fx = load_fixture("3q_crosstalk_free_CPTPLND")
    cm = composite_from_fixture(fx, fx)
    dvs, fds = general_data_from_fixture(cm, fx)
    eps = float(fx["derivative_eps"])
    pl = _plan(fx)
    pl.set_param_map(fx["cmp_spam_kind"], fx["cmp_spam_obj"], fx["cmp_spam_elem"])
    pl.set_composite(cm)
    pl.set_composite_values(fx["cmp_leaf_values"], fx["rhos"], fx["effects"])
    G = pl.get_model()[0]
    assert np.abs(G - fx["gates"]).max() < 1e-15
    assert np.abs(pl.fill_probs() - fx["probs"]).max() < 1e-10
    cols = fx["dprobs_cols"]
    with pytest.raises(_lib.GstError):                      # general leaves: derivative fills need the leaves' data first
        pl.fill_dprobs(param_idx=cols, eps=eps)
    pl.set_composite_general(*cm.pack_general(dvs, fds), fd_eps=eps)
    Jf = pl.fill_dprobs(param_idx=cols, eps=eps)
    assert np.abs(Jf - fx["dprobs_map"]).max() < 1e-8, np.abs(Jf - fx["dprobs_map"]).max()
    with pytest.raises(_lib.GstError):                      # ... and FD fills the step size the leaves were stepped with
        pl.fill_dprobs(param_idx=cols, eps=2 * eps)
    Ja = pl.fill_dprobs(param_idx=cols, mode=_lib.DERIV_ANALYTIC)
    rows = matrix_rows_by_circuit(fx)
    assert np.abs(Ja - fx["matrix_by_circuit_dprobs"][rows]).max() < 1e-8, np.abs(Ja - fx["matrix_by_circuit_dprobs"][rows]).max()
    # every column: exact vs FD (forward differences: eps x second derivative), every parameter of every leaf reached
    allc = np.arange(int(fx["nP"]))
    Jall = pl.fill_dprobs(param_idx=allc, mode=_lib.DERIV_ANALYTIC)
    Jfd = pl.fill_dprobs(param_idx=allc, eps=eps)
    assert np.abs(Jall - Jfd).max() < 1e-4 and np.array_equal(Jall[:, cols], Ja)
    used = np.zeros(int(fx["nP"]), bool)
    for g in range(len(cm.gate_factors)):
        used[cm.gate_params(g)] = True
    assert (np.abs(Jall[:, used]).max(axis=0) > 0).all()
    # the restatement's layer derivatives are what the device built (compare through the host-supplied route)
    ref = _plan(fx)
    ref.set_model(fx["gates"], fx["rhos"], fx["effects"])
    objs = [(0, g, qs, dm) for g, (qs, dm) in enumerate(cm.gate_derivs(fx["cmp_leaf_values"], dvs))]
    spam = [(1, 0, np.arange(64), np.eye(64))] + [(2, e, np.nonzero((fx["cmp_spam_kind"] == 2) & (fx["cmp_spam_obj"] == e))[0][np.argsort(
        fx["cmp_spam_elem"][(fx["cmp_spam_kind"] == 2) & (fx["cmp_spam_obj"] == e)])], np.eye(64)) for e in range(8)]
    ref.set_derivs(int(fx["nP"]), spam + objs)
    Jh = ref.fill_dprobs(param_idx=allc, mode=_lib.DERIV_ANALYTIC)
    assert np.abs(Jall - Jh).max() < 1e-12, np.abs(Jall - Jh).max()


@pytest.mark.parametrize("seed", range(6))
def test_random_composite_structures_against_the_restatement(seed):
    """Randomised: registers of 1-3 qubits, leaves of every dimension that fits (element leaves with static elements and
    parameters shared BETWEEN leaves, general leaves), targets in arbitrary order, layers of 0-4 factors (repeated leaves
    included) -- the device's layers, finite-difference model sets (through the probabilities they give) and exact columns
    against the numpy restatement and the host-supplied derivative route."""
    from pygsti_amd import _lib
    from pygsti_amd.composite import CompositeModel
    rng = np.random.default_rng(1000 + seed)
    nq = int(rng.integers(1, 4)); D = 4 ** nq
    n_leaves = int(rng.integers(2, 6))
    dims = [4 ** int(rng.integers(1, nq + 1)) for _ in range(n_leaves)]
    nG, nEl = int(rng.integers(2, 6)), int(rng.integers(2, 5))
    n_spam = D + nEl * D
    leaf_params, leaf_static, leaf_general, nxt = [], [], [], n_spam
    pool = []                                             # parameters that may be re-used by a later element leaf
    for d in dims:
        if rng.random() < 0.35:                           # general leaf
            n = int(rng.integers(1, 7))
            leaf_params.append(-np.ones(d * d, np.int64)); leaf_general.append(np.arange(nxt, nxt + n)); nxt += n
        else:
            p = -np.ones(d * d, np.int64)
            free = rng.random(d * d) < 0.8
            for e in np.nonzero(free)[0]:
                if pool and rng.random() < 0.1:
                    p[e] = pool[int(rng.integers(len(pool)))]
                else:
                    p[e] = nxt; pool.append(nxt); nxt += 1
            leaf_params.append(p); leaf_general.append(None)
        leaf_static.append(np.eye(d).ravel() + 0.1 * rng.standard_normal(d * d))
    nP = nxt + 2                                          # two parameters nothing uses
    gate_factors = []
    for g in range(nG):
        fs = []
        for _ in range(int(rng.integers(0 if g else 1, 5))):
            l = int(rng.integers(n_leaves))
            k = {4: 1, 16: 2, 64: 3}[dims[l]]
            fs.append((l, tuple(int(t) for t in rng.permutation(nq)[:k])))
        gate_factors.append(fs)
    cm = CompositeModel(D, nP, dims, leaf_params, leaf_static, gate_factors, leaf_general)
    theta = 0.1 * rng.standard_normal(nP)
    theta[:D] = 0; theta[0] = 1.0
    eps = 1e-7
    gen_vals = {l: np.eye(dims[l]).ravel() + 0.1 * rng.standard_normal(dims[l] ** 2) for l in cm.general_leaves}
    gen_dv = {l: rng.standard_normal((dims[l] ** 2, len(cm.leaf_general[l]))) for l in cm.general_leaves}
    gen_fd = {l: gen_vals[l][None] + eps * gen_dv[l].T + 1e-9 * rng.standard_normal((len(cm.leaf_general[l]), dims[l] ** 2)) for l in cm.general_leaves}
    v = cm.values(theta, gen_vals)
    # SPAM: dense elements, parameters 0 .. n_spam - 1
    rhos = theta[:D].reshape(1, D).copy(); effects = 0.2 * rng.standard_normal((nEl, D)); effects[:, 0] += 0.5
    theta[D:n_spam] = effects.ravel()
    kind = -np.ones(nP, np.int32); obj = np.zeros(nP, np.int32); elem = np.zeros(nP, np.int32)
    kind[:D] = 1; elem[:D] = np.arange(D)
    kind[D:n_spam] = 2; obj[D:n_spam] = np.repeat(np.arange(nEl), D); elem[D:n_spam] = np.tile(np.arange(D), nEl)
    # a handful of circuits over the layers
    n_c = 12
    lens = rng.integers(0, 9, n_c)
    ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    cg = rng.integers(0, nG, int(ptr[-1])).astype(np.int32)
    nE = n_c * nEl
    pl = _lib.Plan.from_circuits(D, nG, 1, nEl, nE, np.zeros(n_c, np.int32), ptr, cg, np.arange(n_c + 1, dtype=np.int64) * nEl,
                                 np.tile(np.arange(nEl, dtype=np.int32), n_c), np.arange(nE, dtype=np.int32))
    pl.set_param_map(kind, obj, elem)
    pl.set_composite(cm)
    pl.set_composite_values(v, rhos, effects)
    if cm.general_leaves:
        pl.set_composite_general(*cm.pack_general(gen_dv, gen_fd), fd_eps=eps)
    G = cm.dense_gates(v)
    scale = max(1.0, np.abs(G).max())
    assert np.abs(pl.get_model()[0] - G).max() <= 1e-14 * scale
    ref = _lib.Plan.from_circuits(D, nG, 1, nEl, nE, np.zeros(n_c, np.int32), ptr, cg, np.arange(n_c + 1, dtype=np.int64) * nEl,
                                  np.tile(np.arange(nEl, dtype=np.int32), n_c), np.arange(nE, dtype=np.int32))
    ref.set_model(G, rhos, effects)
    p0 = ref.fill_probs()
    pscale = max(1.0, np.abs(p0).max())
    assert np.abs(pl.fill_probs() - p0).max() <= 1e-13 * pscale
    allc = np.arange(nP)
    Gs, Rs, Es = cm.model_sets(v, rhos, effects, (kind, obj, elem), allc, eps, gen_fd)
    Jm = ref.fill_dprobs_models(Gs, Rs, Es, eps=eps)
    Jf = pl.fill_dprobs(param_idx=allc, eps=eps)
    assert np.abs(Jf - Jm).max() <= 1e-6 * pscale                      # same sets to rounding, / eps
    objs = [(1, 0, np.arange(D), np.eye(D))] + [(2, e, np.arange(D + e * D, D + (e + 1) * D), np.eye(D)) for e in range(nEl)]
    objs += [(0, g, qs, dm) for g, (qs, dm) in enumerate(cm.gate_derivs(v, gen_dv)) if len(qs)]
    ref.set_derivs(nP, objs)
    Jh = ref.fill_dprobs(param_idx=allc, mode=_lib.DERIV_ANALYTIC)
    Ja = pl.fill_dprobs(param_idx=allc, mode=_lib.DERIV_ANALYTIC)
    jscale = max(1.0, np.abs(Jh).max())
    assert np.abs(Ja - Jh).max() <= 1e-11 * jscale
    assert (Ja[:, -2:] == 0).all() and (Jf[:, -2:] == 0).all()        # parameters nothing uses: exact zero columns
