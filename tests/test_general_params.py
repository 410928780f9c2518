"""General parameterisations (TP, CPTP-constrained) in the analytic mode: d p / d parameter = (d p / d element) x the
members' deriv_wrt_params, as MatrixForwardSimulator computes it.  Golden vectors: MatrixForwardSimulator.bulk_fill_dprobs
on `full TP` / `CPTPLND` models, with the members' deriv_wrt_params() and gpindices captured next to them
(tests/golden/make_golden.py, case 'tp')."""
import numpy as np
import pytest

from conftest import force, load_fixture, plan_from_fixture
from oracle import oracle as O

CASES = ["smq1Q_XYI_L4_TP", "smq1Q_XYI_L4_CPTPLND", "smq2Q_XYICNOT_L1_TP"]


@pytest.mark.parametrize("name", CASES)
def test_oracle_chain_rule_matches_matrix_simulator(name):
    fx = load_fixture(name)
    J, P = O.analytic_dprobs_general(fx, fx["dprobs_cols"])
    rows = fx["matrix_rows"]
    assert np.abs(J[rows] - fx["dprobs_matrix"]).max() < 1e-12
    assert np.abs(P[rows] - fx["probs_matrix"]).max() < 1e-13
    # the fixtures are genuinely not `full`: parameters shared between objects (a POVM's effects), and for TP fewer
    # parameters than dense elements
    D = int(fx["D"])
    n_el = (len(fx["rhos"]) + len(fx["effects"])) * D + len(fx["gates"]) * D * D
    assert int(fx["nP"]) <= n_el and (not name.endswith("_TP") or int(fx["nP"]) < n_el)
    pidx = fx["dv_param_idx"]
    assert len(np.unique(pidx)) < len(pidx)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_gpu_general_parameterisation_dprobs(name):
    from pygsti_amd import _lib
    fx = load_fixture(name)
    pl = plan_from_fixture(fx)
    nP = int(fx["nP"])
    pl.set_derivs(nP, O.derivs_from_fixture(fx))
    cols = fx["dprobs_cols"]
    pr = np.empty(int(fx["nE"]))
    J = pl.fill_dprobs(param_idx=cols, probs_out=pr, mode=_lib.DERIV_ANALYTIC)
    rows = fx["matrix_rows"]
    assert np.abs(J[rows] - fx["dprobs_matrix"]).max() < 1e-8          # the north-star bar; observed ~1e-14
    if name.endswith("_TP"):      # dense members: the probabilities are the reference Map path's, bit for bit
        assert np.abs(pr - fx["probs"]).max() == 0.0
    else:                         # CPTPLND members act through composed (non-dense) reps in the reference; here their
        assert np.abs(pr - fx["probs"]).max() < 1e-10   # dense superoperators are used
    Jo, _ = O.analytic_dprobs_general(fx, cols)
    assert np.abs(J - Jo).max() < 1e-11
    # a column window with a destination offset, through the device-resident entry point
    nE = int(fx["nE"])
    sub = cols[3:3 + min(17, len(cols) - 3)]
    out = np.full((nE, len(sub) + 6), -7.0)
    pl.fill_dprobs(out=out, param_idx=sub, dest_idx=np.arange(len(sub)) + 4, mode=_lib.DERIV_ANALYTIC)
    assert np.array_equal(out[:, 4:4 + len(sub)], J[:, 3:3 + len(sub)])
    assert (out[:, :4] == -7.0).all() and (out[:, 4 + len(sub):] == -7.0).all()
    # finite differences are refused loudly in this mode; clearing the derivatives restores the element map
    with pytest.raises(Exception):
        pl.fill_dprobs(param_idx=cols[:2], mode=_lib.DERIV_FD)
    pl.set_derivs(nP, [])
    J0 = pl.fill_dprobs(param_idx=np.arange(min(nP, 5)), mode=_lib.DERIV_FD)     # kind -1 everywhere: exact zeros
    assert (J0 == 0).all()


TP_CASES = ["smq1Q_XYI_L4_TP", "smq2Q_XYICNOT_L1_TP"]


@pytest.mark.parametrize("name", TP_CASES + ["3q_explicit_TP"])
def test_tp_fixtures_describe_the_complement(name):
    """The complement effect stored in the fixture is identity - sum(others) in the recorded order, bit for bit
    (complementeffect.py:72-78), and the TP parameter map covers every parameter exactly once."""
    fx = load_fixture(name)
    ci, others, ident = int(fx["comp_index"]), fx["comp_others"], fx["comp_identity"]
    assert np.array_equal(fx["effects"][ci], ident - sum([fx["effects"][o] for o in others]))
    pk, po, pe = O.tp_param_map(fx)
    assert (pk >= 0).all() and not ((pk == 2) & (po == ci)).any()
    assert len(set(zip(pk.tolist(), po.tolist(), pe.tolist()))) == int(fx["nP"])


def test_complement_effect_arguments_are_validated():
    from pygsti_amd import _lib
    fx = load_fixture("smq1Q_XYI_L4_TP")
    pl = plan_from_fixture(fx)
    ident = fx["comp_identity"]
    with pytest.raises(ValueError):
        pl.set_complement_effect(7, ident, [0])            # no such effect
    with pytest.raises(ValueError):
        pl.set_complement_effect(1, ident, [1])            # the complement cannot be one of its own terms
    with pytest.raises(ValueError):
        pl.set_complement_effect(1, ident, [0, 0])
    pl.set_complement_effect(1, ident, [0])
    pl.set_complement_effect(-1)


@pytest.mark.gpu
@pytest.mark.parametrize("name", TP_CASES)
def test_gpu_tp_model_fd_dprobs_bitwise(name):
    """FD Jacobian of a "full TP" model, bit for bit what the reference's Map simulator returns for it
    (fixture dprobs_map = MapForwardSimulator._bulk_fill_dprobs_atom on the TP model): TPState / FullTPOp parameters
    are dense elements; an effect parameter also moves the TPPOVM's complement outcome."""
    from pygsti_amd import _lib
    fx = load_fixture(name)
    pl = plan_from_fixture(fx)
    pl.set_param_map(*O.tp_param_map(fx))
    pl.set_complement_effect(int(fx["comp_index"]), fx["comp_identity"], fx["comp_others"])
    cols = fx["dprobs_cols"]
    pr = np.empty(int(fx["nE"]))
    J = pl.fill_dprobs(param_idx=cols, probs_out=pr, eps=float(fx["derivative_eps"]), mode=_lib.DERIV_FD)
    assert np.array_equal(pr, fx["probs"])
    assert np.array_equal(J, fx["dprobs_map"])
    pk, po, _ = O.tp_param_map(fx)
    eff_cols = np.nonzero(pk[cols] == 2)[0]
    assert len(eff_cols) > 0
    # the complement's outcome really moves with the other effects' parameters
    comp_rows = np.nonzero(fx["eff_label"] == int(fx["comp_index"]))[0]
    assert np.abs(J[fx["eff_dest"][comp_rows]][:, eff_cols]).max() > 0
    # effect columns only (no walk launch), scattered into a wider output
    sub = cols[eff_cols]
    out = np.full((int(fx["nE"]), len(sub) + 3), -3.0)
    pl.fill_dprobs(out=out, param_idx=sub, dest_idx=np.arange(len(sub)) + 2, eps=float(fx["derivative_eps"]), mode=_lib.DERIV_FD)
    assert np.array_equal(out[:, 2:2 + len(sub)], J[:, eff_cols]) and (out[:, :2] == -3.0).all() and (out[:, -1] == -3.0).all()
    # exact derivatives of such a plan are refused (gst_set_derivs is the exact route)
    with pytest.raises(Exception):
        pl.fill_dprobs(param_idx=cols[:2], mode=_lib.DERIV_ANALYTIC)
    with pytest.raises(Exception):
        pl.fill_hprobs(idx1=cols[:2], idx2=cols[:2], mode=_lib.DERIV_ANALYTIC)
    # FD-of-FD Hessian block of the TP model, bit for bit the Map simulator's (MapForwardSimulator._bulk_fill_hprobs_atom
    # re-derives the complement after each of the two parameter steps)
    H = pl.fill_hprobs(idx1=fx["hprobs_rows"], idx2=fx["hprobs_cols"], eps=float(fx["hessian_eps"]))
    assert np.array_equal(H, fx["hprobs_map"])
    kr, kc = pk[fx["hprobs_rows"]], pk[fx["hprobs_cols"]]
    assert (kr == 2).any() and (kc == 2).any() and (kr == 0).any() and (kc == 0).any()
    assert np.abs(H[fx["eff_dest"][comp_rows]][:, kr == 2][:, :, kc == 2]).max() >= 0      # (effect x effect on the complement outcome)
    assert np.abs(H[fx["eff_dest"][comp_rows]][:, kr == 2][:, :, kc == 0]).max() > 0       # effect x gate moves it
    # without the declaration the complement outcome would be wrong: the test is sensitive to it
    pl.set_complement_effect(-1)
    J2 = pl.fill_dprobs(param_idx=cols, eps=float(fx["derivative_eps"]), mode=_lib.DERIV_FD)
    assert not np.array_equal(J2, fx["dprobs_map"])
    ncomp = np.setdiff1d(np.arange(int(fx["nE"])), fx["eff_dest"][comp_rows])
    assert np.array_equal(J2[ncomp], fx["dprobs_map"][ncomp])


@pytest.mark.gpu
def test_gpu_3q_tp_model_fd_jacobian_and_hessian_bitwise():
    """The 3-qubit model as "full TP" (D = 64, 40,831 parameters, a 7-effect TPPOVM + complement): probabilities, 48 FD
    Jacobian columns and a 6 x 16 FD-of-FD Hessian block, bit for bit the reference Map simulator's
    (tests/golden/make_golden.py, case '3qtp').  The fused D = 64 Hessian kernel has no form that re-derives a complement
    effect; the block is composed from FD Jacobians of stepped models exactly as _mapfill_hprobs_atom composes it
    (mapforwardsim.py:420-436) -- round 2 refused it with GST_EUNSUPPORTED."""
    fx = load_fixture("3q_explicit_TP")
    assert int(fx["D"]) == 64
    pl = plan_from_fixture(fx)
    pk, po, pe = O.tp_param_map(fx)
    pl.set_param_map(pk, po, pe)
    pl.set_complement_effect(int(fx["comp_index"]), fx["comp_identity"], fx["comp_others"])
    cols = fx["dprobs_cols"]
    pr = np.empty(int(fx["nE"]))
    J = pl.fill_dprobs(param_idx=cols, probs_out=pr, eps=float(fx["derivative_eps"]))
    assert np.array_equal(pr, fx["probs"])
    assert np.array_equal(J, fx["dprobs_map"])
    assert (pk[cols] == 2).any() and (pk[cols] == 1).any() and (pk[cols] == 0).any()
    rows, c2 = fx["hprobs_rows"], fx["hprobs_cols"]
    H = pl.fill_hprobs(idx1=rows, idx2=c2, eps=float(fx["hessian_eps"]))
    assert np.array_equal(H, fx["hprobs_map"])
    assert (pk[rows] == 2).any() and (pk[c2] == 2).any() and len(np.intersect1d(rows, c2)) > 0
    comp_rows = fx["eff_dest"][fx["eff_label"] == int(fx["comp_index"])]
    assert np.abs(H[comp_rows][:, pk[rows] == 2][:, :, pk[c2] == 0]).max() > 0          # effect x gate moves the complement's outcome
    # scattered destination, and the model is left as it was
    out = np.full((int(fx["nE"]), len(rows) + 2, len(c2) + 3), -5.0)
    pl.fill_hprobs(out=out, idx1=rows, dest1=np.arange(len(rows))[::-1].copy() + 1, idx2=c2, dest2=np.arange(len(c2)) + 2, eps=float(fx["hessian_eps"]))
    assert np.array_equal(out[:, 1:1 + len(rows), 2:2 + len(c2)], H[:, ::-1, :])
    assert (out[:, 0] == -5.0).all() and (out[:, -1] == -5.0).all() and (out[:, :, :2] == -5.0).all() and (out[:, :, -1] == -5.0).all()
    assert np.array_equal(pl.fill_probs(), fx["probs"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", TP_CASES + ["smq1Q_XYI_L4_depol"])
def test_gpu_composed_hessian_is_the_fused_one(name, monkeypatch):
    """GST_TEST_FORCE hess_composed=1 sends every FD-of-FD block through the composed route (n1 + 2 FD Jacobians of stepped models);
    on plans the fused two-perturbation kernel covers, both equal the reference's block bit for bit."""
    fx = load_fixture(name)
    force(monkeypatch, hess_composed=1)
    pl = plan_from_fixture(fx)
    if "comp_index" in fx:
        pl.set_param_map(*O.tp_param_map(fx))
        pl.set_complement_effect(int(fx["comp_index"]), fx["comp_identity"], fx["comp_others"])
    H = pl.fill_hprobs(idx1=fx["hprobs_rows"], idx2=fx["hprobs_cols"], eps=float(fx["hessian_eps"]))
    assert np.array_equal(H, fx["hprobs_map"])
    assert np.array_equal(pl.fill_probs(), fx["probs"])


def test_oracle_tp_exact_hessian_matches_matrix_simulator():
    """The numpy chain rule of the element Hessian, against MatrixForwardSimulator.bulk_fill_hprobs on the TP model."""
    fx = load_fixture("smq1Q_XYI_L4_TP")
    H = O.analytic_hprobs_general(fx, fx["hprobs_rows"], fx["hprobs_cols"])
    ref = fx["hprobs_matrix"]
    assert np.abs(H[fx["matrix_rows"]] - ref).max() < 1e-10 * max(1.0, np.abs(ref).max())
    assert np.abs(ref).max() > 0.1


@pytest.mark.gpu
def test_gpu_tp_exact_hessian_block():
    """gst_fill_hprobs_analytic with gst_set_derivs (linear parameterisation): <= 1e-8 against MatrixForwardSimulator's
    Hessian of the TP model, and against the numpy chain rule on the 2Q TP fixture (rows/columns over rho, effects that
    share the complement, gates)."""
    from pygsti_amd import _lib
    fx = load_fixture("smq1Q_XYI_L4_TP")
    pl = plan_from_fixture(fx)
    pl.set_derivs(int(fx["nP"]), O.derivs_from_fixture(fx))
    r, c = fx["hprobs_rows"], fx["hprobs_cols"]
    H = pl.fill_hprobs(idx1=r, idx2=c, mode=_lib.DERIV_ANALYTIC)
    ref = fx["hprobs_matrix"]
    assert np.abs(H[fx["matrix_rows"]] - ref).max() < 1e-8
    assert np.abs(H - O.analytic_hprobs_general(fx, r, c)).max() < 1e-10
    # FD of FD on the same model agrees to the accuracy of finite differences
    pl.set_derivs(int(fx["nP"]), [])
    pl.set_param_map(*O.tp_param_map(fx))
    pl.set_complement_effect(int(fx["comp_index"]), fx["comp_identity"], fx["comp_others"])
    Hfd = pl.fill_hprobs(idx1=r, idx2=c, eps=float(fx["hessian_eps"]))
    assert np.abs(Hfd - H).max() < 2e-3 * max(1.0, np.abs(H).max())
    # destination windows inside a larger block
    pl.set_derivs(int(fx["nP"]), O.derivs_from_fixture(fx))
    out = np.full((int(fx["nE"]), len(r) + 2, len(c) + 3), -5.0)
    pl.fill_hprobs(out=out, idx1=r, idx2=c, dest1=np.arange(len(r)) + 1, dest2=np.arange(len(c)) + 2, mode=_lib.DERIV_ANALYTIC)
    assert np.array_equal(out[:, 1:1 + len(r), 2:2 + len(c)], H)
    assert (out[:, 0] == -5.0).all() and (out[:, :, :2] == -5.0).all() and (out[:, -1] == -5.0).all() and (out[:, :, -1] == -5.0).all()
    # 2Q
    fx2 = load_fixture("smq2Q_XYICNOT_L1_TP")
    pl2 = plan_from_fixture(fx2)
    pl2.set_derivs(int(fx2["nP"]), O.derivs_from_fixture(fx2))
    r2, c2 = fx2["hprobs_rows"], fx2["hprobs_cols"]
    H2 = pl2.fill_hprobs(idx1=r2, idx2=c2, mode=_lib.DERIV_ANALYTIC)
    Ho = O.analytic_hprobs_general(fx2, r2, c2)
    assert np.abs(H2 - Ho).max() < 1e-10 * max(1.0, np.abs(Ho).max())
    assert np.abs(H2 - fx2["hprobs_map"]).max() < 2e-3 * max(1.0, np.abs(Ho).max())      # vs the Map simulator's FD of FD


def test_oracle_cptplnd_exact_hessian_matches_matrix_simulator():
    """Members that are not linear in their parameters: the chain rule plus the J_elem . hessian_wrt_params term, against
    MatrixForwardSimulator.bulk_fill_hprobs on the CPTPLND model."""
    fx = load_fixture("smq1Q_XYI_L4_CPTPLND")
    assert fx["dv2_nonzero"].all()
    H = O.analytic_hprobs_general(fx, fx["hprobs_rows"], fx["hprobs_cols"])
    ref = fx["hprobs_matrix"]
    assert np.abs(H[fx["matrix_rows"]] - ref).max() < 1e-11 * max(1.0, np.abs(ref).max())
    # the second-derivative term matters: without it the block is visibly wrong
    fx0 = dict(fx); fx0["dv2_nonzero"] = np.zeros_like(fx["dv2_nonzero"])
    H0 = O.analytic_hprobs_general(fx0, fx["hprobs_rows"], fx["hprobs_cols"])
    assert np.abs(H0[fx["matrix_rows"]] - ref).max() > 1e-3


@pytest.mark.gpu
def test_gpu_cptplnd_exact_hessian_block():
    """gst_set_derivs + gst_set_second_derivs: exact Hessian block of the CPTPLND model, <= 1e-8 against
    MatrixForwardSimulator (every member there is an exponentiated error generator: all six objects carry a second-
    derivative tensor; the two effects share their parameters)."""
    from pygsti_amd import _lib
    fx = load_fixture("smq1Q_XYI_L4_CPTPLND")
    pl = plan_from_fixture(fx)
    pl.set_derivs(int(fx["nP"]), O.derivs_from_fixture(fx))
    pl.set_second_derivs(O.second_derivs_from_fixture(fx))
    r, c = fx["hprobs_rows"], fx["hprobs_cols"]
    H = pl.fill_hprobs(idx1=r, idx2=c, mode=_lib.DERIV_ANALYTIC)
    ref = fx["hprobs_matrix"]
    assert np.abs(H[fx["matrix_rows"]] - ref).max() < 1e-8 * max(1.0, np.abs(ref).max())
    assert np.abs(H - O.analytic_hprobs_general(fx, r, c)).max() < 1e-10 * max(1.0, np.abs(ref).max())
    # destination window, and the whole 60 x 60 Hessian is symmetric
    out = np.full((int(fx["nE"]), len(r) + 1, len(c) + 2), 9.0)
    pl.fill_hprobs(out=out, idx1=r, idx2=c, dest1=np.arange(len(r)) + 1, dest2=np.arange(len(c)) + 1, mode=_lib.DERIV_ANALYTIC)
    assert np.array_equal(out[:, 1:, 1:1 + len(c)], H) and (out[:, 0] == 9.0).all() and (out[:, :, 0] == 9.0).all() and (out[:, :, -1] == 9.0).all()
    full = pl.fill_hprobs(mode=_lib.DERIV_ANALYTIC)
    assert np.abs(full - np.transpose(full, (0, 2, 1))).max() < 1e-9 * max(1.0, np.abs(full).max())
    # without the tensors the linear chain rule alone is returned (and is not the Hessian of this model)
    pl.set_second_derivs([])
    H_lin = pl.fill_hprobs(idx1=r, idx2=c, mode=_lib.DERIV_ANALYTIC)
    assert np.abs(H_lin[fx["matrix_rows"]] - ref).max() > 1e-3


# ---- implicit models: embedded / composed layer operations (opcreps.cpp:93-158, 242-276) -----------------------------------------
def test_oracle_chain_rule_on_the_implicit_3q_model():
    """`3q_crosstalk_free` (a LocalNoiseModel generated by the reference: every circuit layer an EmbeddedOp of a 1Q / 2Q
    dense factor, or a ComposedOp of several, layers SHARING the factors' parameters): the numpy chain rule over the
    layers' deriv_wrt_params equals the Matrix simulator's columns; the Map simulator's FD columns differ from them by the
    FD truncation error only; the dense layer operations reproduce the probabilities the reference got by acting factor
    by factor."""
    from conftest import matrix_rows_by_circuit
    fx = load_fixture("3q_crosstalk_free")
    rows = matrix_rows_by_circuit(fx)
    J, P = O.analytic_dprobs_general(fx, fx["dprobs_cols"])
    assert np.abs(J - fx["matrix_by_circuit_dprobs"][rows]).max() < 1e-12
    assert np.abs(P - fx["probs"]).max() < 1e-14 and np.abs(P - fx["matrix_by_circuit_probs"][rows]).max() < 1e-14
    d = np.abs(fx["dprobs_map"] - fx["matrix_by_circuit_dprobs"][rows]).max()
    assert 1e-9 < d < 1e-4
    off, spans = 0, []
    for n in fx["dv_ncols"]:
        spans.append(set(fx["dv_param_idx"][off:off + n].tolist())); off += n
    assert sum(1 for a in range(len(spans)) for b in range(a) if spans[a] & spans[b]) >= 10      # layers share parameters
    assert any(str(l).startswith("[") for l in fx["op_labels"])                                    # composed layers


@pytest.mark.gpu
def test_gpu_implicit_3q_model_exact_jacobian():
    """The same model on the device (D = 64): the layers' dense operations + their derivative matrices through
    gst_set_derivs -- nothing is re-densified per column on the host -- exact Jacobian <= 1e-8 against the Matrix
    simulator, probabilities <= 1e-10 against the Map simulator (which propagates through the embedded / composed reps);
    then through the drop-in's per-atom logic with stand-in members, whose `auto` derivative mode takes this route."""
    from conftest import matrix_rows_by_circuit
    from pygsti_amd import _lib
    import test_gpu_adapter_modes as M
    fx = load_fixture("3q_crosstalk_free")
    rows = matrix_rows_by_circuit(fx)
    cols = fx["dprobs_cols"]
    nE, nP = int(fx["nE"]), int(fx["nP"])
    pl = plan_from_fixture(fx)
    pl.set_derivs(nP, O.derivs_from_fixture(fx))
    pr = np.empty(nE)
    J = pl.fill_dprobs(param_idx=cols, probs_out=pr, mode=_lib.DERIV_ANALYTIC)
    assert np.abs(J - fx["matrix_by_circuit_dprobs"][rows]).max() < 1e-8
    assert np.abs(pr - fx["probs"]).max() < 1e-10
    atom, model = M._Atom(fx), M._dv_model(fx, None)
    sim = M._Sim(model, "auto")
    Ja = np.full((nE, len(cols)), np.nan)
    sim._bulk_fill_dprobs_atom(Ja, None, atom, cols, None)
    assert atom._hip_plan._hip_mode == "derivs"
    assert np.abs(Ja - fx["matrix_by_circuit_dprobs"][rows]).max() < 1e-8
    p = np.empty(nE); sim._bulk_fill_probs_atom(p, atom, None)
    assert np.abs(p - fx["probs"]).max() < 1e-10


@pytest.mark.gpu
def test_gpu_chain_rule_with_ragged_objects():
    """The chain-rule products (gst_kernels_normal.hip: 128 x 80 tiles for the big objects, 64 x 64 for the small ones) at
    shapes no reference parameterisation has: parameter counts that are no multiple of a tile (97, 41, 240 + 1), a row
    count that is no multiple of 128, four effects sharing their parameters (stacked into one product), a gate with no
    parameters at all, and a shared parameter between two gates (the second object ADDS).  Expected: the same plan's
    element Jacobian (pinned against the Matrix simulator elsewhere) times the derivative matrices, in numpy."""
    from pygsti_amd import _lib
    fx = load_fixture("smq2Q_XYICNOT_L2_depol")
    pl = plan_from_fixture(fx)
    nE, D = int(fx["nE"]), 16
    nG, nEf = fx["gates"].shape[0], fx["effects"].shape[0]
    Jp = pl.fill_dprobs(param_idx=np.arange(int(fx["nP"])), mode=_lib.DERIV_ANALYTIC)
    # ... re-ordered into the element layout [rho | effects | gates] whatever the fixture's parameter order is
    pk, po, pe = (np.asarray(fx[k]) for k in ("pkind", "pobj", "pelem"))
    col = np.where(pk == 1, pe, np.where(pk == 2, D + po * D + pe, D + nEf * D + po * D * D + pe))
    Jel = np.zeros((nE, D + nEf * D + nG * D * D))
    Jel[:, col[pk >= 0]] = Jp[:, pk >= 0]
    rng = np.random.default_rng(11)
    sizes = {"g0": 97, "g1": 41, "g2": 241, "rho": 5, "povm": 23}
    off, p = {}, 0
    for k, n in sizes.items():
        off[k] = p; p += n
    nP = p
    objs, expected = [], np.zeros((nE, nP))
    base_rho, base_eff, base_gate = 0, D, D + nEf * D

    def add(kind, obj, pidx, W, a0):
        objs.append((kind, obj, np.asarray(pidx, np.int64), W))
        expected[:, pidx] += Jel[:, a0:a0 + W.shape[0]] @ W
    for gi, key in enumerate(("g0", "g1", "g2")):
        W = rng.standard_normal((D * D, sizes[key]))
        W[rng.random(W.shape) < 0.5] = 0.0
        add(0, gi, off[key] + np.arange(sizes[key]), W, base_gate + gi * D * D)
    # gate 4 shares the first 30 parameters of g0 (later object: adds); gate 3 has no parameters
    add(0, 4, off["g0"] + np.arange(30), rng.standard_normal((D * D, 30)), base_gate + 4 * D * D)
    add(1, 0, off["rho"] + np.arange(sizes["rho"]), rng.standard_normal((D, sizes["rho"])), base_rho)
    for e in range(nEf):
        add(2, e, off["povm"] + np.arange(sizes["povm"]), rng.standard_normal((D, sizes["povm"])), base_eff + e * D)
    pl.set_derivs(nP, objs)
    J = pl.fill_dprobs(param_idx=np.arange(nP), mode=_lib.DERIV_ANALYTIC)
    scale = np.abs(expected).max()
    assert scale > 1.0 and np.abs(J - expected).max() < 1e-12 * scale, np.abs(J - expected).max()
    # a scattered column request into a wider destination
    sub = np.array([off["g2"] + 240, off["g0"] + 3, off["povm"] + 22, off["g1"] + 40, off["g0"] + 96])
    out = np.full((nE, 9), -3.0)
    pl.fill_dprobs(out=out, param_idx=sub, dest_idx=np.array([7, 0, 3, 5, 2]), mode=_lib.DERIV_ANALYTIC)
    assert np.abs(out[:, [7, 0, 3, 5, 2]] - expected[:, sub]).max() < 1e-12 * scale
    assert (out[:, [1, 4, 6, 8]] == -3.0).all()
