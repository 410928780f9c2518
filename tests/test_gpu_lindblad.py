"""SURVEY 8(f) row f4 on hardware: Lindblad-parameterised (CPTPLND) models with the dense members built ON THE DEVICE
(gst_set_lindblad / gst_set_lindblad_params) -- against the reference's vectors of such models
(tests/golden/smq*_CPTPLND.npz: probabilities and Map-simulator FD columns from the real composed reps; `mm_*`: the
dense model after every FD step the reference took).  Tolerances as the north star states them: dense members 1e-14,
probabilities 1e-10, FD dprobs 1e-8."""
import numpy as np
import pytest

from conftest import load_fixture, plan_from_fixture
from pygsti_amd import lindblad as LB

pytestmark = pytest.mark.gpu

CASES = [("smq1Q_XYI_L4_CPTPLND", 1), ("smq2Q_XYICNOT_L1_CPTPLND", 2)]


def _plan(name, nq):
    fx, lb = load_fixture(name), load_fixture("lindblad_" + name)
    pl = plan_from_fixture(fx)                  # (sets the fixture's dense model; replaced below by the device-built one)
    model = LB.LindbladModel.from_fixture(lb, nq)
    pl.set_lindblad(model)
    pl.set_lindblad_params(lb["paramvec"])
    return fx, lb, model, pl


@pytest.mark.parametrize("name,nq", CASES)
def test_device_built_members_match_the_reference(name, nq):
    fx, lb, model, pl = _plan(name, nq)
    G, R, E = pl.get_model()
    assert np.abs(G - fx["gates"]).max() < 1e-14, np.abs(G - fx["gates"]).max()
    assert np.abs(R - fx["rhos"]).max() < 1e-14 and np.abs(E - fx["effects"]).max() < 1e-14
    cols = fx["dprobs_cols"]
    Gs, Rs, Es = pl.lindblad_model_sets(cols, float(fx["derivative_eps"]))
    assert np.abs(Gs - fx["mm_gates"]).max() < 1e-14, np.abs(Gs - fx["mm_gates"]).max()
    assert np.abs(Rs - fx["mm_rhos"]).max() < 1e-14 and np.abs(Es - fx["mm_effects"]).max() < 1e-14
    # a second parameter vector: larger error generators (scaling and squaring is exercised: ||L|| > 1/4)
    th2 = lb["paramvec"] * 40.0
    pl.set_lindblad_params(th2)
    G2, R2, E2 = pl.get_model()
    Gh, Rh, Eh = model.dense(th2)
    scale = max(1.0, np.abs(Gh).max())
    assert np.abs(G2 - Gh).max() < 1e-12 * scale and np.abs(R2 - Rh).max() < 1e-12 * scale and np.abs(E2 - Eh).max() < 1e-12 * scale


@pytest.mark.parametrize("name,nq", CASES)
def test_probs_and_fd_dprobs_of_cptplnd_models(name, nq):
    fx, lb, model, pl = _plan(name, nq)
    p = pl.fill_probs()
    assert np.abs(p - fx["probs"]).max() < 1e-10
    cols = fx["dprobs_cols"]
    pr = np.full(int(fx["nE"]), np.nan)
    J = np.full((int(fx["nE"]), len(cols) + 3), np.nan)
    pl.fill_dprobs(out=J, param_idx=cols, dest_idx=np.arange(len(cols)) + 2, eps=float(fx["derivative_eps"]), probs_out=pr)
    assert np.isnan(J[:, :2]).all() and np.isnan(J[:, -1]).all()
    assert np.abs(pr - fx["probs"]).max() < 1e-10
    err = np.abs(J[:, 2:-1] - fx["dprobs_map"]).max()
    assert err < 1e-8, err
    # the host-stepped form of the same walk (gst_fill_dprobs_models on the reference's own dense sets) agrees closely
    J2 = pl.fill_dprobs_models(fx["mm_gates"], fx["mm_rhos"], fx["mm_effects"], eps=float(fx["derivative_eps"]))
    assert np.abs(J[:, 2:-1] - J2).max() < 2e-8
    # device-resident output
    nE, n = int(fx["nE"]), len(cols)
    d = pl.device_malloc(nE * n * 8)
    pl.fill_dprobs_dev(d, n, cols, None, float(fx["derivative_eps"]), None)
    Jd = np.empty((nE, n)); pl.memcpy_d2h(Jd, d); pl.device_free(d)
    assert np.array_equal(Jd, J[:, 2:-1])


@pytest.mark.parametrize("name,nq", CASES)
def test_state_sharing_walk_agrees_with_independent_walks(name, nq):
    """The state-sharing form of a Lindblad model's FD Jacobian -- walk_pert_kernel (clean/dirty sharing with the base pass,
    64/D columns per wavefront) -- against one independent walk per (program, perturbed model) over the SAME device-built
    member sets (gst_fill_dprobs_models on gst_get_lindblad_model_sets): same perturbed members, same operation order, so
    they agree to the rounding of (p' - p) / eps -- for ragged column subsets, repeated and shuffled columns, a
    destination window, small and large task counts."""
    fx, lb = load_fixture(name), load_fixture("lindblad_" + name)
    model = LB.LindbladModel.from_fixture(lb, nq)
    nP, nE = model.num_params, int(fx["nE"])
    rng = np.random.default_rng(11)
    cols = np.concatenate([np.arange(nP) if nP <= 64 else np.sort(rng.choice(nP, 150, replace=False)), [3, 3, 0]])
    dest = rng.permutation(len(cols) + 5)[:len(cols)]
    outs = {}
    for tt in (0, 5):
        pl = plan_from_fixture(fx, target_tasks=tt)
        pl.set_lindblad(model); pl.set_lindblad_params(lb["paramvec"])
        J = np.full((nE, len(cols) + 5), np.nan)
        pr = np.empty(nE)
        pl.fill_dprobs(out=J, param_idx=cols, dest_idx=dest, eps=1e-7, probs_out=pr)
        untouched = np.setdiff1d(np.arange(len(cols) + 5), dest)
        assert np.isnan(J[:, untouched]).all() and not np.isnan(J[:, dest]).any()
        assert np.abs(pr - fx["probs"]).max() < 1e-10
        outs[tt] = J[:, dest]
        J_again = np.full_like(J, np.nan)
        pl.fill_dprobs(out=J_again, param_idx=cols, dest_idx=dest, eps=1e-7)
        assert np.array_equal(J_again[:, dest], J[:, dest])
    Gs, Rs, Es = pl.lindblad_model_sets(cols, 1e-7)
    ref = pl.fill_dprobs_models(Gs, Rs, Es, eps=1e-7)
    for k, v in outs.items():
        assert np.abs(v - ref).max() < 2e-8, (k, np.abs(v - ref).max())
    known = np.isin(cols, fx["dprobs_cols"])
    pos = [int(np.nonzero(fx["dprobs_cols"] == c)[0][0]) for c in cols[known]]
    assert np.abs(outs[0][:, known] - fx["dprobs_map"][:, pos]).max() < 1e-8


def test_lindblad_description_is_validated():
    from pygsti_amd import _lib
    fx, lb, model, pl = _plan("smq1Q_XYI_L4_CPTPLND", 1)
    bad = LB.LindbladModel(model.members[:-1], model.num_params, model.n_gates, model.n_rhos, model.n_effects)
    with pytest.raises(ValueError):
        pl.set_lindblad(bad)                        # the POVM belongs to no member
    pl.set_lindblad(None)                           # clears: FD needs a parameter map again
    pl.set_model(fx["gates"], fx["rhos"], fx["effects"])
    assert np.abs(pl.fill_probs() - fx["probs"]).max() < 1e-10


@pytest.mark.parametrize("param", ["CPTPLND", "GLND", "H+S", "H+s"])
def test_other_lindblad_parameterisations_through_the_simulator(param):
    """Every coefficient-block kind ('other' cholesky / elements, 'other_diagonal' cholesky / elements) through the host
    mirror's API: `LindbladExplicitModel` + `HipMapForwardSimulator.bulk_fill_probs / bulk_fill_dprobs`.  The device-built
    members equal the host restatement (scipy expm) to 1e-13; the Jacobian equals the one from host-stepped dense model
    sets (gst_fill_dprobs_models) to FD rounding."""
    from pygsti_amd import modelpacks as MP
    from pygsti_amd.forwardsim import HipMapForwardSimulator
    pack = MP.smq1Q_XYI
    m = LB.LindbladExplicitModel(pack.target_model(), param)
    rng = np.random.default_rng(5)
    th = 0.05 * rng.standard_normal(m.num_params)
    if param == "H+s":
        th[3:] = np.abs(th[3:])                     # ('elements' stochastic rates: keep them physical)
    m.from_vector(th)
    sim = HipMapForwardSimulator(m, num_atoms=2)
    circuits = pack.create_gst_circuits(4)
    lay = sim.create_layout(circuits)
    nE, nP = lay.num_elements, m.num_params
    p = np.empty(nE); sim.bulk_fill_probs(p, lay)
    J = np.empty((nE, nP)); pr = np.empty(nE)
    sim.bulk_fill_dprobs(J, lay, pr_array_to_fill=pr)
    assert np.array_equal(p, pr)
    for atom in lay.atoms:
        pl = atom.plan()
        desc = m.lindblad_description(lay.model_gate_labels, lay.effect_labels)
        G, R, E = pl.get_model()
        Gh, Rh, Eh = desc.dense(th)
        assert np.abs(G - Gh).max() < 1e-13 and np.abs(R - Rh).max() < 1e-13 and np.abs(E - Eh).max() < 1e-13
        Gs, Rs, Es = desc.model_sets(th, np.arange(nP), 1e-7)
        J2 = pl.fill_dprobs_models(Gs, Rs, Es, eps=1e-7)
        # (both quotients carry the rounding of their probabilities, ~1e-16 |p| / eps; an unconstrained GLND generator
        #  is not completely positive, so |p| need not stay below 1)
        tol = 2e-8 * max(1.0, np.abs(pr).max(), np.abs(J2).max())
        assert np.abs(J[atom.element_slice] - J2).max() < tol, (np.abs(J[atom.element_slice] - J2).max(), tol)
    # probabilities of every circuit sum to one (trace preservation of every generator)
    assert np.abs(p.reshape(-1, 2).sum(1) - 1.0).max() < 1e-11 * max(1.0, np.abs(p).max())


def test_analytic_jacobian_of_a_cptplnd_model_with_device_computed_member_derivatives():
    """GST_DERIV_ANALYTIC with gst_set_lindblad: d(dense member)/d(parameter) comes from the device (Frechet derivative of
    the exponential), not from the host's deriv_wrt_params() -- against the Matrix simulator's Jacobian of the same model
    (<= 1e-8, fixture `dprobs_matrix`), against the same analytic path fed with the reference's own derivative matrices
    (gst_set_derivs, fixture `dv_*`), and against the finite-difference columns (to FD accuracy); 2Q: analytic vs FD."""
    fx, lb, model, pl = _plan("smq1Q_XYI_L4_CPTPLND", 1)
    from pygsti_amd import _lib
    nE, nP = int(fx["nE"]), int(fx["nP"])
    rows = fx["matrix_rows"]
    Ja = pl.fill_dprobs(param_idx=np.arange(nP), mode=_lib.DERIV_ANALYTIC)
    err = np.abs(Ja[rows] - fx["dprobs_matrix"]).max()
    assert err < 1e-8, err
    # a column window with destinations
    cols = np.array([59, 3, 17, 30]); win = np.full((nE, 7), np.nan)
    pl.fill_dprobs(out=win, param_idx=cols, dest_idx=np.array([6, 0, 2, 3]), mode=_lib.DERIV_ANALYTIC)
    assert np.abs(win[:, [6, 0, 2, 3]] - Ja[:, cols]).max() < 1e-12 and np.isnan(win[:, [1, 4, 5]]).all()
    # the reference's own member derivatives through gst_set_derivs
    pl2 = plan_from_fixture(fx)
    objs, off_c, off_d = [], 0, 0
    for k, o, n in zip(fx["dv_kind"], fx["dv_obj"], fx["dv_ncols"]):
        K = 16 if k == 0 else 4
        objs.append((int(k), int(o), fx["dv_param_idx"][off_c:off_c + n], fx["dv_deriv"][off_d:off_d + K * n].reshape(K, n)))
        off_c += n; off_d += K * n
    pl2.set_derivs(nP, objs)
    Jr = pl2.fill_dprobs(param_idx=np.arange(nP), mode=_lib.DERIV_ANALYTIC)
    assert np.abs(Ja - Jr).max() < 1e-11, np.abs(Ja - Jr).max()
    Jf = pl.fill_dprobs(param_idx=np.arange(nP), eps=1e-7)
    assert np.abs(Ja - Jf).max() < 1e-4            # (forward differences: eps times the second derivative)
    # two qubits: exact against finite differences on a spread of columns
    fx2, lb2, model2, pl2q = _plan("smq2Q_XYICNOT_L1_CPTPLND", 2)
    c2 = fx2["dprobs_cols"]
    Ja2 = pl2q.fill_dprobs(param_idx=c2, mode=_lib.DERIV_ANALYTIC)
    assert np.abs(Ja2 - fx2["dprobs_map"]).max() < 1e-4, np.abs(Ja2 - fx2["dprobs_map"]).max()


DEEP = [("smq1Q_XYI_L128_CPTPLND", 1), ("smq2Q_XYICNOT_L1024_CPTPLND_deep", 2)]


@pytest.mark.parametrize("name,nq", DEEP)
def test_cptplnd_models_at_depth(name, nq):
    """CPTPLND models where the numbers are quoted: the full 1Q L<=128 design and whole 2Q germ-power families to depth
    1,030, against the reference simulator each derivative mode mirrors -- every offered route inside the stated tolerance.

    * probabilities <= 1e-10 at every depth;
    * ANALYTIC (device-computed member derivatives + chain rule) vs the Matrix simulator: <= 1e-8 ABSOLUTE at every depth;
    * FD over DEVICE-built members is refused at depth (GST_EUNSUPPORTED beyond GST_LINDBLAD_FD_MAX_DEPTH = 16: the device's
      scaled-Taylor exponential and scipy's Pade approximant differ in the last bit of the perturbed member, and the quotient
      amplifies that by occurrences / eps -- 5e-8 ... 1.2e-7 beyond depth 40, profiles/r04_cptplnd_depth_profile_*.json);
    * FD at depth is the walk over the REFERENCE's own dense members after every step (`mm_*` of the fixture: what
      model.set_parameter_value produced in pyGSTi), base model included: <= 1e-8 vs the Map simulator at EVERY depth."""
    from conftest import matrix_rows_by_circuit, element_depth
    from pygsti_amd import _lib
    fx, lb, model, pl = _plan(name, nq)
    depth = element_depth(fx)
    assert depth.max() > _lib.LINDBLAD_FD_MAX_DEPTH and pl.stats()["max_depth"] == depth.max()
    cols = fx["dprobs_cols"]
    p = pl.fill_probs()
    assert np.abs(p - fx["probs"]).max() < 1e-10
    with pytest.raises(_lib.GstUnsupported, match="depth"):
        pl.fill_dprobs(param_idx=cols, eps=float(fx["derivative_eps"]))
    d = pl.device_malloc(int(fx["nE"]) * len(cols) * 8)
    with pytest.raises(_lib.GstUnsupported, match="depth"):
        pl.fill_dprobs_dev(d, len(cols), cols, None, float(fx["derivative_eps"]), None)
    pl.device_free(d)
    Ja = pl.fill_dprobs(param_idx=cols, mode=_lib.DERIV_ANALYTIC)
    rows = matrix_rows_by_circuit(fx)
    e_an = np.abs(Ja - fx["matrix_by_circuit_dprobs"][rows]).max(axis=1)
    assert e_an.max() < 1e-8, ("analytic vs Matrix", e_an.max())
    if "mm_gates" in fx:
        pl.set_lindblad(None)
        pl.set_model(fx["gates"], fx["rhos"], fx["effects"])              # the reference's own base members
        pr = np.empty(int(fx["nE"]))
        Jm = pl.fill_dprobs_models(fx["mm_gates"], fx["mm_rhos"], fx["mm_effects"], eps=float(fx["derivative_eps"]), probs_out=pr)
        assert np.abs(pr - fx["probs"]).max() < 1e-10
        e_fd = np.abs(Jm - fx["dprobs_map"]).max(axis=1)
        for lo, hi in ((0, 16), (17, 80), (81, 600), (601, 2000)):
            m = (depth >= lo) & (depth <= hi)
            if m.any():
                assert e_fd[m].max() < 1e-8, ("host-stepped FD vs Map, depth %d-%d" % (lo, hi), float(e_fd[m].max()))


def test_deep_lindblad_fd_through_the_host_mirror_takes_host_stepped_members():
    """HipMapForwardSimulator (derivative_mode 'fd') on a Lindblad-parameterised model at depth > 16: base and stepped members
    from the model's own (host, scipy) exponential, walked as dense sets -- equal to an explicit gst_fill_dprobs_models over
    `model_sets`, and consistent with the exact route to FD accuracy; shallow layouts keep the device-built route."""
    from pygsti_amd import modelpacks as MP, _lib
    from pygsti_amd.forwardsim import HipMapForwardSimulator
    pack = MP.smq1Q_XYI
    m = LB.LindbladExplicitModel(pack.target_model(), "CPTPLND")
    rng = np.random.default_rng(9)
    th = 0.03 * rng.standard_normal(m.num_params)
    m.from_vector(th)
    circuits = pack.create_gst_circuits(32)
    sim = HipMapForwardSimulator(m, num_atoms=1)
    lay = sim.create_layout(circuits)
    nE, nP = lay.num_elements, m.num_params
    atom = lay.atoms[0]
    assert atom.plan().stats()["max_depth"] > _lib.LINDBLAD_FD_MAX_DEPTH
    J = np.empty((nE, nP)); pr = np.empty(nE)
    sim.bulk_fill_dprobs(J, lay, pr_array_to_fill=pr)
    desc = m.lindblad_description(lay.model_gate_labels, lay.effect_labels)
    pl = atom.plan()
    pl.set_lindblad(None); pl.set_model(*desc.dense(th))
    J2 = pl.fill_dprobs_models(*desc.model_sets(th, np.arange(nP), 1e-7), eps=1e-7)
    assert np.array_equal(J, J2)
    sim2 = HipMapForwardSimulator(m, num_atoms=1, derivative_mode="analytic")
    lay2 = sim2.create_layout(circuits)
    Ja = np.empty((nE, nP)); sim2.bulk_fill_dprobs(Ja, lay2)
    assert np.abs(J - Ja).max() < 2e-4 * max(1.0, np.abs(Ja).max())        # (FD truncation error at L = 32)
    p = np.empty(nE); sim.bulk_fill_probs(p, lay)                           # the next fill re-enters the device-built route
    assert np.abs(p - pr).max() < 1e-12
