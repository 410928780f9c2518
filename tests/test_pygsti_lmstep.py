"""The device LM step reached from the REFERENCE'S OWN optimizer (runs only where the real pyGSTi is
importable -- the build container with PYTHONPATH=/tmp/pgref -- and is skipped on the GPU box, where
tests/test_fit_replay.py drives the same logic classes with stand-ins on the real device).

No GPU here, so `_lib.Plan` is replaced by a RECORDING stand-in (tests/_fake_device.py: the CPU checker behind fake
device pointers).  What is pinned:

  * `HipChi2Function.dlsvec` / `HipPoissonPicDeltaLogLFunction.dlsvec` return a `DeviceJacobian` whose J^T J, J^T f and
    Frobenius norm equal numpy products of the STOCK objective's `dlsvec` array (objectivefns.py:4633-4665) -- with and
    without penalty rows -- and leave `objective.probs` / `objective.obj` as the reference's dlsvec leaves them;
  * `GateSetTomography(objfn_builders=hip_objfn_builders()).run(data, simulator=HipMapForwardSimulator())` performs every
    LM iteration through fill_dprobs_dev -> objective_rows_dev -> fill_jtj_dev -> fill_jtf_dev: ZERO (nE, nP) host
    fills, ZERO host materialisations of the Jacobian, and ends at the stock run's estimate;
  * the simulator's options and the builders survive nice-serialization, a `Model` round trip and a GST checkpoint
    (protocols/gst.py:1497-1504);
  * pyGSTi's own 'ep' array is marked for page-locking by the layout and picked up on its first real fill.
"""
import numpy as np
import pytest

pygsti = pytest.importorskip("pygsti")

from _fake_device import FakePlan                                   # noqa: E402
from pygsti_amd import _lib, lmstep                                 # noqa: E402
from pygsti_amd import pygsti_adapter as A                          # noqa: E402


@pytest.fixture()
def fake_device(monkeypatch, oracle_built):
    FakePlan.log = []
    FakePlan.host_jacobian_fills = 0
    lmstep.DeviceJacobian.materialisations = 0
    monkeypatch.setattr(_lib.Plan, "from_circuits", FakePlan.from_circuits)
    monkeypatch.setattr(_lib, "pin_host_array", lambda arr: FakePlan.log.append(("pin_host_array", arr.nbytes)) or True)
    monkeypatch.setattr(_lib, "unpin_host_array", lambda arr: None)
    monkeypatch.setattr(A.HipMapCOPALayout, "PIN_MIN_BYTES", 0)      # (the 1Q Jacobian of these tests is 90 KB)
    return FakePlan


def _setup(max_length=2, seed=11):
    from pygsti.modelpacks import smq1Q_XYI
    edesign = smq1Q_XYI.create_gst_experiment_design(max_length)
    datagen = smq1Q_XYI.target_model().depolarize(op_noise=0.03, spam_noise=0.01)
    ds = pygsti.data.simulate_data(datagen, edesign.all_circuits_needing_data, 1000, seed=seed)
    return smq1Q_XYI, edesign, ds


def _objective_pair(kind, penalties, fake_device, kick=0.02):
    from pygsti.objectivefns import objectivefns as OF
    pack, edesign, ds = _setup(2)
    circuits = list(edesign.all_circuits_needing_data)
    m_ref = pack.target_model().depolarize(op_noise=0.01, spam_noise=0.01)
    m_ref.from_vector(m_ref.to_vector() + kick * np.random.default_rng(5).standard_normal(m_ref.num_params))
    m_hip = m_ref.copy()
    m_ref.sim = pygsti.forwardsims.MapForwardSimulator()
    m_hip.sim = A.HipMapForwardSimulator(derivative_mode="fd")
    stock = {"chi2": OF.Chi2Function, "logl": OF.PoissonPicDeltaLogLFunction}[kind]
    mine = {"chi2": A.HipChi2Function, "logl": A.HipPoissonPicDeltaLogLFunction}[kind]
    reg = {"chi2": {"min_prob_clip_for_weighting": 1e-4}, "logl": {"min_prob_clip": 1e-4, "radius": 1e-4}}[kind]
    o_ref = stock.create_from(m_ref, ds, circuits, regularization=reg, penalties=penalties, method_names=("lsvec", "dlsvec"))
    o_hip = mine.create_from(m_hip, ds, circuits, regularization=reg, penalties=penalties, method_names=("lsvec", "dlsvec"))
    return o_ref, o_hip


@pytest.mark.parametrize("kind,penalties", [("chi2", None), ("logl", None), ("chi2", {"regularize_factor": 1e-3}),
                                            ("logl", {"cptp_penalty_factor": 0.1, "spam_penalty_factor": 0.1})])
def test_device_dlsvec_equals_the_reference_dlsvec(kind, penalties, fake_device):
    o_ref, o_hip = _objective_pair(kind, penalties, fake_device, kick=0.6 if penalties and "regularize_factor" in penalties else 0.02)
    assert isinstance(o_hip.layout, A.HipMapCOPALayout) and o_hip.ex == o_ref.ex
    x = o_ref.model.to_vector()
    f_ref = o_ref.lsvec(x).copy()
    J_ref = np.array(o_ref.dlsvec(x))
    f_hip = o_hip.lsvec(x).copy()
    dj = o_hip.dlsvec(x)
    assert o_hip.last_dlsvec_route == "device" and isinstance(dj, lmstep.DeviceJacobian)
    assert dj.shape == J_ref.shape and fake_device.host_jacobian_fills == 0
    np.testing.assert_allclose(f_hip, f_ref, rtol=0, atol=1e-12 * np.abs(f_ref).max())
    # the arrays dlsvec leaves behind (objective.obj holds lsvec afterwards, objective.probs the clipped probabilities)
    np.testing.assert_allclose(o_hip.obj, o_ref.obj, rtol=0, atol=1e-12 * np.abs(f_ref).max())
    np.testing.assert_allclose(o_hip.probs, o_ref.probs, rtol=0, atol=1e-15)
    # the layout's products, through the reference's own ArraysInterface
    from pygsti.optimize import arraysinterface as ARI
    ari = ARI.DistributedArraysInterface(o_hip.layout, "normal", o_hip.ex)
    jtj = ari.allocate_jtj(); jtf = ari.allocate_jtf()
    ari.fill_jtj(dj, jtj, None); ari.fill_jtf(dj, f_hip, jtf)
    ref_jtj, ref_jtf = J_ref.T @ J_ref, J_ref.T @ f_ref
    tol = 1e-12 if kind == "chi2" else 1e-10
    assert np.abs(jtj - ref_jtj).max() <= tol * np.abs(ref_jtj).max()
    assert np.abs(jtf - ref_jtf).max() <= tol * np.sqrt(np.abs(ref_jtj).max()) * np.linalg.norm(f_ref)
    assert abs(ari.norm2_jac(dj) - np.linalg.norm(J_ref)**2) <= 1e-10 * np.linalg.norm(J_ref)**2
    assert lmstep.DeviceJacobian.materialisations == 0
    # a caller that does want the array gets the reference's, and it is counted
    np.testing.assert_allclose(np.asarray(dj), J_ref, rtol=0, atol=1e-9 * np.abs(J_ref).max())
    assert lmstep.DeviceJacobian.materialisations == 1
    # switched off -> the reference's own host algebra over a Jacobian the (stand-in) device fills
    o_hip.device_lm_step = False
    J_host = np.array(o_hip.dlsvec(x))
    assert o_hip.last_dlsvec_route == "host" and fake_device.host_jacobian_fills >= 1
    np.testing.assert_allclose(J_host, J_ref, rtol=0, atol=1e-9 * np.abs(J_ref).max())
    # ... into pyGSTi's own 'ep' array, which the layout had marked and the fill page-locked
    assert any(c[0] == "pin_host_array" for c in fake_device.log)


def test_gst_run_takes_every_lm_iteration_through_the_device_step(fake_device, tmp_path):
    pack, edesign, ds = _setup(4)
    data = pygsti.protocols.ProtocolData(edesign, ds)
    start = pack.target_model()

    proto_ref = pygsti.protocols.GateSetTomography(start.copy(), gaugeopt_suite=None, verbosity=0)
    res_ref = proto_ref.run(data, simulator=pygsti.forwardsims.MapForwardSimulator(), disable_checkpointing=True)

    proto = pygsti.protocols.GateSetTomography(start.copy(), gaugeopt_suite=None, objfn_builders=A.hip_objfn_builders(), verbosity=0)
    ckpt = str(tmp_path / "ckpt")
    res = proto.run(data, simulator=A.HipMapForwardSimulator(derivative_mode="fd"), checkpoint_path=ckpt)

    names = [c[0] for c in fake_device.log]
    n_steps = names.count("fill_jtj_dev")
    assert n_steps >= 6                                              # chi2 stages L = 1, 2, 4 and the final logL stage
    assert names.count("fill_dprobs_dev") == n_steps and names.count("objective_rows_dev") == n_steps
    assert names.count("fill_jtf_dev") == n_steps
    assert fake_device.host_jacobian_fills == 0, "an (nE, nP) host array was filled during the fit"
    assert "fill_dprobs" not in names
    assert lmstep.DeviceJacobian.materialisations == 0
    assert not any(c[0] == "pin_host_array" for c in fake_device.log)     # the 'ep' array was never touched, so never pinned
    kinds = [c[1] for c in fake_device.log if c[0] == "objective_rows_dev"]
    assert kinds[0] == "chi2" and kinds[-1] == "logl"

    est_ref = res_ref.estimates["GateSetTomography"].models["final iteration estimate"]
    est = res.estimates["GateSetTomography"].models["final iteration estimate"]
    assert np.abs(est.to_vector() - est_ref.to_vector()).max() < 1e-5
    circuits = list(edesign.all_circuits_needing_data)
    est_chk = est.copy(); est_chk.sim = pygsti.forwardsims.MapForwardSimulator()
    l_ref = pygsti.tools.two_delta_logl(est_ref, ds, circuits)
    l_hip = pygsti.tools.two_delta_logl(est_chk, ds, circuits)
    assert abs(l_hip - l_ref) <= 1e-6 * abs(l_ref)

    # the estimate's simulator is ours, options intact; the checkpoint pyGSTi wrote after every stage round-trips them
    assert isinstance(est.sim, A.HipMapForwardSimulator) and est.sim.derivative_mode == "fd"
    from pygsti.protocols.gst import GateSetTomographyCheckpoint
    import glob
    files = sorted(glob.glob(ckpt + "_iteration_*.json"))
    assert files
    chk = GateSetTomographyCheckpoint.read(files[-1])
    assert all(isinstance(m.sim, A.HipMapForwardSimulator) and m.sim.derivative_mode == "fd" for m in chk.mdl_list)
    # the protocol (with its builders) serializes by class path
    back = pygsti.protocols.GateSetTomography.from_nice_serialization(proto.to_nice_serialization()) \
        if hasattr(proto, "to_nice_serialization") else None
    if back is not None:
        assert back.objfn_builders.iteration_builders[0].cls_to_build is A.HipChi2Function


def test_simulator_options_survive_serialization():
    from pygsti.modelpacks import smq1Q_XYI
    from pygsti.forwardsims import ForwardSimulator
    sim = A.HipMapForwardSimulator(derivative_mode="analytic", device=0, lindblad_on_device=False, derivative_eps=3e-7)
    back = ForwardSimulator.from_nice_serialization(sim.to_nice_serialization())
    assert type(back) is A.HipMapForwardSimulator
    assert (back.derivative_mode, back._hip_device, back.lindblad_on_device, back.derivative_eps) == ("analytic", 0, False, 3e-7)
    m = smq1Q_XYI.target_model()
    m.sim = A.HipMapForwardSimulator(derivative_mode="fd")
    m2 = type(m).from_nice_serialization(m.to_nice_serialization())
    assert type(m2.sim) is A.HipMapForwardSimulator and m2.sim.derivative_mode == "fd" and m2.sim.model is m2
    # states written before these keys existed still load (defaults)
    st = sim.to_nice_serialization()
    for k in ("hip_derivative_mode", "hip_device", "hip_lindblad_on_device"):
        st.pop(k)
    assert ForwardSimulator.from_nice_serialization(st).derivative_mode == "auto"
    b = A.hip_objfn_builders("chi2")
    bb = type(b).from_nice_serialization(b.to_nice_serialization())
    assert bb.iteration_builders[0].cls_to_build is A.HipChi2Function and not bb.final_builders


def test_device_step_falls_back_to_the_reference_algebra_when_it_does_not_apply(fake_device):
    """Sparse data (omitted outcomes couple rows through the zero-frequency corrections, objectivefns.py:4616-4624):
    the objective says why and runs the reference's dlsvec over a device-filled host Jacobian."""
    from pygsti.objectivefns import objectivefns as OF
    pack, edesign, ds_full = _setup(1)
    circuits = list(edesign.all_circuits_needing_data)
    ds = pygsti.data.DataSet(outcome_labels=["0", "1"])
    for i, c in enumerate(circuits):
        row = ds_full[c]
        counts = {"0": row["0"] + row["1"]} if i % 3 == 0 else {"0": row["0"], "1": row["1"]}     # every third circuit: '1' never seen
        ds.add_count_dict(c, counts)
    ds.done_adding_data()
    m = pack.target_model().depolarize(op_noise=0.01, spam_noise=0.01)
    m_ref = m.copy(); m_ref.sim = pygsti.forwardsims.MapForwardSimulator()
    m.sim = A.HipMapForwardSimulator(derivative_mode="fd")
    reg = {"min_prob_clip": 1e-4, "radius": 1e-4}
    o = A.HipPoissonPicDeltaLogLFunction.create_from(m, ds, circuits, regularization=reg, method_names=("lsvec", "dlsvec"))
    o_ref = OF.PoissonPicDeltaLogLFunction.create_from(m_ref, ds, circuits, regularization=reg, method_names=("lsvec", "dlsvec"))
    assert o.firsts is not None
    J = np.array(o.dlsvec())
    assert o.last_dlsvec_route == "host" and "omitted" in o.last_dlsvec_blocker
    np.testing.assert_allclose(J, np.array(o_ref.dlsvec()), rtol=0, atol=1e-9 * np.abs(J).max())


def test_device_dlsvec_with_several_atoms(fake_device):
    """num_atoms = 3 in one process (the reference's single-process shortcut, distforwardsim.py:98-99): one plan and one
    device block per atom, J^T J / J^T f summed over them -- equal to the stock objective's products on the same layout."""
    from pygsti.objectivefns import objectivefns as OF
    from pygsti.optimize import arraysinterface as ARI
    pack, edesign, ds = _setup(4)
    circuits = list(edesign.all_circuits_needing_data)
    m_ref = pack.target_model().depolarize(op_noise=0.02, spam_noise=0.01)
    m_hip = m_ref.copy()
    m_ref.sim = pygsti.forwardsims.MapForwardSimulator(num_atoms=3)
    m_hip.sim = A.HipMapForwardSimulator(num_atoms=3, derivative_mode="fd")
    reg = {"min_prob_clip": 1e-4, "radius": 1e-4}
    o_ref = OF.PoissonPicDeltaLogLFunction.create_from(m_ref, ds, circuits, regularization=reg, method_names=("lsvec", "dlsvec"))
    o_hip = A.HipPoissonPicDeltaLogLFunction.create_from(m_hip, ds, circuits, regularization=reg, method_names=("lsvec", "dlsvec"))
    assert len(o_hip.layout.atoms) == 3
    x = m_ref.to_vector()
    f_ref = o_ref.lsvec(x).copy(); J_ref = np.array(o_ref.dlsvec(x))
    f = o_hip.lsvec(x).copy(); dj = o_hip.dlsvec(x)
    assert o_hip.last_dlsvec_route == "device" and len(dj.parts) == 3
    # (the two layouts order their elements identically: same circuits, same atom count)
    ari = ARI.DistributedArraysInterface(o_hip.layout, "normal", o_hip.ex)
    jtj = ari.allocate_jtj(); jtf = ari.allocate_jtf()
    ari.fill_jtj(dj, jtj, None); ari.fill_jtf(dj, f, jtf)
    ref_jtj, ref_jtf = J_ref.T @ J_ref, J_ref.T @ f_ref
    assert np.abs(jtj - ref_jtj).max() <= 1e-10 * np.abs(ref_jtj).max()
    assert np.abs(jtf - ref_jtf).max() <= 1e-10 * np.sqrt(np.abs(ref_jtj).max()) * np.linalg.norm(f_ref)
    assert np.abs(np.asarray(dj) - J_ref).max() <= 1e-9 * np.abs(J_ref).max()
    names = [c[0] for c in fake_device.log]
    assert names.count("fill_dprobs_dev") == 3 and names.count("fill_jtj_dev") == 3


def test_custom_lm_optimizer_takes_the_device_step_too(fake_device):
    """pyGSTi's other Levenberg-Marquardt implementation (optimize/customlm.py:556-610) uses the Jacobian the same three ways
    (shape / nbytes for its log line, ari.norm2_jac, ari.fill_jtj, ari.fill_jtf): a fit through it runs on the device handle."""
    from pygsti.optimize.customlm import CustomLMOptimizer
    pack, edesign, ds = _setup(2)
    data = pygsti.protocols.ProtocolData(edesign, ds)
    opt = CustomLMOptimizer(maxiter=20, tol=1e-6)
    proto = pygsti.protocols.GateSetTomography(pack.target_model(), gaugeopt_suite=None, objfn_builders=A.hip_objfn_builders("chi2"),
                                               optimizer=opt, verbosity=0)
    res = proto.run(data, simulator=A.HipMapForwardSimulator(derivative_mode="fd"), disable_checkpointing=True)
    names = [c[0] for c in fake_device.log]
    assert names.count("fill_jtj_dev") >= 2 and fake_device.host_jacobian_fills == 0 and lmstep.DeviceJacobian.materialisations == 0
    ref = pygsti.protocols.GateSetTomography(pack.target_model(), gaugeopt_suite=None, objfn_builders=pygsti.protocols.GSTObjFnBuilders.create_from("chi2"),
                                             optimizer=CustomLMOptimizer(maxiter=20, tol=1e-6), verbosity=0).run(
        data, simulator=pygsti.forwardsims.MapForwardSimulator(), disable_checkpointing=True)
    a = res.estimates["GateSetTomography"].models["final iteration estimate"].to_vector()
    b = ref.estimates["GateSetTomography"].models["final iteration estimate"].to_vector()
    assert np.abs(a - b).max() < 1e-5
