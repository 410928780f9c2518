"""The reference's own numerical pins for this path, re-stated against the device simulators:
 * test/unit/objects/test_forwardsim.py:278-348 -- cross-simulator consistency on smq1Q_XYI depolarized
   (op 0.05, spam 0.025), lsgst circuits max_length 4: colinearity of stacked probs >= 1 - 1e-14 (PROBS_TOL) and of
   stacked Jacobians >= 1 - 1e-10 (JACS_TOL) between the Map-like (FD) and Matrix-like (analytic) simulators;
 * test/unit/objects/test_model.py:419-486 -- known-answer probabilities E^T G G rho to 7 places;
 * test/unit/objects/test_forwardsim.py:148-163 -- zero-parameter requests give (nE, 0) / (nE, 0, 0) arrays."""
import numpy as np
import pytest

from pygsti_amd import modelpacks as MP
from pygsti_amd.forwardsim import HipMapForwardSimulator

pytestmark = pytest.mark.gpu


def _colinearity(a, b):
    a, b = a.ravel(), b.ravel()
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))


def test_cross_simulator_consistency():
    model = MP.smq1Q_XYI.target_model().depolarize(op_noise=0.05, spam_noise=0.025)
    circuits = MP.smq1Q_XYI.create_gst_circuits(4)
    res = {}
    for mode in ("fd", "analytic"):
        sim = HipMapForwardSimulator(derivative_mode=mode); model.sim = sim
        lay = sim.create_layout(circuits, array_types=("e", "ep"))
        p = np.empty(lay.num_elements); J = np.empty((lay.num_elements, model.num_params))
        sim.bulk_fill_dprobs(J, lay, pr_array_to_fill=p)
        res[mode] = (p, J)
    assert 1 - _colinearity(res["fd"][0], res["analytic"][0]) < 1e-14
    assert 1 - _colinearity(res["fd"][1], res["analytic"][1]) < 1e-10


def test_known_answer_probabilities():
    model = MP.smq1Q_XYI.target_model().depolarize(op_noise=0.01, spam_noise=0.001)
    model.sim = HipMapForwardSimulator()
    Gx, Gy = model.operations["Gxpi2:0"], model.operations["Gypi2:0"]
    rho = model.preps["rho0"]; E0, E1 = model.povms["Mdefault"]["0"], model.povms["Mdefault"]["1"]
    for circ, prod in ((("Gxpi2:0", "Gypi2:0"), Gy @ Gx), (("Gxpi2:0", "Gxpi2:0"), Gx @ Gx), ((), np.eye(4))):
        pr = model.sim.probs(circ)
        assert abs(pr[("0",)] - float(E0 @ prod @ rho)) < 1e-7 and abs(pr[("1",)] - float(E1 @ prod @ rho)) < 1e-7
    bulk = model.sim.bulk_probs([("Gxpi2:0",), ("Gxpi2:0", "Gypi2:0")])
    assert abs(bulk[("Gxpi2:0", "Gypi2:0")][("0",)] - float(E0 @ Gy @ Gx @ rho)) < 1e-7


def test_zero_parameter_requests_give_empty_derivative_arrays():
    model = MP.smq1Q_XYI.target_model(); sim = HipMapForwardSimulator(); model.sim = sim
    lay = sim.create_layout(MP.smq1Q_XYI.create_gst_circuits(1))
    atom = lay.atoms[0]
    J = np.empty((lay.num_elements, 0)); sim._bulk_fill_dprobs_atom(J, None, atom, slice(0, 0))
    H = np.empty((lay.num_elements, 0, 0)); sim._bulk_fill_hprobs_atom(H, None, None, atom, slice(0, 0), slice(0, 0))
    assert J.shape == (lay.num_elements, 0) and H.shape == (lay.num_elements, 0, 0)
