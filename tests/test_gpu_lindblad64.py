"""SURVEY 8(f) row f4 at three qubits: Lindblad members of FULL dimension 64 built and differentiated ON THE DEVICE
(gst_set_lindblad at D = 64: lindblad64_assemble / _exp / _deriv kernels -- the 4,032-term generator sum, the scaled-Taylor
exponential and its Frechet derivatives as 64 x 64 products on the matrix cores) against a three-qubit explicit CPTPLND model
of the reference (tests/golden/make_golden_r6.py): dense members 1e-12, probabilities 1e-10 (Map simulator through the real
composed reps), exact Jacobian columns 1e-8 vs MatrixForwardSimulator (matrixforwardsim.py:1047-1140), which itself paid 226 s
per 8-column slice for `ExpErrorgenOp.deriv_wrt_params()`."""
import numpy as np
import pytest

from conftest import load_fixture, plan_from_fixture, matrix_rows_by_circuit
from pygsti_amd import lindblad as LB

pytestmark = pytest.mark.gpu

NAME = "3q_explicit_CPTPLND"


@pytest.fixture(scope="module")
def case():
    from pygsti_amd import _lib
    fx, lb = load_fixture(NAME), load_fixture("lindblad_" + NAME)
    model = LB.LindbladModel.from_fixture(lb, 3)
    pl = plan_from_fixture({**fx, "pkind": -np.ones(int(fx["nP"]), np.int32), "pobj": np.zeros(int(fx["nP"]), np.int32),
                            "pelem": np.zeros(int(fx["nP"]), np.int32)})
    pl.set_lindblad(model)
    pl.set_lindblad_params(lb["paramvec"])
    return fx, lb, model, pl, _lib


def test_device_built_three_qubit_members_match_the_reference(case):
    fx, lb, model, pl, _lib = case
    G, R, E = pl.get_model()
    assert np.abs(G - fx["gates"]).max() < 1e-12, np.abs(G - fx["gates"]).max()
    assert np.abs(R - fx["rhos"]).max() < 1e-12 and np.abs(E - fx["effects"]).max() < 1e-12
    # a second parameter vector: larger generators, more squarings
    th2 = lb["paramvec"] * 3.0
    pl.set_lindblad_params(th2)
    G2, R2, E2 = pl.get_model()
    Gh, Rh, Eh = model.dense(th2)
    scale = max(1.0, np.abs(Gh).max())
    assert np.abs(G2 - Gh).max() < 1e-11 * scale and np.abs(R2 - Rh).max() < 1e-11 * scale and np.abs(E2 - Eh).max() < 1e-11 * scale
    pl.set_lindblad_params(lb["paramvec"])


def test_probabilities_and_exact_jacobian_of_a_three_qubit_cptplnd_model(case):
    fx, lb, model, pl, _lib = case
    p = pl.fill_probs()
    assert np.abs(p - fx["probs"]).max() < 1e-10, np.abs(p - fx["probs"]).max()
    cols = fx["matrix_cols"]
    Ja = pl.fill_dprobs(param_idx=cols, mode=_lib.DERIV_ANALYTIC)
    rows = matrix_rows_by_circuit(fx)
    ref = fx["matrix_by_circuit_dprobs"][rows]
    err = np.abs(Ja - ref).max()
    assert err < 1e-8, (err, np.abs(ref).max())
    assert np.abs(ref).max() > 1e-3                                    # (the columns are not trivially zero)
    # every block of columns on its own: preparation, POVM, gate Hamiltonian / Cholesky blocks, the two-qubit gate
    for k in range(0, len(cols), 8):
        assert np.abs(ref[:, k:k + 8]).max() > 1e-6, k
    # against the reference's own finite differences (Map simulator), to FD accuracy
    assert np.abs(Ja - fx["dprobs_map"]).max() < 5e-5 * max(1.0, np.abs(ref).max())
    # a window of the caller's array, shuffled columns
    win = np.full((int(fx["nE"]), 7), np.nan)
    pick = np.array([3, 17, 30]); dest = np.array([5, 0, 2])
    pl.fill_dprobs(out=win, param_idx=cols[pick], dest_idx=dest, mode=_lib.DERIV_ANALYTIC)
    assert np.array_equal(win[:, dest], Ja[:, pick]) and np.isnan(win[:, [1, 3, 4, 6]]).all()


def test_three_qubit_lindblad_routes_that_are_not_offered(case):
    fx, lb, model, pl, _lib = case
    with pytest.raises(_lib.GstUnsupported):
        pl.fill_dprobs(param_idx=fx["matrix_cols"][:4], eps=1e-7)         # FD over device-built members: exact derivatives only
    with pytest.raises(_lib.GstUnsupported):
        pl.lindblad_model_sets(fx["matrix_cols"][:2], 1e-7)
