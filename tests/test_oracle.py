"""Pins the CPU oracle (oracle/) against every golden vector the reference produced
(tests/golden/*.npz from tests/golden/make_golden.py): BITWISE for probs, FD dprobs, FD-of-FD hprobs,
for both back ends (our C restatement and, where built, the reference's own C++ reps)."""
import os

import numpy as np
import pytest

from conftest import load_fixture, assert_bitwise, ROOT

DESIGN_FIXTURES = ["smq1Q_XYI_L4_depol", "smq1Q_XYI_L4_kick", "smq1Q_XYI_L128_depol",
                   "smq2Q_XYICNOT_L2_depol", "smq2Q_XYICNOT_L1024_deep"]
# + two preparations, two POVMs (2 and 3 effects), explicit SPAM labels in the circuits, an empty gate string
# + the 3-qubit explicit model (D = 64, 10 gates, 41,536 parameters; BASELINE configs[4]): 207 seeded random circuits
FIXTURES = DESIGN_FIXTURES + ["smq1Q_multispam_L2", "3q_explicit_L64", "3q_explicit_matrix"]
HAVE_REF = os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libgst_ref.so")) or \
    os.path.isdir("/root/reference/pygsti/evotypes/densitymx")
KINDS = ["port"] + (["reference"] if HAVE_REF else [])


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("name", FIXTURES)
def test_probs_and_dprobs_bitwise(oracle_built, name, kind):
    fx = load_fixture(name)
    orc = oracle_built.from_fixture(fx, kind)
    assert_bitwise(orc.probs(), fx["probs"], "probs")
    J, pr = orc.dprobs(fx["dprobs_cols"], eps=float(fx["derivative_eps"]), return_probs=True)
    assert_bitwise(pr, fx["probs"], "probs from dprobs")
    assert_bitwise(J, fx["dprobs_map"], "dprobs")


@pytest.mark.parametrize("name", ["smq1Q_XYI_L4_depol", "smq1Q_multispam_L2", "3q_explicit_L64"])
def test_hprobs_bitwise(oracle_built, name):
    fx = load_fixture(name)
    H = oracle_built.from_fixture(fx).hprobs(fx["hprobs_rows"], fx["hprobs_cols"], eps=float(fx["hessian_eps"]))
    assert_bitwise(H, fx["hprobs_map"], "hprobs")


@pytest.mark.parametrize("name", ["smq1Q_XYI_L4_depol", "smq1Q_XYI_L4_kick", "smq1Q_XYI_L128_depol",
                                  "smq2Q_XYICNOT_L2_depol", "smq1Q_multispam_L2"])
def test_numpy_analytic_jacobian_matches_matrix_simulator(oracle_built, name):
    """The independent forward/backward analytic Jacobian agrees with MatrixForwardSimulator's golden
    vectors to 1e-10 (both exact derivatives); the FD Map Jacobian does NOT (SURVEY finding 3)."""
    fx = load_fixture(name)
    J, P = oracle_built.analytic_dprobs(fx, fx["dprobs_cols"])
    rows = fx["matrix_rows"]
    assert np.abs(P[rows] - fx["probs_matrix"]).max() < 1e-12
    assert np.abs(J[rows] - fx["dprobs_matrix"]).max() < 1e-10
    assert np.abs(P - fx["probs"]).max() < 1e-12


@pytest.mark.parametrize("name", DESIGN_FIXTURES)
def test_prefix_table_restatement_is_identical_to_reference_table(name):
    from oracle import prefix_table as PT
    fx = load_fixture(name)
    t = PT.build_table(fx["circ_ptr"], fx["circ_gates"], len(fx["outcome_names"]))
    for k in ("t_dest", "t_start", "t_cache", "t_rho", "row_ptr", "gate_idx"):
        assert np.array_equal(t[k], fx[k]), k
    assert t["cache_size"] == int(fx["cache_size"])


def test_fd_hessian_is_loose_against_analytic():
    """Documented in SURVEY App. C: FD-of-FD (eps 1e-5) only agrees with the analytic Hessian loosely."""
    fx = load_fixture("smq1Q_XYI_L4_depol")
    rows = fx["matrix_rows"]
    d = np.abs(fx["hprobs_map"][rows] - fx["hprobs_matrix"]).max()
    assert 1e-9 < d < 1e-2


@pytest.mark.parametrize("name", ["smq1Q_XYI_L4_depol", "smq1Q_multispam_L2"])
def test_analytic_hprobs_oracle_vs_matrix_simulator(name):
    """The numpy exact Hessian (derivative forward/backward states) against MatrixForwardSimulator's hprobs vectors."""
    from oracle import oracle as O
    fx = load_fixture(name)
    H = O.analytic_hprobs(fx, fx["hprobs_rows"], fx["hprobs_cols"])
    assert np.abs(H[fx["matrix_rows"]] - fx["hprobs_matrix"]).max() < 1e-11
    # and it is the quantity the FD-of-FD Map path approximates
    assert np.abs(H - fx["hprobs_map"]).max() < 1e-2


def test_analytic_hprobs_oracle_vs_matrix_simulator_2q_blocks():
    from oracle import oracle as O
    fx = load_fixture("smq2Q_XYICNOT_L2_depol")
    rows = fx["matrix_rows"][::5]
    # (the numpy Hessian is slow: restrict the fixture to a sample of circuits by masking the effect CSR)
    for b in (0, 2):
        i1, i2 = fx["mh%d_idx1" % b][:6], fx["mh%d_idx2" % b][::4]
        H = O.analytic_hprobs(fx, i1, i2)
        ref = fx["mh%d_hprobs" % b][::5][:, :6][:, :, ::4]
        assert np.abs(H[rows] - ref).max() < 1e-11


@pytest.mark.parametrize("name,ptol,jtol", [("smq1Q_XYI_L4_CPTPLND", 1e-15, 1e-8), ("smq2Q_XYICNOT_L1_CPTPLND", 1e-15, 1e-8)])
def test_dense_model_sets_reproduce_the_map_fd_of_cptplnd_models(oracle_built, name, ptol, jtol):
    """General parameterisations: the dense model after every FD step (fixture `mm_*`, written by the reference's own
    set_parameter_value) walked by the oracle gives the reference Map simulator's Jacobian to rounding -- the reference
    propagates composed / exponentiated members factor by factor, so the agreement is ~1e-16 / eps, not bitwise (observed: probs 3e-16, dprobs 4.4e-9 / 6.1e-9; bar: the north star's 1e-8)."""
    fx = load_fixture(name)
    eps = float(fx["derivative_eps"])
    orc = oracle_built.from_fixture({k: np.array(v) for k, v in fx.items()})      # (set_model writes into the oracle's arrays)
    base = orc.probs()
    assert np.abs(base - fx["probs"]).max() <= ptol
    J = np.empty_like(fx["dprobs_map"])
    for c, (g, r, e) in enumerate(zip(fx["mm_gates"], fx["mm_rhos"], fx["mm_effects"])):
        orc.set_model(g, r, e)
        J[:, c] = (orc.probs() - base) / eps
    assert np.abs(J - fx["dprobs_map"]).max() <= jtol, np.abs(J - fx["dprobs_map"]).max()


def test_two_level_model_sets_reproduce_the_map_fd_of_fd_hessian(oracle_built):
    """FD-of-FD Hessian block of the CPTPLND model composed from two-level dense model sets on the oracle vs the
    reference Map simulator's block: rounding amplified by 1 / eps^2 = 1e10 (bound 2e-5 absolute)."""
    fx = load_fixture("smq1Q_XYI_L4_CPTPLND")
    eps = float(fx["hessian_eps"])
    orc = oracle_built.from_fixture({k: np.array(v) for k, v in fx.items()})
    d = []
    for k in range(fx["mm2_gates"].shape[0]):
        orc.set_model(fx["mm2_gates"][k, 0], fx["mm2_rhos"][k, 0], fx["mm2_effects"][k, 0]); base = orc.probs()
        cols = []
        for c in range(1, fx["mm2_gates"].shape[1]):
            orc.set_model(fx["mm2_gates"][k, c], fx["mm2_rhos"][k, c], fx["mm2_effects"][k, c])
            cols.append((orc.probs() - base) / eps)
        d.append(np.array(cols).T)
    H = np.stack([(d[k] - d[0]) / eps for k in range(1, len(d))], axis=1)
    rows = [list(fx["hprobs_rows"]).index(r) for r in fx["mm2_rows"]]
    cols = [list(fx["hprobs_cols"]).index(c) for c in fx["mm2_cols"]]
    ref = fx["hprobs_map"][:, rows][:, :, cols]
    assert np.abs(H - ref).max() <= 2e-5, np.abs(H - ref).max()       # observed 6.7e-6 (max|H| = 2.0)


def test_numpy_analytic_oracle_vs_matrix_simulator_at_d64():
    """The direct analytic reference at D = 64 (round 4): two 16-column Jacobian blocks and two Hessian blocks taken
    straight from MatrixForwardSimulator's per-atom seams on the 3-qubit model (tests/golden/make_golden_r4.py,
    matrixforwardsim.py:1047-1287) pin the numpy forward/backward restatement there as well (it was pinned at D = 4 and
    16 only, and the device's D = 64 analytic kernels were compared with it alone)."""
    from conftest import matrix_rows_by_circuit
    from oracle import oracle as O
    fx = load_fixture("3q_explicit_matrix")
    rows = matrix_rows_by_circuit(fx)
    J, P = O.analytic_dprobs(fx, fx["matrix_cols"])
    assert np.abs(P - fx["matrix_by_circuit_probs"][rows]).max() < 1e-13
    assert np.abs(J - fx["matrix_by_circuit_dprobs"][rows]).max() < 1e-11
    assert np.abs(fx["matrix_by_circuit_dprobs"]).max() > 0.1
    for b in (0, 1):
        H = O.analytic_hprobs(fx, fx["mh%d_idx1" % b], fx["mh%d_idx2" % b])
        ref = fx["mh%d_by_circuit_hprobs" % b][rows]
        assert np.abs(H - ref).max() < 1e-11, (b, np.abs(H - ref).max())
        assert np.abs(ref).max() > 1e-3


@pytest.mark.parametrize("name,jtol", [("smq1Q_XYI_L128_CPTPLND", 1e-10), ("smq2Q_XYICNOT_L1024_CPTPLND_deep", None)])
def test_deep_cptplnd_fixtures(oracle_built, name, jtol):
    """CPTPLND models at depth (round 4): (i) the numpy analytic Jacobian x the members' reference derivative matrices
    equals the Matrix simulator's columns (1Q L<=128, every column); (ii) the reference's own two simulators differ by
    the FD truncation error, which grows with depth (9e-3 at depth 1,030 here) -- parity is per mode; (iii) the dense
    model sets of the 2Q fixture walked by the oracle reproduce the Map simulator's FD columns to 1e-8 at EVERY depth
    (dense products vs factor-by-factor composed reps: 4.4e-9 observed, flat in depth)."""
    from conftest import matrix_rows_by_circuit, element_depth
    from oracle import oracle as O
    fx = load_fixture(name)
    rows = matrix_rows_by_circuit(fx)
    Jm = fx["matrix_by_circuit_dprobs"][rows]
    assert np.abs(fx["matrix_by_circuit_probs"][rows] - fx["probs"]).max() < 1e-12
    d = np.abs(Jm - fx["dprobs_map"])
    depth = element_depth(fx)
    assert d[depth <= 4].max() < 1e-4 and d.max() > 1e-4          # FD truncation: small when shallow, not at depth
    if jtol is not None:
        J, P = O.analytic_dprobs_general(fx)
        assert np.abs(J - Jm).max() < jtol, np.abs(J - Jm).max()
    else:
        eps = float(fx["derivative_eps"])
        orc = oracle_built.from_fixture({k: np.array(v) for k, v in fx.items()})
        base = orc.probs()
        J = np.empty_like(fx["dprobs_map"])
        for c, (g, r, e) in enumerate(zip(fx["mm_gates"], fx["mm_rhos"], fx["mm_effects"])):
            orc.set_model(g, r, e)
            J[:, c] = (orc.probs() - base) / eps
        err = np.abs(J - fx["dprobs_map"])
        assert err.max() <= 1e-8, err.max()
        assert depth.max() >= 1030


def test_shipped_reference_build_is_the_recorded_one(oracle_built, tmp_path, monkeypatch):
    """oracle/_ref cannot be rebuilt where /root/reference does not exist (the GPU box): a "reference"-kind checker only runs
    on the binary whose hash the build container recorded (oracle/ref_build.sha256), and refuses anything else."""
    import shutil
    O = oracle_built
    if not os.path.exists(O.REF_SO):
        pytest.skip("oracle/_ref is not built here")
    monkeypatch.setattr(O, "_ref_verified", None)
    assert O.verify_ref() is True
    fake = tmp_path / "libgst_ref.so"
    shutil.copy(O.REF_SO, fake)
    with open(fake, "ab") as f:
        f.write(b"\0")
    monkeypatch.setattr(O, "REF_SO", str(fake))
    monkeypatch.setattr(O, "_ref_verified", None)
    with pytest.raises(RuntimeError, match="not the build recorded"):
        O.verify_ref()
